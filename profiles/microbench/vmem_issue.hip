// Microbenchmark: cycles per vector-memory WAVE INSTRUCTION per CU, by width and by the number of active lanes, for loads and stores
// that hit the L2 (each workgroup walks its own 64 KB window).  One 1024-thread workgroup per CU, every wave issues `iters`
// instructions back to back (loads: summed into a register; stores: fire and forget).
//   hipcc -O3 --offload-arch=gfx950 -o vmem_issue vmem_issue.hip && ./vmem_issue
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s\n", hipGetErrorString(e_)); return 1; } } while (0)
typedef unsigned u2 __attribute__((ext_vector_type(2)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;

// MODE: 0 load b16, 1 load b32, 2 load b64, 3 load b128, 4 store b16, 5 store b32, 6 store b64, 7 store b128
// ACT: active lanes per wave (64, 32: every second quad ... ) -- lanes [0, ACT) active
template <int MODE, int ACT>
__global__ __launch_bounds__(1024) void k(char* __restrict__ buf, unsigned* __restrict__ out, int iters, unsigned win_bytes) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  constexpr int W = MODE % 4 == 0 ? 2 : MODE % 4 == 1 ? 4 : MODE % 4 == 2 ? 8 : 16;  // bytes per lane
  const rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(buf + (size_t)blockIdx.x * win_bytes, 0, (int)win_bytes, 0x00020000);
  const unsigned mask = win_bytes - 1;
  unsigned acc = 0;
  const bool on = lane < ACT;
  // a wave's instruction covers one contiguous run of 64 * W bytes (a row); consecutive instructions walk rows 1088 bytes apart
  for (int it = 0; it < iters; ++it) {
    const unsigned row = (unsigned)(it * 16 + wave);
    const unsigned off = on ? ((row * 1088u * (W >= 8 ? 4u : 1u) + (unsigned)lane * W) & mask & ~(unsigned)(W - 1)) : 0x80000000u;
    if (MODE == 0) acc += __builtin_amdgcn_raw_buffer_load_b16(r, off, 0, 0);
    if (MODE == 1) acc += __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0);
    if (MODE == 2) { const u2 v = __builtin_bit_cast(u2, __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0)); acc += v.x ^ v.y; }
    if (MODE == 3) { const u4 v = __builtin_bit_cast(u4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0)); acc += v.x ^ v.y ^ v.z ^ v.w; }
    if (MODE == 4) __builtin_amdgcn_raw_buffer_store_b16((unsigned short)(acc + it), r, off, 0, 0);
    if (MODE == 5) __builtin_amdgcn_raw_buffer_store_b32(acc + it, r, off, 0, 0);
    if (MODE == 6) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(__attribute__((__vector_size__(2 * sizeof(unsigned)))) unsigned, (u2){acc + it, acc}), r, off, 0, 0);
    if (MODE == 7) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, (u4){acc + it, acc, acc, acc}), r, off, 0, 0);
  }
  if (acc == 0x12345678u) out[0] = acc;
}

template <int MODE, int ACT>
int run(char* buf, unsigned* out, const char* what) {
  const int cus = 256, iters = 2048;
  const unsigned win = 1u << 16;  // 32 workgroups x 64 KB = 2 MB per XCD: L2-resident
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  hipLaunchKernelGGL((k<MODE, ACT>), dim3(cus), dim3(1024), 0, 0, buf, out, iters, win);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  hipLaunchKernelGGL((k<MODE, ACT>), dim3(cus), dim3(1024), 0, 0, buf, out, iters, win);
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  const double instr_per_cu = 16.0 * iters;
  constexpr int W = MODE % 4 == 0 ? 2 : MODE % 4 == 1 ? 4 : MODE % 4 == 2 ? 8 : 16;
  printf("%-28s active %2d  %7.1f us  %6.1f ns per wave instruction per CU (%5.1f cycles at 2.4 GHz)  %6.2f TB/s\n", what, ACT, ms * 1e3,
         ms * 1e6 / instr_per_cu, ms * 1e6 / instr_per_cu * 2.4, 256.0 * instr_per_cu * ACT * W / (ms * 1e-3) / 1e12);
  return 0;
}

int main() {
  char* buf; unsigned* out;
  CK(hipMalloc(&buf, (size_t)256 << 20)); CK(hipMalloc(&out, 64));
  CK(hipMemset(buf, 1, (size_t)256 << 20));
  run<0, 64>(buf, out, "load  b16"); run<1, 64>(buf, out, "load  b32"); run<2, 64>(buf, out, "load  b64"); run<3, 64>(buf, out, "load  b128");
  run<1, 32>(buf, out, "load  b32"); run<1, 16>(buf, out, "load  b32"); run<1, 8>(buf, out, "load  b32"); run<2, 32>(buf, out, "load  b64");
  run<4, 64>(buf, out, "store b16"); run<5, 64>(buf, out, "store b32"); run<6, 64>(buf, out, "store b64"); run<7, 64>(buf, out, "store b128");
  run<4, 32>(buf, out, "store b16"); run<5, 32>(buf, out, "store b32"); run<5, 16>(buf, out, "store b32");
  return 0;
}
