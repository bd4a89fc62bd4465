// Microbenchmark: bytes per clock per CU the vector-memory path delivers for L2-resident loads, by access
// width and lane layout.  One 1024-thread workgroup per CU re-reads a 2 MB window (fits the XCD's L2) many times.
//   hipcc -O3 --offload-arch=gfx950 -o ta_rate ta_rate.hip && ./ta_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s\n", hipGetErrorString(e_)); return 1; } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

// MODE 0: dword, lanes contiguous (256 B per wave instruction)
// MODE 1: dwordx2, lanes contiguous (512 B)
// MODE 2: dwordx4, lanes contiguous (1 KB)
// MODE 3: dwordx4, lane = 4q+s: quad q of plane s (4 planes x 256 B)
// MODE 4: dwordx4, lane = 16s+q: 16 lanes contiguous per plane (4 planes x 256 B)
// MODE 5: dword, two rows of 32 pixels (2 x 128 B), the tile kernels' pattern
template <int MODE>
__global__ __launch_bounds__(1024) void k(const float* __restrict__ src, float* __restrict__ out, int iters, unsigned win_floats) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const float* base = src + (size_t)blockIdx.x * win_floats;  // per-workgroup window (power of two floats)
  const unsigned mask = win_floats - 1, plane = win_floats / 4, pmask = plane - 1;
  float acc = 0.f;
#pragma unroll 4
  for (int it = 0; it < iters; ++it) {
    const unsigned step = (unsigned)it * 16 + wave;  // each wave walks its share of the window
    if (MODE == 0) {
      acc += base[(step * 64 + lane) & mask];
    } else if (MODE == 1) {
      const f2 v = *(const f2*)(base + (((step * 64 + lane) * 2) & mask));
      acc += v.x + v.y;
    } else if (MODE == 2) {
      const f4 v = *(const f4*)(base + (((step * 64 + lane) * 4) & mask));
      acc += v.x + v.y + v.z + v.w;
    } else if (MODE == 3) {
      const int q = lane >> 2, s = lane & 3;
      const f4 v = *(const f4*)(base + s * plane + (((step * 16 + q) * 4) & pmask));
      acc += v.x + v.y + v.z + v.w;
    } else if (MODE == 4) {
      const int q = lane & 15, s = lane >> 4;
      const f4 v = *(const f4*)(base + s * plane + (((step * 16 + q) * 4) & pmask));
      acc += v.x + v.y + v.z + v.w;
    } else if (MODE == 5) {
      const int row = lane >> 5, x = lane & 31;
      acc += base[((step * 2 + row) * 544 + x) & mask];
    } else if (MODE == 6) {  // dwordx2, 2 rows x 32 lanes x 2 px
      const int row = lane >> 5, x = lane & 31;
      const f2 v = *(const f2*)(base + (((step * 2 + row) * 544 + 2 * x) & mask));
      acc += v.x + v.y;
    } else if (MODE == 8) {  // dwordx2 at an ODD pixel offset (4-byte aligned only): the shifted gathers of a 2-px-per-lane kernel
      const int row = lane >> 5, x = lane & 31;
      f2 v;
      __builtin_memcpy(&v, base + ((((step * 2 + row) * 544 + 2 * x) & mask) + 27), 8);
      acc += v.x + v.y;
    } else {                 // dwordx4, lane = 8s+q within 32, two rows: 8-lane runs per plane
      const int q = lane & 7, s = (lane >> 3) & 3, row = lane >> 5;
      const f4 v = *(const f4*)(base + s * plane + ((((step * 2 + row) * 8 + q) * 4) & pmask));
      acc += v.x + v.y + v.z + v.w;
    }
  }
  if (acc == 12345.678f) out[0] = acc;
}

template <int MODE>
int run(const float* src, float* out, int bytes_per_lane, const char* name) {
  const int iters = 4096;
  const size_t win = 512 * 1024 / 4;  // 512 KB per workgroup: 32 per XCD = 16 MB ... keep it in L2: 128 KB x 32 = 4 MB
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const unsigned w = 128 * 1024 / 4;
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(1024), 0, 0, src, out, 64, w);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(1024), 0, 0, src, out, iters, w);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double bytes = 256.0 * 1024 * iters * bytes_per_lane;
  printf("%-44s %8.1f us  %7.2f TB/s  %6.1f B/clk/CU @2.4GHz  %5.1f clk per wave-instr\n", name, ms * 1e3, bytes / ms / 1e9,
         bytes / 256 / (ms * 1e-3 * 2.4e9), ms * 1e-3 * 2.4e9 / (iters * 16.0));
  (void)win;
  return 0;
}

int main() {
  float *src, *out;
  const size_t n = 256 * (128 * 1024 / 4) + 4096;
  CK(hipMalloc(&src, n * 4)); CK(hipMalloc(&out, 64));
  CK(hipMemset(src, 0, n * 4));
  run<0>(src, out, 4, "dword, contiguous lanes");
  run<5>(src, out, 4, "dword, 2 rows x 32 px");
  run<1>(src, out, 8, "dwordx2, contiguous lanes");
  run<2>(src, out, 16, "dwordx4, contiguous lanes");
  run<3>(src, out, 16, "dwordx4, lane = 4q+s (4 planes interleaved)");
  run<4>(src, out, 16, "dwordx4, lane = 16s+q (4 planes, 16-lane runs)");
  run<7>(src, out, 16, "dwordx4, lane = 32r+8s+q (8-lane runs)");
  run<6>(src, out, 8, "dwordx2, 2 rows x 32 lanes");
  run<8>(src, out, 8, "dwordx2, 2 rows x 32 lanes, odd pixel offset");
  return 0;
}
