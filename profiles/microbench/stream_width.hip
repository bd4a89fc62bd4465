// Microbenchmark: achievable HBM bandwidth of the epilogue's access pattern ([B,K,H,W] planes: read target,
// weight (f32), mask (u8); write affs, g (f32)) as a function of the per-lane access width.
//   hipcc -O3 --offload-arch=gfx950 -o stream_width stream_width.hip && ./stream_width
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s\n", hipGetErrorString(e_)); return 1; } } while (0)

template <int PX, bool NT>
__global__ __launch_bounds__(256) void k(const float* __restrict__ t, const float* __restrict__ w, const uint8_t* __restrict__ m,
                                         float* __restrict__ a, float* __restrict__ g, int K, size_t S) {
  typedef float fv __attribute__((ext_vector_type(PX)));
  typedef uint8_t bv __attribute__((ext_vector_type(PX)));
  const size_t b = blockIdx.y;
  const size_t p = ((size_t)blockIdx.x * 256 + threadIdx.x) * PX;
  if (p >= S) return;
  for (int i = 0; i < K; ++i) {
    const size_t o = (b * K + i) * S + p;
    fv tv, wv; bv mv;
    if (NT) { tv = __builtin_nontemporal_load((const fv*)(t + o)); wv = __builtin_nontemporal_load((const fv*)(w + o)); mv = __builtin_nontemporal_load((const bv*)(m + o)); }
    else { tv = *(const fv*)(t + o); wv = *(const fv*)(w + o); mv = *(const bv*)(m + o); }
    fv av, gv;
    for (int j = 0; j < PX; ++j) { av[j] = tv[j] * wv[j] + (float)mv[j]; gv[j] = tv[j] - wv[j]; }
    if (NT) { __builtin_nontemporal_store(av, (fv*)(a + o)); __builtin_nontemporal_store(gv, (fv*)(g + o)); }
    else { *(fv*)(a + o) = av; *(fv*)(g + o) = gv; }
  }
}

template <int PX, bool NT>
float run(const float* t, const float* w, const uint8_t* m, float* a, float* g, int B, int K, size_t S) {
  dim3 grid((unsigned)((S / PX + 255) / 256), B);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k<PX, NT>), grid, dim3(256), 0, 0, t, w, m, a, g, K, S);
  hipEventRecord(e0);
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((k<PX, NT>), grid, dim3(256), 0, 0, t, w, m, a, g, K, S);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / 20 * 1e3f;
}

int main() {
  const int B = 8, K = 10; const size_t S = 544 * 544, N = (size_t)B * K * S;
  float *t, *w, *a, *g; uint8_t* m;
  CK(hipMalloc(&t, N * 4)); CK(hipMalloc(&w, N * 4)); CK(hipMalloc(&a, N * 4)); CK(hipMalloc(&g, N * 4)); CK(hipMalloc(&m, N));
  CK(hipMemset(t, 0, N * 4)); CK(hipMemset(w, 0, N * 4)); CK(hipMemset(m, 1, N));
  const double bytes = (double)N * 17;
#define R(PX, NT) { float us = run<PX, NT>(t, w, m, a, g, B, K, S); printf("px/lane %d nt %d : %8.1f us  %6.0f GB/s\n", PX, NT, us, bytes / us / 1e3); }
  R(1, false) R(1, true) R(2, false) R(2, true) R(4, false) R(4, true)
  return 0;
}
