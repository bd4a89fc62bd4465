// Does gfx950 serve 4-byte-aligned (not 16-byte-aligned) buffer_load_dwordx4 / dwordx2 correctly, and how fast?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned u4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* src, float* dst, int shift, int n4) {
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (int)0x80000000u, 0x00020000);
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n4) return;
  u4 v = __builtin_bit_cast(u4, __builtin_amdgcn_raw_buffer_load_b128(r, (unsigned)(t * 16 + shift * 4), 0, 0));
  float* o = dst + (size_t)t * 4;
  o[0] = __builtin_bit_cast(float, v.x); o[1] = __builtin_bit_cast(float, v.y); o[2] = __builtin_bit_cast(float, v.z); o[3] = __builtin_bit_cast(float, v.w);
}
int main() {
  const int n4 = 1 << 22; const size_t n = (size_t)n4 * 4 + 64;
  float *h = (float*)malloc(n * 4), *src, *dst, *out = (float*)malloc((size_t)n4 * 16);
  for (size_t i = 0; i < n; ++i) h[i] = (float)(i % 1000003);
  hipMalloc(&src, n * 4); hipMalloc(&dst, (size_t)n4 * 16); hipMemcpy(src, h, n * 4, hipMemcpyHostToDevice);
  for (int shift = 0; shift < 4; ++shift) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k, dim3(n4 / 256), dim3(256), 0, 0, src, dst, shift, n4);
    hipEventRecord(a);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k, dim3(n4 / 256), dim3(256), 0, 0, src, dst, shift, n4);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    hipMemcpy(out, dst, (size_t)n4 * 16, hipMemcpyDeviceToHost);
    if (shift == 1) { for (int i = 0; i < 12; ++i) printf("%g(%g) ", out[i], h[i + shift]); printf("\n"); }
    size_t bad = 0;
    for (size_t i = 0; i < (size_t)n4 * 4; ++i) bad += out[i] != h[i + shift];
    printf("shift %d dwords: %zu wrong of %zu, %.1f us per launch (%.0f GB/s r+w)\n", shift, bad, (size_t)n4 * 4, ms * 100, (double)n4 * 32 / (ms * 100) / 1e3);
  }
  return 0;
}
