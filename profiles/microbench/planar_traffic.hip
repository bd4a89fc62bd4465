// Microbenchmark: the backward's HBM traffic without its arithmetic.  Reads 16 embedding planes + 10 g planes and
// writes 16 gradient planes of B=8 images of 544x544 (NCHW planes), walking tiles XCD-contiguously like the kernels.
// Variants: tile shape (rows x cols per workgroup of 1024 lanes) and bytes per lane per access.
//   hipcc -O3 --offload-arch=gfx950 -o planar_traffic planar_traffic.hip && ./planar_traffic
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s\n", hipGetErrorString(e_)); return 1; } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int B = 8, D = 16, K = 10, H = 544, W = 544, S = H * W;

// TH x TW pixels per workgroup, PX pixels per lane along x (1 or 4); lanes = TH * TW / PX (<= 1024)
template <int TH, int TW, int PX, bool WR>
__global__ __launch_bounds__(1024) void k(const float* __restrict__ e, const float* __restrict__ g, float* __restrict__ de,
                                          int tiles_x, int tiles_per_img, int ntiles, int tpx) {
  const int bid = blockIdx.x;
  const int tile = (bid % 8) * tpx + bid / 8;
  if (tile >= ntiles) return;
  const int b = tile / tiles_per_img, rem = tile % tiles_per_img;
  const int y0 = (rem / tiles_x) * TH, x0 = (rem % tiles_x) * TW;
  constexpr int LW = TW / PX;  // lanes per tile row
  const int ly = threadIdx.x / LW, lx = (threadIdx.x % LW) * PX;
  const int y = y0 + ly, x = x0 + lx;
  if (y >= H || x >= W) return;
  const size_t p = (size_t)y * W + x;
  float acc[PX];
  for (int j = 0; j < PX; ++j) acc[j] = 0.f;
  float v[D][PX];
#pragma unroll
  for (int c = 0; c < D; ++c) {
    const float* src = e + ((size_t)b * D + c) * S + p;
    if (PX == 4) { const f4 t = *(const f4*)src; v[c][0] = t.x; v[c][1 % PX] = t.y; v[c][2 % PX] = t.z; v[c][3 % PX] = t.w; }
    else v[c][0] = *src;
  }
#pragma unroll
  for (int i = 0; i < K; ++i) {
    const float* src = g + ((size_t)b * K + i) * S + p;
    if (PX == 4) { const f4 t = *(const f4*)src; acc[0] += t.x; acc[1 % PX] += t.y; acc[2 % PX] += t.z; acc[3 % PX] += t.w; }
    else acc[0] += *src;
  }
  if (WR) {
#pragma unroll
    for (int c = 0; c < D; ++c) {
      float* dst = de + ((size_t)b * D + c) * S + p;
      if (PX == 4) { f4 t; t.x = v[c][0] + acc[0]; t.y = v[c][1 % PX] + acc[1 % PX]; t.z = v[c][2 % PX] + acc[2 % PX]; t.w = v[c][3 % PX] + acc[3 % PX]; *(f4*)dst = t; }
      else *dst = v[c][0] + acc[0];
    }
  } else {
    float s = 0.f;
    for (int c = 0; c < D; ++c) for (int j = 0; j < PX; ++j) s += v[c][j];
    for (int j = 0; j < PX; ++j) s += acc[j];
    if (s == 1234.5f) de[0] = s;
  }
}

// forward mix: read e (16 planes) + target, weight (10 + 10 f32 planes) + mask (10 u8 planes), write affs + g (20 planes)
template <bool WG>
__global__ __launch_bounds__(1024) void kf(const float* __restrict__ e, const float* __restrict__ t, const float* __restrict__ w,
                                           const unsigned char* __restrict__ m, float* __restrict__ affs, float* __restrict__ gout,
                                           int tiles_x, int tiles_per_img, int ntiles, int tpx) {
  const int bid = blockIdx.x;
  const int tile = (bid % 8) * tpx + bid / 8;
  if (tile >= ntiles) return;
  const int b = tile / tiles_per_img, rem = tile % tiles_per_img;
  const int y = (rem / tiles_x) * 32 + threadIdx.x / 32, x = (rem % tiles_x) * 32 + threadIdx.x % 32;
  if (y >= H || x >= W) return;
  const size_t p = (size_t)y * W + x;
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < D; ++c) s += e[((size_t)b * D + c) * S + p];
#pragma unroll
  for (int i = 0; i < K; ++i) {
    const size_t o = ((size_t)b * K + i) * S + p;
    const float a = s * t[o] + w[o] * (float)m[o];
    affs[o] = a;
    if (WG) gout[o] = a - s;
  }
}

template <bool WG>
int runf(const float* e, const float* t, const float* w, const unsigned char* m, float* affs, float* gout, const char* name) {
  const int tiles_x = 17, tpi = 289, nt = tpi * B, tpx = (nt + 7) / 8;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kf<WG>, dim3(tpx * 8), dim3(1024), 0, 0, e, t, w, m, affs, gout, tiles_x, tpi, nt, tpx);
  (void)hipEventRecord(e0);
  const int reps = 20;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kf<WG>, dim3(tpx * 8), dim3(1024), 0, 0, e, t, w, m, affs, gout, tiles_x, tpi, nt, tpx);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double us = ms / reps * 1e3;
  const double bytes = (double)B * S * (4.0 * D + 9.0 * K + 4.0 * K * (WG ? 2 : 1));
  printf("%-52s %7.1f us  %5.2f TB/s\n", name, us, bytes / us / 1e6);
  return 0;
}

template <int TH, int TW, int PX, bool WR>
int run(const float* e, const float* g, float* de, const char* name) {
  const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH, tpi = tiles_x * tiles_y, nt = tpi * B, tpx = (nt + 7) / 8;
  const int threads = TH * TW / PX;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<TH, TW, PX, WR>), dim3(tpx * 8), dim3(threads), 0, 0, e, g, de, tiles_x, tpi, nt, tpx);
  (void)hipEventRecord(e0);
  const int reps = 20;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k<TH, TW, PX, WR>), dim3(tpx * 8), dim3(threads), 0, 0, e, g, de, tiles_x, tpi, nt, tpx);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double us = ms / reps * 1e3;
  const double bytes = (double)B * S * 4 * (D + K + (WR ? D : 0));
  printf("%-52s %7.1f us  %5.2f TB/s\n", name, us, bytes / us / 1e6);
  return 0;
}

int main() {
  float *e, *g, *de;
  CK(hipMalloc(&e, (size_t)B * D * S * 4)); CK(hipMalloc(&g, (size_t)B * K * S * 4)); CK(hipMalloc(&de, (size_t)B * D * S * 4));
  CK(hipMemset(e, 0, (size_t)B * D * S * 4)); CK(hipMemset(g, 0, (size_t)B * K * S * 4));
  run<32, 32, 1, true>(e, g, de, "rd 26 + wr 16 planes, tile 32x32, 4 B/lane");
  run<16, 64, 1, true>(e, g, de, "rd 26 + wr 16 planes, tile 16x64, 4 B/lane");
  run<8, 128, 1, true>(e, g, de, "rd 26 + wr 16 planes, tile 8x128, 4 B/lane");
  run<2, 512, 1, true>(e, g, de, "rd 26 + wr 16 planes, tile 2x512, 4 B/lane");
  run<32, 128, 4, true>(e, g, de, "rd 26 + wr 16 planes, tile 32x128, 16 B/lane");
  run<64, 64, 4, true>(e, g, de, "rd 26 + wr 16 planes, tile 64x64, 16 B/lane");
  run<32, 32, 4, true>(e, g, de, "rd 26 + wr 16 planes, tile 32x32 (256 lanes), 16 B/lane");
  run<32, 32, 1, false>(e, g, de, "rd 26 planes only, tile 32x32, 4 B/lane");
  run<16, 64, 1, false>(e, g, de, "rd 26 planes only, tile 16x64, 4 B/lane");
  run<32, 128, 4, false>(e, g, de, "rd 26 planes only, tile 32x128, 16 B/lane");
  float *t, *w, *affs; unsigned char* m;
  CK(hipMalloc(&t, (size_t)B * K * S * 4)); CK(hipMalloc(&w, (size_t)B * K * S * 4)); CK(hipMalloc(&affs, (size_t)B * K * S * 4));
  CK(hipMalloc(&m, (size_t)B * K * S));
  CK(hipMemset(t, 0, (size_t)B * K * S * 4)); CK(hipMemset(w, 0, (size_t)B * K * S * 4)); CK(hipMemset(m, 1, (size_t)B * K * S));
  runf<true>(e, t, w, m, affs, g, "forward mix: rd e,t,w,m  wr affs,g  (554 MB)");
  runf<false>(e, t, w, m, affs, g, "forward mix: rd e,t,w,m  wr affs     (459 MB)");
  return 0;
}
