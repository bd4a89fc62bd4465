// bwd_vec.hip -- A/B of the D = 16 cross backward: k_bwd_xdma<16, ..> against its VEC instantiation (16-byte g loads and gradient
// stores, the chunk DMA issued inside the gather; pea_xdma.h), same buffers, alternating batches of launches, outputs compared
// bit for bit.  Standalone:
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -o bwd_vec bwd_vec.hip && ./bwd_vec [B]     (B x 16 x 544^2, shifts 1,3,5,9,27)
// -DBV_W3: the second kernel is k_bwd_xdma_w3 (pea_xdma_w3.h); -DBV_CROP=true: CROP_ZERO border; -DPEA_VEC_NOSPREAD: pea_xdma.h
#include <string.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../pixel-embedded-affinity_amd/csrc/pea_xdma_w3.h"
using namespace pea;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
#ifndef BV_CROP
#define BV_CROP false
#endif
int main(int argc, char** argv) {
  constexpr int D = 16, H = 544, W = 544;
  const int B = argc > 1 ? atoi(argv[1]) : 8;
  const int shifts[5] = {1, 3, 5, 9, 27};
  const int nsh = 5, K = 2 * nsh;
  KParams P;
  memset(&P, 0, sizeof(P));
  P.B = B; P.D = D; P.Z = 1; P.Y = H; P.X = W; P.K = K; P.S = H * W; P.border = BV_CROP ? PEA_BORDER_CROP_ZERO : PEA_BORDER_CIRCULAR; P.eps = 1e-12f;
  P.ksplit = K; P.chunks = (P.S + 255) / 256; P.tiles = B * P.chunks; P.tiles_per_xcd = (P.tiles + 7) / 8;
  P.tbs = P.wbs = P.mbs = (long long)K * P.S;
  for (int i = 0; i < K; ++i) {
    P.off[i][0] = 0; P.off[i][1] = (i % 2 == 0) ? -shifts[i / 2] : 0; P.off[i][2] = (i % 2 == 1) ? -shifts[i / 2] : 0;
    P.lam[i] = 1.f; P.inv_n[i] = 1.f / (B * W); P.gscale[i] = 2.f / (B * W);
  }
  XParams C; size_t lds;
  if (!plan_xdma(P, 16, 32, 51, &C, &lds, 0)) { printf("no plan\n"); return 1; }
  if (argc > 2) { C.skew = atoi(argv[2]); C.skew_slots = 2; C.skew_mode = 0; }  // PEA_SKEW (pea_xdma.h xdma_tile)
  const size_t ne = (size_t)B * D * P.S, nk = (size_t)B * K * P.S, np = (size_t)B * P.S;
  std::vector<float> he(ne), hg(nk), hi(np);
  unsigned s = 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)(s >> 8) / 16777216.f - 0.5f; };
  for (auto& v : he) v = rnd();
  for (auto& v : hg) v = rnd() * 1e-3f;
  for (auto& v : hi) v = 1.0f + 0.1f * rnd();
  for (size_t i = 0; i < np; i += 97) hi[i] = -hi[i];  // some clamped norms (the projection's zero branch)
  float *e, *g, *inv, *dx0, *dx1, *fwd_like;
  CK(hipMalloc(&e, ne * 4)); CK(hipMalloc(&g, nk * 4)); CK(hipMalloc(&inv, np * 4)); CK(hipMalloc(&dx0, ne * 4)); CK(hipMalloc(&dx1, ne * 4));
  CK(hipMalloc(&fwd_like, (size_t)600 << 20));  // what the forward moves between two backwards (the step's cache state)
  CK(hipMemcpy(e, he.data(), ne * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(g, hg.data(), nk * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(inv, hi.data(), np * 4, hipMemcpyHostToDevice));
  CK(hipMemset(dx0, 0xff, ne * 4)); CK(hipMemset(dx1, 0xee, ne * 4));
  constexpr auto k0 = k_bwd_xdma<D, 16, 32, 51, BV_CROP, kXP>;
#ifdef BV_W3  // the three-workgroups-per-CU kernel (pea_xdma_w3.h) instead of the VEC instantiation
  constexpr auto k1 = k_bwd_xdma_w3<16, 32, 51, BV_CROP>;
  const size_t lds1 = 4 * 51 * 256;
#else
  constexpr auto k1 = k_bwd_xdma<D, 16, 32, 51, BV_CROP, kXP, kAuxNT, 0, false, false, true>;
  const size_t lds1 = lds;
#endif
  CK(hipFuncSetAttribute((const void*)k0, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  CK(hipFuncSetAttribute((const void*)k1, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));
  const dim3 grid((unsigned)(C.tiles_per_xcd * 8)), blk(512);
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  auto run = [&](int which, float* dx) {
#ifdef BV_W3
    if (which) hipLaunchKernelGGL(k1, grid, blk, lds1, 0, P, C, e, inv, g, (const float*)nullptr, dx);
#else
    if (which) hipLaunchKernelGGL(k1, grid, blk, lds1, 0, P, C, e, inv, g, (const float*)nullptr, dx, OtherArgs{}, DualArgs{});
#endif
    else hipLaunchKernelGGL(k0, grid, blk, lds, 0, P, C, e, inv, g, (const float*)nullptr, dx, OtherArgs{}, DualArgs{});
  };
  run(0, dx0); run(1, dx1);
  CK(hipDeviceSynchronize());
  std::vector<float> o0(ne), o1(ne);
  CK(hipMemcpy(o0.data(), dx0, ne * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(o1.data(), dx1, ne * 4, hipMemcpyDeviceToHost));
  size_t bad = 0, first = 0;
  for (size_t i = 0; i < ne; ++i)
    if (memcmp(&o0[i], &o1[i], 4) != 0) { if (!bad) first = i; ++bad; }
  printf("B=%d skew=%d: %zu of %zu gradient values differ between k_bwd_xdma and its VEC instantiation", B, C.skew, bad, ne);
  if (bad) {
    const size_t i = first, pl = i / P.S, r = i % P.S;
    printf(" (first: plane %zu y %zu x %zu: %g vs %g)", pl, r / W, r % W, o0[i], o1[i]);
  }
  printf("\n");
  // a second launch of VEC on the same inputs: bit-reproducible?
  CK(hipMemset(dx1, 0xee, ne * 4));
  run(1, dx1);
  CK(hipDeviceSynchronize());
  std::vector<float> o2(ne);
  CK(hipMemcpy(o2.data(), dx1, ne * 4, hipMemcpyDeviceToHost));
  size_t bad2 = 0;
  for (size_t i = 0; i < ne; ++i) bad2 += memcmp(&o1[i], &o2[i], 4) != 0;
  printf("VEC rerun: %zu values differ\n", bad2);
  for (int it = 0; it < 30; ++it) { run(0, dx0); run(1, dx1); }
  const int N = 40;
  for (int rep = 0; rep < 4; ++rep) {
    float ms[2];
    for (int which = 0; which < 2; ++which) {
      CK(hipEventRecord(a));
      for (int it = 0; it < N; ++it) run(which, which ? dx1 : dx0);
      CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      CK(hipEventElapsedTime(&ms[which], a, b));
    }
    // in the step's cache state: 600 MB of other traffic between two backwards (memset = a write stream)
    float msm, ms2[2];
    CK(hipEventRecord(a));
    for (int it = 0; it < N; ++it) hipMemsetAsync(fwd_like, it, (size_t)600 << 20, 0);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    CK(hipEventElapsedTime(&msm, a, b));
    for (int which = 0; which < 2; ++which) {
      CK(hipEventRecord(a));
      for (int it = 0; it < N; ++it) { hipMemsetAsync(fwd_like, it, (size_t)600 << 20, 0); run(which, which ? dx1 : dx0); }
      CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      CK(hipEventElapsedTime(&ms2[which], a, b));
    }
    printf("rep %d: back to back  base %.1f us  VEC %.1f us   |  behind a 600 MB memset (%.1f us)  base %.1f us  VEC %.1f us\n", rep,
           ms[0] * 1e3 / N, ms[1] * 1e3 / N, msm * 1e3 / N, (ms2[0] - msm) * 1e3 / N, (ms2[1] - msm) * 1e3 / N);
  }
  return bad || bad2 ? 2 : 0;
}
