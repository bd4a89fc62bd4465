// stamp_bwd.hip -- diagnostic build of the cross backward with in-kernel s_memtime stamps (cdna_hip_programming.md section 7):
// where do the waves of k_bwd_xdma<D> spend a tile?  Standalone: hipcc -O3 --offload-arch=gfx950 -std=c++17 -DPEA_STAMPS
// -DSTAMP_D=64 -o stamp_bwd stamp_bwd.hip && ./stamp_bwd   (B=8 x D x 544^2, shifts 1,3,5,9[,27], random data).
// Prints, as medians over the stamped waves, the cycles of: prologue issue, prologue wait, and per chunk: gather, wait + barrier,
// DMA issue; then the epilogue.  Read the SHARES, not the length: the stamps' fences forbid overlaps the real kernel has.
#include <string.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#include "../../pixel-embedded-affinity_amd/csrc/pea_xdma.h"
using namespace pea;
#ifndef STAMP_D
#define STAMP_D 64
#endif
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
int main() {
  constexpr int D = STAMP_D, B = 8, H = 544, W = 544;
  const int shifts[5] = {1, 3, 5, 9, 27};
  const int nsh = D == 64 ? 4 : 5, K = 2 * nsh;
  KParams P;
  memset(&P, 0, sizeof(P));
  P.B = B; P.D = D; P.Z = 1; P.Y = H; P.X = W; P.K = K; P.S = H * W; P.border = PEA_BORDER_CIRCULAR; P.eps = 1e-12f;
  P.ksplit = K; P.chunks = (P.S + 255) / 256; P.tiles = B * P.chunks; P.tiles_per_xcd = (P.tiles + 7) / 8;
  P.tbs = P.wbs = P.mbs = (long long)K * P.S;
  for (int i = 0; i < K; ++i) {
    P.off[i][0] = 0; P.off[i][1] = (i % 2 == 0) ? -shifts[i / 2] : 0; P.off[i][2] = (i % 2 == 1) ? -shifts[i / 2] : 0;
    P.lam[i] = 1.f; P.inv_n[i] = 1.f / (B * W); P.gscale[i] = 2.f / (B * W);
  }
  XParams C; size_t lds;
  if (!plan_xdma(P, 16, 32, 51, &C, &lds, 0)) { printf("no plan\n"); return 1; }
  const size_t ne = (size_t)B * D * P.S, nk = (size_t)B * K * P.S, np = (size_t)B * P.S;
  std::vector<float> he(ne), hg(nk), hi(np);
  unsigned s = 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)(s >> 8) / 16777216.f - 0.5f; };
  for (auto& v : he) v = rnd();
  for (auto& v : hg) v = rnd() * 1e-3f;
  for (auto& v : hi) v = 1.0f + 0.1f * rnd();
  float *e, *g, *inv, *dx;
#ifdef PEA_STAMPS
  unsigned long long* st;
#endif
  CK(hipMalloc(&e, ne * 4)); CK(hipMalloc(&g, nk * 4)); CK(hipMalloc(&inv, np * 4)); CK(hipMalloc(&dx, ne * 4));
#ifdef PEA_STAMPS
  CK(hipMalloc(&st, (size_t)kStampWgs * 8 * kStampN * 8));
#endif
  CK(hipMemcpy(e, he.data(), ne * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(g, hg.data(), nk * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(inv, hi.data(), np * 4, hipMemcpyHostToDevice));
  constexpr int XP = D > 32 ? 8 : kXP;
  constexpr auto kern = k_bwd_xdma<D, 16, 32, 51, false, XP>;
  CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const dim3 grid((unsigned)(C.tiles_per_xcd * 8)), blk(512);
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int it = 0; it < 5; ++it) hipLaunchKernelGGL(kern, grid, blk, lds, 0, P, C, e, inv, g, (const float*)nullptr, dx, OtherArgs{}, DualArgs{});
  CK(hipEventRecord(a));
  for (int it = 0; it < 20; ++it) hipLaunchKernelGGL(kern, grid, blk, lds, 0, P, C, e, inv, g, (const float*)nullptr, dx, OtherArgs{}, DualArgs{});
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
#ifndef PEA_STAMPS
  printf("D=%d K=%d: %.1f us per launch (no stamps), %d tiles, region %d quads\n", D, K, ms * 50.f, C.ntiles, C.QA);
  return 0;
#else
  printf("D=%d K=%d: %.1f us per launch (stamped build), %d tiles, region %d quads\n", D, K, ms * 50.f, C.ntiles, C.QA);
  std::vector<unsigned long long> hs((size_t)kStampWgs * 8 * kStampN);
  CK(hipMemcpy(hs.data(), st, hs.size() * 8, hipMemcpyDeviceToHost));
  constexpr int NP = D / 2;
  auto med = [&](int i0, int i1) {  // median over waves of stamp[i1] - stamp[i0]
    std::vector<long long> v;
    for (int w = 0; w < kStampWgs * 8; ++w) {
      const unsigned long long t0 = hs[(size_t)w * kStampN + i0], t1 = hs[(size_t)w * kStampN + i1];
      if (t0 && t1 && t1 > t0) v.push_back((long long)(t1 - t0));
    }
    if (v.empty()) return -1LL;
    std::sort(v.begin(), v.end());
    return v[v.size() / 2];
  };
  printf("prologue issue (DMA inv + chunk 0, 20 g loads, DMA chunk 1): %lld cycles\n", med(0, 1));
  printf("prologue wait + barrier: %lld\n", med(1, 2));
  long long sg = 0, sw = 0, sd = 0;
  int shown = 0;
  for (int ps = 0; ps < NP && 5 + 3 * ps < kStampN - 1; ++ps) {
    const int sprev = ps == 0 ? 2 : 5 + 3 * (ps - 1);
    const long long gth = med(sprev, 3 + 3 * ps);
    const long long wt = ps + 1 < NP ? med(3 + 3 * ps, 4 + 3 * ps) : 0, dm = ps + 1 < NP ? med(4 + 3 * ps, 5 + 3 * ps) : 0;
    sg += gth; sw += wt; sd += dm; ++shown;
    if (ps < 6 || ps == NP - 1) printf("chunk %2d: gather %5lld   wait+barrier %5lld   dma issue %5lld\n", ps, gth, wt, dm);
  }
  printf("mean over %d chunks: gather %lld  wait+barrier %lld  dma issue %lld  (sum per chunk %lld cycles)\n", shown, sg / shown, sw / shown,
         sd / shown, (sg + sw + sd) / shown);
  printf("whole workgroup (first stamp to after the stores): %lld cycles\n", med(0, kStampLast));
  return 0;
#endif
}
