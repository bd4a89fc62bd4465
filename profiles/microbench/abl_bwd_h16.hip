// abl_bwd_h16.hip -- the f16 projection-first backward (k_bwd_xdma_h<.., PF, HW>) timed with phases compiled out: what is a chunk
// made of?  hipcc -O3 --offload-arch=gfx950 -std=c++17 [-DPEA_ABL_H_NOGATHER] [-DPEA_ABL_H_NODMA] [-DPEA_ABL_H_NOSTORE]
// [-DPEA_ABL_H_NOILV] -o abl abl_bwd_h16.hip && ./abl     (B=8 x 64 x 544^2 f16, shifts 1,3,5,9: BASELINE configs[4]; results
// of the ablated builds are wrong by construction).
#include <string.h>
#include <cstdio>
#include <vector>
#include "../../pixel-embedded-affinity_amd/csrc/pea_xdma_h16.h"
using namespace pea;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
int main() {
  constexpr int D = 64, B = 8, H = 544, W = 544, PSU = 28;
  const int shifts[4] = {1, 3, 5, 9};
  const int K = 8;
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
  if (!plan_xdma(P, 16, 32, PSU, &C, &lds, 0)) { printf("no plan\n"); return 1; }
  lds = (size_t)5 * PSU * 256;
  const size_t ne = (size_t)B * D * P.S, nk = (size_t)B * K * P.S, np = (size_t)B * P.S;
  std::vector<__half> he(ne);
  std::vector<float> hg(nk), hi(np);
  unsigned s = 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)(s >> 8) / 16777216.f - 0.5f; };
  for (auto& v : he) v = __float2half(rnd());
  for (auto& v : hg) v = rnd() * 1e-3f;
  for (auto& v : hi) v = 1.0f + 0.1f * rnd();
  __half *e, *dx;
  float *g, *a, *inv;
  CK(hipMalloc(&e, ne * 2)); CK(hipMalloc(&dx, ne * 2)); CK(hipMalloc(&g, nk * 4)); CK(hipMalloc(&a, nk * 4)); CK(hipMalloc(&inv, np * 4));
  CK(hipMemcpy(e, he.data(), ne * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(g, hg.data(), nk * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(a, hg.data(), nk * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(inv, hi.data(), np * 4, hipMemcpyHostToDevice));
  constexpr auto kern = k_bwd_xdma_h<D, 16, 32, PSU, false, kXP, true, 4, true>;
  const dim3 grid((unsigned)(C.tiles_per_xcd * 8)), blk(512);
  hipEvent_t t0, t1; CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
  for (int it = 0; it < 5; ++it) hipLaunchKernelGGL(kern, grid, blk, lds, 0, P, C, e, inv, g, a, (const float*)nullptr, dx, (const __half*)nullptr, (const float*)nullptr);
  CK(hipEventRecord(t0));
  for (int it = 0; it < 20; ++it) hipLaunchKernelGGL(kern, grid, blk, lds, 0, P, C, e, inv, g, a, (const float*)nullptr, dx, (const __half*)nullptr, (const float*)nullptr);
  CK(hipEventRecord(t1)); CK(hipEventSynchronize(t1));
  float ms; CK(hipEventElapsedTime(&ms, t0, t1));
  printf("k_bwd_xdma_h<64, PF, HW> B=8 x 64 x 544^2 K=8: %.1f us per launch, %d tiles, region %d quads\n", ms * 50.f, C.ntiles, C.QA);
  return 0;
}
