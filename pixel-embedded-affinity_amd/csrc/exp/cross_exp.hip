// experiment harness: the LDS-DMA cross backward alone, for quick compile / A-B timing against libpea_hip.so
#include "../pea_xdma.h"
using namespace pea;
static KParams mk(const PeaDesc* d) {
  KParams P;
  P.B = d->B; P.D = d->D; P.Z = d->dims[0]; P.Y = d->dims[1]; P.X = d->dims[2]; P.K = d->K;
  P.S = P.Z * P.Y * P.X; P.border = d->border; P.flags = d->flags; P.eps = d->eps;
  P.chunks = (P.S + kBlock - 1) / kBlock; P.tiles = P.B * P.chunks; P.tiles_per_xcd = (P.tiles + kXcd - 1) / kXcd;
  for (int i = 0; i < PEA_MAX_K; ++i) for (int a = 0; a < 3; ++a) P.off[i][a] = i < d->K ? d->offsets[i][a] : 0;
  return P;
}
template <auto K> static void allow(size_t b) { (void)hipFuncSetAttribute((const void*)K, hipFuncAttributeMaxDynamicSharedMemorySize, (int)b); }
extern "C" int pea_x_inv(const PeaDesc* d, const void* e, float* inv, void* stream) {
  const KParams P = mk(d);
  hipLaunchKernelGGL(k_inv_norm<float>, dim3(P.tiles_per_xcd * kXcd), dim3(kBlock), 0, (hipStream_t)stream, P, (const float*)e, inv);
  return (int)hipGetLastError();
}
extern "C" int pea_x_bwd(const PeaDesc* d, const void* e, const float* inv, const float* g, const float* dl, void* de, int cfg, void* stream) {
  const KParams P = mk(d);
  XParams Xp; size_t lds;
  if (!plan_xdma(P, 16, 32, 51, &Xp, &lds)) return -3;
#define RUN(AUX) { constexpr auto k = k_bwd_xdma<16, 16, 32, 51, false, kXP, AUX>; allow<k>(lds); \
  hipLaunchKernelGGL(k, dim3(Xp.tiles_per_xcd * kXcd), dim3(512), lds, (hipStream_t)stream, P, Xp, (const float*)e, inv, g, dl, (float*)de); }
  if (cfg == 2) RUN(0) else if (cfg == 3) RUN(2) else if (cfg == 4) RUN(16) else if (cfg == 5) RUN(18) else if (cfg == 6) RUN(1) else return -3;
  return (int)hipGetLastError();
}
