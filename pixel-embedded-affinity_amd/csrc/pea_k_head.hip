// pea_k_head.hip -- entry points of the embedding head (pea_head.h: the 1x1 convolution in front of the path and its backward).
// One translation unit of libpea_hip.so (pea_host.h).
#include <algorithm>

#include "pea_host.h"
#include "pea_head.h"

using namespace pea;

extern "C" {

// ---- the embedding head (1x1 convolution, pea_head.h) ---------------------------------------------------------------
// (C, D) pairs of the reference's heads: ResUNet 32 / 64 / 128 / 256 -> 16 (CVPPP) or 32 (BBBC039V1),
// superhuman 3D U-Net 28 / 36 / 48 / 64 / 80 -> 16
#define PEA_HEAD_CASES(X) \
  X(28, 16) X(32, 16) X(36, 16) X(48, 16) X(64, 16) X(80, 16) X(128, 16) X(256, 16) X(32, 32) X(64, 32) X(128, 32) X(256, 32)

static bool head_supported(int C, int D) {
#define PEA_HEAD_Q(c, d) if (C == c && D == d) return true;
  PEA_HEAD_CASES(PEA_HEAD_Q)
#undef PEA_HEAD_Q
  return false;
}

size_t pea_head_workspace_bytes(int C, int D) {
  if (C < 1 || D < 1) return 0;
  return (size_t)kHeadMaxWg * ((size_t)D * C + D) * sizeof(float);
}

int pea_head_fwd(int B, int C, int D, size_t S, const float* x, const float* W, const float* bias, float* e, void* stream) {
  if (B < 1 || C < 1 || D < 1 || S < 1) return PEA_E_DESC;
  if (!x || !W || !e) return PEA_E_NULL;
  if (misaligned(x, 4) || misaligned(W, 4) || misaligned(bias, 4) || misaligned(e, 4)) return PEA_E_ALIGN;
  if (!head_supported(C, D)) return PEA_E_UNSUPPORTED;
  const size_t chunks = (S + kHeadBlock - 1) / kHeadBlock;
  if (chunks * (size_t)B > 0x7fffffffULL) return PEA_E_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid((unsigned)(chunks * B)), blk(kHeadBlock);
#define PEA_HEAD_F(c, d) \
  if (C == c && D == d) hipLaunchKernelGGL((k_head_fwd<c, d>), grid, blk, 0, s, x, W, bias, e, (long long)S, (int)chunks);
  PEA_HEAD_CASES(PEA_HEAD_F)
#undef PEA_HEAD_F
  return hip_rc();
}

int pea_head_bwd(int B, int C, int D, size_t S, const float* x, const float* W, const float* de, float* dx, float* dW, float* db,
                 void* workspace, size_t workspace_bytes, void* stream) {
  if (B < 1 || C < 1 || D < 1 || S < 1) return PEA_E_DESC;
  if (!x || !W || !de || !dW) return PEA_E_NULL;
  if (misaligned(x, 4) || misaligned(W, 4) || misaligned(de, 4) || misaligned(dx, 4) || misaligned(dW, 4) || misaligned(db, 4) ||
      misaligned(workspace, 4))
    return PEA_E_ALIGN;
  if (!head_supported(C, D)) return PEA_E_UNSUPPORTED;
  if (!workspace || workspace_bytes < pea_head_workspace_bytes(C, D)) return PEA_E_WORKSPACE;
  const size_t chunks = (S + kHeadBlock - 1) / kHeadBlock;
  if (chunks * (size_t)B > 0x7fffffffULL) return PEA_E_UNSUPPORTED;
  const int nchunks = (int)(chunks * B);
  // dW: a multiple of the CU count that the instantiation keeps resident, no partial round
  const int nwg = std::min(nchunks, std::min(kHeadMaxWg, ((C <= 48 && D == 16) ? 4 : 2) * device_cus()));
  hipStream_t s = (hipStream_t)stream;
  float* partials = (float*)workspace;
  if (dx) {
    const dim3 grid((unsigned)nchunks), blk(kHeadBlock);
#define PEA_HEAD_X(c, d) \
  if (C == c && D == d) hipLaunchKernelGGL((k_head_dx<c, d>), grid, blk, 0, s, W, de, dx, (long long)S, (int)chunks);
    PEA_HEAD_CASES(PEA_HEAD_X)
#undef PEA_HEAD_X
  }
#define PEA_HEAD_B(c, d)                                                                                              \
  if (C == c && D == d)                                                                                               \
    hipLaunchKernelGGL((k_head_dw<c, d>), dim3((unsigned)nwg), dim3(kHeadBlock), 0, s, x, de, partials, (long long)S, \
                       (int)chunks, nchunks);
  PEA_HEAD_CASES(PEA_HEAD_B)
#undef PEA_HEAD_B
  const int rc = hip_rc();
  if (rc) return rc;
  const int n = D * C + D;
  hipLaunchKernelGGL(k_head_finalize, dim3((unsigned)((n + 15) / 16)), dim3(256), 0, s, partials, nwg, D * C, n, dW, db);
  return hip_rc();
}

}  // extern "C"
