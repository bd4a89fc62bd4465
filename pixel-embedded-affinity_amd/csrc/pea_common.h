// pea_common.h -- kernel parameters and device helpers common to every kernel family.
#pragma once
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pea.h"

namespace pea {

constexpr int kBlock = 256;  // 4 waves of 64
constexpr int kXcd = 8;

struct KParams {
  int B, D, Z, Y, X, K;
  int S;  // Z*Y*X (fits int32: checked on the host)
  int border;
  unsigned flags;
  float eps;
  int chunks;          // workgroups per batch item = ceil(S / kBlock)
  int tiles;           // B * chunks
  int tiles_per_xcd;   // ceil(tiles / 8)
  int off[PEA_MAX_K][3];
  float lam[PEA_MAX_K];
  float inv_n[PEA_MAX_K];   // 1 / N_i
  float gscale[PEA_MAX_K];  // 2 * lambda_i / N_i
  long long tbs, wbs, mbs;  // batch strides (elements) of target / weight / mask
  int ksplit;               // offset channels >= ksplit are addressed through a second buffer resource based ksplit planes further:
                            // K (no split) unless the [K, Z, Y, X] block of a batch item reaches 2 GiB (the raw-buffer range check
                            // counts the scalar plane offset: pea_hip.hip plan_tiles); only k_fwd_tiled / k_bwd_tiled honour it
};

template <typename T>
__device__ __forceinline__ float ld(const T* p, size_t i);
template <>
__device__ __forceinline__ float ld<float>(const float* p, size_t i) { return p[i]; }
template <>
__device__ __forceinline__ float ld<__half>(const __half* p, size_t i) { return __half2float(p[i]); }

__device__ __forceinline__ void st(float* p, size_t i, float v) { p[i] = v; }
__device__ __forceinline__ void st(__half* p, size_t i, float v) { p[i] = __float2half(v); }

// XCD-aware remap: hardware deals consecutive workgroup ids round-robin over the 8 XCDs, so
// id % 8 labels the XCD group.  Give group g the contiguous logical tiles [g*tpx, (g+1)*tpx).
__device__ __forceinline__ int logical_tile(const KParams& P) {
  const int bid = blockIdx.x;
  return (bid % kXcd) * P.tiles_per_xcd + bid / kXcd;
}

// REPLICATE, role B: the coordinates c' along one axis (extent n) with clamp(c' + o) == c, as [lo, hi] (empty: lo > hi).
// Interior c has the one pre-image c - o; a border coordinate collects every c' that the clamp folds onto it.
__device__ __forceinline__ void clamp_preimage(int c, int o, int n, int& lo, int& hi) {
  lo = hi = c - o;
  if (o < 0 && c == 0) { lo = 0; hi = min(-o, n - 1); }
  else if (o > 0 && c == n - 1) { lo = max(n - 1 - o, 0); hi = n - 1; }
  else if (lo < 0 || lo > n - 1) { lo = 1; hi = 0; }
}

// neighbour of (z,y,x) displaced by o; returns flat index or -1 (CROP_ZERO, outside)
__device__ __forceinline__ int neighbour(const KParams& P, int z, int y, int x, int oz, int oy, int ox) {
  int zz = z + oz, yy = y + oy, xx = x + ox;
  if (P.border == PEA_BORDER_CIRCULAR) {  // host guarantees |o| < dim
    zz += (zz < 0) ? P.Z : 0; zz -= (zz >= P.Z) ? P.Z : 0;
    yy += (yy < 0) ? P.Y : 0; yy -= (yy >= P.Y) ? P.Y : 0;
    xx += (xx < 0) ? P.X : 0; xx -= (xx >= P.X) ? P.X : 0;
  } else if (P.border == PEA_BORDER_REPLICATE) {  // index clamped into the volume: every pair exists
    zz = min(max(zz, 0), P.Z - 1);
    yy = min(max(yy, 0), P.Y - 1);
    xx = min(max(xx, 0), P.X - 1);
  } else if ((unsigned)zz >= (unsigned)P.Z || (unsigned)yy >= (unsigned)P.Y || (unsigned)xx >= (unsigned)P.X) {
    return -1;
  }
  return (zz * P.Y + yy) * P.X + xx;
}

// activation of the affs output (include/pea.h PEA_FLAG_*): flags are wave-uniform, the common case (0) is one scalar branch
constexpr unsigned kActMask = PEA_FLAG_RELU_AFFS | PEA_FLAG_ONE_MINUS | PEA_FLAG_HALF_SHIFT | PEA_FLAG_CLAMP01;
__device__ __forceinline__ float act_affs(float a, unsigned af) {
  if (af == 0) return a;
  if (af & PEA_FLAG_HALF_SHIFT) a = (a + 1.0f) * 0.5f;
  if (af & PEA_FLAG_RELU_AFFS) a = fmaxf(a, 0.f);
  if (af & PEA_FLAG_CLAMP01) a = fminf(fmaxf(a, 0.f), 1.0f);
  if (af & PEA_FLAG_ONE_MINUS) a = 1.0f - a;
  return a;
}

__device__ __forceinline__ float inv_norm(float ss, float eps) { return 1.0f / fmaxf(sqrtf(ss), eps); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

}  // namespace pea
