// pea_hip.hip — MI355X (gfx950 / CDNA4) kernels + C ABI of the embedding -> affinity hot path.
//
// Replaces the Python op loops of the reference (weih527/Pixel-Embedded-Affinity):
//   scripts_cvppp/loss/loss_embedding_mse.py:7-95   (2D: normalize -> K x torch.roll/mul/sum -> WeightedMSE)
//   scripts_ac3ac4/loss/loss_embedding_mse.py:7-289 (3D: cropped slices, norm1 / norm5 / ema / inf)
//   loss/loss.py:106-124 WeightedMSE                (fused)
// and the autograd backward of those, by one forward launch (+ a tiny deterministic loss
// reduction) and one backward launch.  See include/pea.h for the contract and DESIGN.md for the
// data layout, the per-kernel roofline and the algorithmic byte counts.
//
// Design notes (gfx950):
//   * HBM-bound op (2-7 flop/B): no MFMA.  One lane = one pixel, D-loop in registers, NCHW planes
//     => every stencil read is a coalesced row read shifted by dx elements.
//   * the L2 norm of the neighbour is accumulated while its channels stream in for the dot product,
//     so ehat is never materialised:  a = <e_p, e_q> / (max(|e_p|,eps) * max(|e_q|,eps)).
//   * 8 XCDs with private L2s: the 1-D grid is remapped so each XCD walks a contiguous span of the
//     batch/rows; stencil re-reads then hit that XCD's own L2 instead of crossing the fabric.
//   * loss: per-lane LDS slots -> per-workgroup partials in a caller-provided workspace -> fixed-order
//     f64 reduction.  No float atomics anywhere: results are bit-reproducible run to run.
//   * backward is in gather form (each pixel pulls its 2K contributions): no atomics.
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pea.h"

namespace {

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

// neighbour of (z,y,x) displaced by sign*o; returns flat index or -1 (CROP_ZERO, outside)
__device__ __forceinline__ int neighbour(const KParams& P, int z, int y, int x, int oz, int oy, int ox) {
  int zz = z + oz, yy = y + oy, xx = x + ox;
  if (P.border == PEA_BORDER_CIRCULAR) {  // host guarantees |o| < dim
    zz += (zz < 0) ? P.Z : 0; zz -= (zz >= P.Z) ? P.Z : 0;
    yy += (yy < 0) ? P.Y : 0; yy -= (yy >= P.Y) ? P.Y : 0;
    xx += (xx < 0) ? P.X : 0; xx -= (xx >= P.X) ? P.X : 0;
  } else if ((unsigned)zz >= (unsigned)P.Z || (unsigned)yy >= (unsigned)P.Y || (unsigned)xx >= (unsigned)P.X) {
    return -1;
  }
  return (zz * P.Y + yy) * P.X + xx;
}

__device__ __forceinline__ float inv_norm(float ss, float eps) { return 1.0f / fmaxf(sqrtf(ss), eps); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

// ------------------------------------------------------------------------------------------------
// forward (direct form): affs, and (TRAIN) per-workgroup loss partials
//   D_T > 0: channels unrolled, own pixel kept in registers;  D_T == 0: generic D, own pixel re-read (L1)
// ------------------------------------------------------------------------------------------------
template <typename T, int D_T, bool TRAIN>
__global__ __launch_bounds__(kBlock) void k_fwd_direct(const KParams P, const T* __restrict__ e,
                                                       const T* __restrict__ eo,
                                                       const float* __restrict__ target,
                                                       const float* __restrict__ weight,
                                                       const uint8_t* __restrict__ mask,
                                                       float* __restrict__ affs, float* __restrict__ partials) {
  extern __shared__ float s_acc[];  // [K][kBlock], TRAIN only
  const int tile = logical_tile(P);
  if (tile >= P.tiles) return;  // whole workgroup exits together (tile is uniform)
  const int b = tile / P.chunks;
  const int p = (tile - b * P.chunks) * kBlock + threadIdx.x;
  const bool live = p < P.S;
  const int D = D_T ? D_T : P.D;
  const size_t S = (size_t)P.S;
  const T* eb = e + (size_t)b * D * S;
  const T* ob = eo + (size_t)b * D * S;
  const size_t kb = (size_t)b * P.K * S;

  int x = 0, y = 0, z = 0;
  float ec[D_T ? D_T : 1];
  float inv_p = 0.f;
  if (live) {
    const int yx = P.Y * P.X;
    z = p / yx;
    const int r = p - z * yx;
    y = r / P.X;
    x = r - y * P.X;
    float ss = 0.f;
    if (D_T) {
#pragma unroll
      for (int c = 0; c < D_T; ++c) {
        ec[c] = ld(eb, c * S + p);
        ss = fmaf(ec[c], ec[c], ss);
      }
    } else {
      for (int c = 0; c < D; ++c) {
        const float v = ld(eb, c * S + p);
        ss = fmaf(v, v, ss);
      }
    }
    inv_p = inv_norm(ss, P.eps);
  }

  for (int i = 0; i < P.K; ++i) {
    float contrib = 0.f;
    if (live) {
      const int q = neighbour(P, z, y, x, P.off[i][0], P.off[i][1], P.off[i][2]);
      float a = 0.f;
      if (q >= 0) {
        float dot = 0.f, sq = 0.f;
        if (D_T) {
#pragma unroll
          for (int c = 0; c < D_T; ++c) {
            const float v = ld(ob, c * S + q);
            dot = fmaf(ec[c], v, dot);
            sq = fmaf(v, v, sq);
          }
        } else {
          for (int c = 0; c < D; ++c) {
            const float v = ld(ob, c * S + q);
            dot = fmaf(ld(eb, c * S + p), v, dot);
            sq = fmaf(v, v, sq);
          }
        }
        a = dot * inv_p * inv_norm(sq, P.eps);
      }
      const size_t ki = kb + (size_t)i * S + p;
      if (affs) affs[ki] = (P.flags & PEA_FLAG_RELU_AFFS) ? fmaxf(a, 0.f) : a;
      if (TRAIN && q >= 0) {
        const size_t in = (size_t)i * S + p;
        const float m = mask ? (float)mask[(size_t)b * P.mbs + in] : 1.f;
        const float r = a * m - target[(size_t)b * P.tbs + in] * m;
        contrib = weight[(size_t)b * P.wbs + in] * r * r;
      }
    }
    if (TRAIN) s_acc[i * kBlock + threadIdx.x] = contrib;
  }

  if (TRAIN) {
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = wave; i < P.K; i += kBlock / 64) {
      const float* row = s_acc + i * kBlock;
      float v = (row[lane] + row[lane + 64]) + (row[lane + 128] + row[lane + 192]);
      v = wave_sum(v);
      if (lane == 0) partials[(size_t)tile * P.K + i] = v;
    }
  }
}

// fixed-order f64 reduction of the per-workgroup partials: loss_out = {loss, L_0..L_{K-1}}
__global__ __launch_bounds__(1024) void k_loss_finalize(const KParams P, const float* __restrict__ partials,
                                                        int nparts, float* __restrict__ loss_out) {
  __shared__ double s_l[PEA_MAX_K];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int i = wave; i < P.K; i += 16) {
    double acc = 0.0;
    for (int t = lane; t < nparts; t += 64) acc += (double)partials[(size_t)t * P.K + i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    if (lane == 0) {
      const double Li = acc * (double)P.inv_n[i];
      s_l[i] = Li;
      loss_out[1 + i] = (float)Li;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double tot = 0.0;
    for (int i = 0; i < P.K; ++i) tot += (double)P.lam[i] * s_l[i];
    loss_out[0] = (float)tot;
  }
}

// ------------------------------------------------------------------------------------------------
// backward (direct gather form).  x = the tensor being differentiated.
//   ROLE_A: x is the first operand:  G(p) += g_i(p)       * nhat(p + o_i),  n = nb (second operand)
//   ROLE_B: x is the second operand: G(p) += g_i(p - o_i) * nhat(p - o_i),  n = nb2 (first operand)
//   self loss: both roles, nb = nb2 = x.   EXPLICIT: g_i = d_affs[b,i,.] (vjp for foreign criteria)
//   de(p) = dloss * (G - xhat <xhat, G>) / n(p)        (G / eps when |x(p)| < eps)
// ------------------------------------------------------------------------------------------------
template <typename T, int D_T, bool ROLE_A, bool ROLE_B, bool EXPLICIT>
__global__ __launch_bounds__(kBlock) void k_bwd_direct(const KParams P, const T* __restrict__ xt,
                                                       const T* __restrict__ nbA, const T* __restrict__ nbB,
                                                       const float* __restrict__ target,
                                                       const float* __restrict__ weight,
                                                       const uint8_t* __restrict__ mask,
                                                       const float* __restrict__ d_affs,
                                                       const float* __restrict__ dloss, T* __restrict__ dx) {
  static_assert(D_T > 0, "backward is specialised on D");
  const int tile = logical_tile(P);
  if (tile >= P.tiles) return;
  const int b = tile / P.chunks;
  const int p = (tile - b * P.chunks) * kBlock + threadIdx.x;
  if (p >= P.S) return;
  const size_t S = (size_t)P.S;
  const T* xb = xt + (size_t)b * D_T * S;
  const size_t kb = (size_t)b * P.K * S;
  const float dl = EXPLICIT ? 1.f : dloss[0];

  const int yx = P.Y * P.X;
  const int z = p / yx;
  const int r0 = p - z * yx;
  const int y = r0 / P.X;
  const int x = r0 - y * P.X;

  float xc[D_T], G[D_T];
  float ss = 0.f;
#pragma unroll
  for (int c = 0; c < D_T; ++c) {
    xc[c] = ld(xb, c * S + p);
    ss = fmaf(xc[c], xc[c], ss);
    G[c] = 0.f;
  }
  const float nrm = sqrtf(ss);
  const float inv_p = 1.0f / fmaxf(nrm, P.eps);

  for (int i = 0; i < P.K; ++i) {
    const int oz = P.off[i][0], oy = P.off[i][1], ox = P.off[i][2];
    if (ROLE_A) {
      const int q = neighbour(P, z, y, x, oz, oy, ox);
      if (q >= 0) {
        const T* nb = nbA + (size_t)b * D_T * S;
        float v[D_T], dot = 0.f, sq = 0.f;
#pragma unroll
        for (int c = 0; c < D_T; ++c) {
          v[c] = ld(nb, c * S + q);
          dot = fmaf(xc[c], v[c], dot);
          sq = fmaf(v[c], v[c], sq);
        }
        const float inv_q = inv_norm(sq, P.eps);
        const size_t in = (size_t)i * S + p;
        float g;
        if (EXPLICIT) {
          g = d_affs[kb + in];
        } else {
          const float m = mask ? (float)mask[(size_t)b * P.mbs + in] : 1.f;
          const float a = dot * inv_p * inv_q;
          g = P.gscale[i] * weight[(size_t)b * P.wbs + in] * m * (a * m - target[(size_t)b * P.tbs + in] * m);
        }
        g *= inv_q;
#pragma unroll
        for (int c = 0; c < D_T; ++c) G[c] = fmaf(g, v[c], G[c]);
      }
    }
    if (ROLE_B) {
      const int q = neighbour(P, z, y, x, -oz, -oy, -ox);
      if (q >= 0) {
        const T* nb = nbB + (size_t)b * D_T * S;
        float v[D_T], dot = 0.f, sq = 0.f;
#pragma unroll
        for (int c = 0; c < D_T; ++c) {
          v[c] = ld(nb, c * S + q);
          dot = fmaf(xc[c], v[c], dot);
          sq = fmaf(v[c], v[c], sq);
        }
        const float inv_q = inv_norm(sq, P.eps);
        const size_t in = (size_t)i * S + q;  // the loss term lives at the first operand's pixel
        float g;
        if (EXPLICIT) {
          g = d_affs[kb + in];
        } else {
          const float m = mask ? (float)mask[(size_t)b * P.mbs + in] : 1.f;
          const float a = dot * inv_p * inv_q;
          g = P.gscale[i] * weight[(size_t)b * P.wbs + in] * m * (a * m - target[(size_t)b * P.tbs + in] * m);
        }
        g *= inv_q;
#pragma unroll
        for (int c = 0; c < D_T; ++c) G[c] = fmaf(g, v[c], G[c]);
      }
    }
  }

  float proj = 0.f;
#pragma unroll
  for (int c = 0; c < D_T; ++c) proj = fmaf(xc[c] * inv_p, G[c], proj);
  if (nrm < P.eps) proj = 0.f;  // clamp_min branch of F.normalize: d ehat / d e = I / eps
  T* db = dx + (size_t)b * D_T * S;
  const float sc = dl * inv_p;
#pragma unroll
  for (int c = 0; c < D_T; ++c) st(db, c * S + p, (G[c] - xc[c] * inv_p * proj) * sc);
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
int validate(const PeaDesc* d) {
  if (!d) return PEA_E_NULL;
  if (d->abi != PEA_ABI_VERSION) return PEA_E_DESC;
  if (d->ndim != 2 && d->ndim != 3) return PEA_E_DESC;
  if (d->B < 1 || d->D < 1 || d->K < 1 || d->K > PEA_MAX_K) return PEA_E_DESC;
  for (int a = 0; a < 3; ++a)
    if (d->dims[a] < 1) return PEA_E_DESC;
  if (d->ndim == 2 && d->dims[0] != 1) return PEA_E_DESC;
  if (d->border != PEA_BORDER_CIRCULAR && d->border != PEA_BORDER_CROP_ZERO) return PEA_E_DESC;
  if (d->dtype != PEA_F32 && d->dtype != PEA_F16) return PEA_E_DESC;
  if (d->norm < PEA_NORM_BX || d->norm > PEA_NORM_FULL) return PEA_E_DESC;
  if (!(d->eps > 0.f)) return PEA_E_DESC;
  if (d->target_bstride < 0 || d->weight_bstride < 0 || d->mask_bstride < 0) return PEA_E_DESC;
  const long long S = (long long)d->dims[0] * d->dims[1] * d->dims[2];
  if (S > 0x7fffffffLL - kBlock) return PEA_E_UNSUPPORTED;
  if ((S + kBlock - 1) / kBlock * (long long)d->B > 0x7fffff00LL) return PEA_E_UNSUPPORTED;
  for (int i = 0; i < d->K; ++i)
    for (int a = 0; a < 3; ++a) {
      const int o = d->offsets[i][a];
      // |o| < dim: torch.roll would wrap further, but no reference stencil does; cropped slices need it
      if (o <= -d->dims[a] || o >= d->dims[a]) return PEA_E_DESC;
    }
  return PEA_OK;
}

KParams make_params(const PeaDesc* d) {
  KParams P;
  P.B = d->B; P.D = d->D; P.Z = d->dims[0]; P.Y = d->dims[1]; P.X = d->dims[2]; P.K = d->K;
  P.S = P.Z * P.Y * P.X;
  P.border = d->border; P.flags = d->flags; P.eps = d->eps;
  P.chunks = (P.S + kBlock - 1) / kBlock;
  P.tiles = P.B * P.chunks;
  P.tiles_per_xcd = (P.tiles + kXcd - 1) / kXcd;
  const long long dense = (long long)P.K * P.S;
  P.tbs = d->target_bstride ? d->target_bstride : dense;
  P.wbs = d->weight_bstride ? d->weight_bstride : dense;
  P.mbs = d->mask_bstride ? d->mask_bstride : dense;
  for (int i = 0; i < PEA_MAX_K; ++i) {
    const bool on = i < d->K;
    double n = 1.0;
    if (on) {
      if (d->norm == PEA_NORM_BX) n = (double)d->B * d->dims[2];
      else if (d->norm == PEA_NORM_FULL) n = (double)d->B * P.S;
      else {
        n = d->B;
        for (int a = 0; a < 3; ++a) n *= (double)(d->dims[a] - (d->offsets[i][a] < 0 ? -d->offsets[i][a] : d->offsets[i][a]));
      }
    }
    for (int a = 0; a < 3; ++a) P.off[i][a] = on ? d->offsets[i][a] : 0;
    P.lam[i] = on ? d->lambda[i] : 0.f;
    P.inv_n[i] = on ? (float)(1.0 / n) : 0.f;
    P.gscale[i] = on ? (float)(2.0 * (double)d->lambda[i] / n) : 0.f;
  }
  return P;
}

inline dim3 grid_of(const KParams& P) { return dim3((unsigned)(P.tiles_per_xcd * kXcd)); }

inline int hip_rc() {
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? PEA_OK : (int)e;
}

template <typename T, bool TRAIN>
int launch_fwd(const KParams& P, const void* e, const void* eo, const float* t, const float* w, const uint8_t* m,
               float* affs, float* partials, hipStream_t s) {
  const T* ep = (const T*)e;
  const T* op = eo ? (const T*)eo : ep;
  const size_t lds = TRAIN ? (size_t)P.K * kBlock * sizeof(float) : 0;
  const dim3 g = grid_of(P), blk(kBlock);
  switch (P.D) {
    case 16: hipLaunchKernelGGL((k_fwd_direct<T, 16, TRAIN>), g, blk, lds, s, P, ep, op, t, w, m, affs, partials); break;
    case 32: hipLaunchKernelGGL((k_fwd_direct<T, 32, TRAIN>), g, blk, lds, s, P, ep, op, t, w, m, affs, partials); break;
    case 64: hipLaunchKernelGGL((k_fwd_direct<T, 64, TRAIN>), g, blk, lds, s, P, ep, op, t, w, m, affs, partials); break;
    default: hipLaunchKernelGGL((k_fwd_direct<T, 0, TRAIN>), g, blk, lds, s, P, ep, op, t, w, m, affs, partials); break;
  }
  return hip_rc();
}

template <typename T, int D_T, bool EXPLICIT>
int launch_bwd_roles(const KParams& P, int roles, const T* x, const T* nbA, const T* nbB, const float* t,
                     const float* w, const uint8_t* m, const float* da, const float* dl, T* dx, hipStream_t s) {
  const dim3 g = grid_of(P), blk(kBlock);
  if (roles == 3) hipLaunchKernelGGL((k_bwd_direct<T, D_T, true, true, EXPLICIT>), g, blk, 0, s, P, x, nbA, nbB, t, w, m, da, dl, dx);
  else if (roles == 1) hipLaunchKernelGGL((k_bwd_direct<T, D_T, true, false, EXPLICIT>), g, blk, 0, s, P, x, nbA, nbB, t, w, m, da, dl, dx);
  else hipLaunchKernelGGL((k_bwd_direct<T, D_T, false, true, EXPLICIT>), g, blk, 0, s, P, x, nbA, nbB, t, w, m, da, dl, dx);
  return hip_rc();
}

template <typename T, bool EXPLICIT>
int launch_bwd(const KParams& P, int roles, const void* x, const void* nbA, const void* nbB, const float* t,
               const float* w, const uint8_t* m, const float* da, const float* dl, void* dx, hipStream_t s) {
  switch (P.D) {
    case 16: return launch_bwd_roles<T, 16, EXPLICIT>(P, roles, (const T*)x, (const T*)nbA, (const T*)nbB, t, w, m, da, dl, (T*)dx, s);
    case 32: return launch_bwd_roles<T, 32, EXPLICIT>(P, roles, (const T*)x, (const T*)nbA, (const T*)nbB, t, w, m, da, dl, (T*)dx, s);
    case 64: return launch_bwd_roles<T, 64, EXPLICIT>(P, roles, (const T*)x, (const T*)nbA, (const T*)nbB, t, w, m, da, dl, (T*)dx, s);
    case 4: return launch_bwd_roles<T, 4, EXPLICIT>(P, roles, (const T*)x, (const T*)nbA, (const T*)nbB, t, w, m, da, dl, (T*)dx, s);
    case 8: return launch_bwd_roles<T, 8, EXPLICIT>(P, roles, (const T*)x, (const T*)nbA, (const T*)nbB, t, w, m, da, dl, (T*)dx, s);
    default: return PEA_E_UNSUPPORTED;  // training needs D in {4, 8, 16, 32, 64}
  }
}

bool misaligned(const void* p, size_t a) { return ((uintptr_t)p & (a - 1)) != 0; }

template <bool EXPLICIT>
int bwd_common(const PeaDesc* desc, const void* e, const void* e_other, const float* target, const float* weight,
               const uint8_t* mask, const float* d_affs, const float* dloss, void* de, void* de_other, void* stream) {
  int rc = validate(desc);
  if (rc) return rc;
  if (!e || !de) return PEA_E_NULL;
  if (EXPLICIT ? !d_affs : (!target || !weight || !dloss)) return PEA_E_NULL;
  if (de_other && !e_other) return PEA_E_NULL;
  const size_t es = desc->dtype == PEA_F16 ? 2 : 4;
  if (misaligned(e, es) || misaligned(e_other, es) || misaligned(de, es) || misaligned(de_other, es) ||
      misaligned(target, 4) || misaligned(weight, 4) || misaligned(d_affs, 4) || misaligned(dloss, 4))
    return PEA_E_ALIGN;
  const KParams P = make_params(desc);
  hipStream_t s = (hipStream_t)stream;
  const bool h = desc->dtype == PEA_F16;
  if (!e_other) {
    return h ? launch_bwd<__half, EXPLICIT>(P, 3, e, e, e, target, weight, mask, d_affs, dloss, de, s)
             : launch_bwd<float, EXPLICIT>(P, 3, e, e, e, target, weight, mask, d_affs, dloss, de, s);
  }
  rc = h ? launch_bwd<__half, EXPLICIT>(P, 1, e, e_other, nullptr, target, weight, mask, d_affs, dloss, de, s)
         : launch_bwd<float, EXPLICIT>(P, 1, e, e_other, nullptr, target, weight, mask, d_affs, dloss, de, s);
  if (rc || !de_other) return rc;
  return h ? launch_bwd<__half, EXPLICIT>(P, 2, e_other, nullptr, e, target, weight, mask, d_affs, dloss, de_other, s)
           : launch_bwd<float, EXPLICIT>(P, 2, e_other, nullptr, e, target, weight, mask, d_affs, dloss, de_other, s);
}

}  // namespace

extern "C" {

int pea_version(void) { return PEA_ABI_VERSION; }

const char* pea_strerror(int code) {
  switch (code) {
    case PEA_OK: return "ok";
    case PEA_E_NULL: return "required pointer is NULL";
    case PEA_E_DESC: return "descriptor field out of range";
    case PEA_E_UNSUPPORTED: return "unsupported combination";
    case PEA_E_WORKSPACE: return "workspace missing or too small";
    case PEA_E_ALIGN: return "pointer not aligned to its element size";
    default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown pea error";
  }
}

int pea_desc_validate(const PeaDesc* desc) { return validate(desc); }

size_t pea_workspace_bytes(const PeaDesc* desc) {
  if (validate(desc)) return 0;
  const KParams P = make_params(desc);
  return (size_t)P.tiles * P.K * sizeof(float);
}

int pea_affinity_infer(const PeaDesc* desc, const void* e, const void* e_other, float* affs, void* stream) {
  const int rc = validate(desc);
  if (rc) return rc;
  if (!e || !affs) return PEA_E_NULL;
  const size_t es = desc->dtype == PEA_F16 ? 2 : 4;
  if (misaligned(e, es) || misaligned(e_other, es) || misaligned(affs, 4)) return PEA_E_ALIGN;
  const KParams P = make_params(desc);
  hipStream_t s = (hipStream_t)stream;
  return desc->dtype == PEA_F16
             ? launch_fwd<__half, false>(P, e, e_other, nullptr, nullptr, nullptr, affs, nullptr, s)
             : launch_fwd<float, false>(P, e, e_other, nullptr, nullptr, nullptr, affs, nullptr, s);
}

int pea_affinity_fwd(const PeaDesc* desc, const void* e, const void* e_other, const float* target,
                     const float* weight, const uint8_t* mask, float* affs, float* loss_out, void* workspace,
                     size_t workspace_bytes, void* stream) {
  int rc = validate(desc);
  if (rc) return rc;
  if (!e || !target || !weight || !loss_out) return PEA_E_NULL;
  const size_t es = desc->dtype == PEA_F16 ? 2 : 4;
  if (misaligned(e, es) || misaligned(e_other, es) || misaligned(affs, 4) || misaligned(target, 4) ||
      misaligned(weight, 4) || misaligned(loss_out, 4) || misaligned(workspace, 4))
    return PEA_E_ALIGN;
  const KParams P = make_params(desc);
  if (!workspace || workspace_bytes < (size_t)P.tiles * P.K * sizeof(float)) return PEA_E_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  float* partials = (float*)workspace;
  rc = desc->dtype == PEA_F16 ? launch_fwd<__half, true>(P, e, e_other, target, weight, mask, affs, partials, s)
                              : launch_fwd<float, true>(P, e, e_other, target, weight, mask, affs, partials, s);
  if (rc) return rc;
  hipLaunchKernelGGL(k_loss_finalize, dim3(1), dim3(1024), 0, s, P, partials, P.tiles, loss_out);
  return hip_rc();
}

int pea_affinity_bwd(const PeaDesc* desc, const void* e, const void* e_other, const float* target,
                     const float* weight, const uint8_t* mask, const float* dloss, void* de, void* de_other,
                     void* stream) {
  return bwd_common<false>(desc, e, e_other, target, weight, mask, nullptr, dloss, de, de_other, stream);
}

int pea_affinity_vjp(const PeaDesc* desc, const void* e, const void* e_other, const float* d_affs, void* de,
                     void* de_other, void* stream) {
  return bwd_common<true>(desc, e, e_other, nullptr, nullptr, nullptr, d_affs, nullptr, de, de_other, stream);
}

}  // extern "C"
