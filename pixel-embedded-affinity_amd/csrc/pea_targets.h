// pea_targets.h -- label image -> target / mask / class-balance weight on the GPU.
//
// What the reference's data providers compute per sample in DataLoader workers with scipy.ndimage.shift and numpy
// (scripts_cvppp/data/data_provider.py:204-225 -> utils/affinity_ours.py:17-39 gen_affs_ours and
// data/data_segmentation.py:205-228 weight_binary_ratio), then ship to the GPU as ~40 MB of f32 per 544^2 sample.
// From an int32 label image it is a trivially parallel integer stencil plus one count per (sample, channel):
//   t_i(p) = [label(p) == label(p + o_i)]  (and both > 0 with PEA_TGT_BOTH_FOREGROUND: seg_to_aff);
//   a neighbour outside the image: mask = 0 and t = 1 (PEA_TGT_PADDING) or 0;
//   f_i = clip(count(t_i != 0) / (Z*Y*X), 0.05, 0.99); the minority class gets max(f,1-f)/min(f,1-f), the majority 1;
//   a single-valued channel gets weight 1 everywhere.
// Integer counts (one u32 atomic per workgroup and channel): bit-reproducible.  The ratio is evaluated in f64 like
// numpy does and rounded to f32 once.
#pragma once
#include "pea_direct.h"

namespace pea {

constexpr int kTgtNit = 8;  // pixels per lane in the label kernels (2048 per workgroup)

struct GParams {
  int B, Z, Y, X, K, S;
  unsigned flags;
  int off[PEA_MAX_K][3];
};

static __global__ __launch_bounds__(256) void k_gen_targets(const GParams G, const int32_t* __restrict__ labels,
                                                     float* __restrict__ target, uint8_t* __restrict__ mask,
                                                     unsigned* __restrict__ counts) {
  // NIT pixels per lane, decoded once; per channel NIT independent neighbour loads in flight; counts go wave ballot ->
  // LDS -> K global atomics per workgroup (one atomic per 256 pixels and channel onto B*K addresses cost 90 us)
  __shared__ unsigned s_cnt[PEA_MAX_K];
  constexpr int NIT = kTgtNit;
  const int b = blockIdx.y;
  if (threadIdx.x < PEA_MAX_K) s_cnt[threadIdx.x] = 0;
  __syncthreads();
  const int yx = G.Y * G.X;
  const int32_t* lb = labels + (size_t)b * G.S;
  const bool pad = G.flags & PEA_TGT_PADDING, fg = G.flags & PEA_TGT_BOTH_FOREGROUND;
  int pp[NIT], pz[NIT], py[NIT], px[NIT], pa[NIT];
#pragma unroll
  for (int j = 0; j < NIT; ++j) {
    const int p = (blockIdx.x * NIT + j) * 256 + (int)threadIdx.x;
    const bool live = p < G.S;
    pp[j] = live ? p : -1;
    pz[j] = live ? p / yx : 0;
    const int r = live ? p - pz[j] * yx : 0;
    py[j] = r / G.X;
    px[j] = r - py[j] * G.X;
    pa[j] = live ? lb[p] : 0;
  }
  constexpr int CH = 4;  // channels per step: CH * NIT independent neighbour loads in flight per lane
  for (int i0 = 0; i0 < G.K; i0 += CH) {
    int nb[CH][NIT];
    bool in[CH][NIT];
#pragma unroll
    for (int u = 0; u < CH; ++u) {
      const int i = min(i0 + u, G.K - 1);
      const int oz = G.off[i][0], oy = G.off[i][1], ox = G.off[i][2];
#pragma unroll
      for (int j = 0; j < NIT; ++j) {
        const int zz = pz[j] + oz, yy = py[j] + oy, xx = px[j] + ox;
        in[u][j] = pp[j] >= 0 && (unsigned)zz < (unsigned)G.Z && (unsigned)yy < (unsigned)G.Y && (unsigned)xx < (unsigned)G.X;
        nb[u][j] = lb[in[u][j] ? (zz * G.Y + yy) * G.X + xx : 0];
      }
    }
#pragma unroll
    for (int u = 0; u < CH; ++u) {
      const int i = i0 + u;
      const bool on = i < G.K;  // uniform
      unsigned c = 0;
#pragma unroll
      for (int j = 0; j < NIT; ++j) {
        const bool t = in[u][j] ? (pa[j] == nb[u][j] && (!fg || (pa[j] > 0 && nb[u][j] > 0))) : pad;
          if (on && pp[j] >= 0) {
            const size_t o = ((size_t)b * G.K + i) * G.S + pp[j];
            target[o] = t ? 1.f : 0.f;
            if (mask) mask[o] = in[u][j] ? 1 : 0;
          }
        c += (unsigned)__popcll(__ballot(pp[j] >= 0 && t));
      }
      if (on && (threadIdx.x & 63) == 0 && c) atomicAdd(&s_cnt[i], c);
    }
  }
  __syncthreads();
  if (threadIdx.x < G.K && s_cnt[threadIdx.x]) atomicAdd(&counts[b * G.K + threadIdx.x], s_cnt[threadIdx.x]);
}

static __global__ __launch_bounds__(256) void k_gen_weights(const GParams G, const float* __restrict__ target,
                                                     const unsigned* __restrict__ counts, float* __restrict__ weight) {
  const int bi = blockIdx.y;  // b * K + i
  const unsigned n = counts[bi];
  float wpos = 1.f, wneg = 1.f;
  if (n != 0 && n != (unsigned)G.S) {
    double f = (double)n / (double)G.S;
    f = fmin(fmax(f, 5e-2), 0.99);
    if (f > 0.5) wneg = (float)(f / (1.0 - f));
    else wpos = (float)((1.0 - f) / f);
  }
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p < G.S) {
    const size_t o = (size_t)bi * G.S + p;
    weight[o] = target[o] != 0.f ? wpos : wneg;
  }
}

}  // namespace pea
