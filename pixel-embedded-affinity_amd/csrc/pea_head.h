// pea_head.h -- the per-pixel embedding head: the 1x1 (1x1x1) convolution that turns the decoder's C feature channels
// into the D-dimensional embedding, e[b,d,p] = bias[d] + sum_c W[d,c] * x[b,c,p], and its backward.
// Reference: OutConv, scripts_cvppp/model/unet2d_residual.py:67-74 (outconv_emb :307, applied :346; same class in
// scripts_bbbc039v1/model/unet2d_residual.py:67,235); 3D: conv3dBlock(.., [(1,1,1)]) out_put*,
// scripts_ac3ac4/model/model_superhuman.py:437-441, applied :486-490.  The step immediately before the path
// (SURVEY.md section 8f, f1).  Included by pea_hip.hip only.
//
// Both kernels stream: one lane per pixel, planar [B,C,S] / [B,D,S] tensors, every access a coalesced 256-byte row of
// one channel.  They are HBM-bound (forward 4(C+D) bytes per pixel, backward 4(2C+D)); the arithmetic (2CD flops per
// pixel forward, 4CD backward) is a fifth of the memory time.  W is read through the scalar cache (uniform addresses,
// compile-time offsets), so the forward and dx are plain v_fmac with an SGPR operand.  dW = sum_p de[:,p] x[:,p]^T is
// the one contraction over PIXELS here and goes to the matrix cores: v_mfma_f32_16x16x4_f32 (exact f32, k-ordered fma
// chain) with A = de[16 d x 4 px], B = x[4 px x 16 c]; the pixel-per-lane registers are turned into that operand
// layout through a per-wave LDS tile (row stride 68 floats: both the pixel-major write and the (l%16, l/16) read are
// bank-conflict free).  Deterministic: lane-private / wave-private accumulators, per-workgroup partial sums, fixed-order
// final reduction.
#pragma once
#include <hip/hip_runtime.h>
#include "pea_tiled.h"  // wave_sum63

namespace pea {

typedef float hv4 __attribute__((ext_vector_type(4)));
constexpr int kHeadBlock = 256;    // 4 waves
constexpr int kHeadMaxWg = 1024;   // backward: workgroups (= partial sums of dW / db)
constexpr int kHeadRow = 68;       // LDS row stride (floats) of a [channel][64 px] tile

template <int C, int D>
__global__ __launch_bounds__(kHeadBlock) void k_head_fwd(const float* __restrict__ x, const float* __restrict__ W,
                                                         const float* __restrict__ bias, float* __restrict__ e, long long S,
                                                         int chunks_per_b) {
  const int b = blockIdx.x / chunks_per_b;
  const long long p = (long long)(blockIdx.x - b * chunks_per_b) * kHeadBlock + threadIdx.x;
  if (p >= S) return;
  const float* xb = x + (size_t)b * C * S + p;
  float* eb = e + (size_t)b * D * S + p;
  float xv[C];
#pragma unroll
  for (int c = 0; c < C; ++c) xv[c] = __builtin_nontemporal_load(xb + (size_t)c * S);
  const bool has_bias = bias != nullptr;
#pragma unroll
  for (int d = 0; d < D; ++d) {
    float a = has_bias ? bias[d] : 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) a = fmaf(W[d * C + c], xv[c], a);
    eb[(size_t)d * S] = a;  // the affinity kernels read it next: keep it in the caches
  }
}

// dx[b,c,p] = sum_d W[d,c] de[b,d,p] (nullable: the input may not need a gradient);
// partials[wg][D*C + D] = this workgroup's share of dW[d,c] = sum_{b,p} de[b,d,p] x[b,c,p] and db[d] = sum_{b,p} de[b,d,p]
template <int C, int D>
__global__ __launch_bounds__(kHeadBlock) void k_head_bwd(const float* __restrict__ x, const float* __restrict__ W,
                                                         const float* __restrict__ de, float* __restrict__ dx,
                                                         float* __restrict__ partials, long long S, int chunks_per_b,
                                                         int nchunks) {
  static_assert(D % 16 == 0, "the dW tiles are 16 x 16");
  constexpr int DT = D / 16, CC = (C + 15) / 16, NW = kHeadBlock / 64;
  constexpr int kTileA = D * kHeadRow, kTileB = 16 * kHeadRow;
  constexpr int kRed = D * CC * 16;  // one wave's dW tile set, [d][16*CC]
  // per wave: A tile (de, [D][68]) + B tile (16 channels of x, [16][68]); reused as [NW][kRed] for the final reduction
  constexpr int kLds = (NW * (kTileA + kTileB) > NW * kRed + NW * D) ? NW * (kTileA + kTileB) : NW * kRed + NW * D;
  __shared__ float lds[kLds];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* tA = lds + wave * (kTileA + kTileB);
  float* tB = tA + kTileA;
  const int mi = lane & 15, mk = lane >> 4;  // MFMA operand coordinates of this lane: row / column mi, k index mk

  hv4 acc[DT][CC];
#pragma unroll
  for (int i = 0; i < DT; ++i)
#pragma unroll
    for (int j = 0; j < CC; ++j) acc[i][j] = hv4{0.f, 0.f, 0.f, 0.f};
  float dbv[D];
#pragma unroll
  for (int d = 0; d < D; ++d) dbv[d] = 0.f;
  const bool want_dx = dx != nullptr;

  for (int chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {  // uniform trip count per workgroup
    const int b = chunk / chunks_per_b;
    const long long p = (long long)(chunk - b * chunks_per_b) * kHeadBlock + threadIdx.x;
    const bool live = p < S;
    const size_t pc = live ? (size_t)p : 0;
    const float* deb = de + (size_t)b * D * S + pc;
    const float* xb = x + (size_t)b * C * S + pc;
    // every load of the iteration is requested up front (addresses are clamped, values masked afterwards): one memory
    // round trip per 256 pixels with D + C rows in flight per wave
    float dv[D], xv[C];
#pragma unroll
    for (int d = 0; d < D; ++d) dv[d] = __builtin_nontemporal_load(deb + (size_t)d * S);
#pragma unroll
    for (int c = 0; c < C; ++c) xv[c] = __builtin_nontemporal_load(xb + (size_t)c * S);
#pragma unroll
    for (int d = 0; d < D; ++d) {
      dv[d] = live ? dv[d] : 0.f;
      dbv[d] += dv[d];
      tA[d * kHeadRow + lane] = dv[d];
    }
    if (want_dx) {
      float dxv[C];
#pragma unroll
      for (int c = 0; c < C; ++c) dxv[c] = 0.f;
#pragma unroll
      for (int d = 0; d < D; ++d)
#pragma unroll
        for (int c = 0; c < C; ++c) dxv[c] = fmaf(W[d * C + c], dv[d], dxv[c]);
      if (live) {
        float* dxb = dx + (size_t)b * C * S + pc;
#pragma unroll
        for (int c = 0; c < C; ++c) __builtin_nontemporal_store(dxv[c], dxb + (size_t)c * S);
      }
    }
    __builtin_amdgcn_wave_barrier();
    // A operands of the 16 k-steps (4 pixels each), kept for every channel chunk
    float av[DT][16];
#pragma unroll
    for (int i = 0; i < DT; ++i)
#pragma unroll
      for (int s = 0; s < 16; ++s) av[i][s] = tA[(16 * i + mi) * kHeadRow + 4 * s + mk];
#pragma unroll
    for (int j = 0; j < CC; ++j) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c = 16 * j + r;
        tB[r * kHeadRow + lane] = (c < C && live) ? xv[c < C ? c : 0] : 0.f;
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const float bv = tB[mi * kHeadRow + 4 * s + mk];
#pragma unroll
        for (int i = 0; i < DT; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i][s], bv, acc[i][j], 0, 0, 0);
      }
      __builtin_amdgcn_wave_barrier();  // the next chunk's writes come after these reads
    }
  }

  // ---- workgroup partial: sum the waves' tiles in wave order
  __syncthreads();
  float* red = lds;                // [NW][kRed]
  float* redb = lds + NW * kRed;   // [NW][D]
#pragma unroll
  for (int i = 0; i < DT; ++i)
#pragma unroll
    for (int j = 0; j < CC; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r)  // C/D layout: row = 4 * (lane >> 4) + r, column = lane & 15
        red[wave * kRed + (16 * i + 4 * mk + r) * (16 * CC) + 16 * j + mi] = acc[i][j][r];
#pragma unroll
  for (int d = 0; d < D; ++d) {
    const float s = wave_sum63(dbv[d]);
    if (lane == 63) redb[wave * D + d] = s;
  }
  __syncthreads();
  float* out = partials + (size_t)blockIdx.x * (D * C + D);
  for (int t = threadIdx.x; t < D * C; t += kHeadBlock) {
    const int d = t / C, c = t - d * C;
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) s += red[w * kRed + d * (16 * CC) + c];
    out[t] = s;
  }
  if (threadIdx.x < D) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) s += redb[w * D + threadIdx.x];
    out[D * C + threadIdx.x] = s;
  }
}

// dW[d,c], db[d] = sum over the workgroup partials, in workgroup order
__global__ __launch_bounds__(256) void k_head_finalize(const float* __restrict__ partials, int nwg, int dc, int n,
                                                       float* __restrict__ dW, float* __restrict__ db) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= n) return;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;  // four interleaved chains: shorter dependency, still a fixed order
  int w = 0;
  for (; w + 3 < nwg; w += 4) {
    s0 += partials[(size_t)w * n + t];
    s1 += partials[(size_t)(w + 1) * n + t];
    s2 += partials[(size_t)(w + 2) * n + t];
    s3 += partials[(size_t)(w + 3) * n + t];
  }
  for (; w < nwg; ++w) s0 += partials[(size_t)w * n + t];
  const float s = (s0 + s1) + (s2 + s3);
  if (t < dc) dW[t] = s;
  else if (db) db[t - dc] = s;
}

}  // namespace pea
