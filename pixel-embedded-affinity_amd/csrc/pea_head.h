// pea_head.h -- the per-pixel embedding head: the 1x1 (1x1x1) convolution that turns the decoder's C feature channels
// into the D-dimensional embedding, e[b,d,p] = bias[d] + sum_c W[d,c] * x[b,c,p], and its backward.
// Reference: OutConv, scripts_cvppp/model/unet2d_residual.py:67-74 (outconv_emb :307, applied :346; same class in
// scripts_bbbc039v1/model/unet2d_residual.py:67,235); 3D: conv3dBlock(.., [(1,1,1)]) out_put*,
// scripts_ac3ac4/model/model_superhuman.py:437-441, applied :486-490.  The step immediately before the path
// (SURVEY.md section 8f, f1).  Included by pea_hip.hip only.
//
// Both kernels stream: one lane per pixel, planar [B,C,S] / [B,D,S] tensors, every access a coalesced 256-byte row of
// one channel.  They are HBM-bound (forward 4(C+D) bytes per pixel, backward 4(2C+D)); the arithmetic (2CD flops per
// pixel forward, 4CD backward) is a fraction of the memory time.  W is read through the scalar cache (uniform addresses,
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
constexpr int kHeadCB = 64;        // input channels held in registers at a time (the wide low-resolution heads are chunked)

template <int C, int D>
__global__ __launch_bounds__(kHeadBlock) void k_head_fwd(const float* __restrict__ x, const float* __restrict__ W,
                                                         const float* __restrict__ bias, float* __restrict__ e, long long S,
                                                         int chunks_per_b) {
  const int b = blockIdx.x / chunks_per_b;
  const long long p = (long long)(blockIdx.x - b * chunks_per_b) * kHeadBlock + threadIdx.x;
  if (p >= S) return;
  const float* xb = x + (size_t)b * C * S + p;
  float* eb = e + (size_t)b * D * S + p;
  const bool has_bias = bias != nullptr;
  if (C <= kHeadCB) {
    float xv[C];
#pragma unroll
    for (int c = 0; c < C; ++c) xv[c] = __builtin_nontemporal_load(xb + (size_t)c * S);
#pragma unroll
    for (int d = 0; d < D; ++d) {
      float a = has_bias ? bias[d] : 0.f;
#pragma unroll
      for (int c = 0; c < C; ++c) a = fmaf(W[d * C + c], xv[c], a);
      eb[(size_t)d * S] = a;  // the affinity kernels read it next: keep it in the caches
    }
  } else {  // wide heads (C = 80 / 128 / 256 at the coarse scales): D accumulators, kHeadCB channels at a time
    float a[D];
#pragma unroll
    for (int d = 0; d < D; ++d) a[d] = has_bias ? bias[d] : 0.f;
    for (int c0 = 0; c0 < C; c0 += kHeadCB) {
      constexpr int kN = kHeadCB;
      float xv[kN];
#pragma unroll
      for (int c = 0; c < kN; ++c) xv[c] = (c0 + c < C) ? __builtin_nontemporal_load(xb + (size_t)(c0 + c) * S) : 0.f;
#pragma unroll
      for (int d = 0; d < D; ++d)
#pragma unroll
        for (int c = 0; c < kN; ++c)
          if (c0 + c < C) a[d] = fmaf(W[d * C + c0 + c], xv[c], a[d]);
    }
#pragma unroll
    for (int d = 0; d < D; ++d) eb[(size_t)d * S] = a[d];
  }
}

// dx[b,c,p] = sum_d W[d,c] de[b,d,p]: a streaming kernel of its own (like the forward: loads, 2CD flops, stores, exit).
// Fused with the dW kernel below it ran 1.6x slower than the two together: vmcnt retires in order, so the next
// pixels' loads of a looping workgroup wait for the acknowledgements of the dx stores before them.
template <int C, int D>
__global__ __launch_bounds__(kHeadBlock) void k_head_dx(const float* __restrict__ W, const float* __restrict__ de,
                                                        float* __restrict__ dx, long long S, int chunks_per_b) {
  const int b = blockIdx.x / chunks_per_b;
  const long long p = (long long)(blockIdx.x - b * chunks_per_b) * kHeadBlock + threadIdx.x;
  if (p >= S) return;
  const float* deb = de + (size_t)b * D * S + p;
  float* dxb = dx + (size_t)b * C * S + p;
  float dv[D];
#pragma unroll
  for (int d = 0; d < D; ++d) dv[d] = deb[(size_t)d * S];  // read again by k_head_dw: no non-temporal hint
  for (int c0 = 0; c0 < C; c0 += kHeadCB) {  // one pass for C <= kHeadCB
    constexpr int kN = C < kHeadCB ? C : kHeadCB;
    float dxv[kN];
#pragma unroll
    for (int c = 0; c < kN; ++c) dxv[c] = 0.f;
#pragma unroll
    for (int d = 0; d < D; ++d)
#pragma unroll
      for (int c = 0; c < kN; ++c)
        if (c0 + c < C) dxv[c] = fmaf(W[d * C + c0 + c], dv[d], dxv[c]);
#pragma unroll
    for (int c = 0; c < kN; ++c)
      if (c0 + c < C) __builtin_nontemporal_store(dxv[c], dxb + (size_t)(c0 + c) * S);
  }
}

// partials[wg][D*C + D] = this workgroup's share of dW[d,c] = sum_{b,p} de[b,d,p] x[b,c,p] and db[d] = sum_{b,p} de[b,d,p].
// Loads only.  What it needs is rows in flight (HBM latency under load is several times the ~1 us a wave spends on 256
// pixels): D + C rows per wave, four workgroups per CU where the registers allow (C <= 48, D = 16).  Prefetching the next
// 256 pixels instead (twice the registers, half the waves) was slower: 114 vs 9x us.
template <int C, int D>
__global__ __launch_bounds__(kHeadBlock, (C <= 48 && D == 16) ? 4 : (C <= 128 ? 2 : 1)) void k_head_dw(const float* __restrict__ x, const float* __restrict__ de,
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

  for (int chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {  // uniform trip count per workgroup
    const int b = chunk / chunks_per_b;
    const long long p = (long long)(chunk - b * chunks_per_b) * kHeadBlock + threadIdx.x;
    const bool live = p < S;
    const size_t pc = live ? (size_t)p : 0;  // clamped address, value masked below: no branches around the loads
    const float* deb = de + (size_t)b * D * S + pc;
    const float* xb = x + (size_t)b * C * S + pc;
    constexpr bool kAll = C <= kHeadCB;  // every row of the 256 pixels requested up front (else 16 channels at a time)
    float dv[D], xv[kAll ? C : 16];
#pragma unroll
    for (int d = 0; d < D; ++d) dv[d] = __builtin_nontemporal_load(deb + (size_t)d * S);
    if (kAll) {
#pragma unroll
      for (int c = 0; c < C; ++c) xv[c] = __builtin_nontemporal_load(xb + (size_t)c * S);
    }
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const float v = live ? dv[d] : 0.f;
      dbv[d] += v;
      tA[d * kHeadRow + lane] = v;
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
        if (!kAll) xv[r] = c < C ? __builtin_nontemporal_load(xb + (size_t)c * S) : 0.f;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c = 16 * j + r;
        tB[r * kHeadRow + lane] = (c < C && live) ? xv[kAll ? (c < C ? c : 0) : r] : 0.f;
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

// dW[d,c], db[d] = sum over the workgroup partials in a fixed order.  One workgroup per 16 outputs: thread (o, g) adds
// the partials of workgroups g, g + 16, g + 32, ... (64-byte coalesced reads, nwg / 16 loads per thread instead of nwg:
// a thread per output walking all 1024 partials took 75 us, longer than the kernel that produced them), then the 16
// group sums are added in group order.
static __global__ __launch_bounds__(256) void k_head_finalize(const float* __restrict__ partials, int nwg, int dc, int n,
                                                       float* __restrict__ dW, float* __restrict__ db) {
  __shared__ float red[16][17];
  const int o = threadIdx.x & 15, g = threadIdx.x >> 4;
  const int t = blockIdx.x * 16 + o;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;  // four independent chains: loads in flight, still a fixed order
  if (t < n) {
    int w = g;
    for (; w + 48 < nwg; w += 64) {
      s0 += partials[(size_t)w * n + t];
      s1 += partials[(size_t)(w + 16) * n + t];
      s2 += partials[(size_t)(w + 32) * n + t];
      s3 += partials[(size_t)(w + 48) * n + t];
    }
    for (; w < nwg; w += 16) s0 += partials[(size_t)w * n + t];
  }
  red[g][o] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (threadIdx.x < 16 && t < n) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += red[k][o];
    if (t < dc) dW[t] = s;
    else if (db) db[t - dc] = s;
  }
}

}  // namespace pea
