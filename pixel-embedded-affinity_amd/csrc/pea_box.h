// pea_box.h -- forward and backward for stencils inside the UNIT BOX (every offset has |dz|, |dy|, |dx| <= 1: the 26-neighbourhood
// BASELINE.json configs[3] names for the AC3/AC4 sub-volumes, and any subset of it), served from LDS, staged by LDS-DMA.
//
// Round 2 ran this stencil on the box kernels of pea_tiled.h: they stage ONE plane (all 16 channels of an 18 x 34 region, 39 KB)
// and gather the 18 neighbours in the planes above and below from global memory, 16 one-dword loads per (pixel, neighbour):
// 4.9 + 6.1 ms on a 24 x 1024^2 sub-volume, 0.17 of the HBM roofline.  Here the channels go through LDS TWO AT A TIME like in the
// cross kernels (pea_xdma.h), but a chunk is the 3-plane box: for each of the two channels the region rows y0-1 .. y0+16, columns
// x0-4 .. x0+35 (whole quads, 16-byte aligned) of planes z-1, z, z+1 -- 6 x 720 floats = 17 KB, moved by 17 wave-level
// buffer_load_dwordx4 ... lds into a ring of three chunk buffers.  Every one of the 26 neighbours is then an LDS read at a
// compile-time displacement from the lane's own slot: the kernels walk ALL 26 displacements (an offset table is a subset of them;
// the host hands over the displacement of every offset), so there is no per-offset address register at all.
//   forward:  dot[26] (packed over the two channels of the chunk); the squared norms are not accumulated per neighbour (26 more
//             accumulators) but once per REGION pixel -- every lane owns 4-5 of the 2160 region pixels -- and exchanged through
//             LDS after the channel loop; a = dot * inv(p) * inv(q) parked as [displacement][tile pixel] over the dead ring and
//             walked four x-adjacent pixels per lane (dwordx4 target / weight / affs / g) like k_fwd_xdma's epilogue.
//   backward: the neighbour at displacement d carries the coefficient g_{o = d}(p) + g_{o = -d}(p + d) (role A of the offset d and
//             role B of the offset -d), times 1 / |e(p + d)| from the staged 1 / norm plane of the forward: 26 coefficients, two to
//             a register pair (pk_fma_c), G(p) = sum_d c_d e(p + d) per channel pair, d e = (G - ehat <ehat, G>) / |e|.
// LDS: forward 3 x 17 KB ring, the parked dot products over it, 9 KB of region norms = 63 KB (two workgroups per CU); backward the
// ring alone, 52 KB (three workgroups per CU at 72 VGPRs: the staged 1 / norm region starts out in the third chunk buffer).
// f32 storage, D = 16, X % 4 == 0, 16-byte aligned planes; CIRCULAR and CROP_ZERO borders.
#pragma once
#include "pea_xdma.h"

namespace pea {

constexpr int kBoxTH = 16, kBoxTW = 32;
constexpr int kBoxRW = 40, kBoxRH = 18;         // region: columns x0 - 4 .. x0 + 35, rows y0 - 1 .. y0 + 16
constexpr int kBoxRP = kBoxRW * kBoxRH;         // 720 region pixels per plane
constexpr int kBoxR3 = 3 * kBoxRP;              // 2160: planes z - 1, z, z + 1
constexpr int kBoxCQ = 2 * kBoxR3 / 4;          // 1080 quads per chunk: [channel j][plane dz][row][col]
constexpr int kBoxNB = (kBoxCQ + 63) / 64;      // 17 blocks of 64 quads (1 KB each); waves take blocks i * 8 + wave
constexpr int kBoxCB = kBoxNB * 1024;           // 17408 bytes per chunk buffer (the last block's tail is padding)
constexpr int kBoxJB = kBoxR3 * 4;              // 8640: byte distance of the chunk's second channel
constexpr int kBoxNQ = kBoxR3 / 4;              // 540 quads of the 1 / norm region
constexpr int kBoxNNB = (kBoxNQ + 63) / 64;     // 9 blocks
constexpr int kBoxND = 26;                      // displacements (the centre excluded)
constexpr int kBoxSA = kBoxND * kBoxTH * kBoxTW * 4;  // 53248: parked a[displacement][tile pixel]
constexpr int kBoxSN = kBoxSA;                  // 1 / norm of the region pixels (2160 floats, 9 blocks reserved)
constexpr int kBoxSP = kBoxSN + kBoxNNB * 1024; // loss partials [K][2]
constexpr int kBoxLds = kBoxSP + 256;           // 62720 bytes
static_assert(3 * kBoxCB <= kBoxSA, "the ring lies under the parked dot products");

// displacement slot s in [0, 26) <-> (dz, dy, dx): index (dz+1)*9 + (dy+1)*3 + (dx+1) with the centre (13) skipped
__host__ __device__ constexpr int box_di(int s) { return s < 13 ? s : s + 1; }
__host__ __device__ constexpr int box_dz(int s) { return box_di(s) / 9 - 1; }
__host__ __device__ constexpr int box_dy(int s) { return box_di(s) / 3 % 3 - 1; }
__host__ __device__ constexpr int box_dx(int s) { return box_di(s) % 3 - 1; }
// byte displacement of slot s from the lane's own slot in plane z (one channel)
__host__ __device__ constexpr int box_lds(int s) { return box_dz(s) * kBoxRP * 4 + (box_dy(s) * kBoxRW + box_dx(s)) * 4; }
constexpr int kBoxBias = kBoxRP * 4 + (kBoxRW + 1) * 4;  // 3044 = -box_lds(0): the read base is biased so every immediate is >= 0

struct BParams {
  int tiles_y, tiles_x, tiles_per_plane, ntiles, tiles_per_xcd;
  int rev, zrun, zgy, zgx, sup_y, sup_x;  // the tile walk of pea_xdma.h (xdma_tile)
  int slot[PEA_MAX_K];         // offset k -> displacement slot
  int kA[kBoxND], kB[kBoxND];  // slot -> offset with o == d (role A) / o == -d (role B), or -1
  float gs[PEA_MAX_K];
};

// this lane's DMA items of a chunk: three quads (the third only in wave 0).  Byte offset inside the batch item's [D][Z][Y][X]
// block, channel j and plane dz folded in (they differ between the lanes of a block); kOOB outside the volume (CROP) / past the
// chunk's 1080 quads: the DMA then writes zeros.
template <bool CROP>
__device__ __forceinline__ unsigned box_item(const KParams& P, int q, int nq_plane, int z, int y0, int x0, bool with_channel) {
  const int pc = q / (kBoxRP / 4), r = q - pc * (kBoxRP / 4);
  const int j = with_channel ? pc / 3 : 0, dzi = with_channel ? pc - j * 3 : pc;
  const int row = r / (kBoxRW / 4), qd = r - row * (kBoxRW / 4);
  bool okz, oky, okx;
  const int gz = wrap1<CROP>(z + dzi - 1, P.Z, okz), gy = wrap1<CROP>(y0 - 1 + row, P.Y, oky), gx = wrap1<CROP>(x0 - 4 + 4 * qd, P.X, okx);
  const bool ok = q < nq_plane && okz && oky && okx && dzi < 3;
  return ok ? (unsigned)(((j * P.Z + gz) * P.Y + gy) * P.X + gx) * 4u : kOOB;  // j * S + voxel < 2^29 (host-checked)
}

#define PEA_BWAIT(n) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(n) : "memory")
// wait until only the youngest chunk's DMA (3 wave instructions in wave 0, 2 in the others) may be in flight, then the barrier
#define PEA_BWAIT1()                   \
  {                                    \
    if (wave == 0) PEA_BWAIT(3);       \
    else PEA_BWAIT(2);                 \
  }
#define PEA_BDMA(buf, so)                                                                                                      \
  {                                                                                                                            \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(xB, (lds_ptr_t)(lds + (buf) * kBoxCB + wave * 1024), 16, vo0, so, 0, 0);           \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(xB, (lds_ptr_t)(lds + (buf) * kBoxCB + (8 + wave) * 1024), 16, vo1, so, 0, 0);     \
    if (wave == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(xB, (lds_ptr_t)(lds + (buf) * kBoxCB + 16 * 1024), 16, vo2, so, 0, 0); \
  }

// ------------------------------------------------------------------------------------------------------------------
// forward (training: affs, g, loss, 1 / norm plane; inference: affs)
// ------------------------------------------------------------------------------------------------------------------
template <int D_T, bool CROP, bool TRAIN>
__global__ __launch_bounds__(kBoxTH* kBoxTW, 4) void k_fwd_box(const KParams P, const BParams C, const float* __restrict__ e,
                                                               const float* __restrict__ target, const float* __restrict__ weight,
                                                               const uint8_t* __restrict__ mask, float* __restrict__ affs,
                                                               float* __restrict__ gout, LossState* __restrict__ st,
                                                               float* __restrict__ inv_out) {
  constexpr int TH = kBoxTH, TW = kBoxTW, NT = TH * TW, NP = D_T / 2, TP = NT, QP = TP / 4, NSL = QP / 64;
  static_assert(D_T % 2 == 0 && NP >= 3, "channel pairs through a ring of three");
  extern __shared__ f4 lds4[];
  char* lds = (char*)lds4;
  float* sA = (float*)lds;
  float* sN = (float*)(lds + kBoxSN);
  float* s_part = (float*)(lds + kBoxSP);
  int tile, b, z, y0, x0;
  if (!march_tile<TH, TW, BParams>(C, P, tile, b, z, y0, x0)) return;
  const size_t S = (size_t)P.S;
  const rsrc_t xB = mkbuf(e + (size_t)b * D_T * S);
  const unsigned ecs = (unsigned)P.S * 4u;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const unsigned vo0 = box_item<CROP>(P, wave * 64 + lane, kBoxCQ, z, y0, x0, true);
  const unsigned vo1 = box_item<CROP>(P, (8 + wave) * 64 + lane, kBoxCQ, z, y0, x0, true);
  const unsigned vo2 = box_item<CROP>(P, 16 * 64 + lane, kBoxCQ, z, y0, x0, true);
  PEA_BDMA(0, 0u)
  PEA_BDMA(1, 2u * ecs)
  const int ly = threadIdx.x >> 5, lx = threadIdx.x & 31;
  const int own = kBoxRP * 4 + ((ly + 1) * kBoxRW + lx + 4) * 4;  // the lane's pixel in plane z of channel 0
  const char* const rb = lds + own - kBoxBias;                    // read base: rb + kBoxBias + box_lds(s)
  PEA_BWAIT1()
  PEA_BDMA(2, 4u * ecs)

  f2 dot[kBoxND], oss = {0.f, 0.f}, rss[5];
#pragma unroll
  for (int s = 0; s < kBoxND; ++s) dot[s] = (f2){0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 5; ++i) rss[i] = (f2){0.f, 0.f};
  const bool own5 = (int)threadIdx.x + 4 * NT < kBoxR3;
  const int r5 = min((int)threadIdx.x + 4 * NT, kBoxR3 - 1) * 4;
#pragma unroll
  for (int ps = 0; ps < NP; ++ps) {
    const int bo = (ps % 3) * kBoxCB;
    f2 o;
    o.x = *(const float*)(lds + bo + own);
    o.y = *(const float*)(lds + bo + kBoxJB + own);
    oss = __builtin_elementwise_fma(o, o, oss);
#pragma unroll
    for (int s = 0; s < kBoxND; ++s) {
      f2 v;
      v.x = *(const float*)(rb + bo + kBoxBias + box_lds(s));
      v.y = *(const float*)(rb + bo + kBoxJB + kBoxBias + box_lds(s));
      dot[s] = __builtin_elementwise_fma(o, v, dot[s]);
      if (s % 6 == 5) asm volatile("" ::: "memory");  // bound the ds_read hoisting
    }
    // the squared norms of the region pixels this lane owns (region pixel t + 512 i of the three planes)
#pragma unroll
    for (int i = 0; i < 5; ++i) {  // (the fifth one of a lane without a fifth pixel: the region's last pixel, never written back)
      f2 v;
      v.x = *(const float*)(lds + bo + (i < 4 ? ((int)threadIdx.x + i * NT) * 4 : r5));
      v.y = *(const float*)(lds + bo + kBoxJB + (i < 4 ? ((int)threadIdx.x + i * NT) * 4 : r5));
      rss[i] = __builtin_elementwise_fma(v, v, rss[i]);
    }
#pragma unroll
    for (int s = 0; s < kBoxND; ++s) asm volatile("" : "+v"(dot[s]));  // the chunk's sums exist before its barrier
#pragma unroll
    for (int i = 0; i < 5; ++i) asm volatile("" : "+v"(rss[i]));
    asm volatile("" : "+v"(oss));
    if (ps + 1 < NP) {
      if (ps + 2 < NP) PEA_BWAIT1()
      else PEA_BWAIT(0);
      if (ps + 3 < NP) PEA_BDMA(ps % 3, (unsigned)(2 * ps + 6) * ecs)
    }
  }

  // ---- 1 / norm of every region pixel -> LDS; the lane's own one (signed) to the plane the backward stages
  const float inv_eps = 1.0f / P.eps;
#pragma unroll
  for (int i = 0; i < 5; ++i)
    if (i < 4 || own5) sN[(int)threadIdx.x + i * NT] = rnorm(rss[i].x + rss[i].y, inv_eps);
  const int py = y0 + ly, px = x0 + lx;
  const bool live = py < P.Y && px < P.X;
  const unsigned pe = live ? (unsigned)((z * P.Y + py) * P.X + px) * 4u : kOOB;
  const float osum = oss.x + oss.y;
  const float inv_own = rnorm(osum, inv_eps);
  if (inv_out) bs32(mkbuf(inv_out + (size_t)b * S), osum < P.eps * P.eps ? -inv_own : inv_own, pe, 0u);
  lds_barrier();  // the ring is dead, the norms are there
  const float* const nb = (const float*)((const char*)sN + own - kBoxBias);
#pragma unroll
  for (int s = 0; s < kBoxND; ++s) {
    const float iq = *(const float*)((const char*)nb + kBoxBias + box_lds(s));
    sA[s * TP + (int)threadIdx.x] = (dot[s].x + dot[s].y) * inv_own * iq;  // a neighbour outside a CROP volume: dot = 0
  }
  lds_barrier();

  // ---- epilogue: item = (offset, quad of 4 x-adjacent tile pixels), dwordx4 everywhere; a wave's items share the offset
  const unsigned af = P.flags & kActMask;
  const bool has_a = affs != nullptr, has_g = gout != nullptr, has_m = mask != nullptr;
  const int nitems = P.K * QP;
  for (int it0 = (int)threadIdx.x; it0 < nitems; it0 += NT) {
    const int k = __builtin_amdgcn_readfirstlane(it0 / QP);
    const int qd = it0 - k * QP;
    const int l4 = qd * 4;
    const int gy = y0 + l4 / TW, gx = x0 + l4 % TW;
    const bool lv = gy < P.Y && gx < P.X;  // X % 4 == 0: a quad is inside or outside as a whole
    const unsigned vo = lv ? (unsigned)((z * P.Y + gy) * P.X + gx) * 4u : kOOB;
    f4 t4 = {0.f, 0.f, 0.f, 0.f}, w4 = t4;
    unsigned m4 = 0x01010101u;
    if (TRAIN) {
      t4 = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(mkbuf(target + (size_t)b * P.tbs + (size_t)k * S), vo, 0u, kAuxNT));
      w4 = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(mkbuf(weight + (size_t)b * P.wbs + (size_t)k * S), vo, 0u, kAuxNT));
      if (has_m) m4 = __builtin_amdgcn_raw_buffer_load_b32(mkbuf(mask + (size_t)b * P.mbs + (size_t)k * S), vo == kOOB ? kOOB : vo >> 2, 0u, kAuxNT);
    }
    const f4 a4 = *(const f4*)(sA + C.slot[k] * TP + l4);
    if (has_a) {
      f4 o = a4;
      if (af) { o.x = act_affs(o.x, af); o.y = act_affs(o.y, af); o.z = act_affs(o.z, af); o.w = act_affs(o.w, af); }
      bs128<true>(mkbuf(affs + ((size_t)b * P.K + k) * S), o, vo, 0u);
    }
    if (TRAIN) {
      float acc = 0.f;
      f4 g4;
      const float gs = C.gs[k];
      const int oz = P.off[k][0], oy = P.off[k][1], ox = P.off[k][2];
      const bool rowok = !CROP || ((unsigned)(z + oz) < (unsigned)P.Z && (unsigned)(gy + oy) < (unsigned)P.Y);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float m = (float)((m4 >> (8 * j)) & 0xffu);
        const float r = a4[j] * m - t4[j] * m;
        float wr = w4[j] * r;
        if (CROP) wr = rowok && (unsigned)(gx + j + ox) < (unsigned)P.X ? wr : 0.f;  // a cropped-away neighbour carries no loss term
        g4[j] = gs * wr * m;
        acc = fmaf(wr, r, acc);
      }
      if (has_g) bs128<false>(mkbuf(gout + ((size_t)b * P.K + k) * S), g4, vo, 0u);
      const float red = wave_sum63(acc);
      if (lane == 63) s_part[k * NSL + (qd >> 6)] = red;
    }
  }
  if (TRAIN) {
    lds_barrier();
    if (wave == 0 && (int)threadIdx.x < P.K) {
      float v = 0.f;
#pragma unroll
      for (int s = 0; s < NSL; ++s) v += s_part[threadIdx.x * NSL + s];
      loss_accumulate(st, tile, threadIdx.x, v);
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// backward, self loss (both roles)
// ------------------------------------------------------------------------------------------------------------------
// LDS: the ring only (52 KB: THREE workgroups per CU at 72 VGPRs) -- the 1 / norm region starts out in the third chunk buffer, which
// is first filled after the coefficients are done (as in k_bwd_xdma).
constexpr int kBoxLdsBwd = 3 * kBoxCB;
template <int D_T, bool CROP>
__global__ __launch_bounds__(kBoxTH* kBoxTW, 6) void k_bwd_box(const KParams P, const BParams C, const float* __restrict__ xt,
                                                               const float* __restrict__ invp, const float* __restrict__ gin,
                                                               const float* __restrict__ dloss, float* __restrict__ dx) {
  constexpr int TH = kBoxTH, TW = kBoxTW, NP = D_T / 2;
  static_assert(D_T == 16, "the lane keeps its pixel and G in registers");
  extern __shared__ f4 lds4[];
  char* lds = (char*)lds4;
  int tile, b, z, y0, x0;
  if (!march_tile<TH, TW, BParams>(C, P, tile, b, z, y0, x0)) return;
  const size_t S = (size_t)P.S;
  const rsrc_t xB = mkbuf(xt + (size_t)b * D_T * S), dB = mkbuf(dx + (size_t)b * D_T * S);
  const rsrc_t iB = mkbuf(invp + (size_t)b * S);
  const unsigned ecs = (unsigned)P.S * 4u;
  const float dl = dloss ? dloss[0] : 1.f;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const unsigned vo0 = box_item<CROP>(P, wave * 64 + lane, kBoxCQ, z, y0, x0, true);
  const unsigned vo1 = box_item<CROP>(P, (8 + wave) * 64 + lane, kBoxCQ, z, y0, x0, true);
  const unsigned vo2 = box_item<CROP>(P, 16 * 64 + lane, kBoxCQ, z, y0, x0, true);
  // the 1 / norm region (one "channel": 540 quads, 9 blocks -- wave 0 moves two of them)
  {
    const unsigned vn0 = box_item<CROP>(P, wave * 64 + lane, kBoxNQ, z, y0, x0, false);
    const unsigned vn1 = box_item<CROP>(P, 8 * 64 + lane, kBoxNQ, z, y0, x0, false);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(iB, (lds_ptr_t)(lds + 2 * kBoxCB + wave * 1024), 16, vn0, 0u, 0, 0);
    if (wave == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(iB, (lds_ptr_t)(lds + 2 * kBoxCB + 8 * 1024), 16, vn1, 0u, 0, 0);
  }
  PEA_BDMA(0, 0u)

  const int ly = threadIdx.x >> 5, lx = threadIdx.x & 31;
  const int py = y0 + ly, px = x0 + lx;
  const bool live = py < P.Y && px < P.X;
  const unsigned pe = live ? (unsigned)((z * P.Y + py) * P.X + px) * 4u : kOOB;
  // ---- coefficients: role A of the offset o == d (g at p) + role B of the offset o == -d (g at p + d)
  float c[kBoxND];
#pragma unroll
  for (int s = 0; s < kBoxND; ++s) {
    const int ka = C.kA[s], kb = C.kB[s];  // uniform
    float v = 0.f;
    if (ka >= 0) v = bl32(mkbuf(gin + ((size_t)b * P.K + ka) * S), pe, 0u);
    if (kb >= 0) {
      bool okz, oky, okx;
      const int qz = wrap1<CROP>(z + box_dz(s), P.Z, okz), qy = wrap1<CROP>(py + box_dy(s), P.Y, oky),
                qx = wrap1<CROP>(px + box_dx(s), P.X, okx);
      const unsigned qo = (live && okz && oky && okx) ? (unsigned)((qz * P.Y + qy) * P.X + qx) * 4u : kOOB;
      v += bl32(mkbuf(gin + ((size_t)b * P.K + kb) * S), qo, 0u);
    }
    c[s] = v;
  }
  PEA_BDMA(1, 2u * ecs)
  PEA_BWAIT1()  // the 1 / norm region, chunk 0 and every g have landed
  const int own = kBoxRP * 4 + ((ly + 1) * kBoxRW + lx + 4) * 4;
  const char* const rb = lds + own - kBoxBias;
  const float invo = *(const float*)(lds + 2 * kBoxCB + own);
  const float inv_own = fabsf(invo);
  f2 c2[kBoxND / 2];
#pragma unroll
  for (int s = 0; s < kBoxND; ++s) {
    const float iq = fabsf(*(const float*)(rb + 2 * kBoxCB + kBoxBias + box_lds(s)));
    const float v = c[s] * iq;
    if (s & 1) c2[s / 2].y = v;
    else c2[s / 2].x = v;
  }
#pragma unroll
  for (int s = 0; s < kBoxND / 2; ++s) asm volatile("" : "+v"(c2[s]));
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // the 1 / norm region is dead: the third buffer may be filled
  PEA_BDMA(2, 4u * ecs)

  f2 G[NP], eh[NP];
#pragma unroll
  for (int ps = 0; ps < NP; ++ps) {
    const int bo = (ps % 3) * kBoxCB;
    f2 o;
    o.x = *(const float*)(lds + bo + own);
    o.y = *(const float*)(lds + bo + kBoxJB + own);
    eh[ps] = o * inv_own;
    asm volatile("" : "+v"(eh[ps]));
    f2 acc = {0.f, 0.f};
#pragma unroll
    for (int s = 0; s < kBoxND; ++s) {
      f2 v;
      v.x = *(const float*)(rb + bo + kBoxBias + box_lds(s));
      v.y = *(const float*)(rb + bo + kBoxJB + kBoxBias + box_lds(s));
      acc = (s & 1) ? pk_fma_c<true>(c2[s / 2], v, acc) : pk_fma_c<false>(c2[s / 2], v, acc);
      if (s % 6 == 5) asm volatile("" ::: "memory");
    }
    asm volatile("" : "+v"(acc));
    G[ps] = acc;
    if (ps + 1 < NP) {
      if (ps + 2 < NP) PEA_BWAIT1()
      else PEA_BWAIT(0);
      if (ps + 3 < NP) PEA_BDMA(ps % 3, (unsigned)(2 * ps + 6) * ecs)
    }
  }
  float proj = 0.f;
#pragma unroll
  for (int ps = 0; ps < NP; ++ps) proj = fmaf(eh[ps].x, G[ps].x, fmaf(eh[ps].y, G[ps].y, proj));
  if (invo < 0.f) proj = 0.f;  // clamp branch of F.normalize
  const float sc = dl * inv_own;
#pragma unroll
  for (int ps = 0; ps < NP; ++ps) {
    const float vx = (G[ps].x - eh[ps].x * proj) * sc, vy = (G[ps].y - eh[ps].y * proj) * sc;
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, vx), dB, pe, (unsigned)(2 * ps) * ecs, kAuxNT);
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, vy), dB, pe, (unsigned)(2 * ps + 1) * ecs, kAuxNT);
  }
}

#undef PEA_BDMA
#undef PEA_BWAIT1
#undef PEA_BWAIT

// host: the plan.  false = an offset leaves the unit box, or the volume does not meet the kernels' addressing limits
inline bool plan_box(const KParams& P, BParams* out) {
  if (P.border == PEA_BORDER_REPLICATE) return false;
  if (P.X % 4 || P.S % 4 || P.K > kBoxND) return false;
  if ((long long)P.D * P.S * 4 >= (1LL << 31)) return false;  // one buffer resource per batch item of e / de
  if (P.Y < kBoxTH + 1 || P.X < kBoxTW + 4) return false;     // the kernels wrap with one conditional add
  BParams C = {};
  for (int s = 0; s < kBoxND; ++s) C.kA[s] = C.kB[s] = -1;
  for (int i = 0; i < P.K; ++i) {
    const int oz = P.off[i][0], oy = P.off[i][1], ox = P.off[i][2];
    if (oz < -1 || oz > 1 || oy < -1 || oy > 1 || ox < -1 || ox > 1 || (oz == 0 && oy == 0 && ox == 0)) return false;
    if (oz != 0 && P.Z < 2) return false;
    const int di = (oz + 1) * 9 + (oy + 1) * 3 + (ox + 1), dn = (-oz + 1) * 9 + (-oy + 1) * 3 + (-ox + 1);
    const int s = di < 13 ? di : di - 1, sn = dn < 13 ? dn : dn - 1;
    if (C.kA[s] >= 0) return false;  // the same offset twice
    C.kA[s] = i;
    C.kB[sn] = i;
    C.slot[i] = s;
    C.gs[i] = P.gscale[i];
  }
  C.tiles_y = (P.Y + kBoxTH - 1) / kBoxTH;
  C.tiles_x = (P.X + kBoxTW - 1) / kBoxTW;
  C.tiles_per_plane = C.tiles_y * C.tiles_x;
  const long long nt = (long long)C.tiles_per_plane * P.Z * P.B;
  if (nt > 0x7fffff00LL) return false;
  C.ntiles = (int)nt;
  C.tiles_per_xcd = (C.ntiles + kXcd - 1) / kXcd;
  C.zrun = P.Z > 1 ? P.Z : 0;
  C.zgy = 4; C.zgx = 2;
  *out = C;
  return true;
}

}  // namespace pea
