// pea_boxm.h -- the unit-box backward (pea_box.h: the 26-neighbourhood of BASELINE.json configs[3] and its subsets) MARCHING along z
// with every channel of the three planes z - 1, z, z + 1 resident in LDS.
//
// Why: k_bwd_box works a (z, tile) per workgroup and stages, per channel pair, the planes z - 1, z, z + 1 of its 18 x 40 region:
// every plane of e is staged by THREE workgroups.  PMC on the 24 x 1024^2 sub-volume (profiles/traffic.json c4n26): 9.3 GB fetched
// beyond L2 for ~ 4.3 GB of compulsory reads, 10.9 GB moved in 2.32 - 2.41 ms -- the kernel is a bandwidth pipe for what it moves,
// and 3.2 GB of it is the same planes again.  Here a workgroup marches through a tile column and keeps ALL 16 channels of three
// planes in LDS: 8 channel pairs x 3 plane slots x (2 channels x 720 region pixels) = 138 KB + three 1 / norm planes (8.6 KB).
// Moving from z to z + 1 replaces one plane per pair -- 46 KB of new data per plane instead of 139 KB -- into the slot of the plane
// that has just died; the slot of plane z' is z' mod 3, a compile-time constant because the steps are unrolled three to a turn.
// One workgroup per CU (147 KB), so it has to hide its own latency:
//   * look-ahead of half a plane at least: after pairs 0 - 3 of plane z are done (barrier B1), their planes z + 2 and the 1 / norm
//     plane z + 2 are requested; after pairs 4 - 7 (barrier B2), theirs.  Nothing else synchronises: TWO barriers per plane.
//   * the 26 coefficient sums g_{o = d}(p) + g_{o = -d}(p + d) of the NEXT plane are requested into registers when a plane starts
//     (52 loads per lane; there is no window here, so the registers exist) and land while the eight pairs are gathered.
//   * hand-offs: before B1 a wave waits until only the 52 g loads may fly (vmcnt(52): the planes requested at the previous B2 are
//     older); before B2 for everything (vmcnt(0): the planes requested at B1 are the youngest loads -- the 16 gradient stores of
//     the plane are issued BEHIND B2, because a store may retire before an older load and must not be counted on).
// A unit (one pair's plane: 360 quads) is moved by one exec-masked dwordx4 LDS-DMA instruction per wave, 45 lanes each (every wave
// issues the same count); the 1 / norm plane (180 quads) by 23 lanes per wave.
// Gather form, no state across planes: a column may be cut into segments without warm-up or drain.  CROP_ZERO border only (a plane
// outside the volume arrives as zeros); every launch issues the full 52 coefficient loads per lane and plane (absent offsets at
// an out-of-range offset), so the counts above hold for subsets of the 26-neighbourhood too.
#pragma once
#include "pea_box.h"

namespace pea {

constexpr int kBmUnit = 2 * kBoxRP * 4;          // 5760 bytes: one pair's plane, [channel j][18 x 40]
constexpr int kBmPair = 3 * kBmUnit;             // 17280: the three plane slots of a pair
constexpr int kBmInv = 8 * kBmPair;              // 138240: the three 1 / norm planes behind them
constexpr int kBmLds = kBmInv + 3 * kBoxRP * 4;  // 146880
constexpr int kBmBias = (kBoxRW + 1) * 4;        // the read base is biased so that every immediate is >= 0

struct BMParams { int zseg, nseg; };

// byte offset (from the biased own-pixel base of a pair) of displacement slot s when the current plane sits in plane slot J
template <int J>
__host__ __device__ constexpr int boxm_lds(int s) {
  return ((J + box_dz(s) + 3) % 3) * kBmUnit + (box_dy(s) * kBoxRW + box_dx(s)) * 4 + kBmBias;
}

#define PEA_BM_WAIT(n) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(n) : "memory")

// plane zp of pair c into its slot (zp mod 3 == slot_); a plane outside the volume arrives as zeros
#define PEA_BM_UNIT(c_, zp_, slot_)                                                                                            \
  {                                                                                                                            \
    const int zq_ = (zp_);                                                                                                     \
    const bool in_ = zq_ >= 0 && zq_ < P.Z;                                                                                    \
    const unsigned so_ = (unsigned)((2 * (c_)) * P.Z + (in_ ? zq_ : 0)) * YX4;                                                 \
    if (lane < 45)                                                                                                             \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xB, (lds_ptr_t)(lds + (c_) * kBmPair + (slot_) * kBmUnit + wave * 720), 16,     \
                                               in_ ? vq : kOOB, so_, 0, 0);                                                    \
  }
#define PEA_BM_INV(zp_, slot_)                                                                                                 \
  {                                                                                                                            \
    const int zq_ = (zp_);                                                                                                     \
    const bool in_ = zq_ >= 0 && zq_ < P.Z;                                                                                    \
    if (lane < 23 && 23 * wave + lane < 180) /* (every wave has such lanes: the same instruction count everywhere) */          \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(iB, (lds_ptr_t)(lds + kBmInv + (slot_) * (kBoxRP * 4) + wave * 368), 16,        \
                                               in_ ? vi : kOOB, (unsigned)(in_ ? zq_ : 0) * YX4, 0, 0);                        \
  }
// the raw coefficient sums of plane zp_ into cr[26]: role A of the offset o == d (g at p) + role B of the offset o == -d (g at p + d)
#define PEA_BM_GLOAD(zp_)                                                                                                      \
  {                                                                                                                            \
    const int zq_ = (zp_);                                                                                                     \
    const bool in_ = zq_ < ze; /* the plane after the segment's last gets no coefficients */                                   \
    int py_o = py, px_o = px;                                                                                                  \
    _Pragma("unroll") for (int s = 0; s < kBoxND; ++s) {                                                                       \
      if (s % 4 == 0) asm volatile("" : "+v"(py_o), "+v"(px_o)); /* offsets formed where they are used, not kept */           \
      const int ka = C.kA[s], kb = C.kB[s];                                                                                    \
      const bool lv = in_ && py_o < P.Y && px_o < P.X;                                                                         \
      const unsigned va = (lv && ka >= 0) ? (unsigned)((zq_ * P.Y + py_o) * P.X + px_o) * 4u : kOOB;                          \
      bool okz, oky, okx;                                                                                                      \
      const int qz = wrap1<true>(zq_ + box_dz(s), P.Z, okz), qy = wrap1<true>(py_o + box_dy(s), P.Y, oky),                     \
                qx = wrap1<true>(px_o + box_dx(s), P.X, okx);                                                                  \
      const unsigned vb = (lv && kb >= 0 && okz && oky && okx) ? (unsigned)((qz * P.Y + qy) * P.X + qx) * 4u : kOOB;           \
      const float v = bl32(mkbuf(gin + ((size_t)b * P.K + (ka >= 0 ? ka : 0)) * S), va, 0u) +                                  \
                      bl32(mkbuf(gin + ((size_t)b * P.K + (kb >= 0 ? kb : 0)) * S), vb, 0u);                                   \
      cr[s] = v;                                                                                                               \
    }                                                                                                                          \
  }

// one plane z whose plane slot is J (= (z - zb) mod 3)
#define PEA_BM_STEP(J, zz)                                                                                                     \
  {                                                                                                                            \
    const int z = (zz);                                                                                                        \
    if (z < ze) {                                                                                                              \
      /* ---- coefficients: raw sums (registers, requested a plane ago) x 1 / |e(p + d)| from the three 1 / norm planes */    \
      const float invo = *(const float*)(lds + kBmInv + J * (kBoxRP * 4) + own);                                               \
      const float inv_own = fabsf(invo);                                                                                       \
      f2 c2[kBoxND / 2];                                                                                                       \
      _Pragma("unroll") for (int s = 0; s < kBoxND; ++s) {                                                                     \
        const float iq = fabsf(*(const float*)(lds + kBmInv + ((J + box_dz(s) + 3) % 3) * (kBoxRP * 4) + own +                 \
                                                (box_dy(s) * kBoxRW + box_dx(s)) * 4));                                        \
        const float v = cr[s] * iq;                                                                                            \
        if (s & 1) c2[s / 2].y = v;                                                                                            \
        else c2[s / 2].x = v;                                                                                                  \
      }                                                                                                                        \
      _Pragma("unroll") for (int s = 0; s < kBoxND / 2; ++s) asm volatile("" : "+v"(c2[s]));                                   \
      PEA_BM_GLOAD(z + 1)                                                                                                      \
      f2 G[8], eh[8];                                                                                                          \
      _Pragma("unroll") for (int c = 0; c < 8; ++c) {                                                                          \
        int pb = c * kBmPair; /* the pair's base: formed here (beyond the first three pairs no immediate reaches) */          \
        asm volatile("" : "+v"(pb));                                                                                           \
        const char* const rb = lds + own - kBmBias + pb;                                                                       \
        f2 o;                                                                                                                  \
        o.x = *(const float*)(rb + J * kBmUnit + kBmBias);                                                                     \
        o.y = *(const float*)(rb + J * kBmUnit + kBoxRP * 4 + kBmBias);                                                        \
        eh[c] = o * inv_own;                                                                                                   \
        f2 acc = {0.f, 0.f};                                                                                                   \
        _Pragma("unroll") for (int s = 0; s < kBoxND; ++s) {                                                                   \
          f2 v;                                                                                                                \
          v.x = *(const float*)(rb + boxm_lds<J>(s));                                                                          \
          v.y = *(const float*)(rb + kBoxRP * 4 + boxm_lds<J>(s));                                                             \
          acc = (s & 1) ? pk_fma_c<true>(c2[s / 2], v, acc) : pk_fma_c<false>(c2[s / 2], v, acc);                              \
          if (s % 13 == 12) asm volatile("" ::: "memory");                                                                     \
        }                                                                                                                      \
        asm volatile("" : "+v"(acc), "+v"(eh[c]));                                                                             \
        G[c] = acc;                                                                                                            \
        if (c == 3) { /* B1: pairs 0 - 3 are done with plane z - 1, the coefficients with 1 / norm plane z - 1 */             \
          PEA_BM_WAIT(2 * kBoxND); /* what was requested at the previous B2 has landed (the g loads behind it may fly) */     \
          PEA_BM_UNIT(0, z + 2, (J + 2) % 3)                                                                                   \
          PEA_BM_UNIT(1, z + 2, (J + 2) % 3)                                                                                   \
          PEA_BM_UNIT(2, z + 2, (J + 2) % 3)                                                                                   \
          PEA_BM_UNIT(3, z + 2, (J + 2) % 3)                                                                                   \
          PEA_BM_INV(z + 2, (J + 2) % 3)                                                                                       \
        }                                                                                                                      \
      }                                                                                                                        \
      float proj = 0.f;                                                                                                        \
      _Pragma("unroll") for (int c = 0; c < 8; ++c) proj = fmaf(eh[c].x, G[c].x, fmaf(eh[c].y, G[c].y, proj));                 \
      if (invo < 0.f) proj = 0.f; /* clamp branch of F.normalize */                                                           \
      const float sc = dl * inv_own;                                                                                           \
      const unsigned pe = live ? (unsigned)((z * P.Y + py) * P.X + px) * 4u : kOOB;                                            \
      /* B2: pairs 4 - 7 are done with plane z - 1; what was requested at B1 has landed: no load is younger than it (the      */ \
      /* gradient stores follow the barrier: a store may retire before an older load, so none may sit between a request and the */ \
      /* wait that counts on it -- pea_zmarch.h zm_bwd_wait)                                                                    */ \
      PEA_BM_WAIT(0);                                                                                                          \
      PEA_BM_UNIT(4, z + 2, (J + 2) % 3)                                                                                       \
      PEA_BM_UNIT(5, z + 2, (J + 2) % 3)                                                                                       \
      PEA_BM_UNIT(6, z + 2, (J + 2) % 3)                                                                                       \
      PEA_BM_UNIT(7, z + 2, (J + 2) % 3)                                                                                       \
      _Pragma("unroll") for (int c = 0; c < 8; ++c) {                                                                          \
        const float vx = (G[c].x - eh[c].x * proj) * sc, vy = (G[c].y - eh[c].y * proj) * sc;                                  \
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, vx), dB, pe, (unsigned)(2 * c) * ecs, kAuxNT);      \
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, vy), dB, pe, (unsigned)(2 * c + 1) * ecs, kAuxNT);  \
      }                                                                                                                        \
    }                                                                                                                          \
  }

// xt: e [B, 16, S]; invp: its signed 1 / norm plane; gin: d loss / d affs [B, K, S].  C: plan_box with the tile walk set up for
// columns (zrun = nseg: xdma_tile's "z" is the segment).  ONE workgroup of 512 lanes per CU (launch bounds: 2 waves per SIMD).
__global__ __launch_bounds__(kBoxTH* kBoxTW, 2) void k_bwd_boxm(const KParams P, const BParams C, const BMParams M,
                                                                const float* __restrict__ xt, const float* __restrict__ invp,
                                                                const float* __restrict__ gin, const float* __restrict__ dloss,
                                                                float* __restrict__ dx) {
  constexpr int TH = kBoxTH, TW = kBoxTW, D_T = 16;
  extern __shared__ f4 lds4[];
  char* lds = (char*)lds4;
  int tile, b, seg, y0, x0;
  if (!march_tile<TH, TW, BParams>(C, P, tile, b, seg, y0, x0)) return;
  const int zb = seg * M.zseg, ze = min(zb + M.zseg, P.Z);
  const size_t S = (size_t)P.S;
  const rsrc_t xB = mkbuf(xt + (size_t)b * D_T * S), dB = mkbuf(dx + (size_t)b * D_T * S);
  const rsrc_t iB = mkbuf(invp + (size_t)b * S);
  const unsigned ecs = (unsigned)P.S * 4u, YX4 = (unsigned)(P.Y * P.X) * 4u;
  const float dl = dloss ? dloss[0] : 1.f;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int ly = threadIdx.x >> 5, lx = threadIdx.x & 31;
  const int py = y0 + ly, px = x0 + lx;
  const bool live = py < P.Y && px < P.X;
  const int own = ((ly + 1) * kBoxRW + lx + 4) * 4;
  // this lane's quad of a unit (45 lanes per wave: quad q = 45 wave + lane of the 360 = [channel j][row][10 quads]) and of a
  // 1 / norm plane (23 lanes per wave, 180 quads); the channel's distance (j * S) rides in the per-lane offset
  unsigned vq = kOOB, vi = kOOB;
  {
    const int q = 45 * wave + lane;
    if (lane < 45) {
      const int j = q / 180, r = q - j * 180, row = r / 10, qd = r - row * 10;
      bool oky, okx;
      const int gy = wrap1<true>(y0 - 1 + row, P.Y, oky), gx = wrap1<true>(x0 - 4 + 4 * qd, P.X, okx);
      if (oky && okx) vq = (unsigned)j * ecs + (unsigned)(gy * P.X + gx) * 4u;  // j * S * 4 + plane offset < 2^31 (host-checked)
    }
    const int qi = 23 * wave + lane;
    if (lane < 23 && qi < 180) {
      const int row = qi / 10, qd = qi - row * 10;
      bool oky, okx;
      const int gy = wrap1<true>(y0 - 1 + row, P.Y, oky), gx = wrap1<true>(x0 - 4 + 4 * qd, P.X, okx);
      if (oky && okx) vi = (unsigned)(gy * P.X + gx) * 4u;
    }
  }
  float cr[kBoxND];
  // ---- head: planes zb - 1, zb, zb + 1 of every pair, their 1 / norm planes, the first plane's coefficient sums
  // (plane zb sits in slot 0, zb + 1 in slot 1, zb - 1 in slot 2)
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    PEA_BM_UNIT(c, zb - 1, 2)
    PEA_BM_UNIT(c, zb, 0)
    PEA_BM_UNIT(c, zb + 1, 1)
  }
  PEA_BM_INV(zb - 1, 2)
  PEA_BM_INV(zb, 0)
  PEA_BM_INV(zb + 1, 1)
  PEA_BM_GLOAD(zb)
  PEA_BM_WAIT(0);
  for (int zq = zb; zq < ze; zq += 3) {
    PEA_BM_STEP(0, zq)
    PEA_BM_STEP(1, zq + 1)
    PEA_BM_STEP(2, zq + 2)
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the look-ahead beyond the segment has drained
}

#undef PEA_BM_STEP
#undef PEA_BM_GLOAD
#undef PEA_BM_INV
#undef PEA_BM_UNIT
#undef PEA_BM_WAIT

}  // namespace pea
