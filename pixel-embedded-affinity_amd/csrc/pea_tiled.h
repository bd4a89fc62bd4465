// pea_tiled.h -- LDS-tiled kernels: the fast path.  Included by pea_hip.hip only.
//
// One workgroup owns a TH x TW tile of one (b, z) plane, one lane per pixel.  It stages the tile plus
// the halo that the "near" offsets reach into LDS ONCE -- already L2-normalised -- and every lane then
// takes its K (forward) or 2K (backward) neighbour vectors from LDS instead of re-reading L2/HBM through
// the 64 B/clk vector-memory path.  Offsets whose reach would blow the LDS budget ("far", e.g. +-27) are
// read straight from global memory (coalesced row reads that hit the XCD's L2).
//
// What bounds these kernels, and what the code does about it (rocprofv3 counters under profiles/):
//   * HBM traffic is at the compulsory minimum (FETCH_SIZE == the tensors' bytes): what is left is
//     instruction issue and latency, so the instruction stream is kept lean.  All global accesses are BUFFER
//     ops: one 128-bit resource per tensor (uniform), ONE 32-bit per-lane byte offset shared by all D
//     channels / K offsets, the channel / offset plane selected by the scalar soffset -- zero VALU per access.
//     Lanes outside the image get an out-of-range offset: loads return 0 and stores are dropped by the
//     hardware bounds check (no exec-mask branches).  Each offset's constants come from one 16-byte table
//     entry (a single s_load_dwordx4; a byte table would be fetched with vector loads, look divergent and
//     turn every buffer access into a readfirstlane waterfall loop -- as would resources captured by a lambda).
//     The staging loop advances (row, col) incrementally; norms use v_rsq_f32; wave reductions use DPP adds;
//     the border mode is a template parameter.
//   * latency: the streaming operands of the first chunk (target / weight / mask, or g) and the own pixel are
//     requested BEFORE the staging loads, the first far vector right after them, and selects are applied
//     where a value is consumed, so nothing waits on those loads before the staging phase.
//   * registers: <= 128 VGPRs, so 16 waves (one 1024-lane or two 512-lane workgroups) stay resident per CU.
//
// LDS layout: the D channels of a pixel are cut into S = D/4 float4 slots; slot q of all region pixels forms
// one plane of float4 with a COMPILE-TIME plane stride.  Consecutive lanes read consecutive float4 of a plane
// (ds_read_b128, all 64 banks, conflict-free at 256 B/clk, for every stencil shift), and the S slot reads of
// one pixel differ only in the instruction's immediate offset: two VALU adds per neighbour vector.
#pragma once
#include "pea_direct.h"

namespace pea {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;

struct OffEnt {  // per-offset constants, one s_load_dwordx4
  int i;         // offset index (channel of target / weight / mask / affs / g)
  int d;         // LDS displacement in pixels (oy * RW + ox); far offsets: oz
  int oyx;       // (oy << 16) | (ox & 0xffff)
  float gscale;  // 2 * lambda_i / N_i
};
__device__ __forceinline__ int ent_oy(const OffEnt& e) { return e.oyx >> 16; }
__device__ __forceinline__ int ent_ox(const OffEnt& e) { return (int)(short)(e.oyx & 0xffff); }

struct TParams {
  int hy0, hy1, hx0, hx1;  // halo: rows above / below, columns left / right of the tile
  int RH, RW, R;           // staged region (rows, columns, pixels)
  int dr, dc;              // NT / RW, NT % RW: (row, col) advance of one staging step
  float inv_rw;            // 1 / RW
  float inv_eps;           // 1 / eps
  int tiles_y, tiles_x, tiles_per_plane;
  int ntiles, tiles_per_xcd;
  int n_near, n_far;
  OffEnt near[PEA_MAX_K];  // served from LDS
  OffEnt far[PEA_MAX_K];   // served from global memory (d = oz: may also leave the z plane)
};

// ---- cheap primitives --------------------------------------------------------------------------
// 1 / max(sqrt(ss), eps) == min(rsqrt(ss), 1/eps); v_rsq_f32 is 1 ulp, ss == 0 gives +inf -> 1/eps
__device__ __forceinline__ float rnorm(float ss, float inv_eps) { return fminf(__builtin_amdgcn_rsqf(ss), inv_eps); }

// Buffer access to one tensor of one batch item.  vo = per-lane byte offset inside a plane (kOOB for lanes
// that must not touch memory: out of range => load returns 0 / store dropped), so = uniform byte offset of
// the plane (channel / offset index and z).  NTL: non-temporal (streamed-once operands).
constexpr int kAuxNT = 2;
constexpr unsigned kOOB = 0xFFFFFFFFu;
__device__ __forceinline__ rsrc_t mkbuf(const void* base) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, -1, 0x00020000);
}
template <bool NTL = false>
__device__ __forceinline__ float bl32(rsrc_t r, unsigned vo, unsigned so) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, vo, so, NTL ? kAuxNT : 0));
}
template <bool NTL = false>
__device__ __forceinline__ float bl8(rsrc_t r, unsigned vo, unsigned so) {
  return (float)__builtin_amdgcn_raw_buffer_load_b8(r, vo, so, NTL ? kAuxNT : 0);
}
template <bool NTL = false>
__device__ __forceinline__ void bs32(rsrc_t r, float v, unsigned vo, unsigned so) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, vo, so, NTL ? kAuxNT : 0);
}
// embedding element of storage type T (f32 or f16); vo in bytes of T
template <typename T>
__device__ __forceinline__ float bl_emb(rsrc_t r, unsigned vo, unsigned so) {
  if (sizeof(T) == 4) return bl32(r, vo, so);
  return __half2float(__builtin_bit_cast(__half, __builtin_amdgcn_raw_buffer_load_b16(r, vo, so, 0)));
}
template <typename T>
__device__ __forceinline__ void bs_emb(rsrc_t r, float v, unsigned vo, unsigned so) {
  if (sizeof(T) == 4) bs32(r, v, vo, so);
  else __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, __float2half(v)), r, vo, so, 0);
}

// sum over the 64 lanes, result valid in lane 63: 6 DPP adds (no LDS traffic)
__device__ __forceinline__ float wave_sum63(float v) {
#define PEA_DPP_ADD(ctrl, rmask) \
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, rmask, 0xf, false))
  PEA_DPP_ADD(0x111, 0xf);  // row_shr:1
  PEA_DPP_ADD(0x112, 0xf);  // row_shr:2
  PEA_DPP_ADD(0x114, 0xf);  // row_shr:4
  PEA_DPP_ADD(0x118, 0xf);  // row_shr:8   -> lane 15 of each row = row total
  PEA_DPP_ADD(0x142, 0xa);  // row_bcast:15 into rows 1,3
  PEA_DPP_ADD(0x143, 0xc);  // row_bcast:31 into rows 2,3 -> lane 63 = wave total
#undef PEA_DPP_ADD
  return v;
}

// ---- LDS geometry: [slot q][region pixel] of float4, PLQ = compile-time plane stride in pixels ----------
template <int D_T, int PLQ>
struct Lds {
  static constexpr int S = D_T / 4;
  static constexpr int kPlaneB = PLQ * 16;
  static constexpr int kBytes = S * kPlaneB;
  static constexpr int kHalf = (S + 1) / 2;  // slots addressed from one base register (immediates < 64 KB)
  static_assert(D_T % 4 == 0, "tiled kernels need D % 4 == 0");
  static_assert((kHalf - 1) * kPlaneB < 65536, "slot offsets must fit the ds immediate");
};

// the D channels of region pixel `pi`
template <int D_T, int PLQ>
__device__ __forceinline__ void lds_pixel(const char* __restrict__ lds, int pi, float* v) {
  typedef Lds<D_T, PLQ> L;
  const char* a = lds + pi * 16;
  const char* a2 = a + L::kHalf * L::kPlaneB;
#pragma unroll
  for (int q = 0; q < L::S; ++q) {
    const f4 t = q < L::kHalf ? *(const f4*)(a + q * L::kPlaneB) : *(const f4*)(a2 + (q - L::kHalf) * L::kPlaneB);
    v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
  }
}

// tile id -> (plane = b*Z + z, y0, x0); XCD-aware: XCD group g walks tiles [g*tpx, (g+1)*tpx) in row-major order
__device__ __forceinline__ int tile_id(const TParams& Q) {
  const int bid = blockIdx.x;
  return (bid % kXcd) * Q.tiles_per_xcd + bid / kXcd;
}

// wrap (CIRCULAR) or test (CROP) an index one step; host guarantees |o| and the halo are <= the extent
template <bool CROP>
__device__ __forceinline__ int wrap1(int v, int n, bool& ok) {
  if (!CROP) {
    v += v < 0 ? n : 0;
    v -= v >= n ? n : 0;
    ok = true;
  } else {
    ok = (unsigned)v < (unsigned)n;
  }
  return v;
}

// stage the normalised region of batch item `eb`, plane byte offset `zo`, into LDS
template <typename T, int D_T, int PLQ, int NT, bool CROP>
__device__ __forceinline__ void stage_region(const KParams& P, const TParams& Q, rsrc_t eb, unsigned zo, unsigned cs,
                                             int y0, int x0, char* __restrict__ lds) {
  typedef Lds<D_T, PLQ> L;
  int idx = threadIdx.x;
  int r = (int)(((float)idx + 0.5f) * Q.inv_rw);
  int c = idx - r * Q.RW;
#pragma unroll 2
  for (; idx < Q.R; idx += NT) {
    bool oky, okx;
    const int gy = wrap1<CROP>(y0 - Q.hy0 + r, P.Y, oky);
    const int gx = wrap1<CROP>(x0 - Q.hx0 + c, P.X, okx);
    const unsigned vo = (oky && okx) ? (unsigned)(gy * P.X + gx) * (unsigned)sizeof(T) : kOOB;  // outside => zeros
    float v[D_T];
    float ss = 0.f;
#pragma unroll
    for (int ch = 0; ch < D_T; ++ch) {
      v[ch] = bl_emb<T>(eb, vo, zo + ch * cs);
      ss = fmaf(v[ch], v[ch], ss);
    }
    const float inv = rnorm(ss, Q.inv_eps);
    char* dst = lds + idx * 16;
#pragma unroll
    for (int q = 0; q < L::S; ++q) {
      f4 t;
      t.x = v[4 * q] * inv; t.y = v[4 * q + 1] * inv; t.z = v[4 * q + 2] * inv; t.w = v[4 * q + 3] * inv;
      *(f4*)(dst + q * L::kPlaneB) = t;
    }
    r += Q.dr;
    c += Q.dc;
    if (c >= Q.RW) { c -= Q.RW; r += 1; }
  }
}

// lane -> its pixel of the tile (one pixel per lane)
template <int TW>
__device__ __forceinline__ void lane_pixel(int& ly, int& lx) {
  static_assert(TW == 64 || TW == 32, "tile width is one or half a wave");
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (TW == 64) { ly = wave; lx = lane; }
  else { ly = 2 * wave + (lane >> 5); lx = lane & 31; }
}

// ---- forward helpers ------------------------------------------------------------------------------
struct FwdU {  // uniform per-workgroup state
  rsrc_t aB, gB, tB, wB, mB;
  unsigned kzo, kcs, S32;  // byte offset of plane z / stride of one offset channel (f32); elements per channel
  bool has_a, has_g, has_m, relu;
};

template <int N>
struct Twm {
  float t[N], w[N], m[N];
};

template <int KN>
__device__ __forceinline__ void fwd_load_twm(Twm<KN>& r, const FwdU U, const OffEnt* __restrict__ ent, int k0, int n,
                                             unsigned pb, unsigned pm) {
#pragma unroll
  for (int u = 0; u < KN; ++u) {
    const unsigned i = ent[min(k0 + u, n - 1)].i;
    r.t[u] = bl32<true>(U.tB, pb, U.kzo + i * U.kcs);
    r.w[u] = bl32<true>(U.wB, pb, U.kzo + i * U.kcs);
    r.m[u] = U.has_m ? bl8<true>(U.mB, pm, (U.kzo >> 2) + i * U.S32) : 1.f;
  }
}

// one offset's epilogue: affs / g stores (dropped by the bounds check for lanes outside the image) and the
// loss partial (wave-reduced, lane 63 writes)
template <bool TRAIN>
__device__ __forceinline__ void fwd_finish(const FwdU U, int K, float* s_part, const OffEnt e, float a, bool valid, float t,
                                           float w, float m, unsigned pb) {
  const unsigned so = U.kzo + (unsigned)e.i * U.kcs;
  if (U.has_a) bs32<true>(U.aB, U.relu ? fmaxf(a, 0.f) : a, pb, so);
  if (TRAIN) {
    const float r = a * m - t * m;
    const float wr = valid ? w * r : 0.f;
    if (U.has_g) bs32(U.gB, e.gscale * wr * m, pb, so);
    const float red = wave_sum63(wr * r);
    if ((threadIdx.x & 63) == 63) s_part[(threadIdx.x >> 6) * K + e.i] = red;
  }
}

// ------------------------------------------------------------------------------------------------
// forward, tiled.  SELF: e_other == e (own pixel comes out of LDS too).
// ------------------------------------------------------------------------------------------------
template <typename T, int D_T, int TH, int TW, int PLQ, bool CROP, bool TRAIN, bool SELF>
__global__ __launch_bounds__(TH* TW, 4) void k_fwd_tiled(const KParams P, const TParams Q, const T* __restrict__ e,
                                                         const T* __restrict__ eo, const float* __restrict__ target,
                                                         const float* __restrict__ weight,
                                                         const uint8_t* __restrict__ mask, float* __restrict__ affs,
                                                         float* __restrict__ gout, float* __restrict__ partials) {
  typedef Lds<D_T, PLQ> L;
  constexpr int NT = TH * TW, NW = NT / 64;
  constexpr int KN = 4;  // near offsets per chunk
  extern __shared__ f4 lds4[];
  char* lds = (char*)lds4;
  float* s_part = (float*)(lds + L::kBytes);  // [NW][K]
  const int tile = tile_id(Q);
  if (tile >= Q.ntiles) return;
  const int plane = tile / Q.tiles_per_plane;
  const int rem = tile - plane * Q.tiles_per_plane;
  const int ty = rem / Q.tiles_x;
  const int y0 = ty * TH, x0 = (rem - ty * Q.tiles_x) * TW;
  const int b = plane / P.Z, z = plane - b * P.Z;
  const size_t S = (size_t)P.S;
  const unsigned YX = (unsigned)(P.Y * P.X);
  // one buffer resource per tensor of this batch item; planes are selected by scalar byte offsets
  const rsrc_t eB = mkbuf(e + (size_t)b * D_T * S), oB = mkbuf(eo + (size_t)b * D_T * S);
  FwdU U;
  U.aB = mkbuf(affs ? affs + (size_t)b * P.K * S : nullptr);
  U.gB = mkbuf(gout ? gout + (size_t)b * P.K * S : nullptr);
  U.tB = mkbuf(target + (size_t)b * P.tbs);
  U.wB = mkbuf(weight + (size_t)b * P.wbs);
  U.mB = mkbuf(mask ? mask + (size_t)b * P.mbs : nullptr);
  U.kcs = (unsigned)P.S * 4u;
  U.kzo = (unsigned)z * YX * 4u;
  U.S32 = (unsigned)P.S;
  U.has_a = affs != nullptr; U.has_g = gout != nullptr; U.has_m = mask != nullptr;
  U.relu = P.flags & PEA_FLAG_RELU_AFFS;
  const unsigned ecs = (unsigned)P.S * (unsigned)sizeof(T);  // embedding channel stride, bytes
  const unsigned ezo = (unsigned)z * YX * (unsigned)sizeof(T);

  int ly, lx;
  lane_pixel<TW>(ly, lx);
  const int py = y0 + ly, px = x0 + lx;
  const bool live = py < P.Y && px < P.X;
  const unsigned po = (unsigned)(py * P.X + px);  // element offset inside the plane
  const unsigned pb = live ? po * 4u : kOOB;      // f32 operands
  const unsigned pm = live ? po : kOOB;           // u8 mask
  const unsigned pe = live ? po * (unsigned)sizeof(T) : kOOB;
  const int pr = (ly + Q.hy0) * Q.RW + lx + Q.hx0;

  // (1) streaming operands of the first near chunk: in flight during the whole staging phase
  Twm<KN> sa;
  if (TRAIN && Q.n_near > 0) fwd_load_twm<KN>(sa, U, Q.near, 0, Q.n_near, pb, pm);
  float own[D_T];
  if (!SELF) {
#pragma unroll
    for (int c = 0; c < D_T; ++c) own[c] = bl_emb<T>(eB, pe, ezo + c * ecs);
  }

  // (2) stage the region
  stage_region<T, D_T, PLQ, NT, CROP>(P, Q, oB, ezo, ecs, y0, x0, lds);

  // (3) first far offset: neighbour vector + its streaming operands, in flight across the barrier
  float fv[D_T];
  Twm<1> sf;
  bool fok = false;
#define PEA_FWD_LOAD_FAR(k)                                                                        \
  {                                                                                                \
    const OffEnt fe_ = Q.far[k];                                                                   \
    bool okz_, oky_, okx_;                                                                         \
    const int zz_ = wrap1<CROP>(z + fe_.d, P.Z, okz_);                                             \
    const int yy_ = wrap1<CROP>(py + ent_oy(fe_), P.Y, oky_);                                      \
    const int xx_ = wrap1<CROP>(px + ent_ox(fe_), P.X, okx_);                                      \
    fok = live && okz_ && oky_ && okx_;                                                            \
    const unsigned zo_ = (unsigned)(CROP ? min(max(zz_, 0), P.Z - 1) : zz_) * YX * (unsigned)sizeof(T); \
    const unsigned vo_ = fok ? (unsigned)(yy_ * P.X + xx_) * (unsigned)sizeof(T) : kOOB;           \
    _Pragma("unroll") for (int c = 0; c < D_T; ++c) fv[c] = bl_emb<T>(oB, vo_, zo_ + c * ecs);     \
    if (TRAIN) fwd_load_twm<1>(sf, U, Q.far, k, Q.n_far, pb, pm);                                  \
  }
  if (Q.n_far > 0) PEA_FWD_LOAD_FAR(0)

  if (!SELF) {
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < D_T; ++c) ss = fmaf(own[c], own[c], ss);
    const float inv = rnorm(ss, Q.inv_eps);
#pragma unroll
    for (int c = 0; c < D_T; ++c) own[c] *= inv;
  }
  __syncthreads();
  if (SELF) lds_pixel<D_T, PLQ>(lds, pr, own);

  // ---- near offsets: neighbour vector from LDS -----------------------------------------------------
  for (int k0 = 0; k0 < Q.n_near; k0 += KN) {
    if (TRAIN && k0 > 0) fwd_load_twm<KN>(sa, U, Q.near, k0, Q.n_near, pb, pm);
#pragma unroll
    for (int u = 0; u < KN; ++u) {
      if (k0 + u < Q.n_near) {  // uniform
        const OffEnt en = Q.near[k0 + u];
        float v[D_T];
        lds_pixel<D_T, PLQ>(lds, pr + en.d, v);
        float a = 0.f;
#pragma unroll
        for (int c = 0; c < D_T; ++c) a = fmaf(own[c], v[c], a);
        bool valid = live;
        if (CROP) {
          const bool inside = (unsigned)(py + ent_oy(en)) < (unsigned)P.Y && (unsigned)(px + ent_ox(en)) < (unsigned)P.X;
          a = inside ? a : 0.f;
          valid = valid && inside;
        }
        fwd_finish<TRAIN>(U, P.K, s_part, en, a, valid, sa.t[u], sa.w[u], sa.m[u], pb);
      }
    }
  }

  // ---- far offsets: neighbour vector straight from global (L2) ---------------------------------------
  for (int k = 0; k < Q.n_far; ++k) {
    if (k > 0) PEA_FWD_LOAD_FAR(k)
    float dot = 0.f, sq = 0.f;
#pragma unroll
    for (int c = 0; c < D_T; ++c) {
      dot = fmaf(own[c], fv[c], dot);
      sq = fmaf(fv[c], fv[c], sq);
    }
    const float a = fok ? dot * rnorm(sq, Q.inv_eps) : 0.f;
    fwd_finish<TRAIN>(U, P.K, s_part, Q.far[k], a, fok, sf.t[0], sf.w[0], sf.m[0], pb);
  }
#undef PEA_FWD_LOAD_FAR

  if (TRAIN) {
    __syncthreads();
    if (threadIdx.x < P.K) {
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) v += s_part[w * P.K + threadIdx.x];
      partials[(size_t)threadIdx.x * Q.ntiles + tile] = v;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// backward, tiled (gather form): a weighted neighbour sum, G(p) = sum over (offset, role) of g * nbhat(q).
// `nb` is staged in LDS and supplies every near neighbour vector; the own pixel x(p) is read raw from
// global (its norm is needed for the projection through F.normalize); g = d loss / d affs [B,K,S].
//   ROLE_A: G(p) += g_i(p)       * nbhat(p + o_i)     (x is the first operand of <x(p), nb(p+o)>)
//   ROLE_B: G(p) += g_i(p - o_i) * nbhat(p - o_i)     (x is the second operand of <nb(p-o), x(p)>)
//   self loss: nb == x, both roles.
// ------------------------------------------------------------------------------------------------
template <typename T, int D_T, int TH, int TW, int PLQ, bool CROP, bool ROLE_A, bool ROLE_B>
__global__ __launch_bounds__(TH* TW, 4) void k_bwd_tiled(const KParams P, const TParams Q, const T* __restrict__ xt,
                                                         const T* __restrict__ nbt, const float* __restrict__ gin,
                                                         const float* __restrict__ dloss, T* __restrict__ dx) {
  constexpr int NT = TH * TW;
  constexpr int NR = (ROLE_A ? 1 : 0) + (ROLE_B ? 1 : 0);
  constexpr int KN = 4;  // near offsets per chunk (x NR roles of g values)
  extern __shared__ f4 lds4[];
  char* lds = (char*)lds4;
  const int tile = tile_id(Q);
  if (tile >= Q.ntiles) return;
  const int plane = tile / Q.tiles_per_plane;
  const int rem = tile - plane * Q.tiles_per_plane;
  const int ty = rem / Q.tiles_x;
  const int y0 = ty * TH, x0 = (rem - ty * Q.tiles_x) * TW;
  const int b = plane / P.Z, z = plane - b * P.Z;
  const size_t S = (size_t)P.S;
  const unsigned YX = (unsigned)(P.Y * P.X);
  const rsrc_t xB = mkbuf(xt + (size_t)b * D_T * S), nB = mkbuf(nbt + (size_t)b * D_T * S);
  const rsrc_t dB = mkbuf(dx + (size_t)b * D_T * S), gB = mkbuf(gin + (size_t)b * P.K * S);
  const unsigned ecs = (unsigned)P.S * (unsigned)sizeof(T);
  const unsigned ezo = (unsigned)z * YX * (unsigned)sizeof(T);
  const unsigned kcs = (unsigned)P.S * 4u;
  const unsigned kzo = (unsigned)z * YX * 4u;
  const float dl = dloss ? dloss[0] : 1.f;

  int ly, lx;
  lane_pixel<TW>(ly, lx);
  const int py = y0 + ly, px = x0 + lx;
  const bool live = py < P.Y && px < P.X;
  const unsigned po = (unsigned)(py * P.X + px);
  const unsigned pe = live ? po * (unsigned)sizeof(T) : kOOB;
  const int pr = (ly + Q.hy0) * Q.RW + lx + Q.hx0;

  // g of (near entry k, role) for this lane: role A reads g at p, role B at the neighbour p - o (wrapped);
  // pairs that do not exist (outside the image, cropped away, past the end of the table) read out of range = 0
#define PEA_BWD_LOAD_GN(k0)                                                                                    \
  {                                                                                                            \
    _Pragma("unroll") for (int u = 0; u < KN; ++u) _Pragma("unroll") for (int r = 0; r < NR; ++r) {            \
      const OffEnt en_ = Q.near[min((k0) + u, Q.n_near - 1)];                                                  \
      const int sg_ = (ROLE_A && r == 0) ? 1 : -1;                                                             \
      bool oky_, okx_;                                                                                         \
      const int yy_ = wrap1<CROP>(py + sg_ * ent_oy(en_), P.Y, oky_);                                          \
      const int xx_ = wrap1<CROP>(px + sg_ * ent_ox(en_), P.X, okx_);                                          \
      const bool ok_ = live && oky_ && okx_ && ((k0) + u < Q.n_near);                                          \
      gn[u][r] = bl32(gB, ok_ ? (sg_ > 0 ? po : (unsigned)(yy_ * P.X + xx_)) * 4u : kOOB, kzo + (unsigned)en_.i * kcs); \
    }                                                                                                          \
  }

  // (1) own raw pixel and the g values of the first near chunk: in flight during staging
  float xh[D_T];
#pragma unroll
  for (int c = 0; c < D_T; ++c) xh[c] = bl_emb<T>(xB, pe, ezo + c * ecs);
  float gn[KN][NR];
  if (Q.n_near > 0) PEA_BWD_LOAD_GN(0)

  // (2) stage
  stage_region<T, D_T, PLQ, NT, CROP>(P, Q, nB, ezo, ecs, y0, x0, lds);

  // (3) first far (offset, role) pair: in flight across the barrier.  Pair j = (far offset j / NR, role j % NR).
  const int n_farp = Q.n_far * NR;
  float fv[D_T], fg = 0.f;
#define PEA_BWD_LOAD_FAR(j)                                                                                   \
  {                                                                                                           \
    const OffEnt fe_ = Q.far[(j) / NR];                                                                       \
    const int sg_ = (ROLE_A && ((j) % NR) == 0) ? 1 : -1;                                                     \
    bool okz_, oky_, okx_;                                                                                    \
    const int zz_ = wrap1<CROP>(z + sg_ * fe_.d, P.Z, okz_);                                                  \
    const int yy_ = wrap1<CROP>(py + sg_ * ent_oy(fe_), P.Y, oky_);                                           \
    const int xx_ = wrap1<CROP>(px + sg_ * ent_ox(fe_), P.X, okx_);                                           \
    const bool ok_ = live && okz_ && oky_ && okx_;                                                            \
    const unsigned zc_ = (unsigned)(CROP ? min(max(zz_, 0), P.Z - 1) : zz_);                                  \
    const unsigned qo_ = (unsigned)(yy_ * P.X + xx_);                                                         \
    const unsigned vo_ = ok_ ? qo_ * (unsigned)sizeof(T) : kOOB;                                              \
    _Pragma("unroll") for (int c = 0; c < D_T; ++c) fv[c] = bl_emb<T>(nB, vo_, zc_ * YX * (unsigned)sizeof(T) + c * ecs); \
    fg = bl32(gB, ok_ ? (sg_ > 0 ? po : qo_) * 4u : kOOB, (sg_ > 0 ? kzo : zc_ * YX * 4u) + (unsigned)fe_.i * kcs); \
  }
  if (n_farp > 0) PEA_BWD_LOAD_FAR(0)

  float G[D_T];
  float ss = 0.f;
#pragma unroll
  for (int c = 0; c < D_T; ++c) {
    ss = fmaf(xh[c], xh[c], ss);
    G[c] = 0.f;
  }
  const bool tiny = ss < P.eps * P.eps;
  const float invp = rnorm(ss, Q.inv_eps);
#pragma unroll
  for (int c = 0; c < D_T; ++c) xh[c] *= invp;
  __syncthreads();

  // ---- near pairs ------------------------------------------------------------------------------------
  for (int k0 = 0; k0 < Q.n_near; k0 += KN) {
    if (k0 > 0) PEA_BWD_LOAD_GN(k0)
#pragma unroll
    for (int u = 0; u < KN; ++u) {
      if (k0 + u < Q.n_near) {  // uniform
        const int d = Q.near[k0 + u].d;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
          float v[D_T];
          lds_pixel<D_T, PLQ>(lds, pr + ((ROLE_A && r == 0) ? d : -d), v);
#pragma unroll
          for (int c = 0; c < D_T; ++c) G[c] = fmaf(gn[u][r], v[c], G[c]);
          // one neighbour vector live at a time: without this fence all ds_read_b128 groups of the chunk are
          // hoisted to its top and the kernel spills
          asm volatile("" ::: "memory");
        }
      }
    }
  }
#undef PEA_BWD_LOAD_GN

  // ---- far pairs (an out-of-range pair read zeros: g = 0, vector = 0) --------------------------------------
  for (int j = 0; j < n_farp; ++j) {
    if (j > 0) PEA_BWD_LOAD_FAR(j)
    float sq = 0.f;
#pragma unroll
    for (int c = 0; c < D_T; ++c) sq = fmaf(fv[c], fv[c], sq);
    const float g = fg * rnorm(sq, Q.inv_eps);
#pragma unroll
    for (int c = 0; c < D_T; ++c) G[c] = fmaf(g, fv[c], G[c]);
  }
#undef PEA_BWD_LOAD_FAR

  float proj = 0.f;
#pragma unroll
  for (int c = 0; c < D_T; ++c) proj = fmaf(xh[c], G[c], proj);
  if (tiny) proj = 0.f;  // clamp_min branch of F.normalize: d ehat / d e = I / eps
  const float sc = dl * invp;
#pragma unroll
  for (int c = 0; c < D_T; ++c) bs_emb<T>(dB, (G[c] - xh[c] * proj) * sc, pe, ezo + c * ecs);
}

}  // namespace pea
