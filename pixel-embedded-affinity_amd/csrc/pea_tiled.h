// pea_tiled.h -- LDS-tiled box kernels (round 1's fast path; since round 2 the fallback of the cross kernels, pea_xdma.h).
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
  float inv_sw;            // 1 / (hx0 + hx1): halo strip width of the tile's rows (stage_region_own)
  float inv_eps;           // 1 / eps
  int tiles_y, tiles_x, tiles_per_plane;
  int ntiles, tiles_per_xcd;
  int zrun;                // > 1: walk z fastest (zrun = Z) so that the planes an oz != 0 offset reaches are in L2
  int n_near, n_far;
  OffEnt near[PEA_MAX_K];  // served from LDS
  OffEnt far[PEA_MAX_K];   // served from global memory (d = oz: may also leave the z plane)
};

// ---- cheap primitives --------------------------------------------------------------------------
// Workgroup barrier that orders LDS traffic only.  __syncthreads() also emits s_waitcnt vmcnt(0): every barrier would
// wait for the global loads that were requested early precisely so that they stay in flight across it (measured with
// s_memtime stamps: 5-6k cycles per barrier).  The hardware barrier itself does not drain VMEM.  No kernel here
// passes data between lanes through global memory, so LDS ordering is all a barrier has to provide.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// 1 / max(sqrt(ss), eps) == min(rsqrt(ss), 1/eps); v_rsq_f32 is 1 ulp, ss == 0 gives +inf -> 1/eps
__device__ __forceinline__ float rnorm(float ss, float inv_eps) { return fminf(__builtin_amdgcn_rsqf(ss), inv_eps); }

// Buffer access to one tensor of one batch item.  vo = per-lane byte offset inside a plane (kOOB for lanes
// that must not touch memory: out of range => load returns 0 / store dropped; every legal offset is < 2^31
// because a plane is, see plan_tiles), so = uniform byte offset of
// the plane (channel / offset index and z).  NTL: non-temporal (streamed-once operands).
constexpr int kAuxNT = 2;
constexpr unsigned kOOB = 0x80000000u;  // == num_records: never near 2^32, so a 16-byte access cannot wrap past the check
__device__ __forceinline__ rsrc_t mkbuf(const void* base) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)kOOB, 0x00020000);
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
// NTL: non-temporal.  The gradient planes are written once and not read again by the kernel that writes them; as plain
// stores they push the embedding / g lines that the neighbouring tiles are about to re-read out of the XCD's 4 MB L2
// (measured on the cross backward: 140 us with plain stores, 97 us with nt stores, 136 us with sc1 alone).
template <typename T, bool NTL = false>
__device__ __forceinline__ void bs_emb(rsrc_t r, float v, unsigned vo, unsigned so) {
  if (sizeof(T) == 4) bs32<NTL>(r, v, vo, so);
  else __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, __float2half(v)), r, vo, so, NTL ? kAuxNT : 0);
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

// tile id -> (plane = b*Z + z, y0, x0); XCD-aware: XCD group g walks tiles [g*tpx, (g+1)*tpx) in row-major order.
// Volumes whose stencil leaves the plane (zrun = Z) walk z FASTEST instead: the tiles running together on an XCD are
// then a few (y, x) columns over all z, and the z-neighbour blocks a tile gathers from global memory were staged by the
// tiles just before it -- L2 hits instead of a second, third, ... HBM read of the volume.  The returned id stays
// plane-major (the partial-sum slot of a tile does not depend on the walk).
__device__ __forceinline__ int tile_id(const TParams& Q) {
  const int bid = blockIdx.x;
  const int lin = (bid % kXcd) * Q.tiles_per_xcd + bid / kXcd;
  if (Q.zrun <= 1 || lin >= Q.ntiles) return lin;
  const int per_b = Q.tiles_per_plane * Q.zrun;
  const int b = lin / per_b, r = lin - b * per_b;
  const int rem = r / Q.zrun, z = r - rem * Q.zrun;
  return (b * Q.zrun + z) * Q.tiles_per_plane + rem;
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

// PEA_BORDER_REPLICATE (embedding_loss_norm6, scripts_ac3ac4/loss/loss_embedding_mse.py:294-354) rides on the CROP instantiations:
// rep (wave-uniform: P.border == PEA_BORDER_REPLICATE) CLAMPS the index into the volume instead of testing it -- every neighbour
// exists.  f32, D = 16 only (pea_k_tiled.hip); the role-B side of the backward keeps the test and adds the border pixels' extra
// pre-images afterwards (k_bwd_tiled).
template <bool CROP>
__device__ __forceinline__ int wrap1r(int v, int n, bool& ok, bool rep) {
  if (CROP && rep) {
    ok = true;
    return min(max(v, 0), n - 1);
  }
  return wrap1<CROP>(v, n, ok);
}

// stage the normalised region of batch item `eb`, plane byte offset `zo`, into LDS.
// WINV (training forward, self loss): the lane that stages one of the tile's OWN pixels also writes its signed 1 / norm
// (negative where |e| < eps: the clamp branch of F.normalize) to the plane the cross backward (pea_xdma.h) reads;
// every other lane's store carries an out-of-range offset and is dropped by the bounds check.
template <typename T, int D_T, int PLQ, int NT, bool CROP, bool WINV>
__device__ __forceinline__ void stage_region_impl(const KParams& P, const TParams& Q, rsrc_t eb, unsigned zo, unsigned cs,
                                                  int y0, int x0, char* __restrict__ lds, rsrc_t ib, unsigned izo, int th, int tw) {
  typedef Lds<D_T, PLQ> L;
  int idx = threadIdx.x;
  int r = (int)(((float)idx + 0.5f) * Q.inv_rw);
  int c = idx - r * Q.RW;
#pragma unroll 2
  for (; idx < Q.R; idx += NT) {
    bool oky, okx;
    const int gy = wrap1r<CROP>(y0 - Q.hy0 + r, P.Y, oky, P.border == PEA_BORDER_REPLICATE);
    const int gx = wrap1r<CROP>(x0 - Q.hx0 + c, P.X, okx, P.border == PEA_BORDER_REPLICATE);
    const unsigned vo = (oky && okx) ? (unsigned)(gy * P.X + gx) * (unsigned)sizeof(T) : kOOB;  // outside => zeros
    float v[D_T];
    float ss = 0.f;
#pragma unroll
    for (int ch = 0; ch < D_T; ++ch) {
      v[ch] = bl_emb<T>(eb, vo, zo + ch * cs);
      ss = fmaf(v[ch], v[ch], ss);
    }
    const float inv = rnorm(ss, Q.inv_eps);
    if (WINV) {
      const int tr = r - Q.hy0, tc = c - Q.hx0;
      const bool mine = (unsigned)tr < (unsigned)th && (unsigned)tc < (unsigned)tw && y0 + tr < P.Y && x0 + tc < P.X;
      bs32(ib, ss < P.eps * P.eps ? -inv : inv, mine ? (unsigned)(gy * P.X + gx) * 4u : kOOB, izo);
    }
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
template <typename T, int D_T, int PLQ, int NT, bool CROP>
__device__ __forceinline__ void stage_region(const KParams& P, const TParams& Q, rsrc_t eb, unsigned zo, unsigned cs,
                                             int y0, int x0, char* __restrict__ lds) {
  stage_region_impl<T, D_T, PLQ, NT, CROP, false>(P, Q, eb, zo, cs, y0, x0, lds, eb, 0u, 0, 0);
}

// stage_region for the self-loss backward: every lane first stages ITS OWN tile pixel (keeping the raw channels and
// 1 / norm in registers: the 16 separate own-pixel loads disappear), then the lanes share the halo pixels.
// Halo enumeration: rows above the tile, rows below it, then the left / right strips of the tile's rows.
template <typename T, int D_T, int PLQ, int TH, int TW, bool CROP>
__device__ __forceinline__ void stage_region_own(const KParams& P, const TParams& Q, rsrc_t eb, unsigned zo, unsigned cs,
                                                 int y0, int x0, char* __restrict__ lds, int ly, int lx, float* own,
                                                 float& own_inv, float& own_ss) {
  typedef Lds<D_T, PLQ> L;
  constexpr int NT = TH * TW;
  const int top = Q.hy0 * Q.RW, bot = Q.hy1 * Q.RW, sw = Q.hx0 + Q.hx1;
  const int nhalo = top + bot + TH * sw;
  for (int j = -NT + (int)threadIdx.x; j < nhalo; j += NT) {  // first trip (j < 0): the own pixel
    int r, c;
    if (j < 0) { r = ly + Q.hy0; c = lx + Q.hx0; }
    else if (j < top) { r = (int)(((float)j + 0.5f) * Q.inv_rw); c = j - r * Q.RW; }
    else if (j < top + bot) { const int k = j - top; const int rr = (int)(((float)k + 0.5f) * Q.inv_rw); r = Q.hy0 + TH + rr; c = k - rr * Q.RW; }
    else { const int k = j - top - bot; const int rr = (int)(((float)k + 0.5f) * Q.inv_sw); const int cc = k - rr * sw; r = Q.hy0 + rr; c = cc < Q.hx0 ? cc : cc + TW; }
    bool oky, okx;
    const int gy = wrap1r<CROP>(y0 - Q.hy0 + r, P.Y, oky, P.border == PEA_BORDER_REPLICATE);
    const int gx = wrap1r<CROP>(x0 - Q.hx0 + c, P.X, okx, P.border == PEA_BORDER_REPLICATE);
    const unsigned vo = (oky && okx) ? (unsigned)(gy * P.X + gx) * (unsigned)sizeof(T) : kOOB;  // outside => zeros
    float v[D_T];
    float ss = 0.f;
#pragma unroll
    for (int ch = 0; ch < D_T; ++ch) {
      v[ch] = bl_emb<T>(eb, vo, zo + ch * cs);
      ss = fmaf(v[ch], v[ch], ss);
    }
    const float inv = rnorm(ss, Q.inv_eps);
    if (j < 0) {
      own_inv = inv;
      own_ss = ss;
#pragma unroll
      for (int ch = 0; ch < D_T; ++ch) own[ch] = v[ch] * inv;
    }
    char* dst = lds + (r * Q.RW + c) * 16;
#pragma unroll
    for (int q = 0; q < L::S; ++q) {
      f4 t;
      t.x = v[4 * q] * inv; t.y = v[4 * q + 1] * inv; t.z = v[4 * q + 2] * inv; t.w = v[4 * q + 3] * inv;
      *(f4*)(dst + q * L::kPlaneB) = t;
    }
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
  rsrc_t aB1, gB1, tB1, wB1;  // the same tensors based `ks` offset planes further (KParams::ksplit): channels >= ks go through these
  unsigned ks;
  unsigned kzo, kcs, S32;  // byte offset of plane z / stride of one offset channel (f32); elements per channel
  bool has_a, has_g, has_m;
  unsigned af;  // activation flags of the affs output (act_affs)
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
    const bool hi = i >= U.ks;  // uniform
    const unsigned so = U.kzo + (hi ? i - U.ks : i) * U.kcs;
    r.t[u] = bl32<true>(hi ? U.tB1 : U.tB, pb, so);
    r.w[u] = bl32<true>(hi ? U.wB1 : U.wB, pb, so);
    r.m[u] = U.has_m ? bl8<true>(U.mB, pm, (U.kzo >> 2) + i * U.S32) : 1.f;
  }
}

// one offset's epilogue: affs / g stores (dropped by the bounds check for lanes outside the image) and the
// loss partial (wave-reduced, lane 63 writes)
template <bool TRAIN>
__device__ __forceinline__ void fwd_finish(const FwdU U, int K, float* s_part, const OffEnt e, float a, bool valid, float t,
                                           float w, float m, unsigned pb) {
  const bool hi = (unsigned)e.i >= U.ks;  // uniform
  const unsigned so = U.kzo + (hi ? (unsigned)e.i - U.ks : (unsigned)e.i) * U.kcs;
  if (U.has_a) bs32<true>(hi ? U.aB1 : U.aB, act_affs(a, U.af), pb, so);
  if (TRAIN) {
    const float r = a * m - t * m;
    const float wr = valid ? w * r : 0.f;
    if (U.has_g) bs32(hi ? U.gB1 : U.gB, e.gscale * wr * m, pb, so);
    const float red = wave_sum63(wr * r);
    if ((threadIdx.x & 63) == 63) s_part[(threadIdx.x >> 6) * K + e.i] = red;
  }
}

// ------------------------------------------------------------------------------------------------
// forward, tiled.  SELF: e_other == e (own pixel comes out of LDS too).
// ------------------------------------------------------------------------------------------------
template <typename T, int D_T, int TH, int TW, int PLQ, bool CROP, bool TRAIN, bool SELF>
__global__ __launch_bounds__(TH* TW, (D_T > 16 ? 2 : 4)) void k_fwd_tiled(const KParams P, const TParams Q, const T* __restrict__ e,
                                                         const T* __restrict__ eo, const float* __restrict__ target,
                                                         const float* __restrict__ weight,
                                                         const uint8_t* __restrict__ mask, float* __restrict__ affs,
                                                         float* __restrict__ gout, LossState* __restrict__ st, float* __restrict__ inv_out) {
  typedef Lds<D_T, PLQ> L;
  constexpr int NT = TH * TW, NW = NT / 64;
  constexpr int KN = 8;  // near offsets per chunk (chunk 0 is requested before the staging loads)
  const bool rep = CROP && P.border == PEA_BORDER_REPLICATE;  // clamped border (wrap1r): every neighbour exists
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
  U.ks = (unsigned)P.ksplit;
  {
    const size_t ko = (size_t)min(P.ksplit, P.K - 1) * S;  // (ksplit == K: never selected, any valid base will do)
    U.aB1 = mkbuf(affs ? affs + (size_t)b * P.K * S + ko : nullptr);
    U.gB1 = mkbuf(gout ? gout + (size_t)b * P.K * S + ko : nullptr);
    U.tB1 = mkbuf(target + (size_t)b * P.tbs + ko);
    U.wB1 = mkbuf(weight + (size_t)b * P.wbs + ko);
  }
  U.kcs = (unsigned)P.S * 4u;
  U.kzo = (unsigned)z * YX * 4u;
  U.S32 = (unsigned)P.S;
  U.has_a = affs != nullptr; U.has_g = gout != nullptr; U.has_m = mask != nullptr;
  U.af = P.flags & kActMask;
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

  // (1) ALL streaming operands of the first near chunk (KN offsets) and of the first two far offsets: in flight
  //     during the whole staging phase.  Exposed memory round trips per tile: one (the staging loads).
  Twm<KN> sa;
  Twm<2> sf;
  if (TRAIN && Q.n_near > 0) fwd_load_twm<KN>(sa, U, Q.near, 0, Q.n_near, pb, pm);
  if (TRAIN && Q.n_far > 0) fwd_load_twm<2>(sf, U, Q.far, 0, Q.n_far, pb, pm);
  float own[D_T];
  if (!SELF) {
#pragma unroll
    for (int c = 0; c < D_T; ++c) own[c] = bl_emb<T>(eB, pe, ezo + c * ecs);
  }

  // (2) stage the region
  if (SELF && inv_out) stage_region_impl<T, D_T, PLQ, NT, CROP, true>(P, Q, oB, ezo, ecs, y0, x0, lds, mkbuf(inv_out + (size_t)b * S), (unsigned)z * YX * 4u, TH, TW);
  else stage_region<T, D_T, PLQ, NT, CROP>(P, Q, oB, ezo, ecs, y0, x0, lds);

  // (3) the first two far neighbour vectors: in flight across the barrier and the near-offset work
  float fvA[D_T], fvB[D_T];
  bool fokA = false, fokB = false;
#define PEA_FWD_LOAD_FAR(fv, fok, k)                                                               \
  {                                                                                                \
    const OffEnt fe_ = Q.far[k];                                                                   \
    bool okz_, oky_, okx_;                                                                         \
    const int zz_ = wrap1r<CROP>(z + fe_.d, P.Z, okz_, rep);                                       \
    const int yy_ = wrap1r<CROP>(py + ent_oy(fe_), P.Y, oky_, rep);                                \
    const int xx_ = wrap1r<CROP>(px + ent_ox(fe_), P.X, okx_, rep);                                \
    fok = live && okz_ && oky_ && okx_;                                                            \
    const unsigned zo_ = (unsigned)(CROP ? min(max(zz_, 0), P.Z - 1) : zz_) * YX * (unsigned)sizeof(T); \
    const unsigned vo_ = fok ? (unsigned)(yy_ * P.X + xx_) * (unsigned)sizeof(T) : kOOB;           \
    _Pragma("unroll") for (int c = 0; c < D_T; ++c) fv[c] = bl_emb<T>(oB, vo_, zo_ + c * ecs);     \
  }
#define PEA_FWD_FAR(fv, fok, k, u)                                                                 \
  {                                                                                                \
    float dot_ = 0.f, sq_ = 0.f;                                                                   \
    _Pragma("unroll") for (int c = 0; c < D_T; ++c) {                                              \
      dot_ = fmaf(own[c], fv[c], dot_);                                                            \
      sq_ = fmaf(fv[c], fv[c], sq_);                                                               \
    }                                                                                              \
    const float a_ = fok ? dot_ * rnorm(sq_, Q.inv_eps) : 0.f;                                     \
    fwd_finish<TRAIN>(U, P.K, s_part, Q.far[k], a_, fok, sf.t[u], sf.w[u], sf.m[u], pb);           \
  }
  if (Q.n_far > 0) PEA_FWD_LOAD_FAR(fvA, fokA, 0)
  if (Q.n_far > 1) PEA_FWD_LOAD_FAR(fvB, fokB, 1)

  if (!SELF) {
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < D_T; ++c) ss = fmaf(own[c], own[c], ss);
    const float inv = rnorm(ss, Q.inv_eps);
#pragma unroll
    for (int c = 0; c < D_T; ++c) own[c] *= inv;
  }
  lds_barrier();
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
        if (CROP && !rep) {
          const bool inside = (unsigned)(py + ent_oy(en)) < (unsigned)P.Y && (unsigned)(px + ent_ox(en)) < (unsigned)P.X;
          a = inside ? a : 0.f;
          valid = valid && inside;
        }
        fwd_finish<TRAIN>(U, P.K, s_part, en, a, valid, sa.t[u], sa.w[u], sa.m[u], pb);
      }
    }
  }

  // ---- far offsets: neighbour vectors straight from global (L2), two at a time ------------------------
  for (int k = 0; k < Q.n_far; k += 2) {
    if (k > 0) {
      if (TRAIN) fwd_load_twm<2>(sf, U, Q.far, k, Q.n_far, pb, pm);
      PEA_FWD_LOAD_FAR(fvA, fokA, k)
      if (k + 1 < Q.n_far) PEA_FWD_LOAD_FAR(fvB, fokB, k + 1)
    }
    PEA_FWD_FAR(fvA, fokA, k, 0)
    if (k + 1 < Q.n_far) PEA_FWD_FAR(fvB, fokB, k + 1, 1)
  }
#undef PEA_FWD_LOAD_FAR
#undef PEA_FWD_FAR

  if (TRAIN) {
    lds_barrier();
    if (threadIdx.x < P.K) {
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) v += s_part[w * P.K + threadIdx.x];
      loss_accumulate(st, tile, threadIdx.x, v);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// forward, tiled, with the epilogue TRANSPOSED THROUGH LDS (the default forward).
//
// These kernels are bound by the vector-memory pipe: one wave-level load/store costs it ~16-20 cycles whatever
// its width (measured: ~1.2 us of kernel time per VMEM instruction per pixel at this size; a plain 4 B/lane
// streaming kernel with the same access pattern runs at exactly that rate, profiles/microbench).  Reading
// target / weight / mask and writing affs / g one pixel per lane costs 5 VMEM instructions per (pixel, offset).
// Here the dot products are parked in LDS as [offset][tile pixel] instead, and a second phase walks that array
// four x-adjacent pixels per lane: every streaming access becomes a 16-byte-per-lane dwordx4 (one dword for the
// four u8 masks), 1.25 VMEM instructions per (pixel, offset).  The operands of that phase are requested at the
// very top of the kernel (their addresses do not depend on anything computed), so they are in flight during
// staging and the dot products.
// Needs: K <= kKV, X % 4 == 0, 16-byte aligned target / weight / affs / g planes, 4-byte aligned mask planes.
// ------------------------------------------------------------------------------------------------
constexpr int kKV = 12;  // max offsets (near) the register-resident / transposed epilogue handles
constexpr int kFV = 4;   // max far offsets
typedef unsigned u4 __attribute__((ext_vector_type(4)));

// 16-byte buffer store.  HAZARD (gfx950, ROCm 7.2): a VALU write to one of the store's data VGPRs in the very
// next instruction corrupts that dword.  LLVM's hazard recognizer skips the ">64-bit VMEM store data" wait state
// when soffset is an SGPR (as it always is here); seen as affs[..].y == the mask byte converted one instruction
// later.  An explicit s_nop after the store restores the wait states.
template <bool NTL>
__device__ __forceinline__ void bs128(rsrc_t r, f4 v, unsigned vo, unsigned so) {
  const u4 d = __builtin_bit_cast(u4, v);
  __builtin_amdgcn_raw_buffer_store_b128(d, r, vo, so, NTL ? kAuxNT : 0);
  // the data registers are an INPUT of the nop, so no VALU write to them can be scheduled between the two (seen: the
  // scheduler moved `v_mov_b32 v44, 0` of the next quad in between, which zeroed g at scattered pixels)
  asm volatile("s_nop 1" ::"v"(d) : "memory");
}

template <typename T, int D_T, int TH, int TW, int PLQ, bool OVL, bool CROP, bool TRAIN, bool SELF>
__global__ __launch_bounds__(TH* TW, (D_T > 16 ? 2 : 4)) void k_fwd_tiled_v(const KParams P, const TParams Q, const T* __restrict__ e,
                                                           const T* __restrict__ eo, const float* __restrict__ target,
                                                           const float* __restrict__ weight,
                                                           const uint8_t* __restrict__ mask, float* __restrict__ affs,
                                                           float* __restrict__ gout, LossState* __restrict__ st, float* __restrict__ inv_out) {
  typedef Lds<D_T, PLQ> L;
  constexpr int NT = TH * TW, TP = NT, QP = TP / 4, NSL = QP / 64;
  constexpr int ITEMS = (kKV * QP + NT - 1) / NT;
  static_assert(QP % 64 == 0 && TW % 4 == 0, "a wave must cover quads of one offset plane");
  const bool rep = CROP && P.border == PEA_BORDER_REPLICATE;  // clamped border (wrap1r): every neighbour exists
  extern __shared__ f4 lds4[];
  char* lds = (char*)lds4;
  // [K][TP] dot products: OVL = laid over the staged region once every wave is done reading it (smaller LDS
  // footprint: two workgroups per CU), else next to it
  float* sA = OVL ? (float*)lds : (float*)(lds + L::kBytes);
  float* s_part = OVL ? (float*)(lds + L::kBytes) : sA + (size_t)P.K * TP;  // [K][NSL]
  const int tile = tile_id(Q);
  if (tile >= Q.ntiles) return;
  const int plane = tile / Q.tiles_per_plane;
  const int rem = tile - plane * Q.tiles_per_plane;
  const int ty = rem / Q.tiles_x;
  const int y0 = ty * TH, x0 = (rem - ty * Q.tiles_x) * TW;
  const int b = plane / P.Z, z = plane - b * P.Z;
  const size_t S = (size_t)P.S;
  const unsigned YX = (unsigned)(P.Y * P.X);
  const rsrc_t eB = mkbuf(e + (size_t)b * D_T * S), oB = mkbuf(eo + (size_t)b * D_T * S);
  const rsrc_t aB = mkbuf(affs ? affs + (size_t)b * P.K * S : nullptr), gB = mkbuf(gout ? gout + (size_t)b * P.K * S : nullptr);
  const rsrc_t tB = mkbuf(target + (size_t)b * P.tbs), wB = mkbuf(weight + (size_t)b * P.wbs);
  const rsrc_t mB = mkbuf(mask ? mask + (size_t)b * P.mbs : nullptr);
  const unsigned kcs = (unsigned)P.S * 4u, kzo = (unsigned)z * YX * 4u;
  const bool has_a = affs != nullptr, has_g = gout != nullptr, has_m = mask != nullptr;
  const unsigned af = P.flags & kActMask;
  const unsigned ecs = (unsigned)P.S * (unsigned)sizeof(T);
  const unsigned ezo = (unsigned)z * YX * (unsigned)sizeof(T);

  // ---- (0) the epilogue's operands: item = (offset slot, quad of 4 x-adjacent tile pixels), wave-uniform slot
  OffEnt ien[ITEMS];
  bool ion[ITEMS];
  unsigned ivo[ITEMS];   // byte offset of the quad in an f32 plane, kOOB if the quad is outside / slot unused
  int iqd[ITEMS], igy[ITEMS], igx[ITEMS], ioz[ITEMS];
  f4 t4[ITEMS], w4[ITEMS];
  unsigned m4[ITEMS];
#pragma unroll
  for (int it = 0; it < ITEMS; ++it) {
    const int tt = it * NT + (int)threadIdx.x;
    const int s = __builtin_amdgcn_readfirstlane(tt / QP);
    ion[it] = s < P.K;
    const int sc = min(s, P.K - 1);
    ien[it] = sc < Q.n_near ? Q.near[sc] : Q.far[max(sc - Q.n_near, 0)];
    ioz[it] = sc < Q.n_near ? 0 : ien[it].d;  // far entries keep oz in d
    const int qd = tt - (tt / QP) * QP;
    iqd[it] = qd;
    const int l4 = qd * 4;
    igy[it] = y0 + l4 / TW;
    igx[it] = x0 + l4 % TW;
    const bool lv = ion[it] && igy[it] < P.Y && igx[it] < P.X;  // X % 4 == 0: a quad is inside or outside as a whole
    ivo[it] = lv ? (unsigned)(igy[it] * P.X + igx[it]) * 4u : kOOB;
    if (TRAIN) {
      const unsigned so = kzo + (unsigned)ien[it].i * kcs;
      t4[it] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(tB, ivo[it], so, kAuxNT));
      w4[it] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(wB, ivo[it], so, kAuxNT));
      m4[it] = has_m ? __builtin_amdgcn_raw_buffer_load_b32(mB, lv ? ivo[it] >> 2 : kOOB, (kzo >> 2) + (unsigned)ien[it].i * (unsigned)P.S, kAuxNT)
                     : 0x01010101u;
    }
  }

  int ly, lx;
  lane_pixel<TW>(ly, lx);
  const int py = y0 + ly, px = x0 + lx;
  const bool live = py < P.Y && px < P.X;
  const unsigned po = (unsigned)(py * P.X + px);
  const unsigned pe = live ? po * (unsigned)sizeof(T) : kOOB;
  const int pr = (ly + Q.hy0) * Q.RW + lx + Q.hx0;
  float* myA = sA + ly * TW + lx;

  float own[D_T];
  if (!SELF) {
#pragma unroll
    for (int c = 0; c < D_T; ++c) own[c] = bl_emb<T>(eB, pe, ezo + c * ecs);
  }

  // ---- (1) stage the region; (2) first two far vectors in flight across the barrier
  if (SELF && inv_out) stage_region_impl<T, D_T, PLQ, NT, CROP, true>(P, Q, oB, ezo, ecs, y0, x0, lds, mkbuf(inv_out + (size_t)b * S), (unsigned)z * YX * 4u, TH, TW);
  else stage_region<T, D_T, PLQ, NT, CROP>(P, Q, oB, ezo, ecs, y0, x0, lds);
  float fvA[D_T], fvB[D_T];
  bool fokA = false, fokB = false;
#define PEA_FWDV_LOAD_FAR(fv, fok, k)                                                              \
  {                                                                                                \
    const OffEnt fe_ = Q.far[k];                                                                   \
    bool okz_, oky_, okx_;                                                                         \
    const int zz_ = wrap1r<CROP>(z + fe_.d, P.Z, okz_, rep);                                       \
    const int yy_ = wrap1r<CROP>(py + ent_oy(fe_), P.Y, oky_, rep);                                \
    const int xx_ = wrap1r<CROP>(px + ent_ox(fe_), P.X, okx_, rep);                                \
    fok = live && okz_ && oky_ && okx_;                                                            \
    const unsigned zo_ = (unsigned)(CROP ? min(max(zz_, 0), P.Z - 1) : zz_) * YX * (unsigned)sizeof(T); \
    const unsigned vo_ = fok ? (unsigned)(yy_ * P.X + xx_) * (unsigned)sizeof(T) : kOOB;           \
    _Pragma("unroll") for (int c = 0; c < D_T; ++c) fv[c] = bl_emb<T>(oB, vo_, zo_ + c * ecs);     \
  }
#define PEA_FWDV_FAR(fv, fok, k)                                                                   \
  {                                                                                                \
    float dot_ = 0.f, sq_ = 0.f;                                                                   \
    _Pragma("unroll") for (int c = 0; c < D_T; ++c) {                                              \
      dot_ = fmaf(own[c], fv[c], dot_);                                                            \
      sq_ = fmaf(fv[c], fv[c], sq_);                                                               \
    }                                                                                              \
    afar[k] = fok ? dot_ * rnorm(sq_, Q.inv_eps) : 0.f;                                            \
  }
  if (Q.n_far > 0) PEA_FWDV_LOAD_FAR(fvA, fokA, 0)
  if (Q.n_far > 1) PEA_FWDV_LOAD_FAR(fvB, fokB, 1)
  if (!SELF) {
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < D_T; ++c) ss = fmaf(own[c], own[c], ss);
    const float inv = rnorm(ss, Q.inv_eps);
#pragma unroll
    for (int c = 0; c < D_T; ++c) own[c] *= inv;
  }
  lds_barrier();
  if (SELF) lds_pixel<D_T, PLQ>(lds, pr, own);

  // ---- (3) dot products, kept in registers (host guarantees n_near <= kKV, n_far <= kFV)
  float anear[kKV], afar[kFV];
#pragma unroll
  for (int k = 0; k < kKV; ++k) {
    if (k < Q.n_near) {  // uniform
      const OffEnt en = Q.near[k];
      float v[D_T];
      lds_pixel<D_T, PLQ>(lds, pr + en.d, v);
      float a = 0.f;
#pragma unroll
      for (int c = 0; c < D_T; ++c) a = fmaf(own[c], v[c], a);
      if (CROP && !rep) {
        const bool inside = (unsigned)(py + ent_oy(en)) < (unsigned)P.Y && (unsigned)(px + ent_ox(en)) < (unsigned)P.X;
        a = inside ? a : 0.f;
      }
      anear[k] = a;
    }
  }
  if (Q.n_far > 0) PEA_FWDV_FAR(fvA, fokA, 0)
  if (Q.n_far > 1) PEA_FWDV_FAR(fvB, fokB, 1)
  if (Q.n_far > 2) PEA_FWDV_LOAD_FAR(fvA, fokA, 2)
  if (Q.n_far > 3) PEA_FWDV_LOAD_FAR(fvB, fokB, 3)
  if (Q.n_far > 2) PEA_FWDV_FAR(fvA, fokA, 2)
  if (Q.n_far > 3) PEA_FWDV_FAR(fvB, fokB, 3)
#undef PEA_FWDV_LOAD_FAR
#undef PEA_FWDV_FAR
  // -> sA[offset][tile pixel]; with OVL the region must be dead first
  if (OVL) lds_barrier();
#pragma unroll
  for (int k = 0; k < kKV; ++k)
    if (k < Q.n_near) myA[Q.near[k].i * TP] = anear[k];
#pragma unroll
  for (int k = 0; k < kFV; ++k)
    if (k < Q.n_far) myA[Q.far[k].i * TP] = afar[k];
  lds_barrier();

  // ---- (4) epilogue: 4 x-adjacent pixels of one offset per lane, dwordx4 everywhere
#pragma unroll
  for (int it = 0; it < ITEMS; ++it) {
    if (!ion[it]) continue;  // wave-uniform
    const OffEnt en = ien[it];
    const f4 a4 = *(const f4*)(sA + en.i * TP + iqd[it] * 4);
    const unsigned so = kzo + (unsigned)en.i * kcs;
    if (has_a) {
      f4 o = a4;
      if (af) { o.x = act_affs(o.x, af); o.y = act_affs(o.y, af); o.z = act_affs(o.z, af); o.w = act_affs(o.w, af); }
      bs128<true>(aB, o, ivo[it], so);
    }
    if (TRAIN) {
      float acc = 0.f;
      f4 g4;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float m = (float)((m4[it] >> (8 * j)) & 0xffu);
        const float r = a4[j] * m - t4[it][j] * m;
        float wr = w4[it][j] * r;
        if (CROP && !rep) {  // a cropped-away neighbour carries no loss term (its a is already 0)
          const bool inside = (unsigned)(igy[it] + ent_oy(en)) < (unsigned)P.Y && (unsigned)(igx[it] + j + ent_ox(en)) < (unsigned)P.X;
          const bool inz = (unsigned)(z + ioz[it]) < (unsigned)P.Z;
          wr = (inside && inz) ? wr : 0.f;
        }
        g4[j] = en.gscale * wr * m;
        acc = fmaf(wr, r, acc);
      }
      if (has_g) bs128<false>(gB, g4, ivo[it], so);
      const float red = wave_sum63(acc);
      if ((threadIdx.x & 63) == 63) s_part[en.i * NSL + (iqd[it] >> 6)] = red;
    }
  }
  if (TRAIN) {
    lds_barrier();
    if (threadIdx.x < P.K) {
      float v = 0.f;
#pragma unroll
      for (int s = 0; s < NSL; ++s) v += s_part[threadIdx.x * NSL + s];
      loss_accumulate(st, tile, threadIdx.x, v);
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
__global__ __launch_bounds__(TH* TW, (D_T > 16 ? 2 : 4)) void k_bwd_tiled(const KParams P, const TParams Q, const T* __restrict__ xt,
                                                         const T* __restrict__ nbt, const float* __restrict__ gin,
                                                         const float* __restrict__ dloss, T* __restrict__ dx) {
  constexpr int NT = TH * TW;
  constexpr int NR = (ROLE_A ? 1 : 0) + (ROLE_B ? 1 : 0);
  constexpr int KN = 8;  // near offsets per chunk (x NR roles of g values; chunk 0 requested before staging)
  // REPLICATE: role A's neighbour is clamp(p + o) (the staged region holds the clamped pixels, far pairs clamp their index); role B
  // keeps the in-volume test for its regular pre-image p - o and adds what the clamp folds onto border pixels at the end
  const bool rep = CROP && P.border == PEA_BORDER_REPLICATE;
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
  // offset channels >= ksplit: a second resource based ksplit planes further (KParams::ksplit; == K when there is no split)
  const rsrc_t gB1 = mkbuf(gin + (size_t)b * P.K * S + (size_t)min(P.ksplit, P.K - 1) * S);
  const unsigned ksp = (unsigned)P.ksplit;
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
      const int yy_ = wrap1r<CROP>(py + sg_ * ent_oy(en_), P.Y, oky_, rep && sg_ > 0);                         \
      const int xx_ = wrap1r<CROP>(px + sg_ * ent_ox(en_), P.X, okx_, rep && sg_ > 0);                         \
      const bool ok_ = live && oky_ && okx_ && ((k0) + u < Q.n_near);                                          \
      gn[u][r] = bl32((unsigned)en_.i >= ksp ? gB1 : gB, ok_ ? (sg_ > 0 ? po : (unsigned)(yy_ * P.X + xx_)) * 4u : kOOB,     \
                      kzo + ((unsigned)en_.i >= ksp ? (unsigned)en_.i - ksp : (unsigned)en_.i) * kcs);                     \
    }                                                                                                          \
  }

  // (1) own raw pixel and the g values of the first near chunk: in flight during staging
  // self loss (both roles, x == nb): the lane stages its own pixel itself, so x needs no loads of its own
  constexpr bool OWN_STAGED = ROLE_A && ROLE_B;
  float xh[D_T];
  if (!OWN_STAGED) {
#pragma unroll
    for (int c = 0; c < D_T; ++c) xh[c] = bl_emb<T>(xB, pe, ezo + c * ecs);
  }
  float gn[KN][NR];
  if (Q.n_near > 0) PEA_BWD_LOAD_GN(0)

  // (2) stage
  float own_inv = 0.f, own_ss = 0.f;
  if (OWN_STAGED) stage_region_own<T, D_T, PLQ, TH, TW, CROP>(P, Q, nB, ezo, ecs, y0, x0, lds, ly, lx, xh, own_inv, own_ss);
  else stage_region<T, D_T, PLQ, NT, CROP>(P, Q, nB, ezo, ecs, y0, x0, lds);

  // (3) far (offset, role) pairs, two at a time: pair j = (far offset j / NR, role j % NR).  Vectors and g of
  //     the first two pairs are in flight across the barrier and the near-pair work.
  const int n_farp = Q.n_far * NR;
  float fvA[D_T], fvB[D_T], fgA = 0.f, fgB = 0.f;
#define PEA_BWD_LOAD_FAR(fv, fg, j)                                                                           \
  {                                                                                                           \
    const OffEnt fe_ = Q.far[(j) / NR];                                                                       \
    const int sg_ = (ROLE_A && ((j) % NR) == 0) ? 1 : -1;                                                     \
    bool okz_, oky_, okx_;                                                                                    \
    const int zz_ = wrap1r<CROP>(z + sg_ * fe_.d, P.Z, okz_, rep && sg_ > 0);                                 \
    const int yy_ = wrap1r<CROP>(py + sg_ * ent_oy(fe_), P.Y, oky_, rep && sg_ > 0);                          \
    const int xx_ = wrap1r<CROP>(px + sg_ * ent_ox(fe_), P.X, okx_, rep && sg_ > 0);                          \
    const bool ok_ = live && okz_ && oky_ && okx_;                                                            \
    const unsigned zc_ = (unsigned)(CROP ? min(max(zz_, 0), P.Z - 1) : zz_);                                  \
    const unsigned qo_ = (unsigned)(yy_ * P.X + xx_);                                                         \
    const unsigned vo_ = ok_ ? qo_ * (unsigned)sizeof(T) : kOOB;                                              \
    _Pragma("unroll") for (int c = 0; c < D_T; ++c) fv[c] = bl_emb<T>(nB, vo_, zc_ * YX * (unsigned)sizeof(T) + c * ecs); \
    fg = bl32((unsigned)fe_.i >= ksp ? gB1 : gB, ok_ ? (sg_ > 0 ? po : qo_) * 4u : kOOB,                        \
              (sg_ > 0 ? kzo : zc_ * YX * 4u) + ((unsigned)fe_.i >= ksp ? (unsigned)fe_.i - ksp : (unsigned)fe_.i) * kcs); \
  }
#define PEA_BWD_FAR(fv, fg)                                                       \
  {                                                                               \
    float sq_ = 0.f;                                                              \
    _Pragma("unroll") for (int c = 0; c < D_T; ++c) sq_ = fmaf(fv[c], fv[c], sq_); \
    const float g_ = fg * rnorm(sq_, Q.inv_eps);                                  \
    _Pragma("unroll") for (int c = 0; c < D_T; ++c) G[c] = fmaf(g_, fv[c], G[c]); \
  }
  if (n_farp > 0) PEA_BWD_LOAD_FAR(fvA, fgA, 0)
  if (n_farp > 1) PEA_BWD_LOAD_FAR(fvB, fgB, 1)

  float G[D_T];
  float ss = 0.f;
#pragma unroll
  for (int c = 0; c < D_T; ++c) {
    ss = fmaf(xh[c], xh[c], ss);
    G[c] = 0.f;
  }
  if (OWN_STAGED) ss = own_ss;  // xh is already normalised
  const bool tiny = ss < P.eps * P.eps;
  const float invp = OWN_STAGED ? own_inv : rnorm(ss, Q.inv_eps);
  if (!OWN_STAGED) {
#pragma unroll
    for (int c = 0; c < D_T; ++c) xh[c] *= invp;
  }
  lds_barrier();

  // Far pairs 0 and 1 were requested before the barrier: consume them first, then request pairs 2 and 3 so that
  // their round trip hides under the LDS-served near pairs, and consume those last.
  if (n_farp > 0) PEA_BWD_FAR(fvA, fgA)
  if (n_farp > 1) PEA_BWD_FAR(fvB, fgB)
  if (n_farp > 2) PEA_BWD_LOAD_FAR(fvA, fgA, 2)
  if (n_farp > 3) PEA_BWD_LOAD_FAR(fvB, fgB, 3)

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

  // ---- remaining far pairs (an out-of-range pair read zeros: g = 0, vector = 0) ------------------------------
  if (n_farp > 2) PEA_BWD_FAR(fvA, fgA)
  if (n_farp > 3) PEA_BWD_FAR(fvB, fgB)
  for (int j = 4; j < n_farp; j += 2) {
    PEA_BWD_LOAD_FAR(fvA, fgA, j)
    if (j + 1 < n_farp) PEA_BWD_LOAD_FAR(fvB, fgB, j + 1)
    PEA_BWD_FAR(fvA, fgA)
    if (j + 1 < n_farp) PEA_BWD_FAR(fvB, fgB)
  }
#undef PEA_BWD_LOAD_FAR
#undef PEA_BWD_FAR

  // ---- REPLICATE, role B: the clamp is not invertible -- a pixel ON a face of the volume is the clamped neighbour of every p' in a
  //      box (clamp_preimage per axis), of which only p - o (if it is inside) was gathered above.  The other ones are read from global
  //      memory here: face pixels only (the lanes of border tiles; both z = 0 and z = Z - 1 planes when an offset steps along z).
  if (CROP && ROLE_B && rep && live && (z == 0 || z == P.Z - 1 || py == 0 || py == P.Y - 1 || px == 0 || px == P.X - 1)) {
    const T* nbb = nbt + (size_t)b * D_T * S;
    const float* gb = gin + (size_t)b * P.K * S;
    for (int i = 0; i < P.K; ++i) {
      const int oz = P.off[i][0], oy = P.off[i][1], ox = P.off[i][2];
      int z0, z1, ya, yb, xa, xb;
      clamp_preimage(z, oz, P.Z, z0, z1);
      clamp_preimage(py, oy, P.Y, ya, yb);
      clamp_preimage(px, ox, P.X, xa, xb);
      if (z1 < z0 || yb < ya || xb < xa || (z1 == z0 && yb == ya && xb == xa)) continue;  // empty, or p - o alone (done above)
      for (int zz = z0; zz <= z1; ++zz)
        for (int yy = ya; yy <= yb; ++yy)
          for (int xx = xa; xx <= xb; ++xx) {
            if (zz == z - oz && yy == py - oy && xx == px - ox) continue;  // the regular pre-image
            const size_t q2 = ((size_t)zz * P.Y + yy) * P.X + xx;
            float v2[D_T], sq2 = 0.f;
#pragma unroll
            for (int c = 0; c < D_T; ++c) {
              v2[c] = ld(nbb, c * S + q2);
              sq2 = fmaf(v2[c], v2[c], sq2);
            }
            const float g2 = gb[(size_t)i * S + q2] * rnorm(sq2, Q.inv_eps);
#pragma unroll
            for (int c = 0; c < D_T; ++c) G[c] = fmaf(g2, v2[c], G[c]);
          }
    }
  }

  float proj = 0.f;
#pragma unroll
  for (int c = 0; c < D_T; ++c) proj = fmaf(xh[c], G[c], proj);
  if (tiny) proj = 0.f;  // clamp_min branch of F.normalize: d ehat / d e = I / eps
  const float sc = dl * invp;
#pragma unroll
  for (int c = 0; c < D_T; ++c) bs_emb<T, true>(dB, (G[c] - xh[c] * proj) * sc, pe, ezo + c * ecs);
}

}  // namespace pea
