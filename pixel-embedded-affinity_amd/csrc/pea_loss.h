// pea_loss.h -- the loss reduction: order-independent, exact, and off the critical path.
//
// L_i = sum_{b,p} w_i (a_i m_i - t_i m_i)^2 / N_i  (WeightedMSE, scripts_cvppp/loss/loss.py:106-124) is a sum over every pixel of
// the batch; a forward kernel holds one f32 partial per (workgroup, offset).  Rounds 1-2 wrote those partials to a [K][ntiles]
// table and summed it in a second launch (one workgroup, fixed order, f64): 7.7 us + a kernel boundary on a 107 us forward,
// because one CU pulls a 185 KB table at ~25 GB/s however the loads are arranged.
//
// Now every workgroup ADDS its partials into a small table of 64-bit INTEGER accumulators with agent-scope atomics: the f32
// partial is converted exactly to a 128-bit fixed-point number (LSB 2^-64) and its three digits (low 32 bits, middle 32 bits,
// upper 64 bits, two's complement) are added to three u64 words.  Integer addition is associative and commutative, so the sum is
// the same in every arrival order -- bit-reproducible like the fixed-order tree it replaces, and independent of the grid too --
// and exact: no rounding happens between the f32 partial and the final f64.  A digit word takes 2^32 additions before it can wrap.
// The table is sharded kLossSlots ways by tile number so that no address sees more than ntiles / 16 adds.
//
// What is left for the end is 16 x K x 3 words: k_loss_finish, one launch of ONE wave, reads the table, writes loss_out and
// zeroes the table again.
// (Built, measured and removed in round 3: finishing inside the forward kernel -- every workgroup's wave 0 waits for its adds,
//  takes a ticket with a returning atomic, the last ticket holder reads the table with atomic exchanges.  Bit-identical results,
//  no faster in the step on one box (114.8 us either way), 3x SLOWER on another (332 us, and 331 us under rocprofv3 anywhere):
//  4624 returning atomics on one word serialise at the memory side at a rate that differs by box.  DESIGN.md section 5.8.)
//
// Contract of the state block (include/pea.h, pea_workspace_init): it is ZERO between calls (except `magic`); the finish puts
// it back to zero.  A block without the magic word (never initialised) yields NaN losses instead of silently wrong ones.
#pragma once
#include "pea_common.h"

namespace pea {

constexpr int kLossSlots = 16;
constexpr unsigned kLossMagic = 0x50454133u;  // "PEA3"

struct LossState {
  unsigned magic;
  unsigned reserved;
  unsigned pad[14];
  unsigned flags[PEA_MAX_K];                            // bit 0: +inf / overflow, bit 1: -inf, bit 2: NaN among the partials
  unsigned long long acc[kLossSlots][PEA_MAX_K][4];     // [0] sum of low digits, [1] of middle digits, [2] of upper words
};
static_assert(sizeof(LossState) % 64 == 0, "states are laid out back to back in a workspace");

typedef unsigned long long u64;
#define PEA_ATOM_ADD(p, v) (void)__hip_atomic_fetch_add((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)

// one workgroup's partial of offset k: st->acc[tile % 16][k] += v * 2^64 (exactly; |v| < 2^60, smaller than 2^-64 counts as 0)
__device__ __forceinline__ void loss_accumulate(LossState* __restrict__ st, int tile, int k, float v) {
  const unsigned u = __builtin_bit_cast(unsigned, v);
  const int e = (int)((u >> 23) & 0xffu);
  if (e == 0) return;  // zero / denormal
  if (e >= 127 + 60) {  // overflow, infinity, NaN: remembered as a flag, the finish reports inf / NaN like a float sum would
    const unsigned f = (e == 0xff && (u & 0x7fffffu)) ? 4u : ((u >> 31) ? 2u : 1u);
    (void)__hip_atomic_fetch_or(&st->flags[k], f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  const u64 m = (u64)((u & 0x7fffffu) | 0x800000u);
  const int sh = e - (127 + 23) + 64;  // value = m * 2^(sh - 64)
  unsigned __int128 X;
  if (sh >= 0) X = (unsigned __int128)m << sh;
  else if (sh > -24) X = (unsigned __int128)(m >> (-sh));
  else return;
  if (u >> 31) X = (unsigned __int128)0 - X;
  const u64 lo = (u64)X & 0xffffffffull, mid = (u64)(X >> 32) & 0xffffffffull, hi = (u64)(X >> 64);
  u64* a = st->acc[tile & (kLossSlots - 1)][k];
  if (lo) PEA_ATOM_ADD(a + 0, lo);
  if (mid) PEA_ATOM_ADD(a + 1, mid);
  if (hi) PEA_ATOM_ADD(a + 2, hi);
}

// digit sums of one offset -> the value (double).  lo / mid: sums of 32-bit digits; hi: sum of the upper words (mod 2^64)
__device__ __forceinline__ double loss_value(u64 lo_s, u64 mid_s, u64 hi_s, unsigned flags) {
  mid_s += lo_s >> 32;
  const u64 lo = lo_s & 0xffffffffull;
  hi_s += mid_s >> 32;
  const u64 mid = mid_s & 0xffffffffull;
  u64 H = hi_s, F = (mid << 32) | lo;  // integer part (two's complement), 64-bit fraction
  const bool neg = (long long)H < 0;
  if (neg) {
    F = ~F + 1ull;
    H = ~H + (F == 0 ? 1ull : 0ull);
  }
  double d = (double)H + (double)F * 0x1p-64;
  if (neg) d = -d;
  if (flags) {
    const double inf = __builtin_huge_val();
    if ((flags & 4u) || ((flags & 3u) == 3u)) d = __builtin_nan("");
    else d = (flags & 1u) ? inf : -inf;
  }
  return d;
}

// The finish: one workgroup of ceil(K / 4) waves, four offsets per wave (lane = 16 * j + s: offset 4 * wave + j, slot s), so every
// accumulator word is requested in ONE round of loads (the one-wave form walked the offsets four at a time: three dependent round
// trips at K = 10, 10 us per launch in the loss section's trace); the weighted total is summed in offset order by one lane.
// Plain loads and plain zero stores: a kernel boundary lies between the adds and these reads.
static __global__ __launch_bounds__(512) void k_loss_finish(const KParams P, LossState* __restrict__ st, float* __restrict__ loss_out) {
  __shared__ double s_l[PEA_MAX_K];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, s = lane & (kLossSlots - 1), j = lane >> 4;
  const bool good = st->magic == kLossMagic;
  const int k = 4 * wave + j;
  const bool on = k < P.K;
  u64 v0 = 0, v1 = 0, v2 = 0;
  unsigned fl = 0;
  if (on) {
    u64* a = st->acc[s][k];
    v0 = a[0]; v1 = a[1]; v2 = a[2];
    a[0] = 0; a[1] = 0; a[2] = 0;
    if (s == 0) { fl = st->flags[k]; st->flags[k] = 0; }
  }
#pragma unroll
  for (int o = 1; o < kLossSlots; o <<= 1) {  // the 16 slots of an offset sit in 16 adjacent lanes
    v0 += __shfl_xor(v0, o, 64);
    v1 += __shfl_xor(v1, o, 64);
    v2 += __shfl_xor(v2, o, 64);
    fl |= __shfl_xor(fl, o, 64);
  }
  double Li = on ? loss_value(v0, v1, v2, fl) * (double)P.inv_n[k] : 0.0;
  if (!good) Li = __builtin_nan("");  // the state block was never initialised (pea_workspace_init): say so
  if (on && s == 0) {
    loss_out[1 + k] = (float)Li;
    s_l[k] = Li;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double tot = 0.0;
    for (int i = 0; i < P.K; ++i) tot += (double)P.lam[i] * s_l[i];
    loss_out[0] = (float)tot;
  }
}

// pea_workspace_init: zero `n` states and mark them initialised
static __global__ __launch_bounds__(256) void k_loss_state_init(LossState* __restrict__ st, int n) {
  unsigned* w = (unsigned*)st;
  const size_t words = (size_t)n * (sizeof(LossState) / 4);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < words; i += (size_t)gridDim.x * 256)
    w[i] = (i % (sizeof(LossState) / 4) == 0) ? kLossMagic : 0u;
}

}  // namespace pea
