// pea_host.h -- host-side glue shared by the translation units of libpea_hip.so.
//
// The library is built from several .hip files compiled in parallel (one per kernel family; a single file took 1 m 47 s):
//   pea_abi.hip        the extern "C" entry points of include/pea.h: validation, descriptor -> KParams, dispatch
//   pea_k_xdma.hip     LDS-DMA cross kernels (pea_xdma.h): the training forward / backward of axis-aligned stencils, f32 storage
//   pea_k_xdma_h.hip   the same for f16 storage (pea_xdma_h16.h)
//   pea_k_xdma_hq.hip  the f16 backward on producer / consumer waves (pea_xdma_hq.h)
//   pea_k_xdma_pf.hip  the projection-first backward for f32 storage, D = 32 / 64 (pea_xdma_pf.h)
//   pea_k_zmarch.hip   z-march kernels (pea_zmarch.h): 3D volumes with axis-aligned stencils that step along z (norm5 / norm1)
//   pea_k_box.hip      unit-box stencils (pea_box.h): the 26-neighbourhood of a 3D volume through an LDS-DMA ring of 3-plane boxes
//   pea_k_tiled.hip    LDS-tiled box kernels (pea_tiled.h, pea_chunked.h): what the cross / box / march kernels do not take -- diagonal
//                      stencils wider than the unit box, unaligned tensors, X % 4 != 0, a second operand at D >= 32
//   pea_k_labels.hip   the labels-in training step (pea_fused_labels.h) and the label-weight tables
//   pea_k_direct.hip   global-memory kernels (pea_direct.h): the general fallback; loss finish; caller epilogues, stitcher
//   pea_k_head.hip     the embedding head (pea_head.h) and target generation (pea_targets.h)
// Each kernel file exports a few plain functions (declared here) that pick the instantiation and launch it; a function returns
// false when its family has no kernel for the descriptor and the caller tries the next one.  No device code crosses a file.
#pragma once
#include <stdint.h>
#include <string.h>

#include "pea_loss.h"

namespace pea {

// ---- process-wide switches (debugging / A-B runs), read ONCE: getenv on every launch showed up in the host profile ------------
struct Env {
  unsigned gen;       // generation of this switch set (env_generation(): memoised plans carry the generation they were made under)
  int force_direct;   // PEA_FORCE_DIRECT=1: global-memory kernels only
  int fwd_xdma;       // PEA_FWD_XDMA=0: no LDS-DMA forward
  int bwd_xdma;       // PEA_BWD_XDMA=0: no LDS-DMA backward
  int bwd_pf;         // PEA_BWD_PF=0: never the projection-first backward (pea_xdma_pf.h: D = 32 / 64)
  int fwd_wg3;        // PEA_FWD_WG3=0: the 2-workgroups-per-CU forward
  int h16_hw;         // PEA_H16_HW: the f16 kernels' working buffer.  0: f32 (packed-f32 gather); 1: f16 [pixel][2 halves] behind an LDS-DMA
                      // ring; 2 (default): as 1, and the backward of small crosses at D = 32 / 64 on producer / consumer waves (pea_xdma_hq.h)
  int box;            // PEA_BOX=0: unit-box stencils (the 26-neighbourhood) on the tiled kernels instead of pea_box.h
  int zmarch;         // PEA_ZMARCH=0: 3D volumes with z offsets on the tile-per-plane cross kernels (pea_xdma.h) instead of the z-march
                      //   kernels (pea_zmarch.h); 2: the march also on volumes with fewer tile columns than CUs
  int boxm;           // PEA_BOXM=0: the unit-box backward per (z, tile) (pea_box.h) instead of marching (pea_boxm.h)
  int zm_sup;         // PEA_ZM_SUP=sx (1, 2, 4, 8): the z-march kernels' eight XCD blocks of a round lie side by side, (8 / sx) x sx, as one
                      //   super-block (march_tile), blocks of 8 x 2 tile columns unless PEA_ZBLK_* says otherwise; 0: every XCD walks its own
                      //   contiguous range of blocks (rounds 4-5; also for the unit-box kernels); default (-1): the measured best per kernel
                      //   where the tile grid is a whole number of super-blocks (march 4 x 2 XCDs of 8 x 2 blocks, k_fwd_box 4 x 2 of 2 x 8)
  int zseg;           // PEA_ZSEG=n: planes per segment of a tile column (0: whole columns where there are enough of them)
  int zblk_y, zblk_x; // PEA_ZBLK_Y / PEA_ZBLK_X: tiles per block of the z-fastest walk of 3D volumes (0: the default 4 x 2; Y < 0: plane-major)
  int bwd_rev;        // PEA_BWD_REV=0: the 2D cross backward walks every XCD's tile range first tile first.  Default 1: LAST tile first -- what
                      //   the forward touched last (the bottom rows of every image: e read, g and 1 / norm written) is what the backward
                      //   asks for first, and the next forward starts where the backward ended (-1.7 % / -0.8 %, profiles/r5_switch1.txt)
  int fwd_dual;       // PEA_FWD_DUAL=0: pea_affinity_fwd_dual_ex reports PEA_E_UNSUPPORTED (the caller runs the two forwards); default: the
                      //   one-launch pair on a ring of two four-plane buffers handed over in HALVES (e pair, ema pair), two workgroups per CU
};
const Env& env();
void env_reload();          // pea_reload_env(): tests that change a switch call it
unsigned env_generation();  // = env().gen; a new one with every env_reload(): memoised plans that baked a switch in (PEA_ZBLK_*, ..) are dropped

inline bool misaligned(const void* p, size_t a) { return ((uintptr_t)p & (a - 1)) != 0; }
int& g_pending_error();  // per thread: an error met while preparing a launch (allow_lds), reported by the next hip_rc()
inline int hip_rc() {
  int& pe = g_pending_error();
  if (pe) {
    const int r = pe;
    pe = 0;
    (void)hipGetLastError();
    return r;
  }
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? PEA_OK : (int)e;
}
int device_cus();

// Raise a kernel's dynamic-LDS limit above the 64 KB default.  The attribute is per device and the call is a host-side table
// update, so it is made once per (kernel, device) and remembered; a failure is returned (a launch that needs more LDS than it
// was granted would otherwise fail with an opaque error).  Returns hipSuccess (0) or the hipError_t.
int allow_lds_impl(const void* kernel, size_t bytes);
template <auto KERNEL>
inline int allow_lds(size_t bytes) {
  return bytes > 64 * 1024 ? allow_lds_impl((const void*)KERNEL, bytes) : 0;
}

// A small per-thread memo of host-side plans, keyed by the kernel parameters (the deep-supervision launches are latency-bound:
// planning per launch was measurable).  PLAN must be trivially copyable.
template <typename PLAN, int N = 8>
struct PlanCache {
  struct Ent { KParams key; int mode; bool ok, used; PLAN plan; };
  Ent ent[N] = {};
  int next = 0;
  unsigned gen = 0;
  template <typename F>
  bool get(const KParams& P, int mode, PLAN* out, F&& make) {
    const unsigned now = env_generation();
    if (gen != now) {  // a switch changed since these plans were made
      for (int i = 0; i < N; ++i) ent[i].used = false;
      gen = now;
    }
    for (int i = 0; i < N; ++i)
      if (ent[i].used && ent[i].mode == mode && memcmp(&ent[i].key, &P, sizeof(KParams)) == 0) {
        if (ent[i].ok) *out = ent[i].plan;
        return ent[i].ok;
      }
    Ent& e = ent[next];
    next = (next + 1) % N;
    memset(&e.key, 0, sizeof(KParams));
    e.key = P; e.mode = mode; e.used = true;
    e.ok = make(&e.plan);
    if (e.ok) *out = e.plan;
    return e.ok;
  }
};

// ---- loss finish (pea_k_direct.hip) -----------------------------------------------------------------------------------------
void launch_loss_finish(const KParams& P, LossState* st, float* loss_out, hipStream_t s);
void launch_loss_state_init(LossState* st, int n, hipStream_t s);

// Arguments of a forward launch, one struct for every family.
struct FwdArgs {
  const void* e;        // [B, D, S]  f32 / f16 (dtype)
  const void* eo;       // second operand or == e
  const float* t;       // training only
  const float* w;
  const uint8_t* m;     // nullable
  float* affs;          // nullable
  float* gout;          // nullable
  LossState* st;        // training only
  float* loss_out;      // training only: [1 + K]
  float* inv_out;       // nullable: signed 1 / norm plane(s)
  int dtype;            // PEA_F32 / PEA_F16
  bool train;
};
// Each returns true if it launched (the caller then launches the loss finish).
bool xdma_fwd_self(const KParams& P, const FwdArgs& A, hipStream_t s);
bool xdma_fwd_other(const KParams& P, const FwdArgs& A, hipStream_t s);   // inv_out: two planes
bool xdma_labels_supported(const KParams& P, int dtype);
bool xdma_fwd_labels(const KParams& P, const FwdArgs& A, const int32_t* labels, const float* wtab, unsigned lflags, hipStream_t s);
bool tiled_fwd(const KParams& P, const FwdArgs& A, hipStream_t s, bool* wrote_inv);      // k_fwd_tiled_v / k_fwd_tiled / chunked
void direct_fwd(const KParams& P, const FwdArgs& A, hipStream_t s);
void launch_inv_norm(const KParams& P, int dtype, const void* e, float* inv, hipStream_t s);
int xdma_cross_supported(const KParams& P, int dtype, int mode);
// z-march kernels (pea_k_zmarch.hip): f32, D = 16, CROP_ZERO, z offsets in {-1 .. -4}; the backward needs the 1 / norm plane and
// the raw affinity map of the forward
bool zmarch_fwd(const KParams& P, const FwdArgs& A, hipStream_t s);
bool zmarch_bwd_supported(const KParams& P, int dtype);
bool zmarch_bwd(const KParams& P, const float* x, const float* inv, const float* g, const float* affs, const float* dl, float* dx,
                hipStream_t s);
// unit-box stencils (pea_k_box.hip): f32, D = 16, self loss; the backward needs the forward's 1 / norm plane
bool box_supported(const KParams& P, int dtype);
bool box_fwd(const KParams& P, const FwdArgs& A, hipStream_t s);
bool box_bwd(const KParams& P, const float* x, const float* inv, const float* g, const float* dl, float* dx, hipStream_t s);

// backward: roles bit 0 = A (x is the first operand, neighbours nbA), bit 1 = B (x is the second operand, neighbours nbB)
bool xdma_bwd_self(const KParams& P, const float* x, const float* inv, const float* g, const float* affs, const float* dl, float* dx,
                   hipStream_t s);  // affs: the raw cosine map or null
bool xdma_bwd_self_h(const KParams& P, const void* x, const float* inv, const float* g, const float* affs, const float* dl, void* dx,
                     hipStream_t s);  // f16 storage (pea_k_xdma_h.hip)
bool xdma_h_fwd_self(const KParams& P, const FwdArgs& A, hipStream_t s);  // f16 storage forward / inference (pea_k_xdma_h.hip)
// f16 storage, the cross loss with a detached second operand (pea_k_xdma_h.hip): forward (two 1 / norm planes) and the role-A backward
// (projection first: needs the cross loss' raw map)
bool xdma_h_fwd_other(const KParams& P, const FwdArgs& A, hipStream_t s);
bool xdma_h_bwd_other(const KParams& P, const void* e, const void* e_other, const float* inv2, const float* g, const float* affs,
                      const float* dl, void* de, hipStream_t s);
// f16 storage backward with producer / consumer waves (pea_k_xdma_hq.hip; D = 32 / 64, small crosses, PEA_H16_HW=2)
bool xdma_hq_bwd_self(const KParams& P, const void* x, const float* inv, const float* g, const float* affs, const float* dl, void* dx,
                      hipStream_t s);
bool xdma_pf_bwd_self(const KParams& P, const float* x, const float* inv, const float* g, const float* affs, const float* dl, float* dx,
                      hipStream_t s);  // the projection-first backward, f32 storage (pea_k_xdma_pf.hip)
bool xdma_bwd_other(const KParams& P, const float* e, const float* e_other, const float* inv2, const float* g, const float* affs,
                    const float* dl, float* de, bool accumulate, hipStream_t s);  // affs: the raw cosine map or null (read at D > 16)
bool xdma_pf_bwd_other(const KParams& P, const float* e, const float* e_other, const float* inv2, const float* g, const float* affs,
                       const float* dl, float* de, hipStream_t s);  // D = 32 / 64, projection first (pea_k_xdma_pf.hip)
// the full-resolution pair (self loss + detached-EMA cross loss on the same target / weight / mask) as ONE forward launch
// (pea_xdma_dual.h): A = the self loss' arguments (affs, gout, st, inv_out: the own plane), A2 = the cross loss' (eo = ema, gout,
// st, inv_out: the second operand's plane); P2 differs from P in lambda only.  true = launched.
bool xdma_fwd_dual_supported(const KParams& P, int dtype);
bool xdma_fwd_dual(const KParams& P, const KParams& P2, const FwdArgs& A, const FwdArgs& A2, hipStream_t s);
bool xdma_bwd_dual(const KParams& P, const float* e, const float* ema, const float* inv, const float* inv_other, const float* g,
                   const float* g_cross, const float* dl, const float* dl_cross, float* de, hipStream_t s);
bool tiled_bwd(const KParams& P, int dtype, int roles, const void* x, const void* nbA, const void* nbB, const float* g,
               const float* dl, void* dx, hipStream_t s);
int direct_bwd(const KParams& P, int dtype, int roles, const void* x, const void* nbA, const void* nbB, const float* g,
               const float* dl, void* dx, hipStream_t s);   // PEA_OK / PEA_E_UNSUPPORTED / hip error

// labels-in step (pea_k_labels.hip)
bool labels_step(const KParams& P, int dtype, const void* e, const void* e_other, const int32_t* labels, const float* wtab,
                 unsigned lflags, float* affs, LossState* st, const float* dl, void* de, hipStream_t s);
bool labels_step_dual(const KParams& P, const KParams& P2, int dtype, const void* e, const void* ema, const int32_t* labels,
                      const float* wtab, unsigned lflags, float* affs, LossState* st, LossState* st2, const float* dl,
                      const float* dl2, void* de, hipStream_t s);
size_t label_counts_bytes(const PeaDesc* d);
int label_weights(const PeaDesc* d, const int32_t* labels, unsigned flags, float* wtab, void* ws, hipStream_t s);
int gen_targets(const PeaDesc* d, const int32_t* labels, unsigned flags, float* target, uint8_t* mask, float* weight, void* ws,
                size_t need, hipStream_t s);

// caller epilogues, stitcher, rescale (pea_k_direct.hip); head and target generation (pea_k_head.hip): the ABI functions
// themselves live in those files (they need no descriptor logic)

}  // namespace pea
