// pea_k_xdma_plan.h -- what the two translation units of the LDS-DMA cross kernels share (pea_k_xdma.hip: f32 storage;
// pea_k_xdma_h.hip: f16 storage and the projection-first backward): tile shape, plane sizes, the memoised plan, the launch macro.
// (One file took 75 s to compile, the longest of the library; each of the two is an independent hipcc job.)
#pragma once
#include "pea_host.h"
#include "pea_xdma.h"

namespace pea {

namespace {

constexpr int kXdmaTH = 16, kXdmaTW = 32;
constexpr int kXdmaPSU = 51;   // 13 KB planes: the backward's two-sided +-27 cross; 6 planes = 78 KB, two workgroups per CU
constexpr int kXdmaPSU3 = 52;  // 3D backward: whole 64-quad blocks
constexpr int kXdmaPSU3F = 32; // 3D forward: the one-sided cross in 8 KB planes (and every ds_read offset of the ring below 2^16: with
                               // 13 KB planes the sixth plane's reads needed computed addresses, which the inference instantiation
                               // spilled -- profiles/kernel_resources.py)
constexpr int kXdmaPSUF = 30;  // the forward's one-sided cross (27 rows up, one strip): 7.5 KB planes, 45 KB, THREE workgroups per CU

constexpr int kXdmaPSUH = 52;   // backward, f16: 13312-byte f32 planes (the half-size ring planes stay whole 256-byte units)
constexpr int kXdmaPSUHS = 28;  // small crosses: 7168-byte planes, 35 KB per workgroup
constexpr int kXdmaPSUS = 27;  // small crosses (reach <= 11 or so): 6912-byte planes, 41 KB of ring, three workgroups per CU

struct XPlan { XParams C; size_t lds; };

// memoised plan_xdma (per thread; keyed by KParams, the plane size and the mode)
// (th x tw: the tile shape; 8 x 64 for the f16 kernels of pea_xdma_hq.h, whose rows are then whole 128-byte lines)
bool plan(const KParams& P, int psu, int mode, XPlan* out, int th = kXdmaTH, int tw = kXdmaTW) {
  static thread_local PlanCache<XPlan, 12> cache;
  return cache.get(P, psu * 4 + mode + (tw == kXdmaTW ? 0 : 4096), out, [&](XPlan* p) {
    if (!plan_xdma(P, th, tw, psu, &p->C, &p->lds, mode)) return false;
    if (env().zblk_y > 0) p->C.zgy = env().zblk_y;
    if (env().zblk_x > 0) p->C.zgx = env().zblk_x;
    if (env().zblk_y < 0) p->C.zrun = 0;  // plane-major walk
    p->C.rev = (mode == 0 || mode == 2) && P.Z == 1 && env().bwd_rev ? 1 : 0;  // backward plans of 2D images
    return true;
  });
}

#define PEA_LAUNCH(kern, grid, blk, lds, s, ...)              \
  {                                                           \
    if (allow_lds<kern>(lds)) return false;                   \
    hipLaunchKernelGGL(kern, grid, blk, lds, s, __VA_ARGS__); \
  }

}  // namespace
}  // namespace pea
