// pea_k_direct.hip -- launchers of the direct (global-memory) kernels (pea_direct.h: the general fallback), the loss finish
// (pea_loss.h), and the small streaming kernels around the path with their entry points: gradient rescale, the callers' relu /
// border-fill epilogue, the 3D stitcher.  Also the process-wide host state (switches, the LDS-attribute memo).
// One translation unit of libpea_hip.so (pea_host.h).
#include <stdlib.h>

#include <algorithm>
#include <atomic>
#include <mutex>
#include <vector>

#include "pea_host.h"
#include "pea_direct.h"
#include "pea_tiled.h"  // f4

using namespace pea;

namespace pea {

// ---- switches ---------------------------------------------------------------------------------------------------------------
namespace {
int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return v && *v ? atoi(v) : dflt;
}
// The switch set is immutable once published: env_reload() builds a NEW set and swaps the pointer (the old one is left in place: a
// handful of bytes per reload, and reloads happen in tests only), so a launch on another thread sees the old set or the new one, never
// a half-written mix (round-4 advice: a plan sized for one ring depth launched with another).
std::atomic<const Env*> g_env_cur{nullptr};
std::mutex g_env_mu;
std::atomic<unsigned> g_env_gen{0};
const Env* env_make() {
  Env* e = new Env();
  e->gen = g_env_gen.fetch_add(1, std::memory_order_relaxed) + 1;  // published WITH the set: a plan made from it is memoised under it
  e->force_direct = env_int("PEA_FORCE_DIRECT", 0);
  e->fwd_xdma = env_int("PEA_FWD_XDMA", 1);
  e->bwd_xdma = env_int("PEA_BWD_XDMA", 1);
  e->fwd_wg3 = env_int("PEA_FWD_WG3", 1);
  e->bwd_pf = env_int("PEA_BWD_PF", 1);
  e->box = env_int("PEA_BOX", 1);
  e->h16_hw = env_int("PEA_H16_HW", 2);
  e->zmarch = env_int("PEA_ZMARCH", 1);
  e->zseg = env_int("PEA_ZSEG", 0);
  e->boxm = env_int("PEA_BOXM", 1);
  e->zm_sup = env_int("PEA_ZM_SUP", -1);
  e->zblk_y = env_int("PEA_ZBLK_Y", 0);
  e->zblk_x = env_int("PEA_ZBLK_X", 0);
  e->bwd_rev = env_int("PEA_BWD_REV", 1);
  e->fwd_dual = env_int("PEA_FWD_DUAL", 1);
  return e;
}
}  // namespace
const Env& env() {
  const Env* e = g_env_cur.load(std::memory_order_acquire);
  if (!e) {
    std::lock_guard<std::mutex> lk(g_env_mu);
    e = g_env_cur.load(std::memory_order_acquire);
    if (!e) {
      e = env_make();
      g_env_cur.store(e, std::memory_order_release);
    }
  }
  return *e;
}
void env_reload() {
  std::lock_guard<std::mutex> lk(g_env_mu);
  g_env_cur.store(env_make(), std::memory_order_release);
}
unsigned env_generation() { return env().gen; }

// CUs of the CURRENT device (asked every time: the reference runs replicas under nn.DataParallel threads, one device each, so a
// process-wide cache of the first device's answer would be wrong for the others)
int device_cus() {
  int dev = 0, v = 0;
  if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
    return v;
  (void)hipGetLastError();
  return 256;
}

// hipFuncAttributeMaxDynamicSharedMemorySize is kept per (kernel, device): remembered per pair so that the attribute call (a
// driver round trip of a few microseconds) is made once, not per launch -- and its failure is reported
int allow_lds_impl(const void* kernel, size_t bytes) {
  struct Key { const void* k; int dev; size_t bytes; };
  static std::mutex mu;
  static std::vector<Key> done;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return (int)hipGetLastError();
  {
    std::lock_guard<std::mutex> lock(mu);
    for (const Key& e : done)
      if (e.k == kernel && e.dev == dev && e.bytes >= bytes) return 0;
  }
  const hipError_t rc = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (rc != hipSuccess) {
    (void)hipGetLastError();
    g_pending_error() = (int)rc;
    return (int)rc;
  }
  std::lock_guard<std::mutex> lock(mu);
  done.push_back(Key{kernel, dev, bytes});
  return 0;
}

int& g_pending_error() {
  static thread_local int e = 0;
  return e;
}

// ---- loss finish ------------------------------------------------------------------------------------------------------------
void launch_loss_finish(const KParams& P, LossState* st, float* loss_out, hipStream_t s) {
  hipLaunchKernelGGL(k_loss_finish, dim3(1), dim3(64u * (unsigned)((P.K + 3) / 4)), 0, s, P, st, loss_out);
}
void launch_loss_state_init(LossState* st, int n, hipStream_t s) {
  const size_t words = (size_t)n * (sizeof(LossState) / 4);
  hipLaunchKernelGGL(k_loss_state_init, dim3((unsigned)std::min<size_t>((words + 255) / 256, 1024)), dim3(256), 0, s, st, n);
}

// ---- direct kernels ---------------------------------------------------------------------------------------------------------
namespace {
template <typename T, bool TRAIN>
void fwd_direct(const KParams& P, const FwdArgs& A, hipStream_t s) {
  const T *ep = (const T*)A.e, *op = (const T*)A.eo;
  const size_t lds = TRAIN ? (size_t)P.K * kBlock * sizeof(float) : 0;
  const dim3 g((unsigned)(P.tiles_per_xcd * kXcd)), blk(kBlock);
  switch (P.D) {
    case 16: hipLaunchKernelGGL((k_fwd_direct<T, 16, TRAIN>), g, blk, lds, s, P, ep, op, A.t, A.w, A.m, A.affs, A.gout, A.st); break;
    case 32: hipLaunchKernelGGL((k_fwd_direct<T, 32, TRAIN>), g, blk, lds, s, P, ep, op, A.t, A.w, A.m, A.affs, A.gout, A.st); break;
    case 64: hipLaunchKernelGGL((k_fwd_direct<T, 64, TRAIN>), g, blk, lds, s, P, ep, op, A.t, A.w, A.m, A.affs, A.gout, A.st); break;
    default: hipLaunchKernelGGL((k_fwd_direct<T, 0, TRAIN>), g, blk, lds, s, P, ep, op, A.t, A.w, A.m, A.affs, A.gout, A.st); break;
  }
}

template <typename T, int D_T>
void bwd_direct_roles(const KParams& P, int roles, const T* x, const T* nbA, const T* nbB, const float* g, const float* dl, T* dx,
                      hipStream_t s) {
  const dim3 grid((unsigned)(P.tiles_per_xcd * kXcd)), blk(kBlock);
  if (roles == 3) hipLaunchKernelGGL((k_bwd_direct<T, D_T, true, true>), grid, blk, 0, s, P, x, nbA, nbB, g, dl, dx);
  else if (roles == 1) hipLaunchKernelGGL((k_bwd_direct<T, D_T, true, false>), grid, blk, 0, s, P, x, nbA, nbB, g, dl, dx);
  else hipLaunchKernelGGL((k_bwd_direct<T, D_T, false, true>), grid, blk, 0, s, P, x, nbA, nbB, g, dl, dx);
}

template <typename T>
int bwd_direct(const KParams& P, int roles, const void* x, const void* nbA, const void* nbB, const float* g, const float* dl, void* dx,
               hipStream_t s) {
  switch (P.D) {
    case 16: bwd_direct_roles<T, 16>(P, roles, (const T*)x, (const T*)nbA, (const T*)nbB, g, dl, (T*)dx, s); return hip_rc();
    case 32: bwd_direct_roles<T, 32>(P, roles, (const T*)x, (const T*)nbA, (const T*)nbB, g, dl, (T*)dx, s); return hip_rc();
    case 64: bwd_direct_roles<T, 64>(P, roles, (const T*)x, (const T*)nbA, (const T*)nbB, g, dl, (T*)dx, s); return hip_rc();
    case 4: bwd_direct_roles<T, 4>(P, roles, (const T*)x, (const T*)nbA, (const T*)nbB, g, dl, (T*)dx, s); return hip_rc();
    case 8: bwd_direct_roles<T, 8>(P, roles, (const T*)x, (const T*)nbA, (const T*)nbB, g, dl, (T*)dx, s); return hip_rc();
    default: break;
  }
  // any other width: the runtime-D kernel; the REPLICATE border keeps the specialised kernels
  if (P.border == PEA_BORDER_REPLICATE) return PEA_E_UNSUPPORTED;
  const size_t lds = (size_t)2 * P.K * kBlock * sizeof(float);
  const dim3 grid((unsigned)(P.tiles_per_xcd * kXcd)), blk(kBlock);
#define PEA_ANYD(RA_, RB_)                                                                                           \
  {                                                                                                                  \
    constexpr auto kern = k_bwd_direct_anyd<T, RA_, RB_>;                                                            \
    const int rc = allow_lds<kern>(lds);                                                                             \
    if (rc) return rc;                                                                                               \
    hipLaunchKernelGGL(kern, grid, blk, lds, s, P, (const T*)x, (const T*)nbA, (const T*)nbB, g, dl, (T*)dx);        \
  }
  if (roles == 3) PEA_ANYD(true, true) else if (roles == 1) PEA_ANYD(true, false) else PEA_ANYD(false, true)
#undef PEA_ANYD
  return hip_rc();
}
}  // namespace

void direct_fwd(const KParams& P, const FwdArgs& A, hipStream_t s) {
  if (A.dtype == PEA_F16) { if (A.train) fwd_direct<__half, true>(P, A, s); else fwd_direct<__half, false>(P, A, s); }
  else { if (A.train) fwd_direct<float, true>(P, A, s); else fwd_direct<float, false>(P, A, s); }
}

int direct_bwd(const KParams& P, int dtype, int roles, const void* x, const void* nbA, const void* nbB, const float* g,
               const float* dl, void* dx, hipStream_t s) {
  return dtype == PEA_F16 ? bwd_direct<__half>(P, roles, x, nbA, nbB, g, dl, dx, s) : bwd_direct<float>(P, roles, x, nbA, nbB, g, dl, dx, s);
}

}  // namespace pea

// ---- small streaming kernels around the path ------------------------------------------------------------------------------
namespace {

template <typename T>
__global__ __launch_bounds__(256) void k_scale_inplace(T* __restrict__ buf, size_t n4, size_t n, const float* __restrict__ scale) {
  const float sc = scale[0];
  if (sc == 1.0f) return;  // the common loss.backward() case: nothing to do, nothing touched
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (sizeof(T) == 4) {
    if (i < n4) {
      f4 v = ((f4*)buf)[i];
      v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc;
      ((f4*)buf)[i] = v;
    }
    if (i < n - 4 * n4) ((float*)buf)[4 * n4 + i] *= sc;
  } else {
    for (size_t k = i * 4; k < min(n, i * 4 + 4); ++k) st(buf, k, ld(buf, k) * sc);
  }
}

// Caller epilogue of the 3D path (scripts_ac3ac4/main.py:233-237,296-300; inference.py:160-164): for c in {0,1,2} the
// first `shift` slices of affs[:, c] along axis c (z, y, x) are overwritten with slices shift .. 2*shift-1, then relu.
// shift == 0: plain F.relu in place, 4 floats per lane (the general kernel below spends its time on index divisions)
__global__ __launch_bounds__(256) void k_relu_inplace(float* __restrict__ a, size_t n4, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n4) {
    f4 v = ((f4*)a)[i];
    v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
    ((f4*)a)[i] = v;
  }
  if (i < n - 4 * n4) a[4 * n4 + i] = fmaxf(a[4 * n4 + i], 0.f);
}

__global__ __launch_bounds__(256) void k_fill_border_relu(float* __restrict__ affs, int B, int K, int Z, int Y, int X, int shift,
                                                          int relu) {
  const size_t S = (size_t)Z * Y * X, n = (size_t)B * K * S;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int c = (int)((i / S) % (size_t)K);
  const size_t p = i % S;
  const int z = (int)(p / ((size_t)Y * X)), y = (int)((p / X) % Y), x = (int)(p % X);
  size_t src = i;
  if (shift > 0 && c < 3) {
    const int a = c == 0 ? z : c == 1 ? y : x;
    const size_t stride = c == 0 ? (size_t)Y * X : c == 1 ? (size_t)X : 1;
    if (a < shift) src = i + (size_t)shift * stride;  // pred[..., :shift] = pred[..., shift:2*shift]
  }
  float v = affs[src];
  if (relu) v = fmaxf(v, 0.f);
  if (src != i || relu) affs[i] = v;
}

// the same, four x-adjacent voxels per lane (X % 4 == 0, 16-byte aligned map): the z and y fills move whole quads, the x fill (its first
// `shift` columns) patches single values.  (The per-voxel kernel above took 34 us on the reference's training batch -- 44 MB -- : 2.6 TB/s,
// spent on index divisions.)
__global__ __launch_bounds__(256) void k_fill_border_relu_v4(float* __restrict__ affs, int K, int Z, int Y, int X, int shift, size_t n4) {
  const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (q >= n4) return;
  const unsigned X4 = (unsigned)X >> 2;
  const size_t row = q / X4;  // (b * K + c) * Z * Y + z * Y + y
  const int x = (int)(q - row * X4) * 4;
  const size_t pl = row / (unsigned)Y;
  const int y = (int)(row - pl * (unsigned)Y);
  const size_t bc = pl / (unsigned)Z;
  const int z = (int)(pl - bc * (unsigned)Z), c = (int)(bc % (unsigned)K);
  const size_t i = q * 4;
  size_t src = i;
  if (c == 0 && z < shift) src = i + (size_t)shift * Y * X;
  if (c == 1 && y < shift) src = i + (size_t)shift * X;
  f4 v = *(const f4*)(affs + src);
  if (c == 2 && x < shift) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (x + j < shift) v[j] = affs[i + j + shift];
  }
  v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
  *(f4*)(affs + i) = v;
}

// the border fill alone (relu == 0, or a map the forward already clamped): one lane per BORDER voxel of the three channels
__global__ __launch_bounds__(256) void k_fill_border_only(float* __restrict__ affs, int B, int K, int Z, int Y, int X, int shift, int relu) {
  const size_t n0 = (size_t)shift * Y * X, n1 = K > 1 ? (size_t)Z * shift * X : 0, n2 = K > 2 ? (size_t)Z * Y * shift : 0, per_b = n0 + n1 + n2;
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= per_b * (size_t)B) return;
  const size_t b = t / per_b, S = (size_t)Z * Y * X;
  size_t r = t - b * per_b;
  int c, z, y, x;
  size_t stride;
  if (r < n0) { c = 0; z = (int)(r / ((size_t)Y * X)); r -= (size_t)z * Y * X; y = (int)(r / X); x = (int)(r - (size_t)y * X); stride = (size_t)Y * X; }
  else if (r < n0 + n1) { r -= n0; c = 1; z = (int)(r / ((size_t)shift * X)); r -= (size_t)z * shift * X; y = (int)(r / X); x = (int)(r - (size_t)y * X); stride = (size_t)X; }
  else { r -= n0 + n1; c = 2; z = (int)(r / ((size_t)Y * shift)); r -= (size_t)z * Y * shift; y = (int)(r / shift); x = (int)(r - (size_t)y * shift); stride = 1; }
  const size_t dst = ((b * K + c) * Z + z) * (size_t)Y * X + (size_t)y * X + x;
  (void)S;
  float v = affs[dst + (size_t)shift * stride];
  if (relu) v = fmaxf(v, 0.f);
  affs[dst] = v;
}

// 3D inference stitcher (scripts_ac3ac4/data/provider_valid.py:320-349): out[:, window] += vol * w ; wmap[window] += w,
// then out /= wmap.  Product and sum are rounded separately (no FMA) so the result is bit-identical to numpy's.
__global__ __launch_bounds__(256) void k_stitch_add(float* __restrict__ out, float* __restrict__ wmap, const float* __restrict__ vol,
                                                    const float* __restrict__ wv, int C, int Z, int Y, int X, int oz, int oy, int ox,
                                                    int z0, int y0, int x0) {
  const size_t n = (size_t)oz * oy * ox;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int z = (int)(i / ((size_t)oy * ox)), y = (int)((i / ox) % oy), x = (int)(i % ox);
  const size_t o = ((size_t)(z0 + z) * Y + (y0 + y)) * X + (x0 + x), S = (size_t)Z * Y * X;
  const float w = wv[i];
  for (int c = 0; c < C; ++c) {
    float prod = vol[c * n + i] * w;
    asm volatile("" : "+v"(prod));  // keep the product a rounded f32: hipcc would contract a * b + c into one FMA
    out[c * S + o] = out[c * S + o] + prod;
  }
  wmap[o] = wmap[o] + w;
}

__global__ __launch_bounds__(256) void k_stitch_finalize(float* __restrict__ out, const float* __restrict__ wmap, int C, size_t S) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= S) return;
  const float w = wmap[i];
  for (int c = 0; c < C; ++c) out[c * S + i] = __fdiv_rn(out[c * S + i], w);
}

struct ScaleMulti { void* buf[8]; unsigned long long n[8]; };
// up to 8 buffers in one launch (blockIdx.y = buffer): the gradients of one loss section share their grad_output
template <typename T>
__global__ __launch_bounds__(256) void k_scale_multi(const ScaleMulti M, const float* __restrict__ scale) {
  const float sc = scale[0];
  if (sc == 1.0f) return;  // the loss.backward() case: a few hundred workgroups that read one float
  T* buf = (T*)M.buf[blockIdx.y];
  const size_t n = (size_t)M.n[blockIdx.y];
  const size_t stride = (size_t)gridDim.x * 256, t = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (sizeof(T) == 4 && (((uintptr_t)buf) & 15) == 0) {
    const size_t n4 = n / 4;
    for (size_t i = t; i < n4; i += stride) {
      f4 v = ((f4*)buf)[i];
      v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc;
      ((f4*)buf)[i] = v;
    }
    for (size_t i = 4 * n4 + t; i < n; i += stride) st(buf, i, ld(buf, i) * sc);
  } else {
    for (size_t i = t; i < n; i += stride) st(buf, i, ld(buf, i) * sc);
  }
}

}  // namespace

extern "C" {

int pea_scale_inplace(void* buf, int dtype, size_t n, const float* scale, void* stream) {
  if (!buf || !scale) return PEA_E_NULL;
  if (dtype != PEA_F32 && dtype != PEA_F16) return PEA_E_DESC;
  if (misaligned(buf, dtype == PEA_F32 ? 16 : 2) || misaligned(scale, 4)) return PEA_E_ALIGN;
  if (n == 0) return PEA_OK;
  hipStream_t s = (hipStream_t)stream;
  const size_t n4 = dtype == PEA_F32 ? n / 4 : 0;
  const size_t items = dtype == PEA_F32 ? std::max(n4, n - 4 * n4) : (n + 3) / 4;
  const size_t blocks = (items + 255) / 256;
  if (blocks > 0x7fffffffULL) return PEA_E_UNSUPPORTED;
  if (dtype == PEA_F32) hipLaunchKernelGGL(k_scale_inplace<float>, dim3((unsigned)blocks), dim3(256), 0, s, (float*)buf, n4, n, scale);
  else hipLaunchKernelGGL(k_scale_inplace<__half>, dim3((unsigned)blocks), dim3(256), 0, s, (__half*)buf, n4, n, scale);
  return hip_rc();
}

int pea_fill_border_relu(float* affs, int B, int K, int Z, int Y, int X, int shift, int relu, void* stream) {
  if (!affs) return PEA_E_NULL;
  if (B < 1 || K < 1 || Z < 1 || Y < 1 || X < 1 || shift < 0) return PEA_E_DESC;
  if (shift > 0 && K >= 3 && (2 * shift > Z || 2 * shift > Y || 2 * shift > X)) return PEA_E_DESC;
  if (misaligned(affs, 4)) return PEA_E_ALIGN;
  const size_t n = (size_t)B * K * Z * Y * X, blocks = (n + 255) / 256;
  if (blocks > 0x7fffffffULL) return PEA_E_UNSUPPORTED;
  if (shift == 0 && !misaligned(affs, 16)) {
    if (relu) {
      const size_t n4 = n / 4, b4 = (std::max(n4, n - 4 * n4) + 255) / 256;
      hipLaunchKernelGGL(k_relu_inplace, dim3((unsigned)b4), dim3(256), 0, (hipStream_t)stream, affs, n4, n);
    }
    return hip_rc();
  }
  // source slices [shift, 2*shift) are never themselves rewritten (relu is idempotent), so in place is race-free
  if (shift > 0 && !relu) {  // nothing but the border slices changes
    const size_t nb = (size_t)B * ((size_t)shift * Y * X + (K > 1 ? (size_t)Z * shift * X : 0) + (K > 2 ? (size_t)Z * Y * shift : 0));
    hipLaunchKernelGGL(k_fill_border_only, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, (hipStream_t)stream, affs, B, K, Z, Y, X, shift, 0);
    return hip_rc();
  }
  if (shift > 0 && X % 4 == 0 && !misaligned(affs, 16)) {
    const size_t n4 = n / 4;
    hipLaunchKernelGGL(k_fill_border_relu_v4, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, affs, K, Z, Y, X, shift, n4);
    return hip_rc();
  }
  hipLaunchKernelGGL(k_fill_border_relu, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, affs, B, K, Z, Y, X, shift, relu);
  return hip_rc();
}

int pea_stitch_add(float* out_affs, float* weight_map, const float* affs_vol, const float* weight_vol, int C, int Z, int Y,
                   int X, int oz, int oy, int ox, int z0, int y0, int x0, void* stream) {
  if (!out_affs || !weight_map || !affs_vol || !weight_vol) return PEA_E_NULL;
  if (C < 1 || oz < 1 || oy < 1 || ox < 1 || z0 < 0 || y0 < 0 || x0 < 0 || z0 + oz > Z || y0 + oy > Y || x0 + ox > X) return PEA_E_DESC;
  if (misaligned(out_affs, 4) || misaligned(weight_map, 4) || misaligned(affs_vol, 4) || misaligned(weight_vol, 4)) return PEA_E_ALIGN;
  const size_t n = (size_t)oz * oy * ox, blocks = (n + 255) / 256;
  if (blocks > 0x7fffffffULL) return PEA_E_UNSUPPORTED;
  hipLaunchKernelGGL(k_stitch_add, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, out_affs, weight_map, affs_vol,
                     weight_vol, C, Z, Y, X, oz, oy, ox, z0, y0, x0);
  return hip_rc();
}

int pea_stitch_finalize(float* out_affs, const float* weight_map, int C, size_t voxels, void* stream) {
  if (!out_affs || !weight_map) return PEA_E_NULL;
  if (C < 1) return PEA_E_DESC;
  if (misaligned(out_affs, 4) || misaligned(weight_map, 4)) return PEA_E_ALIGN;
  const size_t blocks = (voxels + 255) / 256;
  if (blocks > 0x7fffffffULL) return PEA_E_UNSUPPORTED;
  if (voxels) hipLaunchKernelGGL(k_stitch_finalize, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, out_affs, weight_map, C, voxels);
  return hip_rc();
}

int pea_scale_inplace_multi(void* const* bufs, const size_t* counts, int nbuf, int dtype, const float* scale, void* stream) {
  if (!bufs || !counts || !scale) return PEA_E_NULL;
  if (nbuf < 1 || nbuf > 8 || (dtype != PEA_F32 && dtype != PEA_F16)) return PEA_E_DESC;
  ScaleMulti M = {};
  size_t nmax = 0;
  for (int i = 0; i < nbuf; ++i) {
    if (!bufs[i]) return PEA_E_NULL;
    if (misaligned(bufs[i], dtype == PEA_F32 ? 4 : 2)) return PEA_E_ALIGN;
    M.buf[i] = bufs[i];
    M.n[i] = counts[i];
    nmax = std::max(nmax, counts[i]);
  }
  if (misaligned(scale, 4)) return PEA_E_ALIGN;
  if (nmax == 0) return PEA_OK;
  const unsigned gx = (unsigned)std::min<size_t>((nmax / 4 + 255) / 256 + 1, 512);  // grid-stride: a fixed, small grid
  if (dtype == PEA_F32) hipLaunchKernelGGL(k_scale_multi<float>, dim3(gx, (unsigned)nbuf), dim3(256), 0, (hipStream_t)stream, M, scale);
  else hipLaunchKernelGGL(k_scale_multi<__half>, dim3(gx, (unsigned)nbuf), dim3(256), 0, (hipStream_t)stream, M, scale);
  return hip_rc();
}

// out[0] = sum_j w[j] * rows[j * stride]: the total of a loss section's weighted losses (loss row j = loss_out of call j), one wave.
// Fixed order of additions (j = 0, 1, ..): bit-reproducible.
namespace { __global__ void k_weighted_sum(const float* __restrict__ rows, int stride, const float* __restrict__ w, int n, float* __restrict__ out) {
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int j = 0; j < n; ++j) t += rows[(size_t)j * stride] * w[j];
    out[0] = t;
  }
} }
int pea_weighted_sum(const float* rows, int stride, const float* w, int n, float* out, void* stream) {
  if (!rows || !w || !out) return PEA_E_NULL;
  if (n < 1 || n > 64 || stride < 1) return PEA_E_DESC;
  if (misaligned(rows, 4) || misaligned(w, 4) || misaligned(out, 4)) return PEA_E_ALIGN;
  hipLaunchKernelGGL(k_weighted_sum, dim3(1), dim3(64), 0, (hipStream_t)stream, rows, stride, w, n, out);
  return hip_rc();
}

}  // extern "C"
