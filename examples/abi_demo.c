/* abi_demo.c -- the C ABI of include/pea.h from plain C: no Python, no torch, only the HIP runtime for memory.
 *
 * Builds the same call a binding in any host language would make: fill a PeaDesc, hand over
 * device pointers and a stream, read results back.  Checks pea_affinity_infer and the loss of pea_affinity_fwd against a
 * scalar double-precision restatement of scripts_cvppp/loss/loss_embedding_mse.py:18-47 (normalize, circular shift,
 * dot product, WeightedMSE with the B*W normaliser) written out below.
 *
 *   gcc -O2 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude examples/abi_demo.c \
 *       -Lpixel-embedded-affinity_amd/csrc -lpea_hip -L/opt/rocm/lib -lamdhip64 -lm -o abi_demo
 *   LD_LIBRARY_PATH=pixel-embedded-affinity_amd/csrc:/opt/rocm/lib ./abi_demo
 */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "pea.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define CHECK_PEA(x) do { int r_ = (x); if (r_ != 0) { fprintf(stderr, "%s: rc=%d (%s)\n", #x, r_, r_ < 0 ? pea_strerror(r_) : "hip error"); return 3; } } while (0)

enum { B = 2, D = 16, H = 48, W = 64, K = 4 };
static const int OFF[K][2] = {{-1, 0}, {0, -1}, {-3, 0}, {0, -5}};

static unsigned lcg(unsigned *s) { *s = *s * 1664525u + 1013904223u; return *s >> 8; }
static float unif(unsigned *s) { return (float)lcg(s) / 16777216.0f - 0.5f; }

int main(void) {
  const size_t S = (size_t)H * W, ne = (size_t)B * D * S, nk = (size_t)B * K * S;
  float *e = malloc(ne * 4), *t = malloc(nk * 4), *w = malloc(nk * 4), *affs = malloc(nk * 4), loss[1 + K];
  unsigned char *m = malloc(nk);
  unsigned seed = 555;
  for (size_t i = 0; i < ne; ++i) e[i] = unif(&seed);
  for (size_t i = 0; i < nk; ++i) { t[i] = (lcg(&seed) & 1) ? 1.f : 0.f; w[i] = 0.5f + (float)(lcg(&seed) & 255) / 256.f; m[i] = (lcg(&seed) & 7) != 0; }

  /* ---- reference: the reference's op sequence, scalar, double precision */
  double *ref = malloc(nk * sizeof(double)), ref_loss = 0.0;
  for (int b = 0; b < B; ++b)
    for (int k = 0; k < K; ++k)
      for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
          const int yy = ((y + OFF[k][0]) % H + H) % H, xx = ((x + OFF[k][1]) % W + W) % W; /* torch.roll(e, -o): rolled(p) = e(p + o) */
          double dot = 0, na = 0, nb = 0;
          for (int c = 0; c < D; ++c) {
            const double a = e[((size_t)b * D + c) * S + (size_t)y * W + x], q = e[((size_t)b * D + c) * S + (size_t)yy * W + xx];
            dot += a * q; na += a * a; nb += q * q;
          }
          const double v = dot / (fmax(sqrt(na), 1e-12) * fmax(sqrt(nb), 1e-12));
          const size_t o = ((size_t)b * K + k) * S + (size_t)y * W + x;
          ref[o] = v;
          const double r = v * m[o] - (double)t[o] * m[o];
          ref_loss += (double)w[o] * r * r / ((double)B * W); /* WeightedMSE: norm_term = B * W (loss.py:106-124) */
        }

  /* ---- the library */
  PeaDesc d;
  memset(&d, 0, sizeof(d));
  d.abi = PEA_ABI_VERSION; d.ndim = 2; d.B = B; d.D = D; d.K = K;
  d.dims[0] = 1; d.dims[1] = H; d.dims[2] = W;
  d.border = PEA_BORDER_CIRCULAR; d.dtype = PEA_F32; d.norm = PEA_NORM_BX; d.eps = 1e-12f;
  for (int k = 0; k < K; ++k) { d.offsets[k][0] = 0; d.offsets[k][1] = OFF[k][0]; d.offsets[k][2] = OFF[k][1]; d.lambda[k] = 1.f; }
  CHECK_PEA(pea_desc_validate(&d));

  void *de, *dt, *dw, *dm, *da, *dg, *dl, *dwk, *dgrad;
  const size_t wsb = pea_workspace_bytes(&d);
  hipStream_t st;
  CHECK_HIP(hipStreamCreate(&st));
  CHECK_HIP(hipMalloc(&de, ne * 4)); CHECK_HIP(hipMalloc(&dt, nk * 4)); CHECK_HIP(hipMalloc(&dw, nk * 4));
  CHECK_HIP(hipMalloc(&dm, nk)); CHECK_HIP(hipMalloc(&da, nk * 4)); CHECK_HIP(hipMalloc(&dg, nk * 4));
  CHECK_HIP(hipMalloc(&dl, sizeof(loss))); CHECK_HIP(hipMalloc(&dwk, wsb ? wsb : 4)); CHECK_HIP(hipMalloc(&dgrad, ne * 4));
  CHECK_HIP(hipMemcpyAsync(de, e, ne * 4, hipMemcpyHostToDevice, st));
  CHECK_HIP(hipMemcpyAsync(dt, t, nk * 4, hipMemcpyHostToDevice, st));
  CHECK_HIP(hipMemcpyAsync(dw, w, nk * 4, hipMemcpyHostToDevice, st));
  CHECK_HIP(hipMemcpyAsync(dm, m, nk, hipMemcpyHostToDevice, st));

  CHECK_PEA(pea_affinity_infer(&d, de, NULL, (float *)da, st));
  CHECK_HIP(hipMemcpyAsync(affs, da, nk * 4, hipMemcpyDeviceToHost, st));
  CHECK_HIP(hipStreamSynchronize(st));
  double err_inf = 0;
  for (size_t i = 0; i < nk; ++i) err_inf = fmax(err_inf, fabs(affs[i] - ref[i]));

  CHECK_PEA(pea_workspace_init(dwk, wsb, st)); /* once per workspace: the loss-state block (include/pea.h) */
  CHECK_PEA(pea_affinity_fwd(&d, de, NULL, (const float *)dt, (const float *)dw, (const uint8_t *)dm, (float *)da, (float *)dg, (float *)dl, dwk, wsb, st));
  CHECK_PEA(pea_affinity_bwd(&d, de, NULL, (const float *)dg, NULL, dgrad, NULL, st));
  CHECK_HIP(hipMemcpyAsync(loss, dl, sizeof(loss), hipMemcpyDeviceToHost, st));
  CHECK_HIP(hipMemcpyAsync(affs, da, nk * 4, hipMemcpyDeviceToHost, st));
  float *grad = malloc(ne * 4);
  CHECK_HIP(hipMemcpyAsync(grad, dgrad, ne * 4, hipMemcpyDeviceToHost, st));
  CHECK_HIP(hipStreamSynchronize(st));
  double err_fwd = 0, gsum = 0;
  for (size_t i = 0; i < nk; ++i) err_fwd = fmax(err_fwd, fabs(affs[i] - ref[i]));
  int finite = 1;
  for (size_t i = 0; i < ne; ++i) { finite &= isfinite(grad[i]) != 0; gsum += fabs(grad[i]); }
  const double rel_loss = fabs(loss[0] - ref_loss) / ref_loss;
  printf("abi_demo: version %d  |affs err| infer %.2e  fwd %.2e  loss %.6f (reference %.6f, rel %.1e)  sum|de| %.4f\n",
         pea_version(), err_inf, err_fwd, loss[0], ref_loss, rel_loss, gsum);
  const int ok = err_inf < 1e-5 && err_fwd < 1e-5 && rel_loss < 1e-5 && finite && gsum > 0;
  puts(ok ? "abi_demo: OK" : "abi_demo: FAILED");
  return ok ? 0 : 1;
}
