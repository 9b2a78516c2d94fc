// Shared device/host helpers for libnis_hip (gfx950 only: 64-lane wavefronts, fp64 VALU, LDS).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/nis_hip.h"

#define NIS_TPB 256          // threads per workgroup for the streaming kernels (4 waves)
#define NIS_WAVE 64
#define NIS_MAXB 512         // max workgroups of a statistics/apply launch (= max partial records)
#define NIS_BN_EPS 1e-5      // torch.nn.BatchNorm1d default eps
#define NIS_BN_MOM 0.1       // torch.nn.BatchNorm1d default momentum

// Uniform, read-only data (weights, coefficients written by an EARLIER launch) is read through the
// constant address space so that hipcc emits s_load_* into SGPRs and v_fma_f64 takes the weight as
// its scalar operand: no VGPRs, no LDS traffic, one scalar-cache line per 8 weights.
typedef const double __attribute__((address_space(4)))* nis_cptr;
__device__ __forceinline__ nis_cptr as_const(const double* p) { return (nis_cptr)(unsigned long long)p; }

// (count, mean, M2) record, merged with Chan et al.'s pairwise update: robust against the
// cancellation a plain sum-of-squares suffers when |mean| >> sigma.
struct Mom {
  double n, mean, m2;
};
__device__ __forceinline__ Mom mom_merge(const Mom a, const Mom b) {
  if (b.n == 0.0) return a;
  if (a.n == 0.0) return b;
  Mom r;
  r.n = a.n + b.n;
  const double delta = b.mean - a.mean;
  const double fb = b.n / r.n;
  r.mean = a.mean + delta * fb;
  r.m2 = a.m2 + b.m2 + delta * delta * (a.n * fb);
  return r;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ double wave_first(double v) { return __shfl(v, 0, 64); }

static inline int nis_pick_wm(int m) { return m <= 4 ? 4 : m <= 8 ? 8 : m <= 16 ? 16 : m <= 32 ? 32 : m <= 64 ? 64 : -1; }
static inline int nis_pick_km(int k) { return k <= 4 ? 4 : k <= 8 ? 8 : k <= 16 ? 16 : k <= 32 ? 32 : -1; }

// Per-launch view of one coupling cell (passed by value in the kernarg segment: every field is
// wave-uniform, so plane indices / sizes live in SGPRs).
struct CellK {
  int kind, d, P, T, K, L, out, R;       // out = conditioner outputs per transformed column, R = T*out
  int width[NIS_MAX_HIDDEN];
  int in_plane[NIS_MAX_DIM];             // local column j -> state plane
  int jplane;                            // plane of the running Jacobian
  int first;                             // 1: J_in == 1 (AddJacobian), the J plane is not read
  int nparts;                            // partial records written by the producing statistics launch
  int maxb;                              // stride (in records) between layers in `partials`
  int bins_row;                          // first row of this cell in bins_out
  const double* wp;                      // padded hidden weights: L x WM x WM      (packed by k_pack)
  const double* wlast;                   // padded last layer:    R x WM
  const double* bias;                    // R
  const double* gb;                      // (L+1) x 2 x WM  gamma | beta, zero padded
  double* coef;                          // (L+1) x 2 x WM  scale | shift  (finalised BN affine form)
  double* partials;                      // (L+1) x maxb x WM x 3
  double* running;                       // reference-order running-stat blob (updated in train mode)
  int* status;                           // device error word (bit0: PWLin bin index out of range)
};

__host__ __device__ static inline int cellk_feat(const CellK& c, int l) { return l == 0 ? c.P : c.width[l - 1]; }
// offset (in doubles) of BN layer l inside the reference-order running-stat blob
__host__ __device__ static inline int cellk_running_off(const CellK& c, int l) {
  int o = 0;
  for (int i = 0; i < l; ++i) o += 2 * cellk_feat(c, i);
  return o;
}

// ---- optional per-launch HIP-event timing of one kernel class (bench.py's live roofline leg)
enum { NIS_K_APPLY = 0, NIS_K_STATS = 1, NIS_K_INGEST = 2, NIS_K_INTEGRAND = 3, NIS_K_RAMBO = 4, NIS_K_EMIT = 5, NIS_K_PACK = 6,
       NIS_K_MOMENTS = 7 };
void nis_prof_mark(int kernel_class, int begin, hipStream_t s);
struct NisProfScope {
  int cls;
  hipStream_t s;
  NisProfScope(int c, hipStream_t st) : cls(c), s(st) { nis_prof_mark(cls, 1, s); }
  ~NisProfScope() { nis_prof_mark(cls, 0, s); }
};
