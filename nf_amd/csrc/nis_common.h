// Shared device/host helpers for libnis_hip (gfx950 only: 64-lane wavefronts, fp64 VALU, LDS).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/nis_hip.h"

#define NIS_TPB 256          // threads per workgroup of the apply / streaming kernels (4 waves)
#define NIS_WAVE 64
#define NIS_MAXB 2048        // max workgroups of any launch that emits partial records
#define NIS_BN_EPS 1e-5      // torch.nn.BatchNorm1d default eps
#define NIS_BN_MOM 0.1       // torch.nn.BatchNorm1d default momentum
#define NIS_PMAX_FUSED 8     // max pass-through size whose BN_0 statistics ride along in ingest / apply

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
// Sum over the 16 lanes of a DPP row (lanes with equal lane >> 4), the result in every lane of the row: four data-parallel-primitive
// moves on the vector pipe -- lane ^ 1, lane ^ 2 inside a quad, then the quads of an 8-lane half mirrored, then the halves of the row --
// instead of four ds_bpermute round trips through the LDS crossbar (__shfl_xor)
template <int CTRL>
__device__ __forceinline__ double nis_dpp_mov(double v) {
  const int lo = __double2loint(v), hi = __double2hiint(v);
  return __hiloint2double(__builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true), __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ double row16_sum(double v) {
  v += nis_dpp_mov<0xB1>(v);       // quad_perm [1, 0, 3, 2]
  v += nis_dpp_mov<0x4E>(v);       // quad_perm [2, 3, 0, 1]
  v += nis_dpp_mov<0x141>(v);      // row_half_mirror
  v += nis_dpp_mov<0x140>(v);      // row_mirror
  return v;
}

// ---- float64 exp for the bijector: exp(x) = 2^m * T[j] * (1 + p(r)),  k = rint(x*64/ln2) = 64 m + j,
// r = x - k ln2/64 (two-step Cody-Waite), p = degree-5 Taylor (|r| < 5.5e-3 -> truncation 4e-17).
// ~16 VALU + one LDS read instead of ~45 VALU for ocml's exp; max error 1 ulp (measured 2.2e-16 relative).
// Arguments are clamped to +-708 (no inf / denormal paths: conditioner outputs are O(1..10)).
static __device__ const double NIS_EXP_TAB[64] = {
    0x1.0000000000000p+0, 0x1.02c9a3e778061p+0, 0x1.059b0d3158574p+0, 0x1.0874518759bc8p+0,
    0x1.0b5586cf9890fp+0, 0x1.0e3ec32d3d1a2p+0, 0x1.11301d0125b51p+0, 0x1.1429aaea92de0p+0,
    0x1.172b83c7d517bp+0, 0x1.1a35beb6fcb75p+0, 0x1.1d4873168b9aap+0, 0x1.2063b88628cd6p+0,
    0x1.2387a6e756238p+0, 0x1.26b4565e27cddp+0, 0x1.29e9df51fdee1p+0, 0x1.2d285a6e4030bp+0,
    0x1.306fe0a31b715p+0, 0x1.33c08b26416ffp+0, 0x1.371a7373aa9cbp+0, 0x1.3a7db34e59ff7p+0,
    0x1.3dea64c123422p+0, 0x1.4160a21f72e2ap+0, 0x1.44e086061892dp+0, 0x1.486a2b5c13cd0p+0,
    0x1.4bfdad5362a27p+0, 0x1.4f9b2769d2ca7p+0, 0x1.5342b569d4f82p+0, 0x1.56f4736b527dap+0,
    0x1.5ab07dd485429p+0, 0x1.5e76f15ad2148p+0, 0x1.6247eb03a5585p+0, 0x1.6623882552225p+0,
    0x1.6a09e667f3bcdp+0, 0x1.6dfb23c651a2fp+0, 0x1.71f75e8ec5f74p+0, 0x1.75feb564267c9p+0,
    0x1.7a11473eb0187p+0, 0x1.7e2f336cf4e62p+0, 0x1.82589994cce13p+0, 0x1.868d99b4492edp+0,
    0x1.8ace5422aa0dbp+0, 0x1.8f1ae99157736p+0, 0x1.93737b0cdc5e5p+0, 0x1.97d829fde4e50p+0,
    0x1.9c49182a3f090p+0, 0x1.a0c667b5de565p+0, 0x1.a5503b23e255dp+0, 0x1.a9e6b5579fdbfp+0,
    0x1.ae89f995ad3adp+0, 0x1.b33a2b84f15fbp+0, 0x1.b7f76f2fb5e47p+0, 0x1.bcc1e904bc1d2p+0,
    0x1.c199bdd85529cp+0, 0x1.c67f12e57d14bp+0, 0x1.cb720dcef9069p+0, 0x1.d072d4a07897cp+0,
    0x1.d5818dcfba487p+0, 0x1.da9e603db3285p+0, 0x1.dfc97337b9b5fp+0, 0x1.e502ee78b3ff6p+0,
    0x1.ea4afa2a490dap+0, 0x1.efa1bee615a27p+0, 0x1.f50765b6e4540p+0, 0x1.fa7c1819e90d8p+0};
#ifndef NIS_EXP_BITS
#define NIS_EXP_BITS 6      // flow_wm.hip selects 10 for the narrow (VALU) units: +2.5 % on the C2 step; the matrix-core kernels lose a workgroup per CU to the 8 KB table
#endif
#define NIS_EXP_N (1 << NIS_EXP_BITS)
#if NIS_EXP_BITS == 8
// 256-entry table (2 KB of LDS) + degree-4 Taylor polynomial: one FMA fewer per exp than the 64-entry / degree-5 form, and small
// enough to sit beside the matrix-core apply kernel's LDS image without costing the second workgroup per CU.
#include "nis_exp_tab10.h"
__device__ __forceinline__ void nis_exp_init(double* lds_tab) {
  for (int i = threadIdx.x; i < NIS_EXP_N; i += blockDim.x) lds_tab[i] = NIS_EXP_TAB8[i];
}
__device__ __forceinline__ double nis_exp(double x, const double* lds_tab) {
  const double kd = fma(x, 0x1.71547652b82fep+8, 0x1.8p+52);      // k = rint(256 x / ln 2) in the low word (see below)
  const double k = kd - 0x1.8p+52;
  const double r = fma(k, -0x1.62e42fefa39efp-9, x);                // ONE fused step with the correctly rounded ln 2 / 256: the product is exact
  const int ki = __double2loint(kd);                                 // inside the fma, so r is off by |x| 2^-53 at most (~1e-15 relative in exp for
  const double T = lds_tab[ki & 255];                                // |x| < 10); the two-constant Cody-Waite form is for machines without fma
  double p = fma(r, 1.0 / 24.0, 1.0 / 6.0);
  p = fma(r, p, 0.5);
  p = fma(r, p, 1.0);
  p = r * p;
  return ldexp(fma(T, p, T), ki >> 8);
}
#elif NIS_EXP_BITS == 10
#include "nis_exp_tab10.h"
__device__ __forceinline__ void nis_exp_init(double* lds_tab) {
  for (int i = threadIdx.x; i < NIS_EXP_N; i += blockDim.x) lds_tab[i] = NIS_EXP_TAB10[i];
}
__device__ __forceinline__ double nis_exp(double x, const double* lds_tab) {
  const double kd = fma(x, 0x1.71547652b82fep+10, 0x1.8p+52);     // k = rint(1024 x / ln 2) in the low word (see below)
  const double k = kd - 0x1.8p+52;
  const double r = fma(k, -0x1.62e42fefa39efp-11, x);               // one fused step (see the 256-entry form)
  const int ki = __double2loint(kd);
  const double T = lds_tab[ki & 1023];
  double p = fma(r, 1.0 / 6.0, NIS_EXP10_C2);
  p = fma(r, p, 1.0);
  p = r * p;
  return ldexp(fma(T, p, T), ki >> 10);
}
#else
__device__ __forceinline__ void nis_exp_init(double* lds_tab) {
  if (threadIdx.x < 64) lds_tab[threadIdx.x] = NIS_EXP_TAB[threadIdx.x];
}
__device__ __forceinline__ double nis_exp(double x, const double* lds_tab) {
  // k = rint(64 x / ln 2) by the 1.5 * 2^52 trick: the add rounds to nearest-even and leaves k (two's complement) in the
  // low word of the sum -- one FMA + one subtraction instead of multiply, v_rndne and v_cvt_i32.  Valid for |k| < 2^31
  // (|x| < 2.3e7); no clamp: |x| > 709 gives |k >> 6| > 1023 and v_ldexp_f64 returns 0 / inf like exp() would.
  const double kd = fma(x, 0x1.71547652b82fep+6, 0x1.8p+52);
  const double k = kd - 0x1.8p+52;
  const double r = fma(k, -0x1.62e42fefa39efp-7, x);                // one fused step (see the 256-entry form)
  const int ki = __double2loint(kd);
  const double T = lds_tab[ki & 63];
  double p = fma(r, 1.0 / 120.0, 1.0 / 24.0);
  p = fma(r, p, 1.0 / 6.0);
  p = fma(r, p, 0.5);
  p = fma(r, p, 1.0);
  p = r * p;
  return ldexp(fma(T, p, T), ki >> 6);
}

#endif

// 1 / x and 1 / sqrt(x) for x in the normal range: v_rcp_f64 / v_rsq_f64 (~26 bits) + two Newton steps, no scaling / fix-up
// sequence (5 / 8 instructions instead of the ~14 of an IEEE division, ~26 of sqrt + division; results within 1-2 ulp).
__device__ __forceinline__ double nis_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
}
__device__ __forceinline__ double nis_rsqrt(double x) {
  double r = __builtin_amdgcn_rsq(x);
  const double h = 0.5 * x;
  r = fma(r, fma(-(h * r), r, 0.5), r);
  r = fma(r, fma(-(h * r), r, 0.5), r);
  return r;
}

// sqrt(x) for x in the normal range or exactly 0 (negative -> NaN like sqrt): v_rsq_f64 + one Goldschmidt step + the residual
// correction, <= 1 ulp.  11 instructions; the compiler's IEEE sqrt (input scaling by v_ldexp, v_cmp_class fix-ups) is ~20,
// and the 2 -> 4 RAMBO map takes 29 square roots per sample.
__device__ __forceinline__ double nis_sqrt(double x) {
  const double r = __builtin_amdgcn_rsq(x);
  double g = x * r, h = 0.5 * r;
  const double e = fma(-h, g, 0.5);
  g = fma(g, e, g);
  h = fma(h, e, h);
  g = fma(fma(-g, g, x), h, g);
  return x == 0.0 ? 0.0 : g;
}

// matrix-core statistics kernel (k_bn_stats_mfma), widths <= 32: threads per workgroup.  512 = ONE workgroup of eight waves per CU instead
// of two of four: the same waves, half the records -- every consumer merges the records of its freshest BatchNorm layer in its prologue
// (load_coefs), 24-byte-strided loads whose cost grows with records x readers per CU
#ifndef NIS_MF_STATS_TPB
#define NIS_MF_STATS_TPB 512
#endif
// ... for launches of many tiles per wave; small ones (one or two tiles per wave: the reference's default training regime) keep workgroups
// of four waves, whose prologue and epilogue are shorter (C5 topology, 5 x 2 000: 14.4 us per launch against 18.4)
static inline int nis_mf_stats_tpb(int wm, long long nsamples) { return (wm <= 32 && nsamples >= 65536) ? NIS_MF_STATS_TPB : 256; }
static inline int nis_pick_wm(int m) { return m <= 4 ? 4 : m <= 8 ? 8 : m <= 16 ? 16 : m <= 32 ? 32 : m <= 64 ? 64 : -1; }
// bin-array size of the PWQuad kernels: register arrays up to 32 bins; NIS_KM_BIG (33 .. NIS_MAX_BINS bins): the heights / widths of
// a column live in per-thread scratch arrays and every loop over the bins is a runtime loop (pwquad_column_big)
#define NIS_KM_BIG 128
static inline int nis_pick_km(int k) { return k <= 4 ? 4 : k <= 8 ? 8 : k <= 16 ? 16 : k <= 32 ? 32 : k <= NIS_KM_BIG ? NIS_KM_BIG : -1; }
// workgroup size of the statistics kernel: as many waves as the register budget of width WM allows
#ifndef NIS_STATS_TPB_NARROW
#define NIS_STATS_TPB_NARROW 1024     // statistics kernels of the narrow conditioners: ONE workgroup per CU (the prologue is paid per workgroup) of 16 waves
#endif
__host__ __device__ constexpr int nis_stats_tpb(int wm) { return wm <= 8 ? NIS_STATS_TPB_NARROW : wm <= 16 ? 512 : 256; }

// Built-in integrand evaluated inside the last apply kernel (f(X)*J never leaves the chip).
struct FuseF {
  long fres_gs;           // elements between per-sample outputs of consecutive batches
  int maxb;               // record stride between batches
  int kind;               // -1: none, else NIS_F_*
  int loc0, loc1;         // local columns holding logical output columns 0 and 1 (camel)
  int write_planes;       // 0: the transformed columns / Jacobian are not stored (nobody reads them)
  double p[4];
  double* partials;       // nblocks x 4 moment records
  double* fres;           // optional per-sample f*J
};

// Relative image of a PWQuad last Linear (matrix-core apply kernel).  The bijector is invariant under a common factor of a column's
// K + 1 heights and under a common factor of its K widths (coupling_cells.py:176-225: the widths are normalised by their sum, the
// heights by the trapezoid integral), so exp(z_k - z_ref) serves as well as exp(z_k): the packer subtracts the row of V[K] from the other
// height rows and the row of W[K-1] from the other width rows, the two reference rows become the constant 1 and are not computed.
// A column then has 2K - 1 rows; the image gives it 2K (one zero row at the end), so that a batch of columns is a whole number of
// 16-row tiles for every K the image exists for (4, 8, 16, 32) and no tile is computed for two batches: C4's cells with T = 5 take
// 10 tile products per 16 samples instead of 9 + 3 (reference rows: 33 per column, batches of 4 columns = 132 rows), C5's 16 instead of 9 + 9.
#ifndef NIS_MF_REL
#define NIS_MF_REL 1
#endif
#define NIS_PACK_REL 3
__host__ __device__ inline int nis_rel_out(int K) { return 2 * K; }          // rows per column in the relative image
// packed row j -> row of the reference-order last Linear (-1: zero row); *js = the row subtracted from it
__host__ __device__ inline int nis_rel_row(int T, int K, int j, int* js) {
  const int o = nis_rel_out(K), t = j / o, q = j % o, b = t * (2 * K + 1);
  *js = -1;
  if (t >= T || q == o - 1) return -1;
  *js = q < K ? b + K : b + 2 * K;
  return q < K ? b + q : b + q + 1;
}

// Per-launch view of one coupling cell (passed by value in the kernarg segment: every field is
// wave-uniform, so plane indices / sizes live in SGPRs).
struct CellK {
  int kind, d, P, T, K, L, out, R;       // out = conditioner outputs per transformed column, R = T*out
  int mf_ncb;                            // matrix-core apply kernel: transformed columns per bijector batch (4, 2 or 1; set at launch)
  int regmode;                           // row order of the last Linear's matrix-core image: 0 reference order or NIS_PACK_REL
  int width[NIS_MAX_HIDDEN];
  int in_plane[NIS_MAX_DIM];             // local column j -> state plane
  int jplane;                            // plane of the running Jacobian
  int first;                             // 1: J_in == 1 (AddJacobian), the J plane is not read
  int nparts[NIS_MAX_HIDDEN + 1];        // records written for BN layer l by its producing launch
  int maxb;                              // stride (in records) between layers in `partials`
  int bins_row;                          // first row of this cell in bins_out
  int next_P;                            // >0: also emit BN_0 records of the NEXT cell ...
  int next_src[NIS_PMAX_FUSED];          // ... whose pass-through feature f is this cell's local column next_src[f]
  const double* wp;                      // padded hidden weights: L x WM x WM      (packed by k_pack)
  const double* wlast;                   // padded last layer:    R x WM
  const double* bias;                    // R
  const double* gb;                      // (L+1) x 2 x WM  gamma | beta, zero padded
  const double* hb;                      // (L+1) x WM  bias of the Linear feeding BN layer l (NIS_AFFINE; else zeros / null)
  double* coef;                          // (L+1) x 2 x WM  scale | shift  (finalised BN affine form)
  double* partials;                      // (L+1) x maxb x WM x 3
  double* next_partials;                 // BN_0 records of the next cell (maxb x WMnext x 3), stride next_wm
  int next_wm;
  double* running;                       // reference-order running-stat blob (updated in train mode)
  long long* tracked;                    // num_batches_tracked of the L+1 BN layers (incremented in train mode)
  int* status;                           // device error word (bit0: PWLin bin index out of range)
  // ---- batched iterations: blockIdx.y = g selects one of G independent batches (own planes, own BN statistics)
  int G;                                 // number of batches in this launch set (stride of the per-batch arrays)
  int Gy, g0;                            // this launch covers batches [g0, g0 + Gy): blockIdx.y counts from g0 (several chains of a
                                         //   set run on their own streams, flow_run)
  int coef_gs;                           // doubles between the coefficient sets of consecutive g: (L+1)*2*WM
  long plane_gs;                         // doubles between the plane sets of consecutive g: (d+1)*ld
  const double* sync_part;               // non-null (multi-rank exact BatchNorm): the freshest layer's records are the `sync_world` per-rank
  int sync_world;                        //   records gathered here (one record per rank) instead of this launch set's workgroup records
  int pmerge;                            // 1 (always since round 4): consumers merge the records of the freshest BN layer in their prologue
                                         //    (every workgroup of batch g redoes the small tree merge: cheaper than a finalize launch)
  double* gstat;                         // G > 1: (L+1) x G x 2 x WM  batch mean | unbiased var for k_group_epilogue
  // ---- direct row input (d = 2 chains on the SPEC kernels, float32 w): a column that no cell has written yet is read
  //      straight from the (N, 2) float32 rows of w -- the ingest launch then only gathers BN_0's records
  const float* wrows;                    // null: every column lives in its plane
  long wrows_gs;                         // floats between the row blocks of consecutive g: 2 N
  int x_wcol, xb_wcol;                   // column of w that holds the pass-through / the transformed input, -1: read the plane
  int wplanar;                           // 1 (in-kernel draws): w is two float32 column planes (column c at wrows + c * wplane_ld), not rows:
  long wplane_ld;                        //   a pass reads only the column it needs (4 instead of 8 B/sample)
  // ---- training history (for the adjoint kernels): outputs go to the NEXT plane set, BN statistics are kept
  int derive1;                           // 1: BN_1's batch statistics follow analytically from BN_0's (P == 1, or dcov)
  int dcov;                              // 1 (matrix-core cells, 2 <= P <= 8): BN_0's records carry the P x P co-moments of the pass-through block
                                         //   (k_bn_stats0_cov) and BN_1 is derived from them: h1 = W0 bn0(x) is affine in x
                                         //   (h1_j = W0_j0 * BN_0(x) is affine in the single pass-through column),
                                         //   so the layer-1 statistics launch is skipped
  int fold;                              // 1 (P == 1, fp64 VALU kernels): BN_0, Linear_0, BN_1 evaluated as ONE affine map of x (fold_coefs)
  int wx;                                // exact common width of the hidden layers when a specialised kernel exists (WM = 4: 3), else 0
  long out_off;                          // doubles from a cell's input planes to its output planes (0: in place)
  double* bnsave;                        // (L+1) x 4 x WM: scale | shift | batch mean | 1/sqrt(var+eps)  (may be null)
  long bnsave_gs;                        // doubles between the saved statistics of consecutive batches g (the whole model's record)
};

__host__ __device__ static inline int cellk_feat(const CellK& c, int l) { return l == 0 ? c.P : c.width[l - 1]; }
// BN layer whose records the prologue of a launch that needs the coefficients of layers < upto has to merge (G == 1)
// co-moment slot of the pair (a < b) behind the P feature records of a BN_0 record with covariance (CellK::dcov)
__host__ __device__ static inline int cellk_pair(int P, int a, int b) { return a * P - a * (a + 1) / 2 + (b - a - 1); }
__host__ __device__ static inline int cellk_fresh(const CellK& c, int upto) { return (c.derive1 && upto == 2) ? 0 : upto - 1; }
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

// ---- block-level accumulation of (count, mean, M2[, max]) with a wave-uniform shift --------------------
// Every lane adds t = v - shift, t^2 (shift = the wave's first value, so |t| ~ sigma and the final
// S2 - S1^2/n loses nothing); waves are then merged with Chan's update.
template <int NW>
__device__ __forceinline__ void block_record(double shift, double S1, double S2, double cnt, double vmax, double (*wm)[4], double* part) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const double nw = wave_sum(cnt), s1 = wave_sum(S1), s2 = wave_sum(S2), mx = wave_max(vmax);
  if (lane == 0) {
    wm[wave][0] = nw;
    wm[wave][1] = nw > 0.0 ? shift + s1 / nw : 0.0;
    wm[wave][2] = nw > 0.0 ? fmax(s2 - s1 * s1 / nw, 0.0) : 0.0;
    wm[wave][3] = mx;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    Mom t = {0.0, 0.0, 0.0};
    double m = -__builtin_huge_val();
    for (int w = 0; w < NW; ++w) {
      Mom q = {wm[w][0], wm[w][1], wm[w][2]};
      t = mom_merge(t, q);
      m = fmax(m, wm[w][3]);
    }
    part[0] = t.n;
    part[1] = t.mean;
    part[2] = t.m2;
    part[3] = m;
  }
}
