// "RAMBO on diet" (Plaetzer 2013) batched 2 -> n phase-space map: one thread per sample, every
// 4-vector in registers (n is a template parameter so all leg indices are compile-time).
//
// Reference restated: PhaseSpace/flat_phase_space_generator.py:139-308 (generateKinematics_batch),
// :313-359 (bisect_vec_batch), :363-406 (intermediate masses + massive reweighting), :408-441
// (initial state), PhaseSpace/utils.py:5-19 (set_square), :58-106 (boosts), :124-132 (uniform_distr),
// :134-146 (lab boost), :151-187 (pseudo-rapidity, delta-phi, delta-R).  pdf == None (density 1).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "nis_common.h"

struct RamboK {
  nis_rambo_desc d;
  double flat_ab;      // (2pi)^(4-3n) (pi/2)^(n-1)
  double flat_fact;    // (n-1)! (n-2)!
  double flat_const;   // full flat weight for a scalar E_cm (pdf_active == 0)
  double msum[NIS_MAX_LEGS];   // msum[i] = sum_{j>=i} m_j
  double mtot;
  double tau_min;
  nis_pdf1d pdf[2];            // per beam: x f(x) at the fixed scale on the grid's x knots (nx == 0: density 1)
  double cospi_c[11];          // series coefficients of rb_cospi: kernel arguments = scalar operands (as literals they sit in 22 VGPRs)
  int cols[3 * NIS_MAX_LEGS];  // column -> plane (column-major input) or identity
  long r_ld;
  int col_major;
  int ncol;
  // ---- batched plane sets + moment rider (nis_rambo_moments: the integrand side of a fused flow -> RAMBO weight -> moments step).
  //      blockIdx.y = batch g: inputs at r + g r_gs, weights (optional) at weight + g w_gs; mom_part != null: the weight times the
  //      flow's Jacobian jac[g r_gs + i] is reduced to one (count, mean, M2, max) record per workgroup, mom_part[(g maxb + block) 4]
  long r_gs, w_gs;
  const double* jac;
  double* mom_part;
  int mom_maxb;
};

#define RB_EPS 1.4901161193847656e-08     /* np.finfo(float).eps ** 0.5   (utils.py:151) */
#define RB_HUGE 1.7976931348623157e308    /* np.finfo(float).max */

// rho(M, N, m) = rb_rho_num(M, N, m) / (8 M^2)                                   // :107-113
__device__ __forceinline__ double rb_rho_num(double M, double N, double m) {
  const double Msqr = M * M;
  return nis_sqrt((Msqr - (N + m) * (N + m)) * (Msqr - (N - m) * (N - m)));
}
__device__ __forceinline__ double rb_rho(double M, double N, double m) { return rb_rho_num(M, N, m) * nis_rcp(8. * (M * M)); }

__device__ __forceinline__ double rb_ipow(double x, int e) {
  double p = 1.0;
  for (int k = 0; k < e; ++k) p *= x;
  return p;
}

// Root of g(u) = (e+1) u^e - e u^(e+1) = v on (0,1) (the equation bisect_vec_batch solves, :313-359) by Newton's
// method started on the side from which it converges monotonically: g is increasing, convex left of its inflection
// u_i = (e-1)/e and concave right of it, so start right of the root in the convex part (u_s/(1-u_s) >= root, u_s the
// leading-order root) and left of it in the concave part.  8 steps reach the fixed point of the double evaluation
// (<= few ulp from the lattice bisection's limit; identical conditioning: both stall where g' ~ 0 at v -> 1).
__device__ __forceinline__ double rb_solve(double v, int e) {
  if (e == 1) return v * nis_rcp(1.0 + nis_sqrt(1.0 - v));            // 2u - u^2 = v
  const double ui = (double)(e - 1) / (double)e;
  const double vi = rb_ipow(ui, e) * ((double)(e + 1) - (double)e * ui);
  const bool convex = v < vi;
  double u;
  if (convex) {
    const double vr = v * (1.0 / (double)(e + 1));      // starting guesses: reciprocals instead of IEEE divisions (the fixed point does not move)
    const double us = e == 2 ? nis_sqrt(vr) : e == 3 ? cbrt(vr) : e == 4 ? nis_sqrt(nis_sqrt(vr)) : pow(vr, 1.0 / (double)e);
    double c0 = us * nis_rcp(1.0 - fmin(us, 0.9));
    c0 = fmin(c0, ui);
    const double g0 = rb_ipow(c0, e) * ((double)(e + 1) - (double)e * c0);
    u = (g0 >= v) ? c0 : ui;
  } else {
    const double ub = 1.0 - nis_sqrt((1.0 - v) * (2.0 / (double)(e * (e + 1))));
    u = fmax(ui, ub);
  }
#pragma unroll 1
  for (int it = 0; it < 8; ++it) {
    const double um1 = rb_ipow(u, e - 1);
    const double g = um1 * u * ((double)(e + 1) - (double)e * u) - v;
    const double gp = (double)(e * (e + 1)) * um1 * (1.0 - u);
    double un = u - g * nis_rcp(gp);                    // gp == 0 (NaN here) is discarded two lines below
    un = convex ? fmin(fmax(un, 0.0), ui) : fmin(fmax(un, ui), 1.0);
    un = (gp > 0.0) ? un : u;
    const bool moved = fabs(un - u) > 4e-16 * un;       // the iteration is monotone: no movement = fixed point
    u = un;
    if (!__any(moved)) break;                           // wave-uniform exit (typically after 4-5 steps)
  }
  return u;
}

struct V4 {
  double e, x, y, z;
};

// cos(pi x) for the azimuth (x = 2 r, r uniform in [0, 1)): k = rint(x), g = x - k in [-1/2, 1/2] (exact), and with
// h = 1/2 - |g| (exact) cos(pi g) = sin(pi h): ONE odd series on [0, 1/2] that keeps full relative accuracy at the zeros of the
// cosine (h -> 0) -- ~20 instructions against ~50 of the library's cospi (quadrant reduction, both polynomials, select).
// Coefficients (-1)^n pi^(2n+1) / (2n+1)!, n = 0..10, rounded from 80-digit values; truncation 1.3e-18, measured <= 3.4e-16 relative.
static const double RB_COSPI_C[11] = {0x1.921fb54442d18p+1,  -0x1.4abbce625be53p+2, 0x1.466bc6775aae2p+1,  -0x1.32d2cce62bd86p-1,
                                      0x1.50783487ee782p-4,  -0x1.e3074fde8871fp-8, 0x1.e8f434d018d63p-12, -0x1.6fadb9f155744p-16,
                                      0x1.aaec32af93359p-21, -0x1.8a404211f9547p-26, 0x1.2877020d52cf0p-31};
template <typename KT>
__device__ __forceinline__ double rb_cospi(double x, const KT& k) {
  const double kk = rint(x);
  const double h = 0.5 - fabs(x - kk);
  const double z = h * h;
  double p = k.cospi_c[10];
#pragma unroll
  for (int n = 9; n >= 0; --n) p = fma(p, z, k.cospi_c[n]);
  const double c = h * p;
  const int odd = (int)kk << 31;                                  // cos(pi (g + k)) = (-1)^k cos(pi g)
  return __hiloint2double(__double2hiint(c) ^ odd, __double2loint(c));
}

__device__ __forceinline__ void rb_set_square(V4& p, double sq) { p.e = nis_sqrt((p.x * p.x + p.y * p.y + p.z * p.z) + sq); }   // utils.py:5-19

__device__ __forceinline__ void rb_boost(V4& p, double bx, double by, double bz) {                                          // utils.py:58-81
  const double b2 = bx * bx + by * by + bz * bz;
  const double gamma = nis_rsqrt(1.0 - b2);
  const double bp = p.x * bx + p.y * by + p.z * bz;
  const double gamma2 = b2 > 1e-280 ? (gamma - 1.0) * nis_rcp(b2) : 0.0;      // (gamma - 1 is exactly 0 long before b2 gets that small)
  const double factor = gamma2 * bp + gamma * p.e;
  p.x += factor * bx;
  p.y += factor * by;
  p.z += factor * bz;
  p.e = gamma * (p.e + bp);
}

// eta = -log(tan(theta/2)), theta = atan2(pt, pz) (utils.py:151-157).  tan(theta/2) = pt / (|p| + pz) = (|p| - pz) / pt: the
// form without cancellation is taken per sign of pz -- same value to ~1 ulp without atan2 and tan (~200 VALU instructions).
__device__ __forceinline__ double rb_pseudorap(const V4& p) {
  const double pt2 = p.x * p.x + p.y * p.y;
  const double pt = nis_sqrt(pt2);
  const double pa = nis_sqrt(pt2 + p.z * p.z);
  double th2 = p.z >= 0.0 ? pt / (pa + p.z) : (pa - p.z) / pt;
  th2 = fmin(fmax(th2, 6.123e-17), 1.633e16);               // pt == 0 exactly: the reference's tan(atan2(0, pz) / 2) stays finite too
  return (pt < RB_EPS && fabs(p.z) < RB_EPS) ? RB_HUGE : -log(th2);
}

__device__ __forceinline__ double rb_delphi(const V4& a, const V4& b) {                                                     // utils.py:170-180
  const double pt1 = nis_sqrt(a.x * a.x + a.y * a.y), pt2 = nis_sqrt(b.x * b.x + b.y * b.y);
  double tmp = a.x * b.x + a.y * b.y;
  tmp /= (pt1 * pt2);
  double r = fabs(tmp) > 1.0 ? acos(tmp / fabs(tmp)) : acos(tmp);
  if (pt1 == 0. || pt2 == 0.) r = RB_HUGE;
  return r;
}

// x f(x) by LHAPDF 6's log-cubic rule on one table (include/nis_hip.h, nis_pdf1d): binary search for the knot interval
// (uniform trip count: ceil(log2 nx)), cubic Hermite in log x with mean-of-adjacent-differences knot derivatives.
__device__ __forceinline__ double rb_pdf_xf(const nis_pdf1d& t, double x) {
  const int n = t.nx;
  double lx = log(x);
  lx = fmin(fmax(lx, t.logx[0]), t.logx[n - 1]);
  int lo = 0, hi = n - 1;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    const bool le = t.logx[mid] <= lx;
    lo = le ? mid : lo;
    hi = le ? hi : mid;
  }
  const int i = lo;
  const double l0 = t.logx[i], l1 = t.logx[i + 1], f0 = t.xf[i], f1 = t.xf[i + 1];
  const double dl = l1 - l0, s = (f1 - f0) / dl;
  double d0 = s, d1 = s;
  if (i > 0) d0 = 0.5 * (s + (f0 - t.xf[i - 1]) / (l0 - t.logx[i - 1]));
  if (i + 2 < n) d1 = 0.5 * (s + (t.xf[i + 2] - f1) / (t.logx[i + 2] - l1));
  const double u = (lx - l0) / dl, u2 = u * u, u3 = u2 * u;
  return (2 * u3 - 3 * u2 + 1) * f0 + (u3 - 2 * u2 + u) * (d0 * dl) + (-2 * u3 + 3 * u2) * f1 + (u3 - u2) * (d1 * dl);
}

__global__ __launch_bounds__(256) void k_pdf_xfx(nis_pdf1d t, const double* __restrict__ x, long N, double* __restrict__ out) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < N; i += (long)gridDim.x * 256) out[i] = rb_pdf_xf(t, x[i]);
}

#ifndef RB_TPB
#define RB_TPB 128      // threads (= samples) per tile: bounds the LDS staging area to 128*(4(n+2)+1)*8 B
#endif

// The row-major OUTPUT goes through LDS: the (RB_TPB x 4(n+2)) momenta are written to a padded LDS image and stored as
// contiguous 8-byte-per-lane streams; without staging every lane would write 16 bytes into its own 128-byte line per
// instruction (row stride 128-384 B).  Inputs are prefetched into registers one tile ahead (see `fetch`).
#ifndef RB_NT
#define RB_NT 0             // 1: non-temporal hints on the streamed rows in / momenta out (A/B)
#endif
typedef double nis_d2 __attribute__((ext_vector_type(2)));
// 16-byte global loads the compiler does not track (see k_rambo): NV of them off one row pointer, 16 bytes apart
template <int NV, int NR>
__device__ __forceinline__ void rb_issue_row(nis_d2 (&raw)[NR], const double* p) {
  static_assert(NV <= NR, "row wider than its register image");
#if RB_NT
#define RB_LOAD_MOD " nt"
#else
#define RB_LOAD_MOD ""
#endif
#define RB_ISSUE16(J) \
  if constexpr (NV > J) asm volatile("global_load_dwordx4 %0, %1, off offset:%2" RB_LOAD_MOD : "=&v"(raw[J]) : "v"(p), "n"(16 * J) : "memory");
  RB_ISSUE16(0) RB_ISSUE16(1) RB_ISSUE16(2) RB_ISSUE16(3) RB_ISSUE16(4) RB_ISSUE16(5) RB_ISSUE16(6) RB_ISSUE16(7)
  RB_ISSUE16(8) RB_ISSUE16(9) RB_ISSUE16(10) RB_ISSUE16(11) RB_ISSUE16(12) RB_ISSUE16(13) RB_ISSUE16(14) RB_ISSUE16(15)
#undef RB_ISSUE16
  static_assert(NV <= 16, "add RB_ISSUE16 lines");
}
// s_waitcnt vmcnt(CNT) that every later use of the row image depends on
template <int CNT, int NR>
__device__ __forceinline__ void rb_wait_loads(nis_d2 (&raw)[NR]) {
  asm volatile("s_waitcnt vmcnt(%0)" : : "n"(CNT) : "memory");
#pragma unroll
  for (int j = 0; j < NR; ++j) asm volatile("" : "+v"(raw[j]));
}

// MODE: compile-time copy of the feature switches, because the generic kernel (every switch a scalar branch, the union
// of all live ranges) needs 256 VGPRs = one wave per SIMD:
//   0 plain 2 -> n map | 1 + PDF sampling (tau/y or x1,x2), x cut, lab boost | 2 + pT / delta-R / eta cuts
//   3 + the reference's fixed-depth lattice bisection
template <int NF, int MODE = 3>
#ifndef RB_STORE_CHUNK
#define RB_STORE_CHUNK 4
#endif
#ifndef RB_WAVES
#define RB_WAVES 3          // waves per SIMD the plain 2 -> (<= 4) map is allocated for (its LDS staging allows three)
#endif
__global__ __launch_bounds__(RB_TPB, (MODE == 0 && NF <= 4) ? RB_WAVES : 1)
void k_rambo(RamboK k, const double* __restrict__ r, long N, double* __restrict__ momenta,
                                                  double* __restrict__ weight, int* __restrict__ status) {
  extern __shared__ double rb_lds[];
  __shared__ double rb_fmom[RB_TPB / 64][4];
  constexpr int LM = 4 * (NF + 2);                   // doubles of one sample's momenta
  const int tid = threadIdx.x;
  r += (size_t)blockIdx.y * k.r_gs;
  if (weight) weight += (size_t)blockIdx.y * k.w_gs;
  const double* const jac = k.mom_part ? k.jac + (size_t)blockIdx.y * k.r_gs : nullptr;
  double mS1 = 0.0, mS2 = 0.0, mcnt = 0.0, mmax = -__builtin_huge_val(), mshift = 0.0;
  bool mfirst = true;
  constexpr bool HAS_PDF = MODE >= 1, HAS_CUTS = MODE >= 2, GEN = MODE >= 3;
  const bool pdf_active = HAS_PDF ? (k.d.flags & NIS_RAMBO_PDF_ACTIVE) != 0 : false;
  const bool use_tau = HAS_PDF ? (k.d.flags & NIS_RAMBO_TAU) != 0 : false;
  const bool lab_boost = HAS_PDF ? (k.d.flags & NIS_RAMBO_LAB_BOOST) != 0 : false;
  const int bisect_levels = GEN ? k.d.bisect_levels : 0;
  const double pT_mincut = HAS_CUTS ? k.d.pT_mincut : -1.0, delR_mincut = HAS_CUTS ? k.d.delR_mincut : -1.0,
               rap_maxcut = HAS_CUTS ? k.d.rap_maxcut : -1.0;
  const bool col_major = k.col_major != 0;
  constexpr int NPS = 3 * NF - 4;                    // phase-space columns; +2 (tau, ycm | x2, x1) with PDFs
  const long ntiles = (N + RB_TPB - 1) / RB_TPB;
  // Inputs: every thread reads its own row (64-208 contiguous bytes; the lanes of a wave cover one contiguous block, so
  // the cache lines are fully used) or its element of each column plane, one tile AHEAD: the loads of the next tile are
  // in flight while the current one is computed.  PDF columns only exist in the PDF modes.
  constexpr int NIN = HAS_PDF ? NPS + 2 : NPS;
  // row-major rows of exactly NIN doubles (NIN even): 16-byte vector loads, NIN / 2 per sample instead of NIN 8-byte ones
  // ... for maps of at most six final-state legs ONLY: the pipelined loads land in registers the compiler believes idle, which is sound
  // exactly as long as it never spills or re-allocates them between issue and wait.  The larger instantiations do spill (n >= 7 with
  // cuts, n = 10 in every mode: 66 .. 264 spilled registers) and then read stale inputs -- found by the random 2 -> n sweep of round 3
  // (2 -> 10 with cuts: 1-4 % of the weights differed from run to run).  tests/test_host_logic.py checks the built object: no spills
  // in any instantiation that takes this path.
  constexpr bool PIPE_OK = NF <= 6;
  // ... and for calls that return the momenta ONLY: the hand-written wait below counts the momenta stores of a whole tile, so every path
  // through a pipelined tile must issue them (tests/isa_vmem_check.py walks the built code object with the hardware's VM_CNT rule; a
  // weights-only call -- one store per tile -- would need its own loop copy and has no measured user: the fused flow -> weight tail is
  // column-major and takes the compiler-tracked loads anyway).
  const bool vec_rows = PIPE_OK && momenta != nullptr && !col_major && (NIN % 2 == 0) && k.r_ld == NIN && k.ncol == NIN && ((unsigned long long)r & 15ull) == 0;
  // vec_rows: SOFTWARE-PIPELINED inputs.  The 16-byte loads of the NEXT tile are issued by inline asm (rb_issue16) into registers the
  // compiler regards as ready, and the one wait that guards them is written by hand at the top of the next iteration:
  // s_waitcnt vmcnt(<number of momenta stores a whole tile issues after its loads>).  vmcnt counts loads and stores in issue order on
  // gfx9, so that wait lets the loads land without waiting for the tile's own stores; the compiler's bookkeeping cannot express it
  // (at the loop header it merges the pre-loop state, no store behind the loads, into vmcnt(0): every wave then sat out the full
  // HBM latency of its stores once per tile).  Whole tiles only (fixed store count); a ragged last tile waits for vmcnt(0).
  // More operations behind the loads than counted (status atomics) only make the wait stricter.
  constexpr int NRAW = (NPS + 3) / 2;
  auto issue = [&](long t, nis_d2 (&raw)[NRAW]) {
    const long ii = t * RB_TPB + tid;
    const double* p = r + (size_t)(ii < N ? ii : 0) * NIN;
    rb_issue_row<NIN / 2>(raw, p);
  };
  auto fetch = [&](long t, double (&dst)[NPS + 2]) {              // strided / column-major inputs: element loads, used at once
    const long ii = t * RB_TPB + tid;
    const bool ok = ii < N;
#pragma unroll
    for (int c = 0; c < NPS + 2; ++c) {
      double v = 0.5;
      if (c < NIN && ok && c < k.ncol) v = col_major ? r[(size_t)k.cols[c] * k.r_ld + ii] : r[(size_t)ii * k.r_ld + c];
      dst[c] = v;
    }
  };
  double rvn[NPS + 2];
  nis_d2 rawn[NRAW];
#pragma unroll
  for (int c2 = 0; c2 < NRAW; ++c2) rawn[c2] = nis_d2{0.5, 0.5};
  // one tile; FULL: all RB_TPB samples exist (no guards, fixed store count), VEC: inputs come through rawn
  auto process = [&](long tile, auto full_c, auto vec_c) {
    constexpr bool FULL = decltype(full_c)::value, VEC = decltype(vec_c)::value;
    const long i0 = tile * RB_TPB;
    const int ns = FULL ? RB_TPB : (int)min((long)RB_TPB, N - i0);
    const long i = i0 + tid;
    const bool live = FULL ? true : tid < ns;
    double rv[NPS + 2];
    bool nan_in = false;
    const bool has_mom = VEC || momenta != nullptr;     // the pipelined path exists for calls with momenta only (vec_rows)
    if (VEC) {
      // the previous whole tile's LM / 2 momenta stores stay in flight (its weight store, issued last, may or may not exist: not counted,
      // which at worst waits for the OLDEST momenta store as well)
      if (FULL) rb_wait_loads<LM / 2>(rawn);
      else rb_wait_loads<0>(rawn);
#pragma unroll
      for (int c = 0; c < NPS + 2; ++c) {
        const double v = (c & 1) ? rawn[c / 2].y : rawn[c / 2].x;
        rv[c] = (live && c < NIN) ? v : 0.5;
      }
      // whole tiles prefetch unconditionally (last round: a valid tile again, unused).  The ragged tile is the workgroup's last and
      // must NOT: the compiler sees the row image as dead there and reuses its registers while the loads would still be landing.
      if (FULL) issue(min(tile + (long)gridDim.x, ntiles - 1), rawn);
    } else {
#pragma unroll
      for (int c = 0; c < NPS + 2; ++c) rv[c] = rvn[c];
      if (tile + gridDim.x < ntiles) fetch(tile + gridDim.x, rvn);
    }
#pragma unroll
    for (int c = 0; c < NPS + 2; ++c) nan_in |= (rv[c] != rv[c]);
    if (nan_in && live) atomicOr(status, 2);                  // the reference raises PhaseSpaceGeneratorError (:147-149)

    // ---- PDFs: tau / y_cm sampling (:157-187)
    double wgt = 1.0, x1 = 1.0, x2 = 1.0, E = k.d.E_cm;
    if (pdf_active) {
      if (use_tau) {
        const double d_tau = 1.0 - k.tau_min;
        const double tau = k.tau_min + d_tau * rv[NPS];
        const double ymin = 0.5 * log(tau);
        const double d_y = (-ymin) - ymin;
        const double ycm = ymin + d_y * rv[NPS + 1];
        x1 = nis_sqrt(tau) * exp(ycm);
        x2 = nis_sqrt(tau) * exp(-ycm);
        E = nis_sqrt(tau) * k.d.E_cm;
        wgt *= d_tau * d_y;
      } else {
        x1 = rv[NPS + 1];
        x2 = rv[NPS];
        E = nis_sqrt(x1 * x2) * k.d.E_cm;
      }
      if (k.pdf[0].nx > 0) wgt *= rb_pdf_xf(k.pdf[0], x1) / x1;      // get_pdfQ2 (:120-137): x f(x, 91.188^2) / x per beam
      if (k.pdf[1].nx > 0) wgt *= rb_pdf_xf(k.pdf[1], x2) / x2;
      if (x1 < 1e-4 || x2 < 1e-4) wgt *= 0.0;         // x_cut (:185-186)
    }

    // ---- intermediate masses: lattice bisection per column (:313-359), then :363-406
    double M[NF], Kk[NF];
    M[0] = E - k.mtot;
#pragma unroll
    for (int c = 0; c < NF - 2; ++c) {
      constexpr int dummy = 0;
      (void)dummy;
      const int e = NF - 2 - c;
      const double v = rv[c];
      double u;
      if (bisect_levels > 0) {
        // the reference's integer-lattice bisection, fixed depth (bit-compatible path, ~14 VALU per level)
        double left = 0.0, right = 1.0, scale = 0.5;
        u = -1.0;
        for (int level = 0; level < bisect_levels; ++level) {
          u = (left + right) * scale;
          const double chk = rb_ipow(u, e) * ((double)(e + 1) - (double)e * u);      // massless_map :101-103
          left *= 2.;
          right *= 2.;
          const double adder = (v <= chk) ? -0.5 : 0.5;
          left += adder + 0.5;
          right += adder - 0.5;
          scale *= 0.5;
        }
      } else {
        // v == 0 exactly (a float32 draw is 0.0 once per 1.7e7 values): the reference's lattice bisection ends on its lowest
        // lattice point 2^-levels > 0 (120-180 levels), never on 0, and its weight stays finite (every K ratio is 0/0 at u = 0)
        u = fmax(rb_solve(v, e), 0x1p-180);
      }
      M[c + 1] = nis_sqrt(u * (M[c] * M[c]));                                           // :370
    }
    double w_ps = pdf_active ? k.flat_ab * (rb_ipow(E * E, NF - 2) / k.flat_fact) : k.flat_const;   // :81-97
#pragma unroll
    for (int c = 0; c < NF - 1; ++c) {
      Kk[c] = M[c];
      M[c] += k.msum[c];                                                            // :391-392
    }
    // rho numerators and 1 / M_c once: the decays below need the same square roots (q = 4 M rho = num / (2 M)) and reciprocals
    M[NF - 1] = k.d.final_masses[NF - 1];
    double rn[NF - 1], iM[NF - 1];
#pragma unroll
    for (int c = 0; c < NF - 1; ++c) {
      rn[c] = rb_rho_num(M[c], M[c + 1], k.d.final_masses[c]);
      iM[c] = nis_rcp(M[c]);
    }
    w_ps *= rn[NF - 2] * (iM[NF - 2] * iM[NF - 2]);                                        // 8 rho(M_n-2, m_n-1, m_n-2)  :394-397
    double prod = 1.0;
#pragma unroll
    for (int c = 0; c < NF - 2; ++c)
      // rho(M_c, M_c+1, m_c) / rho(K_c, K_c+1, 0) * (M_c+1 / K_c+1) with one reciprocal per factor (the 8 M^2 denominators cross
      // over; every ratio is O(1) although its terms reach 1e-115 in the rows of an exact-zero draw -- one reciprocal for the whole
      // product underflows there); the massless numerator sqrt((K_c^2 - K_c+1^2)^2) is K_c^2 - K_c+1^2 itself (K decreases)   :400-401
      prod *= (rn[c] * (Kk[c] * Kk[c]) * M[c + 1]) * nis_rcp((Kk[c] * Kk[c] - Kk[c + 1] * Kk[c + 1]) * (M[c] * M[c]) * Kk[c + 1]);
    w_ps *= prod;
    w_ps *= rb_ipow(Kk[0] * iM[0], 2 * NF - 4);                                               // :403
    double wtot = wgt * w_ps;

    // ---- sequential two-body decays (:223-278)
    // MODE 0 hands every finished leg straight to the LDS output image (nothing reads the momenta again: no lab boost, no
    // cuts), which frees ~8 (n + 2) registers; the other modes keep them for the lab boost / cut observables.
    constexpr bool STREAM = MODE == 0;
    double* const row = rb_lds + tid * (LM + 1);     // +1: odd row stride -> conflict-free column writes
    auto put_leg = [&](int l, const V4& v) {
      row[4 * l + 0] = v.e;
      row[4 * l + 1] = v.x;
      row[4 * l + 2] = v.y;
      row[4 * l + 3] = v.z;
    };
    V4 mom[STREAM ? 1 : NF + 2];
    V4 Q = {M[0], 0., 0., 0.};
    bool bad_boost = false;
#pragma unroll
    for (int c = 0; c < NF - 1; ++c) {
      const double q = rn[c] * (0.5 * iM[c]);                                         // 4 M rho(M_c, M_c+1, m_c)  :228
      const double cos_t = 2. * rv[NF - 2 + 2 * c] - 1.;
      const double sin_t = nis_sqrt(1. - cos_t * cos_t);
      const double phi = 2 * M_PI * rv[NF - 2 + 2 * c + 1];
      const double cos_p = rb_cospi(2.0 * rv[NF - 2 + 2 * c + 1], k);  // = cos(2 pi r)
      const double sq = nis_sqrt(1. - cos_p * cos_p);
      const double sin_p = phi > M_PI ? -sq : sq;                                   // :241-243
      V4 p = {0., q * sin_t * cos_p, q * sin_t * sin_p, q * cos_t};
      const double m2 = k.d.final_masses[c] * k.d.final_masses[c];
      rb_set_square(p, m2);
      bad_boost |= (Q.e <= 0.) || ((Q.e * Q.e - Q.x * Q.x - Q.y * Q.y - Q.z * Q.z) < 0.);   // utils.py:33
      if (c > 0) {
        // boost into the frame of Q (boostVector_t + boost_t, utils.py:36, 58-81).  Q is on its mass shell M_c (set_square below), so
        // gamma = Q_0 / M_c, gamma b = Q / M_c and (gamma - 1) / b^2 = gamma^2 / (gamma + 1): no 1 - b^2 (the reference's own
        // cancellation), no rsqrt, one reciprocal.  c == 0: Q = (M_0, 0, 0, 0), the boost is the identity.
        const double nx = Q.x * iM[c], ny = Q.y * iM[c], nz = Q.z * iM[c], gam = Q.e * iM[c];
        const double np_ = p.x * nx + p.y * ny + p.z * nz;
        const double f = fma(np_, nis_rcp(gam + 1.0), p.e);
        p.x += f * nx;
        p.y += f * ny;
        p.z += f * nz;
        rb_set_square(p, m2);                                                         // the boosted energy is re-derived from the mass shell
      }
      if constexpr (STREAM) {
        if (has_mom) put_leg(2 + c, p);
      } else {
        mom[2 + c] = p;
      }
      Q.e -= p.e;
      Q.x -= p.x;
      Q.y -= p.y;
      Q.z -= p.z;
      rb_set_square(Q, M[c + 1] * M[c + 1]);
    }
    if constexpr (STREAM) {
      if (has_mom) put_leg(NF + 1, Q);
    } else {
      mom[NF + 1] = Q;
    }
    if (bad_boost && live) atomicOr(status, 8);

    // ---- initial state (:408-441)
    V4 in0, in1;
    if (k.d.initial_masses[0] == 0. || k.d.initial_masses[1] == 0.) {
      in0 = {E / 2.0, 0., 0., E / 2.0};
      in1 = {E / 2.0, 0., 0., -(E / 2.0)};
    } else {
      const double M1 = k.d.initial_masses[0] * k.d.initial_masses[0], M2 = k.d.initial_masses[1] * k.d.initial_masses[1];
      const double E2_ = E * E;
      const double E1 = (E2_ + M1 - M2) / E, E2 = (E2_ - M1 + M2) / E;
      const double Z = nis_sqrt(E2_ * E2_ - 2 * E2_ * M1 - 2 * E2_ * M2 + M1 * M1 - 2 * M1 * M2 + M2 * M2) / E;
      in0 = {E1 / 2.0, 0., 0., Z / 2.0};
      in1 = {E2 / 2.0, 0., 0., -(Z / 2.0)};
    }
    if constexpr (!STREAM) {
      mom[0] = in0;
      mom[1] = in1;
    }
    if (has_mom) {                                   // COM-frame copy is what the reference returns (:282,308)
      if constexpr (STREAM) {
        put_leg(0, in0);
        put_leg(1, in1);
      } else {
#pragma unroll
        for (int l = 0; l < NF + 2; ++l) put_leg(l, mom[l]);
      }
      // each wave stages and streams out ITS OWN 64 rows: the LDS unit serves one wave's accesses in order, so the wave's
      // reads see its own writes without a workgroup barrier (two s_barrier per tile stalled every wave on the slowest one)
      __builtin_amdgcn_wave_barrier();
      const int wv = tid >> 6, ln = tid & 63;
      const int nsw = min(64, ns - wv * 64);           // rows of this wave in the (possibly ragged) tile
      double* dst = momenta + ((size_t)i0 + wv * 64) * LM;
      const double* src = rb_lds + (size_t)wv * 64 * (LM + 1);
      const int tot = nsw * LM;                        // LM is even: a 16-byte pair never straddles two rows
      auto put = [&](int e) {                          // 16 bytes per lane and store: 1 KiB contiguous per wave instruction
        const int s_ = e / LM, q = e - s_ * LM;        // LM is a compile-time constant
        const nis_d2 v = {src[s_ * (LM + 1) + q], src[s_ * (LM + 1) + q + 1]};
#ifdef RB_ABLATE_STORE                                   // timing-only build: everything but the momenta stores (results wrong)
        if (v.x == 1.2345e300) *reinterpret_cast<nis_d2*>(dst + e) = v;
#elif RB_NT
        __builtin_nontemporal_store(v, reinterpret_cast<nis_d2*>(dst + e));
#else
        *reinterpret_cast<nis_d2*>(dst + e) = v;
#endif
      };
      if (FULL || nsw == 64) {                         // whole wave tile: a fixed number of stores (the hand-written wait counts them)
#pragma unroll
        for (int u = 0; u < LM / 2; ++u) {
          put(2 * ln + 128 * u);
          if (u % RB_STORE_CHUNK == RB_STORE_CHUNK - 1) __builtin_amdgcn_sched_barrier(0);   // bound the LDS read-ahead: 4 registers per store in flight
        }
      } else {
        for (int e = 2 * ln; e < tot; e += 128) put(e);
      }
      __builtin_amdgcn_wave_barrier();
    }

    // ---- lab frame (utils.py:134-146); the batch-global "any zero reference" test is the caller's
    if constexpr (HAS_PDF) if (lab_boost) {
      const double rx = mom[0].x * x1 + mom[1].x * x2, ry = mom[0].y * x1 + mom[1].y * x2, rz = mom[0].z * x1 + mom[1].z * x2;
      const double re = mom[0].e * x1 + mom[1].e * x2;
      if (live && rx * rx + ry * ry + rz * rz == 0.) atomicOr(status, 4);
      const double bx = rx / re, by = ry / re, bz = rz / re;
#pragma unroll
      for (int l = 0; l < NF + 2; ++l) rb_boost(mom[l], bx, by, bz);
    }

    // ---- cuts (:285-301)
    double factor = 1.0;
    if constexpr (HAS_CUTS) {
    if (pT_mincut > 0.) {                         // min pT >= 0 can never be below a cut <= 0 (:285-288)
      double ptmin = RB_HUGE;
#pragma unroll
      for (int l = 0; l < NF; ++l) ptmin = fmin(ptmin, fabs(nis_sqrt(mom[2 + l].x * mom[2 + l].x + mom[2 + l].y * mom[2 + l].y)));
      factor = (ptmin < pT_mincut) ? 0.0 : 1.0;
    }
    const bool need_eta = (delR_mincut > 0.) || (rap_maxcut > 0.);
    if (need_eta) {
      double eta[NF];
#pragma unroll
      for (int l = 0; l < NF; ++l) eta[l] = rb_pseudorap(mom[2 + l]);
      if (delR_mincut > 0.) {
#pragma unroll
        for (int a = 1; a < NF; ++a)
#pragma unroll
          for (int b = 0; b < a; ++b) {
            const double de = eta[a] - eta[b], dp = rb_delphi(mom[2 + a], mom[2 + b]);
            if (fabs(nis_sqrt(de * de + dp * dp)) < delR_mincut) factor = 0.0;
          }
      }
      if (rap_maxcut > 0.) {
        double emax = eta[0];
#pragma unroll
        for (int l = 1; l < NF; ++l) emax = fmax(emax, eta[l]);
        if (rap_maxcut < fabs(emax)) factor = 0.0;                              // |max eta|, not max |eta| (:298-301)
      }
    }
    }
    wtot *= factor;
    const double shat = x1 * x2 * (k.d.E_cm * k.d.E_cm);
    const double wv = wtot * nis_rcp(2 * shat);
    if (live && weight) weight[i] = wv;
    if (jac) {                                       // moment rider: f(X) J with f = the phase-space weight (manager.py:397-400)
      const double v = live ? wv * jac[i] : 0.0;
      if (mfirst) {
        mshift = wave_first(v);
        mfirst = false;
      }
      const double t = live ? v - mshift : 0.0;
      mS1 += t;
      mS2 = fma(t, t, mS2);
      mcnt += live ? 1.0 : 0.0;
      mmax = live ? fmax(mmax, v) : mmax;
    }
  };
  const std::integral_constant<bool, true> yes;
  const std::integral_constant<bool, false> no;
  if (PIPE_OK && vec_rows) {                           // (not even instantiated for the larger maps)
    if constexpr (PIPE_OK) {
      if ((long)blockIdx.x >= ntiles) return;
      issue(blockIdx.x, rawn);
      rb_wait_loads<0>(rawn);                          // first tile of the workgroup: nothing is behind its loads yet
      const long nfull = N / RB_TPB;
      long tile = blockIdx.x;
      for (; tile < nfull; tile += gridDim.x) process(tile, yes, yes);
      if (tile < ntiles) process(tile, no, yes);       // the ragged last tile, if it is this workgroup's
    }
  } else {
#ifndef RB_VEC_ONLY                                    // (register-pressure experiments: the pipelined path alone)
    if ((long)blockIdx.x < ntiles) fetch(blockIdx.x, rvn);
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) process(tile, no, no);
#endif
    if (jac) block_record<RB_TPB / 64>(mshift, mS1, mS2, mcnt, mmax, rb_fmom, k.mom_part + ((size_t)blockIdx.y * k.mom_maxb + blockIdx.x) * 4);
  }
}

static double host_fact(int n) {
  double f = 1.0;
  for (int i = 2; i <= n; ++i) f *= i;
  return f;
}

extern "C" int nis_pdf_xfx(const nis_pdf1d* th, const double* x, int64_t N, double* out, void* stream) {
  if (!th || !x || !out || N < 1 || th->nx < 2 || !th->logx || !th->xf) return NIS_E_ARG;
  const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(2048, (N + 255) / 256));
  hipLaunchKernelGGL(k_pdf_xfx, dim3(grid), dim3(256), 0, (hipStream_t)stream, *th, x, (long)N, out);
  return hipGetLastError() == hipSuccess ? NIS_OK : NIS_E_HIP;
}

extern "C" int nis_rambo_forward(const nis_rambo_desc* dh, const double* r, int64_t r_ld, int32_t r_col_major,
                                 const int32_t* r_cols_host, int64_t N, double* momenta, double* weight, int32_t* status,
                                 void* stream) {
  return nis_rambo_forward_pdf(dh, nullptr, r, r_ld, r_col_major, r_cols_host, N, momenta, weight, status, stream);
}

// common host side: descriptor -> kernel constants, grid, MODE / n dispatch.  G plane sets (gs doubles apart) as blockIdx.y;
// mom_part != null adds the moment rider (weights-only, column-major input)
static int rambo_launch(const nis_rambo_desc* dh, const nis_pdf1d* pdfh, const double* r, int64_t r_ld, int32_t r_col_major,
                        const int32_t* r_cols_host, int64_t N, double* momenta, double* weight, int32_t* status, hipStream_t s, int G,
                        long r_gs, long w_gs, const double* jac, double* mom_part, int* grid_out) {
  if (!dh || !r || !status || N < 1 || G < 1) return NIS_E_ARG;
  if (!weight && !mom_part) return NIS_E_ARG;
  if (pdfh)
    for (int b = 0; b < 2; ++b)
      if (pdfh[b].nx != 0 && (pdfh[b].nx < 2 || !pdfh[b].logx || !pdfh[b].xf)) return NIS_E_ARG;
  if (pdfh && (pdfh[0].nx || pdfh[1].nx) && !(dh->flags & NIS_RAMBO_PDF_ACTIVE)) return NIS_E_ARG;
  const int n = dh->n_final;
  if (n < 2 || n > NIS_MAX_LEGS) return NIS_E_UNSUPPORTED;
  RamboK k;
  memset(&k, 0, sizeof(k));
  k.d = *dh;
  if (pdfh) {
    k.pdf[0] = pdfh[0];
    k.pdf[1] = pdfh[1];
  }
  const bool pdf_active = dh->flags & NIS_RAMBO_PDF_ACTIVE;
  k.ncol = 3 * n - 4 + (pdf_active ? 2 : 0);
  k.r_ld = r_ld;
  k.col_major = r_col_major;
  k.r_gs = r_gs;
  k.w_gs = w_gs;
  k.jac = jac;
  k.mom_part = mom_part;
  k.mom_maxb = NIS_MAXB;
  if (!r_col_major && r_ld < k.ncol) return NIS_E_ARG;
  if (r_col_major && r_ld < N) return NIS_E_ARG;
  for (int c = 0; c < k.ncol; ++c) k.cols[c] = (r_col_major && r_cols_host) ? r_cols_host[c] : c;
  for (int i = 0; i < 11; ++i) k.cospi_c[i] = RB_COSPI_C[i];
  k.flat_ab = pow(2 * M_PI, 4 - 3 * n) * pow(M_PI / 2.0, n - 1);
  k.flat_fact = host_fact(n - 1) * host_fact(n - 2);
  k.flat_const = k.flat_ab * (pow(dh->E_cm * dh->E_cm, n - 2) / k.flat_fact);
  double acc = 0.0;
  for (int i = n - 1; i >= 0; --i) {          // torch.cumsum over the flipped masses (:391)
    acc += dh->final_masses[i];
    k.msum[i] = acc;
  }
  double tot = 0.0;
  for (int i = 0; i < n; ++i) tot += dh->final_masses[i];   // torch.sum(masses_t) (:162,:384)
  k.mtot = tot;
  const double base = (tot > 1.0 ? tot : 1.0) / dh->E_cm;   // absolute_Ecm_min = 1 (:68,:164-165)
  k.tau_min = base * base;
  const int LMh = 4 * (n + 2);
  const size_t lds_bytes = (size_t)RB_TPB * (LMh + 1) * sizeof(double);      // output staging only
  const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(8, (160 * 1024) / lds_bytes));
  // workgroups: twice what the chip holds at once (shared by the G plane sets).  An evenly loaded grid -- ceil(tiles / rounds) workgroups of
  // `rounds` tiles each -- measured 7 % SLOWER in round 2 (55.1 vs 51.7 us per 1e6): oversubscribed workgroups balance better
  int64_t ntiles_h = (N + RB_TPB - 1) / RB_TPB, cap = std::max<int64_t>(1, (int64_t)per_cu * 256 * 2 / G);
  if (mom_part) cap = std::min<int64_t>(cap, NIS_MAXB);                // one moment record per workgroup
  const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(cap, ntiles_h));
  if (grid_out) *grid_out = grid;
  const bool has_pdf = (dh->flags & (NIS_RAMBO_PDF_ACTIVE | NIS_RAMBO_LAB_BOOST)) != 0;          // TAU only matters with PDFs
  const bool has_cuts = dh->pT_mincut > 0. || dh->delR_mincut > 0. || dh->rap_maxcut > 0.;
  const bool need_gen = dh->bisect_levels > 0;
  const int mode = need_gen ? 3 : has_cuts ? 2 : has_pdf ? 1 : 0;
#define RB_LAUNCH(NFV, M) \
  hipLaunchKernelGGL((k_rambo<NFV, M>), dim3(grid, G), dim3(RB_TPB), lds_bytes, s, k, r, (long)N, momenta, weight, (int*)status)
#define RB_CASE(NFV)                   \
  case NFV:                            \
    if (mode == 0)                     \
      RB_LAUNCH(NFV, 0);               \
    else if (mode == 1)                \
      RB_LAUNCH(NFV, 1);               \
    else if (mode == 2)                \
      RB_LAUNCH(NFV, 2);               \
    else                               \
      RB_LAUNCH(NFV, 3);               \
    break;
  NisProfScope ps(NIS_K_RAMBO, s);
  switch (n) {
    RB_CASE(2) RB_CASE(3) RB_CASE(4) RB_CASE(5) RB_CASE(6) RB_CASE(7) RB_CASE(8) RB_CASE(9) RB_CASE(10)
    default: return NIS_E_UNSUPPORTED;
  }
#undef RB_CASE
#undef RB_LAUNCH
  return hipGetLastError() == hipSuccess ? NIS_OK : NIS_E_HIP;
}

extern "C" int nis_rambo_forward_pdf(const nis_rambo_desc* dh, const nis_pdf1d* pdfh, const double* r, int64_t r_ld, int32_t r_col_major,
                                     const int32_t* r_cols_host, int64_t N, double* momenta, double* weight, int32_t* status,
                                     void* stream) {
  if (!weight) return NIS_E_ARG;
  return rambo_launch(dh, pdfh, r, r_ld, r_col_major, r_cols_host, N, momenta, weight, status, (hipStream_t)stream, 1, 0, 0, nullptr, nullptr,
                      nullptr);
}

// merge of the per-workgroup records, one workgroup per plane set (api_moments.hip)
hipError_t nis_launch_moments_final(const double* partials, int nparts, double* out4, int maxb, int G, hipStream_t s);

extern "C" size_t nis_rambo_moments_workspace_bytes(int32_t n_batches) {
  return (size_t)std::max(1, n_batches) * NIS_MAXB * 4 * sizeof(double);
}

extern "C" int nis_rambo_moments(const nis_rambo_desc* dh, const nis_pdf1d* pdfh, const double* planes, int64_t ld, int64_t plane_gs,
                                 const int32_t* cols_host, int32_t jplane, int64_t N, int32_t n_batches, double* weight_out, double* out4,
                                 int32_t* status, void* workspace, size_t workspace_bytes, void* stream) {
  if (!planes || !cols_host || !out4 || jplane < 0 || n_batches < 1 || n_batches > 65535 || ld < N) return NIS_E_ARG;
  if (!workspace || workspace_bytes < nis_rambo_moments_workspace_bytes(n_batches)) return NIS_E_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  int grid = 0;
  const int rc = rambo_launch(dh, pdfh, planes, ld, 1, cols_host, N, nullptr, weight_out, status, s, n_batches, (long)plane_gs, (long)N,
                              planes + (size_t)jplane * ld, (double*)workspace, &grid);
  if (rc != NIS_OK) return rc;
  return nis_launch_moments_final((const double*)workspace, grid, out4, NIS_MAXB, n_batches, s) == hipSuccess ? NIS_OK : NIS_E_HIP;
}
