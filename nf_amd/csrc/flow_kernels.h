// Coupling-cell kernels (templates).  One thread owns one sample; the conditioner activations live
// in registers (compile-time width WM, zero padded), weights arrive through scalar loads, BatchNorm
// affine coefficients through LDS broadcast reads.  Train-mode BatchNorm needs the batch statistics
// of every BN layer, so a cell is   [k_bn_stats(l) for l = 1..L]  ->  k_cell_apply   (the BN_0 records
// ride along in the launch that produces the cell's input: k_ingest or the previous k_cell_apply):
// k_bn_stats(l) recomputes layers < l from the pass-through planes and emits per-workgroup
// (count, mean, M2) records; the NEXT launch merges them in its prologue (fixed order ->
// bitwise reproducible) into scale/shift.  No atomics, no grid barrier: the kernel boundary is
// the cheapest inter-workgroup sync on MI355X.
//
// Reference restated: normalizing_flows/layers/coupling_cells.py:107-142 (PWLin.forward),
// :159-228 (PWQuad.forward), :236-254 (RectNN), torch.nn.BatchNorm1d semantics.
#pragma once
#include <cstdlib>
#include "nis_common.h"

// ---------------------------------------------------------------------------------------------
// Prologue: bring the BN affine form (scale, shift) of layers [0, upto) into LDS.
//   layers < fresh : already finalised by an earlier launch -> c.coef (global)
//   layer == fresh : merge the partial records of the producing launch (train mode).
// The records are merged as sums shifted by record 0's mean (no divisions, |shift error| ~ sigma/sqrt(n)),
// TPB/WM slices per feature, then an LDS tree.  lds_coef: [layer][0: scale, 1: shift][WM];
// `red`: 3*TPB doubles.  Workgroup 0 publishes the fresh coefficients, applies the running-stat update
// (momentum 0.1, unbiased variance) and bumps num_batches_tracked.
// `early()` is invoked once, after the prologue's own global loads have been consumed and before its LDS tree: a caller
// puts the first sample loads there, so that their HBM latency overlaps the merge (the vector-memory counter retires in
// order: requests issued any earlier would stall the first use of a record instead).
struct NoEarly {
  __device__ __forceinline__ void operator()() const {}
};
template <int WM, int TPB, class Early = NoEarly>
__device__ __forceinline__ void load_coefs(const CellK& c, int g, int upto, int fresh, bool writer, double* lds_coef, double* red,
                                           Early&& early = Early()) {
  const int tid = threadIdx.x;
  double* coef_g = c.coef + (size_t)g * c.coef_gs;
  // the first strip of published coefficients travels through a register and is stored behind the record loads (one
  // round trip instead of two)
  const int l0 = tid / (2 * WM);
  const bool c0 = tid < upto * 2 * WM && l0 != fresh && !(fresh == 0 && c.derive1 && l0 == 1);
  const double v0 = c0 ? coef_g[tid] : 0.0;
  for (int idx = tid + TPB; idx < upto * 2 * WM; idx += TPB) {
    const int l = idx / (2 * WM);
    if (l != fresh && !(fresh == 0 && c.derive1 && l == 1)) lds_coef[idx] = coef_g[idx];
  }
  if (!(fresh >= 0 && fresh < upto)) {
    early();
    if (c0) lds_coef[tid] = v0;
  }
  if (fresh >= 0 && fresh < upto) {
    constexpr int S = TPB / WM;              // slices of partial records per feature (power of two)
    const int j = tid % WM, s = tid / WM;
    const double* part = c.sync_part ? c.sync_part : c.partials + ((size_t)fresh * c.G + g) * c.maxb * WM * 3;
    const int np = c.sync_part ? c.sync_world : c.nparts[fresh];
    const double ref = part[j * 3 + 1];      // mean of record 0 (workgroup 0 always owns samples)
    // everything the finalisation below needs from global memory is requested here, next to the record loads: the
    // prologue is a chain of dependent round trips otherwise (~1 us each, and every workgroup of the launch pays it)
    const bool d1 = fresh == 0 && c.derive1 && upto >= 1;
    const double gam_e = c.gb[(fresh * 2 + 0) * WM + j], be_e = c.gb[(fresh * 2 + 1) * WM + j];
    const double be00_e = d1 ? c.gb[1 * WM + 0] : 0.0, w0_e = d1 ? c.wp[(size_t)j * WM + 0] : 0.0;
    const double gam1_e = d1 ? c.gb[(1 * 2 + 0) * WM + j] : 0.0, be1_e = d1 ? c.gb[(1 * 2 + 1) * WM + j] : 0.0;
    // this thread's records, EIGHT at a time: the 24 loads go out together and the sums are formed in record order afterwards (same
    // arithmetic, same order -- bit-identical to a one-by-one loop, which is a chain of np / S dependent L2 round trips: with 512 records
    // of a matrix-core launch that chain was ~40 us of EVERY statistics / apply launch, i.e. most of such a launch at 2e5 samples)
    double an = 0.0, a1 = 0.0, a2 = 0.0;
    constexpr int RB = 8;                         // (16 measured no faster)
    for (int b0 = s; b0 < np; b0 += RB * S) {
      double qn[RB], qm[RB], q2[RB];
#pragma unroll
      for (int u = 0; u < RB; ++u) {
        const int b = b0 + u * S;
        const double* q = part + ((size_t)(b < np ? b : s) * WM + j) * 3;
        qn[u] = q[0];
        qm[u] = q[1];
        q2[u] = q[2];
      }
#pragma unroll
      for (int u = 0; u < RB; ++u)
        if (b0 + u * S < np) {
          const double n = qn[u], dl = qm[u] - ref;
          an += n;
          a1 = fma(n, dl, a1);
          a2 += fma(n * dl, dl, q2[u]);
        }
    }
    red[(s * WM + j) * 3 + 0] = an;
    red[(s * WM + j) * 3 + 1] = a1;
    red[(s * WM + j) * 3 + 2] = a2;
    early();
    if (c0) lds_coef[tid] = v0;
    __syncthreads();
#pragma unroll
    for (int h = S / 2; h > 0; h >>= 1) {
      if (s < h) {
        red[(s * WM + j) * 3 + 0] += red[((s + h) * WM + j) * 3 + 0];
        red[(s * WM + j) * 3 + 1] += red[((s + h) * WM + j) * 3 + 1];
        red[(s * WM + j) * 3 + 2] += red[((s + h) * WM + j) * 3 + 2];
      }
      __syncthreads();
    }
    if (s == 0) {
      const double n = red[j * 3 + 0], s1 = red[j * 3 + 1], s2 = red[j * 3 + 2];
      const int F = cellk_feat(c, fresh);
      double scale = 0.0, shift = 0.0;
      if (j < F) {
        const double mean = ref + s1 / n;
        const double m2 = fmax(s2 - s1 * s1 / n, 0.0);
        const double var_b = m2 / n;                             // biased variance normalises the batch
        const double gam = gam_e, be = be_e;
        scale = gam / sqrt(var_b + NIS_BN_EPS);
        shift = be - mean * scale;
        if (writer) {
          const double unb = n > 1.0 ? m2 / (n - 1.0) : var_b;   // running_var takes the unbiased one
          const double mean_r = mean + (c.hb ? c.hb[fresh * WM + j] : 0.0);   // what torch's BN sees: Linear output incl. its bias
          if (c.G == 1) {
            double* run = c.running + cellk_running_off(c, fresh);
            run[j] = (1.0 - NIS_BN_MOM) * run[j] + NIS_BN_MOM * mean_r;
            run[F + j] = (1.0 - NIS_BN_MOM) * run[F + j] + NIS_BN_MOM * unb;
            if (j == 0 && c.tracked) c.tracked[fresh] += 1;
          } else {                                               // applied in batch order by k_group_epilogue
            c.gstat[(((size_t)fresh * c.G + g) * 2 + 0) * WM + j] = mean_r;
            c.gstat[(((size_t)fresh * c.G + g) * 2 + 1) * WM + j] = unb;
          }
        }
      }
      lds_coef[(fresh * 2 + 0) * WM + j] = scale;
      lds_coef[(fresh * 2 + 1) * WM + j] = shift;
      if (writer) {
        coef_g[(fresh * 2 + 0) * WM + j] = scale;
        coef_g[(fresh * 2 + 1) * WM + j] = shift;
        if (c.bnsave) {
          double mean_s = 0.0, rstd_s = 0.0;
          if (j < F) {
            mean_s = ref + s1 / n;
            rstd_s = 1.0 / sqrt(fmax(s2 - s1 * s1 / n, 0.0) / n + NIS_BN_EPS);
          }
          double* bs = c.bnsave + (size_t)g * c.bnsave_gs;
          bs[(fresh * 4 + 0) * WM + j] = scale;
          bs[(fresh * 4 + 1) * WM + j] = shift;
          bs[(fresh * 4 + 2) * WM + j] = mean_s;
          bs[(fresh * 4 + 3) * WM + j] = rstd_s;
        }
      }
      if (fresh == 0 && c.derive1 && j == 0) {       // hand BN_0's statistics to the BN_1 derivation below
        const double mean = ref + s1 / n;
        red[3 * TPB - 4] = n;
        red[3 * TPB - 3] = mean;
        red[3 * TPB - 2] = fmax(s2 - s1 * s1 / n, 0.0) / n;
        red[3 * TPB - 1] = scale;
      }
      if (WM >= 16 && fresh == 0 && c.derive1 && c.dcov && j < F) {     // covariance form: every pass-through feature's statistics
        red[3 * TPB - 64 + j] = ref + s1 / n;                            // mean_a
        red[3 * TPB - 48 + j] = fmax(s2 - s1 * s1 / n, 0.0) / n;         // biased variance
        red[3 * TPB - 32 + j] = scale;                                   // scale0_a
      }
    }
    if (WM >= 16 && d1 && c.dcov) {
      // 2 <= P <= 8: h1_j = sum_a W0[j][a] (scale0_a x_a + shift0_a) is affine in the pass-through block, hence
      //   mean(h1_j) = sum_a W0[j][a] beta0_a,   var_b(h1_j) = sum_ab (W0[j][a] scale0_a) (W0[j][b] scale0_b) Cov(x_a, x_b)
      // The co-moments C_ab sit behind the P feature records of every BN_0 record (k_bn_stats0_cov); merged like the M2's:
      //   C_ab = sum_blocks [C_ab^blk + n_blk dl_a dl_b] - n (M_a - ref_a)(M_b - ref_b),   dl = block mean - ref.
      __syncthreads();
      const int P = c.P, npair = P * (P - 1) / 2;
      int pa = 0, pb = 1;
      {
        int m = j;                                    // pair index -> (a, b), a < b
        for (int a = 0; a < P - 1; ++a) {
          const int row = P - 1 - a;
          if (m < row) { pa = a; pb = a + 1 + m; break; }
          m -= row;
        }
      }
      double acc = 0.0;
      if (j < npair) {
        const double ra = part[pa * 3 + 1], rb = part[pb * 3 + 1];
        for (int b0 = s; b0 < np; b0 += 8 * S) {               // eight records' loads in flight, sums in record order (see above)
          double qn[8], qa[8], qb[8], qc[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int b = b0 + u * S;
            const double* q = part + (size_t)(b < np ? b : s) * WM * 3;
            qn[u] = q[0];
            qa[u] = q[pa * 3 + 1];
            qb[u] = q[pb * 3 + 1];
            qc[u] = q[3 * P + j];
          }
#pragma unroll
          for (int u = 0; u < 8; ++u)
            if (b0 + u * S < np) acc += fma(qn[u] * (qa[u] - ra), qb[u] - rb, qc[u]);
        }
      }
      red[s * WM + j] = acc;
      __syncthreads();
#pragma unroll
      for (int h = S / 2; h > 0; h >>= 1) {
        if (s < h) red[s * WM + j] += red[(s + h) * WM + j];
        __syncthreads();
      }
      if (s == 0 && j < npair) {
        const double n = red[3 * TPB - 4];
        const double ra = part[pa * 3 + 1], rb = part[pb * 3 + 1];
        const double cab = (red[j] - n * (red[3 * TPB - 64 + pa] - ra) * (red[3 * TPB - 64 + pb] - rb)) / n;
        red[3 * TPB - 128 + pa * 8 + pb] = cab;
        red[3 * TPB - 128 + pb * 8 + pa] = cab;
      }
      if (s == 0 && j < P) red[3 * TPB - 128 + j * 8 + j] = red[3 * TPB - 48 + j];
      __syncthreads();
      if (tid < WM) {
        const double n = red[3 * TPB - 4];
        const int F1 = cellk_feat(c, 1);
        double scale = 0.0, shift = 0.0, mean1 = 0.0, var1 = 0.0;
        if (tid < F1) {
          double t[8];
#pragma unroll
          for (int a = 0; a < 8; ++a) {
            const double w = a < P ? c.wp[(size_t)tid * WM + a] : 0.0;
            mean1 = fma(w, a < P ? c.gb[1 * WM + a] : 0.0, mean1);                 // scale0_a mean_a + shift0_a = beta0_a
            t[a] = a < P ? w * red[3 * TPB - 32 + a] : 0.0;
          }
#pragma unroll
          for (int a = 0; a < 8; ++a)
#pragma unroll
            for (int b = 0; b < 8; ++b)
              if (a < P && b < P) var1 = fma(t[a] * t[b], red[3 * TPB - 128 + a * 8 + b], var1);
          var1 = fmax(var1, 0.0);
          const double gam = c.gb[(1 * 2 + 0) * WM + tid], be = c.gb[(1 * 2 + 1) * WM + tid];
          scale = gam / sqrt(var1 + NIS_BN_EPS);
          shift = be - mean1 * scale;
          if (writer) {
            const double unb = n > 1.0 ? var1 * (n / (n - 1.0)) : var1;
            const double mean1_r = mean1 + (c.hb ? c.hb[1 * WM + tid] : 0.0);
            if (c.G == 1) {
              double* run = c.running + cellk_running_off(c, 1);
              run[tid] = (1.0 - NIS_BN_MOM) * run[tid] + NIS_BN_MOM * mean1_r;
              run[F1 + tid] = (1.0 - NIS_BN_MOM) * run[F1 + tid] + NIS_BN_MOM * unb;
              if (tid == 0 && c.tracked) c.tracked[1] += 1;
            } else {
              c.gstat[(((size_t)1 * c.G + g) * 2 + 0) * WM + tid] = mean1_r;
              c.gstat[(((size_t)1 * c.G + g) * 2 + 1) * WM + tid] = unb;
            }
          }
        }
        lds_coef[(1 * 2 + 0) * WM + tid] = scale;
        lds_coef[(1 * 2 + 1) * WM + tid] = shift;
        if (writer) {
          coef_g[(1 * 2 + 0) * WM + tid] = scale;
          coef_g[(1 * 2 + 1) * WM + tid] = shift;
          if (c.bnsave) {
            double* bs = c.bnsave + (size_t)g * c.bnsave_gs;
            bs[(1 * 4 + 0) * WM + tid] = scale;
            bs[(1 * 4 + 1) * WM + tid] = shift;
            bs[(1 * 4 + 2) * WM + tid] = mean1;
            bs[(1 * 4 + 3) * WM + tid] = tid < F1 ? 1.0 / sqrt(var1 + NIS_BN_EPS) : 0.0;
          }
        }
      }
    } else
    if (d1) {
      // P == 1: the pre-BN_1 activation h1_j = W0[j][0] * (scale0 * x + shift0) is affine in x, hence
      //   mean(h1_j) = W0[j][0] * (scale0 * mean0 + shift0),   var_b(h1_j) = (W0[j][0] * scale0)^2 * var_b(x)
      __syncthreads();
      if (tid < WM) {
        const double n = red[3 * TPB - 4], mean0 = red[3 * TPB - 3], var0 = red[3 * TPB - 2], scale0 = red[3 * TPB - 1];
        const double shift0 = be00_e - mean0 * scale0;      // tid < WM: j == tid, the early loads are this thread's
        const int F1 = cellk_feat(c, 1);
        double scale = 0.0, shift = 0.0, mean1 = 0.0, var1 = 0.0;
        if (tid < F1) {
          const double a = w0_e;
          mean1 = a * fma(scale0, mean0, shift0);
          const double as = a * scale0;
          var1 = as * as * var0;
          const double gam = gam1_e, be = be1_e;
          scale = gam / sqrt(var1 + NIS_BN_EPS);
          shift = be - mean1 * scale;
          if (writer) {
            const double unb = n > 1.0 ? var1 * (n / (n - 1.0)) : var1;
            const double mean1_r = mean1 + (c.hb ? c.hb[1 * WM + tid] : 0.0);
            if (c.G == 1) {
              double* run = c.running + cellk_running_off(c, 1);
              run[tid] = (1.0 - NIS_BN_MOM) * run[tid] + NIS_BN_MOM * mean1_r;
              run[F1 + tid] = (1.0 - NIS_BN_MOM) * run[F1 + tid] + NIS_BN_MOM * unb;
              if (tid == 0 && c.tracked) c.tracked[1] += 1;
            } else {
              c.gstat[(((size_t)1 * c.G + g) * 2 + 0) * WM + tid] = mean1_r;
              c.gstat[(((size_t)1 * c.G + g) * 2 + 1) * WM + tid] = unb;
            }
          }
        }
        lds_coef[(1 * 2 + 0) * WM + tid] = scale;
        lds_coef[(1 * 2 + 1) * WM + tid] = shift;
        if (writer) {
          coef_g[(1 * 2 + 0) * WM + tid] = scale;
          coef_g[(1 * 2 + 1) * WM + tid] = shift;
          if (c.bnsave) {
            double* bs = c.bnsave + (size_t)g * c.bnsave_gs;
            bs[(1 * 4 + 0) * WM + tid] = scale;
            bs[(1 * 4 + 1) * WM + tid] = shift;
            bs[(1 * 4 + 2) * WM + tid] = mean1;
            bs[(1 * 4 + 3) * WM + tid] = tid < F1 ? 1.0 / sqrt(var1 + NIS_BN_EPS) : 0.0;
          }
        }
      }
    }
  }
  __syncthreads();
}

// Per-feature records of a workgroup: every thread holds cnt and S1[f], S2[f] (sums of v - shift, (v - shift)^2
// with the wave-uniform shift wshift[wave][f]).  out: record f at out[f*3 .. f*3+2].
template <int F, int NW>
__device__ __forceinline__ void emit_records(const double (&S1)[F], const double (&S2)[F], double cnt, double (*wshift)[F],
                                             double (*wmom)[F][3], int nvalid, double* out) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const double nw = wave_sum(cnt);
#pragma unroll
  for (int f = 0; f < F; ++f) {
    const double s1 = wave_sum(S1[f]);
    const double s2 = wave_sum(S2[f]);
    if (lane == 0) {
      const bool any = nw > 0.0;
      wmom[wave][f][0] = nw;
      wmom[wave][f][1] = any ? wshift[wave][f] + s1 / nw : 0.0;
      wmom[wave][f][2] = any ? fmax(s2 - s1 * s1 / nw, 0.0) : 0.0;
    }
  }
  __syncthreads();
  if (tid < F && tid < nvalid) {
    Mom t = {0.0, 0.0, 0.0};
    for (int w = 0; w < NW; ++w) {
      Mom m = {wmom[w][tid][0], wmom[w][tid][1], wmom[w][tid][2]};
      t = mom_merge(t, m);
    }
    out[tid * 3 + 0] = t.n;
    out[tid * 3 + 1] = t.mean;
    out[tid * 3 + 2] = t.m2;
  }
}

// h <- W(WMxWM, row-major, zero padded) * a.  WX < WM: every layer is known to be at most WX wide (the padding rows and
// columns are zero), so only the WX x WX block is multiplied.
template <int WM, int WX = WM>
__device__ __forceinline__ void dense(nis_cptr W, const double (&a)[WM], double (&b)[WM]) {
#pragma unroll
  for (int j = 0; j < WM; ++j) {
    double acc = 0.0;
    if (j < WX) {
#pragma unroll
      for (int i = 0; i < WX; ++i) acc = fma(W[j * WM + i], a[i], acc);
    }
    b[j] = acc;
  }
}

template <int WM>
__device__ __forceinline__ void load_passthrough(const CellK& c, const double* planes, long ld, long i, double (&a)[WM]) {
#pragma unroll
  for (int p = 0; p < WM; ++p) a[p] = (p < c.P) ? planes[(size_t)c.in_plane[p < NIS_MAX_DIM ? p : 0] * ld + i] : 0.0;
}

// Conditioner up to (and excluding) BN layer `upto`: leaves the pre-BN activations of that layer in `a`
// (upto == 0: the raw pass-through block).  With `finish`, BN_upto (+ReLU) is applied as well.
// `fold` (single pass-through column, see fold_coefs): BN_0, Linear_0 and BN_1 are one affine map of x that sits in
// BN_1's coefficient slot, so the stack starts at layer 1 with x broadcast to every feature.
template <int WM, int WX = WM>
__device__ __forceinline__ void hidden_stack(const CellK& c, const double* lds_coef, int upto, bool finish, double (&a)[WM],
                                             bool fold = false) {
  double b[WM];
  nis_cptr wp = as_const(c.wp);
  int l = 0;
  if (fold && upto >= 1) {
    const double x = a[0];
#pragma unroll
    for (int j = 0; j < WM; ++j) a[j] = j < WX ? x : 0.0;
    l = 1;
  }
  for (; l < upto; ++l) {
    const double* sc = lds_coef + (l * 2 + 0) * WM;
    const double* sh = lds_coef + (l * 2 + 1) * WM;
#pragma unroll
    for (int j = 0; j < WX; ++j) {
      const double v = fma(a[j], sc[j], sh[j]);
      a[j] = (l == 0) ? v : fmax(v, 0.0);                 // BN_0 has no ReLU behind it (coupling_cells.py:236-239)
    }
    dense<WM, WX>(wp + (size_t)l * WM * WM, a, b);
#pragma unroll
    for (int j = 0; j < WM; ++j) a[j] = b[j];
  }
  if (finish) {
    const double* sc = lds_coef + (upto * 2 + 0) * WM;
    const double* sh = lds_coef + (upto * 2 + 1) * WM;
#pragma unroll
    for (int j = 0; j < WX; ++j) {
      const double v = fma(a[j], sc[j], sh[j]);
      a[j] = (upto == 0) ? v : fmax(v, 0.0);
    }
  }
}

// P == 1: h1_j = W0[j][0] (sc0 x + sh0), so BN_1(h1)_j = A_j x + B_j with A_j = sc1_j W0[j][0] sc0 and
// B_j = sc1_j W0[j][0] sh0 + sh1_j.  The pair replaces BN_1's (scale, shift) in the LDS copy of the coefficients
// (the published coefficients and the saved statistics stay unfolded); hidden_stack(.., fold = true) then skips
// BN_0 and Linear_0.  Called by every thread after load_coefs; `upto` = layers present in lds_coef.
// `w_early` = fold_weight(c), requested before load_coefs so that its round trip overlaps the prologue's.
template <int WM>
__device__ __forceinline__ double fold_weight(const CellK& c) {
  return (c.fold && threadIdx.x < WM) ? c.wp[(size_t)threadIdx.x * WM] : 0.0;
}
template <int WM>
__device__ __forceinline__ void fold_coefs(const CellK& c, double* lds_coef, int upto, double w_early) {
  if (!c.fold || upto < 2) return;
  double A = 0.0, B = 0.0;
  const int j = threadIdx.x;
  if (j < WM) {
    const double w = w_early, sc1 = lds_coef[2 * WM + j];
    A = sc1 * (w * lds_coef[0]);
    B = fma(sc1, w * lds_coef[WM], lds_coef[3 * WM + j]);
  }
  __syncthreads();
  if (j < WM) {
    lds_coef[2 * WM + j] = A;
    lds_coef[3 * WM + j] = B;
  }
  __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// Batch statistics of BN layer `layer` (0..L): partial (count, mean, M2) per workgroup & feature.
// P1: the launch is known to have a single pass-through column (one plane load per sample instead of WM guarded ones)
template <int WM, int TPB, int WX = WM, bool P1 = false>
__global__ __launch_bounds__(TPB) void k_bn_stats(CellK c, const double* __restrict__ planes, long ld, long N, int layer) {
  constexpr int NW = TPB / NIS_WAVE;
  __shared__ double lds_coef[NIS_MAX_HIDDEN * 2 * WM];
  __shared__ double red[3 * TPB];
  __shared__ double wshift[NW][WM];
  __shared__ double wmom[NW][WM][3];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = c.g0 + blockIdx.y;
  planes += (size_t)g * c.plane_gs;
  // the previous layer's records are merged here (every workgroup of batch g redoes the small tree merge: cheaper than a launch)
  const double w_fold = fold_weight<WM>(c);
  load_coefs<WM, TPB>(c, g, layer, c.pmerge ? cellk_fresh(c, layer) : -1, blockIdx.x == 0, lds_coef, red);
  fold_coefs<WM>(c, lds_coef, layer, w_fold);
  const bool fold = c.fold && layer >= 2;

  double S1[WM], S2[WM], shf[WM];
#pragma unroll
  for (int j = 0; j < WM; ++j) S1[j] = S2[j] = shf[j] = 0.0;
  double cnt = 0.0;
  const long stride = (long)gridDim.x * TPB;
  constexpr int U = WM <= 8 ? 4 : 1;           // samples in flight per thread: the loop is load-latency bound
  constexpr int PU = P1 ? 1 : WM <= 8 ? WM : 1;
  const int cP = P1 ? 1 : c.P;
  bool firstit = true;
  for (long i0 = (long)blockIdx.x * TPB + tid; i0 < N; i0 += stride * U) {
    double pre[U][PU];
    if (U > 1) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const long i = i0 + u * stride;
#pragma unroll
        for (int p = 0; p < PU; ++p) pre[u][p] = (p < cP && i < N) ? planes[(size_t)c.in_plane[p] * ld + i] : 0.0;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long i = i0 + u * stride;
      if (i < N) {
        double a[WM];
        if (U > 1) {
#pragma unroll
          for (int p = 0; p < WM; ++p) a[p] = p < PU ? pre[u][p < PU ? p : 0] : 0.0;
        } else {
          load_passthrough<WM>(c, planes, ld, i, a);
        }
        hidden_stack<WM, WX>(c, lds_coef, layer, false, a, fold);
        if (firstit) {                  // wave-uniform shift = lane 0's first value (lane 0 owns the lowest index)
#pragma unroll
          for (int j = 0; j < WM; ++j) {
            shf[j] = wave_first(a[j]);  // kept in registers (scalar after readfirstlane): no LDS read per sample
            if (lane == 0) wshift[wave][j] = shf[j];
          }
          firstit = false;
        }
#pragma unroll
        for (int j = 0; j < WX; ++j) {
          const double t = a[j] - shf[j];
          S1[j] += t;
          S2[j] = fma(t, t, S2[j]);
        }
        cnt += 1.0;
      }
    }
  }
  emit_records<WM, NW>(S1, S2, cnt, wshift, wmom, WM, c.partials + (((size_t)layer * c.G + g) * c.maxb + blockIdx.x) * WM * 3);
}

// The same statistics for the folded single-column chain (P == 1, CellK::fold, every layer at most WX wide) with the BN
// layer a compile-time constant: every coefficient and weight of the stack is hoisted into registers before the sample
// loop (the generic kernel re-reads them per sample and layer behind s_waitcnt lgkmcnt(0)), whole wave tiles run
// unguarded with U loads in flight off one scalar base, and the wave's shift enters the last layer's FMA chain as
// its start value.  PMC on C2: 67 -> 44 VALU instructions per sample, no scalar-memory or LDS traffic in the loop.
#ifndef NIS_STATS_U
#define NIS_STATS_U 4
#endif
#ifndef NIS_STATS_PIN
#define NIS_STATS_PIN 0
#endif
// ROWS: the pass-through column is column c.x_wcol of the float32 (N, 2) rows of w (CellK::wrows), not a plane.
// PLN (ROWS kernels): w is two float32 column planes (CellK::wplanar) -- one 4-byte load per sample.
template <int WM, int TPB, int WX, int LAYER, bool ROWS = false, bool PLN = false>
__global__ __launch_bounds__(TPB) void k_bn_stats_fold(CellK c, const double* __restrict__ planes, long ld, long N) {
  static_assert(LAYER >= 2 && LAYER <= NIS_MAX_HIDDEN && WX <= WM, "folded statistics start at BN layer 2");
  constexpr int NW = TPB / NIS_WAVE;
  __shared__ double lds_coef[NIS_MAX_HIDDEN * 2 * WM];
  __shared__ double red[3 * TPB];
  __shared__ double wshift[NW][WM];
  __shared__ double wmom[NW][WM][3];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = c.g0 + blockIdx.y;
  planes += (size_t)g * c.plane_gs;
  // Requests that do not depend on the prologue go out first: the weights (scalar loads), the sample that fixes the
  // wave's shift and the first U tiles -- their latency overlaps the record merge.
  const double w_fold = fold_weight<WM>(c);
  double cs[LAYER][WX], ch[LAYER][WX], W[LAYER][WX][WX];           // rows 1 .. LAYER-1 are used; row 1 of cs/ch holds (A, B)
  nis_cptr wp = as_const(c.wp);
#pragma unroll
  for (int l = 1; l < LAYER; ++l) {
#pragma unroll
    for (int j = 0; j < WX; ++j) {
#pragma unroll
      for (int i = 0; i < WX; ++i) W[l][j][i] = wp[(size_t)l * WM * WM + j * WM + i];
    }
  }
  constexpr int U = NIS_STATS_U;
  const double* px = planes + (size_t)c.in_plane[0] * ld;
  typedef float nis_f2 __attribute__((ext_vector_type(2)));
  const nis_f2* wr = ROWS ? reinterpret_cast<const nis_f2*>(c.wrows + (size_t)g * c.wrows_gs) : nullptr;
  const bool wcol1 = ROWS && c.x_wcol == 1;
  const float* const wpl = (ROWS && PLN) ? c.wrows + (size_t)g * c.wrows_gs + (size_t)c.x_wcol * c.wplane_ld : nullptr;
  auto sample = [&](long tile0, int l) -> double {                // tile0 wave-uniform, l = lane
    if (!ROWS) return (px + tile0)[l];
    if (PLN) return (double)(wpl + tile0)[l];
    const nis_f2 r = (wr + tile0)[l];                             // the whole 8-byte row: same request shape as a plane load
    return (double)(wcol1 ? r.y : r.x);
  };
  const long stride = (long)gridDim.x * TPB;
  long wbase = (long)blockIdx.x * TPB + wave * NIS_WAVE;          // wave-uniform: index of lane 0's next sample
  bool full = wbase + (U - 1) * stride + NIS_WAVE <= N;           // wave-uniform: U whole tiles ahead
  // raw loads (no conversion on them): the values of the NEXT trip are requested before the current one is computed and first
  // touched one whole trip later, so the wave waits on nothing but its own arithmetic (a float -> double conversion right behind
  // the load put the full memory latency in front of every trip)
  struct Raw {
    double d;
    float f;
    nis_f2 r;
  };
  auto load_raw = [&](long tile0, int l) -> Raw {                // tile0 wave-uniform, l = lane
    Raw v = {0.0, 0.f, {0.f, 0.f}};
    if (!ROWS) v.d = (px + tile0)[l];
    else if (PLN) v.f = (wpl + tile0)[l];
    else v.r = (wr + tile0)[l];
    return v;
  };
  auto to_x = [&](const Raw& v) -> double { return !ROWS ? v.d : PLN ? (double)v.f : (double)(wcol1 ? v.r.y : v.r.x); };
  double x_first = 0.0;
  Raw xr[U];
  auto first_loads = [&]() {
    x_first = wbase < N ? sample(wbase, 0) : 0.0;
#pragma unroll
    for (int u = 0; u < U; ++u) xr[u] = load_raw(full ? wbase + u * stride : 0, full ? lane : 0);   // not full: element 0 for every lane (valid, unused;
                                                                                                   // `lane` there read up to 63 elements past a batch of < 64 samples)
  };
  load_coefs<WM, TPB>(c, g, LAYER, c.pmerge ? cellk_fresh(c, LAYER) : -1, blockIdx.x == 0, lds_coef, red, first_loads);
  fold_coefs<WM>(c, lds_coef, LAYER, w_fold);
#pragma unroll
  for (int l = 1; l < LAYER; ++l) {
#pragma unroll
    for (int j = 0; j < WX; ++j) {
      cs[l][j] = lds_coef[(l * 2 + 0) * WM + j];
      ch[l][j] = lds_coef[(l * 2 + 1) * WM + j];
    }
  }
  // pre-BN activations of layer LAYER minus the shift: t_j = h_j - shf_j  (start value of the last FMA chain: -shf_j)
  auto stack = [&](double x, const double (&nshf)[WX], double (&t)[WX]) {
    double a[WX];
#pragma unroll
    for (int j = 0; j < WX; ++j) a[j] = fmax(fma(x, cs[1][j], ch[1][j]), 0.0);
#pragma unroll
    for (int l = 1; l < LAYER; ++l) {
      double h[WX];
#pragma unroll
      for (int j = 0; j < WX; ++j) {
        double acc = (l == LAYER - 1) ? nshf[j] : 0.0;
#pragma unroll
        for (int i = 0; i < WX; ++i) acc = fma(W[l][j][i], a[i], acc);
        h[j] = acc;
      }
#pragma unroll
      for (int j = 0; j < WX; ++j) {
        if (l == LAYER - 1) t[j] = h[j];
        else a[j] = fmax(fma(h[j], cs[l + 1][j], ch[l + 1][j]), 0.0);
      }
    }
  };

  double nshf[WX];
  {                                                               // shift = the value of the wave's first sample
    double zero[WX], t0[WX];
#pragma unroll
    for (int j = 0; j < WX; ++j) zero[j] = 0.0;
    stack(x_first, zero, t0);
#pragma unroll
    for (int j = 0; j < WX; ++j) nshf[j] = -t0[j];
    if (lane == 0) {
#pragma unroll
      for (int j = 0; j < WM; ++j) wshift[wave][j] = j < WX ? t0[j < WX ? j : 0] : 0.0;
    }
  }
  double S1[WM], S2[WM];
#pragma unroll
  for (int j = 0; j < WM; ++j) S1[j] = S2[j] = 0.0;
  double cnt = 0.0;
  // whole tiles, no guards (U loads in flight per wave, requested one trip ahead; the first U above the prologue)
  while (full) {
    double x[U];
#if NIS_STATS_PIN
#pragma unroll
    for (int u = 0; u < U; ++u) {                                // keep the conversions (= the wait) at the top of the trip
      if (!ROWS) asm volatile("" : "+v"(xr[u].d));
      else if (PLN) asm volatile("" : "+v"(xr[u].f));
      else asm volatile("" : "+v"(xr[u].r));
    }
#endif
#pragma unroll
    for (int u = 0; u < U; ++u) x[u] = to_x(xr[u]);
    const long nbase = wbase + U * stride;
    const bool fulln = nbase + (U - 1) * stride + NIS_WAVE <= N;
    const long pbase = fulln ? nbase : wbase;                    // last round: re-request the current tiles (unused)
#pragma unroll
    for (int u = 0; u < U; ++u) xr[u] = load_raw(pbase + u * stride, lane);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      double t[WX];
      stack(x[u], nshf, t);
#pragma unroll
      for (int j = 0; j < WX; ++j) {
        S1[j] += t[j];
        S2[j] = fma(t[j], t[j], S2[j]);
      }
    }
    cnt += (double)U;
    wbase = nbase;
    full = fulln;
  }
  for (; wbase < N; wbase += stride) {                                         // the ragged rest, one tile at a time
    const bool valid = wbase + lane < N;
    double t[WX];
    stack(valid ? sample(wbase, lane) : 0.0, nshf, t);
#pragma unroll
    for (int j = 0; j < WX; ++j) {
      const double tv = valid ? t[j] : 0.0;
      S1[j] += tv;
      S2[j] = fma(tv, tv, S2[j]);
    }
    cnt += valid ? 1.0 : 0.0;
  }
  emit_records<WM, NW>(S1, S2, cnt, wshift, wmom, WM, c.partials + (((size_t)LAYER * c.G + g) * c.maxb + blockIdx.x) * WM * 3);
}

// ---------------------------------------------------------------------------------------------
// BN_0 records WITH the co-moments of the pass-through block (2 <= P <= 8): (n, mean_a, M2_a) per feature as k_bn_stats writes
// them, and behind them C_ab = sum (x_a - mean_a)(x_b - mean_b) for a < b (slot 3 P + cellk_pair(P, a, b)).  BN_1's batch
// statistics follow from these (load_coefs, CellK::dcov), so the layer-1 statistics launch of a matrix-core cell -- a full
// launch with its prologue for 8 MFMAs per 32 samples -- is gone.  Sums are taken about the workgroup's first sample.
// Workgroup reduction + record store shared by every producer of such records (k_bn_stats0_cov, the in-kernel-draw ingest): every
// thread holds sums about the per-feature shifts lsh[p] -- S1[a] = sum (x_a - lsh_a), S2[tri(a, b)] = sum (x_a - lsh_a)(x_b - lsh_b).
#define NIS_COV_PM 8
#define NIS_COV_NS2 (NIS_COV_PM * (NIS_COV_PM + 1) / 2)
template <int TPB>
__device__ __forceinline__ void cov_records_emit(const double (&S1)[NIS_COV_PM], const double (&S2)[NIS_COV_NS2], double cnt, const double* lsh,
                                                 int P, double* out) {
  constexpr int PM = NIS_COV_PM, NS2 = NIS_COV_NS2, NV = PM + NS2 + 1, NW = TPB / 64;
  __shared__ double wsum[NW][NV];
  __shared__ double tot[NV];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int p = 0; p < PM; ++p) {
    const double v = wave_sum(S1[p]);
    if (lane == 0) wsum[wave][p] = v;
  }
#pragma unroll
  for (int k = 0; k < NS2; ++k) {
    const double v = wave_sum(S2[k]);
    if (lane == 0) wsum[wave][PM + k] = v;
  }
  {
    const double v = wave_sum(cnt);
    if (lane == 0) wsum[wave][NV - 1] = v;
  }
  __syncthreads();
  if (tid < NV) {
    double v = 0.0;
    for (int w = 0; w < NW; ++w) v += wsum[w][tid];
    tot[tid] = v;
  }
  __syncthreads();
  const double n = tot[NV - 1];
  auto s2at = [&](int a, int b) { return tot[PM + a * PM - a * (a - 1) / 2 + (b - a)]; };      // a <= b
  if (tid < P) {
    const double s1 = tot[tid];
    out[tid * 3 + 0] = n;
    out[tid * 3 + 1] = n > 0.0 ? lsh[tid] + s1 / n : 0.0;
    out[tid * 3 + 2] = n > 0.0 ? fmax(s2at(tid, tid) - s1 * s1 / n, 0.0) : 0.0;
  }
  const int npair = P * (P - 1) / 2;
  if (tid >= 64 && tid < 64 + npair) {
    int m = tid - 64, pa = 0, pb = 1;
    for (int a = 0; a < P - 1; ++a) {
      const int row = P - 1 - a;
      if (m < row) { pa = a; pb = a + 1 + m; break; }
      m -= row;
    }
    out[3 * P + (tid - 64)] = n > 0.0 ? s2at(pa, pb) - tot[pa] * tot[pb] / n : 0.0;
  }
}

template <int WM, int TPB>
__global__ __launch_bounds__(TPB) void k_bn_stats0_cov(CellK c, const double* __restrict__ planes, long ld, long N) {
  constexpr int PM = NIS_COV_PM, NS2 = NIS_COV_NS2;
  __shared__ double lsh[PM];
  const int tid = threadIdx.x;
  const int g = c.g0 + blockIdx.y, P = c.P;
  planes += (size_t)g * c.plane_gs;
  const long i_blk = (long)blockIdx.x * TPB;
  double sh[PM];
#pragma unroll
  for (int p = 0; p < PM; ++p) sh[p] = (p < P && i_blk < N) ? planes[(size_t)c.in_plane[p] * ld + i_blk] : 0.0;
  if (tid == 0) {
#pragma unroll
    for (int p = 0; p < PM; ++p) lsh[p] = sh[p];
  }
  double S1[PM], S2[NS2];
#pragma unroll
  for (int p = 0; p < PM; ++p) S1[p] = 0.0;
#pragma unroll
  for (int k = 0; k < NS2; ++k) S2[k] = 0.0;
  double cnt = 0.0;
  const long stride = (long)gridDim.x * TPB;
  // U samples per thread and trip: U x P loads in flight (the pass is a latency-bound stream: 54 -> 1e6 samples took 54 us with one)
  constexpr int U = 4;
  const double* pp[PM];
#pragma unroll
  for (int p = 0; p < PM; ++p) pp[p] = planes + (size_t)c.in_plane[p < P ? p : 0] * ld;
  for (long i0 = i_blk + tid; i0 < N; i0 += stride * U) {
    double x[U][PM];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long i = i0 + u * stride;
#pragma unroll
      for (int p = 0; p < PM; ++p) x[u][p] = (p < P && i < N) ? pp[p][i] : sh[p];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long i = i0 + u * stride;
      double xs[PM];
#pragma unroll
      for (int p = 0; p < PM; ++p) xs[p] = x[u][p] - sh[p];                 // missing samples / padding columns: exactly 0
      int k = 0;
#pragma unroll
      for (int a = 0; a < PM; ++a) {
        S1[a] += xs[a];
#pragma unroll
        for (int b = a; b < PM; ++b, ++k) S2[k] = fma(xs[a], xs[b], S2[k]);
      }
      cnt += i < N ? 1.0 : 0.0;
    }
  }
  __syncthreads();                                  // lsh
  cov_records_emit<TPB>(S1, S2, cnt, lsh, P, c.partials + (((size_t)0 * c.G + g) * c.maxb + blockIdx.x) * WM * 3);
}

// ---------------------------------------------------------------------------------------------
// z = bias[row] + <wlast[row,:], a>
template <int WM, int WX = WM>
__device__ __forceinline__ double last_dot(nis_cptr wl, nis_cptr bias, int row, const double (&a)[WM]) {
  double acc = bias[row];
  nis_cptr w = wl + (size_t)row * WM;
#pragma unroll
  for (int i = 0; i < WX; ++i) acc = fma(w[i], a[i], acc);
  return acc;
}

// A wave-uniform double as a scalar value (two v_readfirstlane): lets uniform arithmetic done once on the vector pipe
// feed v_fma_f64 as an SGPR operand afterwards.
__device__ __forceinline__ double nis_uniform(double v) {
  const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v)), hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
  return __hiloint2double(hi, lo);
}

// The opposite: pin a (uniform) double into a VGPR pair.  The K = 4 SPEC kernels hold 21 differenced last-layer weights; as scalars
// they overflow the SGPR file and the compiler parks 16 of them in VGPR lanes, restoring them with 32 v_readlane (+ hazard nops)
// per sample -- 8 % of the loop's vector instructions.  As vector operands they cost registers (three waves per SIMD instead of
// four, measured neutral) and no instructions.
__device__ __forceinline__ double nis_vector(double v) {
  asm volatile("" : "+v"(v));
  return v;
}
#ifndef NIS_WD_VROWS
#define NIS_WD_VROWS 7        // of the 7 live differenced rows, how many live in VGPRs (7: no SGPR spill left in the loop)
#endif

// torch.lerp(start, end, weight) as the CPU kernel evaluates it (used at coupling_cells.py:224)
__device__ __forceinline__ double lerp_torch(double a, double b, double w) {
  const double d = b - a;
  const double lo = a + w * d, hi = b - d * (1.0 - w);      // both sides, one select: no exec-masked branch
  return (w < 0.5) ? lo : hi;
}

// Bin of x among the normalised cumulative widths, with the reference's semantics (coupling_cells.py:198-202):
//   index = argmax([1e-30, c_0, .., c_{K-1}]) (first maximum),  c_k = Wsum_k if Wsum_k <= x else 0,  Wsum_k = fl(run_k / wn),
// run_k the sequential cumsum of the raw widths.  Wsum is non-decreasing, so the kept values form a prefix and the
// maximum is its last element.  "Wsum_k <= x" is decided WITHOUT a division whenever run_k is outside a 2^-50 relative
// band around fl(x * wn) (then the correctly rounded quotient provably falls on the same side of x); inside the band
// -- probability ~1e-15 per bin -- the true division decides.  One division yields the selected left edge.
// Ties (equal Wsum for consecutive bins, i.e. widths that vanish in the cumsum) resolve to the first bin, as argmax does.
// the search with the reference's own comparison (one division per bin), for the columns the multiplied-out tests cannot decide
// *amb (per lane): the decision rests on last bits -- an edge quotient within 2^-50 (relative) of x, or, at the end of the kept prefix, a
// width on the threshold of vanishing in the running sum (2^-55 .. 2^-51 of it; a width below half an ulp, 2^-54 .. 2^-53, leaves the sum unchanged: whether two edges are EQUAL, where the reference's argmax
// takes the first, then turns on the last bit of exp()).  These are the samples on which the reference's own CPU and GPU builds disagree;
// everywhere else the selected bin is the reference's, bit for bit.  Surfaced as NIS_STATUS_BIN_AMBIGUOUS / NIS_BIN_AMBIGUOUS_FLAG.
template <int KM>
__device__ __forceinline__ void pwquad_bin_divide(double xb, int K, const double (&W)[KM], double wn, int* idx_out, double* run_out, bool* amb) {
  double run = 0.0, bestrun = 0.0;
  int idx = 0;
  bool near_x = false, zone = false;
#pragma unroll 1
  for (int k = 0; k < K; ++k) {
    double wk = W[0];
#pragma unroll
    for (int u = 1; u < KM; ++u) wk = (u == k) ? W[u] : wk;
    run += wk;
    const double quot = run / wn;
    const bool le = !(quot > xb);                        // the reference's comparison, verbatim
    const bool up = le && (run > bestrun);
    near_x |= k < K - 1 && fabs(quot - xb) <= xb * 0x1p-50;      // (the last edge is 1: a clamped x is 1e-6 below it)
    const bool z = k > 0 && wk > run * 0x1p-55 && !(wk > run * 0x1p-51);
    zone = le ? (up ? z : (zone || z)) : zone;
    bestrun = up ? run : bestrun;
    idx = up ? k + 1 : idx;
  }
  *idx_out = idx;
  *run_out = bestrun;
  *amb = near_x || zone;
}

// the division-free part of the search: bin and raw left edge from the multiplied-out tests; returns true when THIS lane's decision
// needs the search with the divisions (an edge inside the 2^-50 band around x * wn, or a width at or below 2^-50 of the TOTAL -- a
// superset of "of the running sum at its edge" -- that may tie two edges; one v_min per edge on this path)
template <int KM, bool KEXACT>
__device__ __forceinline__ bool pwquad_bin_fast(double xb, int Krt, const double (&W)[KM], double wn, int* idx_out, double* run_out) {
  const int K = KEXACT ? KM : Krt;
  const double q = xb * wn;
  const double qlo = q * (1.0 - 0x1p-50), qhi = q * (1.0 + 0x1p-50);
  double run = 0.0, bestrun = 0.0;
  int idx = 0;
  bool ambany = false;
  double wmin = wn;
  // the last edge is wn itself (same additions in the same order as the caller's sum): with x <= 1 - 1e-6 (every caller clamps, :167) it is
  // neither at or below q (1 - 2^-50) nor inside the band, so K - 1 edges are examined
#pragma unroll
  for (int k = 0; k < KM - 1; ++k)
    if (KEXACT || k < K - 1) {
      run += W[k];
      if (k > 0) wmin = fmin(wmin, W[k]);
      const bool le = run <= qlo;
      ambany |= (run > qlo) && (run < qhi);
      const bool up = le && (run > bestrun);
      bestrun = up ? run : bestrun;
      idx = up ? k + 1 : idx;
    }
  ambany |= !(wmin > wn * 0x1p-50);
  *idx_out = idx;
  *run_out = bestrun;
  return ambany;
}
// Wsum2[idx] = bestrun / wn (:186, :198) must exceed the 1e-30 sentinel, else bin 0 with left edge 0 wins; tested on the raw sum
// (equivalent unless the quotient is within an ulp of 1e-30); then the clamp of the index
__device__ __forceinline__ void pwquad_bin_finish(int K, double wn, int* idx_io, double* run_io) {
  const bool none = !(*run_io > 1e-30 * wn);
  int idx = none ? 0 : *idx_io;
  *run_io = none ? 0.0 : *run_io;
  *idx_io = (idx > K - 1) ? K - 1 : idx;
}
template <int KM, bool KEXACT>
__device__ __forceinline__ void pwquad_bin_raw(double xb, int Krt, const double (&W)[KM], double wn, int* idx_out, double* run_out, bool* amb_out = nullptr) {
  const int K = KEXACT ? KM : Krt;
  int idx;
  double bestrun;
  const bool ambany = pwquad_bin_fast<KM, KEXACT>(xb, Krt, W, wn, &idx, &bestrun);
  bool amb = false;
  if (__any(ambany)) pwquad_bin_divide<KM>(xb, K, W, wn, &idx, &bestrun, &amb);      // one wave-uniform test per column; practically never taken
  pwquad_bin_finish(K, wn, &idx, &bestrun);
  *idx_out = idx;
  *run_out = bestrun;                                    // raw cumulative width left of the bin
  if (amb_out) *amb_out = amb;
}

// same, returning the normalised left edge Wsum2[idx] (used by the adjoint kernels)
template <int KM, bool KEXACT>
__device__ __forceinline__ void pwquad_bin(double xb, int Krt, const double (&W)[KM], double wn, int* idx_out, double* left_out) {
  double run;
  pwquad_bin_raw<KM, KEXACT>(xb, Krt, W, wn, idx_out, &run);
  *left_out = run / wn;
}

// Piecewise-quadratic transform of one column (coupling_cells.py:167-225).  V: K+1 vertex heights
// (raw exp), W: K bin widths (raw exp).  Returns y; *jf = d y / d x; *bin = bin index.
// The bin is selected with the reference's exact semantics (pwquad_bin_raw).  The value is formed from RAW sums: with
// S_k = (V_k + V_{k+1}) W_k, A = sum_k S_k and wn = sum_k W_k the reference's normalisations (:184-197) amount to
// widths W/wn and heights V 2wn/A, hence
//   alpha = (x wn - run_b) / W_b,   y = ((alpha^2/2 (V_{b+1} - V_b) + alpha V_b) 2 W_b + sum_{k<b} S_k) / A,
//   dy/dx = lerp(V_b, V_{b+1}, alpha) 2 wn / A
// -- two divisions per column instead of four, no normalised copies of V and W (<= few ulp from the reference's order).
// `gather` (optional): this thread's private LDS column, element e at gather[e * NIS_TPB]: V and W are parked there and the
// bin's three values are read back by index (LDS pipe) instead of 3 K selects on the VALU pipe the kernel is bound by.
template <int KM, bool KEXACT>
__device__ __forceinline__ double pwquad_column(double xb, int Krt, const double (&V)[KM + 1], const double (&W)[KM], double* jf, int* bin,
                                                double* gather = nullptr, bool* amb = nullptr) {
  const int K = KEXACT ? KM : Krt;
  xb = (xb > 1.0 - 1e-6) ? 1.0 - 1e-6 : xb;                       // :167
  double wn = 0.0;
#pragma unroll
  for (int k = 0; k < KM; ++k)
    if (k < K) wn += W[k];                                        // Wsum[-1]  :180-182
  if (gather) {
#pragma unroll
    for (int k = 0; k < KM + 1; ++k)
      if (k <= K) gather[k * NIS_TPB] = V[k];
#pragma unroll
    for (int k = 0; k < KM; ++k)
      if (k < K) gather[(KM + 1 + k) * NIS_TPB] = W[k];
  }
  int idx;
  double run;
  pwquad_bin_raw<KM, KEXACT>(xb, Krt, W, wn, &idx, &run, amb);
  double A = 0.0, pre = 0.0, Wb = 1.0, Vb = 0.0, Vb1 = 0.0;
  if (gather) {
#pragma unroll
    for (int k = 0; k < KM; ++k)
      if (k < K) {
        pre = (k == idx) ? A : pre;
        A = fma(V[k] + V[k + 1], W[k], A);
      }
    Vb = gather[idx * NIS_TPB];
    Vb1 = gather[(idx + 1) * NIS_TPB];
    Wb = gather[(KM + 1 + idx) * NIS_TPB];
  } else {
#pragma unroll
  for (int k = 0; k < KM; ++k)
    if (k < K) {
      const bool hit = k == idx;
      pre = hit ? A : pre;                                        // VW[idx] A  :209-211
      Wb = hit ? W[k] : Wb;
      Vb = hit ? V[k] : Vb;
      Vb1 = hit ? V[k + 1] : Vb1;
      A = fma(V[k] + V[k + 1], W[k], A);                          // :194
    }
  }
  // one reciprocal for both divisions: r = 1 / (W_b A)  ->  1/A = r W_b,  1/W_b = r A   (A and W_b are sums of exps: normal range)
  const double r = nis_rcp(Wb * A);
  const double iA = r * Wb;
  const double alpha = (xb * wn - run) * (r * A);                 // :206-207
  const double num = ((alpha * alpha) / 2 * (Vb1 - Vb) + alpha * Vb) * (2 * Wb) + pre;   // :213-220
  *jf = lerp_torch(Vb, Vb1, alpha) * ((2 * wn) * iA);             // :224-225
  *bin = idx;
  return num * iA;
}

// More than 32 bins (up to NIS_MAX_BINS): the same transform with V (K + 1 raw heights) and W (K raw widths) in per-thread arrays
// the compiler keeps in scratch memory, every loop over the bins a runtime loop.  Bin selection is pwquad_bin_raw's, statement by
// statement (guard band, exact re-test with the reference's division when a lane falls inside it, sentinel, clamp).
// (ST: element stride of W -- 1 for a scratch array, 16 for a column of the matrix-core adjoint's LDS ring)
template <int ST = 1>
__device__ __forceinline__ void pwquad_bin_big(double xb, int K, const double* W, double wn, int* idx_out, double* run_out, bool* amb_out = nullptr) {
  const double q = xb * wn;
  const double qlo = q * (1.0 - 0x1p-50), qhi = q * (1.0 + 0x1p-50);
  double run = 0.0, bestrun = 0.0, wmin = wn;
  int idx = 0;
  bool ambany = false;
  for (int k = 0; k < K; ++k) {
    const double wk = W[k * ST];
    run += wk;
    wmin = k > 0 ? fmin(wmin, wk) : wmin;
    const bool le = run <= qlo;
    ambany |= (run > qlo) && (run < qhi);
    const bool up = le && (run > bestrun);
    bestrun = up ? run : bestrun;
    idx = up ? k + 1 : idx;
  }
  ambany |= !(wmin > wn * 0x1p-50);                      // a width that may tie two edges (see pwquad_bin_raw)
  bool amb = false;
  if (__any(ambany)) {                                   // pwquad_bin_divide, on the strided array
    run = 0.0;
    bestrun = 0.0;
    idx = 0;
    bool near_x = false, zone = false;
    for (int k = 0; k < K; ++k) {
      const double wk = W[k * ST];
      run += wk;
      const double quot = run / wn;
      const bool le = !(quot > xb);                      // the reference's comparison, verbatim
      const bool up = le && (run > bestrun);
      near_x |= k < K - 1 && fabs(quot - xb) <= xb * 0x1p-50;
      const bool z = k > 0 && wk > run * 0x1p-55 && !(wk > run * 0x1p-51);
      zone = le ? (up ? z : (zone || z)) : zone;
      bestrun = up ? run : bestrun;
      idx = up ? k + 1 : idx;
    }
    amb = near_x || zone;
  }
  const bool none = !(bestrun > 1e-30 * wn);
  idx = none ? 0 : idx;
  bestrun = none ? 0.0 : bestrun;
  idx = (idx > K - 1) ? K - 1 : idx;
  *idx_out = idx;
  *run_out = bestrun;                                    // raw cumulative width left of the bin
  if (amb_out) *amb_out = amb;
}
__device__ __forceinline__ double pwquad_column_big(double xb, int K, const double* V, const double* W, double* jf, int* bin, bool* amb = nullptr) {
  xb = (xb > 1.0 - 1e-6) ? 1.0 - 1e-6 : xb;                       // :167
  double wn = 0.0;
  for (int k = 0; k < K; ++k) wn += W[k];                         // Wsum[-1]  :180-182
  int idx;
  double bestrun;
  pwquad_bin_big(xb, K, W, wn, &idx, &bestrun, amb);
  double A = 0.0, pre = 0.0;
  for (int k = 0; k < K; ++k) {
    pre = (k == idx) ? A : pre;                                   // VW[idx] A  :209-211
    A = fma(V[k] + V[k + 1], W[k], A);                            // :194
  }
  const double Vb = V[idx], Vb1 = V[idx + 1], Wb = W[idx];
  const double r = nis_rcp(Wb * A);                               // one reciprocal for both divisions (see pwquad_column)
  const double iA = r * Wb;
  const double alpha = (xb * wn - bestrun) * (r * A);             // :206-207
  const double num = ((alpha * alpha) / 2 * (Vb1 - Vb) + alpha * Vb) * (2 * Wb) + pre;   // :213-220
  *jf = lerp_torch(Vb, Vb1, alpha) * ((2 * wn) * iA);             // :224-225
  *bin = idx;
  return num * iA;
}

// Same transform with the raw heights / widths in an LDS image (element k of the column at zr[k * ZS]: rows 0..K heights,
// K+1..2K widths), as the matrix-core kernels keep them: the widths are streamed for the bin search, the trapezoid sums
// stream the heights, and V_b, V_{b+1}, W_b are read BY INDEX instead of through K-long select chains.
// REL: the relative image (nis_common.h, NIS_PACK_REL): rows 0..K-1 = V[k] / V[K], rows K..2K-2 = W[k] / W[K-1], row 2K-1 unused; V[K] = W[K-1] = 1.
template <int KM, bool KEXACT, int ZS, bool REL = false>
__device__ __forceinline__ double pwquad_column_lds(double xb, int Krt, const double* zr, double* jf, int* bin, bool* amb_out = nullptr) {
  static_assert(!REL || KEXACT, "the relative image exists for K == KM only");
  const int K = KEXACT ? KM : Krt;
  const int WO = REL ? K : K + 1;                                 // first width row
  xb = (xb > 1.0 - 1e-6) ? 1.0 - 1e-6 : xb;                       // :167
  double W[KM];
  double wn = 0.0;
#pragma unroll
  for (int k = 0; k < KM; ++k) {
    W[k] = (REL && k == KM - 1) ? 1.0 : (KEXACT || k < K) ? zr[(WO + k) * ZS] : 0.0;
    if (KEXACT || k < K) wn += W[k];                              // Wsum[-1]  :180-182
  }
  // Bin search.  The cumulative widths never decrease, so the bin is the NUMBER of edges at or below the lower end of the band around
  // x * wn (pwquad_bin_raw: 2^-50 relative), and only the two edges next to it can make the multiplied-out test differ from the
  // reference's divisions: the bin's right edge inside the band, or a width that vanished in the sum right below it (the reference's argmax
  // takes the FIRST of equal edges).  Both are checked on W[idx], W[idx - 1] read back by index; a wave with such a column -- probability
  // ~1e-15 per bin -- runs the search with the divisions.  5 vector instructions per bin instead of 11.
  int idx = 0;
  double run = 0.0;
  {
    const double q = xb * wn;
    const double qlo = q * (1.0 - 0x1p-50), qhi = q * (1.0 + 0x1p-50);
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < KM - 1; ++k)                                // (the last edge is wn: never at or below q, see pwquad_bin_raw)
      if (KEXACT || k < K - 1) {
        acc += W[k];
        const bool le = acc <= qlo;
        run = le ? acc : run;
        idx += le ? 1 : 0;
      }
    const int ir = idx > K - 1 ? K - 1 : idx, il = idx > 0 ? idx - 1 : 0;
    const double wr = (REL && ir == K - 1) ? 1.0 : zr[(WO + (REL && ir == K - 1 ? 0 : ir)) * ZS];
    const double wl = (REL && il == K - 1) ? 1.0 : zr[(WO + (REL && il == K - 1 ? 0 : il)) * ZS];
    const bool amb = idx < K && run + wr < qhi;                    // (run + W[idx] IS the next edge of the loop above, bit for bit)
    const bool tie = idx > 0 && !(wl > run * 0x1p-50);             // W[idx - 1] may have vanished in the sum (equal edges) or be on the threshold
    bool flagged = false;
    if (__any(amb || tie)) pwquad_bin_divide<KM>(xb, K, W, wn, &idx, &run, &flagged);
    if (amb_out) *amb_out = flagged;
    const bool none = !(run > 1e-30 * wn);                         // :186, :198 (see pwquad_bin_raw)
    idx = none ? 0 : idx;
    run = none ? 0.0 : run;
    idx = (idx > K - 1) ? K - 1 : idx;
  }
  double A = 0.0, pre = 0.0, vk = zr[0];
#pragma unroll
  for (int k = 0; k < KM; ++k)
    if (KEXACT || k < K) {
      const double vn = (REL && k == KM - 1) ? 1.0 : zr[(k + 1) * ZS];
      pre = (k == idx) ? A : pre;                                 // VW[idx] A  :209-211
      A = fma(vk + vn, W[k], A);                                  // :194
      vk = vn;
    }
  double Vb = zr[idx * ZS], Vb1, Wb;
  if (REL) {                                                      // the two constant rows are not in the image
    const bool lastb = idx == K - 1;
    Vb1 = zr[(lastb ? idx : idx + 1) * ZS];
    Wb = zr[(WO + (lastb ? 0 : idx)) * ZS];
    Vb1 = lastb ? 1.0 : Vb1;
    Wb = lastb ? 1.0 : Wb;
  } else {
    Vb1 = zr[(idx + 1) * ZS];
    Wb = zr[(WO + idx) * ZS];
  }
  const double r = nis_rcp(Wb * A);                               // one reciprocal for both divisions (see pwquad_column)
  const double iA = r * Wb;
  const double alpha = (xb * wn - run) * (r * A);                 // :206-207
  const double num = ((alpha * alpha) / 2 * (Vb1 - Vb) + alpha * Vb) * (2 * Wb) + pre;   // :213-220
  *jf = lerp_torch(Vb, Vb1, alpha) * ((2 * wn) * iA);             // :224-225
  *bin = idx;
  return num * iA;
}

// Full cell: conditioner with finalised BN + bijector, in place on the transformed planes.  Optional riders:
//   * BN_0 records of the NEXT cell (its pass-through columns are all in this thread's registers),
//   * a built-in integrand f(X)*J with its moment records (last cell), optionally without storing X, J.
// SPEC != 0: compile-time copies of the launch's wave-uniform switches for the two cells of a d = 2 chain (P = T = 1,
// folded conditioner input, in place, no bin output): 1 = inner cell (BN_0 rider of the next cell, planes written),
// 2 = last cell with the camel integrand fused and nothing stored.  The generic kernel (SPEC 0) carries every switch
// as a scalar branch; resolving them removes ~60 branches, the SGPR spills and their v_readlane traffic from the loop.
// ROWS (SPEC kernels only): the transformed input -- and, for c.x_wcol >= 0, the pass-through input too -- is read from the
// float32 (N, 2) rows of w (CellK::wrows) instead of its plane: one 8-byte row load per sample.
// FIRST (ROWS kernels): the flow's first cell -- both inputs come from the row, the incoming Jacobian is 1: ONE load per sample.
// BINS (SPEC kernels, test instantiation): the selected bin indices are written to bins_out although the production SPEC
// kernels never output them -- tests assert them bit-exact against the oracle on the very code path the bench runs.
// SPEC 3 / 4 (plane input): the same cells in the TRAINING forward -- outputs and the pass-through column go to the next plane set
// (CellK::out_off, the adjoint's history); 3 carries the next cell's BN_0 rider like SPEC 1, 4 is the last cell (no rider, no integrand).
// PLN (ROWS kernels): w is two float32 column planes (CellK::wplanar): the first cell takes one float from each, a later cell one
// float of its transformed column.
#ifndef NIS_SPEC_WAVES
#define NIS_SPEC_WAVES 1      // 4 = allocate the K = 4 SPEC kernels for four waves per SIMD (<= 128 VGPRs): measured 1 % SLOWER than the 129-VGPR / three-wave last cell
#endif
template <int KIND, int WM, int KM, bool KEXACT, int WX = WM, int SPEC = 0, bool ROWS = false, bool FIRST = false, bool BINS = false, bool PLN = false>
__global__ __launch_bounds__(NIS_TPB, (SPEC != 0 && KM <= 4) ? NIS_SPEC_WAVES : 1) void k_cell_apply(CellK c, double* __restrict__ planes, long ld, long N, int bn_train,
                                                        int* __restrict__ bins_out, FuseF fz) {
  static_assert(!ROWS || SPEC != 0, "row input exists on the SPEC kernels only");
  typedef float nis_f2 __attribute__((ext_vector_type(2)));
  constexpr int NW = NIS_TPB / NIS_WAVE;
  constexpr int NF = SPEC ? 1 : WM <= 4 ? 4 : NIS_PMAX_FUSED;      // next-cell pass-through features tracked here (<= its P)
  constexpr bool SP = SPEC != 0;
  __shared__ double lds_coef[(NIS_MAX_HIDDEN + 1) * 2 * WM];
  __shared__ double red[3 * NIS_TPB];
  __shared__ double wshift[NW][NF];
  __shared__ double wmom[NW][NF][3];
  __shared__ double fmom[NW][4];
  __shared__ double etab[NIS_EXP_N];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = c.g0 + blockIdx.y;
  planes += (size_t)g * c.plane_gs;
  nis_exp_init(etab);
  const double w_fold = fold_weight<WM>(c);
  // SPEC kernels keep the inputs of the NEXT sample in flight while the current one is computed (one sample per thread
  // and iteration: ~0.7 us of arithmetic against ~2 us of HBM latency, 3-4 waves per SIMD); the first request goes out
  // inside the prologue.  jsrc: a first cell has no Jacobian plane to read -- it re-reads x (same line, L1 hit) so that
  // the loop's vector-memory sequence is fixed and the compiler's vmcnt bookkeeping stays exact.
  static_assert(!FIRST || ROWS, "FIRST is a property of the row-input kernels");
  const nis_f2* const rows = ROWS ? reinterpret_cast<const nis_f2*>(c.wrows + (size_t)g * c.wrows_gs) : nullptr;
  constexpr bool xrow = ROWS && FIRST;                             // pass-through input from the row as well
  const bool rswap = ROWS && !PLN && (FIRST ? c.x_wcol == 1 : c.xb_wcol == 0);   // FIRST: (x, xb) = (r.y, r.x);  later cells: xb = r.x
  // PLN: rq is assembled as (pass-through, transformed) from the two float planes, no swap
  const float* const wpx = (ROWS && PLN && FIRST) ? c.wrows + (size_t)g * c.wrows_gs + (size_t)c.x_wcol * c.wplane_ld : nullptr;
  const float* const wpb = (ROWS && PLN) ? c.wrows + (size_t)g * c.wrows_gs + (size_t)c.xb_wcol * c.wplane_ld : nullptr;
  auto row_at = [&](long i) -> nis_f2 {
    if (!PLN) return rows[i];
    return nis_f2{FIRST ? wpx[i] : 0.f, wpb[i]};
  };
  const double* const P0 = planes + (size_t)c.in_plane[0] * ld;
  double* const P1 = planes + (size_t)c.in_plane[SP ? 1 : 0] * ld;
  // plane kernels: a first cell has no Jacobian plane to read -- it re-reads x (same line, L1 hit) so that the loop's
  // vector-memory sequence is fixed and the compiler's vmcnt bookkeeping stays exact
  const double* const jsrc = (!ROWS && c.first) ? P0 : planes + (size_t)c.jplane * ld;
  const long i_first = (long)blockIdx.x * NIS_TPB + tid;
  double xq = 0.0, xbq = 0.0, jq = 0.0;
  nis_f2 rq = {0.f, 0.f};
  auto first_loads = [&]() {
    if (SP && i_first < N) {
      if (!xrow) xq = P0[i_first];
      if (ROWS) rq = row_at(i_first);
      else xbq = P1[i_first];
      if (!FIRST) jq = jsrc[i_first];
    }
  };
  load_coefs<WM, NIS_TPB>(c, g, c.L + 1, (bn_train && c.pmerge) ? cellk_fresh(c, c.L + 1) : -1, blockIdx.x == 0, lds_coef, red, first_loads);
  fold_coefs<WM>(c, lds_coef, c.L + 1, w_fold);
  const bool fold = SP ? true : c.fold != 0;
  const int cP = SP ? 1 : c.P, cT = SP ? 1 : c.T;
  const long out_off = (SPEC == 0 || SPEC >= 3) ? c.out_off : 0;      // SPEC 3 / 4: the training forward's history (next plane set)
  int* const bins_o = SP ? (BINS ? bins_out : nullptr) : bins_out;
  nis_cptr wl = as_const(c.wlast);
  nis_cptr bias = as_const(c.bias);
  constexpr bool GATHER = SP && KM <= 8;    // (2 KM + 1) x 2 KB of LDS per workgroup: 18 KB at KM = 4, 34 KB at KM = 8
  __shared__ double lds_gather[GATHER ? (2 * KM + 1) * NIS_TPB : 1];   // per-thread column of V | W for the indexed read-back
  constexpr bool RIDERS = WM <= 8;          // wide conditioners keep their registers for the GEMV
  const bool next = (SPEC == 1 || SPEC == 3) ? true : (SPEC == 2 || SPEC == 4) ? false : (RIDERS && c.next_P > 0);
  const bool fuse = SPEC == 2 ? true : SPEC != 0 ? false : (RIDERS && fz.kind >= 0);
  const int fkind = SPEC == 2 ? (int)NIS_F_CAMEL : fz.kind;
  const bool wplanes = SPEC == 2 ? false : SPEC != 0 ? true : fz.write_planes != 0;
  double* const fres = SP ? nullptr : fz.fres;
  // SPEC, K = 4: the bijector only sees V up to a common factor and W up to a common factor (y and the Jacobian divide by
  // the area sum (V_k + V_k+1) W_k and by sum W), so the last height and the last width are taken as exp(0) = 1 and every
  // other logit is computed RELATIVE to them: rows k - K and K+1+k - 2K of the last layer, differenced once per workgroup
  // into scalar registers.  7 instead of 9 exps and dot products per sample; values move by a few 1e-16 (rounding only).
  constexpr bool SHIFT = SP && KIND == NIS_PWQUAD && KM == 4;
  constexpr int RS = SHIFT ? 2 * KM + 1 : 1;
  double wd[RS][WX], bd[RS];
  if (SHIFT) {
#pragma unroll
    for (int r = 0; r < RS; ++r) {
      const int ref = r <= KM ? KM : 2 * KM;
      bd[r] = bias[r] - bias[ref];                       // stays in a VGPR: start value of the FMA chain (an SGPR addend would need a v_mov)
#pragma unroll
      for (int i = 0; i < WX; ++i) {
        const double v = wl[r * WM + i] - wl[ref * WM + i];
        wd[r][i] = (r - (r > KM ? 1 : 0)) < NIS_WD_VROWS ? nis_vector(v) : nis_uniform(v);     // rows KM and 2 KM are the references (never used)
      }
    }
  }
  auto rel_exp = [&](int r, const double (&av)[WM]) -> double {
    double acc = bd[SHIFT ? r : 0];
#pragma unroll
    for (int i = 0; i < WX; ++i) acc = fma(wd[SHIFT ? r : 0][i], av[i], acc);
    return nis_exp(acc, etab);
  };
  double nS1[NF], nS2[NF], nsh[NF];
#pragma unroll
  for (int f = 0; f < NF; ++f) nS1[f] = nS2[f] = nsh[f] = 0.0;
  double cnt = 0.0, fS1 = 0.0, fS2 = 0.0, fmax_ = -__builtin_huge_val(), fshift = 0.0;
  bool firstit = true;
  const long stride = (long)gridDim.x * NIS_TPB;
  for (long i = i_first; i < N; i += stride) {
    double a[WM];
    // SPEC: this sample's inputs, requested one iteration ago
    const double x0 = xrow ? (double)(rswap ? rq.y : rq.x) : xq;
    const double xb0 = ROWS ? (double)(rswap ? rq.x : rq.y) : xbq;
    const double jin = ROWS ? (FIRST ? 1.0 : jq) : (c.first ? 1.0 : jq);
    if (SP) {
      const long ic = i + stride < N ? i + stride : i;              // last round: re-read the current sample (unused)
      if (!xrow) xq = P0[ic];
      if (ROWS) rq = row_at(ic);
      else xbq = P1[ic];
      if (!FIRST) jq = jsrc[ic];
#pragma unroll
      for (int p = 0; p < WM; ++p) a[p] = p == 0 ? x0 : 0.0;
    } else {
      load_passthrough<WM>(c, planes, ld, i, a);
    }
    if (out_off != 0) {                             // history mode: the next plane set gets the pass-through block too
#pragma unroll
      for (int p = 0; p < (WM < NIS_MAX_DIM ? WM : NIS_MAX_DIM); ++p)
        if (p < cP) planes[(size_t)c.in_plane[p] * ld + out_off + i] = a[p];
    }
    double nv[NF];
    double c0 = 0.0, c1 = 0.0, nsum = 0.0;
    // riders read pass-through columns straight from their planes (uniform plane index, L1 hit: the line was
    // just loaded for `a`); transformed columns are picked up inside the t loop by one uniform compare each
    if (next) {
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        const int src = c.next_src[f];
        if (SP) nv[f] = 0.0;                                        // SPEC 1: the next cell passes this cell's OUTPUT through (set below)
        else nv[f] = (src >= 0 && src < cP) ? planes[(size_t)c.in_plane[src < 0 ? 0 : src] * ld + i] : 0.0;
      }
    }
    if (fuse) {
      if (SP) {
        c0 = x0;                       // SPEC 2: the camel is symmetric in its two arguments (IEEE addition commutes), so the
      } else if (fkind == NIS_F_CAMEL) {   // pass-through input and the output y are taken in this fixed order
        if (fz.loc0 >= 0 && fz.loc0 < cP) c0 = planes[(size_t)c.in_plane[fz.loc0] * ld + i];
        if (fz.loc1 >= 0 && fz.loc1 < cP) c1 = planes[(size_t)c.in_plane[fz.loc1] * ld + i];
      } else if (fkind == NIS_F_NORMAL) {
#pragma unroll
        for (int p = 0; p < (WM < NIS_MAX_DIM ? WM : NIS_MAX_DIM); ++p)
          if (p < cP) nsum += ((a[p] - fz.p[0]) * (a[p] - fz.p[0])) / (2 * fz.p[1] * fz.p[1]);
      }
    }
    hidden_stack<WM, WX>(c, lds_coef, SP ? 3 : c.L, true, a, fold);      // SPEC: three hidden layers (launch_apply checks)
    double jprod = 1.0;
    for (int t = 0; t < cT; ++t) {
      double* col = SP ? P1 : planes + (size_t)c.in_plane[cP + t] * ld;
      const double xb = SP ? xb0 : col[i];
      const int row0 = t * c.out;
      double y, jf;
      int bin;
      bool amb = false;                                          // PWQuad: the bin decision rests on last bits (pwquad_bin_divide)
      if constexpr (KIND == NIS_PWQUAD && KM > 32) {             // 33 .. NIS_MAX_BINS bins: scratch arrays, runtime loops
        double V[NIS_KM_BIG + 1], W[NIS_KM_BIG];
        const int Kq = c.K;
        for (int k = 0; k <= Kq; ++k) V[k] = nis_exp(last_dot<WM, WX>(wl, bias, row0 + k, a), etab);                     // :173,189
        for (int k = 0; k < Kq; ++k) W[k] = nis_exp(last_dot<WM, WX>(wl, bias, row0 + Kq + 1 + k, a), etab);             // :175,178
        y = pwquad_column_big(xb, Kq, V, W, &jf, &bin, &amb);
      } else if (KIND == NIS_PWQUAD) {
        constexpr int KA = KM > 32 ? 1 : KM;                     // (the branch above took KM > 32)
        double V[KA + 1], W[KA];
        const int Kq = KEXACT ? KM : c.K;
        if (SHIFT) {
#pragma unroll
          for (int k = 0; k < KM + 1; ++k) V[k] = k < KM ? rel_exp(k, a) : 1.0;
#pragma unroll
          for (int k = 0; k < KM; ++k) W[k] = k < KM - 1 ? rel_exp(KM + 1 + k, a) : 1.0;
        } else {
#pragma unroll
          for (int k = 0; k < KM + 1; ++k)
            V[k] = (KEXACT || k <= c.K) ? nis_exp(last_dot<WM, WX>(wl, bias, row0 + k, a), etab) : 0.0;                 // :173,189
#pragma unroll
          for (int k = 0; k < KM; ++k)
            W[k] = (KEXACT || k < c.K) ? nis_exp(last_dot<WM, WX>(wl, bias, row0 + Kq + 1 + k, a), etab) : 0.0;           // :175,178
        }
        y = pwquad_column<KA, KEXACT>(xb, c.K, V, W, &jf, &bin, GATHER ? lds_gather + tid : nullptr, &amb);
      } else if (KIND == NIS_AFFINE) {
        // AffineCoupling (coupling_cells.py:56-68): rows [scale | shift][T];  y = atan(20 e^z0 x + relu(z1)) / (pi/2)
        const double s0 = 20.0 * nis_exp(last_dot<WM, WX>(wl, bias, t, a), etab);
        const double z1 = last_dot<WM, WX>(wl, bias, cT + t, a);
        const double u = xb * s0 + (z1 > 0.0 ? z1 : 0.0);
        y = atan(u) / (M_PI / 2);
        jf = s0 * (1.0 / (u * u + 1.0));
        bin = 0;
      } else {
        // PWLin: equal-width bins -> the index is known before the heights (coupling_cells.py:126-133),
        // so the K heights are streamed: total, prefix below the bin, and the bin's own height.
        const double ak = xb * c.K;
        const double bf = floor(ak);
        const double alpha = (ak - bf) / c.K;
        bin = (int)bf;
        double tot = 0.0, below = 0.0, qb = 0.0;
        for (int k = 0; k < c.K; ++k) {
          const double e = nis_exp(last_dot<WM, WX>(wl, bias, row0 + k, a), etab);                 // :115
          if (k == bin) {
            below = tot;
            qb = e;
          }
          tot += e;                                                                   // :117
        }
        if (bin < 0 || bin >= c.K) {             // the reference's torch.gather raises here (:136)
          if (c.status) atomicOr(c.status, 1);
          qb = __builtin_nan("");
        }
        const double qn = qb / (tot / c.K);                                           // :121
        y = qn * alpha + below / tot;                                                 // :123,136-139
        jf = qn;                                                                      // :141
      }
      if (wplanes) col[out_off + i] = y;
      jprod *= jf;
      if (KIND == NIS_PWQUAD && amb && c.status) {               // (rare: a whole wave skips this)
        atomicOr(c.status, NIS_STATUS_BIN_AMBIGUOUS);
        atomicOr(c.status + 1, NIS_STATUS_BIN_AMBIGUOUS);        // sticky copy: cleared by the caller, not by the next call's first launch
      }
      if (KIND != NIS_AFFINE && bins_o && g == 0) bins_o[(size_t)(c.bins_row + t) * ld + i] = bin | (amb ? NIS_BIN_AMBIGUOUS_FLAG : 0);
      if (next) {
        if (SP) nv[0] = y;             // launch_apply checked next_src[0] == 1
#pragma unroll
        for (int f = 0; f < NF; ++f)
          if (!SP && c.next_src[f] == cP + t) nv[f] = y;
      }
      if (fuse) {
        if (SP) c1 = y;                // launch_apply checked {loc0, loc1} == {0, 1}
        if (!SP && fz.loc0 == cP + t) c0 = y;
        if (!SP && fz.loc1 == cP + t) c1 = y;
        if (fkind == NIS_F_NORMAL) nsum += ((y - fz.p[0]) * (y - fz.p[0])) / (2 * fz.p[1] * fz.p[1]);
      }
    }
    double* jp = planes + (size_t)c.jplane * ld;
    if (KIND == NIS_AFFINE) jprod *= 1.0 / (M_PI / 2);          // the reference applies 2/pi ONCE per cell (:68)
    const double jout = (SP ? jin : c.first ? 1.0 : jp[i]) * jprod;
    if (wplanes) jp[out_off + i] = jout;
    if (next) {
      if (firstit) {
#pragma unroll
        for (int f = 0; f < NF; ++f) {
          nsh[f] = wave_first(nv[f]);
          if (lane == 0) wshift[wave][f] = nsh[f];
        }
      }
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        const double t = nv[f] - nsh[f];
        nS1[f] += t;
        nS2[f] = fma(t, t, nS2[f]);
      }
    }
    if (fuse) {
      double fv = 1.0;
      if (fkind == NIS_F_CAMEL) {                                                   // README.md:35-36
        const double is2 = fz.p[3];                                                   // host: 1 / s^2
        const double e1 = -((c0 - fz.p[0]) * (c0 - fz.p[0]) + (c1 - fz.p[0]) * (c1 - fz.p[0])) * is2;
        const double e2 = -((c0 - fz.p[1]) * (c0 - fz.p[1]) + (c1 - fz.p[1]) * (c1 - fz.p[1])) * is2;
        fv = nis_exp(e1, etab) + nis_exp(e2, etab);
      } else if (fkind == NIS_F_NORMAL) {                                           // manager.py:39-40
        fv = nis_exp(-nsum, etab) / (fz.p[1] * sqrt(pow(2 * M_PI, (double)c.d)));
      }
      const double v = fv * jout;
      if (fres) fres[(size_t)g * fz.fres_gs + i] = v;
      if (firstit) fshift = wave_first(v);
      const double t = v - fshift;
      fS1 += t;
      fS2 = fma(t, t, fS2);
      fmax_ = fmax(fmax_, v);
    }
    cnt += 1.0;
    firstit = false;
  }
  if (next)
    emit_records<NF, NW>(nS1, nS2, cnt, wshift, wmom, SP ? 1 : c.next_P, c.next_partials + ((size_t)g * c.maxb + blockIdx.x) * c.next_wm * 3);
  if (fuse) block_record<NW>(fshift, fS1, fS2, cnt, fmax_, fmom, fz.partials + ((size_t)g * fz.maxb + blockIdx.x) * 4);
}

// Host-side launchers, instantiated once per WM in flow_wm.hip.  `grid` is the x extent; y = c.G batches.
// *_blocks_per_cu: resident workgroups per CU of the kernel (occupancy query, cached) -> grids are sized to ONE
// resident wave of workgroups so that every thread amortises the prologue / epilogue over many samples.
template <int WM>
int stats_blocks_per_cu(int wx, bool folded);
template <int WM>
int apply_blocks_per_cu(int kind, int km, int wx);
template <int WM>
hipError_t launch_stats(const CellK& c, const double* planes, long ld, long N, int layer, int grid, hipStream_t s);
template <int WM>
hipError_t launch_apply(const CellK& c, double* planes, long ld, long N, int bn_train, int km, int* bins, const FuseF& fz, int grid,
                        hipStream_t s);
// matrix-core path (WM >= 16): weight fragments pre-packed per (row tile, k-step) as 64 contiguous doubles
struct MfmaW {
  const double* wm_all;      // L x (WM/16) x (WM/4) x 64
  const double* wm_last;     // RT x (WM/4) x 64,  RT = ceil(R/16)
  const double* bias_pad;    // RT x 16
};
template <int WM>
bool mfma_supported(const CellK& c);                        // LDS image of one column batch fits
template <int WM>
int mfma_blocks_per_cu(const CellK& c);
template <int WM>
hipError_t launch_stats_mfma(const CellK& c, const MfmaW& w, const double* planes, long ld, long N, int layer, int grid, hipStream_t s);
template <int WM>
hipError_t launch_apply_mfma(const CellK& c, const MfmaW& w, double* planes, long ld, long N, int bn_train, int km, int* bins, const FuseF& fz,
                             int grid, hipStream_t s);
// true when launch_apply_mfma has a fused-integrand instantiation for this cell (PWLin, AffineCoupling, PWQuad with K == KM)
static inline bool nis_mfma_fusable(int kind, int km) { return kind != NIS_PWQUAD || km < 0; }
// Row order of the last Linear's matrix-core image: 0 = reference order, NIS_PACK_REL = the relative image (nis_common.h).
// One function decides for the packer and the launcher.
static inline int nis_mf_regmode(int kind, int K, int T, int wm) {
  // PWQuad with K == KM on the matrix cores: the relative image (nis_common.h; -DNIS_MF_REL=0 builds the reference rows for A/B runs)
  if (NIS_MF_REL && wm >= 16 && kind == NIS_PWQUAD && (K == 4 || K == 8 || K == 16 || K == 32)) return NIS_PACK_REL;
  return 0;
}
#define NIS_DECL_WM(W)                                                                                                      \
  template <>                                                                                                               \
  int stats_blocks_per_cu<W>(int wx, bool folded);                                                                                          \
  template <>                                                                                                               \
  int apply_blocks_per_cu<W>(int kind, int km, int wx);                                                                          \
  template <>                                                                                                               \
  hipError_t launch_stats<W>(const CellK& c, const double* planes, long ld, long N, int layer, int grid, hipStream_t s);    \
  template <>                                                                                                               \
  hipError_t launch_apply<W>(const CellK& c, double* planes, long ld, long N, int bn_train, int km, int* bins, const FuseF& fz, \
                             int grid, hipStream_t s);                                                                      \
  template <>                                                                                                               \
  bool mfma_supported<W>(const CellK& c);                                                                                   \
  template <>                                                                                                               \
  int mfma_blocks_per_cu<W>(const CellK& c);                                                                                \
  template <>                                                                                                               \
  hipError_t launch_stats_mfma<W>(const CellK& c, const MfmaW& w, const double* planes, long ld, long N, int layer, int grid, \
                                  hipStream_t s);                                                                           \
  template <>                                                                                                               \
  hipError_t launch_apply_mfma<W>(const CellK& c, const MfmaW& w, double* planes, long ld, long N, int bn_train, int km, int* bins, \
                                  const FuseF& fz, int grid, hipStream_t s);
NIS_DECL_WM(4)
NIS_DECL_WM(8)
NIS_DECL_WM(16)
NIS_DECL_WM(32)
NIS_DECL_WM(64)
#undef NIS_DECL_WM
