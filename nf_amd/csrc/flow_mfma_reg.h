// Matrix-core apply kernel with a REGISTER-RESIDENT bijector (PWQuad with 16 bins, PWLin with a multiple of 16 bins).
//
// k_cell_apply_mfma (flow_mfma.h) moves the exp() values of a column batch through a per-wave LDS image so that ONE lane
// owns a (sample, column) pair and runs the sequential bijector: 20 KB of LDS per wave at K = 16 (two workgroups per CU), a
// 16-long dependent scan per lane and 220-250 VGPRs (two waves per SIMD).  Here the rows of the last Linear are PERMUTED at
// pack time (k_pack_mfma, MF_REG_*) so that the accumulator registers a lane receives from v_mfma_f64_16x16x4_f64 --
// rows (lane >> 4) + 4 r of the tile, r = 0..3 -- hold FOUR CONSECUTIVE bins of one column for the lane's sample:
//   PWQuad K = 16, column t:  tile 2t     row g + 4r  <-  height V[4g + r] RELATIVE to V[16]   (g = lane >> 4)
//                             tile 2t + 1 row g + 4r  <-  width  W[4g + r]
//   PWLin  K = 16 m, column t: tile m t + j, row g + 4r  <-  height Q[16j + 4g + r]
// (PWQuad has 17 heights per column: y and the Jacobian only see the heights up to a common factor -- both divide by the area
// sum (V_k + V_k+1) W_k -- so the rows of the heights are packed as DIFFERENCES to the row of V[16], which then is exp(0) = 1
// exactly: 2 T tiles instead of 2 T + 1, and no seventeenth value to route.  Rounding only: the logits move by ~1e-16 |w|.)
// The four lanes that share a sample (lane groups g = 0..3) then finish the column TOGETHER on registers: local sums of
// four bins, a 4-step cross-lane prefix (ds_bpermute: the LDS crossbar, no LDS storage), bin search on the lane's own
// four edges, and the bin's values fetched from the owning lane.  No image, no per-lane 16-long scan; every lane is busy
// for every column.
//
// Bit-exact bin selection (PWQuad): the reference compares fl(cumsum_k / total) <= x with a SEQUENTIAL cumsum
// (coupling_cells.py:198-202).  The blocked prefix differs from it by <= 2^-49 relative, so a raw edge is trusted only
// outside a 2^-46 band around x * total; a wave with any sample inside the band gathers the 16 widths of that column and
// runs pwquad_bin_raw -- the sequential search with the reference's own divisions -- for the whole wave (probability
// ~1e-11 per column and wave).  Values follow from raw sums exactly as in pwquad_column (<= few ulp).
#pragma once
#include "flow_mfma.h"

#define MF_REG_NONE 0
#define MF_REG_PWQUAD16 1
#define MF_REG_PWLIN16 2

// packed row (tile, row in tile) -> row of the reference-order last Linear, or -1 (zero padding)
__host__ __device__ inline int mf_reg_row(int mode, int T, int K, int tile, int rho) {
  const int g = rho & 3, r = rho >> 2;
  if (mode == MF_REG_PWQUAD16) {
    const int t = tile >> 1, k = 4 * g + r;
    return t < T ? t * 33 + ((tile & 1) ? 17 + k : k) : -1;
  }
  if (mode == MF_REG_PWLIN16) {
    const int m = K / 16, t = tile / m, j = tile % m;
    return t < T ? t * K + 16 * j + 4 * g + r : -1;
  }
  return 16 * tile + rho;
}
// row subtracted from it (PWQuad heights: the row of V[16] of the same column), or -1
__host__ __device__ inline int mf_reg_row_sub(int mode, int T, int K, int tile, int rho) {
  if (mode == MF_REG_PWQUAD16 && !(tile & 1) && (tile >> 1) < T) return (tile >> 1) * 33 + 16;
  return -1;
}
__host__ __device__ inline int mf_reg_tiles(int mode, int T, int K) {
  if (mode == MF_REG_PWQUAD16) return 2 * T;
  if (mode == MF_REG_PWLIN16) return T * (K / 16);
  return 0;
}

// value of `v` held by lane group `gsrc` for this lane's sample
__device__ __forceinline__ double mf_from_group(double v, int gsrc) { return __shfl(v, (gsrc << 4) | (threadIdx.x & 15), 64); }
__device__ __forceinline__ int mf_from_group_i(int v, int gsrc) { return __shfl(v, (gsrc << 4) | (threadIdx.x & 15), 64); }
// inclusive prefix over the four lane groups of a sample (g = lane >> 4), and the total
__device__ __forceinline__ double mf_scan4(double a, int g, double* total) {
  const double u1 = mf_from_group(a, (g + 3) & 3);
  const double a1 = a + (g >= 1 ? u1 : 0.0);
  const double u2 = mf_from_group(a1, (g + 2) & 3);
  const double a2 = a1 + (g >= 2 ? u2 : 0.0);
  *total = mf_from_group(a2, 3);
  return a2;
}
__device__ __forceinline__ double mf_sum4(double a) {
  a += __shfl_xor(a, 16, 64);
  a += __shfl_xor(a, 32, 64);
  return a;
}

// PWQuad column on the registers of the four lanes of a sample.  V[r] = height 4g + r, Vn = height 4g + 4 (g == 3: V[16] = 1),
// W[r] = width 4g + r (raw exps).  Every lane returns the same y / Jacobian factor / bin.
__device__ __forceinline__ double mf_pwquad16_coop(double xb, const double (&V)[4], double Vn, const double (&W)[4], int g, double* jf, int* bin) {
  xb = (xb > 1.0 - 1e-6) ? 1.0 - 1e-6 : xb;                        // :167
  const double p1 = W[0], p2 = p1 + W[1], p3 = p2 + W[2], p4 = p3 + W[3];
  double wn;
  const double incl = mf_scan4(p4, g, &wn);                        // Wsum[-1]  :180-182 (blocked order)
  const double off = incl - p4;
  const double run[4] = {off + p1, off + p2, off + p3, incl};      // raw cumulative width right of bins 4g .. 4g + 3
  const double q = xb * wn;
  const double qlo = q * (1.0 - 0x1p-46), qhi = q * (1.0 + 0x1p-46);
  int cnt = 0;
  bool amb = false;
  double lbest = 0.0;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const bool le = run[r] <= qlo;
    amb |= (run[r] > qlo) && (run[r] < qhi);
    cnt += le ? 1 : 0;
    lbest = le ? run[r] : lbest;                                   // edges are non-decreasing: the last kept one is the largest
  }
  cnt += __shfl_xor(cnt, 16, 64);
  cnt += __shfl_xor(cnt, 32, 64);
  double best = fmax(lbest, __shfl_xor(lbest, 16, 64));
  best = fmax(best, __shfl_xor(best, 32, 64));
  int idx = cnt > 15 ? 15 : cnt;
  // widths that vanish in the cumsum (equal consecutive edges) make the reference's argmax pick the FIRST of them: only the
  // sequential search reproduces that, so such columns take it as well (minimum width below one ulp of the total)
  const double wmin = fmin(fmin(W[0], W[1]), fmin(W[2], W[3]));
  amb |= wmin * 0x1p52 < wn;
  if (__any(amb)) {                                                // wave-uniform; practically never
    // the reference's search, verbatim (pwquad_bin_raw's division branch), on the SEQUENTIAL cumsum: the widths come from their
    // owning lanes one at a time (rolled loops: this path must not cost registers)
    double wseq = 0.0;
#pragma unroll 1
    for (int k = 0; k < 16; ++k) {
      const int r = k & 3;
      const double wl = r == 0 ? W[0] : r == 1 ? W[1] : r == 2 ? W[2] : W[3];
      wseq += mf_from_group(wl, k >> 2);
    }
    double runs = 0.0, bestrun = 0.0;
    int id = 0;
#pragma unroll 1
    for (int k = 0; k < 16; ++k) {
      const int r = k & 3;
      const double wl = r == 0 ? W[0] : r == 1 ? W[1] : r == 2 ? W[2] : W[3];
      runs += mf_from_group(wl, k >> 2);
      const bool le = !((runs / wseq) > xb);                       // the reference's comparison, verbatim
      const bool up = le && (runs > bestrun);
      bestrun = up ? runs : bestrun;
      id = up ? k + 1 : id;
    }
    const bool none = !(bestrun > 1e-30 * wseq);
    id = none ? 0 : id;
    bestrun = none ? 0.0 : bestrun;
    idx = id > 15 ? 15 : id;
    best = bestrun * (wn / wseq);                                  // left edge on the blocked total's scale (<= 1 ulp)
  } else {
    const bool none = !(best > 1e-30 * wn);                        // sentinel of the reference's argmax (:198-202)
    idx = none ? 0 : idx;
    best = none ? 0.0 : best;
  }
  // trapezoid areas S_k = (V_k + V_k+1) W_k: total A and the part left of the bin  (:194, :209-211)
  const double S0 = (V[0] + V[1]) * W[0], S1 = (V[1] + V[2]) * W[1], S2 = (V[2] + V[3]) * W[2], S3 = (V[3] + Vn) * W[3];
  const double s01 = S0 + S1, s012 = s01 + S2, s0123 = s012 + S3;
  const int dloc = idx - 4 * g;                                    // bins of this lane left of the selected one
  const double pl = dloc <= 0 ? 0.0 : dloc == 1 ? S0 : dloc == 2 ? s01 : dloc == 3 ? s012 : s0123;
  const double A = mf_sum4(s0123), pre = mf_sum4(pl);
  // the bin's own heights and width live in lane group idx >> 2, register idx & 3
  const int rr = idx & 3, go = idx >> 2;
  const double vb_l = rr == 0 ? V[0] : rr == 1 ? V[1] : rr == 2 ? V[2] : V[3];
  const double vb1_l = rr == 0 ? V[1] : rr == 1 ? V[2] : rr == 2 ? V[3] : Vn;
  const double wb_l = rr == 0 ? W[0] : rr == 1 ? W[1] : rr == 2 ? W[2] : W[3];
  const double Vb = mf_from_group(vb_l, go), Vb1 = mf_from_group(vb1_l, go), Wb = mf_from_group(wb_l, go);
  const double rcp = nis_rcp(Wb * A);                              // one reciprocal for both divisions (see pwquad_column)
  const double iA = rcp * Wb;
  const double alpha = (xb * wn - best) * (rcp * A);               // :206-207
  const double num = ((alpha * alpha) / 2 * (Vb1 - Vb) + alpha * Vb) * (2 * Wb) + pre;   // :213-220
  *jf = lerp_torch(Vb, Vb1, alpha) * ((2 * wn) * iA);              // :224-225
  *bin = idx;
  return num * iA;
}

// PWLin: running state of one column over its K / 16 tiles (the bin is known before the heights, coupling_cells.py:126-133)
struct MfLinAcc {
  double tot, below, qb;
};
__device__ __forceinline__ void mf_pwlin16_tile(MfLinAcc& a, const double (&e)[4], int d) {    // d = bin - first bin of this lane in this tile
  const double s2 = e[0] + e[1], s3 = s2 + e[2], s4 = s3 + e[3];
  a.tot += s4;
  a.below += d <= 0 ? 0.0 : d == 1 ? e[0] : d == 2 ? s2 : d == 3 ? s3 : s4;
  a.qb += d == 0 ? e[0] : d == 1 ? e[1] : d == 2 ? e[2] : d == 3 ? e[3] : 0.0;
}

#ifndef NIS_MFR_WAVES
#define NIS_MFR_WAVES 3                // waves per SIMD the kernel is allocated for (no LDS image: registers decide)
#endif
#ifndef NIS_MFR_ST
#define NIS_MFR_ST 1                   // sample tiles (of 16) per wave and trip: 1 keeps the activations at 16 VGPRs (three waves without spills)
#endif

template <int KIND, int WM, bool FUSE>
__global__ __launch_bounds__(NIS_MF_TPB, NIS_MFR_WAVES) void k_cell_apply_mfr(CellK c, double* __restrict__ planes, long ld, long N, int bn_train,
                                                                               int* __restrict__ bins_out, const double* __restrict__ wm_all,
                                                                               const double* __restrict__ wm_last,
                                                                               const double* __restrict__ bias_pad, FuseF fz) {
  static_assert(KIND == NIS_PWQUAD || KIND == NIS_PWLIN, "register-resident bijectors: PWQuad K = 16, PWLin K = 16 m");
  constexpr int ST = NIS_MFR_ST, JT = MfCfg<WM>::JT, KQ = MfCfg<WM>::KQ, NW = NIS_MF_TPB / 64;
  __shared__ double lds_coef[(NIS_MAX_HIDDEN + 1) * 2 * WM];
  __shared__ double red[3 * NIS_MF_TPB];
  __shared__ int lds_plane[NIS_MAX_DIM];
  __shared__ double etab[NIS_EXP_N];
  __shared__ double fmom[NW][4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, gq = lane >> 4, s = lane & 15;
  const int g = c.g0 + blockIdx.y;
  planes += (size_t)g * c.plane_gs;
  if (tid < NIS_MAX_DIM) lds_plane[tid] = c.in_plane[tid];
  nis_exp_init(etab);
  load_coefs<WM, NIS_MF_TPB>(c, g, c.L + 1, (bn_train && c.pmerge) ? cellk_fresh(c, c.L + 1) : -1, blockIdx.x == 0, lds_coef, red);
  __syncthreads();
  const bool wplanes = FUSE ? fz.write_planes != 0 : true;
  double fS1 = 0.0, fS2 = 0.0, fcnt = 0.0, fmx = -__builtin_huge_val(), fshift = 0.0;
  bool ffirst = true;
  const double f_i2s = FUSE ? 1.0 / (2 * fz.p[1] * fz.p[1]) : 0.0;
  const double f_norm = (FUSE && fz.kind == NIS_F_NORMAL) ? 1.0 / (fz.p[1] * sqrt(pow(2 * M_PI, (double)c.d))) : 1.0;
  const int T = c.T;
  const int mlin = KIND == NIS_PWLIN ? c.K / 16 : 0;                // tiles per PWLin column
  const int ntile = KIND == NIS_PWQUAD ? 2 * T : T * mlin;
  const long wstride = (long)gridDim.x * NW * 16 * ST;
  // A fragments + bias rows of the NEXT tile are requested right after a tile's MFMA chain has been issued (cyclic, sample-independent)
  double an[KQ];
  nis_d4 bnx = {0.0, 0.0, 0.0, 0.0};
  auto load_tile = [&](int jt) {
#pragma unroll
    for (int q = 0; q < KQ; ++q) an[q] = wm_last[((size_t)jt * KQ + q) * 64 + lane];
#pragma unroll
    for (int r = 0; r < 4; ++r) bnx[r] = bias_pad[16 * jt + gq + 4 * r];
  };
  // tile order of a sample tile: PWQuad: (heights, widths) per column; PWLin: in order; cyclic over the sample tiles
  const int first_tile = 0;
  load_tile(first_tile);
  constexpr int PQ = 2;
  const bool pin = NIS_MF_PREFETCH && c.P <= 4 * PQ;
  double xin[PQ][ST];
  const long base0 = ((long)blockIdx.x * NW + wave) * 16 * ST;
  if (pin && base0 < N) mf_issue_input<WM, PQ, ST>(c, planes, ld, N, base0, lds_plane, xin);
  for (long base = base0; base < N; base += wstride) {
    double H[KQ][ST];
    {
      nis_d4 acc[JT][ST];
      if (pin) {
        mf_finish_input<WM, PQ, ST>(c, N, base, lds_coef, xin, H);
        mf_issue_input<WM, PQ, ST>(c, planes, ld, N, base + wstride < N ? base + wstride : base, lds_plane, xin);
      } else {
        mf_load_input<WM, ST>(c, planes, ld, N, base, lds_plane, lds_coef, H);
      }
      mf_hidden<WM, ST>(c, wm_all, lds_coef, c.L, H, acc);
      mf_bn_relu<WM, ST>(lds_coef + (size_t)c.L * 2 * WM, acc, H);
    }
    if (c.out_off != 0) {                              // history mode: the next plane set gets the pass-through block too
      for (int p = gq; p < c.P; p += 4)
#pragma unroll
        for (int st = 0; st < ST; ++st) {
          const long i = base + 16 * st + s;
          if (i < N) planes[(size_t)lds_plane[p] * ld + c.out_off + i] = planes[(size_t)lds_plane[p] * ld + i];
        }
    }
#pragma unroll 1
    for (int st = 0; st < ST; ++st) {
      const long i = base + 16 * st + s;
      const bool live = i < N;
      double jprod = 1.0;
      double facc = 0.0, fc0 = 0.0, fc1 = 0.0;
      double jinq = 1.0;
      if (!c.first && live) jinq = planes[(size_t)c.jplane * ld + i];
      double xbn = (live && T > 0) ? planes[(size_t)lds_plane[c.P] * ld + i] : 0.0;      // transformed input of column 0 (all four lanes)
      if (FUSE) {
        for (int p = gq; p < c.P; p += 4) {            // pass-through columns, spread over the four lanes of the sample
          const double xp = live ? planes[(size_t)lds_plane[p] * ld + i] : 0.0;
          if (fz.kind == NIS_F_NORMAL) facc = fma((xp - fz.p[0]) * (xp - fz.p[0]), f_i2s, facc);
          if (fz.kind == NIS_F_CAMEL) {
            fc0 = (p == fz.loc0) ? xp : fc0;
            fc1 = (p == fz.loc1) ? xp : fc1;
          }
        }
      }
      double Hs[KQ];
#pragma unroll
      for (int q = 0; q < KQ; ++q) {
        Hs[q] = H[q][0];
#pragma unroll
        for (int u = 1; u < ST; ++u) Hs[q] = (st == u) ? H[q][u] : Hs[q];
      }
      // one tile: MFMA chain on the current fragments, request the next tile's, exp of the four accumulator rows
      auto tile_exp = [&](int next_jt, double (&e)[4]) {
        nis_d4 z = bnx;
#pragma unroll
        for (int q = 0; q < KQ; ++q) z = __builtin_amdgcn_mfma_f64_16x16x4f64(an[q], Hs[q], z, 0, 0, 0);
        load_tile(next_jt);
#pragma unroll
        for (int r = 0; r < 4; ++r) e[r] = nis_exp(z[r], etab);
      };
      if (KIND == NIS_PWQUAD) {
        for (int t = 0; t < T; ++t) {
          double V[4], W[4];
          tile_exp(2 * t + 1, V);
          tile_exp(t + 1 < T ? 2 * t + 2 : first_tile, W);
          const double xb = xbn;
          if (t + 1 < T) xbn = live ? planes[(size_t)lds_plane[c.P + t + 1] * ld + i] : 0.0;
          // height 4g + 4: register 0 of the next lane group; V[16] = exp(0) for the last one (relative heights)
          const double vnx = mf_from_group(V[0], (gq + 1) & 3);
          const double Vn = gq == 3 ? 1.0 : vnx;
          double jf;
          int bin;
          const double y = mf_pwquad16_coop(xb, V, Vn, W, gq, &jf, &bin);
          if (gq == 0 && live) {
            double* col = planes + (size_t)lds_plane[c.P + t] * ld;
            if (wplanes) col[c.out_off + i] = y;
            if (bins_out && g == 0) bins_out[(size_t)(c.bins_row + t) * ld + i] = bin;
          }
          jprod *= jf;
          if (FUSE && gq == (t & 3)) {
            if (fz.kind == NIS_F_NORMAL) facc = fma((y - fz.p[0]) * (y - fz.p[0]), f_i2s, facc);
            if (fz.kind == NIS_F_CAMEL) {
              fc0 = (c.P + t == fz.loc0) ? y : fc0;
              fc1 = (c.P + t == fz.loc1) ? y : fc1;
            }
          }
        }
      } else {
        const double Kd = (double)c.K;
        for (int t = 0; t < T; ++t) {
          const double xb = xbn;
          if (t + 1 < T) xbn = live ? planes[(size_t)lds_plane[c.P + t + 1] * ld + i] : 0.0;
          const double ak = xb * Kd;
          const double bf = floor(ak);
          const double alpha = (ak - bf) / Kd;
          const int bin = (int)bf;
          MfLinAcc a = {0.0, 0.0, 0.0};
          int dl = bin - 4 * gq;
          for (int j = 0; j < mlin; ++j) {
            double e[4];
            const int jt = t * mlin + j;
            tile_exp(jt + 1 < ntile ? jt + 1 : first_tile, e);
            mf_pwlin16_tile(a, e, dl);
            dl -= 16;
          }
          const double tot = mf_sum4(a.tot), below = mf_sum4(a.below);
          double qb = mf_sum4(a.qb);
          if (bin < 0 || bin >= c.K) {
            if (c.status && live) atomicOr(c.status, 1);
            qb = __builtin_nan("");
          }
          const double itot = nis_rcp(tot);
          const double qn = qb * (Kd * itot);                                                 // :121
          const double y = qn * alpha + below * itot;                                         // :123,136-139
          if (gq == 0 && live) {
            double* col = planes + (size_t)lds_plane[c.P + t] * ld;
            if (wplanes) col[c.out_off + i] = y;
            if (bins_out && g == 0) bins_out[(size_t)(c.bins_row + t) * ld + i] = bin;
          }
          jprod *= qn;                                                                        // :141
          if (FUSE && gq == (t & 3)) {
            if (fz.kind == NIS_F_NORMAL) facc = fma((y - fz.p[0]) * (y - fz.p[0]), f_i2s, facc);
            if (fz.kind == NIS_F_CAMEL) {
              fc0 = (c.P + t == fz.loc0) ? y : fc0;
              fc1 = (c.P + t == fz.loc1) ? y : fc1;
            }
          }
        }
      }
      double jout = 1.0;                               // every lane of the sample holds the full product already
      if (live) {
        jout = (c.first ? 1.0 : jinq) * jprod;
        if (gq == 0 && wplanes) planes[(size_t)c.jplane * ld + c.out_off + i] = jout;
      }
      if (FUSE) {
        facc = mf_sum4(facc);
        fc0 = mf_sum4(fc0);
        fc1 = mf_sum4(fc1);
        double fv = 1.0;
        if (fz.kind == NIS_F_NORMAL) {                                                    // manager.py:39-40
          fv = nis_exp(-facc, etab) * f_norm;
        } else if (fz.kind == NIS_F_CAMEL) {                                              // README.md:35-36
          const double is2 = fz.p[3];
          const double e1 = -((fc0 - fz.p[0]) * (fc0 - fz.p[0]) + (fc1 - fz.p[0]) * (fc1 - fz.p[0])) * is2;
          const double e2 = -((fc0 - fz.p[1]) * (fc0 - fz.p[1]) + (fc1 - fz.p[1]) * (fc1 - fz.p[1])) * is2;
          fv = nis_exp(e1, etab) + nis_exp(e2, etab);
        }
        const double v = fv * jout;
        const bool mine = gq == 0 && live;           // one lane per sample contributes
        if (ffirst) {
          fshift = __shfl(v, 0, 64);                 // lane 0 = (group 0, sample 0) of the wave's first tile: always a valid sample
          ffirst = false;
        }
        const double tt = mine ? v - fshift : 0.0;
        fS1 += tt;
        fS2 = fma(tt, tt, fS2);
        fmx = mine ? fmax(fmx, v) : fmx;
        fcnt += mine ? 1.0 : 0.0;
      }
    }
  }
  if (FUSE) block_record<NW>(fshift, fS1, fS2, fcnt, fmx, fmom, fz.partials + ((size_t)g * fz.maxb + blockIdx.x) * 4);
}
