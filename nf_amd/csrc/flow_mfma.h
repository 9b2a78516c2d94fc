// Wide conditioners (register width WM >= 16) on the f64 matrix cores: v_mfma_f64_16x16x4_f64.
//
// Orientation: C[j][s] = sum_k W[j][k] * H[k][s]  (rows = output features, columns = 16 samples), i.e.
//   A operand = weights, lane l holds W[16*jt + (l&15)][4*q + (l>>4)]      (pre-packed per (jt,q) as 64 contiguous doubles)
//   B operand = activations, lane l holds H[4*q + (l>>4)][sample l&15]
//   C/D: lane l holds rows (l>>4) + 4*r (r = 0..3) of column l&15    (probed on gfx950: tools/probe/mfma_f64_layout.hip)
// so accumulator register r of output tile jt IS the B operand of k-step q = 4*jt + r of the next layer: the hidden
// stack runs register-to-register with no LDS and no cross-lane traffic.  BN + ReLU are applied to the accumulators
// (per-lane feature index -> coefficient read from LDS).  A wave owns ST tiles of 16 samples, so every weight
// fragment feeds ST MFMAs.
//
// The last layer is produced in column batches (NCB transformed columns = NCB*out rows at a time): its exp() values
// go through a per-wave LDS image [row][16 samples] and are read back one (sample, column) per lane, where the SAME
// pwquad_column<> / streamed PWLin code as the narrow kernels finishes the cell (bit-identical bin selection).
#pragma once
#include "flow_kernels.h"

typedef double nis_d4 __attribute__((ext_vector_type(4)));

#define NIS_MF_TPB 256                 // 4 waves, one per SIMD
#define NIS_MF_NCB 4                   // transformed columns per bijector batch (4 x 16 samples = 64 lanes); CellK::mf_ncb lowers it to 2 or 1 when
                                       // the LDS image of four columns would leave ONE workgroup per CU (C3: 64 heights per column = 32 KB per wave)
// (two independent accumulation chains per wave in the last layer measured SLOWER: C5 forward 10.3 vs 9.1 ms, C3 26.3 vs 22.2)
#ifndef NIS_MF_PREFETCH
#define NIS_MF_PREFETCH 1              // 0: operands requested where they are used (A/B measurements)
#endif
#ifndef NIS_MF_TILE_INTERLEAVE
#define NIS_MF_TILE_INTERLEAVE 1       // next tile's A fragments requested one by one behind the MFMAs that free their registers (0: all behind the chain; round 3: C3 / C4 / C5 iteration 12.20 / 2.64 / 6.65 -> 11.81 / 2.57 / 6.46 ms)
#endif

// Row tiles of the last layer one column batch touches (its rows [cb NCB out, (cb+1) NCB out) need not start on a tile): the
// per-wave LDS image holds that many whole tiles, so every accumulator row is written without a guard at a fixed offset.
__host__ __device__ inline int mf_cb_tiles(int T, int out, int R, int ncb_cols) {
  int mx = 1;
  for (int cb = 0; cb * ncb_cols < T; ++cb) {
    const int r0 = cb * ncb_cols * out, r1 = (r0 + ncb_cols * out < R) ? r0 + ncb_cols * out : R;
    const int n = (r1 + 15) / 16 - r0 / 16;
    mx = n > mx ? n : mx;
  }
  return mx;
}

#ifndef NIS_MF_ST
#define NIS_MF_ST 2
#endif
#ifndef NIS_MF_WAVES_LIN
#define NIS_MF_WAVES_LIN 0             // 3: allocate the PWLin apply kernel for three waves per SIMD (one column per bijector batch)
#endif
#ifndef NIS_MF_WAVES
#define NIS_MF_WAVES 2                 // waves per SIMD the apply kernel is allocated for (K <= 16)
#endif
template <int WM, int STV = NIS_MF_ST>
struct MfCfg {
  static constexpr int ST = STV;                   // sample tiles (of 16) per wave: keeps VGPR+AGPR <= 256 -> 2 waves/SIMD,
                                                   // so one wave's MFMA phase overlaps the other's VALU (exp / bijector) phase
  static constexpr int JT = WM / 16;               // output row tiles of a hidden layer
  static constexpr int KQ = WM / 4;                // k-steps of a hidden layer
};

// Load the pass-through block as B operands and apply BN_0:  H[q][st] = sc0[k]*x[k][s] + sh0[k],  k = 4q + (lane>>4).
template <int WM, int STV = NIS_MF_ST>
__device__ __forceinline__ void mf_load_input(const CellK& c, const double* planes, long ld, long N, long base, const int* lds_plane,
                                              const double* lds_coef, double (&H)[MfCfg<WM>::KQ][STV]) {
  constexpr int ST = STV, KQ = MfCfg<WM>::KQ;
  const int lane = threadIdx.x & 63, g = lane >> 4, s = lane & 15;
  const int nq = (c.P + 3) / 4;                      // k-steps that hold pass-through columns (wave-uniform: no exec-masked loads)
  const bool whole = base + 16 * ST <= N;            // wave-uniform: every sample of the wave's tiles exists
#pragma unroll
  for (int q = 0; q < KQ; ++q) {
    if (q < nq) {
      const int k = 4 * q + g;
      const bool kv = k < c.P;
      const size_t pl = (size_t)lds_plane[kv ? k : 0] * ld;
      const double sc = kv ? lds_coef[k] : 0.0, sh = kv ? lds_coef[WM + k] : 0.0;
#pragma unroll
      for (int st = 0; st < ST; ++st) {
        const long i = base + 16 * st + s;
        const double x = planes[pl + (whole || i < N ? i : base)];       // a ragged tile re-reads an existing sample (masked below)
        H[q][st] = (whole || i < N) ? fma(x, sc, sh) : 0.0;              // sc = sh = 0 for the padding columns
      }
    } else {
#pragma unroll
      for (int st = 0; st < ST; ++st) H[q][st] = 0.0;
    }
  }
}

// The same in two halves for a software-pipelined caller (P <= 4 PQ): `mf_issue_input` only requests the raw pass-through values of
// the tiles at `base` (a ragged tile re-reads an existing sample), `mf_finish_input` turns them into B operands one iteration later.
template <int WM, int PQ, int STV = NIS_MF_ST>
__device__ __forceinline__ void mf_issue_input(const CellK& c, const double* planes, long ld, long N, long base, const int* lds_plane,
                                               double (&xr)[PQ][STV]) {
  constexpr int ST = STV;
  const int lane = threadIdx.x & 63, g = lane >> 4, s = lane & 15;
  const bool whole = base + 16 * ST <= N;
#pragma unroll
  for (int q = 0; q < PQ; ++q) {
    const int k = 4 * q + g;
    const size_t pl = (size_t)lds_plane[k < c.P ? k : 0] * ld;
#pragma unroll
    for (int st = 0; st < ST; ++st) {
      const long i = base + 16 * st + s;
      xr[q][st] = planes[pl + ((whole || i < N) ? i : base)];
    }
  }
}
template <int WM, int PQ, int STV = NIS_MF_ST>
__device__ __forceinline__ void mf_finish_input(const CellK& c, long N, long base, const double* lds_coef, const double (&xr)[PQ][STV],
                                                double (&H)[MfCfg<WM>::KQ][STV]) {
  constexpr int ST = STV, KQ = MfCfg<WM>::KQ;
  const int lane = threadIdx.x & 63, g = lane >> 4, s = lane & 15;
  const bool whole = base + 16 * ST <= N;
#pragma unroll
  for (int q = 0; q < KQ; ++q) {
#pragma unroll
    for (int st = 0; st < ST; ++st) H[q][st] = 0.0;
    if (q < PQ) {
      const int k = 4 * q + g;
      const bool kv = k < c.P;
      const double sc = kv ? lds_coef[k] : 0.0, sh = kv ? lds_coef[WM + k] : 0.0;
#pragma unroll
      for (int st = 0; st < ST; ++st) H[q][st] = (whole || base + 16 * st + s < N) ? fma(xr[q < PQ ? q : 0][st], sc, sh) : 0.0;
    }
  }
}

// One hidden Linear on the matrix cores: acc[jt][st] = sum_q Wm[jt][q] (x) H[q][st];  nq = k-steps actually populated.
template <int WM, int STV = NIS_MF_ST>
__device__ __forceinline__ void mf_dense(const double* __restrict__ wm, int nq, const double (&H)[MfCfg<WM>::KQ][STV],
                                         nis_d4 (&acc)[MfCfg<WM>::JT][STV]) {
  constexpr int ST = STV, JT = MfCfg<WM>::JT, KQ = MfCfg<WM>::KQ;
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int jt = 0; jt < JT; ++jt) {
#pragma unroll
    for (int st = 0; st < ST; ++st) acc[jt][st] = nis_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int q = 0; q < KQ; ++q) {
      if (q < nq) {
        const double a = wm[((size_t)jt * KQ + q) * 64 + lane];
#pragma unroll
        for (int st = 0; st < ST; ++st) acc[jt][st] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, H[q][st], acc[jt][st], 0, 0, 0);
      }
    }
  }
}

// BN_l + ReLU on the accumulators -> B operands of the next layer (feature j = 16 jt + (lane>>4) + 4 r  ->  k-step 4 jt + r).
template <int WM, int STV = NIS_MF_ST>
__device__ __forceinline__ void mf_bn_relu(const double* lds_coef_l, const nis_d4 (&acc)[MfCfg<WM>::JT][STV],
                                           double (&H)[MfCfg<WM>::KQ][STV]) {
  constexpr int ST = STV, JT = MfCfg<WM>::JT;
  const int g = (threadIdx.x & 63) >> 4;
#pragma unroll
  for (int jt = 0; jt < JT; ++jt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int j = 16 * jt + g + 4 * r;
      const double sc = lds_coef_l[j], sh = lds_coef_l[WM + j];
#pragma unroll
      for (int st = 0; st < ST; ++st) H[4 * jt + r][st] = fmax(fma(acc[jt][st][r], sc, sh), 0.0);
    }
}

// Hidden stack up to (excluding) BN layer `upto` (1..L): leaves the pre-BN accumulators of that layer in acc.
template <int WM, int STV = NIS_MF_ST>
__device__ __forceinline__ void mf_hidden(const CellK& c, const double* wm_all, const double* lds_coef, int upto,
                                          double (&H)[MfCfg<WM>::KQ][STV], nis_d4 (&acc)[MfCfg<WM>::JT][STV]) {
  for (int l = 0; l < upto; ++l) {
    const int nq = l == 0 ? (c.P + 3) / 4 : MfCfg<WM>::KQ;
    mf_dense<WM, STV>(wm_all + (size_t)l * WM * WM, nq, H, acc);
    if (l + 1 < upto) mf_bn_relu<WM, STV>(lds_coef + (size_t)(l + 1) * 2 * WM, acc, H);
  }
}

// Register-resident hidden weights for the statistics kernel (<= 3 BN layers below the one measured, P <= 4 PQ): the fragments
// of Linear_0 .. Linear_2 do not depend on the sample, and the kernel has ~90 VGPRs to spare at its two waves per SIMD, so they are
// loaded ONCE per workgroup instead of 16 x 512 B from L2 per layer and pair of sample tiles (the loop's only memory traffic then
// is the pass-through block).
template <int WM, int PQ>
struct MfWreg {
  double w0[MfCfg<WM>::JT][PQ];
  double w[2][MfCfg<WM>::JT][MfCfg<WM>::KQ];
};
template <int WM, int PQ>
__device__ __forceinline__ void mf_wreg_load(const double* __restrict__ wm_all, int upto, MfWreg<WM, PQ>& W) {
  constexpr int JT = MfCfg<WM>::JT, KQ = MfCfg<WM>::KQ;
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int jt = 0; jt < JT; ++jt) {
#pragma unroll
    for (int q = 0; q < PQ; ++q) W.w0[jt][q] = wm_all[((size_t)jt * KQ + q) * 64 + lane];
#pragma unroll
    for (int l = 1; l <= 2; ++l)
#pragma unroll
      for (int q = 0; q < KQ; ++q) W.w[l - 1][jt][q] = l < upto ? wm_all[(size_t)l * WM * WM + ((size_t)jt * KQ + q) * 64 + lane] : 0.0;
  }
}
template <int WM, int PQ>
__device__ __forceinline__ void mf_hidden_reg(const CellK& c, const MfWreg<WM, PQ>& W, const double* lds_coef, int upto,
                                              double (&H)[MfCfg<WM>::KQ][MfCfg<WM>::ST], nis_d4 (&acc)[MfCfg<WM>::JT][MfCfg<WM>::ST]) {
  constexpr int ST = MfCfg<WM>::ST, JT = MfCfg<WM>::JT, KQ = MfCfg<WM>::KQ;
  const int nq = (c.P + 3) / 4;
#pragma unroll
  for (int jt = 0; jt < JT; ++jt) {
#pragma unroll
    for (int st = 0; st < ST; ++st) acc[jt][st] = nis_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int q = 0; q < PQ; ++q)
      if (q < nq) {
#pragma unroll
        for (int st = 0; st < ST; ++st) acc[jt][st] = __builtin_amdgcn_mfma_f64_16x16x4f64(W.w0[jt][q], H[q][st], acc[jt][st], 0, 0, 0);
      }
  }
#pragma unroll
  for (int l = 1; l <= 2; ++l) {
    if (l < upto) {
      mf_bn_relu<WM>(lds_coef + (size_t)l * 2 * WM, acc, H);
#pragma unroll
      for (int jt = 0; jt < JT; ++jt) {
#pragma unroll
        for (int st = 0; st < ST; ++st) acc[jt][st] = nis_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int q = 0; q < KQ; ++q) {
#pragma unroll
          for (int st = 0; st < ST; ++st) acc[jt][st] = __builtin_amdgcn_mfma_f64_16x16x4f64(W.w[l - 1][jt][q], H[q][st], acc[jt][st], 0, 0, 0);
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Batch statistics of BN layer `layer` >= 1 (pre-BN accumulators of Linear_{layer-1}).
#ifndef NIS_MF_WREG
#define NIS_MF_WREG 1
#endif
#ifndef NIS_MF_STATS_WAVES
#define NIS_MF_STATS_WAVES 2
#endif
template <int WM, int TPBS>
__global__ __launch_bounds__(TPBS, WM <= 32 ? NIS_MF_STATS_WAVES * NIS_MF_TPB / TPBS : 1) void k_bn_stats_mfma(CellK c, const double* __restrict__ planes, long ld, long N, int layer,
                                                              const double* __restrict__ wm_all) {
  constexpr int ST = MfCfg<WM>::ST, JT = MfCfg<WM>::JT, KQ = MfCfg<WM>::KQ, NW = TPBS / 64;
  __shared__ double lds_coef[NIS_MAX_HIDDEN * 2 * WM];
  __shared__ double red[3 * TPBS];
  __shared__ int lds_plane[NIS_MAX_DIM];
  __shared__ double wmom[NW][WM][3];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, gq = lane >> 4, s = lane & 15;
  const int g = c.g0 + blockIdx.y;
  planes += (size_t)g * c.plane_gs;
  if (tid < NIS_MAX_DIM) lds_plane[tid] = c.in_plane[tid];
  load_coefs<WM, TPBS>(c, g, layer, c.pmerge ? cellk_fresh(c, layer) : -1, blockIdx.x == 0, lds_coef, red);

  double S1[JT][4], S2[JT][4], shf[JT][4];
#pragma unroll
  for (int jt = 0; jt < JT; ++jt)
#pragma unroll
    for (int r = 0; r < 4; ++r) S1[jt][r] = S2[jt][r] = shf[jt][r] = 0.0;
  double cnt = 0.0;
  bool first = true;
  const long wstride = (long)gridDim.x * NW * 16 * ST;
  // the pass-through block of the NEXT pair of sample tiles is requested one iteration ahead (cells with P <= 8: two k-steps)
  constexpr int PQ = 2;
  const bool pin = NIS_MF_PREFETCH && c.P <= 4 * PQ;
  double xin[PQ][ST];
  const long base0 = ((long)blockIdx.x * NW + wave) * 16 * ST;
  if (pin && base0 < N) mf_issue_input<WM, PQ>(c, planes, ld, N, base0, lds_plane, xin);
  constexpr bool WREG_OK = NIS_MF_WREG && WM <= 32;           // width 64: 136 fragments per lane do not fit (1.7 KB of scratch)
  const bool wres = WREG_OK && pin && layer <= 3;             // hidden weights resident in registers (mf_hidden_reg)
  MfWreg<WM, WREG_OK ? PQ : 1> Wr;
  if constexpr (WREG_OK) {
    if (wres) mf_wreg_load<WM, PQ>(wm_all, layer, Wr);
  }
  for (long base = base0; base < N; base += wstride) {
    double H[KQ][ST];
    nis_d4 acc[JT][ST];
    if (pin) {
      mf_finish_input<WM, PQ>(c, N, base, lds_coef, xin, H);
      mf_issue_input<WM, PQ>(c, planes, ld, N, base + wstride < N ? base + wstride : base, lds_plane, xin);
    } else {
      mf_load_input<WM>(c, planes, ld, N, base, lds_plane, lds_coef, H);
    }
    bool done = false;
    if constexpr (WREG_OK) {
      if (wres) {
        mf_hidden_reg<WM, PQ>(c, Wr, lds_coef, layer, H, acc);
        done = true;
      }
    }
    if (!done) mf_hidden<WM>(c, wm_all, lds_coef, layer, H, acc);
    if (first) {                       // shift = value of the wave's first sample (lane with s == 0 of each row group)
#pragma unroll
      for (int jt = 0; jt < JT; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) shf[jt][r] = __shfl(acc[jt][0][r], lane & 48, 64);
      first = false;
    }
    if (base + 16 * ST <= N) {                       // whole tiles (wave-uniform): no per-lane masks
      if (gq == 0) cnt += (double)ST;
#pragma unroll
      for (int st = 0; st < ST; ++st)
#pragma unroll
        for (int jt = 0; jt < JT; ++jt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const double t = acc[jt][st][r] - shf[jt][r];
            S1[jt][r] += t;
            S2[jt][r] = fma(t, t, S2[jt][r]);
          }
    } else {
#pragma unroll
      for (int st = 0; st < ST; ++st) {
        const bool ok = base + 16 * st + s < N;
        if (gq == 0 && ok) cnt += 1.0;
#pragma unroll
        for (int jt = 0; jt < JT; ++jt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const double t = ok ? acc[jt][st][r] - shf[jt][r] : 0.0;
            S1[jt][r] += t;
            S2[jt][r] = fma(t, t, S2[jt][r]);
          }
      }
    }
  }
  // reduce over the 16 samples of a row group (lanes with equal lane>>4), then over the count (held by group 0)
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) {
    cnt += __shfl_xor(cnt, o, 64);
#pragma unroll
    for (int jt = 0; jt < JT; ++jt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        S1[jt][r] += __shfl_xor(S1[jt][r], o, 64);
        S2[jt][r] += __shfl_xor(S2[jt][r], o, 64);
      }
  }
  const double nw = __shfl(cnt, 0, 64);
  if (s == 0) {
#pragma unroll
    for (int jt = 0; jt < JT; ++jt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = 16 * jt + gq + 4 * r;
        const bool any = nw > 0.0;
        wmom[wave][j][0] = nw;
        wmom[wave][j][1] = any ? shf[jt][r] + S1[jt][r] / nw : 0.0;
        wmom[wave][j][2] = any ? fmax(S2[jt][r] - S1[jt][r] * S1[jt][r] / nw, 0.0) : 0.0;
      }
  }
  __syncthreads();
  if (tid < WM) {
    Mom t = {0.0, 0.0, 0.0};
    for (int w = 0; w < NW; ++w) {
      Mom m = {wmom[w][tid][0], wmom[w][tid][1], wmom[w][tid][2]};
      t = mom_merge(t, m);
    }
    double* out = c.partials + ((((size_t)layer * c.G + g) * c.maxb + blockIdx.x) * WM + tid) * 3;
    out[0] = t.n;
    out[1] = t.mean;
    out[2] = t.m2;
  }
}

// ---------------------------------------------------------------------------------------------
// Full cell with the conditioner on the matrix cores.
// FUSE (last cell of an integrate step): a built-in integrand f(X) * J and its moment records are produced here -- the four
// lanes that share a sample each hold the contributions of their columns (transformed: the y they just computed;
// pass-through: column p of lane group p % 4) and combine them with two shuffles, like the Jacobian product -- and with
// fz.write_planes == 0 neither the transformed columns nor the Jacobian are stored (nothing reads them).
// REL (PWQuad, K == KM): the last layer is the relative image (nis_common.h, NIS_PACK_REL): c.out = 2K, c.R = T c.out at launch.
template <int KIND, int WM, int KM, bool KEXACT, bool FUSE = false, bool REL = false>
__global__ __launch_bounds__(NIS_MF_TPB, (KIND == NIS_PWLIN && NIS_MF_WAVES_LIN) ? NIS_MF_WAVES_LIN : KM <= 16 ? NIS_MF_WAVES : 1) void k_cell_apply_mfma(CellK c, double* __restrict__ planes, long ld, long N, int bn_train,
                                                                int* __restrict__ bins_out, const double* __restrict__ wm_all,
                                                                const double* __restrict__ wm_last, const double* __restrict__ bias_pad,
                                                                FuseF fz) {
  constexpr int ST = MfCfg<WM>::ST, JT = MfCfg<WM>::JT, KQ = MfCfg<WM>::KQ, NW = NIS_MF_TPB / 64;
  const int NCB = c.mf_ncb;                          // columns per bijector batch: lane groups gq >= NCB idle in the bijector phase
  extern __shared__ double mf_dyn[];                 // per wave: [row tiles of a column batch][16 rows][16 samples]
  __shared__ double lds_coef[(NIS_MAX_HIDDEN + 1) * 2 * WM];
  double* const red = mf_dyn;                        // the prologue's reduction scratch (3 * TPB doubles) lives in the image: 6 KB of LDS
  __shared__ int lds_plane[NIS_MAX_DIM];             // decide between one and two workgroups per CU at K = 16
  __shared__ double etab[NIS_EXP_N];
  __shared__ double fmom[NW][4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, gq = lane >> 4, s = lane & 15;
  const int g = c.g0 + blockIdx.y;
  planes += (size_t)g * c.plane_gs;
  if (tid < NIS_MAX_DIM) lds_plane[tid] = c.in_plane[tid];
  nis_exp_init(etab);
  load_coefs<WM, NIS_MF_TPB>(c, g, c.L + 1, (bn_train && c.pmerge) ? cellk_fresh(c, c.L + 1) : -1, blockIdx.x == 0, lds_coef, red);
  __syncthreads();                                   // the image is reused by the waves from here on
  const int out = c.out;
  const bool wplanes = FUSE ? fz.write_planes != 0 : true;
  double fS1 = 0.0, fS2 = 0.0, fcnt = 0.0, fmx = -__builtin_huge_val(), fshift = 0.0;
  bool ffirst = true;
  // normal(): exp(-sum (x - mu)^2 / (2 sigma^2)) / (sigma sqrt((2 pi)^d)); the constants once per launch, not per sample
  const double f_i2s = FUSE ? 1.0 / (2 * fz.p[1] * fz.p[1]) : 0.0;
  const double f_norm = (FUSE && fz.kind == NIS_F_NORMAL) ? 1.0 / (fz.p[1] * sqrt(pow(2 * M_PI, (double)c.d))) : 1.0;
  const int cb_tiles = mf_cb_tiles(c.T, out, c.R, NCB);
  double* ring = mf_dyn + (size_t)wave * cb_tiles * 256;
  const long wstride = (long)gridDim.x * NW * 16 * ST;
  // Operand requests ahead of their use.  The last layer walks its row tiles in a fixed cyclic order (the weights do not depend
  // on the sample): right after a tile's MFMA chain has been issued, the A fragments and bias rows of the NEXT tile are
  // requested into the same registers, so their L2 round trip (8 x 512 B per tile) runs under the tile's exp phase instead of
  // in front of the next chain.  The transformed input xb of the next column batch and the incoming Jacobian are requested a
  // phase ahead as well (an HBM round trip per column batch otherwise).
  double an[KQ];
  nis_d4 bnx = {0.0, 0.0, 0.0, 0.0};
  auto load_tile = [&](int jt) {
#pragma unroll
    for (int q = 0; q < KQ; ++q) an[q] = wm_last[((size_t)jt * KQ + q) * 64 + lane];
#pragma unroll
    for (int r = 0; r < 4; ++r) bnx[r] = bias_pad[16 * jt + gq + 4 * r];
  };
  const int ncb = (c.T + NCB - 1) / NCB;
  // PWLin with fewer than four columns per batch: the lane groups without a column of their own take a share of the heights of
  // another group's column (nsp lanes per column and sample, each sums K / nsp heights -- whole blocks of eight -- and the partial
  // sums meet in one or two exchanges); lane group `part` 0 of a column owns its outputs.  Other cells: one lane per column.
  const int nsp = (KIND == NIS_PWLIN && NCB < 4 && c.K % (32 / NCB) == 0) ? 4 / NCB : 1;
  const int slot = nsp > 1 ? (gq & (NCB - 1)) : gq, part = nsp > 1 ? gq / NCB : 0;
  const bool colane = nsp > 1 || gq < NCB;
  load_tile(0);
  // the pass-through block of the NEXT pair of sample tiles is requested one iteration ahead (cells with P <= 8: two k-steps)
  constexpr int PQ = 2;
  const bool pin = NIS_MF_PREFETCH && c.P <= 4 * PQ;
  double xin[PQ][ST];
  const long base0 = ((long)blockIdx.x * NW + wave) * 16 * ST;
  if (pin && base0 < N) mf_issue_input<WM, PQ>(c, planes, ld, N, base0, lds_plane, xin);
#ifdef NIS_MF_STAMPS
  long long st_hidden = 0, st_tiles = 0, st_bij = 0, st_tail = 0, st_t0 = clock64(), st_a, st_b;
#define MF_STAMP(acc_) do { st_b = clock64(); acc_ += st_b - st_a; st_a = st_b; } while (0)
#else
#define MF_STAMP(acc_) do { } while (0)
#endif
  for (long base = base0; base < N; base += wstride) {
    double H[KQ][ST];
#ifdef NIS_MF_STAMPS
    st_a = clock64();
#endif
    {
      nis_d4 acc[JT][ST];
      if (pin) {
        mf_finish_input<WM, PQ>(c, N, base, lds_coef, xin, H);
        mf_issue_input<WM, PQ>(c, planes, ld, N, base + wstride < N ? base + wstride : base, lds_plane, xin);
      } else {
        mf_load_input<WM>(c, planes, ld, N, base, lds_plane, lds_coef, H);
      }
      mf_hidden<WM>(c, wm_all, lds_coef, c.L, H, acc);
      mf_bn_relu<WM>(lds_coef + (size_t)c.L * 2 * WM, acc, H);
    }
    MF_STAMP(st_hidden);
    if (c.out_off != 0) {                              // history mode: the next plane set gets the pass-through block too
      for (int p = gq; p < c.P; p += 4)
#pragma unroll
        for (int st = 0; st < ST; ++st) {
          const long i = base + 16 * st + s;
          if (i < N) planes[(size_t)lds_plane[p] * ld + c.out_off + i] = planes[(size_t)lds_plane[p] * ld + i];
        }
    }
    // ---- last layer + bijector, one tile of 16 samples and NCB columns at a time
#pragma unroll 1
    for (int st = 0; st < ST; ++st) {
      double jprod = 1.0;                            // this lane's columns (t = NCB*cb + gq) of sample s
      double facc = 0.0, fc0 = 0.0, fc1 = 0.0;       // FUSE: this lane's share of the integrand's argument
      const long i = base + 16 * st + s;
      // requests of this sample tile that nothing below has to wait for: first column batch's xb, incoming Jacobian
      double xbn = 0.0, jinq = 1.0;
      if (NIS_MF_PREFETCH) {
        if (colane && slot < c.T && i < N) xbn = planes[(size_t)lds_plane[c.P + slot] * ld + i];
        if (!c.first && gq == 0 && i < N) jinq = planes[(size_t)c.jplane * ld + i];
      }
      if (FUSE) {
        for (int p = gq; p < c.P; p += 4) {          // pass-through columns, spread over the four lanes of the sample
          const double xp = i < N ? planes[(size_t)lds_plane[p] * ld + i] : 0.0;
          if (fz.kind == NIS_F_NORMAL) facc = fma((xp - fz.p[0]) * (xp - fz.p[0]), f_i2s, facc);
          if (fz.kind == NIS_F_CAMEL) {
            fc0 = (p == fz.loc0) ? xp : fc0;
            fc1 = (p == fz.loc1) ? xp : fc1;
          }
        }
      }
      double Hs[KQ];                                 // B operands of this sample tile (st is a runtime index: select once)
#pragma unroll
      for (int q = 0; q < KQ; ++q) {
        Hs[q] = H[q][0];
#pragma unroll
        for (int u = 1; u < ST; ++u) Hs[q] = (st == u) ? H[q][u] : Hs[q];
      }
#pragma unroll 1
      for (int cb = 0; cb < ncb; ++cb) {
        const int r0 = cb * NCB * out;
        const int r1 = min(c.R, r0 + NCB * out);
        const int t = cb * NCB + slot;
        const double xbq = xbn;                        // requested one column batch ago
        if (NIS_MF_PREFETCH) {
          const int tn = t + NCB;
          xbn = (colane && tn < c.T && i < N) ? planes[(size_t)lds_plane[c.P + (tn < c.T ? tn : 0)] * ld + i] : 0.0;
        }
        const int jt_first_next = (cb + 1 < ncb) ? ((cb + 1) * NCB * out) / 16 : 0;      // cyclic: the next sample tile starts at row tile 0
        const int jt0 = r0 / 16;
        const bool raw_row = KIND == NIS_AFFINE && (((jt0 * 16 + gq - r0) & 1) != 0);      // affine: odd rows of a column = raw shift (row parity is fixed per lane: 16 jt + 4 r is even)
        double* rw = ring + lane;                      // row (16 (jt - jt0) + gq + 4 r), sample s  ->  rw[256 (jt - jt0) + 64 r]
        for (int jt = jt0; jt * 16 < r1; ++jt) {
          nis_d4 z = bnx;
#if NIS_MF_TILE_INTERLEAVE
          // the next tile's operands are requested AS their registers fall free: the bias rows at once (z holds this tile's copy), each A
          // fragment right behind the MFMA that read it -- the chain is dependent, so the requests go out 64 cycles apart and the first
          // ones have the whole chain + the exp phase to land (all of them behind the chain left ~170 cycles for an L2 round trip, and
          // the stamps of tools/wide_stamps.py showed this loop at 2.4x its datapath time)
          const int jn = (jt + 1) * 16 < r1 ? jt + 1 : jt_first_next;
#pragma unroll
          for (int r = 0; r < 4; ++r) bnx[r] = bias_pad[16 * jn + gq + 4 * r];
#pragma unroll
          for (int q = 0; q < KQ; ++q) {
            z = __builtin_amdgcn_mfma_f64_16x16x4f64(an[q], Hs[q], z, 0, 0, 0);
            an[q] = wm_last[((size_t)jn * KQ + q) * 64 + lane];
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA ...
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);      // ... then one VMEM read
          }
#else
#pragma unroll
          for (int q = 0; q < KQ; ++q) z = __builtin_amdgcn_mfma_f64_16x16x4f64(an[q], Hs[q], z, 0, 0, 0);
          load_tile((jt + 1) * 16 < r1 ? jt + 1 : jt_first_next);
#endif
          // every row of the tile goes to the image (rows of the neighbouring batch and the zero-padded rows past R land in
          // entries nobody reads): no guards, no address arithmetic -- immediate offsets off one pointer
#pragma unroll
          for (int r = 0; r < 4; ++r) rw[64 * r] = raw_row ? z[r] : nis_exp(z[r], etab);
          rw += 256;
        }
        // (sample s, column t) per lane.  A wave executes in lock step and its LDS operations complete in order:
        // the image written above is complete; the wave barriers only pin the compiler's schedule.
        __builtin_amdgcn_wave_barrier();
        MF_STAMP(st_tiles);
        if (colane && t < c.T && i < N) {
          double* col = planes + (size_t)lds_plane[c.P + t] * ld;
          const double xb = NIS_MF_PREFETCH ? xbq : col[i];
          const double* zr = ring + (size_t)((r0 & 15) + slot * out) * 16 + s;
          double y, jf;
          int bin;
          bool amb = false;                                      // the bin decision rests on last bits (pwquad_bin_divide)
          if (KIND == NIS_PWQUAD) {
            y = pwquad_column_lds<KM, KEXACT, 16, REL>(xb, c.K, zr, &jf, &bin, &amb);
            if (amb && c.status) {
              atomicOr(c.status, NIS_STATUS_BIN_AMBIGUOUS);
              atomicOr(c.status + 1, NIS_STATUS_BIN_AMBIGUOUS);
            }
          } else if (KIND == NIS_AFFINE) {
            // rows interleaved at pack time: [exp(z0), z1] per column (coupling_cells.py:56-68)
            const double s0 = 20.0 * zr[0], z1 = zr[16];
            const double u = xb * s0 + (z1 > 0.0 ? z1 : 0.0);
            y = atan(u) / (M_PI / 2);
            jf = s0 * (1.0 / (u * u + 1.0));
            bin = 0;
          } else {
            const double ak = xb * c.K;
            const double bf = floor(ak);
            const double alpha = (ak - bf) / c.K;
            bin = (int)bf;
            // heights streamed from the image EIGHT reads at a time (one read per trip left the LDS round trip, ~130 cycles, in
            // front of every height: at K = 64 that loop was as long as the MFMA phase and pure latency).  Whole blocks of eight are
            // summed as a tree and only the running total in front of the bin's block is kept (11 instead of 32 vector instructions
            // per block); the heights of that one block left of the bin are added in a second pass over its eight values, read by
            // index.  Blocked instead of sequential sums: <= 1-2 ulp of the total, the bin index does not depend on them.
            double tot = 0.0, below = 0.0;
            const int binc = bin < 0 ? 0 : bin >= c.K ? c.K - 1 : bin;
            const int kbeg = nsp > 1 ? part * (c.K / nsp) : 0;                            // this lane's heights [kbeg, kend): whole blocks
            const int kend = nsp > 1 ? kbeg + c.K / nsp : (c.K & ~7);
            const bool inblk = binc < (c.K & ~7);
            const int jb8 = inblk ? (binc & ~7) : 0;                                      // first height of the bin's block
            int k0 = kbeg;
            for (; k0 < kend; k0 += 8) {
              double e[8];
#pragma unroll
              for (int u = 0; u < 8; ++u) e[u] = zr[(k0 + u) * 16];
              const double b = ((e[0] + e[1]) + (e[2] + e[3])) + ((e[4] + e[5]) + (e[6] + e[7]));
              below = (k0 == jb8 && inblk) ? tot : below;
              tot += b;                                                                   // :117
            }
            if (nsp > 1) {
              below = (inblk && jb8 >= kend) ? tot : below;                               // the bin lies in a later lane's share: all of this one is below it
            } else {
              for (; k0 < c.K; ++k0) {                                                    // K % 8 trailing heights, one by one
                const double e = zr[k0 * 16];
                below = (k0 == bin) ? tot : below;
                tot += e;
              }
            }
            if (kend > kbeg) {
              const double* zb = zr + (size_t)jb8 * 16;
              const int rb = (inblk && jb8 >= kbeg && jb8 < kend) ? binc - jb8 : 0;       // (the lane whose share holds the bin's block)
              double bpart = 0.0;
#pragma unroll
              for (int u = 0; u < 7; ++u) {
                const double e = zb[u * 16];
                bpart += (u < rb) ? e : 0.0;
              }
              below += bpart;
            }
            if (nsp > 1) {                                                                // partner lanes: same sample, same column (all active together)
              tot += __shfl_xor(tot, 32, 64);
              below += __shfl_xor(below, 32, 64);
              if (nsp == 4) {
                tot += __shfl_xor(tot, 16, 64);
                below += __shfl_xor(below, 16, 64);
              }
            }
            double qb = zr[(bin < 0 ? 0 : bin >= c.K ? c.K - 1 : bin) * 16];
            if (bin < 0 || bin >= c.K) {
              if (c.status && part == 0) atomicOr(c.status, 1);
              qb = __builtin_nan("");
            }
            const double itot = nis_rcp(tot);                                             // one Newton-refined reciprocal for both
            const double qn = qb * (c.K * itot);                                          // normalisations (<= 1-2 ulp)  :121
            y = qn * alpha + below * itot;                                                // :123,136-139
            jf = qn;                                                                      // :141
          }
          if (part == 0) {                                                                // (every lane of a shared column holds the same y, jf)
            if (wplanes) col[c.out_off + i] = y;
            jprod *= jf;
            if (FUSE) {
              if (fz.kind == NIS_F_NORMAL) facc = fma((y - fz.p[0]) * (y - fz.p[0]), f_i2s, facc);
              if (fz.kind == NIS_F_CAMEL) {
                fc0 = (c.P + t == fz.loc0) ? y : fc0;
                fc1 = (c.P + t == fz.loc1) ? y : fc1;
              }
            }
            if (KIND != NIS_AFFINE && bins_out && g == 0) bins_out[(size_t)(c.bins_row + t) * ld + i] = bin | (amb ? NIS_BIN_AMBIGUOUS_FLAG : 0);
          }
        }
        __builtin_amdgcn_wave_barrier();
        MF_STAMP(st_bij);
      }
      // product over this sample's columns: 4 lanes (row groups) share a sample
      jprod *= __shfl_xor(jprod, 16, 64);
      jprod *= __shfl_xor(jprod, 32, 64);
      if (KIND == NIS_AFFINE) jprod *= 1.0 / (M_PI / 2);       // the reference applies 2/pi ONCE per cell (:68)
      double jout = 1.0;
      if (gq == 0 && i < N) {
        double* jp = planes + (size_t)c.jplane * ld;
        jout = (c.first ? 1.0 : NIS_MF_PREFETCH ? jinq : jp[i]) * jprod;
        if (wplanes) jp[c.out_off + i] = jout;
      }
      if (FUSE) {
        facc += __shfl_xor(facc, 16, 64);            // every column belongs to exactly one of the four lanes: sums of
        facc += __shfl_xor(facc, 32, 64);            // (value or 0) reassemble the sample's argument in all four
        fc0 += __shfl_xor(fc0, 16, 64);
        fc0 += __shfl_xor(fc0, 32, 64);
        fc1 += __shfl_xor(fc1, 16, 64);
        fc1 += __shfl_xor(fc1, 32, 64);
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
        const bool mine = gq == 0 && i < N;          // one lane per sample contributes
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
      MF_STAMP(st_tail);
    }
  }
#ifdef NIS_MF_STAMPS
  if (blockIdx.x == 1 && blockIdx.y == 0 && lane == 0 && c.status) {      // diagnostics: bytes 64.. of the workspace header (nothing reads them)
    long long* o = reinterpret_cast<long long*>(reinterpret_cast<char*>(c.status) + 64) + wave * 5;
    o[0] = st_hidden; o[1] = st_tiles; o[2] = st_bij; o[3] = st_tail; o[4] = clock64() - st_t0;
  }
#endif
  if (FUSE) block_record<NW>(fshift, fS1, fS2, fcnt, fmx, fmom, fz.partials + ((size_t)g * fz.maxb + blockIdx.x) * 4);
}
