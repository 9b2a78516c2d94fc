// Coupling-cell kernels (templates).  One thread owns one sample; the conditioner activations live
// in registers (compile-time width WM, zero padded), weights arrive through scalar loads, BatchNorm
// affine coefficients through LDS broadcast reads.  Train-mode BatchNorm needs the batch statistics
// of every BN layer, so a cell is   [k_bn_stats(l) for l = 0..L]  ->  k_cell_apply :
// k_bn_stats(l) recomputes layers < l from the pass-through planes and emits per-workgroup
// (count, mean, M2) records; the NEXT launch merges them in its prologue (fixed order ->
// bitwise reproducible) into scale/shift.  No atomics, no grid barrier: the kernel boundary is
// the cheapest inter-workgroup sync on MI355X.
//
// Reference restated: normalizing_flows/layers/coupling_cells.py:107-142 (PWLin.forward),
// :159-228 (PWQuad.forward), :236-254 (RectNN), torch.nn.BatchNorm1d semantics.
#pragma once
#include "nis_common.h"

// ---------------------------------------------------------------------------------------------
// Prologue: bring the BN affine form (scale, shift) of layers [0, upto) into LDS.
//   layers < fresh : already finalised by an earlier launch -> c.coef (global)
//   layer == fresh : merge the partial records of the previous statistics launch (train mode)
// lds_coef layout: [layer][0: scale, 1: shift][WM].  `red` is scratch of 3*NIS_TPB doubles.
// Workgroup 0 additionally publishes the fresh coefficients and applies the running-stat update.
template <int WM>
__device__ __forceinline__ void load_coefs(const CellK& c, int upto, int fresh, double* lds_coef, double* red) {
  const int tid = threadIdx.x;
  for (int idx = tid; idx < upto * 2 * WM; idx += NIS_TPB) {
    const int l = idx / (2 * WM);
    if (l != fresh) lds_coef[idx] = c.coef[idx];
  }
  if (fresh >= 0 && fresh < upto) {
    constexpr int S = NIS_TPB / WM;          // slices of partial records per feature
    const int j = tid % WM, s = tid / WM;
    const double* part = c.partials + (size_t)fresh * c.maxb * WM * 3;
    Mom acc = {0.0, 0.0, 0.0};
    for (int b = s; b < c.nparts; b += S) {
      const double* q = part + ((size_t)b * WM + j) * 3;
      Mom m = {q[0], q[1], q[2]};
      acc = mom_merge(acc, m);
    }
    red[(s * WM + j) * 3 + 0] = acc.n;
    red[(s * WM + j) * 3 + 1] = acc.mean;
    red[(s * WM + j) * 3 + 2] = acc.m2;
    __syncthreads();
    if (s == 0) {
      Mom t = {0.0, 0.0, 0.0};
      for (int q = 0; q < S; ++q) {
        Mom m = {red[(q * WM + j) * 3 + 0], red[(q * WM + j) * 3 + 1], red[(q * WM + j) * 3 + 2]};
        t = mom_merge(t, m);
      }
      const int F = cellk_feat(c, fresh);
      double scale = 0.0, shift = 0.0;
      if (j < F) {
        const double var_b = t.m2 / t.n;                         // biased variance normalises the batch
        const double g = c.gb[(fresh * 2 + 0) * WM + j], be = c.gb[(fresh * 2 + 1) * WM + j];
        scale = g / sqrt(var_b + NIS_BN_EPS);
        shift = be - t.mean * scale;
        if (blockIdx.x == 0) {
          double* run = c.running + cellk_running_off(c, fresh);
          const double unb = t.n > 1.0 ? t.m2 / (t.n - 1.0) : var_b;   // running_var takes the unbiased one
          run[j] = (1.0 - NIS_BN_MOM) * run[j] + NIS_BN_MOM * t.mean;
          run[F + j] = (1.0 - NIS_BN_MOM) * run[F + j] + NIS_BN_MOM * unb;
        }
      }
      lds_coef[(fresh * 2 + 0) * WM + j] = scale;
      lds_coef[(fresh * 2 + 1) * WM + j] = shift;
      if (blockIdx.x == 0) {
        c.coef[(fresh * 2 + 0) * WM + j] = scale;
        c.coef[(fresh * 2 + 1) * WM + j] = shift;
      }
    }
  }
  __syncthreads();
}

// h <- W(WMxWM, row-major, zero padded) * a
template <int WM>
__device__ __forceinline__ void dense(nis_cptr W, const double (&a)[WM], double (&b)[WM]) {
#pragma unroll
  for (int j = 0; j < WM; ++j) {
    double acc = 0.0;
#pragma unroll
    for (int i = 0; i < WM; ++i) acc = fma(W[j * WM + i], a[i], acc);
    b[j] = acc;
  }
}

template <int WM>
__device__ __forceinline__ void load_passthrough(const CellK& c, const double* planes, long ld, long i, double (&a)[WM]) {
#pragma unroll
  for (int p = 0; p < WM; ++p) a[p] = (p < c.P) ? planes[(size_t)c.in_plane[p < NIS_MAX_DIM ? p : 0] * ld + i] : 0.0;
}

// Conditioner up to (and excluding) BN layer `upto`: returns the pre-BN activations of that layer
// in `a` (upto == 0: the raw pass-through block).  For the full hidden stack call with upto = L
// and then apply BN_L + ReLU (hidden_stack does that when `finish`).
template <int WM>
__device__ __forceinline__ void hidden_stack(const CellK& c, const double* lds_coef, int upto, bool finish, double (&a)[WM]) {
  double b[WM];
  nis_cptr wp = as_const(c.wp);
  for (int l = 0; l < upto; ++l) {
    const double* sc = lds_coef + (l * 2 + 0) * WM;
    const double* sh = lds_coef + (l * 2 + 1) * WM;
#pragma unroll
    for (int j = 0; j < WM; ++j) {
      const double v = fma(a[j], sc[j], sh[j]);
      a[j] = (l == 0) ? v : fmax(v, 0.0);                 // BN_0 has no ReLU behind it (coupling_cells.py:236-239)
    }
    dense<WM>(wp + (size_t)l * WM * WM, a, b);
#pragma unroll
    for (int j = 0; j < WM; ++j) a[j] = b[j];
  }
  if (finish) {
    const double* sc = lds_coef + (upto * 2 + 0) * WM;
    const double* sh = lds_coef + (upto * 2 + 1) * WM;
#pragma unroll
    for (int j = 0; j < WM; ++j) {
      const double v = fma(a[j], sc[j], sh[j]);
      a[j] = (upto == 0) ? v : fmax(v, 0.0);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Batch statistics of BN layer `layer` (0..L): partial (count, mean, M2) per workgroup & feature.
template <int WM>
__global__ __launch_bounds__(NIS_TPB) void k_bn_stats(CellK c, const double* __restrict__ planes, long ld, long N, int layer) {
  __shared__ double lds_coef[NIS_MAX_HIDDEN * 2 * WM];
  __shared__ double red[3 * NIS_TPB];
  __shared__ double wshift[NIS_TPB / NIS_WAVE][WM];
  __shared__ double wmom[NIS_TPB / NIS_WAVE][WM][3];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  load_coefs<WM>(c, layer, layer - 1, lds_coef, red);

  double S1[WM], S2[WM];
#pragma unroll
  for (int j = 0; j < WM; ++j) S1[j] = S2[j] = 0.0;
  double cnt = 0.0;
  const long stride = (long)gridDim.x * NIS_TPB;
  bool firstit = true;
  for (long i = (long)blockIdx.x * NIS_TPB + tid; i < N; i += stride) {
    double a[WM];
    load_passthrough<WM>(c, planes, ld, i, a);
    hidden_stack<WM>(c, lds_coef, layer, false, a);
    if (firstit) {                      // wave-uniform shift = lane 0's first value (lane 0 owns the lowest index)
#pragma unroll
      for (int j = 0; j < WM; ++j) {
        const double s0 = wave_first(a[j]);
        if (lane == 0) wshift[wave][j] = s0;
      }
      firstit = false;
    }
#pragma unroll
    for (int j = 0; j < WM; ++j) {
      const double t = a[j] - wshift[wave][j];
      S1[j] += t;
      S2[j] = fma(t, t, S2[j]);
    }
    cnt += 1.0;
  }
  // wave -> workgroup -> one record per feature
  const double nw = wave_sum(cnt);
#pragma unroll
  for (int j = 0; j < WM; ++j) {
    const double s1 = wave_sum(S1[j]);
    const double s2 = wave_sum(S2[j]);
    if (lane == 0) {
      if (nw > 0.0) {
        wmom[wave][j][0] = nw;
        wmom[wave][j][1] = wshift[wave][j] + s1 / nw;
        wmom[wave][j][2] = fmax(s2 - s1 * s1 / nw, 0.0);
      } else {
        wmom[wave][j][0] = wmom[wave][j][1] = wmom[wave][j][2] = 0.0;
      }
    }
  }
  __syncthreads();
  if (tid < WM) {
    Mom t = {0.0, 0.0, 0.0};
    for (int w = 0; w < NIS_TPB / NIS_WAVE; ++w) {
      Mom m = {wmom[w][tid][0], wmom[w][tid][1], wmom[w][tid][2]};
      t = mom_merge(t, m);
    }
    double* q = c.partials + (((size_t)layer * c.maxb + blockIdx.x) * WM + tid) * 3;
    q[0] = t.n;
    q[1] = t.mean;
    q[2] = t.m2;
  }
}

// ---------------------------------------------------------------------------------------------
// z = bias[row] + <wlast[row,:], a>
template <int WM>
__device__ __forceinline__ double last_dot(nis_cptr wl, nis_cptr bias, int row, const double (&a)[WM]) {
  double acc = bias[row];
  nis_cptr w = wl + (size_t)row * WM;
#pragma unroll
  for (int i = 0; i < WM; ++i) acc = fma(w[i], a[i], acc);
  return acc;
}

// torch.lerp(start, end, weight) as the CPU kernel evaluates it (used at coupling_cells.py:224)
__device__ __forceinline__ double lerp_torch(double a, double b, double w) {
  return (w < 0.5) ? a + w * (b - a) : b - (b - a) * (1.0 - w);
}

// Piecewise-quadratic transform of one column (coupling_cells.py:167-225).  V: K+1 vertex heights
// (raw exp), W: K bin widths (raw exp).  Returns y; *jf = d y / d x; *bin = bin index.
template <int KM>
__device__ __forceinline__ double pwquad_column(double xb, int K, double (&V)[KM + 1], double (&W)[KM], double* jf, int* bin) {
  xb = (xb > 1.0 - 1e-6) ? 1.0 - 1e-6 : xb;                       // :167
  double wn = 0.0;
#pragma unroll
  for (int k = 0; k < KM; ++k)
    if (k < K) wn += W[k];                                        // Wsum[-1]  :180-182
  // bin search on the normalised cumulative widths, "first maximum of [1e-30, Wsum_k*(Wsum_k<=x)]"  :198-202
  double run = 0.0, best = 1e-30, left = 0.0;
  int idx = 0;
#pragma unroll
  for (int k = 0; k < KM; ++k)
    if (k < K) {
      run += W[k];
      const double cs = run / wn;                                 // Wsum/Wnorms :186
      const double v = (cs > xb) ? 0.0 : cs;
      if (v > best) {
        best = v;
        idx = k + 1;
        left = cs;                                                // Wsum2[idx]
      }
    }
  if (idx > K - 1) idx = K - 1;
#pragma unroll
  for (int k = 0; k < KM; ++k)
    if (k < K) W[k] = W[k] / wn;                                  // :184
  double area = 0.0;
#pragma unroll
  for (int k = 0; k < KM; ++k)
    if (k < K) area += (V[k] + V[k + 1]) / 2 * W[k];              // :194
#pragma unroll
  for (int k = 0; k < KM + 1; ++k)
    if (k <= K) V[k] = V[k] / area;                               // :196-197
  double vw = 0.0, shift = 0.0, Wb = 1.0, Vb = 0.0, Vb1 = 0.0;
#pragma unroll
  for (int k = 0; k < KM; ++k)
    if (k < K) {
      if (k == idx) {
        shift = vw;                                               // VW[idx]  :209-211
        Wb = W[k];
        Vb = V[k];
        Vb1 = V[k + 1];
      }
      vw += (V[k] + V[k + 1]) / 2 * W[k];
    }
  const double alpha = (xb - left) / Wb;                          // :206-207
  const double y1 = (alpha * alpha) / 2 * ((Vb1 - Vb) * Wb);      // :213-214
  const double y2 = (alpha * Vb) * Wb;                            // :216-217
  *jf = lerp_torch(Vb, Vb1, alpha);                               // :224-225
  *bin = idx;
  return y1 + y2 + shift;                                         // :220
}

// Full cell: conditioner with finalised BN + bijector, in place on the transformed planes.
template <int KIND, int WM, int KM>
__global__ __launch_bounds__(NIS_TPB) void k_cell_apply(CellK c, double* __restrict__ planes, long ld, long N, int bn_train,
                                                        int* __restrict__ bins_out) {
  __shared__ double lds_coef[(NIS_MAX_HIDDEN + 1) * 2 * WM];
  __shared__ double red[3 * NIS_TPB];
  const int tid = threadIdx.x;
  load_coefs<WM>(c, c.L + 1, bn_train ? c.L : -1, lds_coef, red);
  nis_cptr wl = as_const(c.wlast);
  nis_cptr bias = as_const(c.bias);
  const long stride = (long)gridDim.x * NIS_TPB;
  for (long i = (long)blockIdx.x * NIS_TPB + tid; i < N; i += stride) {
    double a[WM];
    load_passthrough<WM>(c, planes, ld, i, a);
    hidden_stack<WM>(c, lds_coef, c.L, true, a);
    double jprod = 1.0;
    for (int t = 0; t < c.T; ++t) {
      double* col = planes + (size_t)c.in_plane[c.P + t] * ld;
      const double xb = col[i];
      const int row0 = t * c.out;
      double y, jf;
      int bin;
      if (KIND == NIS_PWQUAD) {
        double V[KM + 1], W[KM];
#pragma unroll
        for (int k = 0; k < KM + 1; ++k) V[k] = (k <= c.K) ? exp(last_dot<WM>(wl, bias, row0 + k, a)) : 0.0;     // :173,189
#pragma unroll
        for (int k = 0; k < KM; ++k) W[k] = (k < c.K) ? exp(last_dot<WM>(wl, bias, row0 + c.K + 1 + k, a)) : 0.0;  // :175,178
        y = pwquad_column<KM>(xb, c.K, V, W, &jf, &bin);
      } else {
        // PWLin: equal-width bins -> the index is known before the heights (coupling_cells.py:126-133),
        // so the K heights are streamed: total, prefix below the bin, and the bin's own height.
        const double ak = xb * c.K;
        const double bf = floor(ak);
        const double alpha = (ak - bf) / c.K;
        bin = (int)bf;
        double tot = 0.0, below = 0.0, qb = 0.0;
        for (int k = 0; k < c.K; ++k) {
          const double e = exp(last_dot<WM>(wl, bias, row0 + k, a));                 // :115
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
      col[i] = y;
      jprod *= jf;
      if (bins_out) bins_out[(size_t)(c.bins_row + t) * ld + i] = bin;
    }
    double* jp = planes + (size_t)c.jplane * ld;
    const double jin = c.first ? 1.0 : jp[i];
    jp[i] = jin * jprod;
  }
}

// Host-side launchers, instantiated once per WM in flow_wm.hip.
template <int WM>
hipError_t launch_stats(const CellK& c, const double* planes, long ld, long N, int layer, int grid, hipStream_t s);
template <int WM>
hipError_t launch_apply(const CellK& c, double* planes, long ld, long N, int bn_train, int km, int* bins, int grid, hipStream_t s);
#define NIS_DECL_WM(W)                                                                                             \
  template <>                                                                                                      \
  hipError_t launch_stats<W>(const CellK& c, const double* planes, long ld, long N, int layer, int grid, hipStream_t s); \
  template <>                                                                                                      \
  hipError_t launch_apply<W>(const CellK& c, double* planes, long ld, long N, int bn_train, int km, int* bins, int grid, hipStream_t s);
NIS_DECL_WM(4)
NIS_DECL_WM(8)
NIS_DECL_WM(16)
NIS_DECL_WM(32)
NIS_DECL_WM(64)
#undef NIS_DECL_WM
