// Hand-written adjoint (backward) of a coupling cell with a narrow conditioner (register width WM = 4 or 8, any
// number of transformed columns, PWQuad n_bins <= 32, PWLin any n_bins, AffineCoupling), train- or eval-mode BatchNorm.  Wider
// conditioners (WM = 16 / 32 / 64) run the same chain on the f64 matrix cores: flow_bwd_wm.hip.
//
// Reverse of:  BN_0 -> Lin_0 -> [BN_l -> ReLU -> Lin_l]_{l=1..L} -> exp / normalise -> quadratic (linear) CDF, J *= dy/dx
// (reference forward: normalizing_flows/layers/coupling_cells.py:107-142, :159-228, :236-254; the reference obtains these
// gradients from torch.autograd at manager.py:275-278).
//
// Train-mode BatchNorm couples all samples of a minibatch, so the adjoint is again a chain of launches, one per BN layer, in reverse:
//   head    : recompute the cell; per transformed column the bijector adjoint -> dZ -> d a_L; dy_L (grad at BN_L's output) +
//             the two batch sums BN_L's backward needs.  d = 2 class (PWQuad, 4 bins, one transformed column): dW_L = dZ^T a_L and
//             db_L accumulate in registers of the same launch (FUSEW); otherwise dZ / a_L go to planes and k_bwd_wlast reduces them
//   layer l : dh_l = scale_l (dy_l - mean(dy_l) - xhat_l mean(dy_l xhat_l)) -> dW_{l-1}, dy_{l-1}, sums of BN_{l-1}
//   input   : BN_0 backward -> gradient of the pass-through columns
//   finish  : every record of the chain -> the gradient blob in the reference's parameter order (one launch per cell)
// The G minibatches of an optimizer step (manager.py:219-270: the loss is their mean, ONE backward) are blockIdx.y of every
// launch (blockIdx.z in the matrix-core chain): own history planes, own BatchNorm records, own batch sums; the parameter gradient is
// the sum over all of them.
// A consumer merges the batch sums of its producer's per-workgroup records in its prologue (fixed order: bitwise reproducible),
// so no reduction launch sits between the links of the chain.
#include <algorithm>
#include <cstring>

#include "flow_bwd.h"

#ifndef NIS_BWD_LAYER_WAVES
#define NIS_BWD_LAYER_WAVES 1     // waves per SIMD the 3-wide layer kernel is allocated for (1: unconstrained, 129 VGPRs = three waves; 4: 128 VGPRs with 14 dwords of scratch measured 15 % slower)
#endif
int nis_grid_for(int blocks_per_cu, int G, int64_t N, int tpb);      // api.hip
int nis_env_int(const char* name, int dflt);

// per-thread accumulators -> slots [0, NACC) of this workgroup's record
template <int NACC>
__device__ __forceinline__ void bw_block_reduce(const double (&acc)[NACC], double* rec) {
  __shared__ double sm[NIS_TPB / 64][NACC];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int a = 0; a < NACC; ++a) {
    const double v = wave_sum(acc[a]);
    if (lane == 0) sm[wave][a] = v;
  }
  __syncthreads();
  if ((int)threadIdx.x < NACC) {
    double t = 0.0;
    for (int w = 0; w < NIS_TPB / 64; ++w) t += sm[w][threadIdx.x];
    rec[threadIdx.x] = t;
  }
  __syncthreads();
}

// Prologue of every chain kernel: this minibatch's saved BatchNorm records -> LDS (lbn: (L+1) x 4 x WM  scale | shift | mean | rstd),
// and the producer's batch sums (2 WM values at k.prev_off of its records) merged in a fixed order -> lsum = sums / N.
template <int WM>
__device__ __forceinline__ void bw_prologue(const BwdK& k, int g, double* lbn, double* lsum) {
  __shared__ double red[NIS_TPB];
  const int nb = (k.c.L + 1) * 4 * WM;
  const double* bn = k.bn + (size_t)g * k.bn_gs;
  for (int i = threadIdx.x; i < nb; i += NIS_TPB) lbn[i] = bn[i];
  if (k.prev) {
    constexpr int NV = 2 * WM, S = NIS_TPB / NV;
    const int j = threadIdx.x % NV, sl = threadIdx.x / NV;
    const double* p = k.prev + (size_t)g * k.nprev * BNACC + k.prev_off + j;
    // every workgroup of the launch pays this merge before its first sample: eight record loads in flight per thread (the chain of
    // one-at-a-time round trips cost ~25 us per launch at 1024 records), fixed order
    double t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] = 0.0;
    for (int b = sl; b < k.nprev; b += 8 * S) {
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = (b + u * S < k.nprev) ? p[(size_t)(b + u * S) * BNACC] : 0.0;
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] += v[u];
    }
    red[sl * NV + j] = ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
    __syncthreads();
#pragma unroll
    for (int h = S / 2; h > 0; h >>= 1) {
      if (sl < h) red[sl * NV + j] += red[(sl + h) * NV + j];
      __syncthreads();
    }
    if (sl == 0) lsum[j] = red[j] * k.invN;
  }
  __syncthreads();
}

// forward recompute of the hidden stack with the SAVED statistics; returns for layer `upto` (1..L):
//   h[]   pre-BN activations of BN_upto        a_in[] input of Linear_{upto-1} (post-ReLU of BN_{upto-1}; BN_0 output if upto == 1)
//   o_in[] BN_{upto-1} output before ReLU      xh_in[] normalised input of BN_{upto-1}
template <int WM>
__device__ __forceinline__ void bw_load_x(const BwdK& k, const double* __restrict__ x, long i, double (&cur)[WM]) {
#pragma unroll
  for (int p = 0; p < WM; ++p) cur[p] = (p < k.c.P) ? x[(size_t)k.c.in_plane[p] * k.ld + i] : 0.0;
}
template <int WM, int WX>
__device__ __forceinline__ void bw_forward(const BwdK& k, const double (&xin)[WM], const double* lbn, int upto, double (&h)[WM],
                                           double (&a_in)[WM], double (&o_in)[WM], double (&xh_in)[WM]) {
  nis_cptr wp = as_const(k.c.wp);
  double cur[WM];
#pragma unroll
  for (int p = 0; p < WM; ++p) cur[p] = xin[p];
  for (int l = 0; l < upto; ++l) {
#pragma unroll
    for (int j = 0; j < WM; ++j) {
      xh_in[j] = (cur[j] - lbn[(l * 4 + 2) * WM + j]) * lbn[(l * 4 + 3) * WM + j];
      o_in[j] = fma(cur[j], lbn[(l * 4 + 0) * WM + j], lbn[(l * 4 + 1) * WM + j]);
      a_in[j] = (l == 0) ? o_in[j] : fmax(o_in[j], 0.0);
    }
    dense<WM, WX>(wp + (size_t)l * WM * WM, a_in, cur);
  }
#pragma unroll
  for (int j = 0; j < WM; ++j) h[j] = cur[j];
}

// ---------------------------------------------------------------------------------------------
// WX: exact common width of the hidden layers (3 for the README / C1 / C2 nets), else WM.  FUSEW (PWQuad, K == KM, T == 1): the
// last Linear's weight / bias gradients accumulate here, record slots [2 WM + r (WM + 1) + q], nothing but dy goes to planes.
template <int KIND, int WM, int KM, int WX, bool FUSEW>
__global__ __launch_bounds__(NIS_TPB) void k_cell_bwd_head(BwdK k) {
  __shared__ double etab[NIS_EXP_N];
  __shared__ double lbn[(NIS_MAX_HIDDEN + 1) * 4 * WM];
  __shared__ double lsum[2 * WM];
  const int g = blockIdx.y;
  nis_exp_init(etab);
  bw_prologue<WM>(k, g, lbn, lsum);
  const CellK& c = k.c;
  nis_cptr wl = as_const(c.wlast);
  nis_cptr bias = as_const(c.bias);
  const int L = c.L, K = FUSEW ? KM : c.K;
  const double* __restrict__ x = k.x + (size_t)g * k.x_gs;
  const double* __restrict__ xo = k.xo + (size_t)g * k.x_gs;
  double* __restrict__ gp = k.g + (size_t)g * k.g_gs;
  double* __restrict__ dyp = k.dyp + (size_t)g * k.wsp_gs;
  double* __restrict__ alp = k.alp + (size_t)g * k.wsp_gs;
  double* __restrict__ dzp = k.dzp + (size_t)g * k.wsp_gs;
  double acc[2 * WM];
#pragma unroll
  for (int a = 0; a < 2 * WM; ++a) acc[a] = 0.0;
  constexpr int NW = FUSEW ? (2 * KM + 1) * (WM + 1) : 1;
  double accw[NW];
#pragma unroll
  for (int a = 0; a < NW; ++a) accw[a] = 0.0;
  const long stride = (long)gridDim.x * NIS_TPB;
  // FUSEW (one transformed column): the five values a sample needs from memory are requested one sample AHEAD -- at its two waves
  // per SIMD the kernel sat out a full HBM round trip per sample otherwise (PMC: 47 % of the fp64 issue rate).  The last round re-reads
  // its own sample; the in-place stores of sample i never touch what was requested for sample i + stride.
  const long i_first = (long)blockIdx.x * NIS_TPB + threadIdx.x;
  const size_t pl_j = (size_t)c.jplane * k.ld, pl_t = (size_t)c.in_plane[c.P] * k.ld;
  double xq[WM], xrq = 0.0, gyq = 0.0, gJq = 0.0, Joq = 0.0;
#pragma unroll
  for (int p = 0; p < WM; ++p) xq[p] = 0.0;
  if (FUSEW && i_first < k.N) {
    bw_load_x<WM>(k, x, i_first, xq);
    xrq = x[pl_t + i_first];
    gyq = gp[pl_t + i_first];
    gJq = gp[pl_j + i_first];
    Joq = xo[pl_j + i_first];
  }
  for (long i = i_first; i < k.N; i += stride) {
    double h[WM], a_in[WM], o_in[WM], xh_in[WM], xin[WM];
    const double xrc = xrq, gyc = gyq, gJc = gJq, Joc = Joq;
    if (FUSEW) {
#pragma unroll
      for (int p = 0; p < WM; ++p) xin[p] = xq[p];
      const long ic = i + stride < k.N ? i + stride : i;
      bw_load_x<WM>(k, x, ic, xq);
      xrq = x[pl_t + ic];
      gyq = gp[pl_t + ic];
      gJq = gp[pl_j + ic];
      Joq = xo[pl_j + ic];
    } else {
      bw_load_x<WM>(k, x, i, xin);
    }
    bw_forward<WM, WX>(k, xin, lbn, L, h, a_in, o_in, xh_in);
    double aL[WM], oL[WM], xhL[WM], da[WM];
#pragma unroll
    for (int j = 0; j < WM; ++j) {
      xhL[j] = (h[j] - lbn[(L * 4 + 2) * WM + j]) * lbn[(L * 4 + 3) * WM + j];
      oL[j] = fma(h[j], lbn[(L * 4 + 0) * WM + j], lbn[(L * 4 + 1) * WM + j]);
      aL[j] = fmax(oL[j], 0.0);
      if (!FUSEW) alp[(size_t)j * k.ldw + i] = aL[j];
      da[j] = 0.0;
    }
    const double gJ = FUSEW ? gJc : gp[(size_t)c.jplane * k.ld + i];
    const double Jout = FUSEW ? Joc : xo[(size_t)c.jplane * k.ld + i];
    double jfprod = 1.0;
    for (int t = 0; t < c.T; ++t) {
      const int colp = c.in_plane[c.P + t];
      const int row0 = t * c.out;
      const double xraw = FUSEW ? xrc : x[(size_t)colp * k.ld + i];
      const double gy = FUSEW ? gyc : gp[(size_t)colp * k.ld + i];
      double gx = 0.0, jf = 1.0;
#define NIS_GV(q, wm1, wq) (((q) >= 1 && (q) - 1 < b ? hgy * (wm1) : 0.0) + ((q) < b ? hgy * (wq) : 0.0) + ((q) == b ? A0 : 0.0) + ((q) == b + 1 ? A1 : 0.0))
#define NIS_GW0(q) (((q) < b ? hgy * (V[q] + V[(q) + 1]) + gleft : 0.0) + ((q) == b ? B0 : 0.0))
      if constexpr (KIND == NIS_PWQUAD && KM > 32) {
        // 33 .. NIS_MAX_BINS bins: the same adjoint with the height / width arrays in per-thread scratch memory and runtime loops
        const bool clamped = xraw > 1.0 - 1e-6;
        const double xb = clamped ? 1.0 - 1e-6 : xraw;
        double V[NIS_KM_BIG + 1], W[NIS_KM_BIG];
        for (int q = 0; q <= K; ++q) V[q] = nis_exp(last_dot<WM, WX>(wl, bias, row0 + q, aL), etab);
        for (int q = 0; q < K; ++q) W[q] = nis_exp(last_dot<WM, WX>(wl, bias, row0 + K + 1 + q, aL), etab);
        double wn = 0.0;
        for (int q = 0; q < K; ++q) wn += W[q];
        int b;
        double left;
        pwquad_bin_big(xb, K, W, wn, &b, &left);
        left = left / wn;
        const double iwn = 1.0 / wn;
        double area = 0.0;
        for (int q = 0; q < K; ++q) {
          W[q] = W[q] * iwn;
          area += (V[q] + V[q + 1]) / 2 * W[q];
        }
        const double iarea = 1.0 / area;
        for (int q = 0; q <= K; ++q) V[q] = V[q] * iarea;
        const double Wb = W[b], Vb = V[b], Vb1 = V[b + 1];
        const double alpha = (xb - left) / Wb;
        const double D = Vb1 - Vb;
        jf = lerp_torch(Vb, Vb1, alpha);
        const double gjf = gJ * Jout / jf;
        const double galpha = gy * jf * Wb + gjf * D;
        gx = clamped ? 0.0 : galpha / Wb;
        const double gleft = -galpha / Wb;
        const double hgy = gy / 2;
        const double A0 = gy * alpha * Wb * (1.0 - alpha / 2) + gjf * (1.0 - alpha);
        const double A1 = gy * alpha * alpha / 2 * Wb + gjf * alpha;
        const double B0 = gy * (alpha * alpha / 2 * D + alpha * Vb) - galpha * alpha / Wb;
        double S = 0.0;
        for (int q = 0; q <= K; ++q) {
          const double wm1 = q >= 1 ? W[q - 1] : 0.0;
          const double wq = q < K ? W[q] : 0.0;
          S += NIS_GV(q, wm1, wq) * V[q];
        }
        double Sw = 0.0;
        for (int q = 0; q < K; ++q) Sw += (NIS_GW0(q) - S * (V[q] + V[q + 1]) / 2) * W[q];
        double wprev = 0.0;
        for (int q = 0; q <= K; ++q) {
          const bool hasw = q < K;
          const double wq = hasw ? W[q] : 0.0;
          const double vq = V[q];
          const double dzv = NIS_GV(q, wprev, wq) * vq - S * vq * (wprev + wq) / 2;
          if (hasw) W[q] = wq * (NIS_GW0(q) - S * (vq + V[q + 1]) / 2 - Sw);
          V[q] = dzv;
          wprev = wq;
        }
        for (int q = 0; q <= K; ++q) {
          const double dz = V[q];
          dzp[(size_t)(row0 + q) * k.ldw + i] = dz;
#pragma unroll
          for (int j = 0; j < WX; ++j) da[j] = fma(wl[(size_t)(row0 + q) * WM + j], dz, da[j]);
        }
        for (int q = 0; q < K; ++q) {
          const double dz = W[q];
          dzp[(size_t)(row0 + K + 1 + q) * k.ldw + i] = dz;
#pragma unroll
          for (int j = 0; j < WX; ++j) da[j] = fma(wl[(size_t)(row0 + K + 1 + q) * WM + j], dz, da[j]);
        }
      } else if (KIND == NIS_PWQUAD) {
        const bool clamped = xraw > 1.0 - 1e-6;
        const double xb = clamped ? 1.0 - 1e-6 : xraw;
        double V[KM + 1], W[KM];                           // KM <= 32 here (the branch above took the scratch-array variant)
#pragma unroll
        for (int q = 0; q < KM + 1; ++q) V[q] = (q <= K) ? nis_exp(last_dot<WM, WX>(wl, bias, row0 + q, aL), etab) : 0.0;
#pragma unroll
        for (int q = 0; q < KM; ++q) W[q] = (q < K) ? nis_exp(last_dot<WM, WX>(wl, bias, row0 + K + 1 + q, aL), etab) : 0.0;
        // forward quantities exactly as k_cell_apply forms them
        double wn = 0.0;
#pragma unroll
        for (int q = 0; q < KM; ++q)
          if (q < K) wn += W[q];
        int b;
        double left;
        pwquad_bin<KM, FUSEW>(xb, K, W, wn, &b, &left);
        const double iwn = 1.0 / wn;
        double area = 0.0;
#pragma unroll
        for (int q = 0; q < KM; ++q)
          if (q < K) {
            W[q] = W[q] * iwn;
            area += (V[q] + V[q + 1]) / 2 * W[q];
          }
        const double iarea = 1.0 / area;
#pragma unroll
        for (int q = 0; q < KM + 1; ++q)
          if (q <= K) V[q] = V[q] * iarea;
        double Wb = 1.0, Vb = 0.0, Vb1 = 0.0;
#pragma unroll
        for (int q = 0; q < KM; ++q)
          if (q < K) {
            const bool hit = q == b;
            Wb = hit ? W[q] : Wb;
            Vb = hit ? V[q] : Vb;
            Vb1 = hit ? V[q + 1] : Vb1;
          }
        const double alpha = (xb - left) / Wb;
        const double D = Vb1 - Vb;
        jf = lerp_torch(Vb, Vb1, alpha);
        // ---- adjoint.  J_out = J_in * prod_t jf_t  ->  dL/d jf_t = gJ * J_out / jf_t
        const double gjf = gJ * Jout / jf;
        const double galpha = gy * jf * Wb + gjf * D;
        gx = clamped ? 0.0 : galpha / Wb;
        const double gleft = -galpha / Wb;
        // Gradients w.r.t. the normalised heights v_q and widths w_q are sparse in q (bins below the hit, the hit, its
        // right edge): re-formed on the fly in each of the three passes instead of being kept in arrays
        //   gv(q) = [q-1 < b] gy w_{q-1}/2 + [q < b] gy w_q/2 + [q == b] A0 + [q == b+1] A1
        //   gw0(q) = [q < b] (gy (v_q + v_{q+1})/2 + gleft) + [q == b] B0
        const double hgy = gy / 2;
        const double A0 = gy * alpha * Wb * (1.0 - alpha / 2) + gjf * (1.0 - alpha);
        const double A1 = gy * alpha * alpha / 2 * Wb + gjf * alpha;
        const double B0 = gy * (alpha * alpha / 2 * D + alpha * Vb) - galpha * alpha / Wb;
        double S = 0.0;                                    // through v = eV / area: S = sum_q gv(q) v_q
#pragma unroll
        for (int q = 0; q < KM + 1; ++q)
          if (q <= K) {
            const double wm1 = q >= 1 ? W[q >= 1 ? q - 1 : 0] : 0.0;
            const double wq = (q < KM && q < K) ? W[q < KM ? q : 0] : 0.0;
            S += NIS_GV(q, wm1, wq) * V[q];
          }
        double Sw = 0.0;                                   // through w = softmax: Sw = sum_q gw(q) w_q
#pragma unroll
        for (int q = 0; q < KM; ++q)
          if (q < K) Sw += (NIS_GW0(q) - S * (V[q] + V[q + 1]) / 2) * W[q];
        // dZ in place (height rows into V, width rows into W), then rows -> planes and d a_L += W_L^T dZ
        double wprev = 0.0;
#pragma unroll
        for (int q = 0; q < KM + 1; ++q)
          if (q <= K) {
            const bool hasw = q < KM && q < K;
            const double wq = hasw ? W[q < KM ? q : 0] : 0.0;
            const double vq = V[q];
            const double dzv = NIS_GV(q, wprev, wq) * vq - S * vq * (wprev + wq) / 2;
            if (hasw) W[q < KM ? q : 0] = wq * (NIS_GW0(q) - S * (vq + V[q + 1]) / 2 - Sw);
            V[q] = dzv;
            wprev = wq;
          }
#pragma unroll
        for (int q = 0; q < KM + 1; ++q)
          if (q <= K) {
            const double dz = V[q];
            if (FUSEW) {
#pragma unroll
              for (int j = 0; j < WX; ++j) accw[q * (WM + 1) + j] = fma(dz, aL[j], accw[q * (WM + 1) + j]);
              accw[q * (WM + 1) + WM] += dz;
            } else {
              dzp[(size_t)(row0 + q) * k.ldw + i] = dz;
            }
#pragma unroll
            for (int j = 0; j < WX; ++j) da[j] = fma(wl[(size_t)(row0 + q) * WM + j], dz, da[j]);
          }
#pragma unroll
        for (int q = 0; q < KM; ++q)
          if (q < K) {
            const double dz = W[q];
            if (FUSEW) {
#pragma unroll
              for (int j = 0; j < WX; ++j) accw[(KM + 1 + q) * (WM + 1) + j] = fma(dz, aL[j], accw[(KM + 1 + q) * (WM + 1) + j]);
              accw[(KM + 1 + q) * (WM + 1) + WM] += dz;
            } else {
              dzp[(size_t)(row0 + K + 1 + q) * k.ldw + i] = dz;
            }
#pragma unroll
            for (int j = 0; j < WX; ++j) da[j] = fma(wl[(size_t)(row0 + K + 1 + q) * WM + j], dz, da[j]);
          }
      } else if (KIND == NIS_AFFINE) {
        // AffineCoupling (coupling_cells.py:56-68): rows [scale | shift][T]; u = 20 e^z0 x + relu(z1), y = atan(u) 2/pi,
        // jf = 20 e^z0 / (u^2 + 1)  (the cell-wide factor 2/pi of the Jacobian is applied once, below)
        const int rs = t, rh = c.T + t;
        const double z1 = last_dot<WM, WX>(wl, bias, rh, aL);
        const double s0 = 20.0 * nis_exp(last_dot<WM, WX>(wl, bias, rs, aL), etab);
        const double u = xraw * s0 + (z1 > 0.0 ? z1 : 0.0);
        const double q = 1.0 / (u * u + 1.0);
        jf = s0 * q;
        const double gjf = gJ * Jout / jf;
        const double gu = gy * (q / (M_PI / 2)) - gjf * (2.0 * u * s0 * q * q);
        gx = gu * s0;
        const double dz0 = (gu * xraw + gjf * q) * s0;
        const double dz1 = z1 > 0.0 ? gu : 0.0;
        dzp[(size_t)rs * k.ldw + i] = dz0;
        dzp[(size_t)rh * k.ldw + i] = dz1;
#pragma unroll
        for (int j = 0; j < WX; ++j) da[j] = fma(wl[(size_t)rs * WM + j], dz0, fma(wl[(size_t)rh * WM + j], dz1, da[j]));
      } else {
        // PWLin: Q_q = K e_q / tot, y = Q_b alpha + sum_{q<b} e_q / tot, jf = Q_b, alpha = x - b/K  (two streamed passes)
        const double ak = xraw * K;
        const double bf = floor(ak);
        const double alpha = (ak - bf) / K;
        const int b = (int)bf;
        double tot = 0.0, below = 0.0, eb = 0.0;
        for (int q = 0; q < K; ++q) {
          const double e = nis_exp(last_dot<WM, WX>(wl, bias, row0 + q, aL), etab);
          const bool hit = q == b;
          below = hit ? tot : below;
          eb = hit ? e : eb;
          tot += e;
        }
        const double Qb = eb / (tot / K);
        const double Qs = below / tot;
        jf = Qb;
        const double gjf = gJ * Jout / jf;
        const double gQb = gy * alpha + gjf, gQs = gy;
        gx = gy * Qb;
        for (int q = 0; q < K; ++q) {
          const double e = nis_exp(last_dot<WM, WX>(wl, bias, row0 + q, aL), etab);
          const double pq = e / tot;
          const double dz = gQb * ((q == b ? Qb : 0.0) - Qb * pq) + gQs * ((q < b ? pq : 0.0) - Qs * pq);
          dzp[(size_t)(row0 + q) * k.ldw + i] = dz;
#pragma unroll
          for (int j = 0; j < WX; ++j) da[j] = fma(wl[(size_t)(row0 + q) * WM + j], dz, da[j]);
        }
      }
#undef NIS_GV
#undef NIS_GW0
      gp[(size_t)colp * k.ld + i] = gx;
      jfprod *= jf;
    }
    if (KIND == NIS_AFFINE) jfprod *= 1.0 / (M_PI / 2);
    gp[(size_t)c.jplane * k.ld + i] = gJ * jfprod;              // dL/dJ_in = dL/dJ_out * prod_t dy_t/dx_t
    // ---- through ReLU to BN_L's output; sums for BN_L
#pragma unroll
    for (int j = 0; j < WX; ++j) {
      const double dy = (oL[j] > 0.0) ? da[j] : 0.0;
      dyp[(size_t)j * k.ldw + i] = dy;
      acc[j] += dy;
      acc[WM + j] = fma(dy, xhL[j], acc[WM + j]);
    }
  }
  double* rec = k.part + ((size_t)g * gridDim.x + blockIdx.x) * BNACC;
  bw_block_reduce<2 * WM>(acc, rec);
  if (FUSEW) bw_block_reduce<NW>(accw, rec + 2 * WM);
}

// dW_L[r][j] = sum_s dZ[r][s] a_L[j][s],  db_L[r] = sum_s dZ[r][s]   for the BRC rows of chunk blockIdx.y, minibatch blockIdx.z
template <int WM>
__global__ __launch_bounds__(NIS_TPB) void k_bwd_wlast(BwdK k) {
  const int g = blockIdx.z;
  const int r0 = blockIdx.y * BRC;
  const int nr = min(BRC, k.c.R - r0);
  const double* __restrict__ alp = k.alp + (size_t)g * k.wsp_gs;
  const double* __restrict__ dzp = k.dzp + (size_t)g * k.wsp_gs;
  double acc[BRC * (WM + 1)];
#pragma unroll
  for (int a = 0; a < BRC * (WM + 1); ++a) acc[a] = 0.0;
  const long stride = (long)gridDim.x * NIS_TPB;
  for (long i = (long)blockIdx.x * NIS_TPB + threadIdx.x; i < k.N; i += stride) {
    double a[WM];
#pragma unroll
    for (int j = 0; j < WM; ++j) a[j] = alp[(size_t)j * k.ldw + i];
#pragma unroll
    for (int r = 0; r < BRC; ++r)
      if (r < nr) {
        const double dz = dzp[(size_t)(r0 + r) * k.ldw + i];
#pragma unroll
        for (int j = 0; j < WM; ++j) acc[r * (WM + 1) + j] = fma(dz, a[j], acc[r * (WM + 1) + j]);
        acc[r * (WM + 1) + WM] += dz;
      }
  }
  bw_block_reduce<BRC*(WM + 1)>(acc, k.part + (((size_t)g * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * BNACC);
}

// BN_l backward (l >= 1) -> Linear_{l-1} backward -> dy_{l-1}.  Record: [dW_{l-1} (WM x WM) | sum dy_{l-1} | sum dy_{l-1} xhat_{l-1}]
template <int WM, int WX>
__global__ __launch_bounds__(NIS_TPB, (WX == 3) ? NIS_BWD_LAYER_WAVES : 1) void k_cell_bwd_layer(BwdK k, int l) {
  __shared__ double lbn[(NIS_MAX_HIDDEN + 1) * 4 * WM];
  __shared__ double lsum[2 * WM];
  const int g = blockIdx.y;
  bw_prologue<WM>(k, g, lbn, lsum);
  const CellK& c = k.c;
  nis_cptr wp = as_const(c.wp);
  const double* __restrict__ x = k.x + (size_t)g * k.x_gs;
  double* __restrict__ dyp = k.dyp + (size_t)g * k.wsp_gs;
  double acc[WM * WM + 2 * WM];
#pragma unroll
  for (int a = 0; a < WM * WM + 2 * WM; ++a) acc[a] = 0.0;
  const long stride = (long)gridDim.x * NIS_TPB;
  // the inputs of the NEXT sample are requested before the current one is worked on (a thread owns ~5 samples of a 1e6 minibatch and
  // the launch streams 56 B per sample: without this every sample sat out a full HBM round trip); the last round re-reads its own sample
  const long i0 = (long)blockIdx.x * NIS_TPB + threadIdx.x;
  double xq[WM], dyq[WX];
  if (i0 < k.N) {
    bw_load_x<WM>(k, x, i0, xq);
#pragma unroll
    for (int j = 0; j < WX; ++j) dyq[j] = dyp[(size_t)j * k.ldw + i0];
  }
  for (long i = i0; i < k.N; i += stride) {
    double h[WM], a_in[WM], o_in[WM], xh_in[WM], xin[WM], dyc[WX];
#pragma unroll
    for (int p = 0; p < WM; ++p) xin[p] = xq[p];
#pragma unroll
    for (int j = 0; j < WX; ++j) dyc[j] = dyq[j];
    {
      const long ic = i + stride < k.N ? i + stride : i;
      bw_load_x<WM>(k, x, ic, xq);
#pragma unroll
      for (int j = 0; j < WX; ++j) dyq[j] = dyp[(size_t)j * k.ldw + ic];
    }
    bw_forward<WM, WX>(k, xin, lbn, l, h, a_in, o_in, xh_in);
    double dh[WM];
#pragma unroll
    for (int j = 0; j < WM; ++j) {
      const double xh = (h[j] - lbn[(l * 4 + 2) * WM + j]) * lbn[(l * 4 + 3) * WM + j];
      const double dy = j < WX ? dyc[j < WX ? j : 0] : 0.0;
      dh[j] = lbn[(l * 4 + 0) * WM + j] * (dy - lsum[j] - xh * lsum[WM + j]);       // scale_l = gamma_l * rstd_l
    }
    double da[WM];
#pragma unroll
    for (int q = 0; q < WM; ++q) da[q] = 0.0;
#pragma unroll
    for (int j = 0; j < WX; ++j)
#pragma unroll
      for (int q = 0; q < WX; ++q) {
        acc[j * WM + q] = fma(dh[j], a_in[q], acc[j * WM + q]);      // dW_{l-1}[j][q]
        da[q] = fma(wp[(size_t)(l - 1) * WM * WM + j * WM + q], dh[j], da[q]);
      }
#pragma unroll
    for (int q = 0; q < WX; ++q) {
      const double dy = (l - 1 == 0 || o_in[q] > 0.0) ? da[q] : 0.0;   // BN_0 has no ReLU behind it
      dyp[(size_t)q * k.ldw + i] = dy;
      acc[WM * WM + q] += dy;
      acc[WM * WM + WM + q] = fma(dy, xh_in[q], acc[WM * WM + WM + q]);
    }
  }
  bw_block_reduce<WM * WM + 2 * WM>(acc, k.part + ((size_t)g * gridDim.x + blockIdx.x) * BNACC);
}

// BN_0 backward: gradient of the pass-through columns
template <int WM>
__global__ __launch_bounds__(NIS_TPB) void k_cell_bwd_input(BwdK k) {
  __shared__ double lbn[(NIS_MAX_HIDDEN + 1) * 4 * WM];
  __shared__ double lsum[2 * WM];
  const int g = blockIdx.y;
  bw_prologue<WM>(k, g, lbn, lsum);
  const CellK& c = k.c;
  const double* __restrict__ x = k.x + (size_t)g * k.x_gs;
  double* __restrict__ gp = k.g + (size_t)g * k.g_gs;
  const double* __restrict__ dyp = k.dyp + (size_t)g * k.wsp_gs;
  const long stride = (long)gridDim.x * NIS_TPB;
  for (long i = (long)blockIdx.x * NIS_TPB + threadIdx.x; i < k.N; i += stride) {
#pragma unroll
    for (int p = 0; p < WM; ++p)
      if (p < c.P) {
        const size_t at = (size_t)c.in_plane[p] * k.ld + i;
        const double xh = (x[at] - lbn[2 * WM + p]) * lbn[3 * WM + p];
        const double dy = dyp[(size_t)p * k.ldw + i];
        gp[at] += lbn[p] * (dy - lsum[p] - xh * lsum[WM + p]);
      }
  }
}

// Every record of a cell's chain -> gradient blob in the reference's parameter order (include/nis_hip.h), summed over the G
// minibatches and the workgroups in a fixed order.  A workgroup owns 4 blob elements and 64 interleaved slices of the records.
//   rh: head records (G, nparts, BNACC): [sum dy_L | sum dy_L xhat_L | fused: dW_L rows (WM + 1 each: weights | bias)]
//   rl: layer-pass records, pass p = L - l + 1 of layer kernel l at rl + (p - 1) G nparts BNACC: [dW_{l-1} | sum dy_{l-1} | sum dy xhat]
//   rw: k_bwd_wlast records (G, nch, nparts, BNACC), chunk = BRC rows of (WM + 1) -- null in the fused mode
// hidden_bias (NIS_AFFINE): 1 = train-mode BatchNorm (zero gradient), 2 = eval mode (`bn`: saved scale | shift | mean | rstd).
struct BwdFin {
  CellK c;
  int WM, G, nparts, nparts_h, nch, hidden_bias, nelem;      // nparts_h: records per minibatch of the head launch, nparts: of every other
  const double* rh;
  const double* rl;
  const double* rw;
  const double* bn;
  double* gparams;
};
__global__ __launch_bounds__(NIS_TPB) void k_bwd_finish(BwdFin f) {
  __shared__ double sm[64][4];
  const CellK& c = f.c;
  const int el = threadIdx.x & 3, slice = threadIdx.x >> 2;
  const int e = blockIdx.x * 4 + el;
  const int WM = f.WM, L = c.L;
  const size_t lay = (size_t)f.G * f.nparts * BNACC;        // one launch's records
  const double* base = nullptr;      // record 0 of minibatch 0
  size_t gstride = (size_t)f.nparts * BNACC;
  int np = f.nparts;                 // records per minibatch in the selected buffer
  bool in_head = false;
  double mult = 1.0;
  if (e < f.nelem) {
    int off = 0;
    bool found = false;
    for (int l = 0; l <= L && !found; ++l) {
      const int F = cellk_feat(c, l);
      const int O = l < L ? c.width[l] : c.R;
      // d gamma_l = sum dy_l xhat_l, d beta_l = sum dy_l: emitted by the launch that produced dy_l (head: l == L, slot 0; layer kernel l+1: slot WM*WM)
      const double* sl = (l == L) ? f.rh : f.rl + (size_t)(L - l - 1) * lay + WM * WM;
      if (e < off + 2 * F) {
        const int j = (e - off) % F;
        base = sl + ((e - off) < F ? WM + j : j);
        in_head = l == L;
        found = true;
        break;
      }
      off += 2 * F;
      if (e < off + O * F) {
        const int r = (e - off) / F, q = (e - off) % F;
        if (l < L) {
          base = f.rl + (size_t)(L - l - 1) * lay + r * WM + q;
        } else if (f.rw) {
          base = f.rw + (size_t)(r / BRC) * f.nparts * BNACC + (r % BRC) * (WM + 1) + q;
          gstride = (size_t)f.nch * f.nparts * BNACC;
        } else {
          base = f.rh + 2 * WM + r * (WM + 1) + q;
          in_head = true;
        }
        found = true;
        break;
      }
      off += O * F;
      if (f.hidden_bias && l < L) {
        if (e < off + O) {
          const int j = e - off;
          base = ((l + 1 == L) ? f.rh : f.rl + (size_t)(L - l - 2) * lay + WM * WM) + j;
          in_head = l + 1 == L;
          mult = f.hidden_bias == 2 ? f.bn[((l + 1) * 4 + 0) * WM + j] : 0.0;
          found = true;
          break;
        }
        off += O;
      }
    }
    if (!found) {                                      // bias of the last Linear
      const int r = e - off;
      if (f.rw) {
        base = f.rw + (size_t)(r / BRC) * f.nparts * BNACC + (r % BRC) * (WM + 1) + WM;
        gstride = (size_t)f.nch * f.nparts * BNACC;
      } else {
        base = f.rh + 2 * WM + r * (WM + 1) + WM;
        in_head = true;
      }
    }
    if (in_head) {
      np = f.nparts_h;
      gstride = (size_t)np * BNACC;
    }
  }
  double t0 = 0.0, t1 = 0.0;
  if (base) {
    const int tot = f.G * np;
    int idx = slice;
    for (; idx + 64 < tot; idx += 128) {
      const int i1 = idx + 64;
      t0 += base[(size_t)(idx / np) * gstride + (size_t)(idx % np) * BNACC];
      t1 += base[(size_t)(i1 / np) * gstride + (size_t)(i1 % np) * BNACC];
    }
    if (idx < tot) t0 += base[(size_t)(idx / np) * gstride + (size_t)(idx % np) * BNACC];
  }
  sm[slice][el] = t0 + t1;
  __syncthreads();
  for (int h = 32; h > 0; h >>= 1) {
    if (slice < h) sm[slice][el] += sm[slice + h][el];
    __syncthreads();
  }
  if (slice == 0 && e < f.nelem) f.gparams[e] = sm[0][el] * mult;
}

// ---------------------------------------------------------------------------------------------
bool nis_bwd_cell_supported(const nis_cell_desc& cd, int wm) {
  if (wm == 16) return bwd_wide_supported<16>(cd);
  if (wm == 32) return bwd_wide_supported<32>(cd);
  if (wm == 64) return bwd_wide_supported<64>(cd);
  if (wm != 4 && wm != 8) return false;
  return cd.K <= NIS_MAX_BINS;                        // PWQuad with more than 32 bins: scratch-array variant of the head
}

static size_t bwd_chunks(int R) { return (size_t)(R + BRC - 1) / BRC; }

// the d = 2 class whose last-layer gradients ride in the head kernel
static bool bwd_fused_w(const nis_cell_desc& cd, int wm) { return wm == 4 && cd.kind == NIS_PWQUAD && cd.K == 4 && cd.d - cd.P == 1; }

// workgroups (x extent) of the chain launches: as many as are resident at once (head: 2 per CU at ~240 VGPRs, layer kernels: 3-4), so a
// thread amortises its prologue (record merge) and epilogue (one wave reduction per accumulator) over as many samples as possible
static int bwd_grid(int G, int64_t N, bool head) {
  return std::min(BMAXB, nis_grid_for(head ? 2 : 3, G, N, NIS_TPB));
}

size_t nis_bwd_cell_params(const nis_cell_desc& cd) {
  const int T = cd.d - cd.P;
  const int R = T * (cd.kind == NIS_PWQUAD ? 2 * cd.K + 1 : cd.kind == NIS_AFFINE ? 2 : cd.K);
  size_t n = 0;
  for (int l = 0; l <= cd.n_hidden; ++l) {
    const int F = l == 0 ? cd.P : cd.width[l - 1];
    const int O = l < cd.n_hidden ? cd.width[l] : R;
    n += 2 * (size_t)F + (size_t)O * F;
    if (cd.kind == NIS_AFFINE && l < cd.n_hidden) n += O;
  }
  return n + R;
}

size_t nis_bwd_cell_workspace_doubles(const nis_cell_desc& cd, int wm, int64_t N, int G) {
  if (wm == 16) return bwd_wide_workspace_doubles<16>(cd, N, G);
  if (wm == 32) return bwd_wide_workspace_doubles<32>(cd, N, G);
  if (wm == 64) return bwd_wide_workspace_doubles<64>(cd, N, G);
  const int T = cd.d - cd.P;
  const int R = T * (cd.kind == NIS_PWQUAD ? 2 * cd.K + 1 : cd.kind == NIS_AFFINE ? 2 : cd.K);
  const bool fused = bwd_fused_w(cd, wm);
  const size_t nch = fused ? 0 : bwd_chunks(R);
  const size_t planes = fused ? (size_t)wm : (size_t)(2 * wm + R);
  const size_t rech = (size_t)G * bwd_grid(G, N, true) * BNACC, recl = (size_t)G * bwd_grid(G, N, false) * BNACC;
  return planes * N * G + rech + (size_t)cd.n_hidden * recl + nch * recl;
}

template <int WM>
static int bwd_run(const CellK& c, BwdK k, const nis_cell_desc& cd, double* ws, double* gparams, hipStream_t s) {
  const long N = k.N;
  const int G = k.G, L = c.L;
  const bool fused = bwd_fused_w(cd, WM);
  const size_t nch = fused ? 0 : bwd_chunks(c.R);
  const size_t planes = fused ? (size_t)WM : (size_t)(2 * WM + c.R);
  k.ldw = N;
  k.wsp_gs = (long)(planes * N);
  k.dyp = ws;
  k.alp = ws + (size_t)WM * N;
  k.dzp = ws + (size_t)2 * WM * N;
  const int gridh = bwd_grid(G, N, true), grid = bwd_grid(G, N, false);
  const size_t rech = (size_t)G * gridh * BNACC, rec = (size_t)G * grid * BNACC;
  double* rh = ws + planes * N * G;
  double* rl = rh + rech;
  double* rw = rl + (size_t)L * rec;
  int wx = WM;
  if (WM == 4 && c.P <= 3) {
    wx = 3;
    for (int l = 0; l < L; ++l)
      if (c.width[l] > 3) wx = WM;
  }
  k.part = rh;
  k.prev = nullptr;
  const dim3 gr(grid, G), grh(gridh, G), tb(NIS_TPB);
  const int km = c.K <= 4 ? 4 : c.K <= 8 ? 8 : c.K <= 16 ? 16 : 32;
#define NIS_HEAD(KIND, KMV)                                                                            \
  do {                                                                                                 \
    if (wx == 3) hipLaunchKernelGGL((k_cell_bwd_head<KIND, WM, KMV, (WM == 4 ? 3 : WM), false>), grh, tb, 0, s, k); \
    else hipLaunchKernelGGL((k_cell_bwd_head<KIND, WM, KMV, WM, false>), grh, tb, 0, s, k);             \
  } while (0)
  if (fused) {
    if (wx == 3) hipLaunchKernelGGL((k_cell_bwd_head<NIS_PWQUAD, WM, 4, (WM == 4 ? 3 : WM), true>), grh, tb, 0, s, k);
    else hipLaunchKernelGGL((k_cell_bwd_head<NIS_PWQUAD, WM, 4, WM, true>), grh, tb, 0, s, k);
  } else if (c.kind == NIS_PWLIN)
    NIS_HEAD(NIS_PWLIN, 4);
  else if (c.kind == NIS_AFFINE)
    NIS_HEAD(NIS_AFFINE, 4);
  else if (km == 4)
    NIS_HEAD(NIS_PWQUAD, 4);
  else if (km == 8)
    NIS_HEAD(NIS_PWQUAD, 8);
  else if (km == 16)
    NIS_HEAD(NIS_PWQUAD, 16);
  else if (c.K <= 32)
    NIS_HEAD(NIS_PWQUAD, 32);
  else
    NIS_HEAD(NIS_PWQUAD, NIS_KM_BIG);
#undef NIS_HEAD
  if (!fused) {                          // last-layer weight / bias gradients: their own records, one chunk of BRC rows per blockIdx.y
    BwdK kw = k;
    kw.part = rw;
    hipLaunchKernelGGL((k_bwd_wlast<WM>), dim3(grid, (unsigned)nch, G), tb, 0, s, kw);
  }
  k.prev = rh;
  k.nprev = gridh;
  k.prev_off = 0;
  for (int l = L; l >= 1; --l) {
    const int pass = L - l + 1;                              // 1..L
    k.part = rl + (size_t)(pass - 1) * rec;
    if (wx == 3) hipLaunchKernelGGL((k_cell_bwd_layer<WM, (WM == 4 ? 3 : WM)>), gr, tb, 0, s, k, l);
    else hipLaunchKernelGGL((k_cell_bwd_layer<WM, WM>), gr, tb, 0, s, k, l);
    k.prev = k.part;
    k.nprev = grid;
    k.prev_off = WM * WM;
  }
  hipLaunchKernelGGL((k_cell_bwd_input<WM>), gr, tb, 0, s, k);
  BwdFin f;
  memset(&f, 0, sizeof(f));
  f.c = c;
  f.WM = WM;
  f.G = G;
  f.nparts = grid;
  f.nparts_h = gridh;
  f.nch = (int)nch;
  f.hidden_bias = c.kind == NIS_AFFINE ? (k.invN == 0.0 ? 2 : 1) : 0;
  f.nelem = (int)nis_bwd_cell_params(cd);
  f.rh = rh;
  f.rl = rl;
  f.rw = fused ? nullptr : rw;
  f.bn = k.bn;
  f.gparams = gparams;
  hipLaunchKernelGGL(k_bwd_finish, dim3((f.nelem + 3) / 4), tb, 0, s, f);
  return hipGetLastError() == hipSuccess ? NIS_OK : NIS_E_HIP;
}

// One cell's adjoint for G minibatches: x / xo / g / bnsave point at minibatch 0, x_gs / g_gs / bn_gs are the strides between
// minibatches.  gparams receives the SUM over the minibatches (narrow chain: blockIdx.y = minibatch; matrix-core chain: blockIdx.z).
int nis_bwd_cell_run(const CellK& c, const nis_cell_desc& cd, int wm, const double* x, const double* xo, double* g, const double* bnsave,
                     long ld, long N, int G, long x_gs, long g_gs, long bn_gs, double* ws, double* gparams, const MfmaW& mw, bool eval_bn,
                     hipStream_t s) {
  BwdK k;
  memset(&k, 0, sizeof(k));
  k.c = c;
  k.x = x;
  k.xo = xo;
  k.g = g;
  k.bn = bnsave;
  k.ld = ld;
  k.ldw = ld;
  k.N = N;
  k.invN = eval_bn ? 0.0 : 1.0 / (double)N;
  k.G = G;
  k.x_gs = x_gs;
  k.g_gs = g_gs;
  k.bn_gs = bn_gs;
  if (wm == 4) return bwd_run<4>(c, k, cd, ws, gparams, s);
  if (wm == 8) return bwd_run<8>(c, k, cd, ws, gparams, s);
  if (wm == 16) return bwd_wide_run<16>(c, k, mw, ws, gparams, s);
  if (wm == 32) return bwd_wide_run<32>(c, k, mw, ws, gparams, s);
  if (wm == 64) return bwd_wide_run<64>(c, k, mw, ws, gparams, s);
  return NIS_E_UNSUPPORTED;
}
