// Hand-written adjoint (backward) of a coupling cell with a narrow conditioner (register width WM = 4 or 8, any
// number of transformed columns, PWQuad n_bins <= 32, PWLin any n_bins, AffineCoupling), train-mode BatchNorm.  Wider conditioners
// (WM = 16 / 32 / 64) run the same chain on the f64 matrix cores: flow_bwd_wm.hip.
//
// Reverse of:  BN_0 -> Lin_0 -> [BN_l -> ReLU -> Lin_l]_{l=1..L} -> exp / normalise -> quadratic (linear) CDF, J *= dy/dx
// (reference forward: normalizing_flows/layers/coupling_cells.py:107-142, :159-228, :236-254; the reference obtains these
// gradients from torch.autograd at manager.py:275-278).
//
// Train-mode BatchNorm couples all samples, so the adjoint is again a chain of launches, one per BN layer, in reverse:
//   head    : recompute the cell; per transformed column the bijector adjoint -> dZ rows (stored as planes) -> d a_L;
//             dy_L (grad at BN_L's output) + the two batch sums BN_L's backward needs
//   wlast   : dW_L = dZ^T a_L, db_L = sum dZ  (rows in chunks of 8: a tall-skinny reduction over the samples)
//   layer l : dh_l = scale_l (dy_l - mean(dy_l) - xhat_l mean(dy_l xhat_l)) -> dW_{l-1}, dy_{l-1}, sums of BN_{l-1}
//   input   : BN_0 backward -> gradient of the pass-through columns
// Per-thread partial sums (registers) -> per-workgroup records -> k_bwd_reduce (fixed order: bitwise reproducible).
#include <algorithm>
#include <cstring>

#include "flow_bwd.h"

template <int NACC>
__device__ __forceinline__ void bw_block_reduce(const double (&acc)[NACC], double* part) {
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
    part[(size_t)blockIdx.x * BNACC + threadIdx.x] = t;
  }
}

// records -> one summed record per y-slice: a workgroup owns 4 elements and sums 64 interleaved slices of the records,
// then an LDS tree -- fixed order (bitwise reproducible).  grid: (ceil(nacc / 4), y-slices).
__global__ __launch_bounds__(NIS_TPB) void k_bwd_reduce(const double* __restrict__ part, int nparts, int nacc, double* __restrict__ out,
                                                        long part_ystride) {
  __shared__ double sm[64][4];
  const int el = threadIdx.x & 3, slice = threadIdx.x >> 2;
  const int e = blockIdx.x * 4 + el;
  part += (size_t)blockIdx.y * part_ystride;
  out += (size_t)blockIdx.y * BNACC;
  double t0 = 0.0, t1 = 0.0, t2 = 0.0, t3 = 0.0;
  if (e < nacc) {
    int b = slice;
    for (; b + 192 < nparts; b += 256) {
      t0 += part[(size_t)b * BNACC + e];
      t1 += part[(size_t)(b + 64) * BNACC + e];
      t2 += part[(size_t)(b + 128) * BNACC + e];
      t3 += part[(size_t)(b + 192) * BNACC + e];
    }
    for (; b < nparts; b += 64) t0 += part[(size_t)b * BNACC + e];
  }
  sm[slice][el] = (t0 + t1) + (t2 + t3);
  __syncthreads();
  for (int h = 32; h > 0; h >>= 1) {
    if (slice < h) sm[slice][el] += sm[slice + h][el];
    __syncthreads();
  }
  if (slice == 0 && e < nacc) out[e] = sm[0][el];
}

// forward recompute of the hidden stack with the SAVED statistics; returns for layer `upto` (1..L):
//   h[]   pre-BN activations of BN_upto        a_in[] input of Linear_{upto-1} (post-ReLU of BN_{upto-1}; BN_0 output if upto == 1)
//   o_in[] BN_{upto-1} output before ReLU      xh_in[] normalised input of BN_{upto-1}
template <int WM>
__device__ __forceinline__ void bw_forward(const BwdK& k, long i, int upto, double (&h)[WM], double (&a_in)[WM], double (&o_in)[WM],
                                           double (&xh_in)[WM]) {
  nis_cptr wp = as_const(k.c.wp);
  nis_cptr bn = as_const(k.bn);
  double cur[WM];
#pragma unroll
  for (int p = 0; p < WM; ++p) cur[p] = (p < k.c.P) ? k.x[(size_t)k.c.in_plane[p] * k.ld + i] : 0.0;
  for (int l = 0; l < upto; ++l) {
#pragma unroll
    for (int j = 0; j < WM; ++j) {
      xh_in[j] = (cur[j] - bn[(l * 4 + 2) * WM + j]) * bn[(l * 4 + 3) * WM + j];
      o_in[j] = fma(cur[j], bn[(l * 4 + 0) * WM + j], bn[(l * 4 + 1) * WM + j]);
      a_in[j] = (l == 0) ? o_in[j] : fmax(o_in[j], 0.0);
    }
    dense<WM>(wp + (size_t)l * WM * WM, a_in, cur);
  }
#pragma unroll
  for (int j = 0; j < WM; ++j) h[j] = cur[j];
}

// ---------------------------------------------------------------------------------------------
template <int KIND, int WM, int KM>
__global__ __launch_bounds__(NIS_TPB) void k_cell_bwd_head(BwdK k) {
  __shared__ double etab[NIS_EXP_N];
  nis_exp_init(etab);
  __syncthreads();
  const CellK& c = k.c;
  nis_cptr wl = as_const(c.wlast);
  nis_cptr bias = as_const(c.bias);
  nis_cptr bn = as_const(k.bn);
  const int L = c.L, K = c.K;
  double acc[2 * WM];
#pragma unroll
  for (int a = 0; a < 2 * WM; ++a) acc[a] = 0.0;
  const long stride = (long)gridDim.x * NIS_TPB;
  for (long i = (long)blockIdx.x * NIS_TPB + threadIdx.x; i < k.N; i += stride) {
    double h[WM], a_in[WM], o_in[WM], xh_in[WM];
    bw_forward<WM>(k, i, L, h, a_in, o_in, xh_in);
    double aL[WM], oL[WM], xhL[WM], da[WM];
#pragma unroll
    for (int j = 0; j < WM; ++j) {
      xhL[j] = (h[j] - bn[(L * 4 + 2) * WM + j]) * bn[(L * 4 + 3) * WM + j];
      oL[j] = fma(h[j], bn[(L * 4 + 0) * WM + j], bn[(L * 4 + 1) * WM + j]);
      aL[j] = fmax(oL[j], 0.0);
      k.alp[(size_t)j * k.ld + i] = aL[j];
      da[j] = 0.0;
    }
    const double gJ = k.g[(size_t)c.jplane * k.ld + i];
    const double Jout = k.xo[(size_t)c.jplane * k.ld + i];
    double jfprod = 1.0;
    for (int t = 0; t < c.T; ++t) {
      const int colp = c.in_plane[c.P + t];
      const int row0 = t * c.out;
      const double xraw = k.x[(size_t)colp * k.ld + i];
      const double gy = k.g[(size_t)colp * k.ld + i];
      double gx = 0.0, jf = 1.0;
      if (KIND == NIS_PWQUAD) {
        const bool clamped = xraw > 1.0 - 1e-6;
        const double xb = clamped ? 1.0 - 1e-6 : xraw;
        double V[KM + 1], W[KM];
#pragma unroll
        for (int q = 0; q < KM + 1; ++q) V[q] = (q <= K) ? nis_exp(last_dot<WM>(wl, bias, row0 + q, aL), etab) : 0.0;
#pragma unroll
        for (int q = 0; q < KM; ++q) W[q] = (q < K) ? nis_exp(last_dot<WM>(wl, bias, row0 + K + 1 + q, aL), etab) : 0.0;
        // forward quantities exactly as k_cell_apply forms them
        double wn = 0.0;
#pragma unroll
        for (int q = 0; q < KM; ++q)
          if (q < K) wn += W[q];
        int b;
        double left;
        pwquad_bin<KM, false>(xb, K, W, wn, &b, &left);
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
#define NIS_GV(q, wm1, wq) (((q) >= 1 && (q) - 1 < b ? hgy * (wm1) : 0.0) + ((q) < b ? hgy * (wq) : 0.0) + ((q) == b ? A0 : 0.0) + ((q) == b + 1 ? A1 : 0.0))
#define NIS_GW0(q) (((q) < b ? hgy * (V[q] + V[(q) + 1]) + gleft : 0.0) + ((q) == b ? B0 : 0.0))
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
#undef NIS_GV
#undef NIS_GW0
#pragma unroll
        for (int q = 0; q < KM + 1; ++q)
          if (q <= K) {
            const double dz = V[q];
            k.dzp[(size_t)(row0 + q) * k.ld + i] = dz;
#pragma unroll
            for (int j = 0; j < WM; ++j) da[j] = fma(wl[(size_t)(row0 + q) * WM + j], dz, da[j]);
          }
#pragma unroll
        for (int q = 0; q < KM; ++q)
          if (q < K) {
            const double dz = W[q];
            k.dzp[(size_t)(row0 + K + 1 + q) * k.ld + i] = dz;
#pragma unroll
            for (int j = 0; j < WM; ++j) da[j] = fma(wl[(size_t)(row0 + K + 1 + q) * WM + j], dz, da[j]);
          }
      } else if (KIND == NIS_AFFINE) {
        // AffineCoupling (coupling_cells.py:56-68): rows [scale | shift][T]; u = 20 e^z0 x + relu(z1), y = atan(u) 2/pi,
        // jf = 20 e^z0 / (u^2 + 1)  (the cell-wide factor 2/pi of the Jacobian is applied once, below)
        const int rs = t, rh = c.T + t;
        const double z1 = last_dot<WM>(wl, bias, rh, aL);
        const double s0 = 20.0 * nis_exp(last_dot<WM>(wl, bias, rs, aL), etab);
        const double u = xraw * s0 + (z1 > 0.0 ? z1 : 0.0);
        const double q = 1.0 / (u * u + 1.0);
        jf = s0 * q;
        const double gjf = gJ * Jout / jf;
        const double gu = gy * (q / (M_PI / 2)) - gjf * (2.0 * u * s0 * q * q);
        gx = gu * s0;
        const double dz0 = (gu * xraw + gjf * q) * s0;
        const double dz1 = z1 > 0.0 ? gu : 0.0;
        k.dzp[(size_t)rs * k.ld + i] = dz0;
        k.dzp[(size_t)rh * k.ld + i] = dz1;
#pragma unroll
        for (int j = 0; j < WM; ++j) da[j] = fma(wl[(size_t)rs * WM + j], dz0, fma(wl[(size_t)rh * WM + j], dz1, da[j]));
      } else {
        // PWLin: Q_q = K e_q / tot, y = Q_b alpha + sum_{q<b} e_q / tot, jf = Q_b, alpha = x - b/K  (two streamed passes)
        const double ak = xraw * K;
        const double bf = floor(ak);
        const double alpha = (ak - bf) / K;
        const int b = (int)bf;
        double tot = 0.0, below = 0.0, eb = 0.0;
        for (int q = 0; q < K; ++q) {
          const double e = nis_exp(last_dot<WM>(wl, bias, row0 + q, aL), etab);
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
          const double e = nis_exp(last_dot<WM>(wl, bias, row0 + q, aL), etab);
          const double pq = e / tot;
          const double dz = gQb * ((q == b ? Qb : 0.0) - Qb * pq) + gQs * ((q < b ? pq : 0.0) - Qs * pq);
          k.dzp[(size_t)(row0 + q) * k.ld + i] = dz;
#pragma unroll
          for (int j = 0; j < WM; ++j) da[j] = fma(wl[(size_t)(row0 + q) * WM + j], dz, da[j]);
        }
      }
      k.g[(size_t)colp * k.ld + i] = gx;
      jfprod *= jf;
    }
    if (KIND == NIS_AFFINE) jfprod *= 1.0 / (M_PI / 2);
    k.g[(size_t)c.jplane * k.ld + i] = gJ * jfprod;              // dL/dJ_in = dL/dJ_out * prod_t dy_t/dx_t
    // ---- through ReLU to BN_L's output; sums for BN_L
#pragma unroll
    for (int j = 0; j < WM; ++j) {
      const double dy = (oL[j] > 0.0) ? da[j] : 0.0;
      k.dyp[(size_t)j * k.ld + i] = dy;
      acc[j] += dy;
      acc[WM + j] = fma(dy, xhL[j], acc[WM + j]);
    }
  }
  bw_block_reduce<2 * WM>(acc, k.part);
}

// dW_L[r][j] = sum_s dZ[r][s] a_L[j][s],  db_L[r] = sum_s dZ[r][s]   for the BRC rows of chunk blockIdx.y
template <int WM>
__global__ __launch_bounds__(NIS_TPB) void k_bwd_wlast(BwdK k) {
  const int r0 = blockIdx.y * BRC;
  const int nr = min(BRC, k.c.R - r0);
  double acc[BRC * (WM + 1)];
#pragma unroll
  for (int a = 0; a < BRC * (WM + 1); ++a) acc[a] = 0.0;
  const long stride = (long)gridDim.x * NIS_TPB;
  for (long i = (long)blockIdx.x * NIS_TPB + threadIdx.x; i < k.N; i += stride) {
    double a[WM];
#pragma unroll
    for (int j = 0; j < WM; ++j) a[j] = k.alp[(size_t)j * k.ld + i];
#pragma unroll
    for (int r = 0; r < BRC; ++r)
      if (r < nr) {
        const double dz = k.dzp[(size_t)(r0 + r) * k.ld + i];
#pragma unroll
        for (int j = 0; j < WM; ++j) acc[r * (WM + 1) + j] = fma(dz, a[j], acc[r * (WM + 1) + j]);
        acc[r * (WM + 1) + WM] += dz;
      }
  }
  bw_block_reduce<BRC*(WM + 1)>(acc, k.part + (size_t)blockIdx.y * BMAXB * BNACC);
}

// BN_l backward (l >= 1) -> Linear_{l-1} backward -> dy_{l-1}
template <int WM>
__global__ __launch_bounds__(NIS_TPB) void k_cell_bwd_layer(BwdK k, int l, int sums_off) {
  const CellK& c = k.c;
  nis_cptr wp = as_const(c.wp);
  nis_cptr bn = as_const(k.bn);
  nis_cptr sm = as_const(k.sums);
  const double invN = k.invN;
  double acc[WM * WM + 2 * WM];
#pragma unroll
  for (int a = 0; a < WM * WM + 2 * WM; ++a) acc[a] = 0.0;
  const long stride = (long)gridDim.x * NIS_TPB;
  for (long i = (long)blockIdx.x * NIS_TPB + threadIdx.x; i < k.N; i += stride) {
    double h[WM], a_in[WM], o_in[WM], xh_in[WM];
    bw_forward<WM>(k, i, l, h, a_in, o_in, xh_in);
    double dh[WM];
#pragma unroll
    for (int j = 0; j < WM; ++j) {
      const double xh = (h[j] - bn[(l * 4 + 2) * WM + j]) * bn[(l * 4 + 3) * WM + j];
      const double dy = k.dyp[(size_t)j * k.ld + i];
      const double m1 = sm[sums_off + j] * invN, m2 = sm[sums_off + WM + j] * invN;
      dh[j] = bn[(l * 4 + 0) * WM + j] * (dy - m1 - xh * m2);       // scale_l = gamma_l * rstd_l
    }
    double da[WM];
#pragma unroll
    for (int q = 0; q < WM; ++q) da[q] = 0.0;
#pragma unroll
    for (int j = 0; j < WM; ++j)
#pragma unroll
      for (int q = 0; q < WM; ++q) {
        acc[j * WM + q] = fma(dh[j], a_in[q], acc[j * WM + q]);      // dW_{l-1}[j][q]
        da[q] = fma(wp[(size_t)(l - 1) * WM * WM + j * WM + q], dh[j], da[q]);
      }
#pragma unroll
    for (int q = 0; q < WM; ++q) {
      const double dy = (l - 1 == 0 || o_in[q] > 0.0) ? da[q] : 0.0;   // BN_0 has no ReLU behind it
      k.dyp[(size_t)q * k.ld + i] = dy;
      acc[WM * WM + q] += dy;
      acc[WM * WM + WM + q] = fma(dy, xh_in[q], acc[WM * WM + WM + q]);
    }
  }
  bw_block_reduce<WM * WM + 2 * WM>(acc, k.part);
}

// BN_0 backward: gradient of the pass-through columns
template <int WM>
__global__ __launch_bounds__(NIS_TPB) void k_cell_bwd_input(BwdK k, int sums_off) {
  const CellK& c = k.c;
  nis_cptr bn = as_const(k.bn);
  nis_cptr sm = as_const(k.sums);
  const double invN = k.invN;
  const long stride = (long)gridDim.x * NIS_TPB;
  for (long i = (long)blockIdx.x * NIS_TPB + threadIdx.x; i < k.N; i += stride) {
#pragma unroll
    for (int p = 0; p < WM; ++p)
      if (p < c.P) {
        const size_t at = (size_t)c.in_plane[p] * k.ld + i;
        const double xh = (k.x[at] - bn[2 * WM + p]) * bn[3 * WM + p];
        const double dy = k.dyp[(size_t)p * k.ld + i];
        const double m1 = sm[sums_off + p] * invN, m2 = sm[sums_off + WM + p] * invN;
        k.g[at] += bn[p] * (dy - m1 - xh * m2);
      }
  }
}

// Summed records -> gradient blob in the reference's parameter order (include/nis_hip.h).
//   sums: [pass 0 = head (BN_L sums) | pass 1 = layer L | ... | pass L = layer 1], each BNACC doubles;
//   wsum: one BNACC record per chunk of BRC last-layer rows, [r][WM weights | bias].
// hidden_bias (NIS_AFFINE): 1 = train-mode BatchNorm, 2 = eval mode (`bn`: the saved scale | shift | mean | rstd records).
__global__ void k_bwd_scatter(CellK c, int WM, const double* __restrict__ sums, const double* __restrict__ wsum,
                              double* __restrict__ gparams, int hidden_bias, const double* __restrict__ bn) {
  const int L = c.L;
  size_t off = 0;
  const int tid = threadIdx.x, nth = blockDim.x;
  for (int l = 0; l <= L; ++l) {
    const int F = cellk_feat(c, l);
    const int O = l < L ? c.width[l] : c.R;
    // d gamma_l = sum dy_l xhat_l, d beta_l = sum dy_l: emitted by the pass that produced dy_l
    //   dy_L by the head (pass 0, offset 0); dy_l (l < L) by layer pass l+1 = pass index L-l, offset WM*WM
    const double* s = (l == L) ? sums : sums + (size_t)(L - l) * BNACC + WM * WM;
    for (int j = tid; j < F; j += nth) {
      gparams[off + j] = s[WM + j];
      gparams[off + F + j] = s[j];
    }
    off += 2 * (size_t)F;
    if (l < L) {                       // dW_l from layer pass l+1 (pass index L-l)
      const double* w = sums + (size_t)(L - l) * BNACC;
      for (int e = tid; e < O * F; e += nth) gparams[off + e] = w[(e / F) * WM + (e % F)];
      if (hidden_bias) {                 // NIS_AFFINE: a bias in front of a train-mode BatchNorm has gradient sum(dh) == 0;
        // in eval mode dh_{l+1} = scale_{l+1} dy_{l+1}, so the bias of Linear_l gets scale_{l+1} * sum dy_{l+1}
        const double* s1 = (l + 1 == L) ? sums : sums + (size_t)(L - l - 1) * BNACC + WM * WM;
        for (int e = tid; e < O; e += nth)
          gparams[off + (size_t)O * F + e] = hidden_bias == 2 ? bn[((l + 1) * 4 + 0) * WM + e] * s1[e] : 0.0;
      }
    } else {
      for (int e = tid; e < O * F; e += nth) {
        const int r = e / F, q = e % F;
        gparams[off + e] = wsum[(size_t)(r / BRC) * BNACC + (r % BRC) * (WM + 1) + q];
      }
    }
    off += (size_t)O * F;
    if (hidden_bias && l < L) off += O;
  }
  for (int r = tid; r < c.R; r += nth) gparams[off + r] = wsum[(size_t)(r / BRC) * BNACC + (r % BRC) * (WM + 1) + WM];
}

// ---------------------------------------------------------------------------------------------
bool nis_bwd_cell_supported(const nis_cell_desc& cd, int wm) {
  if (wm == 16) return bwd_wide_supported<16>(cd);
  if (wm == 32) return bwd_wide_supported<32>(cd);
  if (wm == 64) return bwd_wide_supported<64>(cd);
  if (wm != 4 && wm != 8) return false;
  if (cd.kind == NIS_PWQUAD) return cd.K <= 32;
  return cd.K <= NIS_MAX_BINS;
}

static size_t bwd_chunks(int R) { return (size_t)(R + BRC - 1) / BRC; }

size_t nis_bwd_cell_workspace_doubles(const nis_cell_desc& cd, int wm, int64_t N) {
  if (wm == 16) return bwd_wide_workspace_doubles<16>(cd, N);
  if (wm == 32) return bwd_wide_workspace_doubles<32>(cd, N);
  if (wm == 64) return bwd_wide_workspace_doubles<64>(cd, N);
  const int T = cd.d - cd.P;
  const int R = T * (cd.kind == NIS_PWQUAD ? 2 * cd.K + 1 : cd.kind == NIS_AFFINE ? 2 : cd.K);
  const size_t nch = bwd_chunks(R);
  return (size_t)(2 * wm + R) * N + std::max<size_t>(1, nch) * BMAXB * BNACC + (size_t)(cd.n_hidden + 2) * BNACC + nch * BNACC;
}

template <int WM>
static int bwd_run(const CellK& c, BwdK k, double* ws, double* gparams, hipStream_t s) {
  const long N = k.N;
  k.dyp = ws;
  k.alp = ws + (size_t)WM * N;
  k.dzp = ws + (size_t)2 * WM * N;
  const size_t nch = bwd_chunks(c.R);
  double* part = ws + (size_t)(2 * WM + c.R) * N;
  double* sums = part + std::max<size_t>(1, nch) * BMAXB * BNACC;
  double* wsum = sums + (size_t)(c.L + 2) * BNACC;
  k.part = part;
  const int grid = (int)std::max<long>(1, std::min<long>(BMAXB, (N + NIS_TPB - 1) / NIS_TPB));
  const int L = c.L;
  const int km = c.K <= 4 ? 4 : c.K <= 8 ? 8 : c.K <= 16 ? 16 : 32;
  if (c.kind == NIS_PWLIN)
    hipLaunchKernelGGL((k_cell_bwd_head<NIS_PWLIN, WM, 4>), dim3(grid), dim3(NIS_TPB), 0, s, k);
  else if (c.kind == NIS_AFFINE)
    hipLaunchKernelGGL((k_cell_bwd_head<NIS_AFFINE, WM, 4>), dim3(grid), dim3(NIS_TPB), 0, s, k);
  else if (km == 4)
    hipLaunchKernelGGL((k_cell_bwd_head<NIS_PWQUAD, WM, 4>), dim3(grid), dim3(NIS_TPB), 0, s, k);
  else if (km == 8)
    hipLaunchKernelGGL((k_cell_bwd_head<NIS_PWQUAD, WM, 8>), dim3(grid), dim3(NIS_TPB), 0, s, k);
  else if (km == 16)
    hipLaunchKernelGGL((k_cell_bwd_head<NIS_PWQUAD, WM, 16>), dim3(grid), dim3(NIS_TPB), 0, s, k);
  else
    hipLaunchKernelGGL((k_cell_bwd_head<NIS_PWQUAD, WM, 32>), dim3(grid), dim3(NIS_TPB), 0, s, k);
  hipLaunchKernelGGL(k_bwd_reduce, dim3((2 * WM + 3) / 4, 1), dim3(NIS_TPB), 0, s, (const double*)part, grid, 2 * WM, sums, 0L);
  // last-layer weight / bias gradients (its own records, reduced per chunk)
  {
    const int gw = grid;                 // as many workgroups as the chain kernels: the reduction is issue- and latency-bound
    hipLaunchKernelGGL((k_bwd_wlast<WM>), dim3(gw, (unsigned)nch), dim3(NIS_TPB), 0, s, k);
    hipLaunchKernelGGL(k_bwd_reduce, dim3((BRC * (WM + 1) + 3) / 4, (unsigned)nch), dim3(NIS_TPB), 0, s, (const double*)part, gw, BRC * (WM + 1), wsum,
                       (long)BMAXB * BNACC);
  }
  for (int l = L; l >= 1; --l) {
    const int pass = L - l + 1;                              // 1..L
    k.sums = sums + (size_t)(pass - 1) * BNACC;
    const int soff = (pass == 1) ? 0 : WM * WM;
    hipLaunchKernelGGL((k_cell_bwd_layer<WM>), dim3(grid), dim3(NIS_TPB), 0, s, k, l, soff);
    hipLaunchKernelGGL(k_bwd_reduce, dim3((WM * WM + 2 * WM + 3) / 4, 1), dim3(NIS_TPB), 0, s, (const double*)part, grid, WM * WM + 2 * WM,
                       sums + (size_t)pass * BNACC, 0L);
  }
  k.sums = sums + (size_t)L * BNACC;
  hipLaunchKernelGGL((k_cell_bwd_input<WM>), dim3(grid), dim3(NIS_TPB), 0, s, k, WM * WM);
  hipLaunchKernelGGL(k_bwd_scatter, dim3(1), dim3(NIS_TPB), 0, s, c, WM, (const double*)sums, (const double*)wsum, gparams,
                     c.kind == NIS_AFFINE ? (k.invN == 0.0 ? 2 : 1) : 0, k.bn);
  return hipGetLastError() == hipSuccess ? NIS_OK : NIS_E_HIP;
}

int nis_bwd_cell_run(const CellK& c, int wm, const double* x, const double* xo, double* g, const double* bnsave, long ld, long N,
                     double* ws, double* gparams, const MfmaW& mw, bool eval_bn, hipStream_t s) {
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
  if (wm == 4) return bwd_run<4>(c, k, ws, gparams, s);
  if (wm == 8) return bwd_run<8>(c, k, ws, gparams, s);
  if (wm == 16) return bwd_wide_run<16>(c, k, mw, ws, gparams, s);
  if (wm == 32) return bwd_wide_run<32>(c, k, mw, ws, gparams, s);
  if (wm == 64) return bwd_wide_run<64>(c, k, mw, ws, gparams, s);
  return NIS_E_UNSUPPORTED;
}
