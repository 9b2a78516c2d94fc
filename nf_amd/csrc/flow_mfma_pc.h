// Matrix-core apply kernel with WAVE-SPECIALISED roles: on every SIMD one PRODUCER wave (conditioner GEMMs on
// v_mfma_f64_16x16x4_f64 + exp of the last layer's rows) is paired with one CONSUMER wave (bijector, stores, integrand rider).
//
// k_cell_apply_mfma (flow_mfma.h) runs two identical waves per SIMD, each alternating between a matrix phase and a bijector
// phase that is a chain of LDS round trips with little arithmetic between them; whenever both waves of a SIMD sit in that
// phase the fp64 datapath idles (PMC: MFMA + VALU busy ~75-80 % of the launch).  Here a 512-thread workgroup (one per CU)
// puts waves w and w + 4 on one SIMD: wave w < 4 only produces -- hidden stack, then per item (sample tile, column batch)
// the last-layer tiles whose exp() values go to one of TWO per-pair LDS images -- and wave w + 4 only consumes the image the
// producer finished one item earlier.  One workgroup barrier per item hands the images over.  The two programs are
// complementary by construction (matrix/exp work beside latency-bound selection work), so the datapath stays fed; the
// arithmetic is the SAME code as k_cell_apply_mfma's (mf_hidden, nis_exp, pwquad_column_lds, streamed PWLin), results are
// bit-identical.
#pragma once
#include "flow_mfma.h"

#define NIS_MFPC_TPB 512
#define NIS_MFPC_PAIRS 4               // producer / consumer pairs per workgroup (one per SIMD)

template <int KIND, int WM, int KM, bool KEXACT, bool FUSE = false>
__global__ __launch_bounds__(NIS_MFPC_TPB, 2) void k_cell_apply_mfpc(CellK c, double* __restrict__ planes, long ld, long N, int bn_train,
                                                                      int* __restrict__ bins_out, const double* __restrict__ wm_all,
                                                                      const double* __restrict__ wm_last, const double* __restrict__ bias_pad,
                                                                      FuseF fz) {
  constexpr int ST = MfCfg<WM>::ST, JT = MfCfg<WM>::JT, KQ = MfCfg<WM>::KQ, NP = NIS_MFPC_PAIRS;
  const int NCB = c.mf_ncb;                          // columns per item: lane groups gq >= NCB idle in the consumer
  extern __shared__ double mf_dyn[];                 // [2 buffers][NP pairs][row tiles of a column batch][16 rows][16 samples]
  __shared__ double lds_coef[(NIS_MAX_HIDDEN + 1) * 2 * WM];
  double* const red = mf_dyn;                        // the prologue's reduction scratch (3 * TPB doubles) lives in the images
  __shared__ int lds_plane[NIS_MAX_DIM];
  __shared__ double etab[NIS_EXP_N];
  __shared__ double fmom[NP][4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, gq = lane >> 4, s = lane & 15;
  const bool producer = wave < NP;                   // wave-uniform role; waves w and w + NP share a SIMD
  const int pw = wave & (NP - 1);
  const int g = c.g0 + blockIdx.y;
  planes += (size_t)g * c.plane_gs;
  if (tid < NIS_MAX_DIM) lds_plane[tid] = c.in_plane[tid];
  nis_exp_init(etab);
  load_coefs<WM, NIS_MFPC_TPB>(c, g, c.L + 1, (bn_train && c.pmerge) ? cellk_fresh(c, c.L + 1) : -1, blockIdx.x == 0, lds_coef, red);
  __syncthreads();                                   // the images are reused by the waves from here on
  const int out = c.out;
  const bool wplanes = FUSE ? fz.write_planes != 0 : true;
  const double f_i2s = FUSE ? 1.0 / (2 * fz.p[1] * fz.p[1]) : 0.0;
  const double f_norm = (FUSE && fz.kind == NIS_F_NORMAL) ? 1.0 / (fz.p[1] * sqrt(pow(2 * M_PI, (double)c.d))) : 1.0;
  const int cb_tiles = mf_cb_tiles(c.T, out, c.R, NCB);
  const size_t buf_stride = (size_t)NP * cb_tiles * 256;
  double* const ring0 = mf_dyn + (size_t)pw * cb_tiles * 256;
  const long wstride = (long)gridDim.x * NP * 16 * ST;
  const int ncb = (c.T + NCB - 1) / NCB;

  // ---- producer state
  double an[KQ];
  nis_d4 bnx = {0.0, 0.0, 0.0, 0.0};
  auto load_tile = [&](int jt) {
#pragma unroll
    for (int q = 0; q < KQ; ++q) an[q] = wm_last[((size_t)jt * KQ + q) * 64 + lane];
#pragma unroll
    for (int r = 0; r < 4; ++r) bnx[r] = bias_pad[16 * jt + gq + 4 * r];
  };
  constexpr int PQ = 2;
  const bool pin = NIS_MF_PREFETCH && c.P <= 4 * PQ;
  double xin[PQ][ST];
  double H[KQ][ST];
  double Hs[KQ];
  const long bb0 = (long)blockIdx.x * NP * 16 * ST;
  if (producer) {
    load_tile(0);
    if (pin && bb0 + pw * 16 * ST < N) mf_issue_input<WM, PQ>(c, planes, ld, N, bb0 + pw * 16 * ST, lds_plane, xin);
  }
  // ---- consumer state (one sample tile at a time; `c_*`: the item produced one step ago, workgroup-uniform)
  double jprod = 1.0, facc = 0.0, fc0 = 0.0, fc1 = 0.0, jinq = 1.0;
  double fS1 = 0.0, fS2 = 0.0, fcnt = 0.0, fmx = -__builtin_huge_val(), fshift = 0.0;
  bool ffirst = true;
  long c_base = 0;
  int c_st = 0, c_cb = 0;
  bool have_prev = false;                            // (consumer loop: the item handed over at the last barrier)

  auto consume = [&](long base, int st, int cb, int buf) {
    const long i = base + 16 * st + s;
    const int t = cb * NCB + gq;
    const bool live = i < N;
    if (cb == 0) {                                   // a new sample tile: reset the per-sample accumulators
      jprod = 1.0;
      facc = fc0 = fc1 = 0.0;
      jinq = (!c.first && gq == 0 && live) ? planes[(size_t)c.jplane * ld + i] : 1.0;
      if (c.out_off != 0) {                          // history mode: the next plane set gets the pass-through block too
        for (int p = gq; p < c.P; p += 4)
          if (live) planes[(size_t)lds_plane[p] * ld + c.out_off + i] = planes[(size_t)lds_plane[p] * ld + i];
      }
      if (FUSE) {
        for (int p = gq; p < c.P; p += 4) {          // pass-through columns, spread over the four lanes of the sample
          const double xp = live ? planes[(size_t)lds_plane[p] * ld + i] : 0.0;
          if (fz.kind == NIS_F_NORMAL) facc = fma((xp - fz.p[0]) * (xp - fz.p[0]), f_i2s, facc);
          if (fz.kind == NIS_F_CAMEL) {
            fc0 = (p == fz.loc0) ? xp : fc0;
            fc1 = (p == fz.loc1) ? xp : fc1;
          }
        }
      }
    }
    const int r0 = cb * NCB * out;
    if (gq < NCB && t < c.T && live) {
      double* col = planes + (size_t)lds_plane[c.P + t] * ld;
      const double xb = col[i];
      const double* zr = ring0 + (size_t)buf * buf_stride + (size_t)((r0 & 15) + gq * out) * 16 + s;
      double y, jf;
      int bin;
      if (KIND == NIS_PWQUAD) {
        y = pwquad_column_lds<KM, KEXACT, 16>(xb, c.K, zr, &jf, &bin);
      } else if (KIND == NIS_AFFINE) {
        const double s0 = 20.0 * zr[0], z1 = zr[16];                                         // rows interleaved at pack time (coupling_cells.py:56-68)
        const double u = xb * s0 + (z1 > 0.0 ? z1 : 0.0);
        y = atan(u) / (M_PI / 2);
        jf = s0 * (1.0 / (u * u + 1.0));
        bin = 0;
      } else {
        const double ak = xb * c.K;
        const double bf = floor(ak);
        const double alpha = (ak - bf) / c.K;
        bin = (int)bf;
        double tot = 0.0, below = 0.0;
        int k0 = 0;
        for (; k0 + 8 <= c.K; k0 += 8) {
          double e[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) e[u] = zr[(k0 + u) * 16];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            below = (k0 + u == bin) ? tot : below;
            tot += e[u];                                                                    // :117
          }
        }
        for (; k0 < c.K; ++k0) {
          const double e = zr[k0 * 16];
          below = (k0 == bin) ? tot : below;
          tot += e;
        }
        double qb = zr[(bin < 0 ? 0 : bin >= c.K ? c.K - 1 : bin) * 16];
        if (bin < 0 || bin >= c.K) {
          if (c.status) atomicOr(c.status, 1);
          qb = __builtin_nan("");
        }
        const double itot = nis_rcp(tot);
        const double qn = qb * (c.K * itot);                                                // :121
        y = qn * alpha + below * itot;                                                      // :123,136-139
        jf = qn;                                                                            // :141
      }
      if (wplanes) col[c.out_off + i] = y;
      jprod *= jf;
      if (FUSE) {
        if (fz.kind == NIS_F_NORMAL) facc = fma((y - fz.p[0]) * (y - fz.p[0]), f_i2s, facc);
        if (fz.kind == NIS_F_CAMEL) {
          fc0 = (c.P + t == fz.loc0) ? y : fc0;
          fc1 = (c.P + t == fz.loc1) ? y : fc1;
        }
      }
      if (KIND != NIS_AFFINE && bins_out && g == 0) bins_out[(size_t)(c.bins_row + t) * ld + i] = bin;
    }
    if (cb == ncb - 1) {                             // the sample tile is complete: Jacobian product, integrand rider
      double jp = jprod;
      jp *= __shfl_xor(jp, 16, 64);
      jp *= __shfl_xor(jp, 32, 64);
      if (KIND == NIS_AFFINE) jp *= 1.0 / (M_PI / 2);          // the reference applies 2/pi ONCE per cell (:68)
      double jout = 1.0;
      if (gq == 0 && live) {
        jout = (c.first ? 1.0 : jinq) * jp;
        if (wplanes) planes[(size_t)c.jplane * ld + c.out_off + i] = jout;
      }
      if (FUSE) {
        double fa = facc, f0 = fc0, f1 = fc1;
        fa += __shfl_xor(fa, 16, 64);
        fa += __shfl_xor(fa, 32, 64);
        f0 += __shfl_xor(f0, 16, 64);
        f0 += __shfl_xor(f0, 32, 64);
        f1 += __shfl_xor(f1, 16, 64);
        f1 += __shfl_xor(f1, 32, 64);
        double fv = 1.0;
        if (fz.kind == NIS_F_NORMAL) {                                                    // manager.py:39-40
          fv = nis_exp(-fa, etab) * f_norm;
        } else if (fz.kind == NIS_F_CAMEL) {                                              // README.md:35-36
          const double is2 = fz.p[3];
          const double e1 = -((f0 - fz.p[0]) * (f0 - fz.p[0]) + (f1 - fz.p[0]) * (f1 - fz.p[0])) * is2;
          const double e2 = -((f0 - fz.p[1]) * (f0 - fz.p[1]) + (f1 - fz.p[1]) * (f1 - fz.p[1])) * is2;
          fv = nis_exp(e1, etab) + nis_exp(e2, etab);
        }
        const double v = fv * jout;
        const bool mine = gq == 0 && live;
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
  };

  // Two programs, one barrier schedule: both roles walk the same (trip, sample tile, column batch) item sequence -- the trip count is
  // workgroup-uniform -- and meet at ONE barrier per item; the consumer lags by one item.  The roles are separate loops (the branch is
  // wave-uniform) so that neither program's registers are live in the other's.
  if (producer) {
    int it = 0;
    for (long bb = bb0; bb < N; bb += wstride) {
      const long base = bb + (long)pw * 16 * ST;
      {
        nis_d4 acc[JT][ST];
        if (pin) {
          mf_finish_input<WM, PQ>(c, N, base, lds_coef, xin, H);
          const long nb = bb + wstride < N ? base + wstride : base;
          mf_issue_input<WM, PQ>(c, planes, ld, N, nb < N ? nb : (base < N ? base : 0), lds_plane, xin);
        } else {
          mf_load_input<WM>(c, planes, ld, N, base < N ? base : 0, lds_plane, lds_coef, H);
          if (base >= N) {
#pragma unroll
            for (int q = 0; q < KQ; ++q)
#pragma unroll
              for (int u = 0; u < ST; ++u) H[q][u] = 0.0;
          }
        }
        mf_hidden<WM>(c, wm_all, lds_coef, c.L, H, acc);
        mf_bn_relu<WM>(lds_coef + (size_t)c.L * 2 * WM, acc, H);
      }
#pragma unroll 1
      for (int st = 0; st < ST; ++st) {
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
          const int jt_first_next = (cb + 1 < ncb) ? ((cb + 1) * NCB * out) / 16 : 0;      // cyclic: the next sample tile starts at row tile 0
          const int jt0 = r0 / 16;
          const bool raw_row = KIND == NIS_AFFINE && (((jt0 * 16 + gq - r0) & 1) != 0);
          double* rw = ring0 + (size_t)(it & 1) * buf_stride + lane;
          for (int jt = jt0; jt * 16 < r1; ++jt) {
            nis_d4 z = bnx;
#pragma unroll
            for (int q = 0; q < KQ; ++q) z = __builtin_amdgcn_mfma_f64_16x16x4f64(an[q], Hs[q], z, 0, 0, 0);
            load_tile((jt + 1) * 16 < r1 ? jt + 1 : jt_first_next);
#pragma unroll
            for (int r = 0; r < 4; ++r) rw[64 * r] = raw_row ? z[r] : nis_exp(z[r], etab);
            rw += 256;
          }
          __syncthreads();                           // item `it` is complete in its image (and item `it - 1` has been consumed)
          ++it;
        }
      }
    }
  } else {
    int it = 0;
    for (long bb = bb0; bb < N; bb += wstride) {
      const long base = bb + (long)pw * 16 * ST;
#pragma unroll 1
      for (int st = 0; st < ST; ++st) {
#pragma unroll 1
        for (int cb = 0; cb < ncb; ++cb) {
          if (have_prev) consume(c_base, c_st, c_cb, (it - 1) & 1);
          __syncthreads();
          c_base = base;
          c_st = st;
          c_cb = cb;
          have_prev = true;
          ++it;
        }
      }
    }
    if (have_prev) consume(c_base, c_st, c_cb, (it - 1) & 1);
  }
  if (FUSE) {                                        // moment records: the consumer waves hold them
    const double nw = wave_sum(fcnt), s1 = wave_sum(fS1), s2 = wave_sum(fS2), mx = wave_max(fmx);
    if (!producer && lane == 0) {
      fmom[pw][0] = nw;
      fmom[pw][1] = nw > 0.0 ? fshift + s1 / nw : 0.0;
      fmom[pw][2] = nw > 0.0 ? fmax(s2 - s1 * s1 / nw, 0.0) : 0.0;
      fmom[pw][3] = mx;
    }
    __syncthreads();
    if (tid == 0) {
      Mom tm = {0.0, 0.0, 0.0};
      double m = -__builtin_huge_val();
      for (int w = 0; w < NP; ++w) {
        Mom q = {fmom[w][0], fmom[w][1], fmom[w][2]};
        tm = mom_merge(tm, q);
        m = fmax(m, fmom[w][3]);
      }
      double* part = fz.partials + ((size_t)g * fz.maxb + blockIdx.x) * 4;
      part[0] = tm.n;
      part[1] = tm.mean;
      part[2] = tm.m2;
      part[3] = m;
    }
  }
}
