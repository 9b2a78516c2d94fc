// Adjoint (backward) of a coupling cell with a WIDE conditioner (register width NIS_WM = 16 / 32 / 64), train-mode
// BatchNorm, on the f64 matrix cores (v_mfma_f64_16x16x4_f64).  One translation unit per width (-DNIS_WM=...).
// Reference: the gradients torch.autograd produces at normalizing_flows/manager.py:275-278 for the forward of
// normalizing_flows/layers/coupling_cells.py:107-142, :159-228, :236-254.
//
// Same launch chain as the narrow adjoint (flow_bwd.hip): head -> [layer L .. 1] -> input, every step followed by a
// fixed-order reduction of its per-workgroup records.  Orientation as in flow_mfma.h: tiles are C[feature][sample]
// with 16 samples per tile, so
//   * the forward is recomputed register-to-register (accumulators of one layer are the B operands of the next);
//   * d a = W^T d h needs no transposition either: the k index of the MFMA runs over the rows of W, lane l supplies
//     W[4 ks + (l>>4)][16 qt + (l&15)] straight from the row-major weight image and d h[4 ks + (l>>4)][sample l&15]
//     from its accumulator registers;
//   * in the head the bijector adjoint leaves dZ of (sample l&15, column slot l>>4) in that lane's registers: with the
//     k index running over the 4 column slots of a batch, row q of each slot is one MFMA k-step of d a_L = W_L^T dZ;
//   * only the weight gradients dW = d h a^T reduce over SAMPLES, which this layout keeps in the n index.  The hidden layers'
//     (round 4) are accumulated INSIDE k_bwd_layer_w: the wave's d h and a tiles cross a padded per-wave LDS image and come
//     back with samples on the k index (16 MFMAs per tile at width 32), per-workgroup records go to the reduction the chain
//     runs anyway -- no d h / a planes, no GEMM launch.  The last layer's dW_L = dZ a_L^T is taken from the dZ / a_L planes the
//     head stores by a tall-skinny MFMA GEMM (k_bwd_xty_w) whose k index runs over samples: accumulating it inside the head (dZ in
//     the LDS ring, dW_L in the registers of dedicated GEMM waves) was built and measured slower, profiles/r04_head2_ab.txt.
#include <algorithm>
#include <cstring>

#include "flow_bwd.h"
#include "flow_mfma.h"

#ifndef NIS_WM
#error "compile with -DNIS_WM=<16|32|64>"
#endif

namespace {
constexpr int WM = NIS_WM, JT = WM / 16, KQ = WM / 4, NW = 4, TPB = 256, NCB = NIS_MF_NCB;
constexpr int ST = 1;                           // sample tiles (of 16) per wave pass: keeps the chain kernels at <= 256 registers
constexpr int WPS = WM >= 64 ? 1 : 2;           // waves per SIMD the chain kernels are compiled for
constexpr int RPW = WM >= 64 ? 2 : 4;           // row tiles (of 16) per workgroup of the weight-gradient GEMM
constexpr int GREC = RPW * 16 * (WM + 1);       // its record: RPW*16 rows x (WM products | row sum)
constexpr int SREC = 2 * WM;                    // BatchNorm record: sum dy | sum dy*xhat
constexpr int BNLDS = (NIS_MAX_HIDDEN + 1) * 4 * WM;
constexpr int TS = 17;                          // row stride (doubles) of the staging tiles [feature][16 samples]: transposed reads without bank conflicts
constexpr int STAGE = NW * 2 * WM * TS;         // k_bwd_layer_w: per wave the d h tile and the a tile of its current 16 samples

struct BwdW {
  const double* wm_all;                         // MFMA fragments of the hidden layers / last layer (k_pack_mfma)
  const double* wm_last;
  const double* bias_pad;
  double* partg;                                // k_bwd_layer_w: records of dW_{l-1} (row groups of GREC per workgroup, the GEMM's record layout)
  long partg_gs;                                //   doubles between the minibatches' record areas
  int nb;                                       //   workgroups per minibatch (record index = group * nb + blockIdx.x)
  // small batches (one stream): a layer kernel merges the BatchNorm-sum records of the launch in front of it in its own prologue -- no
  // reduction launch in between -- and its workgroup 0 leaves the reduced sums for the scatter
  const double* prev;                           // nprev records of SREC doubles (nullptr: BwdK::sums holds the reduced sums)
  double* sums_w;
  int nprev;
};

// Workspace buffers (dZ, dy, d h, a) are tile-major: [16-sample tile][feature][16 samples], so one wave's tile is a
// contiguous block (feature stride 128 B: immediate offsets) and the GEMM reads whole 2 KB runs of 16 rows.
__device__ __forceinline__ size_t tmx(int F, int j, long i) { return ((size_t)(i >> 4) * F + j) * 16 + (i & 15); }

__device__ __forceinline__ nis_d4 mfma(double a, double b, nis_d4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }

__device__ __forceinline__ void bw_dense(const double* __restrict__ wm, int nq, const double (&H)[KQ][ST], nis_d4 (&acc)[JT][ST]) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int jt = 0; jt < JT; ++jt) {
#pragma unroll
    for (int st = 0; st < ST; ++st) acc[jt][st] = nis_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int q = 0; q < KQ; ++q)
      if (q < nq) {
        const double a = wm[((size_t)jt * KQ + q) * 64 + lane];
#pragma unroll
        for (int st = 0; st < ST; ++st) acc[jt][st] = mfma(a, H[q][st], acc[jt][st]);
      }
  }
}

// Forward recompute with the SAVED statistics up to BN layer `upto` (1..L).  Feature of H[q] / XH[q]: 4 q + (lane>>4).
// (Saving h_1 .. h_{L-1} from the head so that the layer kernels start one dense layer below their BatchNorm instead of at x -- built,
// parity-green, removed: at C5 / 2e5 the head's 16 extra stores per tile and 34 more spilled registers cost 40 us per cell where the layer
// launches gained 16, their loops being bound by the latency of a tile's loads and MFMA chain, not by the count of dense layers; in the
// 5 x 2 000 regime the step moved by 1-2 %, inside its run-to-run spread.)
//   acc: pre-BN activations of BN_upto | H: input of Linear_{upto-1} | XH: normalised input of BN_{upto-1}
constexpr int XPRE = 2;          // pass-through features 4 q + gq, q < XPRE (the first 8), of the NEXT tile a chain kernel may request ahead (bww_prefetch_x)
__device__ __forceinline__ void bww_prefetch_x(const CellK& c, const double* __restrict__ x, long ld, long N, long base, const int* lds_plane,
                                               double (&xp)[XPRE]) {
  const int lane = threadIdx.x & 63, gq = lane >> 4, s = lane & 15;
  const long i = base + s;
#pragma unroll
  for (int q = 0; q < XPRE; ++q) {
    const int kf = 4 * q + gq;
    xp[q] = (kf < c.P && i < N) ? x[(size_t)lds_plane[kf] * ld + i] : 0.0;
  }
}
__device__ __forceinline__ void bww_forward(const CellK& c, const double* __restrict__ x, long ld, long N, long base, const int* lds_plane,
                                            const double* lds_bn, const double* __restrict__ wm_all, int upto, double (&H)[KQ][ST],
                                            double (&XH)[KQ][ST], nis_d4 (&acc)[JT][ST], const double* xpre = nullptr) {
  const int lane = threadIdx.x & 63, gq = lane >> 4, s = lane & 15;
#pragma unroll
  for (int q = 0; q < KQ; ++q) {
    const int kf = 4 * q + gq;
    const bool kv = kf < c.P;
    const size_t pl = kv ? (size_t)lds_plane[kf] * ld : 0;
    const double sc = lds_bn[kf], sh = lds_bn[WM + kf], mu = lds_bn[2 * WM + kf], rs = lds_bn[3 * WM + kf];
#pragma unroll
    for (int st = 0; st < ST; ++st) {
      const long i = base + 16 * st + s;
      const double xv = (xpre && q < XPRE && ST == 1) ? xpre[q < XPRE ? q : 0] : (kv && i < N) ? x[pl + i] : 0.0;
      H[q][st] = kv ? fma(xv, sc, sh) : 0.0;
      XH[q][st] = kv ? (xv - mu) * rs : 0.0;
    }
  }
  for (int l = 0; l < upto; ++l) {
    bw_dense(wm_all + (size_t)l * WM * WM, l == 0 ? (c.P + 3) / 4 : KQ, H, acc);
    if (l + 1 < upto) {
      const double* bn = lds_bn + (size_t)(l + 1) * 4 * WM;
#pragma unroll
      for (int jt = 0; jt < JT; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int j = 16 * jt + gq + 4 * r;
          const double sc = bn[j], sh = bn[WM + j], mu = bn[2 * WM + j], rs = bn[3 * WM + j];
#pragma unroll
          for (int st = 0; st < ST; ++st) {
            const double v = acc[jt][st][r];
            XH[4 * jt + r][st] = (v - mu) * rs;
            H[4 * jt + r][st] = fmax(fma(v, sc, sh), 0.0);
          }
        }
    }
  }
}

// per-lane BatchNorm sums (feature 4 q + (lane>>4), this lane's samples) -> one record of the workgroup
__device__ __forceinline__ void bww_emit_sums(double (&a1)[KQ], double (&a2)[KQ], double (*sm)[SREC], double* rec) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, gq = lane >> 4, s = lane & 15;
#pragma unroll
  for (int o = 8; o > 0; o >>= 1)
#pragma unroll
    for (int q = 0; q < KQ; ++q) {
      a1[q] += __shfl_xor(a1[q], o, 64);
      a2[q] += __shfl_xor(a2[q], o, 64);
    }
  if (s == 0) {
#pragma unroll
    for (int q = 0; q < KQ; ++q) {
      sm[wave][4 * q + gq] = a1[q];
      sm[wave][WM + 4 * q + gq] = a2[q];
    }
  }
  __syncthreads();
  if (tid < SREC) {
    double t = 0.0;
    for (int w = 0; w < NW; ++w) t += sm[w][tid];
    rec[tid] = t;
  }
}

// d a += W_rows^T dz for NQ consecutive rows of the row-major last-layer image (wr points at row 0, column lane&15):
// row q of the 4 column slots is one MFMA k-step.  The A operands are fetched one chunk of CH rows ahead of their
// use; the compiler barriers keep the scheduler from hoisting ALL loads (which costs > 100 registers).
template <int NQ, bool EXACT>
__device__ __forceinline__ void bww_da(const double* __restrict__ wr, const double (&dz)[NQ], int nq, nis_d4 (&da)[JT]) {
  constexpr int CH = 4, NC = (NQ + CH - 1) / CH;
  double a[2][CH][JT];
#pragma unroll
  for (int u = 0; u < CH; ++u)
#pragma unroll
    for (int jt = 0; jt < JT; ++jt) a[0][u][jt] = (u < NQ && (EXACT || u < nq)) ? wr[(size_t)u * WM + 16 * jt] : 0.0;
#pragma unroll
  for (int ch = 0; ch < NC; ++ch) {
    asm volatile("" ::: "memory");
    if (ch + 1 < NC) {
#pragma unroll
      for (int u = 0; u < CH; ++u) {
        const int q = (ch + 1) * CH + u;
#pragma unroll
        for (int jt = 0; jt < JT; ++jt) a[(ch + 1) & 1][u][jt] = (q < NQ && (EXACT || q < nq)) ? wr[(size_t)q * WM + 16 * jt] : 0.0;
      }
    }
#pragma unroll
    for (int u = 0; u < CH; ++u) {
      const int q = ch * CH + u;
      if (q < NQ && (EXACT || q < nq)) {
#pragma unroll
        for (int jt = 0; jt < JT; ++jt) da[jt] = mfma(a[ch & 1][u][jt], dz[q], da[jt]);
      }
    }
  }
}

// minibatch g = blockIdx.z of a launch: every per-minibatch pointer of the kernel's own copy of the arguments moves to its set
__device__ __forceinline__ void bww_batch(BwdK& k, BwdW* w) {
  const size_t g = blockIdx.z;
  k.x += g * k.x_gs;
  k.xo += g * k.x_gs;
  k.g += g * k.g_gs;
  k.bn += g * k.bn_gs;
  k.dyp += g * k.wsp_gs;
  k.alp += g * k.wsp_gs;
  k.dzp += g * k.wsp_gs;
  k.part += g * k.part_gs;
  if (k.sums) k.sums += g * k.sums_gs;
  if (w) {
    w->partg += g * w->partg_gs;
    if (w->prev) {
      w->prev += g * k.part_gs;
      w->sums_w += g * k.sums_gs;
    }
  }
}

__device__ __forceinline__ void bww_stage(const BwdK& k, double* lds_bn, int* lds_plane) {
  const int tid = threadIdx.x;
  if (tid < NIS_MAX_DIM) lds_plane[tid] = k.c.in_plane[tid];
  for (int idx = tid; idx < (k.c.L + 1) * 4 * WM; idx += TPB) lds_bn[idx] = k.bn[idx];
}

// Timing-only instrumentation (-DNIS_HEAD_STAMPS, never in the shipped library): wave 0 of workgroup 0 accumulates the 100 MHz clock per
// phase of the head and prints the sums (tools/build_variant.sh headstamps "-DNIS_WM=32 -DNIS_HEAD_STAMPS" flow_bwd_wm flow_bwd_wm32).
#ifdef NIS_HEAD_STAMPS
#define HS_T0() unsigned long long hst = __builtin_amdgcn_s_memrealtime(), hsacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define HS_T(k)                                                        \
  do {                                                                 \
    const unsigned long long now = __builtin_amdgcn_s_memrealtime();   \
    hsacc[k] += now - hst;                                             \
    hst = now;                                                         \
  } while (0)
#define HS_PRINT()                                                                                                                        \
  if (blockIdx.x == 0 && blockIdx.z == 0 && threadIdx.x == 0)                                                                             \
  printf("head: recompute %llu | last layer + exp %llu | bijector %llu | dZ store %llu | d a_L %llu | tail %llu (x10 ns)\n", hsacc[0], hsacc[1], \
         hsacc[2], hsacc[3], hsacc[4], hsacc[5])
#else
#define HS_T0()
#define HS_T(k)
#define HS_PRINT()
#endif

// ---------------------------------------------------------------------------------------------
// head: recompute the cell, bijector adjoint -> dZ planes, d a_L = W_L^T dZ, dy_L, sums of BN_L, a_L planes
// BATCH: the launch carries G > 1 minibatches as blockIdx.z.  Moving the kernel's pointers to minibatch g costs 15-30 VGPRs in these
// kernels (the C5 head goes from 52 to 85 spilled dwords, +13 % on a 2e5 minibatch), so a single minibatch runs the BATCH = false build.
template <int KIND, int KM, bool KEXACT, bool BATCH>
__global__ __launch_bounds__(TPB, KM > 16 ? 1 : WPS) void k_bwd_head_w(BwdK k, BwdW w) {
  extern __shared__ double mf_dyn[];                 // per wave: [ncb*out rows][16 samples] of exp(Z); ncb = CellK::mf_ncb columns per batch
  __shared__ double lds_bn[BNLDS];
  __shared__ int lds_plane[NIS_MAX_DIM];
  __shared__ double etab[NIS_EXP_N];
  __shared__ double sm[NW][SREC];
  if (BATCH) bww_batch(k, &w);
  const CellK& c = k.c;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, gq = lane >> 4, s = lane & 15;
  bww_stage(k, lds_bn, lds_plane);
  nis_exp_init(etab);
  __syncthreads();
  const int L = c.L, K = KEXACT ? KM : c.K, out = c.out;
  const long ld = k.ld, N = k.N;
  // columns per bijector batch: four (one per lane group) unless the ring of four does not fit the LDS -- then two or one, the other
  // lane groups run on dummy data with zero gradients (heads with many rows per column: PWLin with > 70 bins, PWQuad with > 32)
  const int ncb = c.mf_ncb;
  double* ring = mf_dyn + (size_t)wave * ncb * out * 16;
  const double* bnL = lds_bn + (size_t)L * 4 * WM;
  // BN_L sums live in LDS (sm[wave]: sum dy | sum dy*xhat), updated once per tile: 32 registers less in the bijector phase
  if (lane < SREC / 2) sm[wave][lane] = 0.0, sm[wave][SREC / 2 + lane] = 0.0;
  const long wstride = (long)gridDim.x * NW * 16 * ST;
  HS_T0();
  for (long base = ((long)blockIdx.x * NW + wave) * 16 * ST; base < N; base += wstride) {
    double H[KQ][ST], XH[KQ][ST];
    {
      nis_d4 acc[JT][ST];
      bww_forward(c, k.x, ld, N, base, lds_plane, lds_bn, w.wm_all, L, H, XH, acc);
#pragma unroll
      for (int jt = 0; jt < JT; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int j = 16 * jt + gq + 4 * r;
          const double sc = bnL[j], sh = bnL[WM + j], mu = bnL[2 * WM + j], rs = bnL[3 * WM + j];
#pragma unroll
          for (int st = 0; st < ST; ++st) {
            const double v = acc[jt][st][r];
            XH[4 * jt + r][st] = (v - mu) * rs;
            H[4 * jt + r][st] = fmax(fma(v, sc, sh), 0.0);            // a_L
          }
        }
    }
    HS_T(0);
#pragma unroll 1
    for (int st = 0; st < ST; ++st) {
      double Hs[KQ], XHs[KQ];
#pragma unroll
      for (int q = 0; q < KQ; ++q) {
        Hs[q] = H[q][0];
        XHs[q] = XH[q][0];
#pragma unroll
        for (int u = 1; u < ST; ++u) {
          Hs[q] = (st == u) ? H[q][u] : Hs[q];
          XHs[q] = (st == u) ? XH[q][u] : XHs[q];
        }
      }
      const long i = base + 16 * st + s;
      const bool iv = i < N;
      nis_d4 da[JT];
#pragma unroll
      for (int jt = 0; jt < JT; ++jt) da[jt] = nis_d4{0.0, 0.0, 0.0, 0.0};
      double jfprod = 1.0;
      const double gJ = iv ? k.g[(size_t)c.jplane * ld + i] : 0.0;
      const double Jout = iv ? k.xo[(size_t)c.jplane * ld + i] : 1.0;
#pragma unroll 1
      for (int cb = 0; cb * ncb < c.T; ++cb) {
        const int r0 = cb * ncb * out;
        const int r1 = min(c.R, r0 + ncb * out);
        // Row tiles of this batch.  As in the forward kernel's tile loop (flow_mfma.h): the NEXT tile's bias rows are requested at once (z
        // holds this tile's copy) and each A fragment right behind the MFMA that read it, into the same registers -- the chain is
        // dependent, so the requests go out 64 cycles apart and have the rest of the chain plus the exp phase to land (round 3 loaded
        // a second set of fragments up front: 16 registers more in a kernel that spills, and this loop ran at 2.1x its datapath time:
        // stamps in DESIGN 4f).  Interior tiles write their four rows unguarded at immediate offsets off one pointer; only a batch's
        // first / last tile (rows of the neighbouring batch, rows past R) takes the guarded form.
        double fa[KQ];
        nis_d4 zb;
        {
          const int jf = r0 / 16;
#pragma unroll
          for (int q = 0; q < KQ; ++q) fa[q] = w.wm_last[((size_t)jf * KQ + q) * 64 + lane];
#pragma unroll
          for (int r = 0; r < 4; ++r) zb[r] = w.bias_pad[16 * jf + gq + 4 * r];
        }
        for (int jt = r0 / 16; jt * 16 < r1; ++jt) {
          const int jn = (jt + 1) * 16 < r1 ? jt + 1 : jt;       // (last tile: re-requests itself, unused)
          nis_d4 z = zb;
#pragma unroll
          for (int r = 0; r < 4; ++r) zb[r] = w.bias_pad[16 * jn + gq + 4 * r];
#pragma unroll
          for (int q = 0; q < KQ; ++q) {
            z = mfma(fa[q], Hs[q], z);
            fa[q] = w.wm_last[((size_t)jn * KQ + q) * 64 + lane];
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA ...
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);      // ... then one VMEM read
          }
          double* rw = ring + (16 * jt + gq - r0) * 16 + s;          // row 16 jt + gq + 4 r of the batch's image -> rw[64 r]
          if (16 * jt >= r0 && 16 * jt + 16 <= r1) {                  // (wave-uniform) every row of the tile belongs to this batch
#pragma unroll
            for (int r = 0; r < 4; ++r) rw[64 * r] = (KIND == NIS_AFFINE && ((gq - r0) & 1)) ? z[r] : nis_exp(z[r], etab);   // affine: odd rows = raw shift (16 jt + 4 r is even)
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int j = 16 * jt + gq + 4 * r - r0;
              if (j >= 0 && j < ncb * out && 16 * jt + gq + 4 * r < c.R) rw[64 * r] = (KIND == NIS_AFFINE && (j & 1)) ? z[r] : nis_exp(z[r], etab);
            }
          }
        }
        __builtin_amdgcn_wave_barrier();
        HS_T(1);
        // (sample s, column slot gq) per lane.  Lanes without a column / sample run on dummy data with zero incoming
        // gradients, so that the MFMAs below stay in wave-uniform control flow and receive dZ = 0 from them.
        const int t = cb * ncb + gq;
        const bool valid = gq < ncb && t < c.T && iv;
        const int tc = min(t, c.T - 1);
        const size_t colp = (size_t)lds_plane[c.P + tc] * ld;
        const int row0 = tc * out;
        const double xraw = valid ? k.x[colp + i] : 0.5;
        const double gy = valid ? k.g[colp + i] : 0.0;
        const double gJv = valid ? gJ : 0.0;
        const double* zr = ring + (size_t)(gq < ncb ? gq : 0) * out * 16 + s;      // (lane groups without a column read slot 0: in bounds, masked)
        const double* wrow = c.wlast + (size_t)row0 * WM + s;
        double gx = 0.0, jf = 1.0;
        if constexpr (KIND == NIS_PWQUAD && KM > 32) {
          // 33 .. NIS_MAX_BINS bins: no register arrays -- the column's raw exponentials stay in its LDS ring (element q of this lane's
          // column and sample at zr[16 q]: heights 0 .. K, widths K+1 .. 2K), every pass over the bins is a runtime loop, and dZ replaces
          // the exponentials in place (the ring is rebuilt for the next column batch).  Same arithmetic as the register branch below.
          const bool clamped = xraw > 1.0 - 1e-6;
          const double xb = clamped ? 1.0 - 1e-6 : xraw;
          double* zw = const_cast<double*>(zr);
          const double* Wr = zr + (size_t)(K + 1) * 16;
          double wn = 0.0;
          for (int q = 0; q < K; ++q) wn += valid ? Wr[q * 16] : 1.0;
          int b = 0;
          double left = 0.0;
          if (valid) pwquad_bin_big<16>(xb, K, Wr, wn, &b, &left);
          left = left / wn;
          const double iwn = 1.0 / wn;
          double area = 0.0;
          for (int q = 0; q < K; ++q) area += valid ? (zr[q * 16] + zr[(q + 1) * 16]) / 2 * (Wr[q * 16] * iwn) : 1.0;
          const double iarea = 1.0 / area;
          const double Wb = valid ? Wr[b * 16] * iwn : 1.0;
          const double Vb = valid ? zr[b * 16] * iarea : 1.0, Vb1 = valid ? zr[(b + 1) * 16] * iarea : 1.0;
          const double alpha = (xb - left) / Wb;
          const double D = Vb1 - Vb;
          jf = lerp_torch(Vb, Vb1, alpha);
          const double gjf = gJv * Jout / jf;
          const double galpha = gy * jf * Wb + gjf * D;
          gx = clamped ? 0.0 : galpha / Wb;
          const double gleft = -galpha / Wb;
          const double hgy = gy / 2;
          const double A0 = gy * alpha * Wb * (1.0 - alpha / 2) + gjf * (1.0 - alpha);
          const double A1 = gy * alpha * alpha / 2 * Wb + gjf * alpha;
          const double B0 = gy * (alpha * alpha / 2 * D + alpha * Vb) - galpha * alpha / Wb;
          auto gv = [&](int q, double wm1, double wq) {
            return ((q >= 1 && q - 1 < b) ? hgy * wm1 : 0.0) + (q < b ? hgy * wq : 0.0) + (q == b ? A0 : 0.0) + (q == b + 1 ? A1 : 0.0);
          };
          auto gw0 = [&](int q, double vq, double vq1) { return (q < b ? hgy * (vq + vq1) + gleft : 0.0) + (q == b ? B0 : 0.0); };
          double S = 0.0, Sw = 0.0;
          if (valid) {
            for (int q = 0; q <= K; ++q) {
              const double wm1 = q >= 1 ? Wr[(q - 1) * 16] * iwn : 0.0;
              const double wq = q < K ? Wr[q * 16] * iwn : 0.0;
              S += gv(q, wm1, wq) * (zr[q * 16] * iarea);
            }
            for (int q = 0; q < K; ++q) {
              const double vq = zr[q * 16] * iarea, vq1 = zr[(q + 1) * 16] * iarea;
              Sw += (gw0(q, vq, vq1) - S * (vq + vq1) / 2) * (Wr[q * 16] * iwn);
            }
            // dZ in place, ascending q: v_q and w_{q-1} of the step are carried in registers, v_{q+1} / w_q are still the raw values
            double wprev = 0.0;
            for (int q = 0; q <= K; ++q) {
              const bool hasw = q < K;
              const double wq = hasw ? Wr[q * 16] * iwn : 0.0;
              const double vq = zr[q * 16] * iarea;
              const double vq1 = hasw ? zr[(q + 1) * 16] * iarea : 0.0;
              const double dzv = gv(q, wprev, wq) * vq - S * vq * (wprev + wq) / 2;
              if (hasw) zw[(size_t)(K + 1 + q) * 16] = wq * (gw0(q, vq, vq1) - S * (vq + vq1) / 2 - Sw);
              zw[q * 16] = dzv;
              wprev = wq;
            }
            for (int q = 0; q < 2 * K + 1; ++q) k.dzp[tmx(c.R, row0 + q, i)] = zr[q * 16];
          }
          // d a_L += W_rows^T dZ: every lane takes part in the MFMAs (lanes without a column / sample contribute dZ = 0)
          for (int q = 0; q < 2 * K + 1; ++q) {
            const double dzq = valid ? zr[q * 16] : 0.0;
#pragma unroll
            for (int jt = 0; jt < JT; ++jt) da[jt] = mfma(wrow[(size_t)q * WM + 16 * jt], dzq, da[jt]);
          }
        } else if (KIND == NIS_PWQUAD) {
          const bool clamped = xraw > 1.0 - 1e-6;
          const double xb = clamped ? 1.0 - 1e-6 : xraw;
          double V[KM + 1], W[KM];
#pragma unroll
          for (int q = 0; q < KM + 1; ++q) V[q] = (KEXACT || q <= K) ? (valid ? zr[q * 16] : 1.0) : 0.0;
#pragma unroll
          for (int q = 0; q < KM; ++q) W[q] = (KEXACT || q < K) ? (valid ? zr[(K + 1 + q) * 16] : 1.0) : 0.0;
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
          // the bin's values by index from the LDS image (raw exponentials, normalised like the arrays) instead of 3 K selects
          const double Wb = valid ? zr[(K + 1 + b) * 16] * iwn : 1.0;
          const double Vb = valid ? zr[b * 16] * iarea : 1.0, Vb1 = valid ? zr[(b + 1) * 16] * iarea : 1.0;
          const double alpha = (xb - left) / Wb;
          const double D = Vb1 - Vb;
          jf = lerp_torch(Vb, Vb1, alpha);
          const double gjf = gJv * Jout / jf;                  // J_out = J_in prod_t jf_t
          const double galpha = gy * jf * Wb + gjf * D;
          gx = clamped ? 0.0 : galpha / Wb;
          const double gleft = -galpha / Wb;
          // Gradients w.r.t. the normalised heights v_q and widths w_q are sparse in q (bins below the hit, the hit, its
          // right edge), so they are re-formed on the fly in each of the three passes instead of being kept in arrays:
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
            if (KEXACT || q <= K) {
              const double wm1 = q >= 1 ? W[q >= 1 ? q - 1 : 0] : 0.0;
              const double wq = (q < KM && (KEXACT || q < K)) ? W[q < KM ? q : 0] : 0.0;
              S += NIS_GV(q, wm1, wq) * V[q];
            }
          double Sw = 0.0;                                   // through w = softmax: Sw = sum_q gw(q) w_q
#pragma unroll
          for (int q = 0; q < KM; ++q)
            if (KEXACT || q < K) Sw += (NIS_GW0(q) - S * (V[q] + V[q + 1]) / 2) * W[q];
          // dZ in place: height rows into V, width rows into W (zero for lanes without a column / sample)
          double wprev = 0.0;
#pragma unroll
          for (int q = 0; q < KM + 1; ++q)
            if (KEXACT || q <= K) {                          // wave-uniform
              const bool hasw = q < KM && (KEXACT || q < K);
              const double wq = hasw ? W[q < KM ? q : 0] : 0.0;
              const double vq = V[q];
              const double dzv = NIS_GV(q, wprev, wq) * vq - S * vq * (wprev + wq) / 2;
              if (hasw) {
                const double dzw = wq * (NIS_GW0(q) - S * (vq + V[q + 1]) / 2 - Sw);
                W[q < KM ? q : 0] = valid ? dzw : 0.0;
              }
              V[q] = valid ? dzv : 0.0;
              wprev = wq;
            }
#undef NIS_GV
#undef NIS_GW0
          HS_T(2);
          if (valid) {
#pragma unroll
            for (int q = 0; q < KM + 1; ++q)
              if (KEXACT || q <= K) k.dzp[tmx(c.R, row0 + q, i)] = V[q];
#pragma unroll
            for (int q = 0; q < KM; ++q)
              if (KEXACT || q < K) k.dzp[tmx(c.R, row0 + K + 1 + q, i)] = W[q];
          }
          HS_T(3);
          bww_da<KM + 1, KEXACT>(wrow, V, K + 1, da);
          bww_da<KM, KEXACT>(wrow + (size_t)(K + 1) * WM, W, K, da);
          HS_T(4);
        } else if (KIND == NIS_AFFINE) {
          // AffineCoupling (coupling_cells.py:56-68); the ring holds [20^-1 s0 = exp(z0), z1] per column slot (rows interleaved
          // at pack time), the dZ planes and W_L^T use the blob's row order [scale | shift][T]
          const double s0 = 20.0 * (valid ? zr[0] : 1.0), z1 = valid ? zr[16] : 0.0;
          const double u = xraw * s0 + (z1 > 0.0 ? z1 : 0.0);
          const double q = 1.0 / (u * u + 1.0);
          jf = s0 * q;
          const double gjf = gJv * Jout / jf;
          const double gu = gy * (q / (M_PI / 2)) - gjf * (2.0 * u * s0 * q * q);
          gx = gu * s0;
          const double dz0 = valid ? (gu * xraw + gjf * q) * s0 : 0.0;
          const double dz1 = (valid && z1 > 0.0) ? gu : 0.0;
          const int rs = tc, rh = c.T + tc;
          if (valid) {
            k.dzp[tmx(c.R, rs, i)] = dz0;
            k.dzp[tmx(c.R, rh, i)] = dz1;
          }
#pragma unroll
          for (int jt = 0; jt < JT; ++jt) {
            da[jt] = mfma(c.wlast[(size_t)rs * WM + 16 * jt + s], dz0, da[jt]);
            da[jt] = mfma(c.wlast[(size_t)rh * WM + 16 * jt + s], dz1, da[jt]);
          }
        } else {
          // PWLin: Q_q = K e_q / tot, y = Q_b alpha + sum_{q<b} e_q / tot, jf = Q_b  (two streamed passes over the ring)
          const double ak = xraw * K;
          const double bf = floor(ak);
          const double alpha = (ak - bf) / K;
          const int b = (int)bf;
          double tot = 0.0, below = 0.0, eb = 0.0;
          for (int q = 0; q < K; ++q) {
            const double e = valid ? zr[q * 16] : 1.0;
            const bool hit = q == b;
            below = hit ? tot : below;
            eb = hit ? e : eb;
            tot += e;
          }
          const double Qb = eb / (tot / K);
          const double Qs = below / tot;
          jf = Qb;
          const double gjf = gJv * Jout / jf;
          const double gQb = gy * alpha + gjf, gQs = gy;
          gx = gy * Qb;
          for (int q0 = 0; q0 < K; q0 += 8) {                 // 8 rows per step: their A operands are in flight while dz is formed
            double a[8][JT], dzv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              const int q = min(q0 + u, K - 1);
#pragma unroll
              for (int jt = 0; jt < JT; ++jt) a[u][jt] = wrow[(size_t)q * WM + 16 * jt];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              const int q = q0 + u;
              const double e = (valid && q < K) ? zr[min(q, K - 1) * 16] : 1.0;
              const double pq = e / tot;
              const double dz = gQb * ((q == b ? Qb : 0.0) - Qb * pq) + gQs * ((q < b ? pq : 0.0) - Qs * pq);
              dzv[u] = (valid && q < K) ? dz : 0.0;
              if (valid && q < K) k.dzp[tmx(c.R, row0 + q, i)] = dzv[u];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
              for (int jt = 0; jt < JT; ++jt) da[jt] = mfma(a[u][jt], dzv[u], da[jt]);
          }
        }
        if (valid) {
          k.g[colp + i] = gx;
          jfprod *= jf;
        }
        __builtin_amdgcn_wave_barrier();
      }
      jfprod *= __shfl_xor(jfprod, 16, 64);
      jfprod *= __shfl_xor(jfprod, 32, 64);
      if (KIND == NIS_AFFINE) jfprod *= 1.0 / (M_PI / 2);
      if (gq == 0 && iv) k.g[(size_t)c.jplane * ld + i] = gJ * jfprod;   // dL/dJ_in = dL/dJ_out prod_t dy_t/dx_t
      // through ReLU to BN_L's output; sums of BN_L; a_L planes for dW_L
#pragma unroll
      for (int q = 0; q < KQ; ++q) {
        const int j = 4 * q + gq;
        const double dy = Hs[q] > 0.0 ? da[q >> 2][q & 3] : 0.0;
        if (iv) {
          k.dyp[tmx(WM, j, i)] = dy;
          k.alp[tmx(WM, j, i)] = Hs[q];
        }
        const double t1 = row16_sum(dy), t2 = row16_sum(dy * XHs[q]);
        if (s == 0) {
          sm[wave][j] += t1;
          sm[wave][WM + j] += t2;
        }
      }
      HS_T(5);
    }
  }
  HS_PRINT();
  __syncthreads();
  if (tid < SREC) {
    double t = 0.0;
    for (int wv = 0; wv < NW; ++wv) t += sm[wv][tid];
    k.part[(size_t)blockIdx.x * SREC + tid] = t;
  }
}

// ---------------------------------------------------------------------------------------------
// layer l (L..1): BN_l backward -> d h_l, dW_{l-1} += d h_l a_{l-1}^T (records), dy_{l-1} = relu'(.) W_{l-1}^T d h_l, sums of BN_{l-1}
template <bool BATCH>
__global__ __launch_bounds__(TPB, WPS) void k_bwd_layer_w(BwdK k, BwdW w, int l) {
  __shared__ double lds_bn[BNLDS];
  __shared__ int lds_plane[NIS_MAX_DIM];
  __shared__ double lds_m[SREC];
  __shared__ double sm[NW][SREC];
  __shared__ double stage[STAGE];                    // staging tiles; the epilogue's cross-wave sum of dW reuses it
  // widths <= 32: the weights this launch multiplies with -- the fragments of the dense layers it recomputes (at most LW) and W_{l-1}
  // row-major for d a -- are copied to LDS once per workgroup: every tile used to stream them from L2, one dependent round trip per layer
  // (stamps: the recompute was 40 of a launch's 64 us at 2e5 samples)
  constexpr int LW = 3;
  constexpr bool LDSW = WM <= 32;
  __shared__ double lds_w[LDSW ? LW * WM * WM : 1];
  __shared__ double lds_wt[LDSW ? WM * WM : 1];
  // the chain is the critical path; the GEMM's waves on the same SIMDs (side stream) are not: issue priority over them (7.14 -> 7.08 ms per C5 step)
  __builtin_amdgcn_s_setprio(3);
  if (BATCH) bww_batch(k, &w);
  const CellK& c = k.c;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, gq = lane >> 4, s = lane & 15;
  bww_stage(k, lds_bn, lds_plane);
  const long ld = k.ld, N = k.N;
  constexpr int ES = SREC <= 64 ? 64 : 128, NS = TPB / ES;      // record entries per slice of threads; slices
  if (w.prev) {                                      // (uniform) the producing launch's records, merged here: slice sl takes records sl, sl + NS, ...
    const int sl = tid / ES, e = tid % ES;
    double t = 0.0;
    if (e < SREC)
      for (int b0 = sl; b0 < w.nprev; b0 += NS * 8) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = b0 + NS * u < w.nprev ? w.prev[(size_t)(b0 + NS * u) * SREC + e] : 0.0;
#pragma unroll
        for (int u = 0; u < 8; ++u) t += v[u];
      }
    if (e < SREC) sm[sl][e] = t;
  } else if (tid < SREC) {
    lds_m[tid] = k.sums[tid] * k.invN;
  }
  const double* bnl = lds_bn + (size_t)l * 4 * WM;
  const bool staged = LDSW && l <= LW;               // (wave-uniform)
  if (staged) {                                      // (same round trip as the records above: one barrier for the whole prologue)
    for (int e = threadIdx.x; e < l * WM * WM; e += TPB) lds_w[e] = w.wm_all[e];
    for (int e = threadIdx.x; e < WM * WM; e += TPB) lds_wt[e] = c.wp[(size_t)(l - 1) * WM * WM + e];
  }
  __syncthreads();
  if (w.prev) {
    if (tid < SREC) {
      double tot = 0.0;
      for (int r = 0; r < NS; ++r) tot += sm[r][tid];
      lds_m[tid] = tot * k.invN;
      if (blockIdx.x == 0) w.sums_w[tid] = tot;
    }
    __syncthreads();                                 // (sm is the epilogue's scratch as well)
  }
  const double* wt = c.wp + (size_t)(l - 1) * WM * WM + (size_t)gq * WM + s;       // W_{l-1}[4 ks + gq][16 qt + s]
  const double* wts = lds_wt + (size_t)gq * WM + s;                                 // ... its LDS copy
  const int nqt = l == 1 ? (c.P + 15) / 16 : JT;
  double a1[KQ], a2[KQ];
#pragma unroll
  for (int q = 0; q < KQ; ++q) a1[q] = a2[q] = 0.0;
  // dW_{l-1}[row 16 it + gq + 4 r][column 16 jt + s]: the same tile layout as every other accumulator of this file
  nis_d4 dW[JT][JT];
#pragma unroll
  for (int it = 0; it < JT; ++it)
#pragma unroll
    for (int jt = 0; jt < JT; ++jt) dW[it][jt] = nis_d4{0.0, 0.0, 0.0, 0.0};
  double* const tdh = stage + (size_t)wave * 2 * WM * TS;
  double* const ta = tdh + WM * TS;
  const long wstride = (long)gridDim.x * NW * 16 * ST;
  HS_T0();
  // the first pass-through columns of the NEXT tile are requested while the current one is worked on (history planes: an HBM round trip
  // in front of every recompute otherwise)
  double xpre[XPRE];
  bww_prefetch_x(c, k.x, ld, N, ((long)blockIdx.x * NW + wave) * 16 * ST, lds_plane, xpre);
  for (long base = ((long)blockIdx.x * NW + wave) * 16 * ST; base < N; base += wstride) {
    double H[KQ][ST], XH[KQ][ST], DH[KQ][ST];
    {
      // the incoming dy of this tile does not depend on the recompute: requested first, so that its round trip (the plane the previous
      // launch wrote) runs beside the hidden stack instead of behind it (stamps: 29 of a launch's 81 us sat in this wait)
      double dyv[KQ];
      {
        const long i = base + s;
#pragma unroll
        for (int q = 0; q < KQ; ++q) dyv[q] = i < N ? k.dyp[tmx(WM, 16 * (q >> 2) + gq + 4 * (q & 3), i)] : 0.0;
      }
      nis_d4 acc[JT][ST];
      if (staged) bww_forward(c, k.x, ld, N, base, lds_plane, lds_bn, lds_w, l, H, XH, acc, xpre);
      else bww_forward(c, k.x, ld, N, base, lds_plane, lds_bn, w.wm_all, l, H, XH, acc, xpre);
      bww_prefetch_x(c, k.x, ld, N, base + wstride, lds_plane, xpre);
      HS_T(0);
#pragma unroll
      for (int jt = 0; jt < JT; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int j = 16 * jt + gq + 4 * r, q = 4 * jt + r;
          const double sc = bnl[j], mu = bnl[2 * WM + j], rs = bnl[3 * WM + j], m1 = lds_m[j], m2 = lds_m[WM + j];
#pragma unroll
          for (int st = 0; st < ST; ++st) {
            const long i = base + 16 * st + s;
            const bool iv = i < N;
            const double xh = (acc[jt][st][r] - mu) * rs;
            const double dy = dyv[q];
            const double dh = iv ? sc * (dy - m1 - xh * m2) : 0.0;          // scale_l = gamma_l * rstd_l
            DH[q][st] = dh;
          }
        }
    }
    // dW_{l-1} += d h_l a_{l-1}^T over this tile's 16 samples: both tiles cross the wave's LDS image ([feature][TS], the wave's own
    // accesses are served in order) and come back with the SAMPLES on the MFMA k index -- lane (i, k) supplies d h[16 it + i][4 m + k],
    // a[16 jt + i][4 m + k].  Rows of samples past N carry d h = 0.
    static_assert(ST == 1, "one sample tile per wave pass");
    HS_T(1);
#pragma unroll
    for (int q = 0; q < KQ; ++q) {
      tdh[(4 * q + gq) * TS + s] = DH[q][0];
      ta[(4 * q + gq) * TS + s] = H[q][0];
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      double bf[JT];
#pragma unroll
      for (int jt = 0; jt < JT; ++jt) bf[jt] = jt < nqt ? ta[(16 * jt + s) * TS + 4 * m + gq] : 0.0;
#pragma unroll
      for (int it = 0; it < JT; ++it) {
        const double af = tdh[(16 * it + s) * TS + 4 * m + gq];
#pragma unroll
        for (int jt = 0; jt < JT; ++jt)
          if (jt < nqt) dW[it][jt] = mfma(af, bf[jt], dW[it][jt]);
      }
    }
    __builtin_amdgcn_wave_barrier();
    HS_T(2);
#pragma unroll
    for (int qt = 0; qt < JT; ++qt) {
      nis_d4 dacc[ST];
#pragma unroll
      for (int st = 0; st < ST; ++st) dacc[st] = nis_d4{0.0, 0.0, 0.0, 0.0};
      if (qt < nqt) {
        if (staged) {                                  // (two loops: one pointer of each address space, no flat loads)
#pragma unroll
          for (int ks = 0; ks < KQ; ++ks) {
            const double a = wts[(4 * ks) * WM + 16 * qt];
#pragma unroll
            for (int st = 0; st < ST; ++st) dacc[st] = mfma(a, DH[ks][st], dacc[st]);
          }
        } else {
#pragma unroll
          for (int ks = 0; ks < KQ; ++ks) {
            const double a = wt[(size_t)(4 * ks) * WM + 16 * qt];
#pragma unroll
            for (int st = 0; st < ST; ++st) dacc[st] = mfma(a, DH[ks][st], dacc[st]);
          }
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = 16 * qt + gq + 4 * r, q = 4 * qt + r;
#pragma unroll
        for (int st = 0; st < ST; ++st) {
          const long i = base + 16 * st + s;
          const double dy = (l == 1 || H[q][st] > 0.0) ? dacc[st][r] : 0.0;   // BN_0 has no ReLU behind it
          if (i < N) k.dyp[tmx(WM, j, i)] = dy;
          a1[q] += dy;
          a2[q] = fma(dy, XH[q][st], a2[q]);
        }
      }
    }
    HS_T(3);
  }
#ifdef NIS_HEAD_STAMPS
  if (blockIdx.x == 0 && blockIdx.z == 0 && threadIdx.x == 0)
    printf("layer %d: recompute %llu | dy load + BN backward %llu | LDS stage + dW %llu | d a + dy store + sums %llu (x10 ns)\n", l, hsacc[0], hsacc[1],
           hsacc[2], hsacc[3]);
#endif
  bww_emit_sums(a1, a2, sm, k.part + (size_t)blockIdx.x * SREC);
  // dW records: wave after wave into one image (fixed summation order), then the workgroup's record in the GEMM's layout
  // (row groups of RPW * 16 rows x (WM + 1); the row-sum column stays zero: no bias behind a hidden Linear's BatchNorm)
  __syncthreads();
  constexpr bool ALLW = NW * WM * WM <= STAGE;        // every wave its own image (one barrier), summed in wave order below -- same sums
  if constexpr (ALLW) {
#pragma unroll
    for (int it = 0; it < JT; ++it)
#pragma unroll
      for (int jt = 0; jt < JT; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) stage[wave * WM * WM + (16 * it + gq + 4 * r) * WM + 16 * jt + s] = dW[it][jt][r];
    __syncthreads();
  } else {
    for (int wv = 0; wv < NW; ++wv) {
      if (wave == wv) {
#pragma unroll
        for (int it = 0; it < JT; ++it)
#pragma unroll
          for (int jt = 0; jt < JT; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int e = (16 * it + gq + 4 * r) * WM + 16 * jt + s;
              stage[e] = (wv == 0 ? 0.0 : stage[e]) + dW[it][jt][r];
            }
      }
      __syncthreads();
    }
  }
  constexpr int NGH = (WM + RPW * 16 - 1) / (RPW * 16);
  for (int e = tid; e < NGH * GREC; e += TPB) {
    const int grp = e / GREC, in = e - grp * GREC;
    const int row = grp * RPW * 16 + in / (WM + 1), col = in % (WM + 1);
    double v = 0.0;
    if (row < WM && col < WM) {
      v = stage[row * WM + col];
      if constexpr (ALLW)
        for (int wv = 1; wv < NW; ++wv) v += stage[wv * WM * WM + row * WM + col];
    }
    w.partg[((size_t)grp * w.nb + blockIdx.x) * GREC + in] = v;
  }
}

// Reduced records -> gradient blob in the reference's parameter order (include/nis_hip.h).
//   sums : pass p (0 = head, L - l + 1 = layer kernel l) holds the sums of BN_{L-p}: [sum dy | sum dy xhat]
//   glast: GEMM result of the last Linear (row groups of GREC); the hidden Linears' come from bww_reduce_hidden
// hidden_bias (NIS_AFFINE): 1 = train-mode BatchNorm, 2 = eval mode (`bn`: the saved scale | shift | mean | rstd records).
// G minibatches: minibatch g's reduced records sit sums_gs / gl_gs doubles behind minibatch g - 1's; the blob receives their SUM.
// Runs as extra workgroups of the cell's LAST chain launch (k_bwd_input_w: both only read the reduced records): as a launch of its own
// it cost 7 us per cell at one minibatch and 21 us at five.
struct ScatterW {
  const double *sums, *glast, *bn;
  double* gparams;
  int hidden_bias, G, nblk;
  long sums_gs, gl_gs;
  // the hidden Linears' weight gradients: the layer kernels' records (one area per layer), reduced HERE by nred more workgroups
  const double* parth;
  long parth_gs, layer_stride;       // doubles between the minibatches' / the layers' record areas
  int nbl, nquads, nred;             // records per row group; workgroups per layer (four entries of WM x (WM + 1) each); nquads * L
};
__device__ __forceinline__ void bww_scatter(const CellK& c, const ScatterW& a, int tid, int nth) {
  const int L = c.L, G = a.G;
  auto over_g = [&](const double* p, long gs) {          // minibatch order; four loads in flight
    double t = 0.0;
    int g = 0;
    for (; g + 4 <= G; g += 4) {
      const double v0 = p[(size_t)g * gs], v1 = p[(size_t)(g + 1) * gs], v2 = p[(size_t)(g + 2) * gs], v3 = p[(size_t)(g + 3) * gs];
      t += v0;
      t += v1;
      t += v2;
      t += v3;
    }
    for (; g < G; ++g) t += p[(size_t)g * gs];
    return t;
  };
  // ONE flat index over the blob ([gamma | beta | W | (bias)] per layer, then the last bias): every thread's loads go out together (walking
  // the layers section by section was a chain of ~12 dependent round trips: 15 us on the cell's critical path)
  auto seg = [&](int l) {
    const size_t F = cellk_feat(c, l), O = l < L ? c.width[l] : c.R;
    return 2 * F + O * F + ((a.hidden_bias && l < L) ? O : 0);
  };
  size_t tot = c.R;
  for (int l = 0; l <= L; ++l) tot += seg(l);
  for (size_t e = tid; e < tot; e += nth) {
    size_t rem = e;
    int l = 0;
    for (; l <= L; ++l) {
      const size_t sg = seg(l);
      if (rem < sg) break;
      rem -= sg;
    }
    double v;
    if (l > L) {                              // bias of the last Linear: row sums of dZ
      const int r = (int)rem;
      v = over_g(a.glast + (size_t)(r / (RPW * 16)) * GREC + (r % (RPW * 16)) * (WM + 1) + WM, a.gl_gs);
    } else {
      const size_t F = cellk_feat(c, l), O = l < L ? c.width[l] : c.R;
      const double* sl = a.sums + (size_t)(L - l) * SREC;
      if (rem < F) {
        v = over_g(sl + WM + rem, a.sums_gs);
      } else if (rem < 2 * F) {
        v = over_g(sl + (rem - F), a.sums_gs);
      } else if (rem < 2 * F + O * F) {
        if (l < L) continue;                  // hidden Linear: bww_reduce_hidden
        const size_t ee = rem - 2 * F;
        const int r = (int)(ee / F), q = (int)(ee % F);
        v = over_g(a.glast + (size_t)(r / (RPW * 16)) * GREC + (r % (RPW * 16)) * (WM + 1) + q, a.gl_gs);
      } else {                                // NIS_AFFINE: a bias in front of a train-mode BatchNorm has gradient sum(dh) == 0;
        const int o = (int)(rem - 2 * F - O * F);                      // eval mode: scale_{l+1} * sum dy_{l+1}  (one BatchNorm record for all minibatches)
        const double* s1 = a.sums + (size_t)(L - l - 1) * SREC;
        v = a.hidden_bias == 2 ? a.bn[((l + 1) * 4 + 0) * WM + o] * over_g(s1 + o, a.sums_gs) : 0.0;
      }
    }
    a.gparams[e] = v;
  }
}

// BN_0 backward: gradient of the pass-through columns (workgroups [0, gridDim.x - sa.nblk)); the rest: bww_scatter.
// (16 us per C5 cell at 2e5 with or without the scatter role, with the columns one by one or eight at a time: it moves 51 MB.)
// dW_m of hidden Linear m = sum over the layer kernel's per-workgroup records (and over the minibatches), straight into the blob: a
// workgroup owns four entries, 64 slices of the records each, fixed order -- what k_bwd_reduce2_w did for them beside the BatchNorm sums
// in the middle of the chain, where it shared the chip with the side-stream GEMM (11-14 us per launch instead of 6)
__device__ __forceinline__ void bww_reduce_hidden(const CellK& c, const ScatterW& a, int wg) {
  __shared__ double sm[4][NW][4];
  const int m = wg / a.nquads, quad = wg - m * a.nquads;
  const int el = threadIdx.x & 3, slice = threadIdx.x >> 2, e = quad * 4 + el;
  const int row = e / (WM + 1), col = e - row * (WM + 1);
  const int F = cellk_feat(c, m), O = c.width[m];
  const bool ok = row < O && col < F;
  size_t off = 0;                                        // blob offset of W_m
  for (int l = 0; l < m; ++l) {
    const size_t Fl = cellk_feat(c, l), Ol = c.width[l];
    off += 2 * Fl + Ol * Fl + (a.hidden_bias ? Ol : 0);
  }
  off += 2 * (size_t)F;
  const size_t in = (size_t)(row / (RPW * 16)) * a.nbl * GREC + (size_t)(row % (RPW * 16)) * (WM + 1) + col;
  double tot = 0.0;
  constexpr int GC = 4;                                  // minibatches per pass: their loads and reductions run side by side
  for (int g0 = 0; g0 < a.G; g0 += GC) {
    double t[GC];
#pragma unroll
    for (int gg = 0; gg < GC; ++gg) {
      t[gg] = 0.0;
      if (ok && g0 + gg < a.G) {
        const double* part = a.parth + (size_t)(g0 + gg) * a.parth_gs + (size_t)m * a.layer_stride + in;
        for (int b0 = slice; b0 < a.nbl; b0 += 512) {
          double v[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) v[u] = b0 + 64 * u < a.nbl ? part[(size_t)(b0 + 64 * u) * GREC] : 0.0;
#pragma unroll
          for (int u = 0; u < 8; ++u) t[gg] += v[u];
        }
      }
    }
#pragma unroll
    for (int gg = 0; gg < GC; ++gg) {
#pragma unroll
      for (int o = 4; o < 64; o <<= 1) t[gg] += __shfl_xor(t[gg], o, 64);
      if ((threadIdx.x & 63) < 4) sm[gg][threadIdx.x >> 6][el] = t[gg];
    }
    __syncthreads();
    if (slice == 0) {
#pragma unroll
      for (int gg = 0; gg < GC; ++gg)
        if (g0 + gg < a.G) tot += (sm[gg][0][el] + sm[gg][1][el]) + (sm[gg][2][el] + sm[gg][3][el]);
    }
    __syncthreads();
  }
  if (slice == 0 && ok) a.gparams[off + (size_t)row * F + col] = tot;
}

__global__ __launch_bounds__(TPB) void k_bwd_input_w(BwdK k, ScatterW sa) {
  const int gi = (int)gridDim.x - sa.nblk - sa.nred;
  if ((int)blockIdx.x >= gi) {
    if (blockIdx.z != 0) return;
    if ((int)blockIdx.x < gi + sa.nblk) bww_scatter(k.c, sa, ((int)blockIdx.x - gi) * TPB + threadIdx.x, sa.nblk * TPB);
    else bww_reduce_hidden(k.c, sa, (int)blockIdx.x - gi - sa.nblk);
    return;
  }
  bww_batch(k, nullptr);
  const CellK& c = k.c;
  const double invN = k.invN;
  const long stride = (long)gi * TPB;
  for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < k.N; i += stride)
    for (int p0 = 0; p0 < c.P; p0 += 8) {          // eight columns at a time, EVERY load (the columns' BatchNorm records and sums too) before
      double xv[8], dv[8], gv[8];                  // the first store: g may alias anything as far as the compiler knows, so column by column
      double sc[8], mu[8], rs[8], m1[8], m2[8];    // each update waited for the previous store and re-read its scalars behind it
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int p = p0 + u < c.P ? p0 + u : p0;
        const size_t at = (size_t)c.in_plane[p] * k.ld + i;
        xv[u] = k.x[at];
        dv[u] = k.dyp[tmx(WM, p, i)];
        gv[u] = k.g[at];
        sc[u] = k.bn[p];
        mu[u] = k.bn[2 * WM + p];
        rs[u] = k.bn[3 * WM + p];
        m1[u] = k.sums[p] * invN;
        m2[u] = k.sums[WM + p];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (p0 + u < c.P) {
          const double xh = (xv[u] - mu[u]) * rs[u];
          k.g[(size_t)c.in_plane[p0 + u] * k.ld + i] = gv[u] + sc[u] * (dv[u] - m1[u] - xh * m2[u] * invN);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// dW_L[r][q] = sum_i dZ[r][i] a_L[q][i] (+ row sums of dZ = db_L): the MFMA k index runs over samples.  Lane l owns samples
// 4 (l>>4) .. 4 (l>>4) + 3 of a 16-sample tile (one 32-byte load per row of the tile-major buffers) and feeds sample 4 (l>>4) + m to
// k-step m.  The four waves of a workgroup split the ROWS (row tiles w + 4 n of the workgroup's block of 4 XRT tiles) and walk the same
// sample tiles, so dZ is read once and a_L once per workgroup (its four waves share the lines; round 3 gave every wave its own samples
// and one row group, and re-read a_L once per row group: 5 x at C5).  The operands of the next tile are in flight while the current one
// is multiplied; a wave owns its rows alone, so its accumulators ARE the workgroup's record (no cross-wave sum).
constexpr int XRT = WM <= 32 ? 5 : 2;                     // row tiles per wave: XRT * JT accumulator tiles of 8 registers
struct XtyOps {
  nis_d4 a[XRT], b[JT];
};
__device__ __forceinline__ void xty_load(const double* __restrict__ Ap, int R, int rt0, const double* __restrict__ Bp, long N, long tile, XtyOps& o) {
  const int lane = threadIdx.x & 63, gq = lane >> 4, s = lane & 15;
  const long i0 = tile * 16 + 4 * gq;
  const nis_d4 zero = {0.0, 0.0, 0.0, 0.0};
  const bool in = i0 < N;                               // the padded tail of a plane is allocated but not initialised
#pragma unroll
  for (int qt = 0; qt < JT; ++qt) o.b[qt] = in ? *reinterpret_cast<const nis_d4*>(Bp + ((size_t)tile * WM + 16 * qt + s) * 16 + 4 * gq) : zero;
#pragma unroll
  for (int n = 0; n < XRT; ++n) {
    const int row = 16 * (rt0 + NW * n) + s;
    o.a[n] = (in && row < R) ? *reinterpret_cast<const nis_d4*>(Ap + ((size_t)tile * R + row) * 16 + 4 * gq) : zero;
  }
  if (in && i0 + 3 >= N) {
#pragma unroll
    for (int m = 1; m < 4; ++m)
      if (i0 + m >= N) {
#pragma unroll
        for (int qt = 0; qt < JT; ++qt) o.b[qt][m] = 0.0;
#pragma unroll
        for (int n = 0; n < XRT; ++n) o.a[n][m] = 0.0;
      }
  }
}

__global__ __launch_bounds__(TPB, 2) void k_bwd_xty_w(const double* __restrict__ Ap, int R, const double* __restrict__ Bp, long N,
                                                      double* __restrict__ part, int ngroups, long wsp_gs, long part_gs) {
  Ap += (size_t)blockIdx.z * wsp_gs;                 // minibatch g = blockIdx.z: its planes, its record area
  Bp += (size_t)blockIdx.z * wsp_gs;
  part += (size_t)blockIdx.z * part_gs;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, gq = lane >> 4, s = lane & 15;
  const int rt0 = blockIdx.y * NW * XRT + wave;      // this wave's row tiles: rt0 + NW n
  nis_d4 acc[XRT][JT];
  double rsum[XRT];
#pragma unroll
  for (int n = 0; n < XRT; ++n) {
    rsum[n] = 0.0;
#pragma unroll
    for (int qt = 0; qt < JT; ++qt) acc[n][qt] = nis_d4{0.0, 0.0, 0.0, 0.0};
  }
  // tiles in DESCENDING order: the head wrote the high tiles last, and what is left of them in the Infinity Cache is read before it is
  // evicted (135 -> 117 us per C5 cell at 2e5; the kernel is bound by its reads either way: with the MFMAs removed it takes 132 us, and
  // staging the tiles through LDS in whole rows -- global_load_lds, swizzled image, ds_read_b128 fragments -- gave 120 us in this order)
  const long ntiles = (N + 15) / 16, tstride = gridDim.x;
  long tile = ntiles - 1 - blockIdx.x;
  XtyOps cur, nxt;
  if (tile >= 0) xty_load(Ap, R, rt0, Bp, N, tile, cur);
  for (; tile >= 0; tile -= tstride) {
    const bool more = tile - tstride >= 0;
    if (more) xty_load(Ap, R, rt0, Bp, N, tile - tstride, nxt);
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int n = 0; n < XRT; ++n) {
        rsum[n] += cur.a[n][m];
#pragma unroll
        for (int qt = 0; qt < JT; ++qt) acc[n][qt] = mfma(cur.a[n][m], cur.b[qt][m], acc[n][qt]);
      }
    if (more) cur = nxt;
  }
  // the wave's rows of the workgroup's record: [row group of RPW * 16][workgroup][rows x (WM products | row sum)]; rows past R hold zeros
  // (their operands were zero), row tiles past the last group are not part of any record
#pragma unroll
  for (int n = 0; n < XRT; ++n) {
    const int rt = rt0 + NW * n;
    if (rt * 16 >= ngroups * RPW * 16) continue;
    double rs = rsum[n];
    rs += __shfl_xor(rs, 16, 64);
    rs += __shfl_xor(rs, 32, 64);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 16 * rt + gq + 4 * r;
      double* rec = part + ((size_t)(row / (RPW * 16)) * gridDim.x + blockIdx.x) * GREC + (size_t)(row % (RPW * 16)) * (WM + 1);
#pragma unroll
      for (int qt = 0; qt < JT; ++qt) rec[16 * qt + s] = acc[n][qt][r];
    }
    if (gq == 0) {
      const int row = 16 * rt + s;
      part[((size_t)(row / (RPW * 16)) * gridDim.x + blockIdx.x) * GREC + (size_t)(row % (RPW * 16)) * (WM + 1) + WM] = rs;
    }
  }
}

// records -> their sum: a workgroup owns 4 elements, 64 slices of the records each; fixed order (bitwise reproducible).
// Two reductions in ONE launch (rows [0, a.ny) of the grid's y extent run job a, the rest job b): the BatchNorm sums of a chain
// kernel and the weight-gradient GEMM's records always come in pairs, and a launch costs more than either reduction.
struct RedJob {
  const double* part;
  double* out;
  int nparts, nacc, ny;
  long part_gs, out_gs;
};
// EL entries per workgroup, 256 / EL slices of the records each: 4 x 64 for the hundreds of records of a big batch (one round trip of
// eight loads per thread), 16 x 16 for small batches (a launch of 4-entry workgroups over the last layer's records was 13 000 workgroups
// for 100 records each: 25 us)
template <int EL>
__global__ __launch_bounds__(TPB) void k_bwd_reduce2_w(RedJob a, RedJob b) {
  constexpr int NSL = TPB / EL;                          // slices
  __shared__ double sm[NW][EL];
  const bool second = (int)blockIdx.y >= a.ny;
  const RedJob& j = second ? b : a;
  const int y = (int)blockIdx.y - (second ? a.ny : 0);
  const int el = threadIdx.x % EL, slice = threadIdx.x / EL;
  const int e = blockIdx.x * EL + el;
  const double* part = j.part + (size_t)blockIdx.z * j.part_gs + (size_t)y * j.nparts * j.nacc;
  double* out = j.out + (size_t)blockIdx.z * j.out_gs;
  double t = 0.0;
  if (e < j.nacc) {                                   // eight records per round trip, summed in record order
    for (int b0 = slice; b0 < j.nparts; b0 += 8 * NSL) {
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = b0 + NSL * u < j.nparts ? part[(size_t)(b0 + NSL * u) * j.nacc + e] : 0.0;
#pragma unroll
      for (int u = 0; u < 8; ++u) t += v[u];
    }
  }
  // the slices of an entry: those inside each wave (lanes differing in the bits above EL), then the four waves' partials
#pragma unroll
  for (int o = EL; o < 64; o <<= 1) t += __shfl_xor(t, o, 64);
  if ((threadIdx.x & 63) < EL) sm[threadIdx.x >> 6][el] = t;
  __syncthreads();
  if (slice == 0) sm[0][el] = (sm[0][el] + sm[1][el]) + (sm[2][el] + sm[3][el]);
  if (slice == 0 && e < j.nacc) out[(size_t)y * j.nacc + e] = sm[0][el];
}

// ---------------------------------------------------------------------------------------------
int device_cus() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0)
      cus = n;
    else
      cus = 256;
  }
  return cus;
}

size_t ring_bytes(int out, int ncb) { return (size_t)NW * ncb * out * 16 * sizeof(double); }
constexpr size_t HEAD_FIXED_LDS = (size_t)(BNLDS + NW * SREC + 64) * sizeof(double) + NIS_MAX_DIM * sizeof(int) + 1024;   // static LDS of the head
// columns per bijector batch of the head: the largest of 4 / 2 / 1 whose ring fits beside the static LDS (0: none does)
int pick_ncb(int out) {
  for (int n = NCB; n >= 1; n >>= 1)
    if (ring_bytes(out, n) + HEAD_FIXED_LDS <= 160 * 1024) return n;
  return 0;
}
long ldw_of(long N) { return (N + 15) / 16 * 16; }
int groups_of(int rows) { return (rows + RPW * 16 - 1) / (RPW * 16); }

// side stream of the last-layer GEMM (one per process: the library is driven from one thread)
struct Side {
  hipStream_t stream = nullptr;
  hipEvent_t fork = nullptr, join = nullptr;
  bool ok = false;
};
Side& side() {                                   // per device; created on first use (never inside a stream capture: a captured step is
  static Side sd[32];                            // preceded by its eager warm-up run)
  static bool tried[32] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 32) dev = 0;
  Side& d = sd[dev];
  if (!tried[dev]) {
    tried[dev] = true;
    d.ok = hipStreamCreateWithFlags(&d.stream, hipStreamNonBlocking) == hipSuccess &&
           hipEventCreateWithFlags(&d.fork, hipEventDisableTiming) == hipSuccess &&
           hipEventCreateWithFlags(&d.join, hipEventDisableTiming) == hipSuccess;
  }
  return d;
}

struct Plan {
  int nb;            // workgroups of the head launch
  int nbl;           // workgroups of the layer launches
  bool side;         // the last layer's GEMM runs beside the layer chain (bwd_wide_run)
  int ngl, ngh;      // row groups of the last-layer / hidden GEMMs
  int nxl, nsg;      // last layer's GEMM: workgroups over the sample tiles, blocks of NW * XRT row tiles (grid y)
  size_t part;       // doubles of the record area of the chain kernels (BatchNorm sums: nb x SREC)
  size_t partx;      // doubles of the record area of the last layer's GEMM (its own: that launch runs beside the layer chain)
  size_t parth;      // doubles of the record area of the hidden layers' weight gradients (one record per workgroup of k_bwd_layer_w)
};

// share of the chip the side-stream GEMM takes, in quarter workgroups per CU (the layer launches get the remaining slots); measured at
// C5 / 2e5: 1 -> 8.96 ms per step (the GEMM becomes the critical path), 2 -> 7.42, 3 -> 7.33, 4 -> 7.45, 5 -> 7.40, 6 -> 7.45; again with the
// layer kernel's issue priority in place: 2 -> 7.30, 3 -> 7.17, 4 -> 7.32, 5 -> 7.65
constexpr int XTY_WG_PER_CU4 = 3;
constexpr long SIDE_MIN_SAMPLES = 65536;
Plan plan_for(int R, long N, int G) {         // the G minibatches of a launch (blockIdx.z) share the chip
  Plan p;
  const int cus = device_cus();
  const long per_wg = (long)NW * 16 * ST;
  p.nb = (int)std::max<long>(1, std::min<long>((N + per_wg - 1) / per_wg, std::max<long>(1, (long)cus * 2 / G)));
  p.ngl = groups_of(R);
  p.ngh = groups_of(WM);
  p.nsg = (p.ngl * RPW + NW * XRT - 1) / (NW * XRT);
  // big batches: the last layer's GEMM on the side stream, ONE of its workgroups per CU and one of the layer kernel's in the other slot
  // (two rounds of 256 layer workgroups would pay the launch's prologue twice).  Small batches keep one stream: the fork / join costs
  // more than the GEMM's few microseconds (C5 topology, 5 x 2000: 2.4 -> 2.8 ms per step with it)
  p.side = (long)N * G >= SIDE_MIN_SAMPLES;
  p.nxl = (int)std::max<long>(1, std::min<long>((N + 15) / 16, std::max(1, cus * (p.side ? XTY_WG_PER_CU4 : 8) / 4 / (p.nsg * G))));
  p.nbl = p.side ? (int)std::max<long>(1, std::min<long>(p.nb, ((long)cus * 2 - (long)p.nxl * p.nsg * G) / G)) : p.nb;      // the slots the GEMM leaves
  p.part = (size_t)p.nb * SREC;
  p.partx = (size_t)p.ngl * p.nxl * GREC;
  p.parth = (size_t)p.ngh * p.nbl * GREC;
  return p;
}
}  // namespace

template <>
bool bwd_wide_supported<NIS_WM>(const nis_cell_desc& cd) {
  const int out = cd.kind == NIS_PWQUAD ? 2 * cd.K + 1 : cd.kind == NIS_AFFINE ? 2 : cd.K;
  return pick_ncb(out) > 0;            // (PWQuad with more than 32 bins: the column's LDS ring doubles as the dZ array)
}

template <>
size_t bwd_wide_workspace_doubles<NIS_WM>(const nis_cell_desc& cd, int64_t N, int G) {
  const int T = cd.d - cd.P;
  const int out = cd.kind == NIS_PWQUAD ? 2 * cd.K + 1 : cd.kind == NIS_AFFINE ? 2 : cd.K;
  const int R = T * out;
  const Plan p = plan_for(R, (long)N, G);
  return (size_t)G * ((size_t)(2 * WM + R) * ldw_of((long)N) + 2 * p.part + p.partx + (size_t)cd.n_hidden * p.parth + (size_t)(cd.n_hidden + 2) * SREC +
                      (size_t)(p.ngl + cd.n_hidden * p.ngh) * GREC);
}

template <>
int bwd_wide_run<NIS_WM>(const CellK& c, BwdK k, const MfmaW& mw, double* ws, double* gparams, hipStream_t s) {
  const long N = k.N;
  const int L = c.L, G = std::max(1, k.G);
  const Plan p = plan_for(c.R, N, G);
  BwdW w;
  w.wm_all = mw.wm_all;
  w.wm_last = mw.wm_last;
  w.bias_pad = mw.bias_pad;
  const long ldw = ldw_of(N);                      // workspace planes: 16-sample aligned rows (vector loads in the GEMM)
  // per minibatch (blockIdx.z): planes [dy | a_L | dZ], then -- behind the planes of all minibatches -- the record areas, the
  // reduced BatchNorm sums and the reduced GEMM results, each G sets
  const size_t wsp = (size_t)(2 * WM + c.R) * ldw;
  const size_t nsum = (size_t)(L + 2) * SREC, ngem = (size_t)(p.ngl + L * p.ngh) * GREC;
  k.ldw = ldw;
  k.wsp_gs = (long)wsp;
  k.part_gs = (long)p.part;
  k.sums_gs = (long)nsum;
  k.dyp = ws;
  k.alp = ws + (size_t)WM * ldw;
  k.dzp = ws + (size_t)2 * WM * ldw;
  double* part = ws + (size_t)G * wsp;                        // two record areas: a layer kernel that merges its predecessor's records
  double* part2 = part + (size_t)G * p.part;                  //   in its prologue writes its own into the other one
  double* partx = part2 + (size_t)G * p.part;
  double* parth = partx + (size_t)G * p.partx;                // minibatch g, hidden Linear m: parth + (g * L + m) * p.parth
  double* sums = parth + (size_t)G * L * p.parth;
  double* glast = sums + (size_t)G * nsum;                 // minibatch g: glast + g * ngem, its hidden-layer results ngl * GREC behind
  k.part = part;
  k.sums = nullptr;
  w.prev = nullptr;
  w.sums_w = nullptr;
  w.nprev = 0;
  w.partg = parth;
  w.partg_gs = (long)((size_t)L * p.parth);
  w.nb = p.nbl;
  CellK& cm = k.c;                                  // (k is this function's own copy)
  const unsigned Gz = (unsigned)G;
  // BatchNorm sums of a chain kernel (job a) and / or the records of a weight-gradient GEMM (job b): ONE launch
  auto reduce_jobs = [&](hipStream_t st, double* sums_out, int nparts, const double* grec, long grec_gs, int ngroups, int nx, double* gemm_out) {
    RedJob a = {k.part, sums_out, nparts, SREC, sums_out ? 1 : 0, (long)p.part, (long)nsum};      // (the area the last producer wrote)
    RedJob b = {grec, gemm_out, nx, GREC, gemm_out ? ngroups : 0, grec_gs, (long)ngem};
    const int nacc = b.ny ? GREC : SREC;
    if (std::max(a.ny ? nparts : 0, b.ny ? nx : 0) <= 128)
      hipLaunchKernelGGL(k_bwd_reduce2_w<16>, dim3((nacc + 15) / 16, a.ny + b.ny, Gz), dim3(TPB), 0, st, a, b);
    else
      hipLaunchKernelGGL(k_bwd_reduce2_w<4>, dim3((nacc + 3) / 4, a.ny + b.ny, Gz), dim3(TPB), 0, st, a, b);
  };
  cm.mf_ncb = pick_ncb(c.out);
  if (cm.mf_ncb < 1) return NIS_E_UNSUPPORTED;
  const size_t ring = ring_bytes(c.out, cm.mf_ncb);
  const int km = c.K <= 4 ? 4 : c.K <= 8 ? 8 : c.K <= 16 ? 16 : 32;
#define NIS_HEAD(KIND, KMV, EX)                                                                                        \
  do {                                                                                                                 \
    if (G > 1) hipLaunchKernelGGL((k_bwd_head_w<KIND, KMV, EX, true>), dim3(p.nb, 1, Gz), dim3(TPB), ring, s, k, w);    \
    else hipLaunchKernelGGL((k_bwd_head_w<KIND, KMV, EX, false>), dim3(p.nb, 1, 1), dim3(TPB), ring, s, k, w);         \
  } while (0)
  if (c.kind == NIS_PWLIN)
    NIS_HEAD(NIS_PWLIN, 4, false);
  else if (c.kind == NIS_AFFINE)
    NIS_HEAD(NIS_AFFINE, 4, false);
  else if (c.K == 4)
    NIS_HEAD(NIS_PWQUAD, 4, true);
  else if (c.K == 8)
    NIS_HEAD(NIS_PWQUAD, 8, true);
  else if (c.K == 16)
    NIS_HEAD(NIS_PWQUAD, 16, true);
  else if (km == 4)
    NIS_HEAD(NIS_PWQUAD, 4, false);
  else if (km == 8)
    NIS_HEAD(NIS_PWQUAD, 8, false);
  else if (km == 16)
    NIS_HEAD(NIS_PWQUAD, 16, false);
  else if (c.K <= 32)
    NIS_HEAD(NIS_PWQUAD, 32, false);
  else
    NIS_HEAD(NIS_PWQUAD, NIS_KM_BIG, false);
#undef NIS_HEAD
  // dW_L = dZ a_L^T, db_L = row sums of dZ, and the reduction of its records: on the SIDE stream.  Nothing in the layer chain reads what
  // they write (only the scatter at the end of the cell does), the GEMM is bound by its 422 MB of dZ reads and the chain by latencies, so
  // the two share the chip: one GEMM workgroup per CU, the chain's workgroups in the other slot (serially the GEMM's 117 us stood in
  // front of the chain).  Joined before the cell's last launch; the next cell's head overwrites dZ / a_L only after that.
  Side* sd = nullptr;
  if (p.side) {
    sd = &side();
    if (!sd->ok) return NIS_E_HIP;
  }
  bool side_err = false;
  auto launch_gemm = [&]() {
    hipStream_t sx = sd ? sd->stream : s;
    if (sd && (hipEventRecord(sd->fork, s) != hipSuccess || hipStreamWaitEvent(sx, sd->fork, 0) != hipSuccess)) side_err = true;
    hipLaunchKernelGGL(k_bwd_xty_w, dim3(p.nxl, p.nsg, Gz), dim3(TPB), 0, sx, (const double*)k.dzp, c.R, (const double*)k.alp, N, partx,
                       p.ngl, (long)wsp, (long)p.partx);
    if (sd) {
      reduce_jobs(sx, nullptr, 0, partx, (long)p.partx, p.ngl, p.nxl, glast);
      if (hipEventRecord(sd->join, sx) != hipSuccess) side_err = true;
    }
  };
  launch_gemm();             // (forked behind the first reduction instead: equal; behind the first layer launch: 7.1 -> 7.3 ms per C5 step)
  if (sd) reduce_jobs(s, sums, p.nb, nullptr, 0, 0, 0, nullptr);
  else reduce_jobs(s, sums, p.nb, partx, (long)p.partx, p.ngl, p.nxl, glast);      // one stream: the head's sums and the GEMM's records in ONE launch
  const bool pmerge = !sd;                               // small batches: the layer kernels below the first merge their predecessor's records
  for (int l = L; l >= 1; --l) {
    const int pass = L - l + 1;
    k.sums = sums + (size_t)(pass - 1) * SREC;
    w.partg = parth + (size_t)(l - 1) * p.parth;
    if (pmerge && l < L) {
      w.prev = k.part;                                   // what layer l + 1 wrote
      w.nprev = p.nbl;
      w.sums_w = sums + (size_t)(pass - 1) * SREC;
      k.part = k.part == part ? part2 : part;
    }
    if (G > 1) hipLaunchKernelGGL(k_bwd_layer_w<true>, dim3(p.nbl, 1, Gz), dim3(TPB), 0, s, k, w, l);
    else hipLaunchKernelGGL(k_bwd_layer_w<false>, dim3(p.nbl, 1, 1), dim3(TPB), 0, s, k, w, l);
    // (dW_{l-1} = d h_l a_{l-1}^T: accumulated by the layer kernel itself, one record per workgroup in the layer's own area; the cell's last
    // launch sums them)
    if (!pmerge || l == 1) reduce_jobs(s, sums + (size_t)pass * SREC, p.nbl, nullptr, 0, 0, 0, nullptr);
  }
  if (side_err) return NIS_E_HIP;
  if (sd && hipStreamWaitEvent(s, sd->join, 0) != hipSuccess) return NIS_E_HIP;
  k.sums = sums + (size_t)L * SREC;
  {
    const int gi = (int)std::max<long>(1, std::min<long>((N + TPB - 1) / TPB, std::max<long>(1, (long)device_cus() * 8 / G)));
    ScatterW sa;
    sa.sums = sums;
    sa.glast = glast;
    sa.bn = k.bn;
    sa.gparams = gparams;
    sa.hidden_bias = c.kind == NIS_AFFINE ? (k.invN == 0.0 ? 2 : 1) : 0;
    sa.G = G;
    sa.nblk = 32;
    sa.sums_gs = (long)nsum;
    sa.gl_gs = (long)ngem;
    sa.parth = parth;
    sa.parth_gs = (long)((size_t)L * p.parth);
    sa.layer_stride = (long)p.parth;
    sa.nbl = p.nbl;
    sa.nquads = (WM * (WM + 1) + 3) / 4;
    sa.nred = sa.nquads * L;
    hipLaunchKernelGGL(k_bwd_input_w, dim3(gi + sa.nblk + sa.nred, 1, Gz), dim3(TPB), 0, s, k, sa);
  }
  return hipGetLastError() == hipSuccess ? NIS_OK : NIS_E_HIP;
}
