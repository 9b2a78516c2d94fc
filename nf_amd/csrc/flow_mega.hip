// k_flow_mega: the whole body of BasicManager.integrate's loop (reference normalizing_flows/manager.py:395-400) for the
// README / C1 / C2 model -- PWQuadManager(n_flow=2).create_model(2, n_bins=4, NN=[w<=3]*3), camel integrand -- as ONE
// persistent launch per set of G batches:   draw w -> cell 0 -> cell 1 -> camel(X) * J -> moment records.
//
// Why: as separate launches the step streams the batch through HBM seven times (BN_0 pass, two statistics passes per
// cell, two apply passes: 88.5 B/sample against 8 algorithmic) and pays 8 launches per set.  Train-mode BatchNorm makes
// every BN layer a batch-wide reduction, so a batch cannot simply stay inside one wave; but one batch of 1e6 samples is
// only 8 MB and the chip has 40 MB of LDS.  Here a batch lives on chip from its draw to its moment record:
//
//   * `teams` (<= 3) teams of NT workgroups (NT = CUs, 256 threads each, 3 resident per CU by LDS) -- team t owns batches
//     t, t + teams, ...; member m of a team owns samples [m n_loc, (m+1) n_loc) of the team's current batch and keeps them
//     in LDS: 8 bytes per sample -- the float32 pair (x_pass, x_trans) while cell 0 works, cell 0's output y (float64, the
//     pass-through input of cell 1) afterwards.  Cell 0's Jacobian and pass-through input (needed again by cell 1's
//     bijector / the integrand) take one round trip through an HBM spill buffer (12 B written + 12 B read per sample,
//     streamed and prefetched): whole-step traffic 24 B/sample with in-kernel draws, 32 B/sample with fed w.
//   * every BN layer with batch statistics (BN_0, BN_2, BN_3 of each cell; BN_1 follows from BN_0 for P = 1) is one
//     team-wide hand-off: each member publishes ONE 128-byte record (count, mean[3], M2[3]) as sixteen 8-byte
//     {tag, value} granules (write-through agent-scope stores, one wave instruction), bumps an arrival counter; a member
//     waits on the counter with one lane, then sweeps the NT records with agent-scope loads whose tags are checked
//     (a granule that is not of this epoch is re-read), and merges them in a fixed order -> bitwise reproducible, no
//     float atomics, no grid-wide barrier.  While a team waits, the other teams' waves on the same CU keep its SIMDs
//     busy: the wave scheduler does the software pipelining.
//   * spins are bounded (s_memrealtime): a hand-off that does not complete within NIS_MEGA_TIMEOUT_TICKS sets status
//     bit 1 and the workgroup leaves; every other workgroup then times out as well and the grid drains.
//
// Arithmetic is the SPEC path of k_cell_apply / k_bn_stats_fold (flow_kernels.h): same exp table, same bin search with
// the reference's exact semantics, same raw-sum column evaluation; reference restated: coupling_cells.py:159-228, :236-254.
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#define NIS_EXP_BITS 6          // the phases below inline the 64-entry / degree-5 exp (LDS budget of the resident batch)
#include "flow_kernels.h"
#include "flow_mega.h"

#define NIS_MEGA_TIMEOUT_TICKS 300000000ull          // 3 s of the 100 MHz s_memrealtime counter
#ifndef NIS_MEGA_U
#define NIS_MEGA_U 1                                  // samples per thread and round of the apply phases (2 gained nothing: one wave alone issues fp64 at half rate whatever its ILP)
#endif

typedef unsigned long long mega_u64;
typedef float nis_f2 __attribute__((ext_vector_type(2)));

// ---------------------------------------------------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al., SC'11; constants of Random123).  Draw mapping (documented in include/nis_hip.h):
//   w[g][n][c] = (philox(ctr = {n>>1 lo, n>>1 hi, s lo, s hi}, key = {seed lo, seed hi})[2 (n & 1) + c] >> 8) * 2^-24,  s = stream0 + g
__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1,
                                              unsigned (&r)[4]) {
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    const mega_u64 p0 = (mega_u64)0xD2511F53u * c0, p1 = (mega_u64)0xCD9E8D57u * c2;
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  r[0] = c0; r[1] = c1; r[2] = c2; r[3] = c3;
}
__device__ __forceinline__ float u24(unsigned r) { return (float)(r >> 8) * 0x1p-24f; }

__global__ __launch_bounds__(256) void k_philox_rows(mega_u64 seed, mega_u64 stream0, long N, float* __restrict__ out) {
  const long npair = (N + 1) / 2;
  const mega_u64 s = stream0 + blockIdx.y;
  float* o = out + (size_t)blockIdx.y * N * 2;
  for (long c = (long)blockIdx.x * 256 + threadIdx.x; c < npair; c += (long)gridDim.x * 256) {
    unsigned r[4];
    philox4x32_10((unsigned)c, (unsigned)((mega_u64)c >> 32), (unsigned)s, (unsigned)(s >> 32), (unsigned)seed, (unsigned)(seed >> 32), r);
    o[4 * c + 0] = u24(r[0]);
    o[4 * c + 1] = u24(r[1]);
    if (2 * c + 1 < N) {
      o[4 * c + 2] = u24(r[2]);
      o[4 * c + 3] = u24(r[3]);
    }
  }
}

hipError_t nis_philox_rows_launch(unsigned long long seed, unsigned long long stream0, int G, long N, float* out, hipStream_t s) {
  const long npair = (N + 1) / 2;
  const int gx = (int)std::max<long>(1, std::min<long>(2048, (npair + 255) / 256));
  hipLaunchKernelGGL(k_philox_rows, dim3(gx, G), dim3(256), 0, s, (mega_u64)seed, (mega_u64)stream0, N, out);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------
// LDS carve (dynamic, base 16-byte aligned, no static __shared__ in this kernel)
struct MegaLds {
  double* st;        // NIS_MEGA_CAP: per-sample state (float32 pair during cell 0, float64 y afterwards)
  double* scr;       // 9 x 256: gather columns of pwquad_column  |  NT x 8 swept records
  double* coef;      // 24: A[3] B[3] | s2[3] h2[3] | s3[3] h3[3]   (each block padded to 4)
  double* etab;      // 64
  double* wmom;      // 4 waves x 3 features x 3
  double* fmom;      // 4 waves x 4
  double* recbuf;    // 8: n, mean[3], M2[3], pad
  double* mstat;     // 3 features x 4: n, mean, M2, -
  int* flag;         // 4
};
#define NIS_MEGA_SCR_DOUBLES (NIS_MEGA_GATHER ? 9 * NIS_MEGA_TPB : NIS_MEGA_SWEEP_REC * 8)
#define NIS_MEGA_LDS_BYTES ((NIS_MEGA_CAP + NIS_MEGA_SCR_DOUBLES + 24 + 64 + 36 + 16 + 8 + 12) * 8 + 16)

__device__ __forceinline__ MegaLds mega_carve(unsigned char* smem) {
  MegaLds L;
  double* p = reinterpret_cast<double*>(smem);
  L.st = p;       p += NIS_MEGA_CAP;
  L.scr = p;      p += NIS_MEGA_SCR_DOUBLES;
  L.coef = p;     p += 24;
  L.etab = p;     p += 64;
  L.wmom = p;     p += 36;
  L.fmom = p;     p += 16;
  L.recbuf = p;   p += 8;
  L.mstat = p;    p += 12;
  L.flag = reinterpret_cast<int*>(p);
  return L;
}

// ---------------------------------------------------------------------------------------------------------------------
// hand-off primitives
// A pointer the compiler must re-derive at this point: without it every slot's granule / counter address (and a mask per
// unrolled load) is hoisted out of the batch loop and kept alive -- hundreds of values that spill to scratch and are
// reloaded on the hand-offs' critical path.
template <typename T>
__device__ __forceinline__ T* mega_opaque(T* p) {
  unsigned long long v = (unsigned long long)p;
  asm volatile("" : "+s"(v));
  return (T*)v;
}
__device__ __forceinline__ mega_u64 mega_ld(const mega_u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Workgroup record (recbuf, 8 doubles) -> 16 granules of this member's slot line, then one arrival.
__device__ __forceinline__ void mega_publish(const MegaArgs& a, const MegaLds& L, int team, int m, int slot, unsigned tag) {
  __syncthreads();                                                  // recbuf complete; every thread is done with the phase
  if (threadIdx.x < 64) {
    if (threadIdx.x < 16) {
      const unsigned half = reinterpret_cast<const unsigned*>(L.recbuf)[threadIdx.x];
      mega_u64* dst = mega_opaque(a.rec) + (((size_t)team * NIS_MEGA_SLOTS + slot) * a.NT + m) * 16 + threadIdx.x;
      __hip_atomic_store(dst, ((mega_u64)tag << 32) | half, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // no drain between the granules and the arrivals: the counter only says "worth sweeping now"; every granule a consumer
    // accepts carries this epoch's tag, one that does not is re-read -- correctness never rests on store ordering
    // one arrival on EVERY replica of the slot's counter (one wave instruction, 32 lanes, 32 lines): a waiter polls only
    // replica (m & 31), so a line sees NT adds and NT * teams / 32 pollers instead of NT * teams pollers on one hot line
    if (threadIdx.x < NIS_MEGA_REPL)
      __hip_atomic_fetch_add(mega_opaque(a.ctr) + ((size_t)(team * 8 + slot) * NIS_MEGA_REPL + threadIdx.x) * 32, 1u, __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
  }
}

// Wait for the team's `target` arrivals on `slot`, sweep the NT records into L.scr (record b at scr[8 b ..]) and merge
// feature j < F (wave j) in a fixed order: mstat[4 j] = {n, mean, M2}.  false: timed out (uniform over the workgroup).
template <int F>
__device__ __forceinline__ bool mega_collect(const MegaArgs& a, const MegaLds& L, int team, int m, int slot, unsigned tag, unsigned target) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) {
    const mega_u64 t0 = __builtin_amdgcn_s_memrealtime();
    int ok = 1;
    const unsigned* mine = mega_opaque(a.ctr) + ((size_t)(team * 8 + slot) * NIS_MEGA_REPL + (m & (NIS_MEGA_REPL - 1))) * 32;
    while (__hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(4);
      if (__builtin_amdgcn_s_memrealtime() - t0 > NIS_MEGA_TIMEOUT_TICKS) {
        ok = 0;
        break;
      }
    }
    L.flag[0] = ok;
    L.flag[1] = 1;
  }
  __syncthreads();
  if (!L.flag[0]) return false;
  const mega_u64* src = mega_opaque(a.rec) + ((size_t)team * NIS_MEGA_SLOTS + slot) * a.NT * 16;
  unsigned* dst = reinterpret_cast<unsigned*>(L.scr);
  const int ne = a.NT * 16;                                          // <= 4096: sixteen granules per thread, all requested at once
  bool good = true;
  mega_u64 x[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    const int e = u * NIS_MEGA_TPB + tid;
    x[u] = e < ne ? mega_ld(src + e) : ((mega_u64)tag << 32);
  }
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    const int e = u * NIS_MEGA_TPB + tid;
    if (e < ne && (unsigned)(x[u] >> 32) != tag) {                    // not this epoch's granule yet: re-read (bounded)
      const mega_u64 t0 = __builtin_amdgcn_s_memrealtime();
      do {
        __builtin_amdgcn_s_sleep(1);
        x[u] = mega_ld(src + e);
      } while ((unsigned)(x[u] >> 32) != tag && __builtin_amdgcn_s_memrealtime() - t0 <= NIS_MEGA_TIMEOUT_TICKS);
      good = good && (unsigned)(x[u] >> 32) == tag;
    }
  }
  if (!good) L.flag[1] = 0;
  // records pass through the LDS scratch in chunks of NIS_MEGA_SWEEP_REC (granule e = 16 b + q of record b sits in
  // x[e / 256] of thread e % 256); feature j < F is merged by wave j, lane s taking records s, s + 64, ... of each chunk
  constexpr int UC = NIS_MEGA_SWEEP_REC / 16;                        // x[] entries per chunk
  double an = 0.0, a1 = 0.0, a2 = 0.0, ref = 0.0;
  auto chunk = [&](auto cc) {
    constexpr int c = decltype(cc)::value;
    if (c > 0) __syncthreads();                                      // the previous chunk has been consumed
#pragma unroll
    for (int u = 0; u < UC; ++u) {
      const int e = (c * UC + u) * NIS_MEGA_TPB + tid;
      if (e < ne) dst[e - c * NIS_MEGA_SWEEP_REC * 16] = (unsigned)x[c * UC + u];
    }
    __syncthreads();
    if (wave < F && L.flag[1]) {
      const int j = wave;
      if (c == 0) ref = L.scr[1 + j];                                 // mean of record 0 (member 0 always owns samples)
      for (int b = lane; b < NIS_MEGA_SWEEP_REC && c * NIS_MEGA_SWEEP_REC + b < a.NT; b += 64) {
        const double* q = L.scr + b * 8;
        const double n = q[0], dl = q[1 + j] - ref;
        an += n;
        a1 = fma(n, dl, a1);
        a2 += fma(n * dl, dl, q[4 + j]);
      }
    }
  };
  chunk(std::integral_constant<int, 0>());
  if (16 / UC > 1 && NIS_MEGA_SWEEP_REC < a.NT) chunk(std::integral_constant<int, (16 / UC > 1 ? 1 : 0)>());
  static_assert(16 / UC <= 2, "at most two sweep chunks");
  if (!L.flag[1]) return false;
  if (wave < F) {
    const int j = wave;
    an = wave_sum(an);
    a1 = wave_sum(a1);
    a2 = wave_sum(a2);
    if (lane == 0) {
      L.mstat[4 * j + 0] = an;
      L.mstat[4 * j + 1] = ref + a1 / an;
      L.mstat[4 * j + 2] = fmax(a2 - a1 * a1 / an, 0.0);
    }
  }
  __syncthreads();
  return true;
}

// Per-wave shifted sums -> workgroup record in recbuf: [n, mean_f.., M2_f..] (Chan merge of the four waves, fixed order).
template <int F>
__device__ __forceinline__ void mega_record(const MegaLds& L, const double (&S1)[F], const double (&S2)[F], double cnt, const double (&shf)[F]) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const double nw = wave_sum(cnt);
#pragma unroll
  for (int f = 0; f < F; ++f) {
    const double s1 = wave_sum(S1[f]), s2 = wave_sum(S2[f]);
    if (lane == 0) {
      const bool any = nw > 0.0;
      L.wmom[(wave * 3 + f) * 3 + 0] = nw;
      L.wmom[(wave * 3 + f) * 3 + 1] = any ? shf[f] + s1 / nw : 0.0;
      L.wmom[(wave * 3 + f) * 3 + 2] = any ? fmax(s2 - s1 * s1 / nw, 0.0) : 0.0;
    }
  }
  __syncthreads();
  if (tid < 3) {
    Mom t = {0.0, 0.0, 0.0};
    if (tid < F) {
      for (int w = 0; w < 4; ++w) {
        Mom q = {L.wmom[(w * 3 + tid) * 3 + 0], L.wmom[(w * 3 + tid) * 3 + 1], L.wmom[(w * 3 + tid) * 3 + 2]};
        t = mom_merge(t, q);
      }
    }
    if (tid == 0) {
      L.recbuf[0] = t.n;
      L.recbuf[7] = 0.0;
    }
    L.recbuf[1 + tid] = t.mean;
    L.recbuf[4 + tid] = t.m2;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// BatchNorm coefficients from the merged statistics (what load_coefs / fold_coefs of flow_kernels.h compute).
// Layer 0 (one feature) + the derived layer 1 + the fold: coef[0..2] = A_j, coef[4..6] = B_j  with
//   BN_1(Linear_0(BN_0(x)))_j = A_j x + B_j.   `writer`: team member 0 records the batch statistics for the running update.
__device__ __forceinline__ void mega_coef_first(const MegaCell& c, const MegaLds& L, int G, int g, bool writer) {
  const int j = threadIdx.x;
  if (j < 3) {
    const double n = L.mstat[0], mean0 = L.mstat[1], var0 = L.mstat[2] / n;
    const double scale0 = c.gb[0] / sqrt(var0 + NIS_BN_EPS);
    const double shift0 = c.gb[4] - mean0 * scale0;
    const double wj = c.wp[(size_t)j * 4];                                        // Linear_0.weight[j][0]
    const double mean1 = wj * fma(scale0, mean0, shift0);
    const double as = wj * scale0;
    const double var1 = as * as * var0;
    const double scale1 = c.gb[(1 * 2 + 0) * 4 + j] / sqrt(var1 + NIS_BN_EPS);
    const double shift1 = c.gb[(1 * 2 + 1) * 4 + j] - mean1 * scale1;
    L.coef[j] = scale1 * (wj * scale0);
    L.coef[4 + j] = fma(scale1, wj * shift0, shift1);
    if (writer) {
      if (j == 0) {
        c.gstat[(((size_t)0 * G + g) * 2 + 0) * 4] = mean0;
        c.gstat[(((size_t)0 * G + g) * 2 + 1) * 4] = n > 1.0 ? L.mstat[2] / (n - 1.0) : var0;
      }
      c.gstat[(((size_t)1 * G + g) * 2 + 0) * 4 + j] = mean1;
      c.gstat[(((size_t)1 * G + g) * 2 + 1) * 4 + j] = n > 1.0 ? var1 * (n / (n - 1.0)) : var1;
    }
  }
  __syncthreads();
}
// Layer l = 2 or 3 (three features): coef[8 (l-1) + j] = scale, coef[8 (l-1) + 4 + j] = shift
__device__ __forceinline__ void mega_coef_layer(const MegaCell& c, const MegaLds& L, int l, int G, int g, bool writer) {
  const int j = threadIdx.x;
  if (j < 3) {
    const double n = L.mstat[4 * j], mean = L.mstat[4 * j + 1], m2 = L.mstat[4 * j + 2];
    const double var_b = m2 / n;
    const double scale = c.gb[(l * 2 + 0) * 4 + j] / sqrt(var_b + NIS_BN_EPS);
    L.coef[8 * (l - 1) + j] = scale;
    L.coef[8 * (l - 1) + 4 + j] = c.gb[(l * 2 + 1) * 4 + j] - mean * scale;
    if (writer) {
      c.gstat[(((size_t)l * G + g) * 2 + 0) * 4 + j] = mean;
      c.gstat[(((size_t)l * G + g) * 2 + 1) * 4 + j] = n > 1.0 ? m2 / (n - 1.0) : var_b;
    }
  }
  __syncthreads();
}

// ---------------------------------------------------------------------------------------------------------------------
// Statistics of BN layer LAYER (2 or 3) over this member's samples: k_bn_stats_fold's loop on the LDS-resident column.
template <int LAYER, bool F32IN>
__device__ __forceinline__ void mega_stats(const MegaCell& c, const MegaLds& L, int n_own) {
  const int tid = threadIdx.x;
  nis_cptr wp = as_const(c.wp);
  double cs[LAYER][3], ch[LAYER][3], W[LAYER][3][3];               // rows 1 .. LAYER-1; row 1 of cs/ch = (A, B)
#pragma unroll
  for (int l = 1; l < LAYER; ++l) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      cs[l][j] = L.coef[8 * (l - 1) + j];
      ch[l][j] = L.coef[8 * (l - 1) + 4 + j];
#pragma unroll
      for (int i = 0; i < 3; ++i) W[l][j][i] = wp[(size_t)l * 16 + j * 4 + i];
    }
  }
  auto stack = [&](double x, const double (&nshf)[3], double (&t)[3]) {
    double av[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) av[j] = fmax(fma(x, cs[1][j], ch[1][j]), 0.0);
#pragma unroll
    for (int l = 1; l < LAYER; ++l) {
      double h[3];
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        double acc = (l == LAYER - 1) ? nshf[j] : 0.0;
#pragma unroll
        for (int i = 0; i < 3; ++i) acc = fma(W[l][j][i], av[i], acc);
        h[j] = acc;
      }
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        if (l == LAYER - 1) t[j] = h[j];
        else av[j] = fmax(fma(h[j], cs[l + 1][j], ch[l + 1][j]), 0.0);
      }
    }
  };
  auto sample = [&](int i) -> double {
    if (F32IN) return (double)reinterpret_cast<const nis_f2*>(L.st)[i].x;
    return L.st[i];
  };
  // wave-uniform shift: the value of the wave's first sample (lane 0, first round)
  double nshf[3], shf[3];
  {
    const double xf = wave_first(sample(tid));
    double zero[3] = {0.0, 0.0, 0.0}, t0[3];
    stack(xf, zero, t0);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      shf[j] = t0[j];
      nshf[j] = -t0[j];
    }
  }
  double S1[3] = {0.0, 0.0, 0.0}, S2[3] = {0.0, 0.0, 0.0};
  double cnt = 0.0;
  const int n_full = n_own >> 8;
  int it = 0;
  for (; it + 4 <= n_full; it += 4) {
    double x[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) x[u] = sample((it + u) * NIS_MEGA_TPB + tid);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      double t[3];
      stack(x[u], nshf, t);
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        S1[j] += t[j];
        S2[j] = fma(t[j], t[j], S2[j]);
      }
    }
    cnt += 4.0;
  }
  for (; it * NIS_MEGA_TPB < n_own; ++it) {
    const int i = it * NIS_MEGA_TPB + tid;
    const bool valid = i < n_own;
    double t[3];
    stack(valid ? sample(i) : 0.0, nshf, t);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const double tv = valid ? t[j] : 0.0;
      S1[j] += tv;
      S2[j] = fma(tv, tv, S2[j]);
    }
    cnt += valid ? 1.0 : 0.0;
  }
  mega_record<3>(L, S1, S2, cnt, shf);
}

// ---------------------------------------------------------------------------------------------------------------------
// Conditioner with finalised BatchNorm + the K = 4 bijector of one sample (k_cell_apply, SPEC path with relative logits).
struct MegaHead {
  double wd[7][3];      // rows of the last layer relative to the last height / the last width (scalar registers)
  double bd[7];
};
__device__ __forceinline__ void mega_head_init(const MegaCell& c, MegaHead& H) {
  nis_cptr wl = as_const(c.wlast);
  nis_cptr bias = as_const(c.bias);
#pragma unroll
  for (int q = 0; q < 7; ++q) {
    const int r = q < 4 ? q : q + 1;                     // heights 0..3 relative to row 4, widths 5..7 relative to row 8
    const int ref = q < 4 ? 4 : 8;
    H.bd[q] = bias[r] - bias[ref];
#pragma unroll
    for (int i = 0; i < 3; ++i) H.wd[q][i] = nis_uniform(wl[r * 4 + i] - wl[ref * 4 + i]);
  }
}
// U samples per call: the conditioners and exps of the U samples are independent instruction streams (one wave per SIMD and
// team: a single dependent fp64 chain leaves the pipe idle ~60 % of the time); the column evaluations then take turns on
// the thread's one gather column.
template <int U>
__device__ __forceinline__ void mega_cell(const MegaCell& c, const MegaLds& L, const MegaHead& H, const double (&x)[U], const double (&xb)[U],
                                          double (&y)[U], double (&jf)[U]) {
  nis_cptr wp = as_const(c.wp);
  double V[U][5], W[U][4];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    double av[3], h[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) av[j] = fmax(fma(x[u], L.coef[j], L.coef[4 + j]), 0.0);
#pragma unroll
    for (int l = 1; l < 3; ++l) {
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        double acc = 0.0;
#pragma unroll
        for (int i = 0; i < 3; ++i) acc = fma(wp[(size_t)l * 16 + j * 4 + i], av[i], acc);
        h[j] = acc;
      }
#pragma unroll
      for (int j = 0; j < 3; ++j) av[j] = fmax(fma(h[j], L.coef[8 * l + j], L.coef[8 * l + 4 + j]), 0.0);
    }
    // the 7 exps as 7 interleaved instruction streams (stage by stage): with one wave per SIMD and team a single exp is a
    // ~14-deep dependent fp64 chain, i.e. mostly pipeline latency
    double z[7], kd[7], kk[7], rr[7], pp[7], T[7];
    int ki[7];
#pragma unroll
    for (int q = 0; q < 7; ++q) {
      double acc = H.bd[q];
#pragma unroll
      for (int i = 0; i < 3; ++i) acc = fma(H.wd[q][i], av[i], acc);
      z[q] = acc;
    }
#pragma unroll
    for (int q = 0; q < 7; ++q) kd[q] = fma(z[q], 0x1.71547652b82fep+6, 0x1.8p+52);
#pragma unroll
    for (int q = 0; q < 7; ++q) {
      kk[q] = kd[q] - 0x1.8p+52;
      ki[q] = __double2loint(kd[q]);
    }
#pragma unroll
    for (int q = 0; q < 7; ++q) T[q] = L.etab[ki[q] & 63];
#pragma unroll
    for (int q = 0; q < 7; ++q) rr[q] = fma(kk[q], -0x1.62e42fef00000p-7, z[q]);
#pragma unroll
    for (int q = 0; q < 7; ++q) rr[q] = fma(kk[q], -0x1.473de6af278edp-40, rr[q]);
#pragma unroll
    for (int q = 0; q < 7; ++q) pp[q] = fma(rr[q], 1.0 / 120.0, 1.0 / 24.0);
#pragma unroll
    for (int q = 0; q < 7; ++q) pp[q] = fma(rr[q], pp[q], 1.0 / 6.0);
#pragma unroll
    for (int q = 0; q < 7; ++q) pp[q] = fma(rr[q], pp[q], 0.5);
#pragma unroll
    for (int q = 0; q < 7; ++q) pp[q] = fma(rr[q], pp[q], 1.0);
#pragma unroll
    for (int q = 0; q < 7; ++q) pp[q] = rr[q] * pp[q];
#pragma unroll
    for (int q = 0; q < 7; ++q) {
      const double e = ldexp(fma(T[q], pp[q], T[q]), ki[q] >> 6);                    // = nis_exp(z[q], etab), same operation order
      if (q < 4) V[u][q] = e;
      else W[u][q - 4] = e;
    }
    V[u][4] = 1.0;
    W[u][3] = 1.0;
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    int bin;
    y[u] = pwquad_column<4, true>(xb[u], 4, V[u], W[u], &jf[u], &bin, NIS_MEGA_GATHER ? L.scr + threadIdx.x : nullptr);
  }
}

// Diagnostic build only (-DNIS_MEGA_STAMPS, never in the shipped library): thread 0 of every workgroup stamps the 100 MHz
// clock at each phase boundary of its first NIS_MEGA_STAMP_IL batches into a buffer of its own (tools/mega_stamps.py).
#ifdef NIS_MEGA_STAMPS
#define NIS_MEGA_STAMP_IL 4
__device__ unsigned long long g_mega_stamps[NIS_MEGA_MAX_TEAMS * NIS_MEGA_MAX_NT * NIS_MEGA_STAMP_IL * 16];
#define MEGA_STAMP(k)                                                                                           \
  do {                                                                                                          \
    if (tid == 0 && il < NIS_MEGA_STAMP_IL) g_mega_stamps[((size_t)blockIdx.x * NIS_MEGA_STAMP_IL + il) * 16 + (k)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
extern "C" int nis_mega_stamps(unsigned long long* out_host, int n) {
  return (int)hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_mega_stamps), sizeof(unsigned long long) * (size_t)n);
}
#else
#define MEGA_STAMP(k) do { } while (0)
#endif

// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NIS_MEGA_TPB, NIS_MEGA_TEAMS) void k_flow_mega(MegaArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char mega_smem[];
  const MegaLds L = mega_carve(mega_smem);
  const int tid = threadIdx.x, lane = tid & 63;
  const int team = blockIdx.x / a.NT, m = blockIdx.x - team * a.NT;
  const long base_n = (long)m * a.n_loc;
  long rest = a.N - base_n;
  rest = rest < 0 ? 0 : rest;
  const int n_own = (int)(rest < a.n_loc ? rest : a.n_loc);
  const int n_it = (n_own + NIS_MEGA_TPB - 1) / NIS_MEGA_TPB;
  if (tid < 64) L.etab[tid] = NIS_EXP_TAB[tid];
  nis_f2* const st2 = reinterpret_cast<nis_f2*>(L.st);
  double* const spj = a.spill_j + ((size_t)team * a.NT + m) * NIS_MEGA_CAP;
  float* const spx = a.spill_x + ((size_t)team * a.NT + m) * NIS_MEGA_CAP;
  const bool writer = m == 0;
  if (blockIdx.x == 0 && tid == 0) *a.status = 0;
  __syncthreads();

  int il = 0;
  for (int g = team; g < a.G; g += a.teams, ++il) {
    const unsigned tag0 = ((a.launch_seq & 0xFFFFFu) << 12) | (unsigned)(il * 8 + 1);
    const unsigned target = (unsigned)a.NT * (unsigned)(il + 1);
    // ---- phase 0: the draws (or the caller's rows) -> LDS; BN_0 sums of cell 0's pass-through column
    MEGA_STAMP(0);
    {
      double S1[1] = {0.0}, S2[1] = {0.0}, shf[1] = {0.0};
      double cnt = 0.0;
      if (a.w) {
        const nis_f2* rows = reinterpret_cast<const nis_f2*>(a.w) + (size_t)g * a.N + base_n;
        for (int it0 = 0; it0 < n_it; it0 += 8) {
          nis_f2 r[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int i = (it0 + u) * NIS_MEGA_TPB + tid;
            r[u] = i < n_own ? rows[i] : nis_f2{0.f, 0.f};
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int i = (it0 + u) * NIS_MEGA_TPB + tid;
            const bool valid = i < n_own;
            const nis_f2 c2 = a.col0 ? nis_f2{r[u].y, r[u].x} : r[u];            // (pass-through, transformed)
            const double x0 = (double)c2.x;
            if (it0 + u == 0) shf[0] = wave_first(x0);
            if (valid) st2[i] = c2;
            const double t = valid ? x0 - shf[0] : 0.0;
            S1[0] += t;
            S2[0] = fma(t, t, S2[0]);
            cnt += valid ? 1.0 : 0.0;
          }
        }
      } else {
        // Lane pairs share one Philox call (4 words = two samples): even lanes evaluate the call of round k, odd lanes
        // the call of round k + 1, and each hands its partner the half it needs (n_loc is even: a pair never straddles members).
        const mega_u64 s = (a.rng_state ? a.rng_state[1] : a.stream0) + (mega_u64)g;
        const mega_u64 seed = a.rng_state ? a.rng_state[0] : a.seed;
        const bool odd = (lane & 1) != 0;
        for (int k = 0; k < n_it; k += 2) {
          const int itc = k + (odd ? 1 : 0);
          const mega_u64 c = (mega_u64)(base_n + (long)itc * NIS_MEGA_TPB + (tid & ~1)) >> 1;
          unsigned r[4];
          philox4x32_10((unsigned)c, (unsigned)(c >> 32), (unsigned)s, (unsigned)(s >> 32), (unsigned)seed, (unsigned)(seed >> 32), r);
          const unsigned s0 = odd ? r[0] : r[2], s1 = odd ? r[1] : r[3];
          const unsigned q0 = (unsigned)__shfl_xor((int)s0, 1, 64), q1 = (unsigned)__shfl_xor((int)s1, 1, 64);
          const unsigned ua[2] = {odd ? q0 : r[0], odd ? r[2] : q0}, ub[2] = {odd ? q1 : r[1], odd ? r[3] : q1};
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const int i = (k + u) * NIS_MEGA_TPB + tid;
            const bool valid = i < n_own;
            const float w0 = u24(ua[u]), w1 = u24(ub[u]);
            const nis_f2 c2 = a.col0 ? nis_f2{w1, w0} : nis_f2{w0, w1};
            const double x0 = (double)c2.x;
            if (k + u == 0) shf[0] = wave_first(x0);
            if (valid) {
              st2[i] = c2;
              if (a.w_out) reinterpret_cast<nis_f2*>(a.w_out)[(size_t)g * a.N + base_n + i] = nis_f2{w0, w1};
            }
            const double t = valid ? x0 - shf[0] : 0.0;
            S1[0] += t;
            S2[0] = fma(t, t, S2[0]);
            cnt += valid ? 1.0 : 0.0;
          }
        }
      }
      mega_record<1>(L, S1, S2, cnt, shf);
    }
    MEGA_STAMP(1);
    mega_publish(a, L, team, m, 0, tag0 + 0);
    // ---- cell 0
    if (!mega_collect<1>(a, L, team, m, 0, tag0 + 0, target)) goto fail;
    mega_coef_first(a.cell[0], L, a.G, g, writer);
    MEGA_STAMP(2);
    mega_stats<2, true>(a.cell[0], L, n_own);
    MEGA_STAMP(3);
    mega_publish(a, L, team, m, 1, tag0 + 1);
    if (!mega_collect<3>(a, L, team, m, 1, tag0 + 1, target)) goto fail;
    mega_coef_layer(a.cell[0], L, 2, a.G, g, writer);
    MEGA_STAMP(4);
    mega_stats<3, true>(a.cell[0], L, n_own);
    MEGA_STAMP(5);
    mega_publish(a, L, team, m, 2, tag0 + 2);
    if (!mega_collect<3>(a, L, team, m, 2, tag0 + 2, target)) goto fail;
    mega_coef_layer(a.cell[0], L, 3, a.G, g, writer);
    MEGA_STAMP(6);
    {
      // apply: (x, xb) -> y (stays in LDS as cell 1's pass-through input); J and x take the HBM round trip
      double S1[1] = {0.0}, S2[1] = {0.0}, shf[1] = {0.0};
      double cnt = 0.0;
      MegaHead H0;
      mega_head_init(a.cell[0], H0);
      for (int it = 0; it < n_it; it += NIS_MEGA_U) {
        nis_f2 r[NIS_MEGA_U];
        double xv[NIS_MEGA_U], xbv[NIS_MEGA_U], y[NIS_MEGA_U], jf[NIS_MEGA_U];
#pragma unroll
        for (int u = 0; u < NIS_MEGA_U; ++u) {
          const int i = (it + u) * NIS_MEGA_TPB + tid;
          r[u] = i < n_own ? st2[i] : nis_f2{0.f, 0.f};
          xv[u] = (double)r[u].x;
          xbv[u] = (double)r[u].y;
        }
        mega_cell<NIS_MEGA_U>(a.cell[0], L, H0, xv, xbv, y, jf);
        if (it == 0) shf[0] = wave_first(y[0]);
#pragma unroll
        for (int u = 0; u < NIS_MEGA_U; ++u) {
          const int i = (it + u) * NIS_MEGA_TPB + tid;
          const bool valid = i < n_own;
          if (valid) {
            L.st[i] = y[u];
            spj[i] = jf[u];
            spx[i] = r[u].x;
          }
          const double t = valid ? y[u] - shf[0] : 0.0;
          S1[0] += t;
          S2[0] = fma(t, t, S2[0]);
          cnt += valid ? 1.0 : 0.0;
        }
      }
      mega_record<1>(L, S1, S2, cnt, shf);
    }
    MEGA_STAMP(7);
    mega_publish(a, L, team, m, 3, tag0 + 3);
    // ---- cell 1 (passes y through, transforms cell 0's pass-through column)
    if (!mega_collect<1>(a, L, team, m, 3, tag0 + 3, target)) goto fail;
    mega_coef_first(a.cell[1], L, a.G, g, writer);
    MEGA_STAMP(8);
    mega_stats<2, false>(a.cell[1], L, n_own);
    MEGA_STAMP(9);
    mega_publish(a, L, team, m, 4, tag0 + 4);
    if (!mega_collect<3>(a, L, team, m, 4, tag0 + 4, target)) goto fail;
    mega_coef_layer(a.cell[1], L, 2, a.G, g, writer);
    MEGA_STAMP(10);
    mega_stats<3, false>(a.cell[1], L, n_own);
    MEGA_STAMP(11);
    mega_publish(a, L, team, m, 5, tag0 + 5);
    if (!mega_collect<3>(a, L, team, m, 5, tag0 + 5, target)) goto fail;
    mega_coef_layer(a.cell[1], L, 3, a.G, g, writer);
    MEGA_STAMP(12);
    {
      double fS1 = 0.0, fS2 = 0.0, cnt = 0.0, fmx = -__builtin_huge_val(), fshift = 0.0;
      const double p0 = a.fp[0], p1 = a.fp[1], is2 = a.fp[3];
      MegaHead H1;
      mega_head_init(a.cell[1], H1);
      // the spilled values of the NEXT round are requested before this round is computed (unconditional loads of clamped indices)
      double jq[NIS_MEGA_U];
      float xq[NIS_MEGA_U];
#pragma unroll
      for (int u = 0; u < NIS_MEGA_U; ++u) {
        const int i = u * NIS_MEGA_TPB + tid;
        const int ic = i < n_own ? i : 0;
        jq[u] = spj[ic];
        xq[u] = spx[ic];
      }
      for (int it = 0; it < n_it; it += NIS_MEGA_U) {
        double xv[NIS_MEGA_U], xbv[NIS_MEGA_U], jin[NIS_MEGA_U], y[NIS_MEGA_U], jf[NIS_MEGA_U];
#pragma unroll
        for (int u = 0; u < NIS_MEGA_U; ++u) {
          const int i = (it + u) * NIS_MEGA_TPB + tid;
          const bool valid = i < n_own;
          jin[u] = jq[u];
          xbv[u] = valid ? (double)xq[u] : 0.0;
          xv[u] = valid ? L.st[i] : 0.0;
          const int in = i + NIS_MEGA_U * NIS_MEGA_TPB;
          const int ic = in < n_own ? in : (valid ? i : 0);
          jq[u] = spj[ic];
          xq[u] = spx[ic];
        }
        mega_cell<NIS_MEGA_U>(a.cell[1], L, H1, xv, xbv, y, jf);
#pragma unroll
        for (int u = 0; u < NIS_MEGA_U; ++u) {
          const int i = (it + u) * NIS_MEGA_TPB + tid;
          const bool valid = i < n_own;
          const double x = xv[u];
          const double jout = jin[u] * jf[u];
          const double e1 = -((x - p0) * (x - p0) + (y[u] - p0) * (y[u] - p0)) * is2;          // README.md:35-36
          const double e2 = -((x - p1) * (x - p1) + (y[u] - p1) * (y[u] - p1)) * is2;
          const double v = (nis_exp(e1, L.etab) + nis_exp(e2, L.etab)) * jout;
          if (it == 0 && u == 0) fshift = wave_first(v);
          const double t = valid ? v - fshift : 0.0;
          fS1 += t;
          fS2 = fma(t, t, fS2);
          fmx = valid ? fmax(fmx, v) : fmx;
          cnt += valid ? 1.0 : 0.0;
        }
      }
      __syncthreads();                                               // fmom is free (block_record has its own barrier after the stores)
      block_record<4>(fshift, fS1, fS2, cnt, fmx, reinterpret_cast<double(*)[4]>(L.fmom), a.fpart + ((size_t)g * a.maxb + m) * 4);
    }
    __syncthreads();
    MEGA_STAMP(13);
  }
  return;
fail:
  if (tid == 0) atomicOr(a.status, 2);
}

// ---------------------------------------------------------------------------------------------------------------------
// host side
static int mega_cus() {
  int dev = 0;
  hipDeviceProp_t pr;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&pr, dev) != hipSuccess) return 0;
  return pr.multiProcessorCount;
}

bool nis_mega_geometry(int* NT, int* teams) {
  static int nt = -1, tm = 0;
  if (nt < 0) {
    nt = mega_cus();
    tm = 0;
    if (nt > NIS_MEGA_MAX_NT) nt = NIS_MEGA_MAX_NT;                  // a team need not span every CU
    if (nt >= 1 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(k_flow_mega), hipFuncAttributeMaxDynamicSharedMemorySize, NIS_MEGA_LDS_BYTES) ==
            hipSuccess) {
      int occ = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_flow_mega, NIS_MEGA_TPB, NIS_MEGA_LDS_BYTES) == hipSuccess) tm = occ;
      // the occupancy API can over-report by one block per CU for SGPR-heavy kernels (MI355X_MICROARCH.md): this kernel is
      // LDS-limited to NIS_MEGA_TEAMS per CU anyway, and a non-resident workgroup only makes the bounded hand-offs time out
      if (tm > NIS_MEGA_MAX_TEAMS) tm = NIS_MEGA_MAX_TEAMS;
      const char* e = getenv("NIS_MEGA_TEAMS");
      if (e && *e && atoi(e) >= 1 && atoi(e) < tm) tm = atoi(e);
    }
    (void)hipGetLastError();
    if (getenv("NIS_DEBUG_OCC")) fprintf(stderr, "[nis] k_flow_mega: %d CUs, %d teams, %d B LDS\n", nt, tm, (int)NIS_MEGA_LDS_BYTES);
  }
  *NT = nt;
  *teams = tm;
  return tm >= 1;
}

static const size_t MEGA_CTR_BYTES = (size_t)NIS_MEGA_MAX_TEAMS * 8 * NIS_MEGA_REPL * 128;   // [teams][8 slots][replicas] counters, one 128-byte line each
static size_t mega_rec_bytes() { return (size_t)NIS_MEGA_MAX_TEAMS * NIS_MEGA_SLOTS * NIS_MEGA_MAX_NT * 16 * sizeof(mega_u64); }
static size_t mega_spill_elems() { return (size_t)NIS_MEGA_MAX_TEAMS * NIS_MEGA_MAX_NT * NIS_MEGA_CAP; }

size_t nis_mega_scratch_bytes() { return MEGA_CTR_BYTES + mega_rec_bytes() + mega_spill_elems() * (sizeof(double) + sizeof(float)); }

hipError_t nis_mega_launch(MegaArgs a, void* scratch, bool fresh, hipStream_t s) {
  static unsigned seq = 0;
  unsigned char* p = reinterpret_cast<unsigned char*>(scratch);
  a.ctr = reinterpret_cast<unsigned*>(p);
  a.rec = reinterpret_cast<mega_u64*>(p + MEGA_CTR_BYTES);
  a.spill_j = reinterpret_cast<double*>(p + MEGA_CTR_BYTES + mega_rec_bytes());
  a.spill_x = reinterpret_cast<float*>(a.spill_j + mega_spill_elems());
  a.launch_seq = ++seq;
  hipError_t e = hipMemsetAsync(p, 0, fresh ? MEGA_CTR_BYTES + mega_rec_bytes() : MEGA_CTR_BYTES, s);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_flow_mega, dim3(a.teams * a.NT), dim3(NIS_MEGA_TPB), NIS_MEGA_LDS_BYTES, s, a);
  return hipGetLastError();
}
