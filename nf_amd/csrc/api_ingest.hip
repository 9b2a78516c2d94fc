// Ingest / emit side of libnis_hip.so: row-major w -> column planes with the BN_0 record riders, the in-kernel Philox draws,
// planes -> row-major XJ, and the record reduction of the multi-rank exact BatchNorm (see api_internal.h for the split).
#include "api_internal.h"

// ---------------------------------------------------------------------------------------------
// ingest / emit.  256 samples per workgroup are transposed through LDS so that both the row-major
// side and the plane side move as contiguous 8-byte-per-lane streams.
template <typename TIN>
__global__ __launch_bounds__(NIS_TPB) void k_ingest(const TIN* __restrict__ w, long N, int d, double* __restrict__ planes, long ld,
                                                    IngestStats st) {
  extern __shared__ double tile[];                              // [sample][d] with +1 skew per 32 rows
  constexpr int NW = NIS_TPB / NIS_WAVE, NF = NIS_PMAX_FUSED;
  __shared__ double wshift[NW][NF];
  __shared__ double wmom[NW][NF][3];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = st.g0 + blockIdx.y;
  w += (size_t)g * N * d;
  planes += (size_t)g * st.plane_gs;
  if (blockIdx.x == 0 && g == 0 && threadIdx.x == 0 && st.status) *st.status = 0;
  double S1[NF], S2[NF];
#pragma unroll
  for (int f = 0; f < NF; ++f) S1[f] = S2[f] = 0.0;
  double cnt = 0.0;
  bool firstit = true;
  const long nblk = (N + NIS_TPB - 1) / NIS_TPB;
  for (long blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    const long i0 = blk * NIS_TPB;
    const int ns = (int)min((long)NIS_TPB, N - i0);
    const int tot = ns * d;
    for (int e = threadIdx.x; e < tot; e += NIS_TPB) {
      const int s = e / d, cidx = e - s * d;
      tile[s * d + cidx + s / 32] = (double)w[i0 * d + e];
    }
    __syncthreads();
    const int s = threadIdx.x;
    if (s < ns) {
      for (int cidx = 0; cidx < d; ++cidx) planes[(size_t)cidx * ld + i0 + s] = tile[s * d + cidx + s / 32];
      if (st.P > 0) {
        double v[NF];
#pragma unroll
        for (int f = 0; f < NF; ++f) v[f] = (f < st.P) ? tile[s * d + st.sel[f] + s / 32] : 0.0;
        if (firstit) {
#pragma unroll
          for (int f = 0; f < NF; ++f) {
            const double s0 = wave_first(v[f]);
            if (lane == 0) wshift[wave][f] = s0;
          }
          firstit = false;
        }
#pragma unroll
        for (int f = 0; f < NF; ++f) {
          const double t = v[f] - wshift[wave][f];
          S1[f] += t;
          S2[f] = fma(t, t, S2[f]);
        }
        cnt += 1.0;
      }
    }
    __syncthreads();
  }
  if (st.P > 0) emit_records<NF, NW>(S1, S2, cnt, wshift, wmom, st.P, st.partials + ((size_t)g * st.maxb + blockIdx.x) * st.wm * 3);
}

// d == 2 (README / C1 / C2): a row is one 8- or 16-byte vector, so every lane loads its own rows directly (coalesced),
// four in flight; no LDS transpose, no barriers.
// WRITE = false (direct row input: the cells read w themselves): only the BN_0 records are gathered.
template <typename TIN, bool WRITE = true>
__global__ __launch_bounds__(NIS_TPB) void k_ingest2(const TIN* __restrict__ w, long N, double* __restrict__ planes, long ld,
                                                     IngestStats st) {
  typedef TIN vec2 __attribute__((ext_vector_type(2)));
  constexpr int NW = NIS_TPB / NIS_WAVE, NF = 2, U = 4;
  __shared__ double wshift[NW][NF];
  __shared__ double wmom[NW][NF][3];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = st.g0 + blockIdx.y;
  const vec2* rows = reinterpret_cast<const vec2*>(w + (size_t)g * N * 2);
  planes += (size_t)g * st.plane_gs;
  if (blockIdx.x == 0 && g == 0 && threadIdx.x == 0 && st.status) *st.status = 0;
  double S1[NF] = {0.0, 0.0}, S2[NF] = {0.0, 0.0};
  double cnt = 0.0;
  bool firstit = true;
  const int s0 = st.P > 0 ? st.sel[0] : 0, s1 = st.P > 1 ? st.sel[1] : 0;
  const long stride = (long)gridDim.x * NIS_TPB;
  for (long i0 = (long)blockIdx.x * NIS_TPB + threadIdx.x; i0 < N; i0 += U * stride) {
    vec2 r[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long i = i0 + u * stride;
      r[u] = i < N ? rows[i] : vec2{0, 0};
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long i = i0 + u * stride;
      if (i < N) {
        const double x0 = (double)r[u].x, x1 = (double)r[u].y;
        if (WRITE) {
          planes[i] = x0;
          planes[(size_t)ld + i] = x1;
        }
        if (st.P > 0) {
          const double v0 = s0 == 0 ? x0 : x1, v1 = s1 == 0 ? x0 : x1;
          if (firstit) {
            const double f0 = wave_first(v0), f1 = wave_first(v1);
            if (lane == 0) wshift[wave][0] = f0, wshift[wave][1] = f1;
            firstit = false;
          }
          const double t0 = v0 - wshift[wave][0], t1 = v1 - wshift[wave][1];
          S1[0] += t0;
          S2[0] = fma(t0, t0, S2[0]);
          S1[1] += t1;
          S2[1] = fma(t1, t1, S2[1]);
          cnt += 1.0;
        }
      }
    }
  }
  if (st.P > 0) emit_records<NF, NW>(S1, S2, cnt, wshift, wmom, st.P, st.partials + ((size_t)g * st.maxb + blockIdx.x) * st.wm * 3);
}

// In-kernel draws for the direct-row chain: the record pass GENERATES batch g's float32 rows (Philox4x32-10, the mapping of
// nis_philox_rows: sample n takes words 2 (n & 1), 2 (n & 1) + 1 of the call with counter n >> 1 on stream
// stream0 + g) instead of reading them, writes them where the cells will read them, and gathers BN_0's records as k_ingest2
// does.  One thread = one Philox call = two consecutive samples = one 16-byte store.
__device__ __forceinline__ void nis_philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1, unsigned (&r)[4]) {
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  r[0] = c0; r[1] = c1; r[2] = c2; r[3] = c3;
}
__global__ __launch_bounds__(NIS_TPB) void k_ingest2_rng(unsigned long long seed, unsigned long long stream0,
                                                         const unsigned long long* __restrict__ state, long N, float* __restrict__ rows,
                                                         long rows_gs, long plane_ld, float* __restrict__ w_out, IngestStats st) {
  typedef float vec4 __attribute__((ext_vector_type(4)));
  constexpr int NW = NIS_TPB / NIS_WAVE, NF = 2;
  __shared__ double wshift[NW][NF];
  __shared__ double wmom[NW][NF][3];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = st.g0 + blockIdx.y;
  if (state) {                                     // device-resident generator state (graph replays): {seed, next stream}
    seed = state[0];
    stream0 = state[1];
  }
  const unsigned long long sid = stream0 + (unsigned long long)g;
  float* dst = rows + (size_t)g * rows_gs;
  if (blockIdx.x == 0 && g == 0 && threadIdx.x == 0 && st.status) *st.status = 0;
  double S1[NF] = {0.0, 0.0}, S2[NF] = {0.0, 0.0};
  double cnt = 0.0;
  bool firstit = true;
  const int s0 = st.P > 0 ? st.sel[0] : 0;
  const long npair = (N + 1) / 2;
  const long stride = (long)gridDim.x * NIS_TPB;
  for (long c = (long)blockIdx.x * NIS_TPB + threadIdx.x; c < npair; c += stride) {
    unsigned r[4];
    nis_philox4x32_10((unsigned)c, (unsigned)((unsigned long long)c >> 32), (unsigned)sid, (unsigned)(sid >> 32), (unsigned)seed,
                      (unsigned)(seed >> 32), r);
    const vec4 v = {(float)(r[0] >> 8) * 0x1p-24f, (float)(r[1] >> 8) * 0x1p-24f, (float)(r[2] >> 8) * 0x1p-24f, (float)(r[3] >> 8) * 0x1p-24f};
    const bool two = 2 * c + 1 < N;
    if (plane_ld > 0) {                       // two float32 column planes: samples 2c, 2c+1 of column 0 | of column 1
      typedef float vec2 __attribute__((ext_vector_type(2)));
      if (two) {
        *reinterpret_cast<vec2*>(dst + 2 * c) = vec2{v.x, v.z};
        *reinterpret_cast<vec2*>(dst + plane_ld + 2 * c) = vec2{v.y, v.w};
      } else {
        dst[2 * c] = v.x;
        dst[plane_ld + 2 * c] = v.y;
      }
    } else if (two) {
      *reinterpret_cast<vec4*>(dst + 4 * c) = v;
    } else {
      dst[4 * c] = v.x;
      dst[4 * c + 1] = v.y;
    }
    if (w_out) {
      if (two) *reinterpret_cast<vec4*>(w_out + (size_t)g * N * 2 + 4 * c) = v;
      else w_out[(size_t)g * N * 2 + 4 * c] = v.x, w_out[(size_t)g * N * 2 + 4 * c + 1] = v.y;
    }
    if (st.P > 0) {
      const double a0 = (double)(s0 == 0 ? v.x : v.y), a1 = (double)(s0 == 0 ? v.z : v.w);   // feature 0 of the pair's two samples
      if (firstit) {
        const double f0 = wave_first(a0);
        if (lane == 0) wshift[wave][0] = f0, wshift[wave][1] = 0.0;
        firstit = false;
      }
      const double t0 = a0 - wshift[wave][0], t1 = two ? a1 - wshift[wave][0] : 0.0;
      S1[0] += t0 + t1;
      S2[0] = fma(t0, t0, S2[0]);
      S2[0] = fma(t1, t1, S2[0]);
      cnt += two ? 2.0 : 1.0;
    }
  }
  if (st.P > 0) emit_records<NF, NW>(S1, S2, cnt, wshift, wmom, st.P, st.partials + ((size_t)g * st.maxb + blockIdx.x) * st.wm * 3);
}

// Multi-rank exact BatchNorm: merge this rank's workgroup records of one BN layer into ONE record per feature
// (count, mean, M2) -- shifted sums like load_coefs, fixed order -- written to out[j*3 ..]; the caller's exchange hook then
// gathers the ranks' records, and the consumers merge those instead of the local workgroup records.
__global__ __launch_bounds__(NIS_TPB) void k_sync_reduce(const double* __restrict__ part, int nparts, int WM, double* __restrict__ out) {
  __shared__ double red[3 * NIS_TPB];
  const int tid = threadIdx.x;
  const int S = NIS_TPB / WM;                 // slices per feature (WM <= 64 -> S >= 4, a power of two)
  const int j = tid % WM, sl = tid / WM;
  const double ref = part[j * 3 + 1];
  double an = 0.0, a1 = 0.0, a2 = 0.0;
  for (int b = sl; b < nparts; b += S) {
    const double* q = part + ((size_t)b * WM + j) * 3;
    const double n = q[0], dl = q[1] - ref;
    an += n;
    a1 = fma(n, dl, a1);
    a2 += fma(n * dl, dl, q[2]);
  }
  red[(sl * WM + j) * 3 + 0] = an;
  red[(sl * WM + j) * 3 + 1] = a1;
  red[(sl * WM + j) * 3 + 2] = a2;
  __syncthreads();
  for (int h = S / 2; h > 0; h >>= 1) {
    if (sl < h)
      for (int q = 0; q < 3; ++q) red[(sl * WM + j) * 3 + q] += red[((sl + h) * WM + j) * 3 + q];
    __syncthreads();
  }
  if (sl == 0) {
    const double n = red[j * 3 + 0], s1 = red[j * 3 + 1], s2 = red[j * 3 + 2];
    out[j * 3 + 0] = n;
    out[j * 3 + 1] = n > 0.0 ? ref + s1 / n : 0.0;
    out[j * 3 + 2] = n > 0.0 ? fmax(s2 - s1 * s1 / n, 0.0) : 0.0;
  }
}

// inverse of k_emit: column c of the row-major (N, cols) matrix goes to plane m.col[c]
__global__ __launch_bounds__(NIS_TPB) void k_scatter_rows(const double* __restrict__ src, long N, int cols, ColMap m,
                                                          double* __restrict__ planes, long ld) {
  extern __shared__ double tile[];
  const long nblk = (N + NIS_TPB - 1) / NIS_TPB;
  for (long blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    const long i0 = blk * NIS_TPB;
    const int ns = (int)min((long)NIS_TPB, N - i0);
    const int tot = ns * cols;
    for (int e = threadIdx.x; e < tot; e += NIS_TPB) {
      const int r = e / cols, cidx = e - r * cols;
      tile[r * cols + cidx + r / 32] = src[i0 * cols + e];
    }
    __syncthreads();
    const int s = threadIdx.x;
    if (s < ns)
      for (int cidx = 0; cidx < cols; ++cidx) planes[(size_t)m.col[cidx] * ld + i0 + s] = tile[s * cols + cidx + s / 32];
    __syncthreads();
  }
}

__global__ __launch_bounds__(NIS_TPB) void k_emit(const double* __restrict__ planes, long ld, long N, int d, ColMap m,
                                                  double* __restrict__ xj, long plane_gs) {
  extern __shared__ double tile[];
  const int dd = d + 1;
  planes += (size_t)blockIdx.y * plane_gs;          // batch g of a launch set: own plane set, own (N, d+1) block of rows
  xj += (size_t)blockIdx.y * N * dd;
  const long nblk = (N + NIS_TPB - 1) / NIS_TPB;
  for (long blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    const long i0 = blk * NIS_TPB;
    const int ns = (int)min((long)NIS_TPB, N - i0);
    const int s = threadIdx.x;
    if (s < ns)
      for (int cidx = 0; cidx < dd; ++cidx) tile[s * dd + cidx + s / 32] = planes[(size_t)m.col[cidx] * ld + i0 + s];
    __syncthreads();
    const int tot = ns * dd;
    for (int e = threadIdx.x; e < tot; e += NIS_TPB) {
      const int r = e / dd, cidx = e - r * dd;
      xj[i0 * dd + e] = tile[r * dd + cidx + r / 32];
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------
// In-kernel draws for every model outside the d = 2 direct-row class: thread = sample, Q = ceil(d / 4) Philox calls,
//   w[g][n][c] = (philox(ctr = {q lo, q hi, s lo, s hi}, key = {seed lo, seed hi})[c & 3] >> 8) * 2^-24,  q = n Q + (c >> 2),  s = stream0 + g
// (the float32 lattice torch's uniform_ produces, manager.py:390), written straight to the float64 state planes -- no input
// buffer, no torch.rand launch, no transpose -- with the first cell's BN_0 records riding along: plain (count, mean, M2) per
// pass-through feature, or (COV) with the co-moments of the pass-through block so that a matrix-core first cell derives BN_1
// too (CellK::dcov).  Sums are taken about 1/2, the mean of the draws.
template <bool COV>
__global__ __launch_bounds__(NIS_TPB) void k_ingest_rng(unsigned long long seed, unsigned long long stream0,
                                                        const unsigned long long* __restrict__ state, long N, int d,
                                                        double* __restrict__ planes, long ld, float* __restrict__ w_out, IngestStats st) {
  constexpr int NW = NIS_TPB / NIS_WAVE, NF = NIS_PMAX_FUSED;
  static_assert(NF == NIS_COV_PM, "one feature array serves both riders");
  __shared__ double wshift[NW][NF];
  __shared__ double wmom[NW][NF][3];
  __shared__ double lsh[NF];
  const int g = st.g0 + blockIdx.y;
  if (state) {
    seed = state[0];
    stream0 = state[1];
  }
  const unsigned long long sid = stream0 + (unsigned long long)g;
  planes += (size_t)g * st.plane_gs;
  if (blockIdx.x == 0 && g == 0 && threadIdx.x == 0 && st.status) *st.status = 0;
  if (threadIdx.x < NW * NF) (&wshift[0][0])[threadIdx.x] = 0.5;
  if (threadIdx.x < NF) lsh[threadIdx.x] = 0.5;
  const int Q = (d + 3) >> 2;
  double S1[NF], S2[COV ? NIS_COV_NS2 : NF];
#pragma unroll
  for (int f = 0; f < NF; ++f) S1[f] = 0.0;
#pragma unroll
  for (int k = 0; k < (COV ? NIS_COV_NS2 : NF); ++k) S2[k] = 0.0;
  double cnt = 0.0;
  const long stride = (long)gridDim.x * NIS_TPB;
  for (long i = (long)blockIdx.x * NIS_TPB + threadIdx.x; i < N; i += stride) {
    double v[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) v[f] = 0.5;                    // padding features: exactly the shift
    for (int q = 0; q < Q; ++q) {
      // d == 2: the pair mapping of the direct-row chain and of nis_philox_rows (one call = two samples: counter n >> 1, words
      // 2 (n & 1) + c), so that EVERY d = 2 model draws the numbers nis_philox_rows materialises -- not only the SPEC class
      const bool pair = d == 2;
      const unsigned long long ctr = pair ? (unsigned long long)(i >> 1) : (unsigned long long)i * (unsigned long long)Q + (unsigned long long)q;
      unsigned r[4];
      nis_philox4x32_10((unsigned)ctr, (unsigned)(ctr >> 32), (unsigned)sid, (unsigned)(sid >> 32), (unsigned)seed, (unsigned)(seed >> 32), r);
      float u[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) u[j] = (float)(r[j] >> 8) * 0x1p-24f;
      if (pair && (i & 1)) {
        u[0] = u[2];
        u[1] = u[3];
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (4 * q + j < d) {
          planes[(size_t)(4 * q + j) * ld + i] = (double)u[j];
          if (w_out) w_out[((size_t)g * N + i) * d + 4 * q + j] = u[j];
        }
#pragma unroll
      for (int f = 0; f < NF; ++f)
        if (f < st.P && (st.sel[f] >> 2) == q) {
          const int j = st.sel[f] & 3;
          v[f] = (double)(j == 0 ? u[0] : j == 1 ? u[1] : j == 2 ? u[2] : u[3]);
        }
    }
    if (st.P > 0) {
      double t[NF];
#pragma unroll
      for (int f = 0; f < NF; ++f) t[f] = v[f] - 0.5;
      if (COV) {
        int k = 0;
#pragma unroll
        for (int a = 0; a < NF; ++a) {
          S1[a] += t[a];
#pragma unroll
          for (int b = a; b < NF; ++b, ++k) S2[COV ? k : 0] = fma(t[a], t[b], S2[COV ? k : 0]);
        }
      } else {
#pragma unroll
        for (int f = 0; f < NF; ++f) {
          S1[f] += t[f];
          S2[COV ? 0 : f] = fma(t[f], t[f], S2[COV ? 0 : f]);
        }
      }
      cnt += 1.0;
    }
  }
  if (st.P > 0) {
    __syncthreads();                                            // wshift / lsh
    double* out = st.partials + ((size_t)g * st.maxb + blockIdx.x) * st.wm * 3;
    if constexpr (COV) {
      cov_records_emit<NIS_TPB>(S1, S2, cnt, lsh, st.P, out);
    } else {
      emit_records<NF, NW>(S1, S2, cnt, wshift, wmom, st.P, out);
    }
  }
}

// the same draws as (n_batches, N, d) float32 rows (tests feed them to the oracle; sample_hook-free fed paths)
__global__ __launch_bounds__(NIS_TPB) void k_philox_rows2(unsigned long long seed, unsigned long long stream0, long N, float* __restrict__ out) {
  const long npair = (N + 1) / 2;
  const unsigned long long s = stream0 + blockIdx.y;
  float* o = out + (size_t)blockIdx.y * N * 2;
  for (long c = (long)blockIdx.x * NIS_TPB + threadIdx.x; c < npair; c += (long)gridDim.x * NIS_TPB) {
    unsigned r[4];
    nis_philox4x32_10((unsigned)c, (unsigned)((unsigned long long)c >> 32), (unsigned)s, (unsigned)(s >> 32), (unsigned)seed, (unsigned)(seed >> 32), r);
    o[4 * c + 0] = (float)(r[0] >> 8) * 0x1p-24f;
    o[4 * c + 1] = (float)(r[1] >> 8) * 0x1p-24f;
    if (2 * c + 1 < N) {
      o[4 * c + 2] = (float)(r[2] >> 8) * 0x1p-24f;
      o[4 * c + 3] = (float)(r[3] >> 8) * 0x1p-24f;
    }
  }
}
__global__ __launch_bounds__(NIS_TPB) void k_philox_rows_d(unsigned long long seed, unsigned long long stream0, long N, int d, float* __restrict__ out) {
  const unsigned long long s = stream0 + blockIdx.y;
  float* o = out + (size_t)blockIdx.y * N * d;
  const int Q = (d + 3) >> 2;
  for (long i = (long)blockIdx.x * NIS_TPB + threadIdx.x; i < N; i += (long)gridDim.x * NIS_TPB)
    for (int q = 0; q < Q; ++q) {
      const unsigned long long ctr = (unsigned long long)i * (unsigned long long)Q + (unsigned long long)q;
      unsigned r[4];
      nis_philox4x32_10((unsigned)ctr, (unsigned)(ctr >> 32), (unsigned)s, (unsigned)(s >> 32), (unsigned)seed, (unsigned)(seed >> 32), r);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (4 * q + j < d) o[(size_t)i * d + 4 * q + j] = (float)(r[j] >> 8) * 0x1p-24f;
    }
}

// ---------------------------------------------------------------------------------------------
// launchers (api.hip plans the grids)
hipError_t nis_launch_ingest(const void* w, int w_dtype, int wcols, long N, double* planes, long ld, const IngestStats& st, int grid,
                             int Gy, bool direct, hipStream_t s) {
  const size_t tile_bytes = ((size_t)NIS_TPB * wcols + 16) * sizeof(double);
  if (direct)
    hipLaunchKernelGGL((k_ingest2<float, false>), dim3(grid, Gy), dim3(NIS_TPB), 0, s, (const float*)w, N, planes, ld, st);
  else if (wcols == 2 && st.P <= 2 && w_dtype == NIS_F64)
    hipLaunchKernelGGL(k_ingest2<double>, dim3(grid, Gy), dim3(NIS_TPB), 0, s, (const double*)w, N, planes, ld, st);
  else if (wcols == 2 && st.P <= 2)
    hipLaunchKernelGGL(k_ingest2<float>, dim3(grid, Gy), dim3(NIS_TPB), 0, s, (const float*)w, N, planes, ld, st);
  else if (w_dtype == NIS_F64)
    hipLaunchKernelGGL(k_ingest<double>, dim3(grid, Gy), dim3(NIS_TPB), tile_bytes, s, (const double*)w, N, wcols, planes, ld, st);
  else
    hipLaunchKernelGGL(k_ingest<float>, dim3(grid, Gy), dim3(NIS_TPB), tile_bytes, s, (const float*)w, N, wcols, planes, ld, st);
  return hipGetLastError();
}

hipError_t nis_launch_ingest2_rng(unsigned long long seed, unsigned long long stream0, const unsigned long long* state, long N,
                                  float* rows, long rows_gs, long plane_ld, float* w_out, const IngestStats& st, int grid, int Gy,
                                  hipStream_t s) {
  hipLaunchKernelGGL(k_ingest2_rng, dim3(grid, Gy), dim3(NIS_TPB), 0, s, seed, stream0, state, N, rows, rows_gs, plane_ld, w_out, st);
  return hipGetLastError();
}

hipError_t nis_launch_ingest_rng(unsigned long long seed, unsigned long long stream0, const unsigned long long* state, long N, int d,
                                 double* planes, long ld, float* w_out, const IngestStats& st, int grid, int Gy, hipStream_t s) {
  if (st.cov)
    hipLaunchKernelGGL(k_ingest_rng<true>, dim3(grid, Gy), dim3(NIS_TPB), 0, s, seed, stream0, state, N, d, planes, ld, w_out, st);
  else
    hipLaunchKernelGGL(k_ingest_rng<false>, dim3(grid, Gy), dim3(NIS_TPB), 0, s, seed, stream0, state, N, d, planes, ld, w_out, st);
  return hipGetLastError();
}

hipError_t nis_launch_philox_rows(unsigned long long seed, unsigned long long stream0, int G, long N, int d, float* out, hipStream_t s) {
  const long work = d == 2 ? (N + 1) / 2 : N;
  const int gx = (int)std::max<long>(1, std::min<long>(2048, (work + NIS_TPB - 1) / NIS_TPB));
  if (d == 2)
    hipLaunchKernelGGL(k_philox_rows2, dim3(gx, G), dim3(NIS_TPB), 0, s, seed, stream0, N, out);
  else
    hipLaunchKernelGGL(k_philox_rows_d, dim3(gx, G), dim3(NIS_TPB), 0, s, seed, stream0, N, d, out);
  return hipGetLastError();
}

hipError_t nis_launch_sync_reduce(const double* part, int nparts, int WM, double* out, hipStream_t s) {
  hipLaunchKernelGGL(k_sync_reduce, dim3(1), dim3(NIS_TPB), 0, s, part, nparts, WM, out);
  return hipGetLastError();
}

hipError_t nis_launch_scatter_rows(const double* src, long N, int cols, const ColMap& m, double* planes, long ld, int grid, hipStream_t s) {
  const size_t tile_bytes = ((size_t)NIS_TPB * cols + 16) * sizeof(double);
  hipLaunchKernelGGL(k_scatter_rows, dim3(grid), dim3(NIS_TPB), tile_bytes, s, src, N, cols, m, planes, ld);
  return hipGetLastError();
}

hipError_t nis_launch_emit(const double* planes, long ld, long N, int d, const ColMap& m, double* xj, int grid, hipStream_t s, int G,
                           long plane_gs) {
  const size_t tile_bytes = ((size_t)NIS_TPB * (d + 1) + 16) * sizeof(double);
  hipLaunchKernelGGL(k_emit, dim3(grid, G), dim3(NIS_TPB), tile_bytes, s, planes, ld, N, d, m, xj, plane_gs);
  return hipGetLastError();
}
