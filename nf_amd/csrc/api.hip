// Host side of libnis_hip.so (C ABI of include/nis_hip.h) + the small streaming kernels:
// parameter packing, ingest (row-major w -> column planes), emit (planes -> row-major XJ),
// built-in integrands and the moment reductions.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "flow_kernels.h"
#include "flow_mega.h"

static thread_local std::string g_err;
static int fail_hip(hipError_t e, const char* what) {
  g_err = std::string(what) + ": " + hipGetErrorString(e);
  return NIS_E_HIP;
}
#define HIP_OK(call)                                   \
  do {                                                 \
    hipError_t e_ = (call);                            \
    if (e_ != hipSuccess) return fail_hip(e_, #call);  \
  } while (0)

extern "C" const char* nis_last_error(void) { return g_err.c_str(); }
extern "C" int nis_version(void) { return 100; }
extern "C" int nis_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

// ---------------------------------------------------------------------------------------------
// live per-launch timing of one kernel class
static struct {
  bool on = false;
  int cls = -1;
  std::vector<hipEvent_t> ev;   // pairs: start, stop
  int used = 0;
} g_prof;

void nis_prof_mark(int kernel_class, int begin, hipStream_t s) {
  if (!g_prof.on || kernel_class != g_prof.cls) return;
  if (g_prof.used + 1 >= (int)g_prof.ev.size() && begin) return;
  if (begin) {
    (void)hipEventRecord(g_prof.ev[g_prof.used], s);
  } else if (g_prof.used + 1 < (int)g_prof.ev.size()) {
    (void)hipEventRecord(g_prof.ev[g_prof.used + 1], s);
    g_prof.used += 2;
  }
}

extern "C" int nis_prof_disable(void) {
  for (hipEvent_t e : g_prof.ev) (void)hipEventDestroy(e);
  g_prof.ev.clear();
  g_prof.on = false;
  g_prof.used = 0;
  return NIS_OK;
}

extern "C" int nis_prof_enable(int32_t kernel_class, int32_t max_launches) {
  nis_prof_disable();
  if (max_launches < 1) return NIS_E_ARG;
  g_prof.ev.resize(2 * (size_t)max_launches);
  for (auto& e : g_prof.ev) HIP_OK(hipEventCreate(&e));
  g_prof.cls = kernel_class;
  g_prof.used = 0;
  g_prof.on = true;
  return NIS_OK;
}

extern "C" int nis_prof_read(double* total_ms, int32_t* launches) {
  if (!total_ms || !launches) return NIS_E_ARG;
  double tot = 0.0;
  for (int i = 0; i + 1 < g_prof.used; i += 2) {
    HIP_OK(hipEventSynchronize(g_prof.ev[i + 1]));
    float ms = 0.f;
    HIP_OK(hipEventElapsedTime(&ms, g_prof.ev[i], g_prof.ev[i + 1]));
    tot += ms;
  }
  *total_ms = tot;
  *launches = g_prof.used / 2;
  return NIS_OK;
}

// ---------------------------------------------------------------------------------------------
// sizes
static inline int cell_T(const nis_cell_desc& c) { return c.d - c.P; }
static inline int cell_out(const nis_cell_desc& c) { return c.kind == NIS_PWQUAD ? 2 * c.K + 1 : c.kind == NIS_AFFINE ? 2 : c.K; }
static inline bool cell_hidden_bias(const nis_cell_desc& c) { return c.kind == NIS_AFFINE; }
static inline int cell_R(const nis_cell_desc& c) { return cell_T(c) * cell_out(c); }
static inline int cell_feat(const nis_cell_desc& c, int l) { return l == 0 ? c.P : c.width[l - 1]; }

extern "C" size_t nis_cell_param_count(const nis_cell_desc* c) {
  size_t n = 0;
  for (int l = 0; l <= c->n_hidden; ++l) {
    const int F = cell_feat(*c, l);
    const int O = l < c->n_hidden ? c->width[l] : cell_R(*c);
    n += 2 * (size_t)F + (size_t)O * F;
    if (l < c->n_hidden && cell_hidden_bias(*c)) n += O;
  }
  return n + cell_R(*c);
}
extern "C" size_t nis_cell_running_count(const nis_cell_desc* c) {
  size_t n = 0;
  for (int l = 0; l <= c->n_hidden; ++l) n += 2 * (size_t)cell_feat(*c, l);
  return n;
}

struct CellLayout {
  int WM, KM;
  size_t off_wp, off_wlast, off_bias, off_gb, off_hb, off_coef, off_part, off_gstat;   // in doubles from the workspace body
  size_t off_wm, off_wmlast, off_biaspad;                                      // matrix-core fragment images (WM >= 16)
  int RT;                                                                      // row tiles of the last layer
};

struct nis_flow {
  int d, n_cells;
  std::vector<nis_cell_desc> cells;
  std::vector<int> bins_row;
  int out_col[NIS_MAX_DIM];
  // packing cache: the padded parameter image in the workspace stays valid while the caller's token,
  // BN mode, workspace, batch count and parameter pointers are unchanged
  int64_t packed_token = 0;
  int packed_mode = -1;
  int packed_G = 0;
  const void* packed_ws = nullptr;
  std::vector<const double*> packed_params, packed_running;
  const void* mega_ws = nullptr;     // workspace whose hand-off granules have been zeroed (flow_mega.hip)
};

static const size_t WS_HEADER = 256;   // bytes: status word + padding

// Workspace body for G batched iterations: per cell the zero-padded parameter image (shared by all batches),
// then per-batch BN coefficients, partial records and batch statistics; finally the integrand's moment records.
// The persistent whole-step kernel (flow_mega.hip) covers the README / C1 / C2 model: d = 2, two PWQuad cells with 4 bins,
// one pass-through column, three hidden layers at most 3 wide.
static bool mega_model(const nis_flow* f) {
  if (f->d != 2 || f->n_cells != 2) return false;
  for (int ci = 0; ci < 2; ++ci) {
    const nis_cell_desc& c = f->cells[ci];
    if (c.kind != NIS_PWQUAD || c.P != 1 || c.K != 4 || c.n_hidden != 3) return false;
    for (int l = 0; l < 3; ++l)
      if (c.width[l] > 3) return false;
  }
  return true;
}

static size_t flow_layout(const nis_flow* f, int G, std::vector<CellLayout>* lay, size_t* off_fpart, size_t* off_mega = nullptr) {
  size_t off = 0;
  for (int i = 0; i < f->n_cells; ++i) {
    const nis_cell_desc& c = f->cells[i];
    int m = c.P;
    for (int l = 0; l < c.n_hidden; ++l) m = std::max(m, c.width[l]);
    CellLayout L;
    L.WM = nis_pick_wm(m);
    L.KM = c.kind == NIS_PWQUAD ? nis_pick_km(c.K) : 4;
    if (c.kind == NIS_PWQUAD && L.KM == c.K) L.KM = -L.KM;        // K == KM: guard-free specialisation
    const size_t WM = L.WM, R = cell_R(c), nl = c.n_hidden;
    L.off_wp = off;      off += nl * WM * WM;
    L.off_wlast = off;   off += R * WM;
    L.off_bias = off;    off += (R + 7) / 8 * 8;
    L.off_gb = off;      off += (nl + 1) * 2 * WM;
    L.off_hb = off;      off += (nl + 1) * WM;        // bias of the Linear in front of BN layer l (zero unless NIS_AFFINE)
    L.off_coef = off;    off += (size_t)G * (nl + 1) * 2 * WM;
    L.off_gstat = off;   off += (size_t)G * (nl + 1) * 2 * WM;
    L.off_part = off;    off += (nl + 1) * (size_t)G * NIS_MAXB * WM * 3;
    L.RT = (int)((R + 15) / 16);
    L.off_wm = L.off_wmlast = L.off_biaspad = 0;
    if (WM >= 16) {
      L.off_wm = off;      off += nl * WM * WM;
      L.off_wmlast = off;  off += (size_t)L.RT * 16 * WM;
      L.off_biaspad = off; off += (size_t)L.RT * 16;
    }
    if (lay) lay->push_back(L);
  }
  if (off_fpart) *off_fpart = off;
  off += (size_t)G * NIS_MAXB * 4;
  off = (off + 31) / 32 * 32;                                   // 256-byte aligned: the scratch starts with its counter block
  if (off_mega) *off_mega = off;
  if (mega_model(f)) off += (nis_mega_scratch_bytes() + 7) / 8;
  return off;
}

static int validate_cell(const nis_cell_desc& c, int d) {
  if (c.kind != NIS_PWLIN && c.kind != NIS_PWQUAD && c.kind != NIS_AFFINE) return NIS_E_ARG;
  if (c.d != d || c.P < 1 || c.P >= d || (c.K < 1 && c.kind != NIS_AFFINE)) return NIS_E_ARG;
  if (c.n_hidden < 1 || c.n_hidden > NIS_MAX_HIDDEN) return NIS_E_UNSUPPORTED;
  int m = c.P;
  for (int l = 0; l < c.n_hidden; ++l) {
    if (c.width[l] < 1) return NIS_E_ARG;
    m = std::max(m, c.width[l]);
  }
  if (nis_pick_wm(m) < 0) return NIS_E_UNSUPPORTED;
  if (c.kind == NIS_PWQUAD && nis_pick_km(c.K) < 0) return NIS_E_UNSUPPORTED;
  if (c.kind == NIS_PWLIN && c.K > NIS_MAX_BINS) return NIS_E_UNSUPPORTED;
  for (int j = 0; j < d; ++j)
    if (c.in_col[j] < 0 || c.in_col[j] >= d) return NIS_E_ARG;
  return NIS_OK;
}

extern "C" int nis_flow_create(const nis_cell_desc* cells, int32_t n_cells, int32_t d, const int32_t* out_col, nis_flow** out) {
  if (!cells || !out || !out_col || n_cells < 1 || n_cells > NIS_MAX_CELLS || d < 2 || d > NIS_MAX_DIM) return NIS_E_ARG;
  nis_flow* f = new nis_flow();
  f->d = d;
  f->n_cells = n_cells;
  int brow = 0;
  for (int i = 0; i < n_cells; ++i) {
    int rc = validate_cell(cells[i], d);
    if (rc != NIS_OK) {
      delete f;
      return rc;
    }
    f->cells.push_back(cells[i]);
    f->bins_row.push_back(brow);
    brow += cell_T(cells[i]);
  }
  for (int j = 0; j < d; ++j) {
    if (out_col[j] < 0 || out_col[j] >= d) {
      delete f;
      return NIS_E_ARG;
    }
    f->out_col[j] = out_col[j];
  }
  f->packed_params.assign(n_cells, nullptr);
  f->packed_running.assign(n_cells, nullptr);
  *out = f;
  return NIS_OK;
}

extern "C" void nis_flow_destroy(nis_flow* f) { delete f; }

extern "C" size_t nis_flow_workspace_bytes(const nis_flow* f, int64_t n_batches) {
  const int G = (int)std::max<int64_t>(1, n_batches);
  return WS_HEADER + flow_layout(f, G, nullptr, nullptr) * sizeof(double);
}

// ---------------------------------------------------------------------------------------------
// k_pack: reference-order parameter blob -> zero-padded WM-strided weights (+ eval-mode BN affine).
__global__ __launch_bounds__(NIS_TPB) void k_pack(CellK c, int WM, const double* __restrict__ prm, const double* __restrict__ run,
                                                  int eval_mode, double* wp, double* wlast, double* bias, double* gb, double* hb,
                                                  int hidden_bias) {
  const int tid = blockIdx.x * NIS_TPB + threadIdx.x, nth = gridDim.x * NIS_TPB;
  size_t off = 0;
  int roff = 0;
  for (int l = 0; l <= c.L; ++l) {
    const int F = cellk_feat(c, l);
    const int O = l < c.L ? c.width[l] : c.R;
    const double* gam = prm + off;
    const double* bet = prm + off + F;
    const double* W = prm + off + 2 * F;
    // bias of the Linear that FEEDS this BN layer (NIS_AFFINE): the kernels evaluate the Linear without it -- a constant
    // in front of a BatchNorm cancels -- and account for it where it is visible: the running mean and the eval-mode shift
    const double* hbin = (hidden_bias && l >= 1) ? prm + off - F : nullptr;
    for (int j = tid; j < WM; j += nth) {
      const double g = j < F ? gam[j] : 0.0, b = j < F ? bet[j] : 0.0;
      const double hbv = (hbin && j < F) ? hbin[j] : 0.0;
      gb[(l * 2 + 0) * WM + j] = g;
      gb[(l * 2 + 1) * WM + j] = b;
      if (hb) hb[l * WM + j] = hbv;
      if (eval_mode) {                       // BatchNorm1d.eval(): running statistics
        double sc = 0.0, sh = 0.0;
        if (j < F) {
          sc = g / sqrt(run[roff + F + j] + NIS_BN_EPS);
          sh = b - (run[roff + j] - hbv) * sc;
        }
        c.coef[(l * 2 + 0) * WM + j] = sc;
        c.coef[(l * 2 + 1) * WM + j] = sh;
      }
    }
    if (l < c.L) {
      double* dst = wp + (size_t)l * WM * WM;
      for (int e = tid; e < WM * WM; e += nth) {
        const int j = e / WM, i = e % WM;
        dst[e] = (j < O && i < F) ? W[(size_t)j * F + i] : 0.0;
      }
    } else {
      for (size_t e = tid; e < (size_t)c.R * WM; e += nth) {
        const int r = (int)(e / WM), i = (int)(e % WM);
        wlast[e] = (i < F) ? W[(size_t)r * F + i] : 0.0;
      }
      const double* bs = W + (size_t)O * F;
      for (int r = tid; r < c.R; r += nth) bias[r] = bs[r];
    }
    off += 2 * (size_t)F + (size_t)O * F;
    if (l < c.L && hidden_bias) off += O;
    roff += 2 * F;
  }
}

// k_pack_mfma: padded row-major images -> per-(row tile, k-step) fragments of 64 contiguous doubles, lane l holding
// W[16 jt + (l & 15)][4 q + (l >> 4)]  (the A operand of v_mfma_f64_16x16x4_f64).
// affT > 0 (NIS_AFFINE with T transformed columns): the last layer's rows [scale | shift][T] are interleaved to
// [t][scale, shift], so that a column's outputs are contiguous like those of the piecewise cells.
__global__ __launch_bounds__(NIS_TPB) void k_pack_mfma(int WM, int L, int R, int RT, const double* __restrict__ wp,
                                                       const double* __restrict__ wlast, const double* __restrict__ bias,
                                                       double* __restrict__ wm_all, double* __restrict__ wm_last,
                                                       double* __restrict__ bias_pad, int affT) {
  const size_t tid = (size_t)blockIdx.x * NIS_TPB + threadIdx.x, nth = (size_t)gridDim.x * NIS_TPB;
  const int JT = WM / 16, KQ = WM / 4;
  const size_t nh = (size_t)L * WM * WM;
  for (size_t e = tid; e < nh; e += nth) {
    const int lane = (int)(e & 63);
    const size_t frag = e >> 6;
    const int q = (int)(frag % KQ), jt = (int)((frag / KQ) % JT), l = (int)(frag / ((size_t)KQ * JT));
    const int j = 16 * jt + (lane & 15), k = 4 * q + (lane >> 4);
    wm_all[e] = wp[(size_t)l * WM * WM + (size_t)j * WM + k];
  }
  const size_t nl = (size_t)RT * KQ * 64;
  for (size_t e = tid; e < nl; e += nth) {
    const int lane = (int)(e & 63);
    const size_t frag = e >> 6;
    const int q = (int)(frag % KQ), jt = (int)(frag / KQ);
    const int j = 16 * jt + (lane & 15), k = 4 * q + (lane >> 4);
    const int jo = affT > 0 ? (j & 1) * affT + (j >> 1) : j;          // packed row j <- blob row jo
    wm_last[e] = j < R ? wlast[(size_t)jo * WM + k] : 0.0;
  }
  for (size_t e = tid; e < (size_t)RT * 16; e += nth) {
    const int j = (int)e, jo = affT > 0 ? (j & 1) * affT + (j >> 1) : j;
    bias_pad[e] = e < (size_t)R ? bias[jo] : 0.0;
  }
}

// ---------------------------------------------------------------------------------------------
// ingest / emit.  256 samples per workgroup are transposed through LDS so that both the row-major
// side and the plane side move as contiguous 8-byte-per-lane streams.
struct ColMap {
  int col[NIS_MAX_DIM + 1];
};

struct IngestStats {
  int P;                           // 0: no statistics
  int sel[NIS_PMAX_FUSED];         // input column of pass-through feature p of the first cell
  int wm;                          // record stride of that cell
  double* partials;                // its BN_0 records (one per workgroup and batch)
  int* status;                     // zeroed here (first launch of a forward)
  int maxb;                        // record stride between batches
  long plane_gs;                   // doubles between the plane sets of consecutive batches
  int g0;                          // first batch of this launch (blockIdx.y counts from it)
};

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
// nis_philox_rows / flow_mega.hip: sample n takes words 2 (n & 1), 2 (n & 1) + 1 of the call with counter n >> 1 on stream
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
                                                  double* __restrict__ xj) {
  extern __shared__ double tile[];
  const int dd = d + 1;
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
// moments: partial record per workgroup = (count, mean, M2, max)
__device__ __forceinline__ void block_moments(double v, bool valid, bool& first, double& shift, double& S1, double& S2, double& cnt,
                                              double& vmax) {
  if (first) {
    shift = wave_first(v);
    first = false;
  }
  if (valid) {
    const double t = v - shift;
    S1 += t;
    S2 = fma(t, t, S2);
    cnt += 1.0;
    vmax = fmax(vmax, v);
  }
}

__device__ __forceinline__ void block_moments_finish(double shift, double S1, double S2, double cnt, double vmax, double* part) {
  __shared__ double wm[NIS_TPB / NIS_WAVE][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const double nw = wave_sum(cnt), s1 = wave_sum(S1), s2 = wave_sum(S2), mx = wave_max(vmax);
  if (lane == 0) {
    wm[wave][0] = nw;
    wm[wave][1] = nw > 0.0 ? shift + s1 / nw : 0.0;
    wm[wave][2] = nw > 0.0 ? fmax(s2 - s1 * s1 / nw, 0.0) : 0.0;
    wm[wave][3] = mx;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    Mom t = {0.0, 0.0, 0.0};
    double m = -__builtin_huge_val();
    for (int w = 0; w < NIS_TPB / NIS_WAVE; ++w) {
      Mom q = {wm[w][0], wm[w][1], wm[w][2]};
      t = mom_merge(t, q);
      m = fmax(m, wm[w][3]);
    }
    part[0] = t.n;
    part[1] = t.mean;
    part[2] = t.m2;
    part[3] = m;
  }
}

struct IntegrandArgs {
  int out_col[NIS_MAX_DIM];
  int d, kind, jplane, pad;
  double p[4];
};

// element (column c, sample i) at base[c * cs + i * ss]: column planes (cs = ld, ss = 1) or row-major rows (cs = 1, ss = row stride)
__device__ __forceinline__ double eval_integrand_at(const IntegrandArgs& a, const double* base, long cs, long ss, long i) {
  if (a.kind == NIS_F_CAMEL) {                                                  // README.md:35-36
    const double x0 = base[(size_t)a.out_col[0] * cs + i * ss], x1 = base[(size_t)a.out_col[1] * cs + i * ss];
    const double s2 = a.p[2] * a.p[2];
    const double e1 = -((x0 - a.p[0]) * (x0 - a.p[0]) + (x1 - a.p[0]) * (x1 - a.p[0])) / s2;
    const double e2 = -((x0 - a.p[1]) * (x0 - a.p[1]) + (x1 - a.p[1]) * (x1 - a.p[1])) / s2;
    return exp(e1) + exp(e2);
  } else if (a.kind == NIS_F_NORMAL) {                                          // manager.py:39-40
    const double mu = a.p[0], sg = a.p[1];
    double s = 0.0;
    for (int c = 0; c < a.d; ++c) {
      const double x = base[(size_t)a.out_col[c] * cs + i * ss] - mu;
      s += (x * x) / (2 * sg * sg);
    }
    return exp(-s) / (sg * sqrt(pow(2 * M_PI, (double)a.d)));
  }
  return 1.0;
}
__device__ __forceinline__ double eval_integrand(const IntegrandArgs& a, const double* planes, long ld, long i) {
  return eval_integrand_at(a, planes, ld, 1, i);
}

// f(x) of a built-in integrand for the rows of a row-major (N, d) matrix (the training loop's f(X), manager.py:241-246)
__global__ __launch_bounds__(NIS_TPB) void k_integrand_rows(const double* __restrict__ x, long rs, long N, IntegrandArgs a,
                                                            double* __restrict__ out) {
  const long stride = (long)gridDim.x * NIS_TPB;
  for (long i = (long)blockIdx.x * NIS_TPB + threadIdx.x; i < N; i += stride) out[i] = eval_integrand_at(a, x, 1, rs, i);
}

__global__ __launch_bounds__(NIS_TPB) void k_integrand_moments(const double* __restrict__ planes, long ld, long N, IntegrandArgs a,
                                                               double* __restrict__ fres, double* __restrict__ partials) {
  double S1 = 0, S2 = 0, cnt = 0, vmax = -__builtin_huge_val(), shift = 0;
  bool first = true;
  const long stride = (long)gridDim.x * NIS_TPB;
  const long nit = (N + stride - 1) / stride;
  for (long it = 0; it < nit; ++it) {
    const long i = it * stride + (long)blockIdx.x * NIS_TPB + threadIdx.x;
    const bool valid = i < N;
    double v = 0.0;
    if (valid) {
      v = eval_integrand(a, planes, ld, i) * planes[(size_t)a.jplane * ld + i];
      if (fres) fres[i] = v;
    }
    block_moments(v, valid, first, shift, S1, S2, cnt, vmax);
  }
  block_moments_finish(shift, S1, S2, cnt, vmax, partials + (size_t)blockIdx.x * 4);
}

__global__ __launch_bounds__(NIS_TPB) void k_moments(const double* __restrict__ va, const double* __restrict__ vb, long N,
                                                     double* __restrict__ partials) {
  double S1 = 0, S2 = 0, cnt = 0, vmax = -__builtin_huge_val(), shift = 0;
  bool first = true;
  const long stride = (long)gridDim.x * NIS_TPB;
  const long nit = (N + stride - 1) / stride;
  for (long it = 0; it < nit; ++it) {
    const long i = it * stride + (long)blockIdx.x * NIS_TPB + threadIdx.x;
    const bool valid = i < N;
    double v = 0.0;
    if (valid) v = vb ? va[i] * vb[i] : va[i];
    block_moments(v, valid, first, shift, S1, S2, cnt, vmax);
  }
  block_moments_finish(shift, S1, S2, cnt, vmax, partials + (size_t)blockIdx.x * 4);
}

// one workgroup: merge the per-workgroup records in a fixed order -> out4 = {n, mean, M2, max}
__device__ __forceinline__ void moments_final_body(const double* __restrict__ partials, int nparts, double* __restrict__ out4) {
  __shared__ double sm[NIS_TPB][4];
  Mom t = {0.0, 0.0, 0.0};
  double m = -__builtin_huge_val();
  for (int b = threadIdx.x; b < nparts; b += NIS_TPB) {
    Mom q = {partials[b * 4 + 0], partials[b * 4 + 1], partials[b * 4 + 2]};
    t = mom_merge(t, q);
    m = fmax(m, partials[b * 4 + 3]);
  }
  sm[threadIdx.x][0] = t.n;
  sm[threadIdx.x][1] = t.mean;
  sm[threadIdx.x][2] = t.m2;
  sm[threadIdx.x][3] = m;
  __syncthreads();
  for (int s = NIS_TPB / 2; s > 0; s >>= 1) {
    if (threadIdx.x < s) {
      Mom a = {sm[threadIdx.x][0], sm[threadIdx.x][1], sm[threadIdx.x][2]};
      Mom b = {sm[threadIdx.x + s][0], sm[threadIdx.x + s][1], sm[threadIdx.x + s][2]};
      a = mom_merge(a, b);
      sm[threadIdx.x][0] = a.n;
      sm[threadIdx.x][1] = a.mean;
      sm[threadIdx.x][2] = a.m2;
      sm[threadIdx.x][3] = fmax(sm[threadIdx.x][3], sm[threadIdx.x + s][3]);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out4[0] = sm[0][0];
    out4[1] = sm[0][1];
    out4[2] = sm[0][2];
    out4[3] = sm[0][3];
  }
}

__global__ __launch_bounds__(NIS_TPB) void k_moments_final(const double* __restrict__ partials, int nparts, double* __restrict__ out4,
                                                           int maxb) {
  moments_final_body(partials + (size_t)blockIdx.x * maxb * 4, nparts, out4 + (size_t)blockIdx.x * 4);
}

// Batched iterations (G > 1): ONE launch at the end of the launch set applies the G momentum-0.1 running-statistics
// updates of every BN layer of every cell in batch order (BatchNorm1d semantics) and merges the fused integrand's moment
// records -- three ~4.5 us launches per C2 launch set otherwise.  Blocks [0, blk_off[ncell]): one per (cell, BN layer);
// blocks behind them: one per batch for the moment records (when fpart != nullptr).
#define NIS_EPI_CELLS 16
struct EpiCell {
  double* running;
  const double* gstat;
  long long* tracked;
  int L, WM, P;
  int width[NIS_MAX_HIDDEN];
};
struct EpiJob {
  EpiCell cell[NIS_EPI_CELLS];
  int blk_off[NIS_EPI_CELLS + 1];
  int ncell, G;
  const double* fpart;
  double* out4;
  int fparts, maxb;
  unsigned long long* rng_state;     // {seed, next stream}: next stream += G (last thing the launch set does)
};
__global__ __launch_bounds__(NIS_TPB) void k_group_epilogue(EpiJob job) {
  const int b = blockIdx.x;
  const int nupd = job.blk_off[job.ncell];
  if (b >= nupd) {
    const int g = b - nupd;
    moments_final_body(job.fpart + (size_t)g * job.maxb * 4, job.fparts, job.out4 + (size_t)g * 4);
    if (g == 0 && threadIdx.x == 0 && job.rng_state) job.rng_state[1] += (unsigned long long)job.G;
    return;
  }
  int ci = 0;
  while (ci + 1 < job.ncell && b >= job.blk_off[ci + 1]) ++ci;
  const EpiCell& c = job.cell[ci];
  const int l = b - job.blk_off[ci], j = threadIdx.x;
  const int F = l == 0 ? c.P : c.width[l - 1];
  int off = 0;
  for (int i = 0; i < l; ++i) off += 2 * (i == 0 ? c.P : c.width[i - 1]);
  if (j < F) {
    double* run = c.running + off;
    double rm = run[j], rv = run[F + j];
    for (int g = 0; g < job.G; ++g) {
      rm = (1.0 - NIS_BN_MOM) * rm + NIS_BN_MOM * c.gstat[(((size_t)l * job.G + g) * 2 + 0) * c.WM + j];
      rv = (1.0 - NIS_BN_MOM) * rv + NIS_BN_MOM * c.gstat[(((size_t)l * job.G + g) * 2 + 1) * c.WM + j];
    }
    run[j] = rm;
    run[F + j] = rv;
  }
  if (j == 0 && c.tracked) c.tracked[l] += job.G;
}

static int stream_grid(int64_t N) { return (int)std::max<int64_t>(1, std::min<int64_t>(NIS_MAXB, (N + NIS_TPB - 1) / NIS_TPB)); }

__global__ __launch_bounds__(64) void k_merge_moments(const double* __restrict__ recs, int world, int steps, double* __restrict__ out) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= steps) return;
  double n = 0.0, s1 = 0.0;
  for (int r = 0; r < world; ++r) {
    const double* q = recs + ((size_t)r * steps + i) * 4;
    if (q[0] > 0.0) {
      n += q[0];
      s1 = fma(q[0], q[1], s1);
    }
  }
  const double mean = n > 0.0 ? s1 / n : 0.0;
  double m2 = 0.0, mx = -__builtin_huge_val();
  for (int r = 0; r < world; ++r) {
    const double* q = recs + ((size_t)r * steps + i) * 4;
    if (q[0] > 0.0) {
      const double d = q[1] - mean;
      m2 += fma(q[0] * d, d, q[2]);
      mx = fmax(mx, q[3]);
    }
  }
  out[i * 4 + 0] = n;
  out[i * 4 + 1] = mean;
  out[i * 4 + 2] = m2;
  out[i * 4 + 3] = mx;
}

extern "C" int nis_merge_moments(const double* recs, int32_t world, int32_t steps, double* out, void* stream) {
  if (!recs || !out || world < 1 || steps < 1) return NIS_E_ARG;
  hipLaunchKernelGGL(k_merge_moments, dim3((steps + 63) / 64), dim3(64), 0, (hipStream_t)stream, recs, (int)world, (int)steps, out);
  HIP_OK(hipGetLastError());
  return NIS_OK;
}

extern "C" size_t nis_moments_workspace_bytes(void) { return (size_t)NIS_MAXB * 4 * sizeof(double); }

extern "C" int nis_moments(const double* a, const double* b, int64_t N, double* out4, void* workspace, size_t workspace_bytes,
                           void* stream) {
  if (!a || !out4 || N < 1) return NIS_E_ARG;
  if (!workspace || workspace_bytes < nis_moments_workspace_bytes()) return NIS_E_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  const int grid = stream_grid(N);
  hipLaunchKernelGGL(k_moments, dim3(grid), dim3(NIS_TPB), 0, s, a, b, (long)N, (double*)workspace);
  HIP_OK(hipGetLastError());
  hipLaunchKernelGGL(k_moments_final, dim3(1), dim3(NIS_TPB), 0, s, (const double*)workspace, grid, out4, 0);
  HIP_OK(hipGetLastError());
  return NIS_OK;
}

extern "C" int nis_integrand_moments(const double* planes, int64_t ld, int64_t N, int32_t d, const int32_t* out_col_host,
                                     int32_t f_kind, const double* f_params_host, double* fres_out, double* out4, void* workspace,
                                     size_t workspace_bytes, void* stream) {
  if (!planes || !out_col_host || !out4 || N < 1 || d < 1 || d > NIS_MAX_DIM || ld < N) return NIS_E_ARG;
  if (f_kind == NIS_F_CAMEL && d < 2) return NIS_E_ARG;
  if (!workspace || workspace_bytes < nis_moments_workspace_bytes()) return NIS_E_WORKSPACE;
  IntegrandArgs a;
  memset(&a, 0, sizeof(a));
  for (int c = 0; c < d; ++c) a.out_col[c] = out_col_host[c];
  a.d = d;
  a.kind = f_kind;
  a.jplane = d;
  for (int k = 0; k < 4; ++k) a.p[k] = f_params_host ? f_params_host[k] : 0.0;
  hipStream_t s = (hipStream_t)stream;
  const int grid = stream_grid(N);
  {
    NisProfScope ps(NIS_K_INTEGRAND, s);
    hipLaunchKernelGGL(k_integrand_moments, dim3(grid), dim3(NIS_TPB), 0, s, planes, (long)ld, (long)N, a, fres_out,
                       (double*)workspace);
  }
  HIP_OK(hipGetLastError());
  hipLaunchKernelGGL(k_moments_final, dim3(1), dim3(NIS_TPB), 0, s, (const double*)workspace, grid, out4, 0);
  HIP_OK(hipGetLastError());
  return NIS_OK;
}

extern "C" int nis_integrand_eval(int32_t f_kind, const double* f_params_host, const double* x, int64_t row_stride, int64_t N,
                                  int32_t d, double* out, void* stream) {
  if (!x || !out || N < 1 || d < 1 || d > NIS_MAX_DIM || row_stride < d) return NIS_E_ARG;
  if (f_kind == NIS_F_CAMEL && d < 2) return NIS_E_ARG;
  if (f_kind != NIS_F_CAMEL && f_kind != NIS_F_NORMAL && f_kind != NIS_F_ONE) return NIS_E_ARG;
  IntegrandArgs a;
  memset(&a, 0, sizeof(a));
  for (int c = 0; c < d; ++c) a.out_col[c] = c;
  a.d = d;
  a.kind = f_kind;
  for (int k = 0; k < 4; ++k) a.p[k] = f_params_host ? f_params_host[k] : 0.0;
  hipLaunchKernelGGL(k_integrand_rows, dim3(stream_grid(N)), dim3(NIS_TPB), 0, (hipStream_t)stream, x, (long)row_stride, (long)N, a, out);
  HIP_OK(hipGetLastError());
  return NIS_OK;
}

// ---------------------------------------------------------------------------------------------
static int device_cus() {
  static int n = -1;
  if (n < 0) {
    int dev = 0;
    hipDeviceProp_t pr;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&pr, dev) != hipSuccess) return 256;
    n = pr.multiProcessorCount > 0 ? pr.multiProcessorCount : 256;
  }
  return n;
}

// x extent of a launch over G batches: blocks_per_cu * CUs workgroups shared by the batches (rounded DOWN: one
// workgroup more than a multiple of the CU count costs a whole extra round), never more workgroups than the batch has
// chunks of `tpb` samples, never more than NIS_MAXB records.
static int grid_for(int blocks_per_cu, int G, int64_t N, int tpb) {
  const int64_t resident = (int64_t)blocks_per_cu * device_cus();
  int64_t gx = resident / G;
  gx = std::min<int64_t>(gx, (N + tpb - 1) / tpb);
  gx = std::min<int64_t>(gx, NIS_MAXB);
  return (int)std::max<int64_t>(1, gx);
}

static int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return (v && *v) ? atoi(v) : dflt;
}

// exact common width of the hidden layers when a specialised instantiation exists (WM = 4: 3-wide nets), else 0
static int cell_wx(const nis_cell_desc& cd, int WM) {
  if (WM != 4 || cd.P > 3 || env_int("NIS_NO_WX", 0)) return 0;
  for (int l = 0; l < cd.n_hidden; ++l)
    if (cd.width[l] > 3) return 0;
  return 3;
}

template <int WM>
static void cell_grids(int kind, int km, int wx, bool folded, int G, int64_t N, int* gs, int* ga) {
  static const int bs = env_int("NIS_STATS_BPC", 0), ba = env_int("NIS_APPLY_BPC", 0);   // tuning overrides
  *gs = grid_for(bs > 0 ? bs : stats_blocks_per_cu<WM>(wx, folded), G, N, nis_stats_tpb(WM));
  *ga = grid_for(ba > 0 ? ba : apply_blocks_per_cu<WM>(kind, km, wx), G, N, NIS_TPB);
}

struct RunOpts {
  int G = 1;                       // batched iterations (w: (G,N,d), planes: (G,d+1,ld), out4: (G,4))
  int w_has_jac = 0;
  int bn_mode = NIS_BN_TRAIN;
  int64_t pack_token = 0;          // 0: always repack
  long long* const* tracked = nullptr;
  double* xj_out = nullptr;
  int32_t* bins_out = nullptr;
  int f_kind = -1;                 // >= 0: built-in integrand, moments -> out4
  const double* f_params = nullptr;
  double* fres_out = nullptr;
  double* out4 = nullptr;
  int keep_planes = 1;             // 0 with f_kind >= 0: X / J are not stored
  int hist = 0;                    // 1: `planes` holds n_cells+1 plane sets; cell c reads set c and writes set c+1
  double* bnsave = nullptr;        // hist: per cell (L+1) x 4 x WM saved BN scale | shift | mean | rstd
  // multi-rank exact BatchNorm (nis_flow_forward_sync): after every record-producing launch the rank's reduced record goes
  // to sync_buf[rank] and `sync_fn` gathers all ranks' records into sync_buf (world x 64 x 3 doubles)
  nis_sync_fn sync_fn = nullptr;
  void* sync_ctx = nullptr;
  double* sync_buf = nullptr;
  int sync_world = 1, sync_rank = 0;
  // in-kernel draws (nis_integrate_steps_rng): w == nullptr, the persistent kernel generates batch g from Philox stream stream0 + g
  int rng = 0;
  unsigned long long seed = 0, stream0 = 0;
  float* w_out = nullptr;
  int spec_bins = 0;                         // test entry (nis_integrate_steps_bins): the SPEC kernels of the bench path write bins_out
  unsigned long long* rng_state = nullptr;   // device {seed, next stream}: read by the drawing launch, advanced by n_batches in the
                                             //   set's epilogue launch -- a captured launch set draws new numbers on every replay
};

// declared in flow_bwd.hip
bool nis_bwd_cell_supported(const nis_cell_desc& cd, int wm);
size_t nis_bwd_cell_workspace_doubles(const nis_cell_desc& cd, int wm, int64_t N);
int nis_bwd_cell_run(const CellK& c, int wm, const double* x, const double* xo, double* g, const double* bnsave, long ld, long N,
                     double* ws, double* gparams, const MfmaW& mw, hipStream_t s);

static bool use_mfma() {
  static const int off = env_int("NIS_NO_MFMA", 0);
  return !off;
}

// multi-rank exact BatchNorm: local records -> one record per feature in this rank's slot -> exchange hook
static int sync_records(const RunOpts& o, const double* records, int nparts, int WM, hipStream_t s) {
  if (!o.sync_fn) return NIS_OK;
  hipLaunchKernelGGL(k_sync_reduce, dim3(1), dim3(NIS_TPB), 0, s, records, nparts, WM, o.sync_buf + (size_t)o.sync_rank * WM * 3);
  HIP_OK(hipGetLastError());
  o.sync_fn(o.sync_ctx, o.sync_buf, WM * 3, (void*)s);
  return NIS_OK;
}

// Side streams of the multi-chain mode (flow_run): created once per device, non-blocking; fork / join events without timing.
struct SidePool {
  std::vector<hipStream_t> stream;
  std::vector<hipEvent_t> join, stag;
  hipEvent_t fork = nullptr;
};
static SidePool* side_pool(int n) {
  static SidePool pools[16];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  SidePool& p = pools[dev];
  if (!p.fork && hipEventCreateWithFlags(&p.fork, hipEventDisableTiming) != hipSuccess) return nullptr;
  while ((int)p.stream.size() < n) {
    hipStream_t st;
    hipEvent_t ev, ev2;
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) return nullptr;
    if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return nullptr;
    if (hipEventCreateWithFlags(&ev2, hipEventDisableTiming) != hipSuccess) return nullptr;
    p.stream.push_back(st);
    p.join.push_back(ev);
    p.stag.push_back(ev2);
  }
  return &p;
}

template <int WM>
static int run_cell(const CellK& c, const MfmaW& mw, bool mfma, double* planes, long ld, long N, int bn_mode, int km, int* bins,
                    const FuseF& fz, bool need_stats0, int grid_stats, int grid_apply, int grid_mf, int grid_mf_stats, hipStream_t s,
                    const RunOpts& o, hipEvent_t pre_apply = nullptr) {
  if (bn_mode == NIS_BN_TRAIN) {
    if (need_stats0) {
      NisProfScope ps(NIS_K_STATS, s);
      HIP_OK(launch_stats<WM>(c, planes, ld, N, 0, c.dcov ? c.nparts[0] : grid_stats, s));
      if (int rc = sync_records(o, c.partials, c.nparts[0], WM, s)) return rc;
    }
    if (!c.pmerge) HIP_OK(launch_finalize<WM>(c, 0, s));
    for (int l = 1; l <= c.L; ++l) {
      if (l == 1 && c.derive1) continue;                       // BN_1 was derived from BN_0 (single pass-through column)
      {
        NisProfScope ps(NIS_K_STATS, s);
        if (mfma)
          HIP_OK(launch_stats_mfma<WM>(c, mw, planes, ld, N, l, grid_mf_stats, s));
        else
          HIP_OK(launch_stats<WM>(c, planes, ld, N, l, grid_stats, s));
        if (int rc = sync_records(o, c.partials + (size_t)l * c.G * c.maxb * WM * 3, c.nparts[l], WM, s)) return rc;
      }
      if (!c.pmerge) HIP_OK(launch_finalize<WM>(c, l, s));
    }
  }
  if (pre_apply) HIP_OK(hipEventRecord(pre_apply, s));      // the next chain of the set starts here (flow_run: staggered chains)
  {
    NisProfScope ps(NIS_K_APPLY, s);
    if (mfma)
      HIP_OK(launch_apply_mfma<WM>(c, mw, planes, ld, N, bn_mode == NIS_BN_TRAIN ? 1 : 0, km, bins, fz, grid_mf, s));
    else
      HIP_OK(launch_apply<WM>(c, planes, ld, N, bn_mode == NIS_BN_TRAIN ? 1 : 0, km, bins, fz, grid_apply, s));
  }
  return NIS_OK;                      // G > 1: the running-statistics updates are applied by the launch set's k_group_epilogue
}

template <int WM>
static void mfma_plan(const CellK& c, int G, int64_t N, bool* ok, int* grid, int* grid_stats) {
  *ok = WM >= 16 && use_mfma() && mfma_supported<WM>(c);
  // one wave owns 2 tiles of 16 samples per pass; 4 waves per workgroup
  const int per_wg = 4 * 16 * 2;
  static const int bpc = env_int("NIS_MFMA_BPC", 0);          // tuning override
  *grid = *ok ? grid_for(bpc > 0 ? bpc : mfma_blocks_per_cu<WM>(c), G, N, per_wg) : 1;
  // the statistics kernels run on the apply kernel's grid.  They hold no LDS image and ~100 VGPRs, but more workgroups per CU
  // (NIS_MFMA_STATS_BPC = 4, 6, 8) measured 0 ... 8 % SLOWER (C3 13.9 -> 14.2 ms, C5 7.6 -> 8.2 ms per iteration): every
  // workgroup's prologue merges the previous layer's records, one per workgroup
  static const int bpc_s = env_int("NIS_MFMA_STATS_BPC", 0);
  *grid_stats = *ok ? (bpc_s > 0 ? std::max(*grid, grid_for(bpc_s, G, N, per_wg)) : *grid) : 1;
}

// Descriptor-level mirror of launch_apply's SPEC conditions (flow_wm.hip): every cell of a d = 2 PWQuad chain with 3-wide
// conditioners and a compiled bin count, camel integrand fused into the last cell, nothing materialised.
static bool direct_rows_ok(const nis_flow* f, const std::vector<CellLayout>& lay, const RunOpts& o, int w_dtype, int wcols, int ingest_P) {
  if (f->d != 2 || wcols != 2 || w_dtype != NIS_F32 || o.w_has_jac || o.hist || o.keep_planes || o.xj_out || (o.bins_out && !o.spec_bins) ||
      o.fres_out || o.sync_fn || o.f_kind != NIS_F_CAMEL || !o.out4)
    return false;
  if (o.bn_mode != NIS_BN_TRAIN) return false;     // eval mode: inner cells carry no BN_0 rider and run the generic kernel
  if (env_int("NIS_NO_FOLD", 0) || env_int("NIS_NO_WX", 0) || env_int("NIS_NO_DERIVE1", 0) || getenv("NIS_NO_STATS_FOLD")) return false;
  if (ingest_P != 1) return false;
  bool in_w[2] = {true, true};
  for (int ci = 0; ci < f->n_cells; ++ci) {
    const nis_cell_desc& cd = f->cells[ci];
    if (cd.kind != NIS_PWQUAD || cd.P != 1 || lay[ci].WM != 4 || cell_wx(cd, 4) != 3) return false;
    if (!(lay[ci].KM == -4 || lay[ci].KM == -8 || lay[ci].KM == -16)) return false;
    if (cd.n_hidden != 3) return false;                                 // SPEC apply kernels: three hidden layers, unrolled
    if (cd.in_col[0] < 0 || cd.in_col[0] > 1 || cd.in_col[1] < 0 || cd.in_col[1] > 1) return false;
    if (in_w[cd.in_col[0]] && !in_w[cd.in_col[1]]) return false;        // the row kernels take xb (and optionally x) from w
    if (ci + 1 < f->n_cells && f->cells[ci + 1].in_col[0] != cd.in_col[1]) return false;   // SPEC 1: the next cell passes this output through
    if (ci + 1 == f->n_cells && !((f->out_col[0] == 0 && f->out_col[1] == 1) || (f->out_col[0] == 1 && f->out_col[1] == 0))) return false;
    in_w[cd.in_col[1]] = false;
  }
  return true;
}

static int flow_run(nis_flow* f, const void* w, int32_t w_dtype, int64_t N, const double* const* params, double* const* running,
                    double* planes, int64_t ld, void* workspace, size_t workspace_bytes, hipStream_t s, const RunOpts& o) {
  if (!f || (!w && !o.rng) || !params || !running || N < 1 || (planes && ld < N)) return NIS_E_ARG;
  if (o.bn_mode != NIS_BN_EVAL && o.bn_mode != NIS_BN_TRAIN) return NIS_E_ARG;
  if (w_dtype != NIS_F64 && w_dtype != NIS_F32) return NIS_E_ARG;
  const int G = o.G;
  if (G < 1 || G > 65535) return NIS_E_ARG;
  if (G > 1 && (o.xj_out || o.bins_out || o.w_has_jac || o.hist)) return NIS_E_UNSUPPORTED;
  if (!workspace || workspace_bytes < nis_flow_workspace_bytes(f, G)) return NIS_E_WORKSPACE;
  std::vector<CellLayout> lay;
  size_t off_fpart = 0, off_mega = 0;
  flow_layout(f, G, &lay, &off_fpart, &off_mega);
  int* status = (int*)workspace;
  double* body = (double*)((char*)workspace + WS_HEADER);
  const int d = f->d;
  const bool train = o.bn_mode == NIS_BN_TRAIN;
  const size_t tile_bytes = ((size_t)NIS_TPB * (d + 1) + 16) * sizeof(double);
  const int wcols = o.w_has_jac ? d + 1 : d;
  const long plane_gs = (long)(d + 1) * ld;

  // is the padded parameter image in this workspace still valid?
  bool repack = o.pack_token == 0 || o.pack_token != f->packed_token || o.bn_mode != f->packed_mode || workspace != f->packed_ws ||
                G != f->packed_G;
  for (int ci = 0; ci < f->n_cells && !repack; ++ci) repack = params[ci] != f->packed_params[ci] || running[ci] != f->packed_running[ci];

  // ---- ingest (+ BN_0 records of the first cell)
  const nis_cell_desc& c0 = f->cells[0];
  IngestStats ist;
  memset(&ist, 0, sizeof(ist));
  ist.status = status;
  ist.maxb = NIS_MAXB;
  ist.plane_gs = plane_gs;
  // a matrix-core first cell with 2..8 pass-through columns takes its BN_0 records from the covariance pass (CellK::dcov: BN_1 is
  // derived, the layer-1 statistics launch disappears) instead of the ingest rider, which has no co-moments
  const bool cell0_cov = train && lay[0].WM >= 16 && use_mfma() && c0.P >= 2 && c0.P <= 8 && c0.n_hidden >= 2 &&
                         3 * c0.P + c0.P * (c0.P - 1) / 2 <= 3 * lay[0].WM && !o.sync_fn && !env_int("NIS_NO_DERIVE1", 0) &&
                         !env_int("NIS_NO_DCOV", 0);
  const bool ingest_stats = train && c0.P <= NIS_PMAX_FUSED && !cell0_cov;
  if (ingest_stats) {
    ist.P = c0.P;
    for (int p = 0; p < c0.P; ++p) ist.sel[p] = c0.in_col[p];
    ist.wm = lay[0].WM;
    ist.partials = body + lay[0].off_part;
  }
  // Direct row input: a d = 2 chain of SPEC kernels with the camel integrand fused never needs X as planes, so the
  // (N, 2) float32 rows of w are read by the cells themselves (a column lives in w until a cell overwrites it) and the
  // ingest launch only gathers BN_0's records: 80 MB read instead of 80 MB read + 160 MB written per 1e7 samples.
  const bool direct = direct_rows_ok(f, lay, o, w_dtype, wcols, ist.P) && !env_int("NIS_NO_DIRECT_ROWS", 0);
  // ---- parameter image (before anything forks): zero-padded weights (+ MFMA fragments, + eval-mode BN affine)
  auto cell_shape = [&](int ci) {
    const nis_cell_desc& cd = f->cells[ci];
    CellK c;
    memset(&c, 0, sizeof(c));
    c.kind = cd.kind;
    c.d = d;
    c.P = cd.P;
    c.T = d - cd.P;
    c.K = cd.K;
    c.L = cd.n_hidden;
    c.out = cell_out(cd);
    c.R = cell_R(cd);
    for (int l = 0; l < cd.n_hidden; ++l) c.width[l] = cd.width[l];
    for (int j = 0; j < d; ++j) c.in_plane[j] = cd.in_col[j];
    c.jplane = d;
    c.coef = body + lay[ci].off_coef;
    return c;
  };
  for (int ci = 0; ci < f->n_cells; ++ci)
    if (!params[ci] || !running[ci]) return NIS_E_ARG;
  if (repack) {
    for (int ci = 0; ci < f->n_cells; ++ci) {
      const nis_cell_desc& cd = f->cells[ci];
      const CellLayout& L = lay[ci];
      const CellK c = cell_shape(ci);
      const int pgrid = (int)std::min<size_t>(64, ((size_t)c.R * L.WM + NIS_TPB - 1) / NIS_TPB);
      NisProfScope ps(NIS_K_PACK, s);
      hipLaunchKernelGGL(k_pack, dim3(std::max(1, pgrid)), dim3(NIS_TPB), 0, s, c, L.WM, params[ci], (const double*)running[ci],
                         train ? 0 : 1, body + L.off_wp, body + L.off_wlast, body + L.off_bias, body + L.off_gb, body + L.off_hb,
                         cell_hidden_bias(cd) ? 1 : 0);
      HIP_OK(hipGetLastError());
      if (L.WM >= 16 && use_mfma()) {
        hipLaunchKernelGGL(k_pack_mfma, dim3(std::max(1, pgrid)), dim3(NIS_TPB), 0, s, L.WM, c.L, c.R, L.RT,
                           (const double*)(body + L.off_wp), (const double*)(body + L.off_wlast), (const double*)(body + L.off_bias),
                           body + L.off_wm, body + L.off_wmlast, body + L.off_biaspad, cd.kind == NIS_AFFINE ? d - cd.P : 0);
        HIP_OK(hipGetLastError());
      }
    }
  }
  auto packed_done = [&]() {
    f->packed_token = o.pack_token;
    f->packed_mode = o.bn_mode;
    f->packed_ws = workspace;
    f->packed_G = G;
    for (int ci = 0; ci < f->n_cells; ++ci) {
      f->packed_params[ci] = params[ci];
      f->packed_running[ci] = running[ci];
    }
  };
  // G > 1, train mode: ONE launch at the end applies the running-statistics updates of all batches in batch order and
  // merges the fused integrand's moment records
  EpiJob epi;
  memset(&epi, 0, sizeof(epi));
  epi.G = G;
  auto epi_add = [&](int ci) {
    const nis_cell_desc& cd = f->cells[ci];
    EpiCell& e = epi.cell[epi.ncell];
    e.running = running[ci];
    e.gstat = body + lay[ci].off_gstat;
    e.tracked = (train && o.tracked) ? o.tracked[ci] : nullptr;
    e.L = cd.n_hidden;
    e.WM = lay[ci].WM;
    e.P = cd.P;
    for (int l = 0; l < NIS_MAX_HIDDEN; ++l) e.width[l] = l < cd.n_hidden ? cd.width[l] : 0;
    epi.blk_off[epi.ncell + 1] = epi.blk_off[epi.ncell] + cd.n_hidden + 1;
    ++epi.ncell;
  };
  auto epi_flush = [&](const double* fpart, int fparts) -> int {
    if (epi.ncell == 0 && !fpart) return NIS_OK;
    epi.fpart = fpart;
    epi.fparts = fparts;
    epi.out4 = o.out4;
    epi.maxb = NIS_MAXB;
    epi.rng_state = (fpart && o.rng) ? o.rng_state : nullptr;
    NisProfScope ps(NIS_K_MOMENTS, s);
    hipLaunchKernelGGL(k_group_epilogue, dim3(epi.blk_off[epi.ncell] + (fpart ? G : 0)), dim3(NIS_TPB), 0, s, epi);
    HIP_OK(hipGetLastError());
    epi.ncell = 0;
    return NIS_OK;
  };

  // ---- persistent whole-step kernel (flow_mega.hip, opt-in: NIS_MEGA=1): the README / C1 / C2 model with everything the
  //      direct-row mode requires (float32 rows or in-kernel draws, train-mode BN, camel fused, nothing materialised),
  //      batch resident in LDS.  24-32 B/sample of HBM traffic instead of 88, but slower than the launch-per-phase path
  //      (one wave per SIMD and team issues fp64 at half rate; 6 hand-offs of ~9 us per batch).
  {
    int NT = 0, teams = 0;
    const bool mega = direct && mega_model(f) && env_int("NIS_MEGA", 0) && G <= NIS_MEGA_MAX_G && nis_mega_geometry(&NT, &teams) &&
                      ((N + NT - 1) / NT + 1) / 2 * 2 <= NIS_MEGA_CAP;
    if (mega) {
      MegaArgs a;
      memset(&a, 0, sizeof(a));
      for (int ci = 0; ci < 2; ++ci) {
        const CellLayout& L = lay[ci];
        a.cell[ci].wp = body + L.off_wp;
        a.cell[ci].wlast = body + L.off_wlast;
        a.cell[ci].bias = body + L.off_bias;
        a.cell[ci].gb = body + L.off_gb;
        a.cell[ci].gstat = body + L.off_gstat;
      }
      a.w = o.rng ? nullptr : (const float*)w;
      a.w_out = o.rng ? o.w_out : nullptr;
      a.N = (long)N;
      a.G = G;
      a.NT = NT;
      a.teams = std::min(teams, G);
      a.n_loc = (int)(((N + NT - 1) / NT + 1) / 2 * 2);
      a.col0 = f->cells[0].in_col[0];
      a.seed = o.seed;
      a.stream0 = o.stream0;
      a.rng_state = o.rng ? o.rng_state : nullptr;
      a.status = status;
      for (int k = 0; k < 3; ++k) a.fp[k] = o.f_params ? o.f_params[k] : 0.0;
      a.fp[3] = 1.0 / (a.fp[2] * a.fp[2]);
      a.fpart = body + off_fpart;
      a.maxb = NIS_MAXB;
      {
        NisProfScope ps(NIS_K_APPLY, s);
        HIP_OK(nis_mega_launch(a, body + off_mega, f->mega_ws != workspace, s));
      }
      f->mega_ws = workspace;
      epi_add(0);
      epi_add(1);
      if (int rc2 = epi_flush(body + off_fpart, NT)) return rc2;
      packed_done();
      return NIS_OK;
    }
  }
  if (!planes) return NIS_E_ARG;
  if (o.rng && !direct) return NIS_E_UNSUPPORTED;        // in-kernel draws exist for the direct-row chain (d = 2 SPEC kernels)
  // In-kernel draws on the launch-per-phase path: the record pass GENERATES the float32 rows (same Philox mapping as
  // nis_philox_rows) into the state plane the d = 2 direct-row chain never touches -- the plane of cell 0's pass-through
  // column, N float64 = N float32 pairs -- and the cells read them from there.
  const float* wrows = (const float*)w;
  long wrows_gs = 2 * (long)N;
  long wplane_ld = 0;                      // > 0: two float32 column planes instead of rows (K = 4 chains: PLN kernel instantiations)
  if (o.rng) {
    wrows = (const float*)(planes + (size_t)f->cells[0].in_col[0] * ld);
    wrows_gs = 2 * plane_gs;
    bool k4 = N % 2 == 0 && !env_int("NIS_NO_PLANAR", 0);          // even N: the second plane starts 8-byte aligned
    for (int ci = 0; ci < f->n_cells; ++ci) k4 = k4 && lay[ci].KM == -4;
    if (k4) wplane_ld = (long)N;
  }

  // ---- one chain of launches for batches [g0, g0 + Gy) on stream cs: ingest -> per cell [statistics ..., apply].
  //      Gg (>= Gy) sizes the grids: every chain of a call uses the same grids, so every batch has the same record counts.
  int epi_fparts = 0;
  bool epi_fused = false;
  auto chain = [&](hipStream_t cs, int g0, int Gy, int Gg, bool collect_epi, hipEvent_t stagger) -> int {
    IngestStats is2 = ist;
    is2.g0 = g0;
    // ingest / emit: 256-thread streaming workgroups, 8 per CU; the read-only record pass of the direct mode runs best at 4
    // (half the BN_0 records for the first statistics launch to merge: C2 bench 21.8e9 -> 22.2e9 samples/s)
    const int tgrid = grid_for(env_int("NIS_INGEST_BPC", direct ? 4 : 8), Gg, N, NIS_TPB);
    bool col_in_w[2] = {direct, direct};
    nis_prof_mark(NIS_K_INGEST, 1, cs);
    if (direct && o.rng)
      hipLaunchKernelGGL(k_ingest2_rng, dim3(tgrid, Gy), dim3(NIS_TPB), 0, cs, (unsigned long long)o.seed, (unsigned long long)o.stream0,
                         (const unsigned long long*)o.rng_state, (long)N, const_cast<float*>(wrows), wrows_gs, wplane_ld, o.w_out, is2);
    else if (direct)
      hipLaunchKernelGGL((k_ingest2<float, false>), dim3(tgrid, Gy), dim3(NIS_TPB), 0, cs, (const float*)w, (long)N, planes, (long)ld, is2);
    else if (wcols == 2 && is2.P <= 2 && w_dtype == NIS_F64)
      hipLaunchKernelGGL(k_ingest2<double>, dim3(tgrid, Gy), dim3(NIS_TPB), 0, cs, (const double*)w, (long)N, planes, (long)ld, is2);
    else if (wcols == 2 && is2.P <= 2)
      hipLaunchKernelGGL(k_ingest2<float>, dim3(tgrid, Gy), dim3(NIS_TPB), 0, cs, (const float*)w, (long)N, planes, (long)ld, is2);
    else if (w_dtype == NIS_F64)
      hipLaunchKernelGGL(k_ingest<double>, dim3(tgrid, Gy), dim3(NIS_TPB), tile_bytes, cs, (const double*)w, (long)N, wcols, planes,
                         (long)ld, is2);
    else
      hipLaunchKernelGGL(k_ingest<float>, dim3(tgrid, Gy), dim3(NIS_TPB), tile_bytes, cs, (const float*)w, (long)N, wcols, planes,
                         (long)ld, is2);
    nis_prof_mark(NIS_K_INGEST, 0, cs);
    HIP_OK(hipGetLastError());
    if (o.sync_fn) {
      if (G != 1 || !train || o.hist) return NIS_E_UNSUPPORTED;
      if (ingest_stats)
        if (int rc = sync_records(o, is2.partials, tgrid, lay[0].WM, cs)) return rc;
    }
    int prev_records = tgrid;           // records of BN_0 of the current cell written by the previous launch
    bool have_stats0 = ingest_stats;
    double* pl = planes;
    size_t bn_off = 0;
    for (int ci = 0; ci < f->n_cells; ++ci) {
      const nis_cell_desc& cd = f->cells[ci];
      const CellLayout& L = lay[ci];
      if (o.hist) pl = planes + (size_t)ci * plane_gs;
      int grid_stats = 1, grid_apply = 1;
      switch (L.WM) {
        // folded statistics launches (k_bn_stats_fold): train mode, one pass-through column, BN_1 derived
        case 4: cell_grids<4>(cd.kind, L.KM, cell_wx(cd, L.WM), train && cd.P == 1 && !env_int("NIS_NO_DERIVE1", 0) && !env_int("NIS_NO_FOLD", 0),
                              Gg, N, &grid_stats, &grid_apply); break;
        case 8: cell_grids<8>(cd.kind, L.KM, 0, false, Gg, N, &grid_stats, &grid_apply); break;
        case 16: cell_grids<16>(cd.kind, L.KM, 0, false, Gg, N, &grid_stats, &grid_apply); break;
        case 32: cell_grids<32>(cd.kind, L.KM, 0, false, Gg, N, &grid_stats, &grid_apply); break;
        default: cell_grids<64>(cd.kind, L.KM, 0, false, Gg, N, &grid_stats, &grid_apply); break;
      }
      CellK c = cell_shape(ci);
      c.first = (ci == 0 && !o.w_has_jac) ? 1 : 0;
      c.G = G;
      c.Gy = Gy;
      c.g0 = g0;
      bool mfma = false;
      int grid_mf = 1, grid_mf_stats = 1;
      switch (L.WM) {
        case 16: mfma_plan<16>(c, Gg, N, &mfma, &grid_mf, &grid_mf_stats); break;
        case 32: mfma_plan<32>(c, Gg, N, &mfma, &grid_mf, &grid_mf_stats); break;
        case 64: mfma_plan<64>(c, Gg, N, &mfma, &grid_mf, &grid_mf_stats); break;
        default: break;
      }
      c.nparts[0] = have_stats0 ? prev_records : grid_stats;
      for (int l = 1; l <= cd.n_hidden; ++l) c.nparts[l] = mfma ? grid_mf_stats : grid_stats;
      c.maxb = NIS_MAXB;
      c.bins_row = f->bins_row[ci];
      c.wp = body + L.off_wp;
      c.wlast = body + L.off_wlast;
      c.bias = body + L.off_bias;
      c.gb = body + L.off_gb;
      c.hb = body + L.off_hb;
      c.partials = body + L.off_part;
      c.gstat = body + L.off_gstat;
      c.running = running[ci];
      c.tracked = (train && o.tracked) ? o.tracked[ci] : nullptr;
      c.status = status;
      c.coef_gs = train ? (cd.n_hidden + 1) * 2 * L.WM : 0;     // eval: every batch shares the running-stat coefficients
      c.plane_gs = plane_gs;
      c.derive1 = (train && cd.P == 1 && !env_int("NIS_NO_DERIVE1", 0)) ? 1 : 0;
      // matrix-core cells with 2..8 pass-through columns: BN_0's records carry the co-moments of the block, BN_1 is derived from
      // them and the layer-1 statistics launch is skipped (not with the multi-rank record exchange: its records are per feature)
      c.dcov = (train && mfma && cd.P >= 2 && cd.P <= 8 && cd.n_hidden >= 2 && 3 * cd.P + cd.P * (cd.P - 1) / 2 <= 3 * L.WM && !o.sync_fn &&
                have_stats0 == false && !env_int("NIS_NO_DERIVE1", 0) && !env_int("NIS_NO_DCOV", 0)) ? 1 : 0;
      if (c.dcov) {
        c.derive1 = 1;
        // the covariance pass is a latency-bound stream (8 loads, 44 FMAs per sample, <= 128 VGPRs): two workgroups per CU (four: equal at G = 3, slower at G = 1 -- every consumer merges the records)
        static const int bpc_cov = env_int("NIS_COV_BPC", 2);
        c.nparts[0] = grid_for(bpc_cov, Gg, N, 256);
      }
      // single pass-through column on the fp64 VALU kernels: BN_0, Linear_0 and BN_1 collapse into one affine map of x
      // (needs the layer-1 statistics launch to be gone: train mode with derive1, or eval mode)
      c.fold = (cd.P == 1 && !mfma && (c.derive1 || !train) && !env_int("NIS_NO_FOLD", 0)) ? 1 : 0;
      c.wx = cell_wx(cd, L.WM);
      c.sync_part = o.sync_fn ? o.sync_buf : nullptr;
      c.sync_world = o.sync_world;
      c.pmerge = (G == 1 || !env_int("NIS_FINALIZE_KERNEL", 0)) ? 1 : 0;   // a launch costs >= 4.3 us, the prologue merge ~1 us
      if (direct) {
        c.wrows = wrows;
        c.wrows_gs = wrows_gs;
        c.wplanar = wplane_ld > 0 ? 1 : 0;
        c.wplane_ld = wplane_ld;
        c.x_wcol = col_in_w[cd.in_col[0]] ? cd.in_col[0] : -1;
        c.xb_wcol = col_in_w[cd.in_col[1]] ? cd.in_col[1] : -1;
        col_in_w[cd.in_col[1]] = false;            // an inner cell writes its transformed column to the plane; the last one stores nothing
      } else {
        c.x_wcol = c.xb_wcol = -1;
      }
      c.out_off = o.hist ? plane_gs : 0;
      c.bnsave = (o.hist && o.bnsave && train) ? o.bnsave + bn_off : nullptr;
      bn_off += (size_t)(cd.n_hidden + 1) * 4 * L.WM;
      // rider 1: BN_0 records of the next cell (every state column is in this thread's registers)
      bool next_rider = false;
      if (train && ci + 1 < f->n_cells && L.WM <= 8 && f->cells[ci + 1].P <= (L.WM <= 4 ? 4 : NIS_PMAX_FUSED)) {
        const nis_cell_desc& nx = f->cells[ci + 1];
        next_rider = true;
        c.next_P = nx.P;
        for (int p = 0; p < nx.P; ++p) {
          int src = -1;
          for (int j = 0; j < d; ++j)
            if (cd.in_col[j] == nx.in_col[p]) src = j;
          c.next_src[p] = src;
        }
        for (int p = nx.P; p < NIS_PMAX_FUSED; ++p) c.next_src[p] = -1;
        c.next_partials = body + lay[ci + 1].off_part;
        c.next_wm = lay[ci + 1].WM;
      }
      // rider 2: built-in integrand + moment records in the last cell
      FuseF fz;
      memset(&fz, 0, sizeof(fz));
      fz.kind = -1;
      fz.write_planes = 1;
      fz.maxb = NIS_MAXB;
      fz.fres_gs = N;
      const bool last = ci + 1 == f->n_cells;
      bool fused_f = false;
      if (last && o.f_kind >= 0 && (L.WM <= 8 || (mfma && nis_mfma_fusable(cd.kind, L.KM) && !o.fres_out))) {
        fused_f = true;
        fz.kind = o.f_kind;
        fz.loc0 = fz.loc1 = -1;
        for (int j = 0; j < d; ++j) {
          if (cd.in_col[j] == f->out_col[0]) fz.loc0 = j;
          if (d > 1 && cd.in_col[j] == f->out_col[1]) fz.loc1 = j;
        }
        for (int k = 0; k < 4; ++k) fz.p[k] = o.f_params ? o.f_params[k] : 0.0;
        if (o.f_kind == NIS_F_CAMEL) fz.p[3] = 1.0 / (fz.p[2] * fz.p[2]);
        fz.partials = body + off_fpart;
        fz.fres = o.fres_out;
        fz.write_planes = (o.keep_planes || o.xj_out) ? 1 : 0;
      }
      if (last && o.f_kind >= 0 && !fused_f && G > 1) return NIS_E_UNSUPPORTED;
      MfmaW mw;
      mw.wm_all = body + L.off_wm;
      mw.wm_last = body + L.off_wmlast;
      mw.bias_pad = body + L.off_biaspad;
      int rc;
      const bool need0 = !have_stats0;
      switch (L.WM) {
        case 4: rc = run_cell<4>(c, mw, false, pl, ld, N, o.bn_mode, L.KM, o.bins_out, fz, need0, grid_stats, grid_apply, 1, 1, cs, o, ci == 0 ? stagger : nullptr); break;
        case 8: rc = run_cell<8>(c, mw, false, pl, ld, N, o.bn_mode, L.KM, o.bins_out, fz, need0, grid_stats, grid_apply, 1, 1, cs, o, ci == 0 ? stagger : nullptr); break;
        case 16: rc = run_cell<16>(c, mw, mfma, pl, ld, N, o.bn_mode, L.KM, o.bins_out, fz, need0, grid_stats, grid_apply, grid_mf, grid_mf_stats, cs, o, ci == 0 ? stagger : nullptr); break;
        case 32: rc = run_cell<32>(c, mw, mfma, pl, ld, N, o.bn_mode, L.KM, o.bins_out, fz, need0, grid_stats, grid_apply, grid_mf, grid_mf_stats, cs, o, ci == 0 ? stagger : nullptr); break;
        case 64: rc = run_cell<64>(c, mw, mfma, pl, ld, N, o.bn_mode, L.KM, o.bins_out, fz, need0, grid_stats, grid_apply, grid_mf, grid_mf_stats, cs, o, ci == 0 ? stagger : nullptr); break;
        default: rc = NIS_E_UNSUPPORTED;
      }
      if (rc != NIS_OK) return rc;
      have_stats0 = next_rider;
      prev_records = grid_apply;
      if (next_rider)
        if (int rc2 = sync_records(o, c.next_partials, grid_apply, c.next_wm, cs)) return rc2;
      if (train && G > 1 && collect_epi) {
        epi_add(ci);
        if (epi.ncell == NIS_EPI_CELLS && !(last && fused_f))
          if (int rc2 = epi_flush(nullptr, 0)) return rc2;
      }
      if (last && o.f_kind >= 0) {
        if (fused_f) {
          epi_fparts = mfma ? grid_mf : grid_apply;
          epi_fused = true;
        } else {
          int rc2 = nis_integrand_moments(pl, ld, N, d, f->out_col, o.f_kind, o.f_params, o.fres_out, o.out4, body + off_fpart,
                                          (size_t)NIS_MAXB * 4 * sizeof(double), (void*)cs);
          if (rc2 != NIS_OK) return rc2;
        }
      }
    }
    return NIS_OK;
  };

  // Independent batches on several streams: one chain's HBM-bound statistics passes run beside another's fp64-issue-bound
  // apply kernels (C2 at G = 10: 33.3 -> 28.8 us per step).  The chains share nothing but read-only parameters (records,
  // coefficients, planes and moment records are indexed by batch); they fork from and join the caller's stream through
  // events, so the call stays capturable into a hipGraph, and the one epilogue launch behind the join applies the
  // running-statistics updates of ALL batches in batch order.
  int n_chains = 1;
  if (G >= 2 && !o.sync_fn && !o.hist && train && f->n_cells <= NIS_EPI_CELLS) n_chains = std::min(G, std::max(1, env_int("NIS_STREAMS", 1)));   // default 1: started together or staggered, the chains of ONE call gained nothing; free-running sets on separate streams do
  SidePool* pool = n_chains > 1 ? side_pool(n_chains - 1) : nullptr;
  if (n_chains > 1 && !pool) n_chains = 1;
  if (n_chains == 1) {
    if (int rc = chain(s, 0, G, G, true, nullptr)) return rc;
  } else {
    // Chains start STAGGERED: chain k + 1 waits until chain k reaches its first apply launch, so that the memory-bound
    // head of one chain (record pass, statistics) runs beside the fp64-bound apply kernel of its predecessor; started
    // together the chains would run the same kernel class in lockstep and overlap nothing.
    const int Gg = (G + n_chains - 1) / n_chains;
    static const bool stagger_on = env_int("NIS_NO_STAGGER", 0) == 0;
    HIP_OK(hipEventRecord(pool->fork, s));
    int g0 = 0;
    for (int k = 0; k < n_chains; ++k) {
      const int Gy = G / n_chains + (k < G % n_chains ? 1 : 0);
      hipStream_t cs = k == 0 ? s : pool->stream[k - 1];
      if (k > 0) HIP_OK(hipStreamWaitEvent(cs, (stagger_on ? pool->stag[k - 1] : pool->fork), 0));
      if (int rc = chain(cs, g0, Gy, Gg, k == 0, (stagger_on && k + 1 < n_chains) ? pool->stag[k] : nullptr)) return rc;
      if (k > 0) {
        HIP_OK(hipEventRecord(pool->join[k - 1], cs));
        HIP_OK(hipStreamWaitEvent(s, pool->join[k - 1], 0));
      }
      g0 += Gy;
    }
  }
  if (epi_fused) {
    if (int rc2 = epi_flush(body + off_fpart, epi_fparts)) return rc2;
  }
  if (int rc2 = epi_flush(nullptr, 0)) return rc2;
  packed_done();
  if (o.xj_out) {
    ColMap m;
    for (int j = 0; j < d; ++j) m.col[j] = f->out_col[j];
    m.col[d] = d;
    const int tgrid = grid_for(8, 1, N, NIS_TPB);
    double* pl = o.hist ? planes + (size_t)f->n_cells * plane_gs : planes;      // hist: the final state lives in the last plane set
    NisProfScope ps(NIS_K_EMIT, s);
    hipLaunchKernelGGL(k_emit, dim3(tgrid), dim3(NIS_TPB), tile_bytes, s, (const double*)pl, (long)ld, (long)N, d, m, o.xj_out);
    HIP_OK(hipGetLastError());
  }
  return NIS_OK;
}

extern "C" int nis_flow_forward(nis_flow* f, const void* w, int32_t w_dtype, int32_t w_has_jac, int64_t N, const double* const* params,
                                double* const* running, long long* const* tracked, int32_t bn_mode, int64_t pack_token, double* planes,
                                int64_t ld, double* xj_out, int32_t* bins_out, void* workspace, size_t workspace_bytes, void* stream) {
  RunOpts o;
  o.w_has_jac = w_has_jac;
  o.bn_mode = bn_mode;
  o.pack_token = pack_token;
  o.tracked = tracked;
  o.xj_out = xj_out;
  o.bins_out = bins_out;
  return flow_run(f, w, w_dtype, N, params, running, planes, ld, workspace, workspace_bytes, (hipStream_t)stream, o);
}

extern "C" int nis_flow_forward_sync(nis_flow* f, const void* w, int32_t w_dtype, int32_t w_has_jac, int64_t N,
                                     const double* const* params, double* const* running, long long* const* tracked,
                                     int64_t pack_token, double* planes, int64_t ld, double* xj_out, void* workspace,
                                     size_t workspace_bytes, void* stream, nis_sync_fn sync_fn, void* sync_ctx, double* sync_buf,
                                     int32_t world, int32_t rank) {
  if (!sync_fn || !sync_buf || world < 1 || rank < 0 || rank >= world) return NIS_E_ARG;
  RunOpts o;
  o.w_has_jac = w_has_jac;
  o.bn_mode = NIS_BN_TRAIN;
  o.pack_token = pack_token;
  o.tracked = tracked;
  o.xj_out = xj_out;
  o.sync_fn = sync_fn;
  o.sync_ctx = sync_ctx;
  o.sync_buf = sync_buf;
  o.sync_world = world;
  o.sync_rank = rank;
  return flow_run(f, w, w_dtype, N, params, running, planes, ld, workspace, workspace_bytes, (hipStream_t)stream, o);
}

extern "C" int nis_integrate_steps(nis_flow* f, const void* w, int32_t w_dtype, int32_t n_batches, int64_t N,
                                   const double* const* params, double* const* running, long long* const* tracked, int32_t bn_mode,
                                   int64_t pack_token, int32_t f_kind, const double* f_params_host, double* planes, int64_t ld,
                                   int32_t keep_planes, double* fres_out, double* out4, void* workspace, size_t workspace_bytes,
                                   void* stream) {
  if (!out4 || f_kind < 0) return NIS_E_ARG;
  if (f_kind == NIS_F_CAMEL && f && f->d < 2) return NIS_E_ARG;
  RunOpts o;
  o.G = n_batches;
  o.bn_mode = bn_mode;
  o.pack_token = pack_token;
  o.tracked = tracked;
  o.f_kind = f_kind;
  o.f_params = f_params_host;
  o.fres_out = fres_out;
  o.out4 = out4;
  o.keep_planes = keep_planes;
  return flow_run(f, w, w_dtype, N, params, running, planes, ld, workspace, workspace_bytes, (hipStream_t)stream, o);
}

extern "C" int nis_integrate_steps_rng(nis_flow* f, int32_t n_batches, int64_t N, const double* const* params, double* const* running,
                                       long long* const* tracked, int32_t bn_mode, int64_t pack_token, int32_t f_kind,
                                       const double* f_params_host, uint64_t seed, uint64_t stream0, uint64_t* rng_state, float* w_out,
                                       double* planes, int64_t ld, double* out4, void* workspace, size_t workspace_bytes, void* stream) {
  if (!out4 || f_kind < 0) return NIS_E_ARG;
  RunOpts o;
  o.G = n_batches;
  o.bn_mode = bn_mode;
  o.pack_token = pack_token;
  o.tracked = tracked;
  o.f_kind = f_kind;
  o.f_params = f_params_host;
  o.out4 = out4;
  o.keep_planes = 0;
  o.rng = 1;
  o.seed = seed;
  o.stream0 = stream0;
  o.rng_state = (unsigned long long*)rng_state;
  o.w_out = w_out;
  return flow_run(f, nullptr, NIS_F32, N, params, running, planes, planes ? ld : N, workspace, workspace_bytes, (hipStream_t)stream, o);
}

// Test entry: ONE iteration of the fused integrate step on the production (direct-row SPEC) code path that additionally writes
// the selected bin index of every cell's transformed column: bins_out int32 [n_cells][N] (tests/test_gpu_parity.py asserts them
// bit-exact against the oracle at the bench's full batch size).  Models outside the SPEC class: NIS_E_UNSUPPORTED.
extern "C" int nis_integrate_steps_bins(nis_flow* f, const void* w, int32_t w_dtype, int64_t N, const double* const* params,
                                        double* const* running, long long* const* tracked, int64_t pack_token, int32_t f_kind,
                                        const double* f_params_host, double* planes, int64_t ld, int32_t* bins_out, double* out4,
                                        void* workspace, size_t workspace_bytes, void* stream) {
  if (!out4 || !bins_out || f_kind < 0 || !f) return NIS_E_ARG;
  for (int ci = 0; ci < f->n_cells; ++ci)
    if (f->cells[ci].kind != NIS_PWQUAD || f->cells[ci].K != 4) return NIS_E_UNSUPPORTED;
  RunOpts o;
  o.G = 1;
  o.bn_mode = NIS_BN_TRAIN;
  o.pack_token = pack_token;
  o.tracked = tracked;
  o.f_kind = f_kind;
  o.f_params = f_params_host;
  o.out4 = out4;
  o.keep_planes = 0;
  o.bins_out = bins_out;
  o.spec_bins = 1;
  std::vector<CellLayout> lay;
  flow_layout(f, 1, &lay, nullptr);
  if (!direct_rows_ok(f, lay, o, w_dtype, f->d, f->cells[0].P) || env_int("NIS_NO_DIRECT_ROWS", 0) || env_int("NIS_MEGA", 0)) return NIS_E_UNSUPPORTED;
  return flow_run(f, w, w_dtype, N, params, running, planes, ld, workspace, workspace_bytes, (hipStream_t)stream, o);
}

// 1 when nis_integrate_steps can fuse a built-in integrand into the last cell's kernel (narrow conditioners: the fp64 VALU
// apply kernel's rider; wide ones: the FUSE instantiations of the matrix-core apply kernel)
extern "C" int nis_flow_fusable(const nis_flow* f) {
  if (!f) return 0;
  std::vector<CellLayout> lay;
  flow_layout(f, 1, &lay, nullptr);
  const nis_cell_desc& cd = f->cells[f->n_cells - 1];
  const CellLayout& L = lay[f->n_cells - 1];
  if (L.WM <= 8) return 1;
  if (!use_mfma() || !nis_mfma_fusable(cd.kind, L.KM)) return 0;
  CellK c;
  memset(&c, 0, sizeof(c));
  c.kind = cd.kind;
  c.K = cd.K;
  c.out = cell_out(cd);
  bool ok = false;
  switch (L.WM) {
    case 16: ok = mfma_supported<16>(c); break;
    case 32: ok = mfma_supported<32>(c); break;
    case 64: ok = mfma_supported<64>(c); break;
    default: break;
  }
  return ok ? 1 : 0;
}

extern "C" int nis_integrate_resident(const nis_flow* f, int32_t n_batches, int64_t N, int32_t bn_mode, int32_t f_kind) {
  int NT = 0, teams = 0;
  return f && mega_model(f) && bn_mode == NIS_BN_TRAIN && f_kind == NIS_F_CAMEL && n_batches >= 1 && n_batches <= NIS_MEGA_MAX_G &&
                 env_int("NIS_MEGA", 0) && !env_int("NIS_NO_DIRECT_ROWS", 0) && !env_int("NIS_NO_FOLD", 0) && !env_int("NIS_NO_WX", 0) &&
                 !env_int("NIS_NO_DERIVE1", 0) && !getenv("NIS_NO_STATS_FOLD") && nis_mega_geometry(&NT, &teams) &&
                 ((N + NT - 1) / NT + 1) / 2 * 2 <= NIS_MEGA_CAP
             ? 1
             : 0;
}

extern "C" int nis_philox_rows(uint64_t seed, uint64_t stream0, int32_t n_batches, int64_t N, float* out, void* stream) {
  if (!out || n_batches < 1 || n_batches > 65535 || N < 1) return NIS_E_ARG;
  HIP_OK(nis_philox_rows_launch(seed, stream0, n_batches, (long)N, out, (hipStream_t)stream));
  return NIS_OK;
}

// ---------------------------------------------------------------------------------------------
// training forward with history + hand-written adjoint
static int cell_wm(const nis_cell_desc& c) {
  int m = c.P;
  for (int l = 0; l < c.n_hidden; ++l) m = std::max(m, c.width[l]);
  return nis_pick_wm(m);
}

extern "C" int nis_flow_backward_supported(const nis_flow* f) {
  if (!f) return 0;
  for (int ci = 0; ci < f->n_cells; ++ci)
    if (!nis_bwd_cell_supported(f->cells[ci], cell_wm(f->cells[ci]))) return 0;
  return 1;
}

extern "C" size_t nis_flow_bnsave_count(const nis_flow* f) {
  size_t n = 0;
  for (int ci = 0; ci < f->n_cells; ++ci) n += (size_t)(f->cells[ci].n_hidden + 1) * 4 * cell_wm(f->cells[ci]);
  return n;
}

extern "C" int nis_flow_forward_train(nis_flow* f, const void* w, int32_t w_dtype, int32_t w_has_jac, int64_t N,
                                      const double* const* params, double* const* running, long long* const* tracked,
                                      int64_t pack_token, double* planes_hist, int64_t ld, double* bnsave, double* xj_out,
                                      void* workspace, size_t workspace_bytes, void* stream) {
  if (!planes_hist || !bnsave) return NIS_E_ARG;
  RunOpts o;
  o.w_has_jac = w_has_jac;
  o.bn_mode = NIS_BN_TRAIN;
  o.pack_token = pack_token;
  o.tracked = tracked;
  o.xj_out = xj_out;
  o.hist = 1;
  o.bnsave = bnsave;
  return flow_run(f, w, w_dtype, N, params, running, planes_hist, ld, workspace, workspace_bytes, (hipStream_t)stream, o);
}

// doubles of the packed parameter image a cell's adjoint works from (row-major images + MFMA fragments when wide)
static size_t bwd_pack_doubles(const nis_cell_desc& cd, size_t WM) {
  const size_t R = cell_R(cd), nl = cd.n_hidden, RT = (R + 15) / 16;
  size_t n = nl * WM * WM + R * WM + (R + 7) / 8 * 8 + 2 * (nl + 1) * 2 * WM;
  if (WM >= 16) n += nl * WM * WM + RT * 16 * WM + RT * 16;
  return n;
}

extern "C" size_t nis_flow_backward_workspace_bytes(const nis_flow* f, int64_t N) {
  size_t cellws = 0, pk = 0;
  std::vector<CellLayout> lay;
  flow_layout(f, 1, &lay, nullptr);
  for (int ci = 0; ci < f->n_cells; ++ci) {
    cellws = std::max(cellws, nis_bwd_cell_workspace_doubles(f->cells[ci], lay[ci].WM, N));
    pk = std::max(pk, bwd_pack_doubles(f->cells[ci], lay[ci].WM));
  }
  return ((size_t)(f->d + 1) * N + cellws + pk + 64) * sizeof(double);
}

extern "C" int nis_flow_backward(nis_flow* f, const double* planes_hist, int64_t ld, int64_t N, int32_t w_has_jac,
                                 const double* const* params, const double* bnsave, const double* gxj, double* const* gparams,
                                 double* gx_out, void* workspace, size_t workspace_bytes, void* stream) {
  if (!f || !planes_hist || !params || !bnsave || !gxj || !gparams || N < 1 || ld != N) return NIS_E_ARG;   // gradient planes share ld
  if (!nis_flow_backward_supported(f)) return NIS_E_UNSUPPORTED;
  if (!workspace || workspace_bytes < nis_flow_backward_workspace_bytes(f, N)) return NIS_E_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  const int d = f->d;
  const long set = (long)(d + 1) * ld;
  double* g = (double*)workspace;                       // gradient planes (d+1, ld)
  double* cellws = g + ((size_t)(d + 1) * N + 15) / 16 * 16;      // 128-byte aligned: the wide adjoint uses vector loads
  size_t cw = 0;
  std::vector<CellLayout> lay;
  flow_layout(f, 1, &lay, nullptr);
  for (int ci = 0; ci < f->n_cells; ++ci) cw = std::max(cw, nis_bwd_cell_workspace_doubles(f->cells[ci], lay[ci].WM, N));
  double* pk = cellws + cw;
  const int tgrid = grid_for(8, 1, N, NIS_TPB);
  const size_t tile_bytes = ((size_t)NIS_TPB * (d + 1) + 16) * sizeof(double);
  // dL/dXJ (N, d+1) row-major in the model's OUTPUT column order -> gradient planes (output column c lives in plane out_col[c])
  {
    ColMap m;
    for (int j = 0; j < d; ++j) m.col[j] = f->out_col[j];
    m.col[d] = d;
    hipLaunchKernelGGL(k_scatter_rows, dim3(tgrid), dim3(NIS_TPB), tile_bytes, s, gxj, (long)N, d + 1, m, g, (long)N);
    HIP_OK(hipGetLastError());
  }
  size_t bn_total = nis_flow_bnsave_count(f);
  size_t bn_off = bn_total;
  for (int ci = f->n_cells - 1; ci >= 0; --ci) {
    const nis_cell_desc& cd = f->cells[ci];
    const CellLayout& L = lay[ci];
    bn_off -= (size_t)(cd.n_hidden + 1) * 4 * L.WM;
    CellK c;
    memset(&c, 0, sizeof(c));
    c.kind = cd.kind;
    c.d = d;
    c.P = cd.P;
    c.T = d - cd.P;
    c.K = cd.K;
    c.L = cd.n_hidden;
    c.out = cell_out(cd);
    c.R = cell_R(cd);
    for (int l = 0; l < cd.n_hidden; ++l) c.width[l] = cd.width[l];
    for (int j = 0; j < d; ++j) c.in_plane[j] = cd.in_col[j];
    c.jplane = d;
    c.first = (ci == 0 && !w_has_jac) ? 1 : 0;
    c.G = 1;
    c.Gy = 1;
    const size_t WM = L.WM, R = c.R, nl = c.L;
    double* wp = pk;
    double* wlast = wp + nl * WM * WM;
    double* bias = wlast + R * WM;
    double* gb = bias + (R + 7) / 8 * 8;
    double* coef = gb + (nl + 1) * 2 * WM;
    c.wp = wp;
    c.wlast = wlast;
    c.bias = bias;
    c.gb = gb;
    c.coef = coef;
    hipLaunchKernelGGL(k_pack, dim3(WM >= 16 ? 32 : 1), dim3(NIS_TPB), 0, s, c, (int)WM, params[ci], (const double*)nullptr, 0, wp, wlast,
                       bias, gb, (double*)nullptr, cell_hidden_bias(cd) ? 1 : 0);
    HIP_OK(hipGetLastError());
    MfmaW mw;
    mw.wm_all = mw.wm_last = mw.bias_pad = nullptr;
    if (WM >= 16) {
      const size_t RT = (R + 15) / 16;
      double* wm_all = coef + (nl + 1) * 2 * WM;
      double* wm_last = wm_all + nl * WM * WM;
      double* bias_pad = wm_last + RT * 16 * WM;
      hipLaunchKernelGGL(k_pack_mfma, dim3(32), dim3(NIS_TPB), 0, s, (int)WM, c.L, c.R, (int)RT, (const double*)wp, (const double*)wlast,
                         (const double*)bias, wm_all, wm_last, bias_pad, cd.kind == NIS_AFFINE ? d - cd.P : 0);
      HIP_OK(hipGetLastError());
      mw.wm_all = wm_all;
      mw.wm_last = wm_last;
      mw.bias_pad = bias_pad;
    }
    int rc = nis_bwd_cell_run(c, L.WM, planes_hist + (size_t)ci * set, planes_hist + (size_t)(ci + 1) * set, g, bnsave + bn_off,
                              (long)ld, (long)N, cellws, gparams[ci], mw, s);
    if (rc != NIS_OK) return rc;
  }
  if (gx_out) {
    ColMap m;
    for (int j = 0; j <= d; ++j) m.col[j] = j;
    hipLaunchKernelGGL(k_emit, dim3(tgrid), dim3(NIS_TPB), tile_bytes, s, (const double*)g, (long)N, (long)N, d, m, gx_out);
    HIP_OK(hipGetLastError());
  }
  return NIS_OK;
}
