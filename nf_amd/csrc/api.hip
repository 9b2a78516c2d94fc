// Host side of libnis_hip.so (C ABI of include/nis_hip.h): launch planning of a forward / a launch set of integrate()
// iterations (flow_run), the entry points, the adjoint driver and the per-launch timing hooks.  The kernels it launches live in
// flow_wm.hip (cells), api_pack.hip, api_ingest.hip, api_moments.hip, rambo.hip and flow_bwd*.hip (see api_internal.h).
#include "api_internal.h"

static thread_local std::string g_err;
int nis_fail_hip(hipError_t e, const char* what) {
  g_err = std::string(what) + ": " + hipGetErrorString(e);
  return NIS_E_HIP;
}

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
int nis_device_cus() {
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
int nis_grid_for(int blocks_per_cu, int G, int64_t N, int tpb) {
  const int64_t resident = (int64_t)blocks_per_cu * nis_device_cus();
  int64_t gx = resident / G;
  gx = std::min<int64_t>(gx, (N + tpb - 1) / tpb);
  gx = std::min<int64_t>(gx, NIS_MAXB);
  return (int)std::max<int64_t>(1, gx);
}

int nis_env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return (v && *v) ? atoi(v) : dflt;
}

// exact common width of the hidden layers when a specialised instantiation exists (WM = 4: 3-wide nets), else 0
static int cell_wx(const nis_cell_desc& cd, int WM) {
  if (WM != 4 || cd.P > 3) return 0;
  for (int l = 0; l < cd.n_hidden; ++l)
    if (cd.width[l] > 3) return 0;
  return 3;
}

template <int WM>
static void cell_grids(int kind, int km, int wx, bool folded, int G, int64_t N, int* gs, int* ga) {
  *gs = nis_grid_for(stats_blocks_per_cu<WM>(wx, folded), G, N, nis_stats_tpb(WM));
  *ga = nis_grid_for(apply_blocks_per_cu<WM>(kind, km, wx), G, N, NIS_TPB);
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
size_t nis_bwd_cell_workspace_doubles(const nis_cell_desc& cd, int wm, int64_t N, int G);
size_t nis_bwd_cell_params(const nis_cell_desc& cd);
int nis_bwd_cell_run(const CellK& c, const nis_cell_desc& cd, int wm, const double* x, const double* xo, double* g, const double* bnsave,
                     long ld, long N, int G, long x_gs, long g_gs, long bn_gs, double* ws, double* gparams, const MfmaW& mw, bool eval_bn,
                     hipStream_t s);

// BN_1 derived from the pass-through block's co-moments (CellK::dcov): 2 <= P <= 8, the record (3 P feature entries + P (P - 1) / 2
// co-moments) must fit the 3 WM doubles of a BN record, AND the merge in load_coefs gives every pair one of its WM feature columns --
// width 16 with P = 7 has 21 pairs: the record fits, the merge does not (pairs 16 .. 20 were never summed: BN_1 variances ~1 % off,
// found by a shifted-seed sweep in round 3).
static bool dcov_fits(int P, int WM) {
  const int npair = P * (P - 1) / 2;
  return P >= 2 && P <= 8 && 3 * P + npair <= 3 * WM && npair <= WM;
}
static bool use_mfma() {
  static const int off = nis_env_int("NIS_NO_MFMA", 0);
  return !off;
}

// multi-rank exact BatchNorm: local records -> one record per feature in this rank's slot -> exchange hook
static int sync_records(const RunOpts& o, const double* records, int nparts, int WM, hipStream_t s) {
  if (!o.sync_fn) return NIS_OK;
  HIP_OK(nis_launch_sync_reduce(records, nparts, WM, o.sync_buf + (size_t)o.sync_rank * WM * 3, s));
  o.sync_fn(o.sync_ctx, o.sync_buf, WM * 3, (void*)s);
  return NIS_OK;
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
  *grid = *ok ? nis_grid_for(mfma_blocks_per_cu<WM>(c), G, N, per_wg) : 1;
  // the statistics kernels keep THEIR residency (two 256-thread workgroups per CU) whatever the apply kernel's workgroup shape is.  They
  // hold no LDS image and ~100 VGPRs, but 4 / 6 / 8 workgroups per CU measured 0 ... 8 % SLOWER (round 2: C3 13.9 -> 14.2 ms, C5
  // 7.6 -> 8.2 ms per iteration): every workgroup's prologue merges the previous layer's records, one per workgroup
  const int sw = nis_mf_stats_tpb(WM, (long long)G * N) / 256;                  // workgroups of 256 threads one statistics workgroup stands for
  *grid_stats = *ok ? nis_grid_for(2 / sw, G, N, per_wg * sw) : 1;      // (one per CU: C5 training step -1 %, C3 / C4 / C5 iterations +4 / +2 / +2.5 %)
}

// Descriptor-level mirror of launch_apply's SPEC conditions (flow_wm.hip): every cell of a d = 2 PWQuad chain with 3-wide
// conditioners and a compiled bin count, camel integrand fused into the last cell, nothing materialised.
static bool direct_rows_ok(const nis_flow* f, const std::vector<CellLayout>& lay, const RunOpts& o, int w_dtype, int wcols, int ingest_P) {
  if (f->d != 2 || wcols != 2 || w_dtype != NIS_F32 || o.w_has_jac || o.hist || o.keep_planes || o.xj_out || (o.bins_out && !o.spec_bins) ||
      o.fres_out || o.sync_fn || o.f_kind != NIS_F_CAMEL || !o.out4)
    return false;
  if (o.bn_mode != NIS_BN_TRAIN) return false;     // eval mode: inner cells carry no BN_0 rider and run the generic kernel
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
  if (G > 1 && (o.bins_out || o.w_has_jac || (o.xj_out && !o.hist))) return NIS_E_UNSUPPORTED;
  if (!workspace || workspace_bytes < nis_flow_workspace_bytes(f, G)) return NIS_E_WORKSPACE;
  std::vector<CellLayout> lay;
  size_t off_fpart = 0;
  nis_flow_layout(f, G, &lay, &off_fpart);
  int* status = (int*)workspace;
  double* body = (double*)((char*)workspace + WS_HEADER);
  const int d = f->d;
  const bool train = o.bn_mode == NIS_BN_TRAIN;
  const int wcols = o.w_has_jac ? d + 1 : d;
  const long set = (long)(d + 1) * ld;                                   // one plane set
  const long plane_gs = o.hist ? (long)(f->n_cells + 1) * set : set;     // history: batch g owns n_cells + 1 consecutive sets

  // is the padded parameter image in this workspace still valid?
  bool repack = o.pack_token == 0 || o.pack_token != f->packed_token || o.bn_mode != f->packed_mode || workspace != f->packed_ws ||
                G != f->packed_G;
  for (int ci = 0; ci < f->n_cells && !repack; ++ci) repack = params[ci] != f->packed_params[ci] || running[ci] != f->packed_running[ci];
  if (o.hist && !train && o.bnsave) repack = true;        // eval-mode forward with history: the pack launch also fills the adjoint's BN record

  // ---- ingest (+ BN_0 records of the first cell)
  const nis_cell_desc& c0 = f->cells[0];
  IngestStats ist;
  memset(&ist, 0, sizeof(ist));
  ist.status = status;
  ist.maxb = NIS_MAXB;
  ist.plane_gs = plane_gs;
  // a matrix-core first cell with 2..8 pass-through columns takes its BN_0 records from the covariance pass (CellK::dcov: BN_1 is
  // derived, the layer-1 statistics launch disappears) instead of the ingest rider, which has no co-moments
  const bool cell0_cov = train && lay[0].WM >= 16 && use_mfma() && c0.n_hidden >= 2 && dcov_fits(c0.P, lay[0].WM) && !o.sync_fn &&
                         !nis_env_int("NIS_NO_DCOV", 0);
  // ... unless the batch is DRAWN by the ingest launch (o.rng): that kernel holds a sample's whole row in registers and adds the
  // co-moments to its records, so the first cell needs neither the covariance pass nor its layer-1 statistics launch
  const bool ingest_cov = cell0_cov && o.rng;
  const bool ingest_stats = train && c0.P <= NIS_PMAX_FUSED && (!cell0_cov || ingest_cov);
  if (ingest_stats) {
    ist.P = c0.P;
    for (int p = 0; p < c0.P; ++p) ist.sel[p] = c0.in_col[p];
    ist.wm = lay[0].WM;
    ist.partials = body + lay[0].off_part;
    ist.cov = ingest_cov ? 1 : 0;
  }
  // Direct row input: a d = 2 chain of SPEC kernels with the camel integrand fused never needs X as planes, so the
  // (N, 2) float32 rows of w are read by the cells themselves (a column lives in w until a cell overwrites it) and the
  // ingest launch only gathers BN_0's records: 80 MB read instead of 80 MB read + 160 MB written per 1e7 samples.
  const bool direct = direct_rows_ok(f, lay, o, w_dtype, wcols, ist.P) && !nis_env_int("NIS_NO_DIRECT_ROWS", 0);
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
  if (repack) {                       // ONE launch for all cells (k_pack_cells)
    size_t bn_pack_off = 0;
    std::vector<PackJob> jobs(f->n_cells);
    for (int ci = 0; ci < f->n_cells; ++ci) {
      const nis_cell_desc& cd = f->cells[ci];
      const CellLayout& L = lay[ci];
      PackJob& j = jobs[ci];
      memset(&j, 0, sizeof(j));
      j.prm = params[ci];
      j.run = (const double*)running[ci];
      j.wp = body + L.off_wp;
      j.wlast = body + L.off_wlast;
      j.bias = body + L.off_bias;
      j.gb = body + L.off_gb;
      j.hb = body + L.off_hb;
      j.coef = body + L.off_coef;
      j.bnsave = (o.hist && !train && o.bnsave) ? o.bnsave + bn_pack_off : nullptr;
      bn_pack_off += (size_t)(cd.n_hidden + 1) * 4 * L.WM;
      j.WM = L.WM;
      j.P = cd.P;
      j.L = cd.n_hidden;
      j.R = cell_R(cd);
      j.RT = L.RT;
      for (int l = 0; l < cd.n_hidden; ++l) j.width[l] = cd.width[l];
      j.eval_mode = train ? 0 : 1;
      j.hidden_bias = cell_hidden_bias(cd) ? 1 : 0;
      if (L.WM >= 16 && use_mfma()) {
        j.wm_all = body + L.off_wm;
        j.wm_last = body + L.off_wmlast;
        j.bias_pad = body + L.off_biaspad;
        j.affT = cd.kind == NIS_AFFINE ? d - cd.P : 0;
        j.regmode = nis_mf_regmode(cd.kind, cd.K, d - cd.P, L.WM);
        j.T = d - cd.P;
        j.K = cd.K;
      }
    }
    NisProfScope ps(NIS_K_PACK, s);
    HIP_OK(nis_launch_pack_cells(jobs, s));
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
    HIP_OK(nis_launch_group_epilogue(epi, epi.blk_off[epi.ncell] + (fpart ? G : 0), s));
    epi.ncell = 0;
    return NIS_OK;
  };

  if (!planes) return NIS_E_ARG;
  if (o.rng && (o.w_has_jac || o.hist || o.sync_fn)) return NIS_E_UNSUPPORTED;      // the draw belongs to integrate()'s loop body
  // In-kernel draws of the d = 2 direct-row chain: the record pass GENERATES the float32 rows (same Philox mapping as
  // nis_philox_rows) into the state plane that chain never touches -- the plane of cell 0's pass-through column, N float64 =
  // N float32 pairs -- and the cells read them from there.  Every other model: k_ingest_rng writes the float64 planes directly.
  const float* wrows = (const float*)w;
  long wrows_gs = 2 * (long)N;
  long wplane_ld = 0;                      // > 0: two float32 column planes instead of rows (K = 4 chains: PLN kernel instantiations)
  if (o.rng && direct) {
    wrows = (const float*)(planes + (size_t)f->cells[0].in_col[0] * ld);
    wrows_gs = 2 * plane_gs;
    // even N: the second plane starts 8-byte aligned.  Two cells only: an INNER cell writes its float64 output plane over the
    // float32 planes other workgroups still read (row layout: a thread overwrites exactly the row it has read, any chain length)
    bool k4 = N % 2 == 0 && f->n_cells == 2;
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
    const int tgrid = nis_grid_for(direct ? 4 : 8, Gg, N, NIS_TPB);
    bool col_in_w[2] = {direct, direct};
    nis_prof_mark(NIS_K_INGEST, 1, cs);
    hipError_t ie;
    if (direct && o.rng)
      ie = nis_launch_ingest2_rng(o.seed, o.stream0, o.rng_state, (long)N, const_cast<float*>(wrows), wrows_gs, wplane_ld, o.w_out, is2,
                                  tgrid, Gy, cs);
    else if (o.rng)
      ie = nis_launch_ingest_rng(o.seed, o.stream0, o.rng_state, (long)N, d, planes, (long)ld, o.w_out, is2, tgrid, Gy, cs);
    else
      ie = nis_launch_ingest(w, w_dtype, wcols, (long)N, planes, (long)ld, is2, tgrid, Gy, direct, cs);
    nis_prof_mark(NIS_K_INGEST, 0, cs);
    HIP_OK(ie);
    if (o.sync_fn) {
      if (G != 1 || !train || o.hist) return NIS_E_UNSUPPORTED;
      if (ingest_stats)
        if (int rc = sync_records(o, is2.partials, tgrid, lay[0].WM, cs)) return rc;
    }
    int prev_records = tgrid;           // records of BN_0 of the current cell written by the previous launch
    bool have_stats0 = ingest_stats;
    bool stats0_cov = ingest_cov;       // ... and they carry the co-moments of the pass-through block (CellK::dcov format)
    double* pl = planes;
    size_t bn_off = 0;
    for (int ci = 0; ci < f->n_cells; ++ci) {
      const nis_cell_desc& cd = f->cells[ci];
      const CellLayout& L = lay[ci];
      if (o.hist) pl = planes + (size_t)ci * set;
      int grid_stats = 1, grid_apply = 1;
      switch (L.WM) {
        // folded statistics launches (k_bn_stats_fold): train mode, one pass-through column, BN_1 derived
        case 4: cell_grids<4>(cd.kind, L.KM, cell_wx(cd, L.WM), train && cd.P == 1,
                              Gg, N, &grid_stats, &grid_apply); break;
        case 8: cell_grids<8>(cd.kind, L.KM, 0, false, Gg, N, &grid_stats, &grid_apply); break;
        case 16: cell_grids<16>(cd.kind, L.KM, 0, false, Gg, N, &grid_stats, &grid_apply); break;
        case 32: cell_grids<32>(cd.kind, L.KM, 0, false, Gg, N, &grid_stats, &grid_apply); break;
        default: cell_grids<64>(cd.kind, L.KM, 0, false, Gg, N, &grid_stats, &grid_apply); break;
      }
      CellK c = cell_shape(ci);
      c.regmode = nis_mf_regmode(cd.kind, cd.K, d - cd.P, L.WM);
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
      c.derive1 = (train && cd.P == 1) ? 1 : 0;
      // matrix-core cells with 2..8 pass-through columns: BN_0's records carry the co-moments of the block, BN_1 is derived from
      // them and the layer-1 statistics launch is skipped (not with the multi-rank record exchange: its records are per feature)
      c.dcov = (train && mfma && cd.n_hidden >= 2 && dcov_fits(cd.P, L.WM) && !o.sync_fn &&
                (have_stats0 == false || stats0_cov) && !nis_env_int("NIS_NO_DCOV", 0)) ? 1 : 0;
      if (c.dcov && have_stats0) {
        c.derive1 = 1;                  // the producing launch (ingest with in-kernel draws) already wrote them: nparts[0] = its grid
      } else if (c.dcov) {
        c.derive1 = 1;
        // the covariance pass is a latency-bound stream (8 loads, 44 FMAs per sample, <= 128 VGPRs): two workgroups per CU (four: equal at G = 3, slower at G = 1 -- every consumer merges the records)
        constexpr int bpc_cov = 2;                   // (one 512-thread workgroup per CU, as the matrix-core statistics kernel has: no gain here)
        // ... and at least eight samples per thread: at 2e5 samples 98 workgroups instead of 512 -- the pass itself is a chain of latencies
        // either way (15-20 us), its consumer merges a fifth of the records (C5 training step 7.06 -> 7.00 ms; 1e6-sample iterations equal)
        c.nparts[0] = nis_grid_for(bpc_cov, Gg, N, 256 * 8);
      }
      // single pass-through column on the fp64 VALU kernels: BN_0, Linear_0 and BN_1 collapse into one affine map of x
      // (needs the layer-1 statistics launch to be gone: train mode with derive1, or eval mode)
      c.fold = (cd.P == 1 && !mfma && (c.derive1 || !train)) ? 1 : 0;
      c.wx = cell_wx(cd, L.WM);
      c.sync_part = o.sync_fn ? o.sync_buf : nullptr;
      c.sync_world = o.sync_world;
      c.pmerge = 1;                                  // a k_bn_finalize launch costs >= 4.3 us, the consumers' prologue merge ~1 us
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
      c.out_off = o.hist ? set : 0;
      c.bnsave = (o.hist && o.bnsave && train) ? o.bnsave + bn_off : nullptr;
      c.bnsave_gs = (long)nis_flow_bnsave_count(f);
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
      stats0_cov = false;
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

  if (int rc = chain(s, 0, G, G, true, nullptr)) return rc;
  if (epi_fused) {
    if (int rc2 = epi_flush(body + off_fpart, epi_fparts)) return rc2;
  }
  if (int rc2 = epi_flush(nullptr, 0)) return rc2;
  packed_done();
  if (o.xj_out) {
    ColMap m;
    for (int j = 0; j < d; ++j) m.col[j] = f->out_col[j];
    m.col[d] = d;
    const int tgrid = nis_grid_for(8, G, N, NIS_TPB);
    double* pl = o.hist ? planes + (size_t)f->n_cells * set : planes;      // hist: the final state lives in the last plane set
    NisProfScope ps(NIS_K_EMIT, s);
    HIP_OK(nis_launch_emit(pl, (long)ld, (long)N, d, m, o.xj_out, tgrid, s, G, plane_gs));
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
  // f_kind < 0 (NIS_F_NONE): no integrand -- the G iterations only leave X and J in `planes` (keep_planes) for a caller-side f
  if (f_kind < 0 ? (!keep_planes || !planes || fres_out) : !out4) return NIS_E_ARG;
  if (f_kind == NIS_F_CAMEL && f && f->d < 2) return NIS_E_ARG;
  RunOpts o;
  o.G = n_batches;
  o.bn_mode = bn_mode;
  o.pack_token = pack_token;
  o.tracked = tracked;
  o.f_kind = f_kind < 0 ? -1 : f_kind;
  o.f_params = f_params_host;
  o.fres_out = fres_out;
  o.out4 = out4;
  o.keep_planes = keep_planes;
  return flow_run(f, w, w_dtype, N, params, running, planes, ld, workspace, workspace_bytes, (hipStream_t)stream, o);
}

extern "C" int nis_integrate_steps_rng(nis_flow* f, int32_t n_batches, int64_t N, const double* const* params, double* const* running,
                                       long long* const* tracked, int32_t bn_mode, int64_t pack_token, int32_t f_kind,
                                       const double* f_params_host, uint64_t seed, uint64_t stream0, uint64_t* rng_state, float* w_out,
                                       double* planes, int64_t ld, int32_t keep_planes, double* out4, void* workspace,
                                       size_t workspace_bytes, void* stream) {
  if (f_kind < 0 ? (!keep_planes || !planes) : !out4) return NIS_E_ARG;      // NIS_F_NONE: see nis_integrate_steps
  if (f_kind == NIS_F_CAMEL && f && f->d < 2) return NIS_E_ARG;
  RunOpts o;
  o.G = n_batches;
  o.bn_mode = bn_mode;
  o.pack_token = pack_token;
  o.tracked = tracked;
  o.f_kind = f_kind < 0 ? -1 : f_kind;
  o.f_params = f_params_host;
  o.out4 = out4;
  o.keep_planes = keep_planes;
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
  nis_flow_layout(f, 1, &lay, nullptr);
  if (!direct_rows_ok(f, lay, o, w_dtype, f->d, f->cells[0].P) || nis_env_int("NIS_NO_DIRECT_ROWS", 0)) return NIS_E_UNSUPPORTED;
  return flow_run(f, w, w_dtype, N, params, running, planes, ld, workspace, workspace_bytes, (hipStream_t)stream, o);
}

// 1 when nis_integrate_steps can fuse a built-in integrand into the last cell's kernel (narrow conditioners: the fp64 VALU
// apply kernel's rider; wide ones: the FUSE instantiations of the matrix-core apply kernel)
extern "C" int nis_flow_fusable(const nis_flow* f) {
  if (!f) return 0;
  std::vector<CellLayout> lay;
  nis_flow_layout(f, 1, &lay, nullptr);
  const nis_cell_desc& cd = f->cells[f->n_cells - 1];
  const CellLayout& L = lay[f->n_cells - 1];
  if (L.WM <= 8) return 1;
  if (!use_mfma() || !nis_mfma_fusable(cd.kind, L.KM)) return 0;
  CellK c;
  memset(&c, 0, sizeof(c));
  c.kind = cd.kind;
  c.K = cd.K;
  c.T = f->d - cd.P;
  c.R = cell_R(cd);
  c.out = cell_out(cd);
  c.regmode = nis_mf_regmode(cd.kind, cd.K, c.T, L.WM);
  bool ok = false;
  switch (L.WM) {
    case 16: ok = mfma_supported<16>(c); break;
    case 32: ok = mfma_supported<32>(c); break;
    case 64: ok = mfma_supported<64>(c); break;
    default: break;
  }
  return ok ? 1 : 0;
}

// How nis_integrate_steps_rng would draw for this model / call: 2 = the d = 2 direct-row chain (float32 rows in a spare state
// plane, SPEC kernels), 1 = k_ingest_rng (any model: float64 planes generated), 0 = never (kept for callers that must know).
extern "C" int nis_integrate_rng_mode(const nis_flow* f, int32_t n_batches, int64_t N, int32_t bn_mode, int32_t f_kind, int32_t keep_planes) {
  if (!f || n_batches < 1 || N < 1 || f_kind < 0) return 0;
  RunOpts o;
  o.G = n_batches;
  o.bn_mode = bn_mode;
  o.f_kind = f_kind;
  o.keep_planes = keep_planes;
  o.rng = 1;
  o.out4 = reinterpret_cast<double*>(8);          // "present": direct_rows_ok only tests it
  std::vector<CellLayout> lay;
  nis_flow_layout(f, n_batches, &lay, nullptr);
  const int ingest_P = bn_mode == NIS_BN_TRAIN ? f->cells[0].P : 0;
  if (direct_rows_ok(f, lay, o, NIS_F32, f->d, ingest_P) && !nis_env_int("NIS_NO_DIRECT_ROWS", 0)) return 2;
  return 1;
}

extern "C" int nis_philox_rows(uint64_t seed, uint64_t stream0, int32_t n_batches, int64_t N, int32_t d, float* out, void* stream) {
  if (!out || n_batches < 1 || n_batches > 65535 || N < 1 || d < 1 || d > NIS_MAX_DIM) return NIS_E_ARG;
  HIP_OK(nis_launch_philox_rows(seed, stream0, n_batches, (long)N, d, out, (hipStream_t)stream));
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

extern "C" int nis_flow_forward_hist(nis_flow* f, const void* w, int32_t w_dtype, int32_t w_has_jac, int64_t N,
                                     const double* const* params, double* const* running, long long* const* tracked, int32_t bn_mode,
                                     int64_t pack_token, double* planes_hist, int64_t ld, double* bnsave, double* xj_out,
                                     void* workspace, size_t workspace_bytes, void* stream) {
  if (!planes_hist || !bnsave) return NIS_E_ARG;
  RunOpts o;
  o.w_has_jac = w_has_jac;
  o.bn_mode = bn_mode;
  o.pack_token = pack_token;
  o.tracked = tracked;
  o.xj_out = xj_out;
  o.hist = 1;
  o.bnsave = bnsave;
  return flow_run(f, w, w_dtype, N, params, running, planes_hist, ld, workspace, workspace_bytes, (hipStream_t)stream, o);
}

extern "C" int nis_flow_forward_train(nis_flow* f, const void* w, int32_t w_dtype, int32_t w_has_jac, int64_t N,
                                      const double* const* params, double* const* running, long long* const* tracked,
                                      int64_t pack_token, double* planes_hist, int64_t ld, double* bnsave, double* xj_out,
                                      void* workspace, size_t workspace_bytes, void* stream) {
  return nis_flow_forward_hist(f, w, w_dtype, w_has_jac, N, params, running, tracked, NIS_BN_TRAIN, pack_token, planes_hist, ld, bnsave,
                               xj_out, workspace, workspace_bytes, stream);
}

// doubles of the packed parameter image a cell's adjoint works from (row-major images + MFMA fragments when wide)
static size_t bwd_pack_doubles(const nis_cell_desc& cd, size_t WM) {
  const size_t R = cell_R(cd), nl = cd.n_hidden, RT = (R + 15) / 16;
  size_t n = nl * WM * WM + R * WM + (R + 7) / 8 * 8 + 2 * (nl + 1) * 2 * WM;
  if (WM >= 16) n += nl * WM * WM + RT * 16 * WM + RT * 16;
  return n;
}

// doubles of one call's adjoint scratch: gradient planes of the G minibatches, the largest cell workspace, the packed parameter images
static void bwd_ws_layout(const nis_flow* f, int G, int64_t N, size_t* gplanes, size_t* cellws, size_t* pk) {
  std::vector<CellLayout> lay;
  nis_flow_layout(f, 1, &lay, nullptr);
  *cellws = *pk = 0;
  for (int ci = 0; ci < f->n_cells; ++ci) {
    *cellws = std::max(*cellws, nis_bwd_cell_workspace_doubles(f->cells[ci], lay[ci].WM, N, G));
    *pk += (bwd_pack_doubles(f->cells[ci], lay[ci].WM) + 15) / 16 * 16;          // every cell its own image: packed in one launch
  }
  *gplanes = ((size_t)G * (f->d + 1) * N + 15) / 16 * 16;          // 128-byte aligned: the wide adjoint uses vector loads
  *cellws = (*cellws + 15) / 16 * 16;
  *pk = (*pk + 15) / 16 * 16;
}

extern "C" size_t nis_flow_backward_workspace_bytes_batched(const nis_flow* f, int32_t n_batches, int64_t N) {
  if (!f || n_batches < 1 || N < 1) return 0;
  size_t gp, cw, pk;
  bwd_ws_layout(f, n_batches, N, &gp, &cw, &pk);
  return (gp + cw + pk + 64) * sizeof(double);
}
extern "C" size_t nis_flow_backward_workspace_bytes(const nis_flow* f, int64_t N) { return nis_flow_backward_workspace_bytes_batched(f, 1, N); }

// where the output cotangent comes from: a (N, d+1) row-major dL/dXJ, or the minibatch loss of nis_loss_terms (dL/dJ only)
struct BwdSeed {
  const double* gxj = nullptr;
  const double* fv = nullptr;          // loss seed: integrand values (G, N), device scalars maxf / gloss, nis_loss_terms' records (G, 12)
  const double* maxf = nullptr;
  const double* rec12 = nullptr;
  const double* gloss = nullptr;
  int est = 0;
  bool eval_bn = false;                // the forward ran eval-mode BatchNorm (running statistics): no batch-mean terms in its backward
};

// G minibatches: planes_hist (G, n_cells + 1, d + 1, ld), bnsave (G, nis_flow_bnsave_count) (eval mode: one record), gradient
// planes (G, d + 1, ld); gparams[c] receives the SUM over the minibatches.
static int flow_backward_impl(nis_flow* f, const double* planes_hist, int64_t ld, int G, int64_t N, int32_t w_has_jac,
                              const double* const* params, const double* bnsave, const BwdSeed& seed, double* const* gparams,
                              double* gx_out, void* workspace, size_t workspace_bytes, void* stream) {
  const double* gxj = seed.gxj;
  if (!f || !planes_hist || !params || !bnsave || (!gxj && !seed.fv) || !gparams || N < 1 || ld != N || G < 1 || G > 65535) return NIS_E_ARG;   // gradient planes share ld
  if (G > 1 && (gxj || gx_out || w_has_jac)) return NIS_E_UNSUPPORTED;
  if (!nis_flow_backward_supported(f)) return NIS_E_UNSUPPORTED;
  if (!workspace || workspace_bytes < nis_flow_backward_workspace_bytes_batched(f, G, N)) return NIS_E_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  const int d = f->d;
  const long set = (long)(d + 1) * ld;
  const long hist_gs = (long)(f->n_cells + 1) * set;
  size_t gpl, cw, pkd;
  bwd_ws_layout(f, G, N, &gpl, &cw, &pkd);
  double* g = (double*)workspace;                       // gradient planes (G, d+1, ld)
  double* cellws = g + gpl;
  double* pk = cellws + cw;
  std::vector<CellLayout> lay;
  nis_flow_layout(f, 1, &lay, nullptr);
  const int tgrid = nis_grid_for(8, G, N, NIS_TPB);
  // dL/dXJ (N, d+1) row-major in the model's OUTPUT column order -> gradient planes (output column c lives in plane out_col[c])
  if (!gxj) {                         // seeded by the minibatch loss: dL/dJ from the records, zeros for the coordinates
    HIP_OK(nis_launch_loss_cotangent(seed.fv, planes_hist + (size_t)f->n_cells * set + (size_t)d * ld, 1, (long)N, seed.maxf, seed.rec12,
                                     seed.gloss, seed.est, d, g, (long)N, s, G, hist_gs, set));
  } else {
    ColMap m;
    for (int j = 0; j < d; ++j) m.col[j] = f->out_col[j];
    m.col[d] = d;
    HIP_OK(nis_launch_scatter_rows(gxj, (long)N, d + 1, m, g, (long)N, tgrid, s));
  }
  // parameter images of all cells (reference row order, train-mode BatchNorm affine): one launch, every cell its own region of `pk`
  std::vector<PackJob> jobs(f->n_cells);
  {
    double* at = pk;
    for (int ci = 0; ci < f->n_cells; ++ci) {
      const nis_cell_desc& cd = f->cells[ci];
      const size_t WM = lay[ci].WM, R = cell_R(cd), nl = cd.n_hidden, RT = (R + 15) / 16;
      PackJob& j = jobs[ci];
      memset(&j, 0, sizeof(j));
      j.prm = params[ci];
      j.wp = at;
      j.wlast = j.wp + nl * WM * WM;
      j.bias = j.wlast + R * WM;
      j.gb = j.bias + (R + 7) / 8 * 8;
      j.coef = j.gb + (nl + 1) * 2 * WM;
      double* end = j.coef + (nl + 1) * 2 * WM;
      j.WM = (int)WM;
      j.P = cd.P;
      j.L = cd.n_hidden;
      j.R = (int)R;
      j.RT = (int)RT;
      for (int l = 0; l < cd.n_hidden; ++l) j.width[l] = cd.width[l];
      j.hidden_bias = cell_hidden_bias(cd) ? 1 : 0;
      if (WM >= 16) {
        j.wm_all = end;
        j.wm_last = j.wm_all + nl * WM * WM;
        j.bias_pad = j.wm_last + RT * 16 * WM;
        end = j.bias_pad + RT * 16;
        j.affT = cd.kind == NIS_AFFINE ? d - cd.P : 0;
      }
      if ((size_t)(end - at) != bwd_pack_doubles(cd, WM)) return NIS_E_ARG;
      at += (bwd_pack_doubles(cd, WM) + 15) / 16 * 16;
    }
    HIP_OK(nis_launch_pack_cells(jobs, s));
  }
  const size_t bn_total = nis_flow_bnsave_count(f);
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
    const PackJob& pj = jobs[ci];
    c.wp = pj.wp;
    c.wlast = pj.wlast;
    c.bias = pj.bias;
    c.gb = pj.gb;
    c.coef = pj.coef;
    MfmaW mw;
    mw.wm_all = pj.wm_all;
    mw.wm_last = pj.wm_last;
    mw.bias_pad = pj.bias_pad;
    int rc = nis_bwd_cell_run(c, cd, L.WM, planes_hist + (size_t)ci * set, planes_hist + (size_t)(ci + 1) * set, g, bnsave + bn_off,
                              (long)ld, (long)N, G, hist_gs, set, seed.eval_bn ? 0L : (long)bn_total, cellws, gparams[ci], mw,
                              seed.eval_bn, s);
    if (rc != NIS_OK) return rc;
  }
  if (gx_out) {
    ColMap m;
    for (int j = 0; j <= d; ++j) m.col[j] = j;
    HIP_OK(nis_launch_emit(g, (long)N, (long)N, d, m, gx_out, tgrid, s));
  }
  return NIS_OK;
}


extern "C" int nis_flow_backward_eval(nis_flow* f, const double* planes_hist, int64_t ld, int64_t N, int32_t w_has_jac,
                                      const double* const* params, const double* bnsave, const double* gxj, double* const* gparams,
                                      double* gx_out, void* workspace, size_t workspace_bytes, void* stream) {
  if (!gxj) return NIS_E_ARG;
  BwdSeed seed;
  seed.gxj = gxj;
  seed.eval_bn = true;
  return flow_backward_impl(f, planes_hist, ld, 1, N, w_has_jac, params, bnsave, seed, gparams, gx_out, workspace, workspace_bytes, stream);
}

extern "C" int nis_flow_backward(nis_flow* f, const double* planes_hist, int64_t ld, int64_t N, int32_t w_has_jac,
                                 const double* const* params, const double* bnsave, const double* gxj, double* const* gparams,
                                 double* gx_out, void* workspace, size_t workspace_bytes, void* stream) {
  if (!gxj) return NIS_E_ARG;
  BwdSeed seed;
  seed.gxj = gxj;
  return flow_backward_impl(f, planes_hist, ld, 1, N, w_has_jac, params, bnsave, seed, gparams, gx_out, workspace, workspace_bytes, stream);
}

extern "C" int nis_flow_backward_loss(nis_flow* f, const double* planes_hist, int64_t ld, int64_t N, int32_t w_has_jac,
                                      const double* const* params, const double* bnsave, const double* fvals, const double* maxf_dev,
                                      const double* rec12, const double* gloss_dev, int32_t est, double* const* gparams, void* workspace,
                                      size_t workspace_bytes, void* stream) {
  if (!fvals || !maxf_dev || !rec12 || !gloss_dev) return NIS_E_ARG;
  BwdSeed seed;
  seed.fv = fvals;
  seed.maxf = maxf_dev;
  seed.rec12 = rec12;
  seed.gloss = gloss_dev;
  seed.est = est;
  return flow_backward_impl(f, planes_hist, ld, 1, N, w_has_jac, params, bnsave, seed, gparams, nullptr, workspace, workspace_bytes, stream);
}

// ---- G minibatches of one optimizer step per call (manager.py:219-270: the loss is their mean and backward runs once): the
//      minibatches are blockIdx.y of every launch, each with its own train-mode BatchNorm statistics.
//      planes_hist (G, n_cells + 1, d + 1, ld), bnsave (G, nis_flow_bnsave_count) (eval mode: one record), xj_out (G, N, d + 1) optional.
extern "C" int nis_flow_forward_hist_batched(nis_flow* f, const void* w, int32_t w_dtype, int32_t n_batches, int64_t N,
                                             const double* const* params, double* const* running, long long* const* tracked,
                                             int32_t bn_mode, int64_t pack_token, double* planes_hist, int64_t ld, double* bnsave,
                                             double* xj_out, void* workspace, size_t workspace_bytes, void* stream) {
  if (!planes_hist || !bnsave || n_batches < 1) return NIS_E_ARG;
  RunOpts o;
  o.G = n_batches;
  o.bn_mode = bn_mode;
  o.pack_token = pack_token;
  o.tracked = tracked;
  o.xj_out = xj_out;
  o.hist = 1;
  o.bnsave = bnsave;
  return flow_run(f, w, w_dtype, N, params, running, planes_hist, ld, workspace, workspace_bytes, (hipStream_t)stream, o);
}

// fvals (G, N), rec12 (G, 12) of nis_loss_terms_batched; gloss_dev: device scalar applied to every minibatch's loss (1 / n_minibatches);
// gparams[c] = sum over the minibatches of gloss * d loss_g / d(parameter blob of cell c).
extern "C" int nis_flow_backward_loss_batched(nis_flow* f, const double* planes_hist, int64_t ld, int32_t n_batches, int64_t N,
                                              const double* const* params, const double* bnsave, int32_t bn_mode, const double* fvals,
                                              const double* maxf_dev, const double* rec12, const double* gloss_dev, int32_t est,
                                              double* const* gparams, void* workspace, size_t workspace_bytes, void* stream) {
  if (!fvals || !maxf_dev || !rec12 || !gloss_dev) return NIS_E_ARG;
  BwdSeed seed;
  seed.fv = fvals;
  seed.maxf = maxf_dev;
  seed.rec12 = rec12;
  seed.gloss = gloss_dev;
  seed.est = est;
  seed.eval_bn = bn_mode == NIS_BN_EVAL;
  return flow_backward_impl(f, planes_hist, ld, n_batches, N, 0, params, bnsave, seed, gparams, nullptr, workspace, workspace_bytes, stream);
}
