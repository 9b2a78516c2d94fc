// Shared declarations of the host side of libnis_hip.so: the C ABI's translation units are split by concern --
//   api.hip          launch planning (flow_run), the entry points, the adjoint driver, per-launch timing
//   api_pack.hip     model handle, workspace layout, parameter packing (k_pack, k_pack_mfma)
//   api_ingest.hip   row-major w -> column planes (+ BN_0 record riders), in-kernel Philox draws, emit / scatter
//   api_moments.hip  moment reductions, built-in integrands on planes / rows, launch-set epilogue, record merge, loss terms
// Kernels are launched through the host functions declared here (no relocatable device code between the units).
#pragma once
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "flow_kernels.h"

int nis_fail_hip(hipError_t e, const char* what);          // records the message for nis_last_error, returns NIS_E_HIP
#define HIP_OK(call)                                       \
  do {                                                     \
    hipError_t e_ = (call);                                \
    if (e_ != hipSuccess) return nis_fail_hip(e_, #call);  \
  } while (0)

// sizes
static inline int cell_T(const nis_cell_desc& c) { return c.d - c.P; }
static inline int cell_out(const nis_cell_desc& c) { return c.kind == NIS_PWQUAD ? 2 * c.K + 1 : c.kind == NIS_AFFINE ? 2 : c.K; }
static inline bool cell_hidden_bias(const nis_cell_desc& c) { return c.kind == NIS_AFFINE; }
static inline int cell_R(const nis_cell_desc& c) { return cell_T(c) * cell_out(c); }
static inline int cell_feat(const nis_cell_desc& c, int l) { return l == 0 ? c.P : c.width[l - 1]; }

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
};

static const size_t WS_HEADER = 256;   // bytes: status word + padding
size_t nis_flow_layout(const nis_flow* f, int G, std::vector<CellLayout>* lay, size_t* off_fpart);

int nis_device_cus();
int nis_grid_for(int blocks_per_cu, int G, int64_t N, int tpb);
int nis_env_int(const char* name, int dflt);
static inline int nis_stream_grid(int64_t N) { return (int)std::max<int64_t>(1, std::min<int64_t>(NIS_MAXB, (N + NIS_TPB - 1) / NIS_TPB)); }

// ---- api_pack.hip
// one cell's parameter images (api_pack.hip, k_pack_cells)
struct PackJob {
  const double *prm, *run;                                   // the cell's parameter blob; running statistics (eval mode)
  double *wp, *wlast, *bias, *gb, *hb, *coef, *bnsave;       // hb, bnsave optional; coef: eval mode
  double *wm_all, *wm_last, *bias_pad;                       // matrix-core fragment images (nullptr: none)
  int WM, P, L, R, RT, width[NIS_MAX_HIDDEN];
  int eval_mode, hidden_bias, affT, regmode, T, K;
};
#define NIS_PACK_CELLS 16
struct PackJobs {
  PackJob j[NIS_PACK_CELLS];
};
hipError_t nis_launch_pack_cells(const std::vector<PackJob>& jobs, hipStream_t s);

// ---- api_ingest.hip
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
  int cov;                         // 1 (in-kernel draws, matrix-core first cell with 2 <= P <= 8): the records also carry the
                                   //   co-moments of the pass-through block in k_bn_stats0_cov's format (CellK::dcov)
};
// row-major w (n_batches, N, wcols) -> planes, with the BN_0 records of the first cell riding along; `direct`: the d = 2
// direct-row chain reads w itself, only the records are gathered
hipError_t nis_launch_ingest(const void* w, int w_dtype, int wcols, long N, double* planes, long ld, const IngestStats& st, int grid,
                             int Gy, bool direct, hipStream_t s);
// in-kernel draws of the d = 2 direct-row chain: float32 rows (plane_ld == 0) or two float32 column planes written to `rows`
hipError_t nis_launch_ingest2_rng(unsigned long long seed, unsigned long long stream0, const unsigned long long* state, long N,
                                  float* rows, long rows_gs, long plane_ld, float* w_out, const IngestStats& st, int grid, int Gy,
                                  hipStream_t s);
// in-kernel draws of any other model: the d float64 planes of batch g are GENERATED (float32 lattice values), BN_0 records (with the
// co-moments when st.cov) ride along; w_out optional (n_batches, N, d) float32
hipError_t nis_launch_ingest_rng(unsigned long long seed, unsigned long long stream0, const unsigned long long* state, long N, int d,
                                 double* planes, long ld, float* w_out, const IngestStats& st, int grid, int Gy, hipStream_t s);
hipError_t nis_launch_philox_rows(unsigned long long seed, unsigned long long stream0, int G, long N, int d, float* out, hipStream_t s);
hipError_t nis_launch_sync_reduce(const double* part, int nparts, int WM, double* out, hipStream_t s);
hipError_t nis_launch_scatter_rows(const double* src, long N, int cols, const ColMap& m, double* planes, long ld, int grid, hipStream_t s);
hipError_t nis_launch_emit(const double* planes, long ld, long N, int d, const ColMap& m, double* xj, int grid, hipStream_t s, int G = 1,
                           long plane_gs = 0);

// ---- api_moments.hip
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
hipError_t nis_launch_group_epilogue(const EpiJob& job, int blocks, hipStream_t s);
// per-batch merge of the workgroup records (count, mean, M2, max): partials (G, maxb, 4) -> out4 (G, 4)
hipError_t nis_launch_moments_final(const double* partials, int nparts, double* out4, int maxb, int G, hipStream_t s);
// first launch of nis_flow_backward_loss: dL/dJ of the minibatch loss -> gradient planes (rec12 = nis_loss_terms' records)
hipError_t nis_launch_loss_cotangent(const double* fv, const double* jac, long jstride, long N, const double* maxf, const double* rec12,
                                     const double* gloss, int est, int d, double* g, long ld, hipStream_t s, int G = 1, long jac_gs = 0,
                                     long g_gs = 0);
