// Shared declarations of the adjoint kernels: flow_bwd.hip (narrow conditioners, fp64 VALU) and
// flow_bwd_wm.hip (wide conditioners, v_mfma_f64_16x16x4_f64).
#pragma once
#include "flow_kernels.h"

#define BNACC 80              // record length (>= WM*WM + 2*WM for WM = 8, >= BRC*(WM+1))
#define BRC 8                 // rows of the last layer per wlast chunk
#define BMAXB 1024            // max workgroups of an adjoint launch
#define BKMAX 32              // max PWQuad bins of the adjoint kernels (2K + 1 in-place dZ values per lane)

struct BwdK {
  CellK c;                    // same per-cell view as the forward (in_plane, sizes, packed weights)
  const double* x;            // plane set of the cell's input  (d+1 planes, leading dimension ld)
  const double* xo;           // plane set of the cell's output (J_out lives in plane d)
  double* g;                  // gradient planes (d+1): in = dL/d(output set), out = dL/d(input set), in place
  double* dyp;                // WM planes: dL/d(BN_l output) of the layer being processed
  double* alp;                // WM planes: a_L (input of the last Linear)
  double* dzp;                // R planes: dL/dZ
  const double* bn;           // saved (L+1) x 4 x WM: scale | shift | mean | rstd
  double* part;               // records of this launch
  const double* sums;         // reduced record of the previous launch (BNACC)
  long ld, N;
  long ldw;                   // leading dimension of the workspace planes dyp / alp / dzp (wide adjoint: multiple of 16)
  double invN;                // 1 / N with train-mode BatchNorm; 0 in eval mode: dx = scale dy, the batch-mean terms of the
                              //   train-mode backward (mean(dy), xhat mean(dy xhat)) vanish, d gamma / d beta keep their sums
  // ---- batched minibatches (narrow adjoint, flow_bwd.hip): blockIdx.y = g selects one of G independent minibatches
  int G;
  long x_gs;                  // doubles between the history plane sets (x, xo) of consecutive minibatches
  long g_gs;                  // ... between their gradient plane sets
  long wsp_gs;                // ... between their workspace planes (dyp / alp / dzp)
  long bn_gs;                 // ... between their saved BatchNorm records (0 in eval mode: running statistics, one record)
  const double* prev;         // records of the launch that produced dy of the layer being processed: (G, nprev, BNACC); the consumer
  int nprev, prev_off;        //   merges the 2 WM batch sums at prev_off in its prologue (no reduction launch in between)
  long part_gs, sums_gs;      // wide adjoint (flow_bwd_wm.hip, minibatch = blockIdx.z): doubles between the minibatches' record areas / reduced sums
};

// ---- wide conditioners (register width 16 / 32 / 64), one translation unit per width
template <int WM>
bool bwd_wide_supported(const nis_cell_desc& cd);
template <int WM>
size_t bwd_wide_workspace_doubles(const nis_cell_desc& cd, int64_t N, int G);
template <int WM>
int bwd_wide_run(const CellK& c, BwdK k, const MfmaW& mw, double* ws, double* gparams, hipStream_t s);
#define NIS_DECL_BWD_WM(W)                                                                            \
  template <>                                                                                         \
  bool bwd_wide_supported<W>(const nis_cell_desc& cd);                                                \
  template <>                                                                                         \
  size_t bwd_wide_workspace_doubles<W>(const nis_cell_desc& cd, int64_t N, int G);                    \
  template <>                                                                                         \
  int bwd_wide_run<W>(const CellK& c, BwdK k, const MfmaW& mw, double* ws, double* gparams, hipStream_t s);
NIS_DECL_BWD_WM(16)
NIS_DECL_BWD_WM(32)
NIS_DECL_BWD_WM(64)
#undef NIS_DECL_BWD_WM
