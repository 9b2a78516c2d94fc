// Model handle, workspace layout and parameter packing of libnis_hip.so (see api_internal.h for the split).
#include "api_internal.h"

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

// Workspace body for G batched iterations: per cell the zero-padded parameter image (shared by all batches),
// then per-batch BN coefficients, partial records and batch statistics; finally the integrand's moment records.
size_t nis_flow_layout(const nis_flow* f, int G, std::vector<CellLayout>* lay, size_t* off_fpart) {
  size_t off = 0;
  for (int i = 0; i < f->n_cells; ++i) {
    const nis_cell_desc& c = f->cells[i];
    int m = c.P;
    for (int l = 0; l < c.n_hidden; ++l) m = std::max(m, c.width[l]);
    CellLayout L;
    L.WM = nis_pick_wm(m);
    L.KM = c.kind == NIS_PWQUAD ? nis_pick_km(c.K) : 4;
    if (c.kind == NIS_PWQUAD && L.KM == c.K && L.KM <= 32) L.KM = -L.KM;        // K == KM: guard-free specialisation
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
  return WS_HEADER + nis_flow_layout(f, G, nullptr, nullptr) * sizeof(double);
}
// ---------------------------------------------------------------------------------------------
// k_pack_cells: reference-order parameter blobs -> the images the kernels read, for up to NIS_PACK_CELLS cells per launch (grid.y):
//   zero-padded WM-strided weights wp / wlast / bias, the BatchNorm affine gb (+ hb, + eval-mode coefficients and the adjoint's saved
//   record), and for the matrix-core kernels (wm_all != nullptr) the per-(row tile, k-step) fragments of 64 contiguous doubles, lane l
//   holding W[16 jt + (l & 15)][4 q + (l >> 4)] (the A operand of v_mfma_f64_16x16x4_f64), all taken from the blob directly.
//   affT > 0 (NIS_AFFINE with T transformed columns): the last layer's rows [scale | shift][T] are interleaved to [t][scale, shift], so
//   that a column's outputs are contiguous like those of the piecewise cells.  regmode == NIS_PACK_REL: the relative image of a PWQuad
//   last layer (nis_common.h, nis_rel_row).
// One launch per optimizer step and direction instead of two per cell (the launches were 5 us each, in series: 0.17 ms of a C5 step).
__global__ __launch_bounds__(NIS_TPB) void k_pack_cells(PackJobs J) {
  const PackJob& c = J.j[blockIdx.y];
  const int tid = blockIdx.x * NIS_TPB + threadIdx.x, nth = gridDim.x * NIS_TPB;
  const int WM = c.WM;
  size_t off = 0;
  int roff = 0;
  auto feat = [&](int l) { return l == 0 ? c.P : c.width[l - 1]; };
  for (int l = 0; l <= c.L; ++l) {
    const int F = feat(l);
    const int O = l < c.L ? c.width[l] : c.R;
    const double* gam = c.prm + off;
    const double* bet = c.prm + off + F;
    const double* W = c.prm + off + 2 * F;
    // bias of the Linear that FEEDS this BN layer (NIS_AFFINE): the kernels evaluate the Linear without it -- a constant
    // in front of a BatchNorm cancels -- and account for it where it is visible: the running mean and the eval-mode shift
    const double* hbin = (c.hidden_bias && l >= 1) ? c.prm + off - F : nullptr;
    for (int j = tid; j < WM; j += nth) {
      const double g = j < F ? gam[j] : 0.0, b = j < F ? bet[j] : 0.0;
      const double hbv = (hbin && j < F) ? hbin[j] : 0.0;
      c.gb[(l * 2 + 0) * WM + j] = g;
      c.gb[(l * 2 + 1) * WM + j] = b;
      if (c.hb) c.hb[l * WM + j] = hbv;
      if (c.eval_mode) {                     // BatchNorm1d.eval(): running statistics
        double sc = 0.0, sh = 0.0;
        if (j < F) {
          sc = g / sqrt(c.run[roff + F + j] + NIS_BN_EPS);
          sh = b - (c.run[roff + j] - hbv) * sc;
        }
        c.coef[(l * 2 + 0) * WM + j] = sc;
        c.coef[(l * 2 + 1) * WM + j] = sh;
        if (c.bnsave) {                      // eval-mode forward with history: what the adjoint reads (scale | shift | mean | rstd)
          c.bnsave[(l * 4 + 0) * WM + j] = sc;
          c.bnsave[(l * 4 + 1) * WM + j] = sh;
          c.bnsave[(l * 4 + 2) * WM + j] = j < F ? c.run[roff + j] - hbv : 0.0;
          c.bnsave[(l * 4 + 3) * WM + j] = j < F ? 1.0 / sqrt(c.run[roff + F + j] + NIS_BN_EPS) : 0.0;
        }
      }
    }
    if (l < c.L) {
      double* dst = c.wp + (size_t)l * WM * WM;
      for (int e = tid; e < WM * WM; e += nth) {
        const int j = e / WM, i = e % WM;
        dst[e] = (j < O && i < F) ? W[(size_t)j * F + i] : 0.0;
      }
    } else {
      for (size_t e = tid; e < (size_t)c.R * WM; e += nth) {
        const int r = (int)(e / WM), i = (int)(e % WM);
        c.wlast[e] = (i < F) ? W[(size_t)r * F + i] : 0.0;
      }
      const double* bs = W + (size_t)O * F;
      for (int r = tid; r < c.R; r += nth) c.bias[r] = bs[r];
    }
    off += 2 * (size_t)F + (size_t)O * F;
    if (l < c.L && c.hidden_bias) off += O;
    roff += 2 * F;
  }
  if (!c.wm_all) return;
  const int KQ = WM / 4;
  size_t offw = 0;                                          // blob offset of layer l's weight matrix
  for (int l = 0; l <= c.L; ++l) {
    const int F = feat(l), O = l < c.L ? c.width[l] : c.R;
    offw += 2 * (size_t)F;
    if (l == c.L) break;
    for (size_t e = tid; e < (size_t)WM * WM; e += nth) {
      const int lane = (int)(e & 63);
      const size_t frag = e >> 6;
      const int q = (int)(frag % KQ), jt = (int)(frag / KQ);
      const int j = 16 * jt + (lane & 15), k = 4 * q + (lane >> 4);
      c.wm_all[(size_t)l * WM * WM + e] = (j < O && k < F) ? c.prm[offw + (size_t)j * F + k] : 0.0;
    }
    offw += (size_t)O * F + (c.hidden_bias ? O : 0);
  }
  const int FL = feat(c.L);
  const double* WL = c.prm + offw;
  const double* bs = WL + (size_t)c.R * FL;
  auto rows_of = [&](int j, int* js) {                      // packed row j <- blob row (minus blob row *js); < 0: a zero row
    int jo = c.affT > 0 ? (j & 1) * c.affT + (j >> 1) : j;
    *js = -1;
    if (c.regmode == NIS_PACK_REL) jo = nis_rel_row(c.T, c.K, j, js);
    else if (j >= c.R) return -1;
    return (jo >= 0 && jo < c.R) ? jo : -1;
  };
  const size_t nl = (size_t)c.RT * KQ * 64;
  for (size_t e = tid; e < nl; e += nth) {
    const int lane = (int)(e & 63);
    const size_t frag = e >> 6;
    const int q = (int)(frag % KQ), jt = (int)(frag / KQ);
    const int j = 16 * jt + (lane & 15), k = 4 * q + (lane >> 4);
    int js;
    const int jo = rows_of(j, &js);
    double v = (jo >= 0 && k < FL) ? WL[(size_t)jo * FL + k] : 0.0;
    if (js >= 0 && k < FL) v -= WL[(size_t)js * FL + k];
    c.wm_last[e] = v;
  }
  for (size_t e = tid; e < (size_t)c.RT * 16; e += nth) {
    int js;
    const int jo = rows_of((int)e, &js);
    double v = jo >= 0 ? bs[jo] : 0.0;
    if (js >= 0) v -= bs[js];
    c.bias_pad[e] = v;
  }
}

// jobs of any number of cells: NIS_PACK_CELLS per launch
hipError_t nis_launch_pack_cells(const std::vector<PackJob>& jobs, hipStream_t s) {
  for (size_t at = 0; at < jobs.size(); at += NIS_PACK_CELLS) {
    PackJobs J;
    const int n = (int)std::min<size_t>(NIS_PACK_CELLS, jobs.size() - at);
    size_t most = 0;
    for (int i = 0; i < n; ++i) {
      J.j[i] = jobs[at + i];
      most = std::max(most, (size_t)std::max(jobs[at + i].R, jobs[at + i].RT * 16) * jobs[at + i].WM);
    }
    const int grid = (int)std::min<size_t>(32, (most + NIS_TPB - 1) / NIS_TPB);
    hipLaunchKernelGGL(k_pack_cells, dim3(std::max(1, grid), n), dim3(NIS_TPB), 0, s, J);
  }
  return hipGetLastError();
}
