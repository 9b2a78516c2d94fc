// Reductions of libnis_hip.so: (count, mean, M2, max) records, built-in integrands on planes / rows, the launch-set epilogue
// (running-statistics updates in batch order + record merge), the multi-rank record merge (see api_internal.h for the split).
#include "api_internal.h"

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
  const int grid = nis_stream_grid(N);
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
  const int grid = nis_stream_grid(N);
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
  hipLaunchKernelGGL(k_integrand_rows, dim3(nis_stream_grid(N)), dim3(NIS_TPB), 0, (hipStream_t)stream, x, (long)row_stride, (long)N, a, out);
  HIP_OK(hipGetLastError());
  return NIS_OK;
}

hipError_t nis_launch_group_epilogue(const EpiJob& job, int blocks, hipStream_t s) {
  hipLaunchKernelGGL(k_group_epilogue, dim3(blocks), dim3(NIS_TPB), 0, s, job);
  return hipGetLastError();
}

hipError_t nis_launch_moments_final(const double* partials, int nparts, double* out4, int maxb, int G, hipStream_t s) {
  hipLaunchKernelGGL(k_moments_final, dim3(G), dim3(NIS_TPB), 0, s, partials, nparts, out4, maxb);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Training-loop glue of one minibatch (manager.py:223-268) as TWO launches instead of ~15 elementwise torch kernels and three
// separate moment reductions: with fv_i = f(X_i) (or f(w_i) during the preburn, :237-239) and J_i the flow's Jacobian,
//   a_i = fv_i J_i  (preburn: fv_i)        -> integ / err bookkeeping   (:241-249: mean and unbiased variance)
//   u_i = fv_i J_i / maxf                  -> loss: var(u) ("var", :254-255) or mean((u maxf)^2) ("est", :257-258)
//   u_i^2                                  -> variance of the mean (:268)
// k_loss_terms reduces the three series in ONE pass (per-workgroup records, fixed-order merge in k_loss_final) and evaluates a
// built-in integrand on the rows of X itself (fv_out keeps the values for the cotangent); k_loss_cotangent (api.hip: first launch
// of nis_flow_backward_loss) turns the final records into dL/dJ_i, written straight to the adjoint's gradient planes.
struct LossSrc {
  const double* fv_in;      // given integrand values (G, N) (arbitrary callable), or null:
  const double* xbase;      //   element (column c, sample i) of minibatch g at xbase[g * x_gs + out_col[c] * xcs + i * xss]: the rows of X
  long xcs, xss, x_gs;      //   (or of w during the preburn) or the column planes of the model output, for the built-in integrand `ia`
  double* fv_out;           // (G, N) receives the evaluated values (needed again by the cotangent)
  const double* jac;        // J_i of minibatch g at jac[g * jac_gs + i * jstride]
  long jstride, jac_gs;
  const double* maxf;       // device scalar
  int preburn;
};
// grid (records per minibatch, G).  Record of workgroup b, minibatch g: partials[((g * gridDim.x + b) * 3 + q) * 4 ...], q = a | u | u^2
__global__ __launch_bounds__(NIS_TPB) void k_loss_terms(LossSrc src, IntegrandArgs ia, long N, double* __restrict__ partials) {
  __shared__ double wm[3][NIS_TPB / NIS_WAVE][4];
  double S1[3] = {0, 0, 0}, S2[3] = {0, 0, 0}, sh[3] = {0, 0, 0}, mx[3] = {-__builtin_huge_val(), -__builtin_huge_val(), -__builtin_huge_val()};
  double cnt = 0.0;
  bool first = true;
  const int g = blockIdx.y;
  const double imax = 1.0 / src.maxf[0];
  const double* __restrict__ fvi = src.fv_in ? src.fv_in + (size_t)g * N : nullptr;
  double* __restrict__ fvo = src.fv_out ? src.fv_out + (size_t)g * N : nullptr;
  const double* __restrict__ xb = src.xbase ? src.xbase + (size_t)g * src.x_gs : nullptr;
  const double* __restrict__ jac = src.jac + (size_t)g * src.jac_gs;
  const long stride = (long)gridDim.x * NIS_TPB;
  const long nit = (N + stride - 1) / stride;
  for (long it = 0; it < nit; ++it) {
    const long i = it * stride + (long)blockIdx.x * NIS_TPB + threadIdx.x;
    const bool valid = i < N;
    double v[3] = {0.0, 0.0, 0.0};
    if (valid) {
      double fv;
      if (fvi) {
        fv = fvi[i];
      } else {
        fv = eval_integrand_at(ia, xb, src.xcs, src.xss, i);
        fvo[i] = fv;
      }
      const double fj = fv * jac[i * src.jstride];
      v[0] = src.preburn ? fv : fj;
      v[1] = fj * imax;
      v[2] = v[1] * v[1];
    }
    if (first) {
#pragma unroll
      for (int q = 0; q < 3; ++q) sh[q] = wave_first(v[q]);
      first = false;
    }
    if (valid) {
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const double t = v[q] - sh[q];
        S1[q] += t;
        S2[q] = fma(t, t, S2[q]);
        mx[q] = fmax(mx[q], v[q]);
      }
      cnt += 1.0;
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const double nw = wave_sum(cnt);
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const double s1 = wave_sum(S1[q]), s2 = wave_sum(S2[q]), m = wave_max(mx[q]);
    if (lane == 0) {
      wm[q][wave][0] = nw;
      wm[q][wave][1] = nw > 0.0 ? sh[q] + s1 / nw : 0.0;
      wm[q][wave][2] = nw > 0.0 ? fmax(s2 - s1 * s1 / nw, 0.0) : 0.0;
      wm[q][wave][3] = m;
    }
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    const int q = threadIdx.x;
    Mom t = {0.0, 0.0, 0.0};
    double m = -__builtin_huge_val();
    for (int w = 0; w < NIS_TPB / NIS_WAVE; ++w) {
      Mom r = {wm[q][w][0], wm[q][w][1], wm[q][w][2]};
      t = mom_merge(t, r);
      m = fmax(m, wm[q][w][3]);
    }
    double* part = partials + (((size_t)g * gridDim.x + blockIdx.x) * 3 + q) * 4;
    part[0] = t.n;
    part[1] = t.mean;
    part[2] = t.m2;
    part[3] = m;
  }
}

// one workgroup per minibatch (blockIdx.x = g), wave q merges series q: every lane folds its share of the records as sums about
// record 0's mean (no divisions; the records' means differ by ~sigma / sqrt(n)), then a wave reduction -- fixed order.
//   rec (G, 12) = 3 x {n, mean, M2, max} (a | u | u^2), then the scalars the host loop adds up:
//   out (G, 8): out[0] = mean(a), out[1] = var(a) (unbiased), out[2] = loss, out[3] = var(u^2) (unbiased), out[4] = n, out[5] = mean(u)
__global__ __launch_bounds__(NIS_TPB) void k_loss_final(const double* __restrict__ partials, int nparts, const double* __restrict__ maxf, int est,
                                                        double* __restrict__ rec, double* __restrict__ out) {
  const int g = blockIdx.x;
  partials += (size_t)g * nparts * 12;
  rec += (size_t)g * 12;
  out += (size_t)g * 8;
  const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
  if (q < 3) {
    const double ref = partials[(size_t)q * 4 + 1];
    double an = 0.0, a1 = 0.0, a2 = 0.0, m = -__builtin_huge_val();
    for (int b = lane; b < nparts; b += 64) {
      const double* p = partials + ((size_t)b * 3 + q) * 4;
      const double n = p[0], dl = p[1] - ref;
      an += n;
      a1 = fma(n, dl, a1);
      a2 += fma(n * dl, dl, p[2]);
      m = fmax(m, p[3]);
    }
    an = wave_sum(an);
    a1 = wave_sum(a1);
    a2 = wave_sum(a2);
    m = wave_max(m);
    if (lane == 0) {
      rec[q * 4 + 0] = an;
      rec[q * 4 + 1] = ref + a1 / an;
      rec[q * 4 + 2] = fmax(a2 - a1 * a1 / an, 0.0);
      rec[q * 4 + 3] = m;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const double n = rec[0];
    out[0] = rec[1];
    out[1] = rec[2] / (n - 1.0);
    out[2] = est ? (maxf[0] * maxf[0]) * rec[9] : rec[6] / (n - 1.0);      // mean((u maxf)^2)  |  unbiased var(u)
    out[3] = rec[10] / (n - 1.0);
    out[4] = n;
    out[5] = rec[5];
    out[6] = out[7] = 0.0;
  }
}

static int loss_grid(int G, int64_t N) { return std::min(nis_stream_grid(N), std::max(1, 2 * nis_device_cus() / G)); }
extern "C" size_t nis_loss_workspace_bytes_batched(int32_t n_batches, int64_t N) {
  if (n_batches < 1 || N < 1) return 0;
  return (size_t)n_batches * loss_grid(n_batches, N) * 12 * sizeof(double);
}
extern "C" size_t nis_loss_workspace_bytes(void) { return (size_t)NIS_MAXB * 12 * sizeof(double); }

// G minibatches per call: fv_in / fv_out (G, N), rec12 (G, 12), out8 (G, 8).  Built-in integrand (fv_in == null): element
// (column c, sample i) of minibatch g is read at xbase[g * x_batch_stride + x_cols[c] * x_col_stride + i * x_row_stride] -- the
// rows of a (G, N, d) matrix (col stride 1, row stride d) or the column planes of the model output (col stride ld, row stride 1,
// x_cols = the model's output column map; null: identity).
extern "C" int nis_loss_terms_batched(const double* fv_in, const double* xbase, int64_t x_col_stride, int64_t x_row_stride,
                                      int64_t x_batch_stride, const int32_t* x_cols_host, int32_t d, int32_t f_kind,
                                      const double* f_params_host, double* fv_out, const double* jac, int64_t jstride, int64_t jac_batch_stride,
                                      int32_t n_batches, int64_t N, const double* maxf_dev, int32_t preburn, int32_t est, double* rec12,
                                      double* out8, void* workspace, size_t workspace_bytes, void* stream) {
  if (!jac || !maxf_dev || !rec12 || !out8 || N < 1 || jstride < 1 || n_batches < 1 || n_batches > 65535) return NIS_E_ARG;
  if (!fv_in && (!xbase || !fv_out || d < 1 || d > NIS_MAX_DIM)) return NIS_E_ARG;
  if (!fv_in && f_kind != NIS_F_CAMEL && f_kind != NIS_F_NORMAL && f_kind != NIS_F_ONE) return NIS_E_ARG;
  if (!fv_in && f_kind == NIS_F_CAMEL && d < 2) return NIS_E_ARG;
  if (!workspace || workspace_bytes < nis_loss_workspace_bytes_batched(n_batches, N)) return NIS_E_WORKSPACE;
  LossSrc src;
  memset(&src, 0, sizeof(src));
  src.fv_in = fv_in;
  src.xbase = xbase;
  src.xcs = (long)x_col_stride;
  src.xss = (long)x_row_stride;
  src.x_gs = (long)x_batch_stride;
  src.fv_out = fv_out;
  src.jac = jac;
  src.jstride = (long)jstride;
  src.jac_gs = (long)jac_batch_stride;
  src.maxf = maxf_dev;
  src.preburn = preburn;
  IntegrandArgs a;
  memset(&a, 0, sizeof(a));
  for (int c = 0; c < d && c < NIS_MAX_DIM; ++c) a.out_col[c] = x_cols_host ? x_cols_host[c] : c;
  a.d = d;
  a.kind = f_kind;
  for (int k = 0; k < 4; ++k) a.p[k] = f_params_host ? f_params_host[k] : 0.0;
  hipStream_t s = (hipStream_t)stream;
  // two workgroups per CU over all minibatches: the pass streams 16-24 B per sample, and every record costs the merge three updates
  const int grid = loss_grid(n_batches, N);
  {
    NisProfScope ps(NIS_K_MOMENTS, s);
    hipLaunchKernelGGL(k_loss_terms, dim3(grid, n_batches), dim3(NIS_TPB), 0, s, src, a, (long)N, (double*)workspace);
  }
  HIP_OK(hipGetLastError());
  hipLaunchKernelGGL(k_loss_final, dim3(n_batches), dim3(NIS_TPB), 0, s, (const double*)workspace, grid, maxf_dev, est, rec12, out8);
  HIP_OK(hipGetLastError());
  return NIS_OK;
}

extern "C" int nis_loss_terms(const double* fv_in, const double* xrows, int64_t xstride, int32_t d, int32_t f_kind, const double* f_params_host,
                              double* fv_out, const double* jac, int64_t jstride, int64_t N, const double* maxf_dev, int32_t preburn,
                              int32_t est, double* rec12, double* out8, void* workspace, size_t workspace_bytes, void* stream) {
  if (!fv_in && xstride < d) return NIS_E_ARG;
  return nis_loss_terms_batched(fv_in, xrows, 1, xstride, 0, nullptr, d, f_kind, f_params_host, fv_out, jac, jstride, 0, 1, N, maxf_dev, preburn,
                                est, rec12, out8, workspace, workspace_bytes, stream);
}

// dL/dJ_i of the variance ("var") or second-moment ("est") loss of minibatch g = blockIdx.y, times the upstream cotangent gloss
// (device scalar), into gradient plane d of that minibatch; the d coordinate planes get zeros (f is evaluated on the detached X,
// manager.py:244).  fv (G, N), rec (G, 12), jac at jac[g * jac_gs + i * jstride], gradient planes at g[g * g_gs + ...].
__global__ __launch_bounds__(NIS_TPB) void k_loss_cotangent(const double* __restrict__ fv, const double* __restrict__ jac, long jstride, long N,
                                                            const double* __restrict__ maxf, const double* __restrict__ rec,
                                                            const double* __restrict__ gloss, int est, int d, double* __restrict__ g, long ld,
                                                            long jac_gs, long g_gs) {
  const int gi = blockIdx.y;
  fv += (size_t)gi * N;
  jac += (size_t)gi * jac_gs;
  rec += (size_t)gi * 12;
  g += (size_t)gi * g_gs;
  const double imax = 1.0 / maxf[0], n = rec[4], mean_u = rec[5], gl = gloss[0];
  const double cvar = 2.0 / (n - 1.0) * gl, cest = 2.0 / n * gl;
  const long stride = (long)gridDim.x * NIS_TPB;
  for (long i = (long)blockIdx.x * NIS_TPB + threadIdx.x; i < N; i += stride) {
    const double f = fv[i], J = jac[i * jstride];
    const double gj = est ? cest * (f * J) * f : cvar * (f * J * imax - mean_u) * (f * imax);
    for (int p = 0; p < d; ++p) g[(size_t)p * ld + i] = 0.0;
    g[(size_t)d * ld + i] = gj;
  }
}

hipError_t nis_launch_loss_cotangent(const double* fv, const double* jac, long jstride, long N, const double* maxf, const double* rec12,
                                     const double* gloss, int est, int d, double* g, long ld, hipStream_t s, int G, long jac_gs, long g_gs) {
  hipLaunchKernelGGL(k_loss_cotangent, dim3(nis_grid_for(8, G, N, NIS_TPB), G), dim3(NIS_TPB), 0, s, fv, jac, jstride, N, maxf, rec12, gloss,
                     est, d, g, ld, jac_gs, g_gs);
  return hipGetLastError();
}
