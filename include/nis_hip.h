/*
 * nis_hip.h -- C ABI of libnis_hip.so: MI355X (gfx950) kernels for the neural-importance-sampling
 * hot path of NGoetz/NF (`nisrep`).
 *
 * The reference is pure Python on PyTorch and has no FFI / plugin registry; the boundary a
 * maintainer would bind is therefore the set of tensor-level operations its Python methods
 * perform.  Each entry point below names the reference code it replaces (paths relative to
 * /root/reference/nisrep).  INTEGRATION.md shows the ctypes stubs.
 *
 * Conventions
 *   - plain pointers + sizes only; every pointer is a DEVICE pointer unless it says "host".
 *   - the caller owns all buffers (inputs, outputs, workspace); nothing is allocated or freed
 *     on the launch path, so every call may be captured into a hipGraph.
 *   - all work is enqueued asynchronously on `stream` (a hipStream_t passed as void*).
 *   - return value: 0 = ok, negative = NIS_E_* below.  No exceptions cross the boundary.
 *   - all arithmetic is IEEE float64 (the reference runs in torch.double); bin indices int32.
 *
 * Data layout ("column planes")
 *   The flow state is kept as struct-of-arrays: plane p holds one variable for all N samples,
 *   planes + p*ld (ld >= N).  A permutation layer (RollLayer / MaskLayer / DeMaskLayer,
 *   normalizing_flows/layers/layers.py:6-51,80-91) never moves data: it only renames planes.
 */
#ifndef NIS_HIP_H
#define NIS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NIS_MAX_DIM     32   /* max flow dimension d (n_flow)                        */
#define NIS_MAX_HIDDEN  8    /* max hidden layers of a conditioner                    */
#define NIS_MAX_WIDTH   64   /* max hidden width / pass-through size                  */
#define NIS_MAX_BINS    128  /* max n_bins (PWLin and PWQuad; PWQuad > 32: scratch-array kernels) */
#define NIS_MAX_CELLS   64
#define NIS_MAX_LEGS    10   /* RAMBO: max number of final-state particles            */

/* Device status words at the head of a flow workspace (int32 [0], [1]).  Word 0 is cleared by the first launch of every flow call,
 * word 1 is STICKY: the kernels only ever OR into it, the caller clears it (e.g. once per integrate()) and reads it when it reads
 * its results.
 *   NIS_STATUS_PWLIN_RANGE     a PWLin input coordinate was 1.0 or negative (the reference's torch.gather raises, coupling_cells.py:136)
 *   NIS_STATUS_BIN_AMBIGUOUS   some PWQuad bin decision rested on last bits: an edge quotient cumsum(W)/sum(W) within 2^-50 (relative) of
 *                              x, or a width on the threshold of vanishing in the running sum at the end of the kept prefix (whether two
 *                              edges are EQUAL -- the reference's argmax then takes the first, coupling_cells.py:198-202 -- turns on the
 *                              last bit of exp()).  These are the samples on which the reference's own CPU and GPU builds disagree; for
 *                              every sample that does not raise the flag the selected bin is the reference's, bit for bit.
 * bins_out entries carry the same flag per (column, sample): bin | NIS_BIN_AMBIGUOUS_FLAG. */
#define NIS_STATUS_PWLIN_RANGE    1
#define NIS_STATUS_BIN_AMBIGUOUS  2
#define NIS_BIN_AMBIGUOUS_FLAG    0x40000000

enum { NIS_PWLIN = 0, NIS_PWQUAD = 1, NIS_AFFINE = 2 };
enum { NIS_BN_EVAL = 0, NIS_BN_TRAIN = 1 };
enum { NIS_F64 = 0, NIS_F32 = 1 };

enum {
  NIS_OK = 0,
  NIS_E_ARG = -1,        /* invalid argument (shape / size limits)            */
  NIS_E_UNSUPPORTED = -2,/* configuration outside the compiled kernel set     */
  NIS_E_WORKSPACE = -3,  /* workspace too small                               */
  NIS_E_HIP = -4,        /* a HIP runtime call failed (see nis_last_error)    */
  NIS_E_NODEVICE = -5
};

/* Built-in integrands (device functions), so that f(X)*J never leaves the chip.
 * NIS_F_CAMEL : README.md:35-36   exp(-|x-.75|^2/s^2)+exp(-|x-.25|^2/s^2), p0=0.75 p1=0.25 p2=0.2 (2 columns)
 * NIS_F_NORMAL: manager.py:39-40  exp(-sum (x-mu)^2/(2 sigma^2))/(sigma sqrt((2pi)^d)), p0=mu p1=sigma
 * NIS_F_ONE   : f == 1 (integrates the Jacobian; mean J -> 1)
 * NIS_F_NONE  : nis_integrate_steps[_rng] only, with keep_planes: no integrand, X / J stay in the planes for a caller-side f
 */
enum { NIS_F_NONE = -1, NIS_F_CAMEL = 0, NIS_F_NORMAL = 1, NIS_F_ONE = 2 };

/* One coupling cell (PWLin: layers/coupling_cells.py:73-142, PWQuad: :144-228, AffineCoupling: :6-70; K is
 * ignored for NIS_AFFINE, whose conditioner has 2 outputs per transformed column laid out [scale|shift][T]) with its
 * conditioner (RectNN :230-254).  `in_col[j]` is the state plane that feeds the cell's local
 * column j (0..P-1 pass-through, P..d-1 transformed): the composition of all permutation
 * layers in front of the cell.  Transformed columns are written back to the same planes. */
typedef struct nis_cell_desc {
  int32_t kind;                    /* NIS_PWLIN | NIS_PWQUAD | NIS_AFFINE                      */
  int32_t d;                       /* flow_size                                                */
  int32_t P;                       /* pass_through_size                                        */
  int32_t K;                       /* n_bins                                                   */
  int32_t n_hidden;                /* number of hidden layers L (len(NN_layers))               */
  int32_t width[NIS_MAX_HIDDEN];   /* hidden widths                                            */
  int32_t in_col[NIS_MAX_DIM];     /* plane index per local column                             */
} nis_cell_desc;

/* Parameter blob of one cell, float64, in the order of the reference's nn.Sequential
 * (coupling_cells.py:236-252):
 *   BN_0.weight[P] BN_0.bias[P] Lin_0.weight[w0*P]
 *   BN_1.weight[w0] BN_1.bias[w0] Lin_1.weight[w1*w0] ... BN_L.weight BN_L.bias
 *   Lin_L.weight[(T*out)*w_{L-1}] Lin_L.bias[T*out]          (Linear weights row-major (out,in))
 * NIS_AFFINE: the hidden Linears carry a bias too (coupling_cells.py:26-40): Lin_l.bias[w_l] follows Lin_l.weight.
 * Running-statistics blob:  for l = 0..L: running_mean[f_l] running_var[f_l]   (f_0=P, f_l=w_{l-1})
 */
size_t nis_cell_param_count(const nis_cell_desc* c);
size_t nis_cell_running_count(const nis_cell_desc* c);

typedef struct nis_flow nis_flow;   /* opaque model handle (topology only, no parameters) */

/* Build a handle for a Sequential of cells (manager.py:474-499 PWLinManager.create_model,
 * :518-600 PWQuadManager.create_model).  out_col[c] = plane holding logical output column c
 * after the last permutation layer.  Host call. */
int nis_flow_create(const nis_cell_desc* cells, int32_t n_cells, int32_t d, const int32_t* out_col,
                    nis_flow** out);
void nis_flow_destroy(nis_flow* f);

/* Exchange hook of nis_flow_forward_sync: `records` holds `world` slots of `doubles_per_rank` float64; slot `rank` has
 * just been written by work enqueued on `stream`.  The hook must leave every rank's slot filled in on every rank (an
 * in-place all-gather enqueued on, or ordered after, `stream`). */
typedef void (*nis_sync_fn)(void* ctx, double* records, int32_t doubles_per_rank, void* stream);

/* Bytes of device workspace for n_batches batched iterations (1 for nis_flow_forward); independent of N. */
size_t nis_flow_workspace_bytes(const nis_flow* f, int64_t n_batches);

/* Forward pass of the whole model on N samples: AddJacobian (layers.py:75-77) + every cell.
 *   w        (N,d) row-major, dtype w_dtype (integrate() draws float32, manager.py:390); with
 *            w_has_jac != 0 it is (N,d+1) and its last column is the incoming Jacobian, i.e. the
 *            tensor AddJacobian / a previous cell produced (coupling_cells.py:169)
 *   params   host array [n_cells] of device pointers to the parameter blobs
 *   running  host array [n_cells] of device pointers to the running-stat blobs
 *            (read in NIS_BN_EVAL; read-modify-written in NIS_BN_TRAIN, momentum 0.1, unbiased var)
 *   tracked  host array [n_cells] of device pointers to int64[L+1] num_batches_tracked counters,
 *            incremented in NIS_BN_TRAIN (may be NULL)
 *   pack_token  0: re-pack the parameter image every call; otherwise the zero-padded image kept in the
 *            workspace is reused while token, bn_mode, workspace and blob pointers are unchanged
 *            (the caller changes the token whenever parameters or running statistics changed)
 *   planes   (d+1, ld) float64 state; on return plane out_col[c] holds X[:,c], plane d holds J
 *   xj_out   optional (N,d+1) row-major float64 = the reference's model output XJ (may be NULL)
 *   bins_out optional int32 [sum_cells T_c][ld] bin index per transformed column (may be NULL)
 * bn_mode NIS_BN_TRAIN reproduces train-mode BatchNorm1d (batch statistics over all N rows),
 * which is what the reference uses during training AND integrate(). */
int nis_flow_forward(nis_flow* f, const void* w, int32_t w_dtype, int32_t w_has_jac, int64_t N,
                     const double* const* params, double* const* running, long long* const* tracked,
                     int32_t bn_mode, int64_t pack_token,
                     double* planes, int64_t ld, double* xj_out, int32_t* bins_out,
                     void* workspace, size_t workspace_bytes, void* stream);

/* n_batches iterations of BasicManager.integrate's loop body (manager.py:395-400) in ONE call: the
 * iterations are independent given fixed parameters (each is its own train-mode BatchNorm batch), so they
 * run as the y dimension of every launch and the per-launch latency is paid once for all of them.
 *   w     (n_batches, N, d) row-major;  planes (n_batches, d+1, ld);  out4 (n_batches, 4) = {count, mean, M2, max}
 * The built-in integrand f(X)*J is evaluated inside the last cell's kernel.  keep_planes == 0: X and J are
 * never written (nothing reads them); fres_out (n_batches, N) optional.  Running statistics receive the
 * n_batches momentum updates in batch order.  Workspace: nis_flow_workspace_bytes(f, n_batches).
 * f_kind == NIS_F_NONE (-1) with keep_planes != 0: no integrand -- the call only leaves X and J of the n_batches iterations in
 * `planes` (out4 is not written and may be NULL): the flow part of integrate()'s loop for an arbitrary caller-side f
 * (manager.py:397-400 with a Python callable), still one launch set for all iterations, with the draw inside in the _rng form. */
int nis_integrate_steps(nis_flow* f, const void* w, int32_t w_dtype, int32_t n_batches, int64_t N,
                        const double* const* params, double* const* running, long long* const* tracked,
                        int32_t bn_mode, int64_t pack_token, int32_t f_kind, const double* f_params_host,
                        double* planes, int64_t ld, int32_t keep_planes, double* fres_out, double* out4,
                        void* workspace, size_t workspace_bytes, void* stream);

/* The same loop body with the draw INSIDE the step (manager.py:395 `uniform_(w)` is part of integrate()'s loop): batch g
 * is drawn on the device from the counter-based generator Philox4x32-10 (Salmon et al. 2011, Random123 constants) on the
 * float32 lattice torch's uniform_ produces (manager.py:390), by the launch that gathers BN_0's records -- no input buffer,
 * no separate generator pass.  With s = stream0 + g and u24(r) = (r >> 8) * 2^-24:
 *   d == 2:  w[g][n][c] = u24(philox(ctr = {n>>1 lo, n>>1 hi, s lo, s hi}, key = {seed lo, seed hi})[2 (n & 1) + c])
 *   d != 2:  w[g][n][c] = u24(philox(ctr = {q lo, q hi, s lo, s hi}, key)[c & 3]),  q = n * ceil(d / 4) + (c >> 2)
 * The d = 2 PWQuad chain with a fused camel (nis_integrate_steps' direct-row class, nis_integrate_rng_mode == 2) keeps the
 * draws as float32 rows in the state plane it leaves unused (8 B/sample) and its cells read them from there; every other
 * model (mode 1) gets its d float64 state planes written directly, with the first cell's BN_0 records -- and, for a
 * matrix-core first cell with 2..8 pass-through columns, their co-moments, so that BN_1 is derived as well -- riding along.
 * w_out (n_batches, N, d) float32 optional: the draws that were used (tests replay them through the oracle; nis_philox_rows
 * materialises the same numbers).  planes / keep_planes as for nis_integrate_steps.
 * rng_state (optional, device): {seed, next stream} as two uint64 -- when given it overrides seed / stream0, is read by the
 * drawing launch and advanced by n_batches in the set's last launch, so a launch set captured into a hipGraph draws fresh
 * numbers on every replay. */
int nis_integrate_steps_rng(nis_flow* f, int32_t n_batches, int64_t N,
                            const double* const* params, double* const* running, long long* const* tracked,
                            int32_t bn_mode, int64_t pack_token, int32_t f_kind, const double* f_params_host,
                            uint64_t seed, uint64_t stream0, uint64_t* rng_state, float* w_out, double* planes, int64_t ld,
                            int32_t keep_planes, double* out4, void* workspace, size_t workspace_bytes, void* stream);
/* 2: direct-row class, 1: generated planes (any model), 0: invalid arguments. */
int nis_integrate_rng_mode(const nis_flow* f, int32_t n_batches, int64_t N, int32_t bn_mode, int32_t f_kind, int32_t keep_planes);
/* Test entry: ONE iteration of the fused step on the production direct-row kernels that additionally writes the bin index
 * selected for every cell's transformed column, bins_out int32 [n_cells][ld] (K = 4 PWQuad chains of that kernel class;
 * others: NIS_E_UNSUPPORTED).  The parity tests assert the bins bit-exact against the oracle at the bench's batch size. */
int nis_integrate_steps_bins(nis_flow* f, const void* w, int32_t w_dtype, int64_t N,
                             const double* const* params, double* const* running, long long* const* tracked,
                             int64_t pack_token, int32_t f_kind, const double* f_params_host,
                             double* planes, int64_t ld, int32_t* bins_out, double* out4,
                             void* workspace, size_t workspace_bytes, void* stream);
/* 1 when the built-in integrand of nis_integrate_steps is fused into the last cell's kernel for this model. */
int nis_flow_fusable(const nis_flow* f);
int nis_philox_rows(uint64_t seed, uint64_t stream0, int32_t n_batches, int64_t N, int32_t d, float* out /* (n_batches, N, d) */, void* stream);

/* Train-mode forward whose BatchNorm statistics are those of the batch that is SHARDED over `world` ranks (SURVEY.md 8(e)
 * option 1, "SyncBN-exact": every rank passes its N rows; results equal the reference's single-batch forward of the
 * concatenated batch up to summation order).  After each launch that produces BatchNorm records the rank's records are
 * reduced to one (count, mean, M2) record per feature, `sync_fn` gathers the ranks' records into `sync_buf`
 * (>= world * 64 * 3 doubles of device memory), and the consumers merge those.  Other arguments as nis_flow_forward. */
int nis_flow_forward_sync(nis_flow* f, const void* w, int32_t w_dtype, int32_t w_has_jac, int64_t N,
                          const double* const* params, double* const* running, long long* const* tracked,
                          int64_t pack_token, double* planes, int64_t ld, double* xj_out, void* workspace,
                          size_t workspace_bytes, void* stream, nis_sync_fn sync_fn, void* sync_ctx, double* sync_buf,
                          int32_t world, int32_t rank);

/* Training forward + hand-written adjoint (replaces loss.backward() through the flow, manager.py:275-278, for the
 * cell class the adjoint kernels cover: every model nis_flow_create accepts -- conditioner widths and pass-through size <= 64
 * (<= 8: fp64 VALU kernels; wider: v_mfma_f64_16x16x4_f64 kernels), up to NIS_MAX_BINS bins, any number of transformed columns;
 * nis_flow_backward_supported tells).
 *   nis_flow_forward_train: train-mode forward that keeps every cell's input: planes_hist holds n_cells+1 plane
 *     sets of (d+1, ld) doubles (set c = input of cell c, last set = model output); bnsave
 *     (nis_flow_bnsave_count doubles) receives scale | shift | batch mean | rstd of every BN layer.
 *   nis_flow_backward: gxj = dL/dXJ (N, d+1) row-major in output column order -> gparams[c] = dL/d(parameter
 *     blob of cell c) (same layout as params[c], overwritten) and optionally gx_out = dL/d(input) (N, d+1);
 *     ld must equal N here (the gradient planes in the workspace share the leading dimension). */
int nis_flow_backward_supported(const nis_flow* f);
size_t nis_flow_bnsave_count(const nis_flow* f);
int nis_flow_forward_train(nis_flow* f, const void* w, int32_t w_dtype, int32_t w_has_jac, int64_t N,
                           const double* const* params, double* const* running, long long* const* tracked,
                           int64_t pack_token, double* planes_hist, int64_t ld, double* bnsave, double* xj_out,
                           void* workspace, size_t workspace_bytes, void* stream);
/* The same pair for a forward that ran EVAL-mode BatchNorm (running statistics; the reference's autograd differentiates a model in
 * any mode): nis_flow_forward_hist takes bn_mode (NIS_BN_TRAIN: nis_flow_forward_train); with NIS_BN_EVAL bnsave receives the
 * running-statistics affine form, and nis_flow_backward_eval runs the adjoint without the batch-mean terms of the train-mode
 * BatchNorm backward (dx = scale dy; d gamma = sum dy xhat, d beta = sum dy as before). */
int nis_flow_forward_hist(nis_flow* f, const void* w, int32_t w_dtype, int32_t w_has_jac, int64_t N,
                          const double* const* params, double* const* running, long long* const* tracked, int32_t bn_mode,
                          int64_t pack_token, double* planes_hist, int64_t ld, double* bnsave, double* xj_out,
                          void* workspace, size_t workspace_bytes, void* stream);
int nis_flow_backward_eval(nis_flow* f, const double* planes_hist, int64_t ld, int64_t N, int32_t w_has_jac,
                           const double* const* params, const double* bnsave, const double* gxj, double* const* gparams,
                           double* gx_out, void* workspace, size_t workspace_bytes, void* stream);
size_t nis_flow_backward_workspace_bytes(const nis_flow* f, int64_t N);
int nis_flow_backward(nis_flow* f, const double* planes_hist, int64_t ld, int64_t N, int32_t w_has_jac,
                      const double* const* params, const double* bnsave, const double* gxj,
                      double* const* gparams, double* gx_out, void* workspace, size_t workspace_bytes, void* stream);

/* f(X)*J for a built-in integrand straight from the planes + streaming moments of the result:
 * out4 = {count, mean, M2 = sum (v-mean)^2, max}   (torch.mean/var/max at manager.py:399-400,
 * :241-249).  fres_out (N) optional.  workspace >= nis_moments_workspace_bytes(). */
int nis_integrand_moments(const double* planes, int64_t ld, int64_t N, int32_t d, const int32_t* out_col_host,
                          int32_t f_kind, const double* f_params_host, double* fres_out, double* out4,
                          void* workspace, size_t workspace_bytes, void* stream);

/* f(x_i) of a built-in integrand for the rows of a row-major (N, d) matrix with row stride `row_stride` doubles: what
 * the training loop evaluates as f(X) on the model output (manager.py:241-246; README.md:35-36 camel, manager.py:39-40
 * normal).  f_params_host: 4 doubles (camel: a, b, s; normal: mu, sigma). */
int nis_integrand_eval(int32_t f_kind, const double* f_params_host, const double* x, int64_t row_stride, int64_t N,
                       int32_t d, double* out, void* stream);

/* The training loop's glue of one minibatch (manager.py:223-268) in two launches.  With fv_i the integrand on sample i -- given
 * (fv_in, any callable evaluated by the caller) or evaluated here by a built-in f_kind on the rows xrows (X of the model output; w
 * during the preburn, :237-239; the values are kept in fv_out) -- and J_i = jac[i * jstride]:
 *   a_i = fv_i J_i (preburn: fv_i),  u_i = fv_i J_i / maxf,  u_i^2
 * are reduced in one pass: rec12 = three {count, mean, M2, max} records (a | u | u^2) and
 *   out8 = {mean(a), var(a) unbiased, loss, var(u^2) unbiased, n, mean(u), -, -},  loss = var(u) unbiased ("var", :254-255) or
 *   mean((u maxf)^2) (est != 0, :257-258).  maxf_dev: device scalar.  workspace >= nis_loss_workspace_bytes().
 * nis_flow_backward_loss: the adjoint seeded by that loss -- dL/dJ_i = gloss * d loss / d J_i from the records (no (N, d+1)
 * cotangent tensor, no scatter pass), dL/dX = 0 (f sees the detached X, :244); gloss_dev: device scalar (1 / n_minibatches). */
size_t nis_loss_workspace_bytes(void);
int nis_loss_terms(const double* fv_in, const double* xrows, int64_t xstride, int32_t d, int32_t f_kind, const double* f_params_host,
                   double* fv_out, const double* jac, int64_t jstride, int64_t N, const double* maxf_dev, int32_t preburn, int32_t est,
                   double* rec12, double* out8, void* workspace, size_t workspace_bytes, void* stream);
int nis_flow_backward_loss(nis_flow* f, const double* planes_hist, int64_t ld, int64_t N, int32_t w_has_jac,
                           const double* const* params, const double* bnsave, const double* fvals, const double* maxf_dev,
                           const double* rec12, const double* gloss_dev, int32_t est, double* const* gparams,
                           void* workspace, size_t workspace_bytes, void* stream);

/* The G minibatches of ONE optimizer step per call (manager.py:219-270: the epoch's loss is the mean of the minibatch losses and
 * loss.backward() runs once, so the minibatches are independent given the parameters): every launch carries them as its y dimension,
 * each with its own train-mode BatchNorm statistics; the reference's default regime (mini_batch_size = 2000) is launch-bound otherwise.
 *   nis_flow_forward_hist_batched: w (G, N, d); planes_hist (G, n_cells + 1, d + 1, ld); bnsave (G, nis_flow_bnsave_count) (NIS_BN_EVAL:
 *     one record); xj_out (G, N, d + 1) optional; workspace >= nis_flow_workspace_bytes(f, G).  The running statistics advance in
 *     minibatch order.
 *   nis_loss_terms_batched: fv_in / fv_out (G, N), rec12 (G, 12), out8 (G, 8).  A built-in integrand (fv_in == NULL) reads element
 *     (column c, sample i) of minibatch g at xbase[g x_batch_stride + x_cols[c] x_col_stride + i x_row_stride]: rows of a (G, N, d)
 *     matrix (strides 1, d, N d) or the output planes of the history (strides ld, 1, (n_cells + 1)(d + 1) ld; x_cols = out_col).
 *     J_i of minibatch g at jac[g jac_batch_stride + i jstride].  workspace >= nis_loss_workspace_bytes_batched(G, N).
 *   nis_flow_backward_loss_batched: gparams[c] = sum over the minibatches of gloss * d loss_g / d(parameter blob of cell c).
 *     workspace >= nis_flow_backward_workspace_bytes_batched(f, G, N).
 * Streams: every backward entry point orders ALL its work on `stream` (what the caller enqueues behind the call sees the results).  For
 *   matrix-core cells at N * n_batches >= 65 536 the last layer's weight-gradient GEMM runs on a side stream the library creates once per
 *   device (forked from and joined back into `stream` with events inside the call), so the first such call must not happen while
 *   `stream` is being captured (capture a step only after one eager run, as flow.TrainPlan does); one host thread per device. */
int nis_flow_forward_hist_batched(nis_flow* f, const void* w, int32_t w_dtype, int32_t n_batches, int64_t N,
                                  const double* const* params, double* const* running, long long* const* tracked, int32_t bn_mode,
                                  int64_t pack_token, double* planes_hist, int64_t ld, double* bnsave, double* xj_out,
                                  void* workspace, size_t workspace_bytes, void* stream);
size_t nis_loss_workspace_bytes_batched(int32_t n_batches, int64_t N);
int nis_loss_terms_batched(const double* fv_in, const double* xbase, int64_t x_col_stride, int64_t x_row_stride, int64_t x_batch_stride,
                           const int32_t* x_cols_host, int32_t d, int32_t f_kind, const double* f_params_host, double* fv_out,
                           const double* jac, int64_t jstride, int64_t jac_batch_stride, int32_t n_batches, int64_t N,
                           const double* maxf_dev, int32_t preburn, int32_t est, double* rec12, double* out8,
                           void* workspace, size_t workspace_bytes, void* stream);
size_t nis_flow_backward_workspace_bytes_batched(const nis_flow* f, int32_t n_batches, int64_t N);
int nis_flow_backward_loss_batched(nis_flow* f, const double* planes_hist, int64_t ld, int32_t n_batches, int64_t N,
                                   const double* const* params, const double* bnsave, int32_t bn_mode, const double* fvals,
                                   const double* maxf_dev, const double* rec12, const double* gloss_dev, int32_t est,
                                   double* const* gparams, void* workspace, size_t workspace_bytes, void* stream);

/* Streaming moments of a vector (optionally of the product a[i]*b[i], b may be NULL). */
size_t nis_moments_workspace_bytes(void);
int nis_moments(const double* a, const double* b, int64_t N, double* out4,
                void* workspace, size_t workspace_bytes, void* stream);
/* Merge of moment records over the shards of a multi-GPU job (nf_amd/parallel.py; the reference has no distributed code, this is
 * the one reduction of the sample-sharded path): recs (world, steps, 4) = every rank's {count, mean, M2, max} per iteration as
 * gathered by ONE all-gather, out (steps, 4).  n = sum n_r, mean = sum n_r mean_r / n, M2 = sum M2_r + sum n_r (mean_r - mean)^2,
 * max over the non-empty shards; one launch, ranks in index order (identical on every rank). */
int nis_merge_moments(const double* recs, int32_t world, int32_t steps, double* out, void* stream);

/* RAMBO on diet, FlatInvertiblePhasespace.generateKinematics_batch with pdf=None
 * (PhaseSpace/flat_phase_space_generator.py:139-308 and helpers :313-441, PhaseSpace/utils.py).
 *   r        random numbers, (N, ncol) row-major (r_ld = ncol) or column planes (r_col_major != 0,
 *            column c at r + r_cols[c]*r_ld); ncol = 3n-4 (+2 when pdf_active)
 *   momenta  optional (N, 2+n, 4) row-major, COM frame (flat_phase_space_generator.py:282,308)
 *   weight   (N)
 * flags: bit0 pdf_active, bit1 tau, bit2 "lab boost applied" decided by the caller
 *        (batch-global test of utils.py:139). */
typedef struct nis_rambo_desc {
  int32_t n_final;
  int32_t flags;
  double  initial_masses[2];
  double  final_masses[NIS_MAX_LEGS];
  double  E_cm;
  double  pT_mincut, delR_mincut, rap_maxcut;
  int32_t bisect_levels;           /* 0: Newton solver for the intermediate-mass equation; > 0: the
                                      reference's lattice bisection with this many levels (it runs 120-180) */
  int32_t reserved;
} nis_rambo_desc;
enum { NIS_RAMBO_PDF_ACTIVE = 1, NIS_RAMBO_TAU = 2, NIS_RAMBO_LAB_BOOST = 4 };

int nis_rambo_forward(const nis_rambo_desc* desc_host, const double* r, int64_t r_ld, int32_t r_col_major,
                      const int32_t* r_cols_host, int64_t N, double* momenta, double* weight,
                      int32_t* status, void* stream);

/* Parton densities inside the map (flat_phase_space_generator.py:120-137, :183-187: `pdf.xfxQ2(pdg, x, 91.188^2) / x` per
 * beam, an lhapdf.PDF call per sample in the reference).  One table per beam: x f(x, Q^2) of that beam's parton at the
 * FIXED scale on the x knots of an lhagrid1 grid (device arrays: logx[nx] ascending, xf[nx]; nx == 0: density 1, which is
 * also what the reference uses for particle ids other than the gluon and the quarks).  The kernel evaluates LHAPDF 6's
 * log-cubic rule in x (cubic Hermite in log x, knot derivatives = mean of the adjacent differences, one-sided at the
 * edges; the log Q^2 direction of the log-bicubic rule is folded into the table by the caller, nf_amd/PhaseSpace/
 * pdf_grid.py), x frozen at the grid boundaries.  nis_pdf_xfx evaluates x f(x) alone (PdfGrid.xfxQ2). */
typedef struct nis_pdf1d {
  const double* logx;
  const double* xf;
  int32_t nx;
  int32_t reserved;
} nis_pdf1d;
int nis_rambo_forward_pdf(const nis_rambo_desc* desc_host, const nis_pdf1d* pdf_beams_host /* [2] */,
                          const double* r, int64_t r_ld, int32_t r_col_major, const int32_t* r_cols_host, int64_t N,
                          double* momenta, double* weight, int32_t* status, void* stream);
/* Flow -> RAMBO weight -> moments as ONE launch per launch set: the integrand side of integrate()'s loop body (manager.py:397-400)
 * with f = the phase-space weight of generateKinematics_batch (flat_phase_space_generator.py:139-308, unit matrix element).
 * planes: n_batches plane sets, plane_gs doubles apart, each (d+1, ld) as nis_integrate_steps leaves them with keep_planes;
 * phase-space column c is plane cols_host[c], the flow's Jacobian is plane jplane.  No momenta and (unless weight_out
 * (n_batches, N) is given) no weights are written: weight * J is reduced to out4 (n_batches, 4) = {count, mean, M2, max}.
 * status (device int32) ACCUMULATES the bits of nis_rambo_forward over the calls; the caller reads it once per integrate().
 * workspace >= nis_rambo_moments_workspace_bytes(n_batches). */
size_t nis_rambo_moments_workspace_bytes(int32_t n_batches);
int nis_rambo_moments(const nis_rambo_desc* desc_host, const nis_pdf1d* pdf_beams_host /* [2] or NULL */, const double* planes,
                      int64_t ld, int64_t plane_gs, const int32_t* cols_host, int32_t jplane, int64_t N, int32_t n_batches,
                      double* weight_out, double* out4, int32_t* status, void* workspace, size_t workspace_bytes, void* stream);
int nis_pdf_xfx(const nis_pdf1d* table_host, const double* x, int64_t N, double* out, void* stream);

/* Live timing of ONE kernel class with HIP events recorded on the launch stream around every launch of
 * that class (bench.py's roofline leg).  kernel_class: 0 cell-apply, 1 BN statistics, 2 ingest,
 * 3 integrand+moments, 4 rambo, 5 emit, 6 pack, 7 moments.  nis_prof_read synchronises the recorded
 * events and returns the summed duration and the launch count since nis_prof_enable. */
int nis_prof_enable(int32_t kernel_class, int32_t max_launches);
int nis_prof_read(double* total_ms, int32_t* launches);
int nis_prof_disable(void);

/* Diagnostics. */
const char* nis_last_error(void);      /* host string describing the last NIS_E_HIP failure */
int nis_version(void);
int nis_device_count(void);

#ifdef __cplusplus
}
#endif
#endif /* NIS_HIP_H */
