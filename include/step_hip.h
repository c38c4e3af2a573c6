/*
 * step_hip.h - C ABI of libstep_hip.so: the MI355X (gfx950) kernels of STEP's ST-GCN
 * block stack.
 *
 * The reference (UttaranB127/STEP) is pure Python/PyTorch and has no FFI layer of its
 * own: every entry point below replaces the PyTorch library ops that one reference
 * statement lowers to; the statement is cited as file:line under /root/reference.
 * The Python host (step_amd/net/...) binds these with ctypes; INTEGRATION.md shows the stub.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller; the library keeps nothing
 *     after return except work in flight on `stream` (a hipStream_t passed as void*);
 *   - activations are fp32, contiguous (N, C, T, V) with V == 16 joints (graph.py:41);
 *   - return value: 0 on success, a negative step_status otherwise; the message is in
 *     step_last_error() (thread-local).  Shapes are checked on the host BEFORE a launch;
 *   - re-entrant; no global mutable state; one process per GPU under data parallelism.
 */
#ifndef STEP_HIP_H
#define STEP_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define STEP_ABI_VERSION 1
#define STEP_V 16                 /* joints of the gait skeleton, graph.py:41 */

enum step_status {
    STEP_OK = 0,
    STEP_ERR_ARG = -1,            /* bad shape / null pointer / unsupported size     */
    STEP_ERR_LAUNCH = -2,         /* hipGetLastError() after the launch              */
    STEP_ERR_WORKSPACE = -3       /* workspace too small                             */
};

int step_abi_version(void);
const char* step_last_error(void);

/* ---------------------------------------------------------------- BatchNorm --------
 * group layout `bn_mode`:
 *   0: group = c        nn.BatchNorm2d over (N,T,V)            classifier.py:154,163,180
 *   1: group = v*C + c  `data_bn`, BatchNorm1d on (N, V*C, T)  classifier.py:68-74
 * G = C (mode 0) or V*C (mode 1).  `inner` = T*V.
 */

/* eval mode: scale = gamma*rsqrt(var+eps), shift = beta - mean*scale (F.batch_norm, training=False) */
int step_bn_eval_prepare(const float* gamma, const float* beta, const float* running_mean,
                         const float* running_var, float eps, float* scale, float* shift,
                         int G, void* stream);

/* train mode: batch statistics of x (biased var for normalisation), running-stat update
 * (momentum, unbiased var) exactly as nn.BatchNorm; emits mean, invstd, scale, shift.
 * ws: step_bn_ws_bytes(G) bytes of scratch.  running_* may be NULL (no update).          */
size_t step_bn_ws_bytes(int G);
int step_bn_train_prepare(const float* x, int N, int C, int T, int V, int bn_mode,
                          const float* gamma, const float* beta,
                          float* running_mean, float* running_var, float momentum, float eps,
                          float* mean, float* invstd, float* scale, float* shift,
                          void* ws, size_t ws_bytes, void* stream);

/* The two halves of step_bn_train_prepare, for batch statistics taken over SEVERAL GPUs (synchronised BatchNorm:
 * with per-rank statistics an 8 x 512 data-parallel step is not the reference's single-device batch-4096
 * BatchNorm, nn.BatchNorm2d in classifier.py:154-164).  _stats leaves SUM x, SUM x^2 per group in `sums`
 * (2*G doubles, caller-owned: the host all-reduces them and the element counts across ranks); _finalize takes
 * the combined sums and the combined count and does what step_bn_train_prepare does after its own reduction;
 * count == 0: the count is read from sums[2*G] on the device (a host that all-reduced [sums, count] in one
 * buffer needs no device-to-host copy).                                                                    */
int step_bn_train_stats(const float* x, int N, int C, int T, int V, int bn_mode, double* sums,
                        void* ws, size_t ws_bytes, void* stream);
int step_bn_train_finalize(const double* sums, double count, const float* gamma, const float* beta,
                           float* running_mean, float* running_var, float momentum, float eps,
                           float* mean, float* invstd, float* scale, float* shift, int G, void* stream);

/* step_bn_train_prepare for callers that cannot allocate (code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc).  It has NO counterpart in the reference.  Same arguments, results and
 * launches as step_bn_train_prepare (bit-identical outputs) minus the workspace: the chunk partials, totals
 * and coefficients live in a file-scope device array of the library (1.06 MB per GPU, held until the process
 * ends), whose address is looked up once per device on the first call there (the current device must be the
 * stream's).  Limit: G = C (mode 0) or 16*C (mode 1) <= STEP_BN_STATIC_MAX_GROUPS, larger shapes are
 * rejected with STEP_ERR_ARG.  Calls no hipMalloc / hipFree.  Every call on a GPU shares that GPU's array:
 * calls that overlap on one GPU (two streams, two host threads) race - serialise them or use the
 * workspace form.  Nothing in the array has to start at zero.  Under HIP graph capture: make ONE ordinary
 * call on the device before capturing (the first call's hipGetSymbolAddress is not a capturable call); every
 * later call, captured or not, only launches kernels.                                                  */
#define STEP_BN_STATIC_MAX_GROUPS 256
int step_bn_train_prepare_static(const float* x, int N, int C, int T, int V, int bn_mode,
                                 const float* gamma, const float* beta,
                                 float* running_mean, float* running_var, float momentum, float eps,
                                 float* mean, float* invstd, float* scale, float* shift, void* stream);

/* out = act( x*scale[g] + shift[g] + (res ? res*rscale[g] + rshift[g] : 0) ); rscale/rshift
 * NULL means identity.  The tail of st_gcn.forward: `self.relu(self.tcn(x) + res)`
 * classifier.py:189-191, and the Decoder's final data_bn CVAE_stgcn.py:224-229.            */
int step_affine_act(const float* x, const float* scale, const float* shift,
                    const float* res, const float* rscale, const float* rshift,
                    int relu, float* out, int N, int C, int T, int V, int bn_mode, void* stream);

/* BatchNorm (+ReLU) backward.  g = dy * mask, mask_mode:
 *   0 none; 1 (mask_src > 0) [mask_src = the block output]; 2 (x*scale+shift > 0) [BN+ReLU].
 * reduce: sums[2g] = SUM g, sums[2g+1] = SUM g*x   (fp64, 2*G doubles, caller-owned)
 * apply : training: dx = gamma*invstd*(g - S1/M - xhat*S(g*xhat)/M); eval: dx = g*scale;
 *         accumulate != 0 adds into dx.  dgamma/dbeta (nullable) are written by apply;
 *         dx may be NULL (only dgamma/dbeta wanted).  ws: step_bn_ws_bytes(G) bytes.          */
int step_bn_bwd_reduce(const float* dy, const float* x, const float* mask_src,
                       const float* scale, const float* shift, int mask_mode,
                       int N, int C, int T, int V, int bn_mode,
                       double* sums, void* ws, size_t ws_bytes, void* stream);
int step_bn_bwd_apply(const float* dy, const float* x, const float* mask_src,
                      const float* scale, const float* shift, int mask_mode,
                      const float* mean, const float* invstd, const float* gamma,
                      const double* sums, int training, int accumulate,
                      float* dx, float* dgamma, float* dbeta,
                      int N, int C, int T, int V, int bn_mode,
                      void* ws, size_t ws_bytes, void* stream);
/* Tail of a block with a residual conv, out = relu(bn2(u) + bn_r(r)) classifier.py:163,175-181,189-191:
 * both BatchNorm backward passes (g = dout * (out > 0) for either) as one reduce and one apply, i.e.
 * step_bn_bwd_reduce + step_bn_bwd_apply with mask_mode 1 for (u, BN2) and for (r, residual BN), sharing the
 * reads of dout and out; results are bit-identical to the two separate calls.  bn_mode 0 only.
 * ws: 2 * step_bn_ws_bytes(C) bytes.                                                              */
int step_bn_bwd_tail_pair(const float* dout, const float* out, const float* u, const float* r,
                          const float* mean_u, const float* invstd_u, const float* gamma_u,
                          const float* mean_r, const float* invstd_r, const float* gamma_r,
                          int training, float* du, float* dr, float* dgamma_u, float* dbeta_u,
                          float* dgamma_r, float* dbeta_r, int N, int C, int T, int V,
                          void* ws, size_t ws_bytes, void* stream);
/* dx (+)= dy * (mask_src > 0): identity residual branch, classifier.py:170-171 */
int step_masked_add(const float* dy, const float* mask_src, float* dx, int accumulate,
                    size_t n, void* stream);

/* ------------------------------------------------------------- graph convolution ----
 * ConvTemporalGraphical.forward tgcn.py:58-67 (1x1 conv to K*Co channels + einsum
 * 'nkctv,kvw->nctw') and, with w_transposed, ConvTransposeTemporalGraphical tgcn.py:121-130.
 * Fused: optional data_bn affine on load (in_scale/in_shift indexed v*Ci+ci), optional
 * residual branch `Conv2d(Ci,Co,1,stride=(s,1))` classifier.py:175-179 sharing the x tile.
 *   x (N,Ci,T,16)  w (K*Co,Ci) | (Ci,K*Co) if w_transposed   b (K*Co)
 *   A_eff (K,16,16) = A * edge_importance (classifier.py:78); a_nnz_max bounds the
 *   non-zeros of every (k, column) AND every (k, row) of A_eff (16 = dense, always valid).
 *   y (N,Co,T,16) raw graph-conv output (pre tcn[0]);  res_out (N,Co,To,16) raw, To=(T-1)/s+1.
 */
int step_gcn_fwd(const float* x, const float* in_scale, const float* in_shift,
                 const float* w, const float* b, int w_transposed,
                 const float* A_eff, int a_nnz_max, float* y,
                 const float* res_w, const float* res_b, int res_transposed, float* res_out,
                 int N, int Ci, int Co, int K, int T, int stride, void* stream);
/* step_gcn_fwd with both contractions on the bf16 matrix cores, every fp32 operand (x, A_eff, w, the aggregated
 * z) as bf16 hi + lo and the lo * lo term dropped (about 2^-16 relative per product, fp32 accumulation): the
 * graph-conv half of the opt-in "f32_split" mode.  Same arguments and layouts.  Covers Ci % 16 == 0, Co % 32 == 0,
 * both <= 256, no data_bn; any other layer (the first block) runs step_gcn_fwd's fp32 kernels unchanged.       */
int step_gcn_fwd_split(const float* x, const float* in_scale, const float* in_shift,
                       const float* w, const float* b, int w_transposed,
                       const float* A_eff, int a_nnz_max, float* y,
                       const float* res_w, const float* res_b, int res_transposed, float* res_out,
                       int N, int Ci, int Co, int K, int T, int stride, void* stream);

/* step_gcn_fwd for a TRAIN-mode block: also returns the batch statistics the BatchNorms behind the two outputs need
 * (tcn[0] on y, residual[1] on res_out: nn.BatchNorm2d in classifier.py:154,180) as y_sums / r_sums = [SUM, SUM of
 * squares] per output channel over (N, T, V), 2*Co doubles each, in the layout step_bn_train_finalize takes
 * (r_sums NULL iff res_w is NULL).  For Ci, Co in {32, 64} without data_bn the sums are taken in the kernel's
 * epilogue (per-wave fp32 over at most 64 frames, then fp64 partials per block, ordered sums: bit-reproducible)
 * and y / res_out are not read again; every other shape runs step_gcn_fwd followed by step_bn_train_stats.
 * ws: step_gcn_fwd_stats_ws_bytes(Co) bytes.                                                              */
size_t step_gcn_fwd_stats_ws_bytes(int Co);
int step_gcn_fwd_stats(const float* x, const float* in_scale, const float* in_shift,
                       const float* w, const float* b, int w_transposed,
                       const float* A_eff, int a_nnz_max, float* y,
                       const float* res_w, const float* res_b, int res_transposed, float* res_out,
                       int N, int Ci, int Co, int K, int T, int stride,
                       double* y_sums, double* r_sums, void* ws, size_t ws_bytes, void* stream);

/* backward of the above (K16 of SURVEY 2.2).  Any of dx, dw/db, dA, dres_w/dres_b may be
 * NULL.  dA is returned on the support of A_eff only (d importance = A * dA needs no more).
 * ws: step_gcn_bwd_ws_bytes(...) bytes.                                                    */
size_t step_gcn_bwd_ws_bytes(int N, int Ci, int Co, int K, int T);
int step_gcn_bwd(const float* dy, const float* dres, const float* x,
                 const float* in_scale, const float* in_shift,
                 const float* w, const float* b, int w_transposed,
                 const float* A_eff, int a_nnz_max,
                 const float* res_w, int res_transposed,
                 float* dx, float* dw, float* db, float* dA, float* dres_w, float* dres_b,
                 int N, int Ci, int Co, int K, int T, int stride,
                 void* ws, size_t ws_bytes, void* stream);

/* step_gcn_bwd for a block with an IDENTITY residual (in == out channels, stride 1: `self.residual = lambda x: x`,
 * classifier.py:170-171): the block input also receives g = id_dout * [id_out > 0] (id_out = the block's output,
 * id_dout its gradient; relu(tcn(x) + x) of classifier.py:189-191), added to dx in the data kernel's store
 * where that kernel form exists, by step_masked_add after it otherwise.  id_dout == NULL: step_gcn_bwd.   */
int step_gcn_bwd_idres(const float* dy, const float* dres, const float* x,
                       const float* in_scale, const float* in_shift,
                       const float* w, const float* b, int w_transposed,
                       const float* A_eff, int a_nnz_max,
                       const float* res_w, int res_transposed,
                       const float* id_dout, const float* id_out,
                       float* dx, float* dw, float* db, float* dA, float* dres_w, float* dres_b,
                       int N, int Ci, int Co, int K, int T, int stride,
                       void* ws, size_t ws_bytes, void* stream);
/* step_gcn_bwd_idres with the MFMA products of the 32 / 64-channel kernels (dz = W^T dy, dA += x^T dz, z = x A,
 * dW += dy z^T, dWr += dres x^T) on the bf16 matrix cores, every fp32 operand as bf16 hi + lo and the lo * lo term
 * dropped (the backward half of the opt-in "f32_split_gcn" mode; dx's sum over joints and every reduction stay fp32).
 * Same arguments, layouts and workspace.  Layers outside Ci, Co in {32, 64} (first block, wide layers) run
 * step_gcn_bwd_idres's fp32 kernels unchanged.                                                              */
int step_gcn_bwd_split(const float* dy, const float* dres, const float* x,
                       const float* in_scale, const float* in_shift,
                       const float* w, const float* b, int w_transposed,
                       const float* A_eff, int a_nnz_max,
                       const float* res_w, int res_transposed,
                       const float* id_dout, const float* id_out,
                       float* dx, float* dw, float* db, float* dA, float* dres_w, float* dres_b,
                       int N, int Ci, int Co, int K, int T, int stride,
                       void* ws, size_t ws_bytes, void* stream);

/* ---------------------------------------------------------- temporal convolution ----
 * tcn[2] = Conv2d(C,C,(Kt,1),(s,1),((Kt-1)//2,0)) classifier.py:156-162 and its
 * ConvTranspose2d twin CVAE_stgcn.py:359-365, as an fp32 MFMA implicit GEMM.
 *
 * Weights are first re-laid out for the matrix cores (once per weight update):
 *   pack_mode 0: forward of Conv2d          w (Co,Ci,Kt)
 *             1: forward of ConvTranspose2d w (Ci,Co,Kt), stride 1
 *             2: input-gradient of Conv2d   w (Co,Ci,Kt), stride 1 or 2
 *             3: input-gradient of ConvTranspose2d w (Ci,Co,Kt), stride 1
 * Kt must be 75 or 9 (the reference builds only Kt=75; 9 is the original ST-GCN value).
 */
size_t step_tcn_packed_elems(int Co, int Ci, int Kt, int stride, int pack_mode);
int step_tcn_pack_weight(const float* w, float* wp, int Co, int Ci, int Kt, int stride,
                         int pack_mode, void* stream);

/* out = epi( conv( pre(in) ) ):
 *   pre : in*in_scale[c]+in_shift[c] then ReLU if in_relu (tcn[0], tcn[1] applied on load;
 *         zero padding is applied AFTER it, as in the reference); NULL scale = identity.
 *   epi : + bias; then if out_scale: *out_scale[c]+out_shift[c] (tcn[3], eval);
 *         then if res: + res*res_scale[c]+res_shift[c] (NULL res_scale = identity);
 *         then ReLU if out_relu (classifier.py:189-191, eval-mode fusion).
 * For pack_mode 0/1: in (N,Ci,Tin,16) -> out (N,Co,Tout,16), Tout=(Tin-1)/s+1.
 * For pack_mode 2/3 (dgrad): in = dout (N,Co,Tout,16) -> out = din (N,Ci,Tin,16).        */
int step_tcn_conv(const float* in, const float* in_scale, const float* in_shift, int in_relu,
                  const float* wp, const float* bias,
                  const float* out_scale, const float* out_shift,
                  const float* res, const float* res_scale, const float* res_shift,
                  int out_relu, float* out,
                  int N, int Ci, int Co, int Tin, int Tout, int Kt, int stride, int pack_mode,
                  void* stream);

/* step_tcn_conv for a TRAIN-mode block (pack_mode 0 / 1, no fused epilogue): the raw conv output u (+ bias) and, from
 * the kernel's epilogue, sums = [SUM u, SUM u^2] per output channel over (N, Tout, V) (2*Co doubles) for the batch
 * statistics of tcn[3] (nn.BatchNorm2d, classifier.py:163), in the layout step_bn_train_finalize takes: u is not read
 * again for statistics.  Per-block fp32 partials (one sequence, at most 40 frames), then fp64, fixed summation
 * order: bit-reproducible.  ws: step_tcn_conv_stats_ws_bytes(N, Co, Tout) bytes.                             */
size_t step_tcn_conv_stats_ws_bytes(int N, int Co, int Tout);
int step_tcn_conv_stats(const float* in, const float* in_scale, const float* in_shift, int in_relu,
                        const float* wp, const float* bias, float* out,
                        int N, int Ci, int Co, int Tin, int Tout, int Kt, int stride, int pack_mode,
                        double* sums, void* ws, size_t ws_bytes, void* stream);

/* Temporal conv of a frame-independent input: the decoder repeats its latent vector over T and V before the
 * first st_gctn block (generator_cvae/net/CVAE_stgcn.py Decoder.forward, `z.repeat([1, 1, T, V])`), so that
 * block's conv is one Co x Ci matrix per output frame (the taps inside the sequence there), stride 1:
 *   wcum (T,Co,Ci) from w (Conv2d (Co,Ci,Kt) / ConvTranspose2d (Ci,Co,Kt) when transposed);
 *   out[n,co,t,w] = epi( SUM_ci wcum[t][co][ci] * a0[n,ci,w] ), a0 (N,Ci,16) the activated graph-conv output of
 *   ONE frame, epi as step_tcn_conv with res0 (N,Co,16) the residual of one frame.  Eval mode only.          */
int step_tcn_const_weights(const float* w, float* wcum, int Co, int Ci, int Kt, int T, int transposed,
                           void* stream);
int step_tcn_const_fwd(const float* a0, const float* wcum, const float* bias,
                       const float* out_scale, const float* out_shift,
                       const float* res0, const float* res_scale, const float* res_shift, int out_relu,
                       float* out, int N, int Ci, int Co, int T, void* stream);

/* bf16-operand twins of the three entry points above (BASELINE config "classifier_stgcn_real_and_synth
 * training, bf16"; same reference statements: classifier.py:156-162 and its input gradient).
 * Tensors stay fp32 in HBM; activations (after `pre`) and weights are rounded to bf16 on their
 * way into v_mfma_f32_32x32x16_bf16, accumulation and the epilogue are fp32.  Built for
 * pack_mode 0 and 2, Kt = 75, channels 32 / 64 at stride 1 and 64 at stride 2 (every temporal
 * conv of the three classifiers); step_tcn_packed_bytes_bf16 returns 0 for anything else and the
 * caller keeps the fp32 entry points.                                                        */
size_t step_tcn_packed_bytes_bf16(int Co, int Ci, int Kt, int stride, int pack_mode);
int step_tcn_pack_weight_bf16(const float* w, void* wp, int Co, int Ci, int Kt, int stride,
                              int pack_mode, void* stream);
int step_tcn_conv_bf16(const float* in, const float* in_scale, const float* in_shift, int in_relu,
                       const void* wp, const float* bias,
                       const float* out_scale, const float* out_shift,
                       const float* res, const float* res_scale, const float* res_shift,
                       int out_relu, float* out,
                       int N, int Ci, int Co, int Tin, int Tout, int Kt, int stride, int pack_mode,
                       void* stream);

/* weight gradient: dw[co][ci][j] = SUM_{n,t,v} dout[n,co,t,v] * a[n,ci,s*t+j-pad,v], with
 * a = relu?(in*in_scale+in_shift) recomputed on load; db[co] = SUM dout.  transposed != 0
 * writes the ConvTranspose2d layout (Ci,Co,Kt) instead.                                   */
size_t step_tcn_wgrad_ws_bytes(int N, int Ci, int Co, int Tin, int Tout, int Kt);
int step_tcn_wgrad(const float* dout, const float* in, const float* in_scale,
                   const float* in_shift, int in_relu, float* dw, float* db,
                   int N, int Ci, int Co, int Tin, int Tout, int Kt, int stride, int transposed,
                   void* ws, size_t ws_bytes, void* stream);

/* bf16-operand twin of step_tcn_wgrad (dout and the recomputed activation rounded to bf16 on their
 * way into the MFMA, fp32 accumulation; db stays an exact fp32 sum).  Built for Kt = 75, square
 * 32- or 64-channel convs, stride 1 or 2; the ws_bytes helper returns 0 for anything else and the
 * caller keeps step_tcn_wgrad.  Writes the Conv2d layout (Co,Ci,Kt).                          */
size_t step_tcn_wgrad_bf16_ws_bytes(int N, int Ci, int Co, int Tin, int Tout, int Kt, int stride);
int step_tcn_wgrad_bf16(const float* dout, const float* in, const float* in_scale,
                        const float* in_shift, int in_relu, float* dw, float* db,
                        int N, int Ci, int Co, int Tin, int Tout, int Kt, int stride,
                        void* ws, size_t ws_bytes, void* stream);
/* the same with `transposed` as in step_tcn_wgrad: the ConvTranspose2d layout (Ci,Co,Kt) of the st_gctn blocks'
 * tcn[2] (generator_cvae/net/CVAE_stgcn.py:359-365; stride 1 only), so that the bf16 mode covers the CVAE     */
int step_tcn_wgrad_bf16_t(const float* dout, const float* in, const float* in_scale,
                          const float* in_shift, int in_relu, float* dw, float* db,
                          int N, int Ci, int Co, int Tin, int Tout, int Kt, int stride, int transposed,
                          void* ws, size_t ws_bytes, void* stream);

/* The split-operand twins of step_tcn_packed_bytes_bf16 / step_tcn_pack_weight_bf16 / step_tcn_conv_bf16: fp32
 * operands enter the bf16 matrix cores as hi = bf16(v) and lo = bf16(v - hi) (16 mantissa bits); a group is 8 input
 * channels x {hi, lo} and a tap takes two MFMAs, w_hi * (v_hi + v_lo) and w_lo * v_hi, fp32 accumulate; the dropped
 * w_lo * v_lo is 2^-16 relative.  Same statements of the reference as step_tcn_conv (classifier.py:156-162 and its
 * input gradient), same shapes as the bf16 twins, held to the fp32 tolerance of the parity tests.  Opt-in
 * (step_amd.set_mfma_dtype(model, "f32_split")).                                                            */
size_t step_tcn_packed_bytes_split(int Co, int Ci, int Kt, int stride, int pack_mode);
int step_tcn_pack_weight_split(const float* w, void* wp, int Co, int Ci, int Kt, int stride,
                               int pack_mode, void* stream);
int step_tcn_conv_split(const float* in, const float* in_scale, const float* in_shift, int in_relu,
                        const void* wp, const float* bias,
                        const float* out_scale, const float* out_shift,
                        const float* res, const float* res_scale, const float* res_shift,
                        int out_relu, float* out,
                        int N, int Ci, int Co, int Tin, int Tout, int Kt, int stride, int pack_mode,
                        void* stream);

/* step_tcn_conv_stats (train-mode forward conv that also returns SUM out, SUM out^2 per channel for the BatchNorm
 * behind it, classifier.py:156-164) on the bf16 matrix cores: split == 0 as step_tcn_conv_bf16, else as
 * step_tcn_conv_split; wp packed by the matching pack entry point; forward pack modes (0, 1) only.  The sums are fp32
 * per tile and lane, fp64 from there on, in a fixed order (bit-reproducible).                                   */
size_t step_tcn_conv_stats_bf16_ws_bytes(int Co);
int step_tcn_conv_stats_bf16(const float* in, const float* in_scale, const float* in_shift, int in_relu,
                             const void* wp, const float* bias, float* out,
                             int N, int Ci, int Co, int Tin, int Tout, int Kt, int stride, int pack_mode, int split,
                             double* sums, void* ws, size_t ws_bytes, void* stream);

/* step_tcn_wgrad with every product taken as THREE bf16 MFMAs on split fp32 operands (v = hi + lo, hi = bf16(v),
 * lo = bf16(v - hi): 16 mantissa bits; a*b ~ a_lo*b_hi + a_hi*b_lo + a_hi*b_hi, fp32 accumulate; the dropped
 * lo*lo term is 2^-16 relative).  Same statement of the reference as step_tcn_wgrad (the weight part of
 * convolution_backward for tcn[2], classifier.py:156-162; `transposed` as in step_tcn_wgrad), same shapes and workspace as step_tcn_wgrad_bf16; held
 * to the fp32 tolerance of the parity tests, at about three times the speed of the fp32-MFMA kernel.  Opt-in
 * (step_amd.set_mfma_dtype(model, "f32_split_wgrad" | "f32_split")): whether split-operand products count as fp32 arithmetic is
 * the caller's decision.                                                                                       */
int step_tcn_wgrad_split(const float* dout, const float* in,
                         const float* in_scale, const float* in_shift, int in_relu,
                         float* dw, float* db,
                         int N, int Ci, int Co, int Tin, int Tout, int Kt, int stride, int transposed,
                         void* ws, size_t ws_bytes, void* stream);

/* ---------------------------------------------------- temporal conv in the frequency domain ----------
 * The same three operators as step_tcn_conv / its input gradient / step_tcn_wgrad (tcn[2], classifier.py:156-162;
 * ConvTranspose2d twin CVAE_stgcn.py:359-365) computed as  transform -> per-bin channel mixing -> inverse transform,
 * every stage an fp32 GEMM on the matrix cores (csrc/tcn_dft.hip has the algebra).  Spectrum layout:
 * S[n][f2][c][v] fp32, f2 in [0, rows), rows = 2 NF = step_dft_rows(T, Kt), f2 < NF real parts, else imaginary;
 * L = rows - 1.  Both transforms are HALF the dense DFT's size: the frames are paired about their centre (hi = Kh + j,
 * lo = Kh - 1 - j, Kh = ceil(K / 2)), the cos rows act on the pair sums, the sin rows on the pair differences, and a
 * spectrum's time origin is that centre, step * (Kh - 1/2); the inverse produces E[j] and O[j] and writes E + O, E - O.
 * The origin shift between a conv's input and output (0 at stride 1, half a frame at stride 2) is a phase in the tap
 * tables tc / ts = cos / sin(2 pi f (j - P + delta) / L) that step_dft_wspec and step_dft_wgrad read.
 * The cos / sin matrices are the caller's (step_amd/dft.py builds them in float64), tiled for the kernels:
 * [group][k][32][mgp] with (mg, mgp, groups) = step_dft_table_geom(rows of the matrix, inverse), zero past the real
 * extents; the forward table is the cos half's tiling (NF rows, k < Kh) followed by the sin half's, the inverse table has
 * ceil(T_out / 2) rows and k < 2 NF (cos part against the real rows, sin part against the imaginary rows).  C % 8 == 0.
 *   step_dft_fwd   S = F . act(in)      in (N,C,K,16) [BN scale/shift + ReLU on the load]
 *   step_dft_mix   out[n][.][co][v] = W[f] . in[n][.][ci][v] per bin, complex, as THREE real products (Gauss: k1 = (a+b) Xr,
 *                  k2 = a (Xi - Xr), k3 = b (Xr + Xi); Re = k1 - k3, Im = k1 + k2); wmat = the triples (a+b, a, b, 0) per
 *                  (row, k), [NF][rows / 32][K / 2][2][32][4] floats = NF * 4 * Co * Ci, from step_dft_wspec
 *   step_dft_inv   out (N,C,T_out,16) = G . S (+ bias); optional SUM out, SUM out^2 per channel (as step_tcn_conv_stats),
 *                  or step_tcn_conv's eval-mode epilogue (BN2 affine, residual with its affine, ReLU: classifier.py:189-191)
 *   step_dft_wspec taps (Co,Ci,Kt) [or the ConvTranspose2d layout] -> forward and backward mixing matrices
 *   step_dft_wgrad dw, db from the spectra of dout and of the activated input (kept from the forward pass); the per-bin
 *                  complex products again as three real GEMMs
 *   step_dft_groups(T_out): partial-sum rows per sample that step_dft_inv's statistics use (step_dft_inv_ws_bytes)  */
int step_dft_rows(int T, int Kt);
int step_dft_table_geom(int rows, int inverse, int* mgp, int* groups);
int step_dft_groups(int T_out);
int step_dft_fwd(const float* in, const float* in_scale, const float* in_shift, int in_relu,
                 const float* matt, float* spec, int N, int C, int K, int rows, void* stream);
int step_dft_mix(const float* in, const float* wmat, float* out, int N, int Ci, int Co, int NF, void* stream);
size_t step_dft_inv_ws_bytes(int N, int C, int T_out);
int step_dft_inv(const float* spec, const float* matt, const float* bias,
                 const float* out_scale, const float* out_shift,
                 const float* res, const float* res_scale, const float* res_shift, int out_relu,
                 float* out, int N, int C, int rows_in, int T_out,
                 double* sums, void* ws, size_t ws_bytes, void* stream);
/* Long sequences at stride 1, overlap-save (no counterpart in the reference: the same tcn[2] conv, classifier.py:156-162,
 * evaluated segment by segment so that the transforms cost S * K^2 instead of T^2): the row of T frames is cut into S
 * segments of seg_step output frames; segment s transforms the K input frames starting at frame0 + s * seg_step (frame0 =
 * -(Kt-1)/2; frames outside [0, T) are the conv's zero padding, applied after the activation) and writes its T_seg output
 * frames at s * seg_step (frames past T_out are dropped).  Spectra are [N * S][rows][C][16] with rows = step_dft_rows(K -
 * (Kt-1)/2, Kt) >= K + 1: the outputs kept never wrap.  step_dft_mix takes N * S samples; the tap tables carry the
 * segment's origin shift (step_amd/dft.py).  ws for sums: S * step_dft_inv_ws_bytes(N, C, T_seg).                  */
int step_dft_fwd_seg(const float* in, const float* in_scale, const float* in_shift, int in_relu,
                     const float* matt, float* spec, int N, int C, int T, int K, int rows, int S, int seg_step,
                     int frame0, void* stream);
int step_dft_inv_seg(const float* spec, const float* matt, const float* bias,
                     const float* out_scale, const float* out_shift,
                     const float* res, const float* res_scale, const float* res_shift, int out_relu,
                     float* out, int N, int C, int rows_in, int T_seg, int T_out, int S, int seg_step,
                     double* sums, void* ws, size_t ws_bytes, void* stream);
/* A stride-2 conv as two stride-1 convs (polyphase; the same tcn[2] statement of a block with stride 2, classifier.py:156-162
 * and the blocks of classifier.py:40-51): output frame t' reads the frames 2 t' + j - P, i.e. the even frames through the
 * odd taps and the odd frames through the even taps.  The tensor (N, C, 2 Tp, 16) is taken as `phases` = 2 interleaved
 * sequences of Tp frames per channel row; step_dft_fwd_ph writes the spectrum of sequence (c, ph) as channel 2 c + ph of
 * spec [N][rows][phases * C][16] with rows = step_dft_rows(Tp, 2 ceil(P / 2) + 1) - every transform of the path works on
 * Tp frames and ~half the bins.  step_dft_mix then mixes 2 C -> C channels per bin forward (C -> 2 C backward, wmat from
 * step_dft_wspec on the taps split by phase, (Co, 2 ci + ph, Kt) with zeros where a tap belongs to the other phase, and a tap
 * table whose column j holds the tap's offset within its phase), step_dft_inv gives the Tp output frames, step_dft_wgrad
 * takes Ci = 2 Co, and step_dft_inv_ph returns the input gradient's two phases to their interleaved frames.           */
int step_dft_fwd_ph(const float* in, const float* in_scale, const float* in_shift, int in_relu,
                    const float* matt, float* spec, int N, int C, int Tp, int rows, int phases, void* stream);
int step_dft_inv_ph(const float* spec, const float* matt, float* out, int N, int C, int rows_in, int Tp,
                    int phases, void* stream);
int step_dft_wspec(const float* w, const float* tc, const float* ts, float* wf, float* wb, int Ci, int Co,
                   int NF, int Kt, int transposed, void* stream);
size_t step_dft_wgrad_ws_bytes(int N, int Ci, int Co, int NF);
int step_dft_wgrad(const float* dspec, const float* spec, const float* tc, const float* ts, float* dw,
                   float* db, int N, int Ci, int Co, int NF, int Kt, int transposed, void* ws,
                   size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------- head ----------
 * F.avg_pool2d(x, x.size()[2:]) + mean over M=1, classifier.py:81-82                      */
int step_pool_fwd(const float* x, float* feat, int N, int C, int inner, void* stream);
int step_pool_bwd(const float* dfeat, float* dx, int N, int C, int inner, void* stream);
/* y[n,o] = SUM_i x[n,i]*w[o,i] + b[o]  (w_transposed: w[i,o]) - the 1x1 `fcn`
 * classifier.py:63,86, z_mean/z_lsig CVAE_stgcn.py:105-106, Decoder.fcn :171, hybrid head.   */
int step_dense_fwd(const float* x, const float* w, const float* b, int w_transposed,
                   float* y, int N, int I, int O, void* stream);
int step_dense_bwd(const float* dy, const float* x, const float* w, int w_transposed,
                   float* dx, float* dw, float* db, int N, int I, int O, void* stream);
/* The classifier's inference head in one launch: feat = avg_pool(x) (N,C), out = feat W^T + b (N,O), softmaxed when
 * `softmax` (classifier.py:81-87: avg_pool2d, fcn, F.softmax; the real_and_synth variant returns the logits).
 * Bit-identical to step_pool_fwd + step_dense_fwd (+ step_softmax_fwd), which the autograd path keeps.     */
int step_head_fwd(const float* x, const float* w, const float* b, int w_transposed, int softmax,
                  float* feat, float* out, int N, int C, int inner, int O, void* stream);
/* nn.BatchNorm1d on a (N, G) matrix - `data_bn2` of the hybrid head,
 * classifier_hybrid/net/classifier.py:46,83.  Emits mean/invstd (batch or running).        */
int step_bn_rows_fwd(const float* x, const float* gamma, const float* beta,
                     float* running_mean, float* running_var, int training,
                     float momentum, float eps, float* y, float* mean, float* invstd,
                     int N, int G, void* stream);
int step_bn_rows_bwd(const float* dy, const float* x, const float* mean, const float* invstd,
                     const float* gamma, int training, float* dx, float* dgamma, float* dbeta,
                     int N, int G, void* stream);
/* The same BatchNorm1d in train mode with its batch taken over SEVERAL GPUs (synchronised BatchNorm, as
 * step_bn_train_stats / _finalize do for the 4-D layers; data_bn2, classifier_hybrid/net/classifier.py:46,83):
 *   _stats     sums[2g] = SUM x, sums[2g+1] = SUM x^2 over this rank's N rows (2G doubles of a 2G+1 buffer; the host
 *              writes its row count into slot 2G and all-reduces the buffer);
 *   _apply     mean / biased variance from the combined buffer (count read from slot 2G), running-stat update, y;
 *   _bwd_sums  sums[2g] = SUM dy, sums[2g+1] = SUM dy * xhat (this rank), dgamma / dbeta from them (nullable);
 *   _bwd_apply dx = gamma * invstd * (dy - S1/M - xhat * S2/M) with the combined sums and M = slot 2G.       */
int step_bn_rows_stats(const float* x, double* sums, int N, int G, void* stream);
int step_bn_rows_apply(const float* x, const double* sums, const float* gamma, const float* beta,
                       float* running_mean, float* running_var, float momentum, float eps, float* y,
                       float* mean, float* invstd, int N, int G, void* stream);
int step_bn_rows_bwd_sums(const float* dy, const float* x, const float* mean, const float* invstd,
                          double* sums, float* dgamma, float* dbeta, int N, int G, void* stream);
int step_bn_rows_bwd_apply(const float* dy, const float* x, const float* mean, const float* invstd,
                           const float* gamma, const double* sums, float* dx, int N, int G, void* stream);
/* F.softmax(x, dim=1) classifier.py:87 and its backward */
int step_softmax_fwd(const float* x, float* y, int N, int O, void* stream);
int step_softmax_bwd(const float* dy, const float* y, float* dx, int N, int O, void* stream);

/* ------------------------------------------------------------- CVAE loss ----
 * vae_loss = between_frame_loss(x_in, x_out) + beta * KLD(mean, lsig):
 * generator_cvae/utils/losses.py:7-42 (a Python loop of scalar mse_loss calls in the reference) and
 * generator_cvae/utils/processor_stgcn.py:34-40, value and gradients in one pass.
 * x_in, x_out (N,C,T,V) (M = 1); mean, lsig (N,Z) or both NULL (between_frame_loss alone).
 * loss[3] = {total, between_frame, KLD}; d_x_out (N,C,T,V), d_mean, d_lsig (N,Z) nullable
 * (d loss / d x_in = -d_x_out).  ws: step_vae_loss_ws_bytes(N).  C*T*V <= 15360.             */
size_t step_vae_loss_ws_bytes(int N);
int step_vae_loss(const float* x_in, const float* x_out, const float* mean, const float* lsig, float beta,
                  float* loss, float* d_x_out, float* d_mean, float* d_lsig,
                  int N, int C, int T, int V, int Z, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------ input pipeline ----
 * out[n, c, t, v] = SUM_d M[n][c][d] * (gaits[n, t, v, d] - gaits[n, t, 0, d]) + M[n][c][3]:
 * the root-relative similarity alignment of utils/common.py:105-112 (first 48 features of
 * get_affective_features, all utils/loader.py:37 keeps) fused with the (N,T,V*C) -> (N,C,T,V)
 * reshape + moveaxis of utils/loader.py:65-66.  gaits (N,T,V,3), M (N,3,4) = [cR | t] of
 * get_transformation (common.py:230-279, computed on the host), out (N,3,T,V).              */
int step_pose_align(const float* gaits, const float* M, float* out, int N, int T, int V, void* stream);

/* The same for a RAGGED set of gaits, fused with the padding of utils/loader.py:33-36: gait n is frames
 * [offsets[n], offsets[n+1]) of `frames` (sum_len, V, 3); every gait is tiled to `time_steps` frames (the
 * longest gait: np.tile(...)[0:time_steps]) and that block is repeated `cycles` times, so output frame t of
 * gait n is source frame (t mod time_steps) mod len_n.  offsets: N+1 ints on the device, every gait at least
 * one frame (checked by the host caller).  out (N, 3, time_steps*cycles, V).                       */
int step_pose_align_ragged(const float* frames, const int* offsets, const float* M, float* out, int N,
                           int time_steps, int cycles, int V, void* stream);

/* ------------------------------------------------ affective features ----
 * feat[n, :] = compute_features(gaits[n], time_step, add_stride_features) of
 * compute_aff_features/compute_features.py:428-468 (helpers compute_aff_features/utils.py:3-34): box
 * volume, 5 angles, 4 distances, 2 Heron areas, mean speed / acceleration / jerk magnitude of hands,
 * head and feet (27 values), + stride length and gait-cycle time (:393-425) when add_stride_features.
 * gaits (N,T,48) float32 or float64 (in_f64), feat (N, 27 | 29) float64; float64 arithmetic as in the
 * reference; means over an empty set (T < 4) are NaN as numpy's.                                    */
int step_aff_features(const void* gaits, int in_f64, double* feat, int N, int T, double time_step,
                      int add_stride_features, void* stream);
/* compute_aff_features/normalize_features.py:1-22: out[n,j] = (feat[n,j] - (max_j+min_j)/2) /
 * ((max_j-min_j)/2), 0 where max_j == min_j; min/max over the N rows.  out (N,F) float32 or float64. */
int step_aff_normalize(const double* feat, void* out, int out_f64, int N, int F, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* STEP_HIP_H */
