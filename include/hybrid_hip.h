/*
 * hybrid_hip.h -- C ABI of libhybrid_hip.so, the MI355X (gfx950) implementation of
 * the CNN + temporal-transformer hot path (forward and backward).
 *
 * The reference (spygaurad/Transformer-CNN-Hybrid-Network-for-Video-Processing) has NO
 * native / FFI interface: every op on its hot path is a stock torch.nn call made from
 * nn.Module.forward.  Each entry point below therefore replaces
 * the torch dispatch the cited reference line makes; INTEGRATION.md shows the ctypes
 * binding a maintainer adds on the Python side.
 *
 * Contract:
 *   - extern "C", plain pointers and sizes only; no torch types.
 *   - the CALLER owns every buffer (inputs, outputs, saved activations, workspace);
 *     the library never allocates or frees device memory.  Process-global state is limited to
 *     (i) the optional measurement hooks of hyb_profile_set (off unless set; measurement runs only),
 *     (ii) per-kernel "dynamic-LDS attribute already set on device d" bitmasks (lock-free, idempotent) and
 *     (iii) the HYB_* debug switches, each read from the environment once.
 *     None of it depends on the data or changes results.
 *   - every function enqueues work on `stream` (a hipStream_t passed as void*) and
 *     returns immediately; no internal synchronisation; re-entrant (concurrent calls from several host threads
 *     and on several devices are allowed).
 *   - return value: 0 = success; negative = argument check failed (HYB_E_*);
 *     positive = hipError_t from a launch.  Nothing throws across the ABI.
 *   - `dtype` selects the storage/MFMA-operand type T of activations:
 *       HYB_F32  : fp32 storage, v_mfma_f32_16x16x4_f32 (exact fp32; the parity gate)
 *       HYB_BF16 : bf16 storage, v_mfma_f32_16x16x32_bf16, fp32 accumulate/statistics
 *     A second build of the SAME sources and the SAME ABI, libhybrid_hip_x3.so (-DHYB_F32_X3), gives HYB_F32 a third meaning: fp32
 *     storage, every contraction product from three bf16 MFMAs on two-term splits (x = hi + lo; csrc/hyb_common.h) -- within 1e-5 of the
 *     exact mode at 1.8 x its speed.  The Python host side selects it with compute_dtype="bf16x3" (_lib.py routes by library).
 *     Parameters (weights, biases, BN/LN affine, running stats) and parameter
 *     gradients are always fp32.
 *   - internal activation layout is NHWC with the channel count padded to a multiple
 *     of 32 ("Cp"); padded channels hold zeros.  The user-facing clip tensor stays
 *     NCHW fp32 ([B*T, C, H, W]) and is read directly by the first conv stage.
 */
#ifndef HYBRID_HIP_H_
#define HYBRID_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HYB_F32 0
#define HYB_BF16 1
/* May be OR-ed into the `dtype` of hyb_temporal_{fwd,bwd} / hyb_temporal_ce_{fwd,bwd} when dtype is HYB_F32: the last pooled map `h` (and its
 * gradient `dh`) is bf16 although the temporal part stores and computes in fp32 -- the "mixed" mode of the host side (bf16 conv stages,
 * fp32 / split-bf16 temporal part): the global-average-pool kernels read / write the other type, no cast launches.  Workspace / saved-size
 * queries take the plain dtype. */
#define HYB_H_BF16 0x100

#define HYB_E_ARG (-1)      /* bad argument (null pointer, unsupported size) */
#define HYB_E_WORKSPACE (-2) /* workspace too small */

/* ---- misc ------------------------------------------------------------------------- */
int hyb_abi_version(void);
/* bytes per element of dtype (4 or 2); HYB_E_ARG otherwise */
int hyb_dtype_size(int dtype);
/* round a channel count up to the internal padded count */
int hyb_pad_channels(int c);

/* ---- measurement hook (bench.py): time ONE kernel launch inside a real step with HIP events -----------
 * hyb_profile_set(slot, kernel_id, a, b, ev_start, ev_stop): the next launches of kernel `kernel_id` whose shape key is
 * (a, b) record hipEvent_t ev_start / ev_stop on the launch stream immediately before / after that kernel only.
 *   kernel_id 1: the conv contraction kernel of hyb_conv3x3_fwd -- conv3x3_v2_kernel (bf16, shapes with an asynchronous
 *                variant) or the first-generation conv3x3_nhwc_kernel -- (a = Cip, b = Cop as passed; forward and dgrad launches)
 *   kernel_id 2: the weight-gradient contraction kernel -- wgrad_v3_kernel / wgrad_v2_kernel or the first-generation conv3x3_wgrad_kernel --
 *                (a = Cip, b = Cop)
 *   kernel_id 4: gemm_nt_lds_kernel / gemm_nt_tall_kernel (fewer than 64 columns), the fp32 pixel-side GEMM behind hyb_conv2d_* / hyb_fct_conv_* (a = output columns, b = K as launched)
 *   kernel_id 5: flash_bwd4_dkv_kernel, the dK / dV kernel of FCT's attention backward over narrow heads (a = tokens L, b = heads)
 * slot in [0, 16).  hyb_profile_clear() removes all hooks.  The hook table is never touched unless a hook is set, and it is
 * not thread-safe (measurement runs only). */
int hyb_profile_set(int slot, int kernel_id, int a, int b, void* ev_start, void* ev_stop);
int hyb_profile_clear(void);

/* layout conversion helpers (used by tests and by in_channels > 3):
 * NCHW fp32 [N,C,H,W] <-> NHWC T [N,H,W,Cp] (padded channels written as zero). */
int hyb_nchw_to_nhwc(int dtype, const float* src, void* dst, int N, int C, int H, int W, int Cp, void* stream);
int hyb_nhwc_to_nchw(int dtype, const void* src, float* dst, int N, int C, int H, int W, int Cp, void* stream);
/* T <-> fp32 elementwise cast of n elements */
int hyb_cast_to_f32(int dtype, const void* src, float* dst, long long n, void* stream);
int hyb_cast_from_f32(int dtype, const float* src, void* dst, long long n, void* stream);

/* ---- conv stage pieces: replace nn.Conv2d / BatchNorm2d / ReLU / MaxPool2d ----------
 * reference: UNet.py:58 (Conv2d 3x3 pad 1 bias=False), UNet.py:59 (BatchNorm2d),
 * UNet.py:60 (ReLU), UNet.py:13 (MaxPool2d(2,2)).                                      */

/* number of T elements of a packed weight: first ? Cop*32 : Cop*9*Cip */
long long hyb_conv_packed_elems(int first, int Cip, int Cop);
/* w fp32 [Co,Ci,3,3] -> packed T.  mode 0: forward   wp[co][tap][ci]   (Cop x 9 x Cip)
 *                                  mode 1: dgrad     wp[ci][8-tap][co] (Cip x 9 x Cop)
 *                                  mode 2: first layer wp[co][k=tap*Ci+ci] (Cop x 32), Ci<=3 */
int hyb_conv_pack_weight(int dtype, int mode, const float* w, void* wp, int Co, int Ci, int Cop, int Cip, void* stream);

/* y[N,H,W,Cop] = conv3x3(x, wp), stride 1, zero pad 1.
 * first=1: x is NCHW fp32 [N,Ci,H,W] with Ci<=3 (Cip ignored), wp packed with mode 2.
 * first=0: x is NHWC T [N,H,W,Cip], wp packed with mode 0 (or mode 1 for dgrad, with
 *          Cip/Cop exchanged by the caller).
 * stats: NULL, or fp32 [2][Cop] (sum, sum of squares per output channel over N*H*W), overwritten.
 *        Computed from the fp32 accumulators as per-workgroup partial rows in `stats_partials`
 *        (caller workspace of hyb_conv_stats_workspace(Cop) bytes) summed in a fixed order:
 *        bitwise reproducible, no float atomics. */
size_t hyb_conv_stats_workspace(int Cop);
/* number of partial rows hyb_conv3x3_fwd writes for this problem (stats == NULL with stats_partials != NULL leaves only
 * the partial rows: feed them to hyb_bn_stats_finalize) */
int hyb_conv_stats_rows(int first, int N, int H, int W, int Cop);
int hyb_conv3x3_fwd(int dtype, int first, const void* x, const void* wp, void* y, float* stats, float* stats_partials,
                    int N, int H, int W, int Ci, int Cip, int Cop, void* stream);

/* BatchNorm2d statistics -> per-channel scale/shift (UNet.py:59; torch semantics:
 * training: batch mean, biased var for normalisation, running stats updated with
 * momentum and UNBIASED var, num_batches_tracked += 1; eval: running stats).
 * scale_shift fp32 [2][Cop]; mean_invstd fp32 [2][Cop] (saved for backward). */
int hyb_bn_finalize(const float* stats, const float* gamma, const float* beta, float* running_mean,
                    float* running_var, long long* num_batches_tracked, int training, float momentum,
                    float eps, long long count, int Co, int Cop, float* scale_shift, float* mean_invstd,
                    float* running_out /* NULL: update running_mean/var/num_batches_tracked in place (nn.BatchNorm2d semantics);
                                          else fp32 [2][Co] receiving the UPDATED (mean, var) while the inputs stay untouched and
                                          num_batches_tracked is left to the caller (functional form, for torch custom ops) */,
                    void* stream);

/* training-mode shortcut: the same, computed straight from the G per-workgroup partial rows hyb_conv3x3_fwd left in
 * `stats_partials` (fixed-order sum + finalize in one launch).  G = hyb_conv_stats_rows(...). */
int hyb_bn_stats_finalize(const float* stats_partials, int G, const float* gamma, const float* beta, float* running_mean,
                          float* running_var, long long* num_batches_tracked, float momentum, float eps, long long count,
                          int Co, int Cop, float* scale_shift, float* mean_invstd, float* running_out /* as above */, void* stream);

/* pooled[N,H/2,W/2,Cop] = maxpool2x2(relu(y*scale+shift)) (floor; H,W >= 2) */
int hyb_bn_relu_pool_fwd(int dtype, const void* y, const float* scale_shift, void* pooled,
                         int N, int H, int W, int Cop, void* stream);

/* backward of pool+relu+BN, two passes:
 *  reduce: sums[2][Cop] = (sum dy, sum dy*xhat) with dy routed through pool argmax and relu (overwritten; per-block
 *          partial rows summed in a fixed order: reproducible, no float atomics)
 *  dx    : dyraw[N,H,W,Cop] = gamma*invstd*(dy - sum_dy/count - xhat*sum_dyxhat/count) (training)
 *                             gamma*invstd*dy                                        (eval)
 *          and writes dgamma[Co] = sum dy*xhat, dbeta[Co] = sum dy. */
size_t hyb_bn_bwd_reduce_workspace(int Cop);
int hyb_bn_relu_pool_bwd_reduce(int dtype, const void* dpooled, const void* y,
                                const void* pooled /* NULL, or the forward's pooled output [N,H/2,W/2,Cop]: the sums are then formed from it
                                                      (xhat at the arg-max = (pooled - beta)/gamma where pooled > 0) without reading y */,
                                const float* scale_shift, const float* mean_invstd, float* sums, float* partials /* hyb_bn_bwd_reduce_workspace bytes */,
                                float* dgamma /* [Co] or NULL */, float* dbeta /* [Co] or NULL */,
                                int N, int H, int W, int Co, int Cop, void* stream);
int hyb_bn_relu_pool_bwd_dx(int dtype, const void* dpooled, const void* y, const float* scale_shift,
                            const float* mean_invstd, const float* gamma, const float* sums, int training,
                            long long count, void* dyraw, float* dgamma, float* dbeta,
                            int N, int H, int W, int Co, int Cop, void* stream);

/* weight gradient of the conv: dw fp32 [Co,Ci,3,3] = sum_{n,h,w} dy (x) patch(x).
 * x as in hyb_conv3x3_fwd (first selects NCHW fp32 input).  workspace: fp32 partial slabs, summed in a fixed order.
 * dw == NULL leaves only the partial slabs in the workspace (skips the final reduce). */
size_t hyb_conv3x3_wgrad_workspace(int first, int N, int H, int W, int Cip, int Cop);
int hyb_conv3x3_wgrad(int dtype, int first, const void* x, const void* dy, float* dw,
                      int N, int H, int W, int Ci, int Cip, int Co, int Cop,
                      void* workspace, size_t workspace_bytes, void* stream);

/* ---- whole conv stage (Conv3x3 -> BN -> ReLU -> MaxPool), UNet.py:58-60 + :13 -------- */
size_t hyb_convstage_fwd_workspace(int dtype, int first, int Cip, int Cop);
int hyb_convstage_fwd(int dtype, int first, const void* x, const float* weight, const float* gamma,
                      const float* beta, float* running_mean, float* running_var,
                      long long* num_batches_tracked, int training, float momentum, float eps,
                      int N, int H, int W, int Ci, int Cip, int Co, int Cop,
                      void* y_raw /* [N,H,W,Cop] T, saved */, void* pooled /* [N,H/2,W/2,Cop] T */,
                      float* scale_shift /* [2][Cop] saved */, float* mean_invstd /* [2][Cop] saved */,
                      void* packed_bwd /* NULL, or hyb_convstage_packed_bwd_elems() T elements: weights packed for backward, saved */,
                      float* running_out /* NULL = in-place running statistics; else [2][Co], see hyb_bn_finalize (training only) */,
                      void* workspace, size_t workspace_bytes, void* stream);
long long hyb_convstage_packed_bwd_elems(int first, int Cip, int Cop);
/* The FIRST stage (first = 1) never materialises its raw conv output; its y_raw argument is instead an optional buffer of
 * hyb_convstage_route_elems() T elements (0 = this dtype / shape keeps no codes: pass NULL) that the forward fills with the stage's
 * pooling / ReLU routing decisions (4 bits per pooled element: which pixel of the 2x2 window is the first maximum in torch's scan order,
 * and whether the ReLU passed) and the backward, given the same buffer as y_raw, reads instead of recomputing the convolution to re-derive
 * them.  NULL on either side: the backward recomputes (same results, bit for bit). */
long long hyb_convstage_route_elems(int dtype, int N, int H, int W, int Cop);
size_t hyb_convstage_bwd_workspace(int dtype, int first, int N, int H, int W, int Cip, int Cop);
int hyb_convstage_bwd(int dtype, int first, const void* dpooled, const void* x, const void* y_raw,
                      const void* pooled /* NULL, or the stage's forward output (see hyb_bn_relu_pool_bwd_reduce) */,
                      const float* weight, const float* gamma, const float* scale_shift,
                      const float* mean_invstd, int training,
                      int N, int H, int W, int Ci, int Cip, int Co, int Cop,
                      void* dx /* [N,H,W,Cip] T, NULL when first */, float* dweight, float* dgamma, float* dbeta,
                      const void* packed_bwd /* from hyb_convstage_fwd, or NULL to repack */,
                      void* workspace, size_t workspace_bytes, void* stream);

/* ---- frame token: global average pool over H*W (the composite's own glue) ----------- */
int hyb_gap_fwd(int dtype, const void* x /* [N,HW,Cp] */, void* feat /* [N,Cp] */, int N, int HW, int Cp, void* stream);
int hyb_gap_bwd(int dtype, const void* dfeat /* [N,Cp] */, void* dx /* [N,HW,Cp] */, int N, int HW, int Cp, void* stream);

/* ---- nn.Linear: y = x W^T + b (TransformerEncoder.pyc src L12-15, L69, L87, L107) ----
 * x [M,K] T with row stride ldx (elements), W fp32 [N,K], b fp32 [N] or NULL, y [M,N] T.
 * relu=1 applies ReLU in the epilogue (src L70 / the FFN's nn.ReLU). */
int hyb_linear_fwd(int dtype, const void* x, int ldx, const float* W, const float* b, void* y,
                   int M, int N, int K, int relu, void* stream);
/* backward: if relu, dy is first masked by (y > 0) (y = saved forward output).
 * dx [M,K] T (ldx stride; accumulate_dx=1 adds into it), dW fp32 [N,K], db fp32 [N] (both overwritten).
 * dx, dW or db may be NULL to skip.  workspace: M*N T elements (masked dy) when relu. */
int hyb_linear_bwd(int dtype, const void* x, int ldx, const float* W, const void* y, const void* dy,
                   void* dx, int accumulate_dx, float* dW, float* db, int M, int N, int K, int relu,
                   void* workspace, size_t workspace_bytes, void* stream);

/* ---- attention core: TransformerEncoder.pyc src L49-62 with the head split of L22-45 --
 * q,k,v,out: [B,S,D] T (token-major; head h = feature slice [h*D/H,(h+1)*D/H), batch index
 * b*H+h).  scores = q k^T / sqrt(D) (D = d_model, quirk Q1); mask: NULL or fp32 [B,S,S],
 * row b*H+h uses mask[(b*H+h) % B] (quirk Q4: mask.repeat(H,1,1)); masked_fill(mask==0,-1e9);
 * softmax; dropout(p_drop) on the weights (src L58, counter-based RNG keyed by seed);
 * out = weights v.  stats: fp32 [B*H,S,2] = (row max, row sum of exp) of the scaled, masked scores -- all the backward needs to
 * recompute the probabilities (the fp32 probability matrix of the first generation is gone); pass the SAME mask, p_drop and seed
 * to the backward call.
 * Limits: S <= 64 (longer sequences: hyb_attention_long_* below; hyb_encoder_* switch by themselves), D/H a multiple of 8, <= 128. */
int hyb_attention_fwd(int dtype, const void* q, const void* k, const void* v, const float* mask,
                      void* out, float* stats, int B, int S, int D, int H, float p_drop,
                      unsigned long long seed, void* stream);
int hyb_attention_bwd(int dtype, const void* q, const void* k, const void* v, const float* mask, const float* stats,
                      const void* dout, void* dq, void* dk, void* dv, int B, int S, int D, int H,
                      float p_drop, unsigned long long seed, void* stream);
/* The same attention() for sequences of ANY length (the reference has no limit on S; TransformerEncoder.pyc src L49-62): online-softmax
 * kernels over 64-key blocks, same scale / mask / dropout semantics as above.  q, k, v: rows at stride ld_qkv elements (D for separate
 * tensors, 3 D for a packed q|k|v row), dq, dk, dv at stride ld_d; out and dout are dense [B*S, D].  lse: fp32 [B*H, S], the
 * log-sum-exp of each query's scaled, masked scores (forward output, backward input, with the forward's `out`).  seed_inc: NULL, or a
 * device counter added to the seed (graph replay).  workspace: hyb_attention_long_workspace bytes.  The core computes on fp32 operands;
 * a bf16 caller's tensors are widened into the workspace and results rounded once.  Limits: D/H a multiple of 8, <= 128; B*H <= 65535. */
size_t hyb_attention_long_workspace(int dtype, int B, int S, int D, int H);
int hyb_attention_long_fwd(int dtype, const void* q, const void* k, const void* v, int ld_qkv, const float* mask, void* out, float* lse,
                           int B, int S, int D, int H, float p_drop, unsigned long long seed, const unsigned long long* seed_inc,
                           void* workspace, size_t workspace_bytes, void* stream);
int hyb_attention_long_bwd(int dtype, const void* q, const void* k, const void* v, int ld_qkv, const float* mask, const void* out,
                           const float* lse, const void* dout, void* dq, void* dk, void* dv, int ld_d, int B, int S, int D, int H,
                           float p_drop, unsigned long long seed, const unsigned long long* seed_inc, void* workspace,
                           size_t workspace_bytes, void* stream);

/* ---- LayerNorm + residual (+ scale, + dropout): src L116-117 / L120-123 ---------------
 * y = dropout_p( (LayerNorm(x)*gamma + beta + skip) * out_scale ).  stats fp32 [2][M]. */
int hyb_ln_residual_fwd(int dtype, const void* x, const void* skip, const float* gamma, const float* beta,
                        void* y, float* stats, int M, int D, float eps, float out_scale, float p_drop,
                        unsigned long long seed, void* stream);
/* dx, dskip [M,D] T; dgamma/dbeta fp32 [D] are ACCUMULATED into (one LayerNorm instance is
 * applied twice per layer, quirk Q3).  accumulate_dskip=1 adds into dskip. */
int hyb_ln_residual_bwd(int dtype, const void* dy, const void* x, const float* gamma, const float* stats,
                        void* dx, void* dskip, int accumulate_dskip, float* dgamma, float* dbeta,
                        int M, int D, float out_scale, float p_drop, unsigned long long seed, void* stream);

/* ---- whole TransformerEncoder.forward(input, mask): src L110-126 ----------------------
 * params: host array of L*14 device pointers (fp32), per layer in this order:
 *   Wq,bq,Wk,bk,Wv,bv,Wo,bo (attention_layers.i.{query,key,value,output}_layer.{weight,bias}),
 *   W1,b1,W2,b2 (feedforward_layers.i.{0,2}.{weight,bias}), ln_w, ln_b (layer_norm.i).
 * grads: same order, fp32, overwritten.  saved: caller buffer of hyb_encoder_saved_bytes.
 * attn_p = attention-weight dropout (0.1 in train mode, 0 in eval: quirk Q5);
 * layer_p = per-layer dropout (always active: quirk Q6).
 * seed_inc: NULL, or a DEVICE pointer to one 64-bit counter that the kernels add to `seed` when they run.  A launch captured
 * in a hipGraph replays with the same by-value `seed`; advancing the counter between replays (any kernel on the stream) gives
 * every replay its own dropout masks.  Pass the same pointer (and value) to the backward call of the same step. */
size_t hyb_encoder_saved_bytes(int dtype, int B, int S, int D, int Hid, int L, int H);
size_t hyb_encoder_workspace_bytes(int dtype, int B, int S, int D, int Hid, int L, int H);
int hyb_encoder_fwd(int dtype, const void* x, const float* mask, const float* const* params, void* out,
                    void* saved, int B, int S, int D, int Hid, int L, int H, float attn_p, float layer_p,
                    unsigned long long seed, const unsigned long long* seed_inc, void* stream);
int hyb_encoder_bwd(int dtype, const void* dout, const float* mask, const float* const* params,
                    float* const* grads, const void* saved, void* dx, int B, int S, int D, int Hid, int L,
                    int H, float attn_p, float layer_p, unsigned long long seed, const unsigned long long* seed_inc,
                    void* workspace, size_t workspace_bytes, void* stream);

/* ---- head: mean over T then Linear(d, classes) (composite's own) ---------------------- */
int hyb_head_fwd(int dtype, const void* x /* [B,S,D] T */, const float* W /* [C,D] */, const float* b,
                 float* logits /* [B,C] fp32 */, int B, int S, int D, int C, void* stream);
int hyb_head_bwd(int dtype, const void* x, const float* W, const float* dlogits /* [B,C] */,
                 void* dx /* [B,S,D] T */, float* dW, float* db, int B, int S, int D, int C, void* stream);

/* ---- loss: mean cross-entropy over the batch (composite's own) ------------------------
 * A target outside [0, C) (torch raises a device-side assert there) makes the loss and that row of dlogits NaN; nothing is
 * read out of bounds. */
int hyb_cross_entropy_fwd(const float* logits, const long long* target, float* loss /* [1] */,
                          int B, int C, void* stream);
int hyb_cross_entropy_bwd(const float* logits, const long long* target, const float* dloss /* [1] */,
                          float* dlogits, int B, int C, void* stream);

/* ---- model-level entry points: whole CNN backbone / whole temporal part in ONE call each way ---------------------------
 * They chain the stage-level entry points above on the caller's stream; their purpose is host time (a training step is three
 * operator calls each way), not different arithmetic: results are bit-identical to calling the stages one by one.
 *
 * hyb_backbone_*: `stages` conv stages, UNet.py:58-60 + UNet.py:13 each, in the order UNet.forward chains them (UNet.py:32-37).
 *   channels [stages+1] = {C_in (<= 4: the clip frames are read as NCHW fp32), C_1, ..., C_stages}.
 *   fwd params: HOST array of stages*5 device pointers {weight, gamma, beta, running_mean, running_var};
 *   fwd outs:   HOST array of stages*6 device pointers {y_raw (unused for stage 0), pooled, scale_shift, mean_invstd, packed_bwd,
 *               running_out}; pooled of stage s is the input of stage s+1, sizes as in hyb_convstage_fwd.
 *   BatchNorm running statistics in training mode, per stage: running_out [2][C_s] != NULL -> functional (the updated statistics are
 *               written there, running_mean / running_var are only read; see hyb_bn_finalize); running_out == NULL -> nn.BatchNorm2d's
 *               own behaviour: running_mean / running_var are updated in place and *num_batches_tracked[s] += 1 (HOST array of
 *               `stages` device pointers to 64-bit counters; the array or single entries may be NULL).
 *   bwd params: stages*2 {weight, gamma}; saved: stages*5 {y_raw, stage input (ignored for stage 0: x is passed), scale_shift,
 *               mean_invstd, packed_bwd}; grads: stages*3 {dweight, dgamma, dbeta}.  The clip tensor gets no gradient. */
size_t hyb_backbone_fwd_workspace(int dtype, int stages, const int* channels);
int hyb_backbone_fwd(int dtype, int stages, const int* channels, const float* x, const float* const* params,
                     long long* const* num_batches_tracked, int training, float momentum, float eps, int N, int H, int W,
                     void* const* outs, void* workspace, size_t workspace_bytes, void* stream);
size_t hyb_backbone_bwd_workspace(int dtype, int stages, const int* channels, int N, int H, int W);
int hyb_backbone_bwd(int dtype, int stages, const int* channels, const void* dpooled_last,
                     const void* pooled_last /* NULL, or the last stage's forward output */, const float* x, const float* const* params,
                     const void* const* saved, int training, int N, int H, int W, float* const* grads, void* workspace,
                     size_t workspace_bytes, void* stream);

/* hyb_temporal_*: frame tokens (global average pool over HW + Linear(C, D); composite's own) -> hyb_encoder_* (TransformerEncoder.pyc
 *   src L110-126) -> head (mean over S + Linear(D, classes); composite's own).  h [B*S, HW, Cp] T is the last pooled map.
 *   Saved by the caller for backward: feat [B*S, Cp] T, enc_saved (hyb_encoder_saved_bytes), enc_out [B,S,D] T; tok [B,S,D] T is scratch.
 *   bwd writes every parameter gradient and dh [B*S, HW, Cp] T. */
int hyb_temporal_fwd(int dtype, const void* h, const float* token_w, const float* token_b, const float* const* enc_params,
                     const float* head_w, const float* head_b, const float* mask, void* feat, void* tok, void* enc_saved,
                     void* enc_out, float* logits, int B, int S, int HW, int C, int Cp, int D, int Hid, int L, int H, int classes,
                     float attn_p, float layer_p, unsigned long long seed, const unsigned long long* seed_inc, void* stream);
size_t hyb_temporal_bwd_workspace(int dtype, int B, int S, int HW, int Cp, int D, int Hid, int L, int H);
int hyb_temporal_bwd(int dtype, const float* dlogits, const float* token_w, const float* const* enc_params, const float* head_w,
                     const float* mask, const void* feat, const void* enc_saved, const void* enc_out, float* dtoken_w,
                     float* dtoken_b, float* const* enc_grads, float* dhead_w, float* dhead_b, void* dh, int B, int S, int HW, int C,
                     int Cp, int D, int Hid, int L, int H, int classes, float attn_p, float layer_p, unsigned long long seed,
                     const unsigned long long* seed_inc, void* workspace, size_t workspace_bytes, void* stream);

/* hyb_temporal_ce_*: hyb_temporal_* with the mean cross-entropy loss (the composite's own loss, the harness line `loss = criterion(model(x), y)`,
 *   Model.py:56-57 / FCT.py:330-331) inside the same launches: the forward's last launch is LayerNorm + head + loss, the backward's first is
 *   loss backward + head backward + LayerNorm backward -- every dependent launch costs >= 4.6 us on this chip, and these were 8 x 512 x 8
 *   numbers in four launches each way.  target [B] class indices; loss [1]; dloss [1] = d(objective)/d(loss).
 *   ce_scratch: B + 1 floats owned by the caller, ZERO before the first call and left zero-ticketed by every call (per-clip loss terms and
 *   the ticket word the last workgroup resets); calls that share it must be stream-ordered.  Results equal hyb_temporal_* followed by
 *   hyb_cross_entropy_* bit for bit. */
int hyb_temporal_ce_fwd(int dtype, const void* h, const float* token_w, const float* token_b, const float* const* enc_params,
                        const float* head_w, const float* head_b, const float* mask, const long long* target, void* feat, void* tok,
                        void* enc_saved, void* enc_out, float* logits, float* loss, float* ce_scratch, int B, int S, int HW, int C, int Cp,
                        int D, int Hid, int L, int H, int classes, float attn_p, float layer_p, unsigned long long seed,
                        const unsigned long long* seed_inc, void* stream);
int hyb_temporal_ce_bwd(int dtype, const float* dloss, const float* logits, const long long* target, const float* token_w,
                        const float* const* enc_params, const float* head_w, const float* mask, const void* feat, const void* enc_saved,
                        const void* enc_out, float* dtoken_w, float* dtoken_b, float* const* enc_grads, float* dhead_w, float* dhead_b,
                        void* dh, int B, int S, int HW, int C, int Cp, int D, int Hid, int L, int H, int classes, float attn_p,
                        float layer_p, unsigned long long seed, const unsigned long long* seed_inc, void* workspace, size_t workspace_bytes,
                        void* stream);

/* ---- FCT, the reference's "Fully Convolutional Transformer" (FCT.py:24-254; SURVEY.md section 8f-1, first "next" row) -----------
 * FORWARD entry points (the backward is the next step of this row).  Arrays are NHWC fp32 with the TRUE channel count
 * ([N,H,W,C]; a pixel's channels are contiguous: the token view of the spatial attention, FCT.py:69-74, is free).  Arithmetic is
 * exact fp32 (fp32-input MFMA for the contractions). */
#define HYB_ACT_NONE 0
#define HYB_ACT_RELU 1
#define HYB_ACT_GELU 2    /* nn.GELU(): erf form, FCT.py:114 */
#define HYB_ACT_SIGMOID 3 /* FCT.py:205 */
/* y = act(conv3x3(x, w, stride 1, "same" zero padding, dilation) + b): nn.Conv2d(.., 3, 1, padding="same"[, dilation=d]) of
 * FCT.py:140-143, 110-113, 172-174, 194-196 followed by the ReLU / GELU / Sigmoid the reference applies next.  w [Co,Ci,3,3], b [Co] or NULL. */
size_t hyb_fct_conv_workspace(int N, int H, int W, int Ci, int Co);
int hyb_fct_conv_fwd(const float* x, const float* w, const float* b, float* y, float* z_out /* NULL, or the pre-activation [N,H,W,Co]
                     (kept for the GELU backward) */, int N, int H, int W, int Ci, int Co, int dilation,
                     int act, void* workspace, size_t workspace_bytes, void* stream);
/* Attention._build_projection for q, k and v in one pass (FCT.py:41-57): depthwise Conv2d(C, C, 3, padding 1, groups=C) + bias
 * -> ReLU -> LayerNorm over C.  HOST arrays of three device pointers each (q, k, v order): w [C,1,3,3], b [C] (or NULL), LayerNorm
 * weight / bias [C].  C a power of two. */
int hyb_fct_qkv_proj_fwd(const float* x, const float* const* w3, const float* const* b3, const float* const* g3, const float* const* beta3,
                         float* q, float* k, float* v, int N, int H, int W, int C, float eps, void* stream);
/* LayerNorm over C of P = N*H*W pixel rows (Transformer.layernorm, FCT.py:97-99) */
int hyb_fct_ln_fwd(const float* x, const float* g, const float* b, float* y, long long P, int C, float eps, void* stream);
/* nn.MultiheadAttention(embed_dim=C, num_heads, batch_first=True)(query=q, key=k, value=v, need_weights=False), FCT.py:37,75:
 * q, k, v, out [N, L, C]; in_w [3C, C], in_b [3C] (or NULL), out_w [C, C], out_b [C] (or NULL); softmax scale 1/sqrt(C/heads);
 * L = H*W tokens per image are streamed through an online softmax (nothing of size L x L is stored). */
size_t hyb_fct_mha_workspace(int N, int L, int C, int heads);
size_t hyb_fct_mha_saved_bytes(int N, int L, int C, int heads);
int hyb_fct_mha_fwd(const float* q, const float* k, const float* v, const float* in_w, const float* in_b, const float* out_w,
                    const float* out_b, float* out, void* saved /* NULL, or hyb_fct_mha_saved_bytes: kept by the caller for the backward */,
                    int N, int L, int C, int heads, void* workspace, size_t workspace_bytes, void* stream);
size_t hyb_fct_mha_bwd_workspace(int N, int L, int C, int heads);
int hyb_fct_mha_bwd(const float* dout, const float* q, const float* k, const float* v, const float* in_w, const float* out_w,
                    const void* saved, float* dq, float* dk, float* dv, float* din_w, float* din_b /* or NULL */, float* dout_w,
                    float* dout_b /* or NULL */, int N, int L, int C, int heads, void* workspace, size_t workspace_bytes, void* stream);
/* y = a + b (torch.add, FCT.py:96,101,127-128) */
int hyb_fct_add(const float* a, const float* b, float* y, long long n, void* stream);
/* mode 0: nn.MaxPool2d(2) (FCT.py:147), 1: nn.AvgPool2d(2,2) (FCT.py:222), 2: nn.Upsample(scale_factor=2) nearest (FCT.py:170).
 * H, W are the input sizes; pooling floors odd sizes like torch. */
int hyb_fct_resample(int mode, const float* x, float* y, int N, int H, int W, int C, void* stream);
/* torch.cat((a, b), dim=channels) (FCT.py:157,180) */
int hyb_fct_concat(const float* a, int Ca, const float* b, int Cb, float* y, long long P, void* stream);
/* DiceLoss.forward (Metrics.py:14-22) on channel 0 of NCHW pred / true [N,C,H,W]: 1 - (2 sum(p t) + smooth) / (sum p + sum t + smooth) */
size_t hyb_dice_workspace(void);
int hyb_dice_fwd(const float* pred, const float* tru, float* loss /* [1] */, int N, int C, long long HW, float smooth, void* workspace,
                 size_t workspace_bytes, void* stream);

/* ---- FCT backward (autograd of the forward entry points above; all gradients overwritten, reductions in a fixed order) -----------
 * conv: saved = y for ReLU / sigmoid, the pre-activation z for GELU, ignored for NONE; dx may be NULL (first layer). */
size_t hyb_fct_conv_bwd_workspace(int N, int H, int W, int Ci, int Co);
int hyb_fct_conv_bwd(const float* dy, const float* x, const float* w, const float* saved, float* dx, float* dw, float* db /* or NULL */, int N, int H,
                     int W, int Ci, int Co, int dilation, int act, void* workspace, size_t workspace_bytes, void* stream);
size_t hyb_fct_qkv_proj_bwd_workspace(int N, int H, int W, int C);
int hyb_fct_qkv_proj_bwd(const float* x, const float* const* w3, const float* const* b3, const float* const* g3, const float* const* dq3,
                         float* dx, float* const* dw3, float* const* db3, float* const* dg3, float* const* dbeta3, int N, int H, int W,
                         int C, float eps, void* workspace, size_t workspace_bytes, void* stream);
size_t hyb_fct_ln_bwd_workspace(long long P, int C);
int hyb_fct_ln_bwd(const float* dy, const float* x, const float* g, float* dx, float* dg, float* db, long long P, int C, float eps,
                   void* workspace, size_t workspace_bytes, void* stream);
/* mode 0: MaxPool2d(2) backward (x = the pool's input [N,H,W,C]; the gradient goes to the first maximum in scan order, like torch);
 * mode 2: Upsample x2 backward (dy [N,2H,2W,C] -> dx [N,H,W,C]) */
int hyb_fct_resample_bwd(int mode, const float* dy, const float* x, float* dx, int N, int H, int W, int C, void* stream);
int hyb_fct_concat_bwd(const float* dy, float* da /* or NULL */, int Ca, float* db /* or NULL */, int Cb, long long P, void* stream);
int hyb_dice_bwd(const float* pred, const float* tru, const float* dloss, float* dpred /* NCHW like pred */, int N, int C, long long HW, float smooth,
                 void* workspace /* >= 4096 bytes */, size_t workspace_bytes, void* stream);
/* nn.Dropout in train mode (FCT.py:115,146,175): y = x * keep / (1 - p), mask from the counter-based RNG keyed by (seed + *seed_inc);
 * the backward is the same call on the gradient. */
int hyb_fct_dropout(const float* x, float* y, long long n, float p, unsigned long long seed, const unsigned long long* seed_inc, void* stream);

/* ---- clip input pipeline: torchvision's to-tensor transform on the device -----------------------------
 * dst[f][c][h][w] = src[f][h][w][c] / 255 for `frames` uint8 HWC frames (what PIL / cv2 decode to; Dataloader.py:19-23,
 * dataset.pyc src L106-113): frames cross PCIe as bytes and become the fp32 NCHW clip tensor the first conv stage reads. */
int hyb_frames_u8hwc_to_f32chw(const unsigned char* src, float* dst, long long frames, int H, int W, int C, void* stream);

/* ---- ResNet-bottleneck backbone `Encoder_32K` (only bytecode of it ships with the reference:
 * __pycache__/AE_256_32K.cpython-38.pyc, read as data -- `Bottleneck` src L21-53, `Encoder_32K` src L58-137).  NHWC fp32 with the
 * true channel counts, exact-fp32 arithmetic like the FCT entry points above, which these generalise.
 *
 * nn.Conv2d(Ci, Co, k, stride, padding, dilation, bias) forward / backward: the stem Conv2d(3, 64, 7, 2, 3, bias=False) (src L62),
 * the bottleneck's 1x1 / 3x3-stride-s / 1x1 (src L25-30), the down-sampling Conv2d(.., 1, stride, bias=False) (src L99-102) and the
 * tail's Conv2d(.., 3, 1, 1) with bias (src L70-88).  x [N,H,W,Ci], w [Co,Ci,k,k], y [N,Ho,Wo,Co] with torch's output size
 * floor((H + 2 pad - dilation (k-1) - 1) / stride) + 1; act / z_out / saved as for hyb_fct_conv_*, which are these at k=3, stride 1. */
size_t hyb_conv2d_workspace(int N, int H, int W, int Ci, int Co, int k, int stride, int pad, int dilation);
int hyb_conv2d_fwd(const float* x, const float* w, const float* b /* or NULL */, float* y, float* z_out /* or NULL */, int N, int H, int W, int Ci,
                   int Co, int k, int stride, int pad, int dilation, int act, void* workspace, size_t workspace_bytes, void* stream);
size_t hyb_conv2d_bwd_workspace(int N, int H, int W, int Ci, int Co, int k, int stride, int pad, int dilation);
int hyb_conv2d_bwd(const float* dy, const float* x, const float* w, const float* saved, float* dx /* or NULL */, float* dw, float* db /* or NULL */,
                   int N, int H, int W, int Ci, int Co, int k, int stride, int pad, int dilation, int act, void* workspace,
                   size_t workspace_bytes, void* stream);
/* nn.BatchNorm2d over P = N*H*W pixel rows of C channels (C % 4 == 0, C/4 a divisor of 256), optionally fused with the bottleneck's
 * `out += residual` and nn.ReLU (src L44-52): y = relu?( (x - mean) / sqrt(var + eps) * gamma + beta (+ residual) ).
 * training != 0: batch statistics (biased variance), running_mean / running_var (or NULL) updated in place with `momentum` and the
 * unbiased variance, like torch; training == 0: the running statistics normalise.  coef [4][C] (a, b, mean, invstd) is written by
 * the forward and read by the backward; y in the backward is the forward's output (ReLU mask; ignored when relu == 0) -- NULL when the
 * forward had no residual: the mask is then recomputed as fmaf(x, a, b) > 0, the forward's own expression (one array less to read). */
size_t hyb_bn2d_workspace(long long P, int C);
int hyb_bn2d_fwd(const float* x, const float* gamma, const float* beta, const float* residual /* or NULL */, float* y, float* coef,
                 float* running_mean, float* running_var, long long P, int C, float eps, float momentum, int training, int relu,
                 void* workspace, size_t workspace_bytes, void* stream);
int hyb_bn2d_bwd(const float* dy, const float* x, const float* y, const float* gamma, const float* coef, float* dx,
                 float* dresidual /* or NULL */, float* dgamma, float* dbeta, long long P, int C, int training, int relu, void* workspace,
                 size_t workspace_bytes, void* stream);
/* nn.Dropout2d(p) in train mode (src L92, L113, L136) on [N,H,W,C]: one decision per (image, channel) plane from the counter-based RNG
 * keyed by (seed + *seed_inc); the backward is the same call on the gradient. */
int hyb_dropout2d(const float* x, float* y, int N, long long HW, int C, float p, unsigned long long seed, const unsigned long long* seed_inc,
                  void* stream);

/* ---- optimizer step: torch.optim.AdamW of Model.py:153 / FCT.py:305, all tensors in one launch ---------
 * Same update as torch.optim.AdamW(betas=(beta1,beta2), eps, weight_decay, amsgrad=False, maximize=False) at step number
 * `step` (1-based).  params/grads/exp_avg/exp_avg_sq: HOST arrays of `count` device pointers (fp32 tensors of numel[i]
 * elements); state tensors are caller-owned and must be zero before step 1.  Hyper-parameters are doubles (Python floats):
 * 1 - beta etc. are formed in double and rounded to fp32 once, like torch does.
 * step_inc: NULL, or a DEVICE pointer to one 64-bit counter: the step number used is step + *step_inc, read when the kernel runs
 * (the bias corrections are then formed on the device, in double) -- lets one captured launch serve every replay of a hipGraph.
 * advance_ticket != NULL (needs step_inc): the call also adds 1 to *step_inc once every workgroup has read it (the last workgroup to
 * finish does it), so the replayed step needs no separate "counter += 1" launch.  advance_ticket is a DEVICE pointer to one 32-bit word
 * owned by the caller, zero at rest, used by this counter's advancing launches only (the launch counts its finished workgroups there and
 * leaves it zero): advancing calls on different counters -- two optimizers on two streams -- each bring their own word and never
 * interfere; two advancing calls on the SAME counter must be stream-ordered, as any two steps of one optimizer are. */
int hyb_adamw_step(int count, float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                   const long long* numel, double lr, double beta1, double beta2, double eps, double weight_decay, long long step,
                   long long* step_inc, unsigned int* advance_ticket, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* HYBRID_HIP_H_ */
