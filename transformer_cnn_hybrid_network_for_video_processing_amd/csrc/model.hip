// Stage-level entry points: each is a fixed sequence of the kernels in this library, enqueued back to back on
// the caller's stream (no allocation, no synchronisation -> capturable in a hipGraph).
//   hyb_convstage_{fwd,bwd} : Conv3x3 -> BatchNorm2d -> ReLU -> MaxPool2d      (UNet.py:58-60, UNet.py:13)
//   hyb_encoder_{fwd,bwd}   : TransformerEncoder.forward, all layers             (TransformerEncoder.pyc src L110-126)
#include <math.h>
#include <stdlib.h>
#include "hyb_common.h"

int hyb_gemm_nt(int dtype, int groups, const void* const* A, const void* const* B, void* const* C, const float* const* bias, int out_f32,
                int Mo, int No, int R, int lda, int ldb, int ldc, int relu, int accumulate, hipStream_t st, const void* const* Amask = nullptr,
                const void* const* Cmask = nullptr);
int hyb_convert_weights(int dtype, int count, const float* const* W, void* const* Wc, void* const* Wt, const int* N, const int* K,
                        const int* ldt, hipStream_t st);
int hyb_ln_bwd_rows(int M);
int hyb_gemm_nt_ln(int dtype, int groups, const void* x, const void* skip, const float* gamma, const float* beta, void* y, float* stats, float eps,
                   float out_scale, float p_drop, unsigned long long seed, const unsigned long long* seed_inc, const void* const* B, void* const* C,
                   const float* const* bias, int Mo, int No, int R, int ldb, int ldc, int relu, hipStream_t st);
int hyb_ln_residual_bwd_rows(int dtype, const void* dy, const void* x, const float* gamma, const float* stats, void* dx, void* dskip,
                             int accumulate_dskip, float* part, int M, int D, float out_scale, float p_drop, unsigned long long seed,
                             const unsigned long long* seed_inc, hipStream_t st);
int hyb_ln_residual_fwd_inc(int dtype, const void* x, const void* skip, const float* gamma, const float* beta, void* y, float* stats,
                            int M, int D, float eps, float out_scale, float p_drop, unsigned long long seed, const unsigned long long* seed_inc,
                            void* stream);
int hyb_ln_rows_reduce(const float* part, int rows, int D, float* dgamma, float* dbeta, hipStream_t st);
int hyb_linear_dw_multi(int dtype, int groups, const void* const* dy, const void* const* mask, const void* const* x, float* const* dW,
                        float* const* db, const int* N, const int* K, const int* lddy, const int* ldx, int M, hipStream_t st,
                        int nriders, const HybDwRider* riders);
int hyb_linear_dw_grouped(int dtype, int groups, const void* const* dy, const void* const* mask, const void* x, float* const* dW,
                          float* const* db, int M, int N, int K, int lddy, int ldx, hipStream_t st);
int hyb_attention_fwd_packed(int dtype, const void* qkv, const float* mask, void* out, float* stats, int B, int S, int D, int H, float p_drop,
                             unsigned long long seed, const unsigned long long* seed_inc, hipStream_t st);
int hyb_attention_bwd_packed(int dtype, const void* qkv, const float* mask, const float* stats, const void* dout, void* dqkv, int B, int S, int D,
                             int H, float p_drop, unsigned long long seed, const unsigned long long* seed_inc, hipStream_t st, int relu_out);
extern "C" size_t hyb_attention_long_workspace(int dtype, int B, int S, int D, int H);
int hyb_linear_bwd_wt(int dtype, const void* x, int ldx, const float* W, const void* Wt, const void* y, const void* dy, void* dx, int accumulate_dx,
                      float* dW, float* db, int M, int N, int K, int relu, void* ws, size_t ws_bytes, hipStream_t st);

size_t hyb_stage1_fwd_workspace(int dtype, int Cop);
size_t hyb_stage1_bwd_workspace(int dtype, int Cop);
int hyb_stage1_fwd(int dtype, const float* x, const float* weight, const float* gamma, const float* beta, float* running_mean,
                   float* running_var, long long* nbt, int training, float momentum, float eps, int N, int H, int W, int Ci, int Co, int Cop,
                   void* pooled, float* scale_shift, float* mean_invstd, void* packed_out, void* workspace, float* running_out, int prepacked,
                   void* route, hipStream_t st);
long long hyb_stage1_route_elems(int dtype, int N, int H, int W, int Cop);
int hyb_stage1_bwd(int dtype, const void* dpooled, const float* x, const float* weight, const float* gamma, const float* scale_shift,
                   const float* mean_invstd, int training, int N, int H, int W, int Ci, int Co, int Cop, float* dweight, float* dgamma,
                   float* dbeta, const void* packed_in, void* workspace, const void* route, hipStream_t st);
int hyb_conv3x3_wgrad_fused(int dtype, const void* x, const void* y, const void* dp, const float* ss, const float* mi, const float* gamma,
                            const float* sums, int training, long long count, void* dyraw_out, long long dyraw_blk, float* dw, int N, int H, int W,
                            int Ci, int Cip, int Co, int Cop, void* workspace, size_t workspace_bytes, hipStream_t st, HybSlabInfo* defer);
int hyb_wgrad_v2_supported(int dtype, int W, int Cip, int Cop);
int hyb_conv_dgrad_planar_ok(int dtype, int W, int Cin_p, int Cout_p);
int hyb_conv3x3_planar_in(const void* x, const void* wp, void* y, int N, int H, int W, int Cin_p, int Cout_p, hipStream_t st);
int hyb_conv_pack_weight_dual(int dtype, const float* w, void* wp0, void* wp1, int Co, int Ci, int Cop, int Cip, hipStream_t st);

namespace {

inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

#define HYB_TRY(call) do { int rc_ = (call); if (rc_ != 0) return rc_; } while (0)
#define HYB_HIP_TRY(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return (int)e_; } while (0)

struct EncLayout {       // byte offsets inside `saved` for one layer, plus per-layer stride
    size_t x_in, qkv, probs, attn, o, st1, x1, hmid, f, st2, long_ws, long_ws_bytes, layer_bytes;
    size_t wc[6], wt[6];   // T copies (plain / transposed) of Wq, Wk, Wv, Wo, W1, W2, converted once per forward;
                           // wt[0..2] are column blocks of ONE [D][3D] matrix (K-concatenated dX of the Q, K, V projections)
};
inline EncLayout enc_layout(int dtype, int B, int S, int D, int Hid, int H) {
    const size_t es = dtype == HYB_F32 ? 4 : 2;
    const size_t M = (size_t)B * S;
    EncLayout L;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += align256(bytes); return o; };
    L.x_in = take(M * D * es);
    L.qkv = take(M * 3 * D * es);                  // post-ReLU q | k | v, token-major [M][3D]
    L.probs = take((size_t)B * H * S * 2 * 4);       // softmax row statistics (max, sum) per (clip, head, query): P is recomputed in backward
    L.attn = take(M * D * es);
    L.o = take(M * D * es);
    L.st1 = take(2 * M * 4);
    L.x1 = take(M * D * es);
    L.hmid = take(M * Hid * es);
    L.f = take(M * D * es);
    L.st2 = take(2 * M * 4);
    // more than 64 tokens per clip: attention() goes through hyb_attention_long_* (attention.hip), whose scratch (dense fp32 copies of the
    // packed q|k|v, delta) lives here, per layer, in the caller's saved blob: the forward entry point has no workspace argument
    L.long_ws_bytes = S > 64 ? hyb_attention_long_workspace(dtype, B, S, D, H) : 0;
    L.long_ws = take(L.long_ws_bytes);
    const size_t wsz[6] = {(size_t)D * D, (size_t)D * D, (size_t)D * D, (size_t)D * D, (size_t)Hid * D, (size_t)D * Hid};
    // (fp32 storage: the forward reads the master weights in place, no plain copies -- only the transposed ones below)
    for (int i = 0; i < 6; ++i) L.wc[i] = take(dtype == HYB_F32 ? 0 : wsz[i] * es);
    L.wt[0] = take(3 * wsz[0] * es);
    L.wt[1] = L.wt[0] + (size_t)D * es;
    L.wt[2] = L.wt[0] + 2 * (size_t)D * es;
    for (int i = 3; i < 6; ++i) L.wt[i] = take(wsz[i] * es);
    L.layer_bytes = off;
    return L;
}

inline unsigned long long attn_seed(unsigned long long seed, int layer) { return seed + 0x9E3779B97F4A7C15ull * (unsigned long long)(2 * layer + 1); }
inline unsigned long long drop_seed(unsigned long long seed, int layer) { return seed + 0x9E3779B97F4A7C15ull * (unsigned long long)(2 * layer + 2); }

}  // namespace

// ----------------------------------------------------------------------------------------------------------
// conv stage
// ----------------------------------------------------------------------------------------------------------
// first stage: the packed weights in both K orders of conv_first.hip / conv_first_wave.hip, [2][Cop][64], followed by the Gram matrix of
// the patches in double (2306 x 8 bytes, counted here in 2-byte elements so that the buffer is large enough for either dtype)
extern "C" long long hyb_convstage_packed_bwd_elems(int first, int Cip, int Cop) { return first ? (long long)Cop * 128 + 2306 * 4 : (long long)Cip * 9 * Cop; }

extern "C" long long hyb_convstage_route_elems(int dtype, int N, int H, int W, int Cop) { return hyb_stage1_route_elems(dtype, N, H, W, Cop); }

extern "C" size_t hyb_convstage_fwd_workspace(int dtype, int first, int Cip, int Cop) {
    if (first) return hyb_stage1_fwd_workspace(dtype, Cop);
    const size_t es = dtype == HYB_F32 ? 4 : 2;
    return align256((size_t)hyb_conv_packed_elems(first, Cip, Cop) * es) + align256(2 * (size_t)Cop * 4) + align256(hyb_conv_stats_workspace(Cop));
}

// prepacked_fwd != NULL: the caller (hyb_backbone_fwd) has packed this stage's forward AND backward weights already (first stage: both
// layouts in packed_bwd; the pointer is only a flag there)
int hyb_convstage_fwd_impl(int dtype, int first, const void* x, const float* weight, const float* gamma, const float* beta,
                           float* running_mean, float* running_var, long long* nbt, int training, float momentum, float eps,
                           int N, int H, int W, int Ci, int Cip, int Co, int Cop, void* y_raw, void* pooled, float* scale_shift,
                           float* mean_invstd, void* packed_bwd, float* running_out, void* workspace, size_t workspace_bytes, void* stream,
                           const void* prepacked_fwd) {
    HYB_CHECK_ARG(x && weight && gamma && beta && running_mean && running_var && (first || y_raw) && pooled && scale_shift && mean_invstd && workspace);
    HYB_CHECK_ARG(dtype == HYB_F32 || dtype == HYB_BF16);
    HYB_CHECK_ARG(H >= 2 && W >= 2 && Cop % 32 == 0 && Cop >= Co && Co > 0 && N > 0);
    if (workspace_bytes < hyb_convstage_fwd_workspace(dtype, first, Cip, Cop)) return HYB_E_WORKSPACE;
    if (first)      // stage 1: the raw conv output is never materialised; y_raw, when given, receives the pooling / ReLU routing codes
        return hyb_stage1_fwd(dtype, (const float*)x, weight, gamma, beta, running_mean, running_var, nbt, training, momentum, eps, N, H, W, Ci,
                              Co, Cop, pooled, scale_shift, mean_invstd, packed_bwd, workspace, running_out, prepacked_fwd != nullptr, y_raw,
                              (hipStream_t)stream);
    const size_t es = dtype == HYB_F32 ? 4 : 2;
    char* ws = (char*)workspace;
    void* wp = ws;
    float* stats = (float*)(ws + align256((size_t)hyb_conv_packed_elems(first, Cip, Cop) * es));
    float* part = (float*)((char*)stats + align256(2 * (size_t)Cop * 4));
    if (prepacked_fwd) wp = const_cast<void*>(prepacked_fwd);
    else if (packed_bwd) HYB_TRY(hyb_conv_pack_weight_dual(dtype, weight, wp, packed_bwd, Co, Ci, Cop, Cip, (hipStream_t)stream));
    else HYB_TRY(hyb_conv_pack_weight(dtype, 0, weight, wp, Co, Ci, Cop, Cip, stream));
    if (training) {   // conv leaves per-workgroup partial sums; one launch sums them in a fixed order and finalises BN
        HYB_TRY(hyb_conv3x3_fwd(dtype, 0, x, wp, y_raw, nullptr, part, N, H, W, Ci, Cip, Cop, stream));
        HYB_TRY(hyb_bn_stats_finalize(part, hyb_conv_stats_rows(0, N, H, W, Cop), gamma, beta, running_mean, running_var, nbt, momentum, eps,
                                      (long long)N * H * W, Co, Cop, scale_shift, mean_invstd, running_out, stream));
    } else {
        HYB_TRY(hyb_conv3x3_fwd(dtype, 0, x, wp, y_raw, nullptr, nullptr, N, H, W, Ci, Cip, Cop, stream));
        HYB_TRY(hyb_bn_finalize(stats, gamma, beta, running_mean, running_var, nbt, 0, momentum, eps, (long long)N * H * W, Co, Cop,
                                scale_shift, mean_invstd, nullptr, stream));
    }
    HYB_TRY(hyb_bn_relu_pool_fwd(dtype, y_raw, scale_shift, pooled, N, H, W, Cop, stream));
    return 0;
}

extern "C" int hyb_convstage_fwd(int dtype, int first, const void* x, const float* weight, const float* gamma, const float* beta,
                                 float* running_mean, float* running_var, long long* nbt, int training, float momentum, float eps,
                                 int N, int H, int W, int Ci, int Cip, int Co, int Cop, void* y_raw, void* pooled, float* scale_shift,
                                 float* mean_invstd, void* packed_bwd, float* running_out, void* workspace, size_t workspace_bytes, void* stream) {
    return hyb_convstage_fwd_impl(dtype, first, x, weight, gamma, beta, running_mean, running_var, nbt, training, momentum, eps, N, H, W, Ci, Cip, Co,
                                  Cop, y_raw, pooled, scale_shift, mean_invstd, packed_bwd, running_out, workspace, workspace_bytes, stream, nullptr);
}

extern "C" size_t hyb_convstage_bwd_workspace(int dtype, int first, int N, int H, int W, int Cip, int Cop) {
    if (first) return hyb_stage1_bwd_workspace(dtype, Cop);
    const size_t es = dtype == HYB_F32 ? 4 : 2;
    size_t b = align256(2 * (size_t)Cop * 4) + align256(hyb_bn_bwd_reduce_workspace(Cop));   // sums + partial rows
    b += align256((size_t)N * H * W * Cop * es);                                  // dense grad of the raw conv output
    if (!first) b += align256((size_t)Cip * 9 * Cop * es);                        // dgrad-packed weights
    b += align256(hyb_conv3x3_wgrad_workspace(first, N, H, W, Cip, Cop));         // wgrad slabs
    return b;
}

int hyb_convstage_bwd_impl(int dtype, int first, const void* dpooled, const void* x, const void* y_raw, const void* pooled, const float* weight,
                           const float* gamma, const float* scale_shift, const float* mean_invstd, int training, int N, int H, int W,
                           int Ci, int Cip, int Co, int Cop, void* dx, float* dweight, float* dgamma, float* dbeta,
                           const void* packed_bwd, void* workspace, size_t workspace_bytes, void* stream, void* slab_ws, HybSlabInfo* defer);
extern "C" int hyb_convstage_bwd(int dtype, int first, const void* dpooled, const void* x, const void* y_raw, const void* pooled, const float* weight,
                                 const float* gamma, const float* scale_shift, const float* mean_invstd, int training, int N, int H, int W,
                                 int Ci, int Cip, int Co, int Cop, void* dx, float* dweight, float* dgamma, float* dbeta,
                                 const void* packed_bwd, void* workspace, size_t workspace_bytes, void* stream) {
    return hyb_convstage_bwd_impl(dtype, first, dpooled, x, y_raw, pooled, weight, gamma, scale_shift, mean_invstd, training, N, H, W, Ci, Cip, Co, Cop, dx,
                                  dweight, dgamma, dbeta, packed_bwd, workspace, workspace_bytes, stream, nullptr, nullptr);
}
// slab_ws + defer (hyb_backbone_bwd): the weight-gradient slabs go to the caller's buffer, which outlives this call, and their fixed-order sum
// is left to the caller (*defer describes it; S = 0 when this stage's kernel path summed them itself)
int hyb_convstage_bwd_impl(int dtype, int first, const void* dpooled, const void* x, const void* y_raw, const void* pooled, const float* weight,
                           const float* gamma, const float* scale_shift, const float* mean_invstd, int training, int N, int H, int W,
                           int Ci, int Cip, int Co, int Cop, void* dx, float* dweight, float* dgamma, float* dbeta,
                           const void* packed_bwd, void* workspace, size_t workspace_bytes, void* stream, void* slab_ws, HybSlabInfo* defer) {
    if (defer) defer->S = 0;
    HYB_CHECK_ARG(dpooled && x && (first || y_raw) && weight && gamma && scale_shift && mean_invstd && dweight && workspace);
    HYB_CHECK_ARG(dtype == HYB_F32 || dtype == HYB_BF16);
    HYB_CHECK_ARG(first || dx);
    HYB_CHECK_ARG(Cop % 32 == 0 && Cop >= Co && Co > 0 && N > 0 && H >= 2 && W >= 2);
    if (workspace_bytes < hyb_convstage_bwd_workspace(dtype, first, N, H, W, Cip, Cop)) return HYB_E_WORKSPACE;
    if (first)
        return hyb_stage1_bwd(dtype, dpooled, (const float*)x, weight, gamma, scale_shift, mean_invstd, training, N, H, W, Ci, Co, Cop, dweight,
                              dgamma, dbeta, packed_bwd, workspace, y_raw, (hipStream_t)stream);
    const size_t es = dtype == HYB_F32 ? 4 : 2;
    char* ws = (char*)workspace;
    float* sums = (float*)ws;                    ws += align256(2 * (size_t)Cop * 4);
    float* sum_part = (float*)ws;                ws += align256(hyb_bn_bwd_reduce_workspace(Cop));
    void* dyraw = ws;                            ws += align256((size_t)N * H * W * Cop * es);
    void* wpd = nullptr;
    if (!first) { wpd = ws;                      ws += align256((size_t)Cip * 9 * Cop * es); }
    void* slabs = (slab_ws && defer) ? slab_ws : (void*)ws;
    const size_t slab_bytes = hyb_conv3x3_wgrad_workspace(first, N, H, W, Cip, Cop);
    const long long count = (long long)N * H * W;
    HYB_TRY(hyb_bn_relu_pool_bwd_reduce(dtype, dpooled, y_raw, pooled, scale_shift, mean_invstd, sums, sum_part, dgamma, dbeta, N, H, W, Co, Cop, stream));
    // dense BN/ReLU/pool backward is computed inside the wgrad tile staging; the tile is also written once (dyraw) for dgrad
    // Layout of the dense gradient between the two kernels.  NHWC makes the dgrad conv read each 128-byte line of a >= 64-channel
    // gradient once per 32-channel block (stage 2: 555 MB fetched for 308 MB, by PMC); when both kernels are the second-generation
    // ones the tensor is written block-planar, [Cop/32][N][H][W][32], and a block's halo uses whole lines.
    static const int planar_env = getenv("HYB_DYRAW_PLANAR") ? atoi(getenv("HYB_DYRAW_PLANAR")) : 1;
    const bool planar = planar_env && !first && Cop >= 64 && hyb_wgrad_v2_supported(dtype, W, Cip, Cop) && hyb_conv_dgrad_planar_ok(dtype, W, Cop, Cip);
    const long long dyraw_blk = planar ? (long long)N * H * W * 32 : 0;
    HYB_TRY(hyb_conv3x3_wgrad_fused(dtype, x, y_raw, dpooled, scale_shift, mean_invstd, gamma, sums, training, count, dyraw, dyraw_blk, dweight, N, H,
                                    W, Ci, Cip, Co, Cop, slabs, slab_bytes, (hipStream_t)stream, (slab_ws && defer) ? defer : nullptr));
    if (!first) {
        // dgrad = conv3x3 of the dense output gradient with the transposed, tap-flipped weights
        const void* wd = packed_bwd;
        if (!wd) { HYB_TRY(hyb_conv_pack_weight(dtype, 1, weight, wpd, Co, Ci, Cop, Cip, stream)); wd = wpd; }
        if (planar) HYB_TRY(hyb_conv3x3_planar_in(dyraw, wd, dx, N, H, W, Cop, Cip, (hipStream_t)stream));
        else HYB_TRY(hyb_conv3x3_fwd(dtype, 0, dyraw, wd, dx, nullptr, nullptr, N, H, W, Co, Cop, Cip, stream));
    }
    return 0;
}

// ----------------------------------------------------------------------------------------------------------
// TransformerEncoder
// ----------------------------------------------------------------------------------------------------------
// Internal (fused.hip): where the first layer's input lives inside the saved blob -- a producer that writes there saves the copy
size_t hyb_encoder_xin_offset(int dtype, int B, int S, int D, int Hid, int H) { return enc_layout(dtype, B, S, D, Hid, H).x_in; }

extern "C" size_t hyb_encoder_saved_bytes(int dtype, int B, int S, int D, int Hid, int L, int H) {
    if (B <= 0 || S <= 0 || D <= 0 || Hid <= 0 || L <= 0 || H <= 0) return 0;
    return enc_layout(dtype, B, S, D, Hid, H).layer_bytes * (size_t)L;
}
extern "C" size_t hyb_encoder_workspace_bytes(int dtype, int B, int S, int D, int Hid, int L, int H) {
    if (B <= 0 || S <= 0 || D <= 0 || Hid <= 0 || L <= 0 || H <= 0) return 0;
    const size_t es = dtype == HYB_F32 ? 4 : 2;
    const size_t M = (size_t)B * S;
    const size_t big = (size_t)(Hid > D ? Hid : D);
    // (the buffers a layer's weight gradients read -- g1, dqkv, dh, g1b, LayerNorm partial rows -- exist twice, one set per layer parity)
    return 12 * align256(M * D * es) + 4 * align256(M * big * es) + 2 * align256((size_t)2 * 32 * 2 * D * sizeof(float)) +
           (S > 64 ? align256(hyb_attention_long_workspace(dtype, B, S, D, H)) : 0);      // scratch of the long-sequence attention backward
}

int hyb_encoder_fwd_impl(int dtype, const void* x, const float* mask, const float* const* params, void* out, void* saved, int B, int S,
                         int D, int Hid, int L, int H, float attn_p, float layer_p, unsigned long long seed, const unsigned long long* seed_inc, void* stream,
                         HybEncTail* tail);
extern "C" int hyb_encoder_fwd(int dtype, const void* x, const float* mask, const float* const* params, void* out, void* saved, int B, int S,
                               int D, int Hid, int L, int H, float attn_p, float layer_p, unsigned long long seed, const unsigned long long* seed_inc, void* stream) {
    return hyb_encoder_fwd_impl(dtype, x, mask, params, out, saved, B, S, D, Hid, L, H, attn_p, layer_p, seed, seed_inc, stream, nullptr);
}
// tail != NULL (hyb_temporal_fwd): the LAST layer's second LayerNorm is not launched -- the caller's fused tail launch (layernorm.hip:
// LayerNorm + head + loss) runs it; *tail receives its operands.
int hyb_encoder_fwd_impl(int dtype, const void* x, const float* mask, const float* const* params, void* out, void* saved, int B, int S,
                         int D, int Hid, int L, int H, float attn_p, float layer_p, unsigned long long seed, const unsigned long long* seed_inc, void* stream,
                         HybEncTail* tail) {
    HYB_CHECK_ARG(x && params && out && saved && B > 0 && S > 0 && D > 0 && Hid > 0 && L > 0 && H > 0 && D % H == 0);
    HYB_CHECK_ARG(dtype == HYB_F32 || dtype == HYB_BF16);
    HYB_CHECK_ARG(D % 8 == 0 && Hid % 8 == 0);
    const size_t es = dtype == HYB_F32 ? 4 : 2;
    const int M = B * S;
    const EncLayout lay = enc_layout(dtype, B, S, D, Hid, H);
    hipStream_t st = (hipStream_t)stream;
    char* sv = (char*)saved;
    if ((const void*)x != (const void*)(sv + lay.x_in))       // (hyb_temporal_fwd writes the tokens there directly)
        HYB_HIP_TRY(hipMemcpyAsync(sv + lay.x_in, x, (size_t)M * D * es, hipMemcpyDeviceToDevice, st));
    const bool wc_master = dtype == HYB_F32;                 // fp32 storage: the forward reads the master weights in place (16-byte fragment loads)
    for (int i = 0; i < L && wc_master; ++i)
        for (int j = 0; j < 6; ++j) HYB_CHECK_ARG((uintptr_t)params[(size_t)i * 14 + 2 * j] % 16 == 0);
    if (L <= 3) {   // the weight copies of every layer in ONE launch (they depend on the master weights only)
        const float* Wsrc[18]; void* Wc[18]; void* Wt[18]; int Ns[18], Ks[18], ldt[18];
        for (int i = 0; i < L; ++i) {
            char* base = sv + (size_t)i * lay.layer_bytes;
            const float* const* P = params + (size_t)i * 14;
            const int n6[6] = {D, D, D, D, Hid, D}, k6[6] = {D, D, D, D, D, Hid}, l6[6] = {3 * D, 3 * D, 3 * D, D, Hid, D};
            for (int j = 0; j < 6; ++j) {
                Wsrc[i * 6 + j] = P[2 * j]; Wc[i * 6 + j] = wc_master ? nullptr : base + lay.wc[j]; Wt[i * 6 + j] = base + lay.wt[j];
                Ns[i * 6 + j] = n6[j]; Ks[i * 6 + j] = k6[j]; ldt[i * 6 + j] = l6[j];
            }
        }
        HYB_TRY(hyb_convert_weights(dtype, 6 * L, Wsrc, Wc, Wt, Ns, Ks, ldt, st));
    }
    bool ln2_pending = false;
    for (int i = 0; i < L; ++i) {
        char* base = sv + (size_t)i * lay.layer_bytes;
        const float* const* P = params + (size_t)i * 14;
        void* x_in = base + lay.x_in;
        void* y_out = (i == L - 1) ? out : (void*)(base + lay.layer_bytes + lay.x_in);
        if (L > 3) {   // fp32 master weights -> T copies (plain for forward, transposed for dX); deep stacks: one launch per layer
            const float* Wsrc[6] = {P[0], P[2], P[4], P[6], P[8], P[10]};
            void* Wc[6]; void* Wt[6];
            for (int j = 0; j < 6; ++j) { Wc[j] = wc_master ? nullptr : base + lay.wc[j]; Wt[j] = base + lay.wt[j]; }
            const int Ns[6] = {D, D, D, D, Hid, D}, Ks[6] = {D, D, D, D, D, Hid};
            const int ldt[6] = {3 * D, 3 * D, 3 * D, D, Hid, D};
            HYB_TRY(hyb_convert_weights(dtype, 6, Wsrc, Wc, Wt, Ns, Ks, ldt, st));
        }
        // the forward's weight operands: the T copies, or -- fp32 storage -- the master weights themselves (no copy is written)
        auto WC = [&](int j) -> const void* { return wc_master ? (const void*)P[2 * j] : (const void*)(base + lay.wc[j]); };
        const void* Wq3[3] = {WC(0), WC(1), WC(2)};
        const float* bs[3] = {P[1], P[3], P[5]};
        void* ys[3] = {base + lay.qkv, base + lay.qkv + (size_t)D * es, base + lay.qkv + 2 * (size_t)D * es};
        // Q | K | V projections (src L69-70).  From the second layer on their input is the previous layer's second LayerNorm (src L120-123), which
        // the projection launch forms for itself when the shape allows (hyb_gemm_nt_ln; it also writes x_in and the statistics): ln2_pending
        bool projected = false;
        if (ln2_pending) {
            const char* pb = sv + (size_t)(i - 1) * lay.layer_bytes;
            const float* const* PP = params + (size_t)(i - 1) * 14;
            const int rc = hyb_gemm_nt_ln(dtype, 3, pb + lay.f, pb + lay.x1, PP[12], PP[13], x_in, (float*)(pb + lay.st2), 1e-5f, (float)sqrt(0.5), layer_p,
                                          drop_seed(seed, i - 1), seed_inc, Wq3, ys, bs, M, D, D, D, 3 * D, 1, st);
            if (rc == 0) projected = true;
            else if (rc != -100) return rc;
            else HYB_TRY(hyb_ln_residual_fwd_inc(dtype, pb + lay.f, pb + lay.x1, PP[12], PP[13], x_in, (float*)(pb + lay.st2), M, D, 1e-5f,
                                                 (float)sqrt(0.5), layer_p, drop_seed(seed, i - 1), seed_inc, stream));
            ln2_pending = false;
        }
        if (!projected) {
            const void* xs[3] = {x_in, x_in, x_in};
            HYB_TRY(hyb_gemm_nt(dtype, 3, xs, Wq3, ys, bs, 0, M, D, D, D, D, 3 * D, 1, 0, st));
        }
        if (S > 64)
            HYB_TRY(hyb_attention_long_fwd(dtype, base + lay.qkv, base + lay.qkv + (size_t)D * es, base + lay.qkv + 2 * (size_t)D * es, 3 * D, mask,
                                           base + lay.attn, (float*)(base + lay.probs), B, S, D, H, attn_p, attn_seed(seed, i), seed_inc,
                                           base + lay.long_ws, lay.long_ws_bytes, st));
        else
        HYB_TRY(hyb_attention_fwd_packed(dtype, base + lay.qkv, mask, base + lay.attn, (float*)(base + lay.probs), B, S, D, H, attn_p,
                                         attn_seed(seed, i), seed_inc, st));                                          // src L73-84
        { const void* A_[1] = {base + lay.attn}; const void* B_[1] = {WC(3)}; void* C_[1] = {base + lay.o}; const float* b_[1] = {P[7]};
          HYB_TRY(hyb_gemm_nt(dtype, 1, A_, B_, C_, b_, 0, M, D, D, D, D, D, 0, 0, st)); }                            // src L87
        // first LayerNorm (src L116-117) + the feed-forward's first Linear with ReLU (src L119): one launch when the shape allows
        {
            const void* B_[1] = {WC(4)}; void* C_[1] = {base + lay.hmid}; const float* b_[1] = {P[9]};
            const int rc = hyb_gemm_nt_ln(dtype, 1, base + lay.o, x_in, P[12], P[13], base + lay.x1, (float*)(base + lay.st1), 1e-5f, 1.0f, 0.f, 0ull,
                                          nullptr, B_, C_, b_, M, Hid, D, D, Hid, 1, st);
            if (rc == -100) {
                HYB_TRY(hyb_ln_residual_fwd_inc(dtype, base + lay.o, x_in, P[12], P[13], base + lay.x1, (float*)(base + lay.st1), M, D, 1e-5f, 1.0f, 0.f,
                                                0ull, nullptr, stream));
                const void* A_[1] = {base + lay.x1};
                HYB_TRY(hyb_gemm_nt(dtype, 1, A_, B_, C_, b_, 0, M, Hid, D, D, D, Hid, 1, 0, st));
            } else if (rc != 0) return rc;
        }
        { const void* A_[1] = {base + lay.hmid}; const void* B_[1] = {WC(5)}; void* C_[1] = {base + lay.f}; const float* b_[1] = {P[11]};
          HYB_TRY(hyb_gemm_nt(dtype, 1, A_, B_, C_, b_, 0, M, D, Hid, Hid, Hid, D, 0, 0, st)); }                      // src L119 (Linear)
        if (tail && i == L - 1) {
            *tail = HybEncTail{base + lay.f, base + lay.x1, (float*)(base + lay.st2), P[12], P[13], 1e-5f, (float)sqrt(0.5), layer_p, drop_seed(seed, i)};
            break;
        }
        if (i < L - 1) { ln2_pending = true; continue; }           // src L120-123: formed by the next layer's projection launch (above)
        HYB_TRY(hyb_ln_residual_fwd_inc(dtype, base + lay.f, base + lay.x1, P[12], P[13], y_out, (float*)(base + lay.st2), M, D, 1e-5f,
                                        (float)sqrt(0.5), layer_p, drop_seed(seed, i), seed_inc, stream));                          // src L120-123
    }
    return 0;
}

int hyb_encoder_bwd_impl(int dtype, const void* dout, const float* mask, const float* const* params, float* const* grads,
                         const void* saved, void* dx, int B, int S, int D, int Hid, int L, int H, float attn_p, float layer_p,
                         unsigned long long seed, const unsigned long long* seed_inc, void* workspace, size_t workspace_bytes, void* stream,
                         const HybDwExtra* extra, int tail_done, const HybDwRider* extra_rider);
// Where the backward of the LAST layer's second LayerNorm reads and writes (hyb_temporal_bwd runs it inside its fused tail launch and then
// calls hyb_encoder_bwd_impl with tail_done = 1): the same buffers hyb_ln_residual_bwd_rows is given below.
HybEncBwdTail hyb_encoder_bwd_tail(int dtype, const float* const* params, const void* saved, void* workspace, int B, int S, int D, int Hid, int L, int H,
                                   float layer_p, unsigned long long seed) {
    const size_t es = dtype == HYB_F32 ? 4 : 2;
    const int M = B * S, i = L - 1;
    const EncLayout lay = enc_layout(dtype, B, S, D, Hid, H);
    const char* base = (const char*)saved + (size_t)i * lay.layer_bytes;
    char* ws = (char*)workspace;
    const size_t md = align256((size_t)M * D * es);
    const size_t big = align256((size_t)M * (Hid > D ? Hid : D) * es);
    const size_t lnb = align256((size_t)2 * 32 * 2 * D * sizeof(float));
    char* q = ws + 4 * md + (size_t)(i & 1) * (md + 3 * md + 2 * big + lnb);        // parity set of the last layer (see hyb_encoder_bwd_impl)
    HybEncBwdTail t;
    t.f = base + lay.f; t.stats = (const float*)(base + lay.st2); t.gamma = params[(size_t)i * 14 + 12];
    t.dx = q;                                     // set.g1
    t.dskip = ws;                                 // g2
    t.ln_part = (float*)(q + md + 3 * md + 2 * big);
    t.ln_rows = hyb_ln_bwd_rows(M);
    t.out_scale = (float)sqrt(0.5); t.p_drop = layer_p; t.seed = drop_seed(seed, i);
    return t;
}
extern "C" int hyb_encoder_bwd(int dtype, const void* dout, const float* mask, const float* const* params, float* const* grads,
                               const void* saved, void* dx, int B, int S, int D, int Hid, int L, int H, float attn_p, float layer_p,
                               unsigned long long seed, const unsigned long long* seed_inc, void* workspace, size_t workspace_bytes, void* stream) {
    return hyb_encoder_bwd_impl(dtype, dout, mask, params, grads, saved, dx, B, S, D, Hid, L, H, attn_p, layer_p, seed, seed_inc, workspace,
                                workspace_bytes, stream, nullptr, 0, nullptr);
}
// extra != NULL: one more weight gradient whose dy is this function's dx (the frame-token projection of hyb_temporal_bwd) rides in the last
// multi-matrix launch.  With L <= 2 the weight gradients of ALL layers are that one launch at the end (each layer's operands live in its own
// parity set of the workspace until then): every dependent launch costs >= 4.6 us, and two 768-tile launches fill the chip worse than one.
int hyb_encoder_bwd_impl(int dtype, const void* dout, const float* mask, const float* const* params, float* const* grads,
                         const void* saved, void* dx, int B, int S, int D, int Hid, int L, int H, float attn_p, float layer_p,
                         unsigned long long seed, const unsigned long long* seed_inc, void* workspace, size_t workspace_bytes, void* stream,
                         const HybDwExtra* extra, int tail_done, const HybDwRider* extra_rider) {
    // tail_done: the caller has already run the last layer's LN2 backward (fused tail launch) into the buffers hyb_encoder_bwd_tail names;
    // extra_rider: one more fixed-order row sum (the head's weight / bias gradient terms) for the final multi-matrix launch
    HYB_CHECK_ARG((dout || tail_done) && params && grads && saved && dx && workspace && B > 0 && S > 0 && D > 0 && Hid > 0 && L > 0 && H > 0 && D % H == 0);
    HYB_CHECK_ARG(dtype == HYB_F32 || dtype == HYB_BF16);
    if (workspace_bytes < hyb_encoder_workspace_bytes(dtype, B, S, D, Hid, L, H)) return HYB_E_WORKSPACE;
    const size_t es = dtype == HYB_F32 ? 4 : 2;
    const int M = B * S;
    const EncLayout lay = enc_layout(dtype, B, S, D, Hid, H);
    hipStream_t st = (hipStream_t)stream;
    const char* sv = (const char*)saved;
    char* ws = (char*)workspace;
    const size_t md = align256((size_t)M * D * es);
    const size_t big = align256((size_t)M * (Hid > D ? Hid : D) * es);
    const size_t lnb = align256((size_t)2 * 32 * 2 * D * sizeof(float));
    const int lnrows = hyb_ln_bwd_rows(M);
    // shared by all layers (consumed inside a layer's dX chain)
    void* g2 = ws;            // d(x1)
    void* g4 = ws + md;       // d(attn)
    void* gin[2] = {ws + 2 * md, ws + 3 * md};
    // per layer parity: what the layer's weight-gradient launch reads
    struct Set { void* g1; void* dqkv; void* dh; void* g1b; float* lnpart; } set[2];
    {
        char* q = ws + 4 * md;
        for (int j = 0; j < 2; ++j) {
            set[j].g1 = q; q += md;                 // d(LN2 input)
            set[j].dqkv = q; q += 3 * md;           // d(q|k|v) packed [M][3D]
            set[j].dh = q; q += big;                // d(hmid)
            set[j].g1b = q; q += big;               // d(LN1 input)
            set[j].lnpart = (float*)q; q += lnb;    // LayerNorm affine-gradient partial rows of the layer's two calls
        }
    }
    void* const long_ws = ws + 12 * md + 4 * big + 2 * lnb;       // S > 64 only (hyb_encoder_workspace_bytes)
    // (The layer's weight-gradient launch is off the dX chain; issuing it as a parallel branch of the replayed graph was measured in
    // round 2 and lost 4.7 % of the step -- every fork / join edge costs more than the 5 us kernel it hides -- so it stays in line.)

    const bool defer = L <= 2;                              // (two parity sets: the operands of both layers stay valid until the end)
    const void* d_dy[13]; const void* d_mk[13]; const void* d_x[13]; float* d_dW[13]; float* d_db[13];
    int d_N[13], d_K[13], d_lddy[13], d_ldx[13], nd = 0;
    HybDwRider d_rd[3]; int nrd = 0;
    const void* gA = dout;
    for (int i = L - 1; i >= 0; --i) {
        const char* base = sv + (size_t)i * lay.layer_bytes;
        const float* const* P = params + (size_t)i * 14;
        float* const* G = grads + (size_t)i * 14;
        void* gx = (i == 0) ? dx : gin[i & 1];
        const Set& b = set[i & 1];
        // LN2 + residual + sqrt(.5) + dropout
        if (!(tail_done && i == L - 1))
        HYB_TRY(hyb_ln_residual_bwd_rows(dtype, gA, base + lay.f, P[12], (const float*)(base + lay.st2), b.g1, g2, 0, b.lnpart, M, D,
                                         (float)sqrt(0.5), layer_p, drop_seed(seed, i), seed_inc, st));
        // FFN second Linear: dX = g1 . W2 (pre-transposed copy)
        // (its epilogue applies the ReLU backward of the first Linear, dh *= (hmid > 0): the two products that read dh need no mask operand)
        { const void* A_[1] = {b.g1}; const void* B_[1] = {base + lay.wt[5]}; void* C_[1] = {b.dh}; const void* CM_[1] = {base + lay.hmid};
          HYB_TRY(hyb_gemm_nt(dtype, 1, A_, B_, C_, nullptr, 0, M, Hid, D, D, D, Hid, 0, 0, st, nullptr, CM_)); }
        // FFN first Linear (+ReLU, applied above)
        { const void* A_[1] = {b.dh}; const void* B_[1] = {base + lay.wt[4]}; void* C_[1] = {g2};
          HYB_TRY(hyb_gemm_nt(dtype, 1, A_, B_, C_, nullptr, 0, M, D, Hid, Hid, Hid, D, 0, 1, st)); }
        // LN1 + residual
        HYB_TRY(hyb_ln_residual_bwd_rows(dtype, g2, base + lay.o, P[12], (const float*)(base + lay.st1), b.g1b, gx, 0,
                                         b.lnpart + (size_t)lnrows * 2 * D, M, D, 1.0f, 0.f, 0ull, nullptr, st));
        // output projection
        { const void* A_[1] = {b.g1b}; const void* B_[1] = {base + lay.wt[3]}; void* C_[1] = {g4};
          HYB_TRY(hyb_gemm_nt(dtype, 1, A_, B_, C_, nullptr, 0, M, D, D, D, D, D, 0, 0, st)); }
        // attention core: d(q|k|v) packed [M][3D]
        if (S > 64)
            HYB_TRY(hyb_attention_long_bwd(dtype, base + lay.qkv, base + lay.qkv + (size_t)D * es, base + lay.qkv + 2 * (size_t)D * es, 3 * D, mask,
                                           base + lay.attn, (const float*)(base + lay.probs), g4, b.dqkv, (char*)b.dqkv + (size_t)D * es,
                                           (char*)b.dqkv + 2 * (size_t)D * es, 3 * D, B, S, D, H, attn_p, attn_seed(seed, i), seed_inc,
                                           long_ws, lay.long_ws_bytes, st));
        else
        // (short sequences: the attention backward zeroes d(q|k|v) where the projections' ReLU was off, so their consumers need no mask)
        HYB_TRY(hyb_attention_bwd_packed(dtype, base + lay.qkv, mask, (const float*)(base + lay.probs), g4, b.dqkv, B, S, D, H, attn_p,
                                         attn_seed(seed, i), seed_inc, st, 1));
        // Q, K, V projections (+ReLU) share the layer input: one K-concatenated dX GEMM
        { const void* A_[1] = {b.dqkv}; const void* B_[1] = {base + lay.wt[0]}; void* C_[1] = {gx}; const void* M_[1] = {base + lay.qkv};
          HYB_TRY(hyb_gemm_nt(dtype, 1, A_, B_, C_, nullptr, 0, M, D, 3 * D, 3 * D, 3 * D, D, 0, 1, st, S > 64 ? M_ : nullptr)); }
        // the six weight (+ bias) gradients of the layer in ONE launch (768 tiles at config 2 instead of four 64-256-tile launches); the
        // layer's one LayerNorm is applied twice (quirk Q3): both calls' partial rows -> its weight/bias gradients ride in the same launch
        {
            const char* dq_ = (const char*)b.dqkv; const char* qk_ = base + lay.qkv;
            const void* dy_[6] = {b.g1, b.dh, b.g1b, dq_, dq_ + (size_t)D * es, dq_ + 2 * (size_t)D * es};
            const bool qm = S > 64;                        // (the long-sequence attention backward leaves d(q|k|v) unmasked)
            const void* mk_[6] = {nullptr, nullptr, nullptr, qm ? qk_ : nullptr, qm ? qk_ + (size_t)D * es : nullptr, qm ? qk_ + 2 * (size_t)D * es : nullptr};
            const void* x_[6] = {base + lay.hmid, base + lay.x1, base + lay.attn, base + lay.x_in, base + lay.x_in, base + lay.x_in};
            float* dW_[6] = {G[10], G[8], G[6], G[0], G[2], G[4]};
            float* db_[6] = {G[11], G[9], G[7], G[1], G[3], G[5]};
            const int N_[6] = {D, Hid, D, D, D, D}, K_[6] = {Hid, D, D, D, D, D};
            const int lddy_[6] = {D, Hid, D, 3 * D, 3 * D, 3 * D}, ldx_[6] = {Hid, D, D, D, D, D};
            if (defer) {
                const int o = nd * 6;
                for (int j = 0; j < 6; ++j) {
                    d_dy[o + j] = dy_[j]; d_mk[o + j] = mk_[j]; d_x[o + j] = x_[j]; d_dW[o + j] = dW_[j]; d_db[o + j] = db_[j];
                    d_N[o + j] = N_[j]; d_K[o + j] = K_[j]; d_lddy[o + j] = lddy_[j]; d_ldx[o + j] = ldx_[j];
                }
                d_rd[nrd++] = HybDwRider{b.lnpart, G[12], G[13], 2 * lnrows, 2ll * D, (long long)D};
                ++nd;
            } else {
                const HybDwRider rd1{b.lnpart, G[12], G[13], 2 * lnrows, 2ll * D, (long long)D};
                HYB_TRY(hyb_linear_dw_multi(dtype, 6, dy_, mk_, x_, dW_, db_, N_, K_, lddy_, ldx_, M, st, 1, &rd1));
            }
        }
        gA = gx;
    }
    int ng = defer ? nd * 6 : 0;
    if (extra) {
        d_dy[ng] = extra->dy; d_mk[ng] = nullptr; d_x[ng] = extra->x; d_dW[ng] = extra->dW; d_db[ng] = extra->db;
        d_N[ng] = extra->N; d_K[ng] = extra->K; d_lddy[ng] = extra->lddy; d_ldx[ng] = extra->ldx;
        ++ng;
    }
    if (extra_rider) d_rd[nrd++] = *extra_rider;
    if (ng > 0)
        HYB_TRY(hyb_linear_dw_multi(dtype, ng, d_dy, d_mk, d_x, d_dW, d_db, d_N, d_K, d_lddy, d_ldx, M, st, nrd, d_rd));
    else if (nrd > 0) return HYB_E_ARG;              // (callers pass a rider only when a final launch exists: L <= 2 or a riding weight gradient)
    return 0;
}
