// Model-level entry points: the whole CNN backbone and the whole temporal part (frame tokens -> TransformerEncoder -> head) as
// ONE C call each way.  They only chain the stage-level entry points of this library on the caller's stream; what they save is
// host time: a training step becomes three operator calls each way instead of seven, which keeps the step GPU-bound.
//   hyb_backbone_{fwd,bwd} : `stages` x [Conv3x3 -> BatchNorm2d -> ReLU -> MaxPool2d]   (UNet.py:58-60 + UNet.py:13, UNet.py:32-37 order)
//   hyb_temporal_{fwd,bwd} : global-average-pool + Linear frame token (composite's own) -> TransformerEncoder.forward
//                            (TransformerEncoder.pyc src L110-126) -> mean over T + Linear head (composite's own)
#include <stdlib.h>
#include "hyb_common.h"

size_t hyb_encoder_xin_offset(int dtype, int B, int S, int D, int Hid, int H);
int hyb_encoder_bwd_impl(int dtype, const void* dout, const float* mask, const float* const* params, float* const* grads,
                         const void* saved, void* dx, int B, int S, int D, int Hid, int L, int H, float attn_p, float layer_p,
                         unsigned long long seed, const unsigned long long* seed_inc, void* workspace, size_t workspace_bytes, void* stream,
                         const HybDwExtra* extra, int tail_done, const HybDwRider* extra_rider);
int hyb_encoder_fwd_impl(int dtype, const void* x, const float* mask, const float* const* params, void* out, void* saved, int B, int S,
                         int D, int Hid, int L, int H, float attn_p, float layer_p, unsigned long long seed, const unsigned long long* seed_inc, void* stream,
                         HybEncTail* tail);
HybEncBwdTail hyb_encoder_bwd_tail(int dtype, const float* const* params, const void* saved, void* workspace, int B, int S, int D, int Hid, int L, int H,
                                   float layer_p, unsigned long long seed);
int hyb_ln_bwd_rows(int M);
int hyb_temporal_tail_ok(int B, int S, int D, int C, int ln_rows);
int hyb_temporal_tail_fwd(int dtype, const void* f, const void* x1, const float* gamma, const float* beta, void* enc_out, float* stats, int B, int S,
                          int D, float eps, float out_scale, float p_drop, unsigned long long seed, const unsigned long long* seed_inc, const float* W,
                          const float* bias, float* logits, int C, const long long* target, float* loss, float* ce_scratch, hipStream_t st);
int hyb_temporal_tail_bwd(int dtype, const float* dlogits, const float* logits, const long long* target, const float* dloss, const float* W,
                          const void* enc_out, const void* f, const float* gamma, const float* stats, void* dx, void* dskip, float* ln_part,
                          int ln_rows, float* head_part, int B, int S, int D, int C, float out_scale, float p_drop, unsigned long long seed,
                          const unsigned long long* seed_inc, hipStream_t st);
int hyb_convstage_fwd_impl(int dtype, int first, const void* x, const float* weight, const float* gamma, const float* beta,
                           float* running_mean, float* running_var, long long* nbt, int training, float momentum, float eps,
                           int N, int H, int W, int Ci, int Cip, int Co, int Cop, void* y_raw, void* pooled, float* scale_shift,
                           float* mean_invstd, void* packed_bwd, float* running_out, void* workspace, size_t workspace_bytes, void* stream,
                           const void* prepacked_fwd);
int hyb_convstage_bwd_impl(int dtype, int first, const void* dpooled, const void* x, const void* y_raw, const void* pooled, const float* weight,
                           const float* gamma, const float* scale_shift, const float* mean_invstd, int training, int N, int H, int W,
                           int Ci, int Cip, int Co, int Cop, void* dx, float* dweight, float* dgamma, float* dbeta,
                           const void* packed_bwd, void* workspace, size_t workspace_bytes, void* stream, void* slab_ws, HybSlabInfo* defer);
int hyb_wgrad_reduce_multi(int n, const HybSlabInfo* infos, hipStream_t st);
int hyb_conv_pack_weight_many(int dtype, int n, const float* const* w, void* const* wp0, void* const* wp1, const int* Co, const int* Ci, const int* Cop,
                              const int* Cip, const float* s1_w, void* s1_wp, int s1_Co, int s1_Ci, int s1_Cop, hipStream_t st);

namespace {
inline size_t al256(size_t v) { return (v + 255) & ~(size_t)255; }
inline size_t smax(size_t a, size_t b) { return a > b ? a : b; }
#define HYB_TRY(call) do { int rc_ = (call); if (rc_ != 0) return rc_; } while (0)
inline int padc(int c) { return (c + 31) / 32 * 32; }
}  // namespace

extern "C" size_t hyb_backbone_fwd_workspace(int dtype, int stages, const int* channels) {
    if (stages < 1 || !channels) return 0;
    size_t b = 0;
    for (int s = 0; s < stages; ++s) b = smax(b, hyb_convstage_fwd_workspace(dtype, s == 0, s == 0 ? 0 : padc(channels[s]), padc(channels[s + 1])));
    // + one forward weight pack per non-first stage: all of them are written by ONE launch before the first convolution
    const size_t es = dtype == HYB_F32 ? 4 : 2;
    size_t packs = 0;
    for (int s = 1; s < stages; ++s) packs += al256((size_t)hyb_conv_packed_elems(0, padc(channels[s]), padc(channels[s + 1])) * es);
    return al256(b) + packs;
}

extern "C" int hyb_backbone_fwd(int dtype, int stages, const int* channels, const float* x, const float* const* params,
                                long long* const* num_batches_tracked, int training, float momentum, float eps, int N, int H, int W,
                                void* const* outs, void* workspace, size_t workspace_bytes, void* stream) {
    HYB_CHECK_ARG(stages >= 1 && stages <= 16 && channels && x && params && outs && workspace && N > 0);
    HYB_CHECK_ARG(channels[0] >= 1 && channels[0] <= 4);                  // the first stage reads NCHW fp32 frames directly
    if (workspace_bytes < hyb_backbone_fwd_workspace(dtype, stages, channels)) return HYB_E_WORKSPACE;
    const void* in = x;
    int h = H, w = W;
    // weight packs of stages 1.. (forward layout into the workspace behind the per-stage scratch, backward layout into the caller's saved
    // packed_bwd buffers) in one launch; a stage without a packed_bwd buffer packs by itself as before
    const size_t es = dtype == HYB_F32 ? 4 : 2;
    size_t stage_ws = 0;
    for (int s = 0; s < stages; ++s) stage_ws = smax(stage_ws, hyb_convstage_fwd_workspace(dtype, s == 0, s == 0 ? 0 : padc(channels[s]), padc(channels[s + 1])));
    stage_ws = al256(stage_ws);
    const void* prepacked[17] = {nullptr};
    {
        const float* pw[16]; void* p0[16]; void* p1[16]; int co[16], ci[16], cop[16], cip[16];
        char* q = (char*)workspace + stage_ws;
        int n = 0;
        for (int s = 1; s < stages; ++s) {
            void* bwdpack = outs[(size_t)s * 6 + 4];
            void* fwdpack = q;
            q += al256((size_t)hyb_conv_packed_elems(0, padc(channels[s]), padc(channels[s + 1])) * es);
            if (!bwdpack) continue;
            pw[n] = params[(size_t)s * 5]; p0[n] = fwdpack; p1[n] = bwdpack;
            co[n] = channels[s + 1]; ci[n] = channels[s]; cop[n] = padc(channels[s + 1]); cip[n] = padc(channels[s]);
            prepacked[s] = fwdpack;
            ++n;
        }
        // the first stage's two layouts ([2][Cop][64] at the head of its packed_bwd buffer) ride in the same launch
        void* s1pack = outs[4];
        if (s1pack) prepacked[0] = s1pack;
        if (n > 0 || s1pack)
            HYB_TRY(hyb_conv_pack_weight_many(dtype, n, pw, p0, p1, co, ci, cop, cip, params[0], s1pack, channels[1], channels[0], padc(channels[1]),
                                              (hipStream_t)stream));
    }
    for (int s = 0; s < stages; ++s) {
        HYB_CHECK_ARG(h >= 2 && w >= 2);
        const float* const* P = params + (size_t)s * 5;
        void* const* O = outs + (size_t)s * 6;
        const int Ci = channels[s], Co = channels[s + 1];
        // the per-stage scratch is reused by every stage: the stages are ordered on the stream
        // training: running_out (O[5]) != NULL -> functional BatchNorm (the updated statistics go there, the inputs stay untouched);
        // NULL -> nn.BatchNorm2d's own in-place update of running_mean / running_var and num_batches_tracked += 1
        HYB_TRY(hyb_convstage_fwd_impl(dtype, s == 0, in, P[0], P[1], P[2], (float*)P[3], (float*)P[4],
                                       num_batches_tracked ? num_batches_tracked[s] : nullptr, training, momentum, eps, N, h, w, Ci,
                                       s == 0 ? 0 : padc(Ci), Co, padc(Co), O[0], O[1], (float*)O[2], (float*)O[3], O[4], training ? (float*)O[5] : nullptr,
                                       workspace, stage_ws, stream, prepacked[s]));
        in = O[1];
        h /= 2; w /= 2;
    }
    return 0;
}

// The weight-gradient slabs of stages 2.. outlive their stage: their fixed-order sums are ONE launch at the end of the backward (they feed only
// the optimizer).  Up to four stages are deferred (the first ones met walking backwards); offsets[s] = byte offset of stage s's region.
static size_t backbone_slab_bytes(int stages, const int* channels, int N, int H, int W, size_t* offsets) {
    size_t total = 0;
    int hs[17], wsz[17];
    { int h = H, w = W; for (int s = 0; s < stages; ++s) { hs[s] = h; wsz[s] = w; h /= 2; w /= 2; } }
    int deferred = 0;
    for (int s = stages - 1; s >= 1; --s) {
        if (offsets) offsets[s] = (size_t)-1;
        if (deferred >= 4) continue;
        if (offsets) offsets[s] = total;
        total += al256(hyb_conv3x3_wgrad_workspace(0, N, hs[s], wsz[s], padc(channels[s]), padc(channels[s + 1])));
        ++deferred;
    }
    return total;
}

extern "C" size_t hyb_backbone_bwd_workspace(int dtype, int stages, const int* channels, int N, int H, int W) {
    if (stages < 1 || !channels || N <= 0) return 0;
    const size_t es = dtype == HYB_F32 ? 4 : 2;
    size_t stage_ws = 0, dx_bytes = 0;
    int h = H, w = W;
    for (int s = 0; s < stages; ++s) {
        stage_ws = smax(stage_ws, hyb_convstage_bwd_workspace(dtype, s == 0, N, h, w, s == 0 ? 0 : padc(channels[s]), padc(channels[s + 1])));
        if (s > 0) dx_bytes = smax(dx_bytes, (size_t)N * h * w * padc(channels[s]) * es);      // d(input of stage s) = d(pooled of stage s-1)
        h /= 2; w /= 2;
    }
    return al256(stage_ws) + 2 * al256(dx_bytes) + backbone_slab_bytes(stages, channels, N, H, W, nullptr);
}

extern "C" int hyb_backbone_bwd(int dtype, int stages, const int* channels, const void* dpooled_last, const void* pooled_last, const float* x,
                                const float* const* params,
                                const void* const* saved, int training, int N, int H, int W, float* const* grads, void* workspace,
                                size_t workspace_bytes, void* stream) {
    HYB_CHECK_ARG(stages >= 1 && stages <= 16 && channels && dpooled_last && x && params && saved && grads && workspace && N > 0);
    if (workspace_bytes < hyb_backbone_bwd_workspace(dtype, stages, channels, N, H, W)) return HYB_E_WORKSPACE;
    const size_t es = dtype == HYB_F32 ? 4 : 2;
    size_t stage_ws = 0, dx_bytes = 0;
    int hs[17], wsz[17];
    {
        int h = H, w = W;
        for (int s = 0; s < stages; ++s) {
            hs[s] = h; wsz[s] = w;
            stage_ws = smax(stage_ws, hyb_convstage_bwd_workspace(dtype, s == 0, N, h, w, s == 0 ? 0 : padc(channels[s]), padc(channels[s + 1])));
            if (s > 0) dx_bytes = smax(dx_bytes, (size_t)N * h * w * padc(channels[s]) * es);
            h /= 2; w /= 2;
        }
    }
    char* ws = (char*)workspace;
    void* dxbuf[2] = {ws + al256(stage_ws), ws + al256(stage_ws) + al256(dx_bytes)};
    size_t slab_off[17];
    backbone_slab_bytes(stages, channels, N, H, W, slab_off);
    char* const slab_base = ws + al256(stage_ws) + 2 * al256(dx_bytes);
    HybSlabInfo pending[4];
    int npending = 0;
    const void* dp = dpooled_last;
    for (int s = stages - 1; s >= 0; --s) {
        const float* const* P = params + (size_t)s * 2;          // weight, gamma
        const void* const* S = saved + (size_t)s * 5;            // y_raw, stage input, scale_shift, mean_invstd, packed_bwd
        float* const* G = grads + (size_t)s * 3;                 // dweight, dgamma, dbeta
        const int Ci = channels[s], Co = channels[s + 1];
        void* dx = s == 0 ? nullptr : dxbuf[s & 1];
        const void* pooled = s + 1 < stages ? saved[(size_t)(s + 1) * 5 + 1] : pooled_last;      // a stage's output is the next stage's saved input
        const bool can_defer = s > 0 && slab_off[s] != (size_t)-1;
        HybSlabInfo info{};
        HYB_TRY(hyb_convstage_bwd_impl(dtype, s == 0, dp, s == 0 ? (const void*)x : S[1], S[0], pooled, P[0], P[1], (const float*)S[2], (const float*)S[3],
                                       training, N, hs[s], wsz[s], Ci, s == 0 ? 0 : padc(Ci), Co, padc(Co), dx, G[0], G[1], G[2], S[4], ws, al256(stage_ws),
                                       stream, can_defer ? slab_base + slab_off[s] : nullptr, can_defer ? &info : nullptr));
        if (info.S > 0) pending[npending++] = info;
        dp = dx;
    }
    if (npending > 0) HYB_TRY(hyb_wgrad_reduce_multi(npending, pending, (hipStream_t)stream));
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------------
// (the head's per-clip weight / bias gradient rows of the fused tail, [B][C*D + C] floats, are sized for the largest class count the head takes: 64)
static size_t head_part_bytes(int B, int D) { return al256((size_t)B * ((size_t)64 * D + 64) * sizeof(float)); }

extern "C" size_t hyb_temporal_bwd_workspace(int dtype, int B, int S, int HW, int Cp, int D, int Hid, int L, int H) {
    if (B <= 0 || S <= 0 || D <= 0) return 0;
    const size_t es = dtype == HYB_F32 ? 4 : 2;
    const size_t M = (size_t)B * S;
    return al256(hyb_encoder_workspace_bytes(dtype, B, S, D, Hid, L, H)) + 2 * al256(M * D * es) + al256(M * Cp * es) + head_part_bytes(B, D);
}

// The tail of the temporal part runs as ONE launch each way when the shapes allow (hyb_temporal_tail_ok): forward = the last layer's second
// LayerNorm + head (+ cross-entropy when a target is given), backward = (cross-entropy backward +) head backward + that LayerNorm's
// backward.  HYB_TEMPORAL_TAIL=0: the separate launches (A/B; same results up to the order of two fixed-order sums).
static int tail_enabled() {
    static const int env = getenv("HYB_TEMPORAL_TAIL") ? atoi(getenv("HYB_TEMPORAL_TAIL")) : 1;
    return env;
}

int hyb_gap_fwd_h16(const void* x, float* feat, int N, int HW, int Cp, hipStream_t st);      // bn_pool.hip
int hyb_gap_bwd_h16(const float* dfeat, void* dx, int N, int HW, int Cp, hipStream_t st);

static int temporal_fwd_impl(int dtype, const void* h, const float* token_w, const float* token_b, const float* const* enc_params,
                             const float* head_w, const float* head_b, const float* mask, void* feat, void* tok, void* enc_saved,
                             void* enc_out, float* logits, int B, int S, int HW, int C, int Cp, int D, int Hid, int L, int H, int classes,
                             float attn_p, float layer_p, unsigned long long seed, const unsigned long long* seed_inc, const long long* target,
                             float* loss, float* ce_scratch, void* stream) {
    HYB_CHECK_ARG(h && token_w && enc_params && head_w && feat && tok && enc_saved && enc_out && logits && B > 0 && S > 0 && HW > 0);
    const int N = B * S;
    const bool h16 = (dtype & HYB_H_BF16) != 0;                 // the pooled map is bf16, everything from the frame features on is fp32
    dtype &= 0xff;
    HYB_CHECK_ARG(!h16 || dtype == HYB_F32);
    if (h16) HYB_TRY(hyb_gap_fwd_h16(h, (float*)feat, N, HW, Cp, (hipStream_t)stream));
    else HYB_TRY(hyb_gap_fwd(dtype, h, feat, N, HW, Cp, stream));
    // the tokens are written straight into the encoder's saved input slot (`tok` stays an unused scratch argument of the ABI)
    void* tok_dst = (char*)enc_saved + hyb_encoder_xin_offset(dtype, B, S, D, Hid, H);
    (void)tok;
    HYB_TRY(hyb_linear_fwd(dtype, feat, Cp, token_w, token_b, tok_dst, N, D, C, 0, stream));
    const bool tail = tail_enabled() && hyb_temporal_tail_ok(B, S, D, classes, hyb_ln_bwd_rows(N));
    HybEncTail t{};
    HYB_TRY(hyb_encoder_fwd_impl(dtype, tok_dst, mask, enc_params, enc_out, enc_saved, B, S, D, Hid, L, H, attn_p, layer_p, seed, seed_inc, stream,
                                 tail ? &t : nullptr));
    if (tail)
        return hyb_temporal_tail_fwd(dtype, t.f, t.x1, t.gamma, t.beta, enc_out, t.st2, B, S, D, t.eps, t.out_scale, t.p_drop, t.seed, seed_inc, head_w,
                                     head_b, logits, classes, target, loss, ce_scratch, (hipStream_t)stream);
    HYB_TRY(hyb_head_fwd(dtype, enc_out, head_w, head_b, logits, B, S, D, classes, stream));
    if (target) HYB_TRY(hyb_cross_entropy_fwd(logits, target, loss, B, classes, stream));
    return 0;
}

extern "C" int hyb_temporal_fwd(int dtype, const void* h, const float* token_w, const float* token_b, const float* const* enc_params,
                                const float* head_w, const float* head_b, const float* mask, void* feat, void* tok, void* enc_saved,
                                void* enc_out, float* logits, int B, int S, int HW, int C, int Cp, int D, int Hid, int L, int H, int classes,
                                float attn_p, float layer_p, unsigned long long seed, const unsigned long long* seed_inc, void* stream) {
    return temporal_fwd_impl(dtype, h, token_w, token_b, enc_params, head_w, head_b, mask, feat, tok, enc_saved, enc_out, logits, B, S, HW, C, Cp, D,
                             Hid, L, H, classes, attn_p, layer_p, seed, seed_inc, nullptr, nullptr, nullptr, stream);
}
extern "C" int hyb_temporal_ce_fwd(int dtype, const void* h, const float* token_w, const float* token_b, const float* const* enc_params,
                                   const float* head_w, const float* head_b, const float* mask, const long long* target, void* feat, void* tok,
                                   void* enc_saved, void* enc_out, float* logits, float* loss, float* ce_scratch, int B, int S, int HW, int C, int Cp,
                                   int D, int Hid, int L, int H, int classes, float attn_p, float layer_p, unsigned long long seed,
                                   const unsigned long long* seed_inc, void* stream) {
    HYB_CHECK_ARG(target && loss && ce_scratch);
    return temporal_fwd_impl(dtype, h, token_w, token_b, enc_params, head_w, head_b, mask, feat, tok, enc_saved, enc_out, logits, B, S, HW, C, Cp, D,
                             Hid, L, H, classes, attn_p, layer_p, seed, seed_inc, target, loss, ce_scratch, stream);
}

static int temporal_bwd_impl(int dtype, const float* dlogits, const float* logits, const long long* target, const float* dloss, const float* token_w,
                             const float* const* enc_params, const float* head_w, const float* mask, const void* feat, const void* enc_saved,
                             const void* enc_out, float* dtoken_w, float* dtoken_b, float* const* enc_grads, float* dhead_w, float* dhead_b, void* dh,
                             int B, int S, int HW, int C, int Cp, int D, int Hid, int L, int H, int classes, float attn_p, float layer_p,
                             unsigned long long seed, const unsigned long long* seed_inc, void* workspace, size_t workspace_bytes, void* stream) {
    HYB_CHECK_ARG((dlogits || (logits && target && dloss)) && token_w && enc_params && head_w && feat && enc_saved && enc_out && dtoken_w && enc_grads &&
                  dhead_w && dh && workspace);
    const bool h16 = (dtype & HYB_H_BF16) != 0;                 // dh is written as bf16
    dtype &= 0xff;
    HYB_CHECK_ARG(!h16 || dtype == HYB_F32);
    if (workspace_bytes < hyb_temporal_bwd_workspace(dtype, B, S, HW, Cp, D, Hid, L, H)) return HYB_E_WORKSPACE;
    const size_t es = dtype == HYB_F32 ? 4 : 2;
    const int N = B * S;
    const size_t enc_ws = al256(hyb_encoder_workspace_bytes(dtype, B, S, D, Hid, L, H));
    char* ws = (char*)workspace;
    void* denc = ws + enc_ws;                                   // d(encoder output)
    void* dtok = ws + enc_ws + al256((size_t)N * D * es);       // d(tokens)
    void* dfeat = ws + enc_ws + 2 * al256((size_t)N * D * es);  // d(frame features), padded channels zero
    float* head_part = (float*)(ws + enc_ws + 2 * al256((size_t)N * D * es) + al256((size_t)N * Cp * es));
    // the token projection's weight gradient (dtok^T feat) rides in the encoder backward's final multi-matrix launch
    const bool ride = C % 8 == 0;
    const HybDwExtra tokdw{dtok, feat, dtoken_w, dtoken_b, D, C, D, Cp};
    const bool tail = tail_enabled() && dhead_b && hyb_temporal_tail_ok(B, S, D, classes, hyb_ln_bwd_rows(N)) && (ride || L <= 2);
    if (tail) {
        const HybEncBwdTail t = hyb_encoder_bwd_tail(dtype, enc_params, enc_saved, ws, B, S, D, Hid, L, H, layer_p, seed);
        HYB_TRY(hyb_temporal_tail_bwd(dtype, dlogits, logits, target, dloss, head_w, enc_out, t.f, t.gamma, t.stats, t.dx, t.dskip, t.ln_part, t.ln_rows,
                                      head_part, B, S, D, classes, t.out_scale, t.p_drop, t.seed, seed_inc, (hipStream_t)stream));
        const long long cd = (long long)classes * D;
        const HybDwRider hr{head_part, dhead_w, dhead_b, B, cd + classes, cd};
        HYB_TRY(hyb_encoder_bwd_impl(dtype, nullptr, mask, enc_params, enc_grads, enc_saved, dtok, B, S, D, Hid, L, H, attn_p, layer_p, seed, seed_inc, ws,
                                     enc_ws, stream, ride ? &tokdw : nullptr, 1, &hr));
    } else {
        const float* dl = dlogits;
        if (!dl) {      // cross-entropy backward as its own launch, into the head of the (not yet used) dfeat scratch
            HYB_CHECK_ARG((size_t)B * classes * sizeof(float) <= al256((size_t)N * Cp * es));
            HYB_TRY(hyb_cross_entropy_bwd(logits, target, dloss, (float*)dfeat, B, classes, stream));
            dl = (const float*)dfeat;
        }
        HYB_TRY(hyb_head_bwd(dtype, enc_out, head_w, dl, denc, dhead_w, dhead_b, B, S, D, classes, stream));
        HYB_TRY(hyb_encoder_bwd_impl(dtype, denc, mask, enc_params, enc_grads, enc_saved, dtok, B, S, D, Hid, L, H, attn_p, layer_p, seed, seed_inc, ws,
                                     enc_ws, stream, ride ? &tokdw : nullptr, 0, nullptr));
    }
    if (Cp > C) { hipError_t e = hipMemsetAsync(dfeat, 0, (size_t)N * Cp * es, (hipStream_t)stream); if (e != hipSuccess) return (int)e; }
    HYB_TRY(hyb_linear_bwd(dtype, feat, Cp, token_w, nullptr, dtok, dfeat, 0, ride ? nullptr : dtoken_w, ride ? nullptr : dtoken_b, N, D, C, 0, nullptr, 0,
                           stream));
    if (h16) HYB_TRY(hyb_gap_bwd_h16((const float*)dfeat, dh, N, HW, Cp, (hipStream_t)stream));
    else HYB_TRY(hyb_gap_bwd(dtype, dfeat, dh, N, HW, Cp, stream));
    return 0;
}

extern "C" int hyb_temporal_bwd(int dtype, const float* dlogits, const float* token_w, const float* const* enc_params, const float* head_w,
                                const float* mask, const void* feat, const void* enc_saved, const void* enc_out, float* dtoken_w,
                                float* dtoken_b, float* const* enc_grads, float* dhead_w, float* dhead_b, void* dh, int B, int S, int HW, int C,
                                int Cp, int D, int Hid, int L, int H, int classes, float attn_p, float layer_p, unsigned long long seed,
                                const unsigned long long* seed_inc, void* workspace, size_t workspace_bytes, void* stream) {
    HYB_CHECK_ARG(dlogits);
    return temporal_bwd_impl(dtype, dlogits, nullptr, nullptr, nullptr, token_w, enc_params, head_w, mask, feat, enc_saved, enc_out, dtoken_w, dtoken_b,
                             enc_grads, dhead_w, dhead_b, dh, B, S, HW, C, Cp, D, Hid, L, H, classes, attn_p, layer_p, seed, seed_inc, workspace,
                             workspace_bytes, stream);
}
extern "C" int hyb_temporal_ce_bwd(int dtype, const float* dloss, const float* logits, const long long* target, const float* token_w,
                                   const float* const* enc_params, const float* head_w, const float* mask, const void* feat, const void* enc_saved,
                                   const void* enc_out, float* dtoken_w, float* dtoken_b, float* const* enc_grads, float* dhead_w, float* dhead_b,
                                   void* dh, int B, int S, int HW, int C, int Cp, int D, int Hid, int L, int H, int classes, float attn_p,
                                   float layer_p, unsigned long long seed, const unsigned long long* seed_inc, void* workspace, size_t workspace_bytes,
                                   void* stream) {
    HYB_CHECK_ARG(dloss && logits && target);
    return temporal_bwd_impl(dtype, nullptr, logits, target, dloss, token_w, enc_params, head_w, mask, feat, enc_saved, enc_out, dtoken_w, dtoken_b,
                             enc_grads, dhead_w, dhead_b, dh, B, S, HW, C, Cp, D, Hid, L, H, classes, attn_p, layer_p, seed, seed_inc, workspace,
                             workspace_bytes, stream);
}
