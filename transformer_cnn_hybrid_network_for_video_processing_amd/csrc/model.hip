// Stage-level entry points: each is a fixed sequence of the kernels in this library, enqueued back to back on
// the caller's stream (no allocation, no synchronisation -> capturable in a hipGraph).
//   hyb_convstage_{fwd,bwd} : Conv3x3 -> BatchNorm2d -> ReLU -> MaxPool2d      (UNet.py:58-60, UNet.py:13)
//   hyb_encoder_{fwd,bwd}   : TransformerEncoder.forward, all layers             (TransformerEncoder.pyc src L110-126)
#include <math.h>
#include "hyb_common.h"

int hyb_linear_fwd_grouped3(int dtype, const void* const* x, const float* const* W, const float* const* b, void* const* y, int groups,
                            int M, int N, int K, int relu, hipStream_t st);

namespace {

inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

#define HYB_TRY(call) do { int rc_ = (call); if (rc_ != 0) return rc_; } while (0)
#define HYB_HIP_TRY(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return (int)e_; } while (0)

struct EncLayout {       // byte offsets inside `saved` for one layer, plus per-layer stride
    size_t x_in, q, k, v, probs, attn, o, st1, x1, hmid, f, st2, layer_bytes;
};
inline EncLayout enc_layout(int dtype, int B, int S, int D, int Hid, int H) {
    const size_t es = dtype == HYB_F32 ? 4 : 2;
    const size_t M = (size_t)B * S;
    EncLayout L;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += align256(bytes); return o; };
    L.x_in = take(M * D * es);
    L.q = take(M * D * es);
    L.k = take(M * D * es);
    L.v = take(M * D * es);
    L.probs = take((size_t)B * H * S * S * 4);
    L.attn = take(M * D * es);
    L.o = take(M * D * es);
    L.st1 = take(2 * M * 4);
    L.x1 = take(M * D * es);
    L.hmid = take(M * Hid * es);
    L.f = take(M * D * es);
    L.st2 = take(2 * M * 4);
    L.layer_bytes = off;
    return L;
}

inline unsigned long long attn_seed(unsigned long long seed, int layer) { return seed + 0x9E3779B97F4A7C15ull * (unsigned long long)(2 * layer + 1); }
inline unsigned long long drop_seed(unsigned long long seed, int layer) { return seed + 0x9E3779B97F4A7C15ull * (unsigned long long)(2 * layer + 2); }

}  // namespace

// ----------------------------------------------------------------------------------------------------------
// conv stage
// ----------------------------------------------------------------------------------------------------------
extern "C" size_t hyb_convstage_fwd_workspace(int dtype, int first, int Cip, int Cop) {
    const size_t es = dtype == HYB_F32 ? 4 : 2;
    return align256((size_t)hyb_conv_packed_elems(first, Cip, Cop) * es) + align256(2 * (size_t)Cop * 4) + align256(hyb_conv_stats_workspace(Cop));
}

extern "C" int hyb_convstage_fwd(int dtype, int first, const void* x, const float* weight, const float* gamma, const float* beta,
                                 float* running_mean, float* running_var, long long* nbt, int training, float momentum, float eps,
                                 int N, int H, int W, int Ci, int Cip, int Co, int Cop, void* y_raw, void* pooled, float* scale_shift,
                                 float* mean_invstd, void* workspace, size_t workspace_bytes, void* stream) {
    HYB_CHECK_ARG(x && weight && gamma && beta && running_mean && running_var && y_raw && pooled && scale_shift && mean_invstd && workspace);
    HYB_CHECK_ARG(dtype == HYB_F32 || dtype == HYB_BF16);
    HYB_CHECK_ARG(H >= 2 && W >= 2);
    if (workspace_bytes < hyb_convstage_fwd_workspace(dtype, first, Cip, Cop)) return HYB_E_WORKSPACE;
    const size_t es = dtype == HYB_F32 ? 4 : 2;
    char* ws = (char*)workspace;
    void* wp = ws;
    float* stats = (float*)(ws + align256((size_t)hyb_conv_packed_elems(first, Cip, Cop) * es));
    float* part = (float*)((char*)stats + align256(2 * (size_t)Cop * 4));
    HYB_TRY(hyb_conv_pack_weight(dtype, first ? 2 : 0, weight, wp, Co, Ci, Cop, Cip, stream));
    HYB_TRY(hyb_conv3x3_fwd(dtype, first, x, wp, y_raw, training ? stats : nullptr, part, N, H, W, Ci, Cip, Cop, stream));
    HYB_TRY(hyb_bn_finalize(stats, gamma, beta, running_mean, running_var, nbt, training, momentum, eps, (long long)N * H * W, Co, Cop,
                            scale_shift, mean_invstd, stream));
    HYB_TRY(hyb_bn_relu_pool_fwd(dtype, y_raw, scale_shift, pooled, N, H, W, Cop, stream));
    return 0;
}

extern "C" size_t hyb_convstage_bwd_workspace(int dtype, int first, int N, int H, int W, int Cip, int Cop) {
    const size_t es = dtype == HYB_F32 ? 4 : 2;
    size_t b = align256(2 * (size_t)Cop * 4);                                     // sums
    b += align256((size_t)N * H * W * Cop * es);                                  // dense grad of the raw conv output
    if (!first) b += align256((size_t)Cip * 9 * Cop * es);                        // dgrad-packed weights
    b += align256(hyb_conv3x3_wgrad_workspace(first, N, H, W, Cip, Cop));         // wgrad slabs
    return b;
}

extern "C" int hyb_convstage_bwd(int dtype, int first, const void* dpooled, const void* x, const void* y_raw, const float* weight,
                                 const float* gamma, const float* scale_shift, const float* mean_invstd, int training, int N, int H, int W,
                                 int Ci, int Cip, int Co, int Cop, void* dx, float* dweight, float* dgamma, float* dbeta, void* workspace,
                                 size_t workspace_bytes, void* stream) {
    HYB_CHECK_ARG(dpooled && x && y_raw && weight && gamma && scale_shift && mean_invstd && dweight && workspace);
    HYB_CHECK_ARG(dtype == HYB_F32 || dtype == HYB_BF16);
    HYB_CHECK_ARG(first || dx);
    if (workspace_bytes < hyb_convstage_bwd_workspace(dtype, first, N, H, W, Cip, Cop)) return HYB_E_WORKSPACE;
    const size_t es = dtype == HYB_F32 ? 4 : 2;
    char* ws = (char*)workspace;
    float* sums = (float*)ws;                    ws += align256(2 * (size_t)Cop * 4);
    void* dyraw = ws;                            ws += align256((size_t)N * H * W * Cop * es);
    void* wpd = nullptr;
    if (!first) { wpd = ws;                      ws += align256((size_t)Cip * 9 * Cop * es); }
    void* slabs = ws;
    const size_t slab_bytes = hyb_conv3x3_wgrad_workspace(first, N, H, W, Cip, Cop);
    const long long count = (long long)N * H * W;
    HYB_HIP_TRY(hipMemsetAsync(sums, 0, 2 * (size_t)Cop * 4, (hipStream_t)stream));
    HYB_TRY(hyb_bn_relu_pool_bwd_reduce(dtype, dpooled, y_raw, scale_shift, mean_invstd, sums, N, H, W, Cop, stream));
    HYB_TRY(hyb_bn_relu_pool_bwd_dx(dtype, dpooled, y_raw, scale_shift, mean_invstd, gamma, sums, training, count, dyraw, dgamma, dbeta, N, H,
                                    W, Co, Cop, stream));
    HYB_TRY(hyb_conv3x3_wgrad(dtype, first, x, dyraw, dweight, N, H, W, Ci, Cip, Co, Cop, slabs, slab_bytes, stream));
    if (!first) {
        // dgrad = conv3x3 of the dense output gradient with the transposed, tap-flipped weights
        HYB_TRY(hyb_conv_pack_weight(dtype, 1, weight, wpd, Co, Ci, Cop, Cip, stream));
        HYB_TRY(hyb_conv3x3_fwd(dtype, 0, dyraw, wpd, dx, nullptr, nullptr, N, H, W, Co, Cop, Cip, stream));
    }
    return 0;
}

// ----------------------------------------------------------------------------------------------------------
// TransformerEncoder
// ----------------------------------------------------------------------------------------------------------
extern "C" size_t hyb_encoder_saved_bytes(int dtype, int B, int S, int D, int Hid, int L, int H) {
    if (B <= 0 || S <= 0 || D <= 0 || Hid <= 0 || L <= 0 || H <= 0) return 0;
    return enc_layout(dtype, B, S, D, Hid, H).layer_bytes * (size_t)L;
}
extern "C" size_t hyb_encoder_workspace_bytes(int dtype, int B, int S, int D, int Hid, int L, int H) {
    if (B <= 0 || S <= 0 || D <= 0 || Hid <= 0 || L <= 0 || H <= 0) return 0;
    const size_t es = dtype == HYB_F32 ? 4 : 2;
    const size_t M = (size_t)B * S;
    const size_t big = (size_t)(Hid > D ? Hid : D);
    return 8 * align256(M * D * es) + 2 * align256(M * big * es);
}

extern "C" int hyb_encoder_fwd(int dtype, const void* x, const float* mask, const float* const* params, void* out, void* saved, int B, int S,
                               int D, int Hid, int L, int H, float attn_p, float layer_p, unsigned long long seed, void* stream) {
    HYB_CHECK_ARG(x && params && out && saved && B > 0 && S > 0 && D > 0 && Hid > 0 && L > 0 && H > 0 && D % H == 0);
    HYB_CHECK_ARG(dtype == HYB_F32 || dtype == HYB_BF16);
    HYB_CHECK_ARG(D % 8 == 0 && Hid % 8 == 0);
    const size_t es = dtype == HYB_F32 ? 4 : 2;
    const int M = B * S;
    const EncLayout lay = enc_layout(dtype, B, S, D, Hid, H);
    hipStream_t st = (hipStream_t)stream;
    char* sv = (char*)saved;
    HYB_HIP_TRY(hipMemcpyAsync(sv + lay.x_in, x, (size_t)M * D * es, hipMemcpyDeviceToDevice, st));
    for (int i = 0; i < L; ++i) {
        char* base = sv + (size_t)i * lay.layer_bytes;
        const float* const* P = params + (size_t)i * 14;
        void* x_in = base + lay.x_in;
        void* y_out = (i == L - 1) ? out : (void*)(base + lay.layer_bytes + lay.x_in);
        const void* xs[3] = {x_in, x_in, x_in};
        const float* Ws[3] = {P[0], P[2], P[4]};
        const float* bs[3] = {P[1], P[3], P[5]};
        void* ys[3] = {base + lay.q, base + lay.k, base + lay.v};
        HYB_TRY(hyb_linear_fwd_grouped3(dtype, xs, Ws, bs, ys, 3, M, D, D, 1, st));                                   // src L69-70
        HYB_TRY(hyb_attention_fwd(dtype, base + lay.q, base + lay.k, base + lay.v, mask, base + lay.attn, (float*)(base + lay.probs), B, S, D,
                                  H, attn_p, attn_seed(seed, i), stream));                                            // src L73-84
        HYB_TRY(hyb_linear_fwd(dtype, base + lay.attn, D, P[6], P[7], base + lay.o, M, D, D, 0, stream));             // src L87
        HYB_TRY(hyb_ln_residual_fwd(dtype, base + lay.o, x_in, P[12], P[13], base + lay.x1, (float*)(base + lay.st1), M, D, 1e-5f, 1.0f, 0.f,
                                    0ull, stream));                                                                   // src L116-117
        HYB_TRY(hyb_linear_fwd(dtype, base + lay.x1, D, P[8], P[9], base + lay.hmid, M, Hid, D, 1, stream));          // src L119 (Linear, ReLU)
        HYB_TRY(hyb_linear_fwd(dtype, base + lay.hmid, Hid, P[10], P[11], base + lay.f, M, D, Hid, 0, stream));       // src L119 (Linear)
        HYB_TRY(hyb_ln_residual_fwd(dtype, base + lay.f, base + lay.x1, P[12], P[13], y_out, (float*)(base + lay.st2), M, D, 1e-5f,
                                    (float)sqrt(0.5), layer_p, drop_seed(seed, i), stream));                          // src L120-123
    }
    return 0;
}

extern "C" int hyb_encoder_bwd(int dtype, const void* dout, const float* mask, const float* const* params, float* const* grads,
                               const void* saved, void* dx, int B, int S, int D, int Hid, int L, int H, float attn_p, float layer_p,
                               unsigned long long seed, void* workspace, size_t workspace_bytes, void* stream) {
    (void)mask;
    HYB_CHECK_ARG(dout && params && grads && saved && dx && workspace && B > 0 && S > 0 && D > 0 && Hid > 0 && L > 0 && H > 0 && D % H == 0);
    HYB_CHECK_ARG(dtype == HYB_F32 || dtype == HYB_BF16);
    if (workspace_bytes < hyb_encoder_workspace_bytes(dtype, B, S, D, Hid, L, H)) return HYB_E_WORKSPACE;
    const size_t es = dtype == HYB_F32 ? 4 : 2;
    const int M = B * S;
    const EncLayout lay = enc_layout(dtype, B, S, D, Hid, H);
    hipStream_t st = (hipStream_t)stream;
    const char* sv = (const char*)saved;
    char* ws = (char*)workspace;
    const size_t md = align256((size_t)M * D * es);
    void* g1 = ws;            // d(LN input)
    void* g2 = ws + md;       // d(x1)
    void* g4 = ws + 2 * md;   // d(attn)
    void* g5 = ws + 3 * md;   // dq
    void* g6 = ws + 4 * md;   // dk
    void* g7 = ws + 5 * md;   // dv
    void* gin[2] = {ws + 6 * md, ws + 7 * md};
    const size_t big = align256((size_t)M * (Hid > D ? Hid : D) * es);
    void* dh = ws + 8 * md;             // d(hmid)
    void* scratch = ws + 8 * md + big;  // relu-masked dy
    const size_t scratch_bytes = big;

    const void* gA = dout;
    for (int i = L - 1; i >= 0; --i) {
        const char* base = sv + (size_t)i * lay.layer_bytes;
        const float* const* P = params + (size_t)i * 14;
        float* const* G = grads + (size_t)i * 14;
        void* gx = (i == 0) ? dx : gin[i & 1];
        HYB_HIP_TRY(hipMemsetAsync(G[12], 0, (size_t)D * 4, st));
        HYB_HIP_TRY(hipMemsetAsync(G[13], 0, (size_t)D * 4, st));
        // LN2 + residual + sqrt(.5) + dropout
        HYB_TRY(hyb_ln_residual_bwd(dtype, gA, base + lay.f, P[12], (const float*)(base + lay.st2), g1, g2, 0, G[12], G[13], M, D,
                                    (float)sqrt(0.5), layer_p, drop_seed(seed, i), stream));
        // FFN
        HYB_TRY(hyb_linear_bwd(dtype, base + lay.hmid, Hid, P[10], nullptr, g1, dh, 0, G[10], G[11], M, D, Hid, 0, nullptr, 0, stream));
        HYB_TRY(hyb_linear_bwd(dtype, base + lay.x1, D, P[8], base + lay.hmid, dh, g2, 1, G[8], G[9], M, Hid, D, 1, scratch, scratch_bytes, stream));
        // LN1 + residual
        HYB_TRY(hyb_ln_residual_bwd(dtype, g2, base + lay.o, P[12], (const float*)(base + lay.st1), g1, gx, 0, G[12], G[13], M, D, 1.0f, 0.f,
                                    0ull, stream));
        // output projection
        HYB_TRY(hyb_linear_bwd(dtype, base + lay.attn, D, P[6], nullptr, g1, g4, 0, G[6], G[7], M, D, D, 0, nullptr, 0, stream));
        // attention core
        HYB_TRY(hyb_attention_bwd(dtype, base + lay.q, base + lay.k, base + lay.v, (const float*)(base + lay.probs), g4, g5, g6, g7, B, S, D, H,
                                  attn_p, attn_seed(seed, i), stream));
        // Q, K, V projections (ReLU), all three feed from the layer input
        HYB_TRY(hyb_linear_bwd(dtype, base + lay.x_in, D, P[0], base + lay.q, g5, gx, 1, G[0], G[1], M, D, D, 1, scratch, scratch_bytes, stream));
        HYB_TRY(hyb_linear_bwd(dtype, base + lay.x_in, D, P[2], base + lay.k, g6, gx, 1, G[2], G[3], M, D, D, 1, scratch, scratch_bytes, stream));
        HYB_TRY(hyb_linear_bwd(dtype, base + lay.x_in, D, P[4], base + lay.v, g7, gx, 1, G[4], G[5], M, D, D, 1, scratch, scratch_bytes, stream));
        gA = gx;
    }
    return 0;
}
