// FCT, the reference's "Fully Convolutional Transformer" (FCT.py:24-254) -- SURVEY.md section 8f-1, the first "next" row.
// FORWARD kernels (inference path; the backward is the next step of this row and fails loudly on the Python side until then).
//
// Layout: NHWC fp32 with the true channel count (8 .. 128 channels: LayerNorm-over-C, the pixel-token view of the spatial
// attention and the im2col rows are all contiguous that way).  Arithmetic: exact fp32 (v_mfma_f32_16x16x4_f32 for the
// contractions), which is what the 1e-3 gate against the reference needs.
//
//   conv3x3 (+bias, dilation 1/2/3, ReLU / GELU / sigmoid)   FCT.py:140-145, 110-113, 194-196    im2col -> MFMA GEMM -> activation
//   depthwise 3x3 + bias + ReLU -> LayerNorm over C, x3       FCT.py:41-57 (q, k, v projections)   one fused pass
//   LayerNorm over C                                           FCT.py:97-99
//   nn.MultiheadAttention over H*W pixel tokens, 2 heads       FCT.py:37,67-79                      GEMM in-proj, flash core, GEMM out-proj
//   MaxPool2d(2), AvgPool2d(2,2), Upsample(x2 nearest), cat, add   FCT.py:147,170,180,222,238-240
//   DiceLoss                                                   Metrics.py:5-22
#include <math.h>
#include <stdlib.h>
#include "hyb_common.h"
#include "conv_geo.h"

int hyb_gemm_nt(int dtype, int groups, const void* const* A, const void* const* B, void* const* C, const float* const* bias, int out_f32,
                int Mo, int No, int R, int lda, int ldb, int ldc, int relu, int accumulate, hipStream_t st, const void* const* Amask = nullptr,
                const void* const* Cmask = nullptr);
bool hyb_conv_implicit_ok(int Ci, long long rows);
int hyb_conv_implicit_gemm(const float* x, const float* wp, const float* bias, float* y, int n_img, int H, int W, int Ci, int Ho, int Wo, int Co,
                           int Kp, int k, int stride, int pad, int dil, int ldy, int relu, hipStream_t st);
int hyb_flash_attention_fwd(int dtype, const void* q, const void* k, const void* v, void* out, float* lse, int N, int L, int H, int dhp, int ld,
                            float scale, hipStream_t st, int dh_true);

namespace {

inline size_t al256(size_t v) { return (v + 255) & ~(size_t)255; }
inline int up8(int v) { return (v + 7) / 8 * 8; }
#define FCT_TRY(call) do { int rc_ = (call); if (rc_ != 0) return rc_; } while (0)

__device__ __forceinline__ float gelu_erf(float z) { return 0.5f * z * (1.f + erff(z * 0.70710678118654752f)); }      // nn.GELU() default (erf form)

__global__ void act_kernel(const float* __restrict__ zin, float* __restrict__ y, long long n, int act) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float z = zin[i];
    y[i] = act == HYB_ACT_GELU ? gelu_erf(z) : act == HYB_ACT_SIGMOID ? 1.f / (1.f + expf(-z)) : act == HYB_ACT_RELU ? fmaxf(z, 0.f) : z;
}

// ---- the three depthwise projections of Attention._build_projection (FCT.py:41-57), fused:
//      out_j[pix][c] = LayerNorm_j( relu( dwconv3x3_j(x)[pix][c] + b_j[c] ) ) for j = q, k, v.
// LPP lanes share a pixel (LPP = min(C, 64), a power of two), CPL = C / LPP channels per lane; LayerNorm's mean / variance are
// reductions over those lanes.
struct ProjArgs { const float* w[3]; const float* b[3]; const float* g[3]; const float* beta[3]; float* out[3]; };

template <int CPL>
__global__ __launch_bounds__(256) void qkv_proj_kernel(const float* __restrict__ x, ProjArgs a, long long P, int H, int W, int C, int LPP, float eps) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long pix = t / LPP;
    const int sub = (int)(t - pix * LPP);
    const bool live = pix < P;
    const long long pc = live ? pix : P - 1;
    const int w0 = (int)(pc % W), h0 = (int)((pc / W) % H);
    float xin[9][CPL];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int hh = h0 + tap / 3 - 1, ww = w0 + tap % 3 - 1;
        const bool in = hh >= 0 && hh < H && ww >= 0 && ww < W;
#pragma unroll
        for (int j = 0; j < CPL; ++j) xin[tap][j] = in ? x[(pc + (long long)(hh - h0) * W + (ww - w0)) * C + sub + j * LPP] : 0.f;
    }
#pragma unroll
    for (int pj = 0; pj < 3; ++pj) {
        float r[CPL];
        float s1 = 0.f;
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
            const int c = sub + j * LPP;
            float acc = a.b[pj] ? a.b[pj][c] : 0.f;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) acc = fmaf(a.w[pj][c * 9 + tap], xin[tap][j], acc);
            r[j] = fmaxf(acc, 0.f);
            s1 += r[j];
        }
        for (int o = LPP >> 1; o > 0; o >>= 1) s1 += __shfl_xor(s1, o, 64);
        const float mean = s1 / (float)C;
        float s2 = 0.f;
#pragma unroll
        for (int j = 0; j < CPL; ++j) { const float dlt = r[j] - mean; s2 += dlt * dlt; }
        for (int o = LPP >> 1; o > 0; o >>= 1) s2 += __shfl_xor(s2, o, 64);
        const float rstd = rsqrtf(s2 / (float)C + eps);
        if (live) {
#pragma unroll
            for (int j = 0; j < CPL; ++j) {
                const int c = sub + j * LPP;
                a.out[pj][pix * C + c] = (r[j] - mean) * rstd * a.g[pj][c] + a.beta[pj][c];
            }
        }
    }
}

// LayerNorm over C of NHWC rows (FCT.py:97-99)
template <int CPL>
__global__ __launch_bounds__(256) void ln_c_kernel(const float* __restrict__ x, const float* __restrict__ g, const float* __restrict__ b,
                                                   float* __restrict__ y, long long P, int C, int LPP, float eps) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long pix = t / LPP;
    const int sub = (int)(t - pix * LPP);
    const bool live = pix < P;
    const long long pc = live ? pix : P - 1;
    float r[CPL];
    float s1 = 0.f;
#pragma unroll
    for (int j = 0; j < CPL; ++j) { r[j] = x[pc * C + sub + j * LPP]; s1 += r[j]; }
    for (int o = LPP >> 1; o > 0; o >>= 1) s1 += __shfl_xor(s1, o, 64);
    const float mean = s1 / (float)C;
    float s2 = 0.f;
#pragma unroll
    for (int j = 0; j < CPL; ++j) { const float dlt = r[j] - mean; s2 += dlt * dlt; }
    for (int o = LPP >> 1; o > 0; o >>= 1) s2 += __shfl_xor(s2, o, 64);
    const float rstd = rsqrtf(s2 / (float)C + eps);
    if (live)
#pragma unroll
        for (int j = 0; j < CPL; ++j) { const int c = sub + j * LPP; y[pix * C + c] = (r[j] - mean) * rstd * g[c] + b[c]; }
}

// packed in/out projection weights of nn.MultiheadAttention with every head zero-padded from dh to dhp features:
//   win  [3][Cp][C] <- in_proj_weight [3C][C],  bin [3][Cp] <- in_proj_bias [3C],  wout [C][Cp] <- out_proj.weight [C][C]
__global__ void mha_pack_kernel(const float* __restrict__ in_w, const float* __restrict__ in_b, const float* __restrict__ out_w,
                                float* __restrict__ win, float* __restrict__ bin, float* __restrict__ wout, int C, int Hh, int dh, int dhp) {
    const int Cp = Hh * dhp;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int n_in = 3 * Cp * C;
    if (i < n_in) {
        const int j = i / (Cp * C), rem = i - j * Cp * C, rp = rem / C, c = rem - rp * C;
        const int h = rp / dhp, f = rp - h * dhp;
        win[i] = f < dh ? in_w[((long long)j * C + h * dh + f) * C + c] : 0.f;
    } else if (i < n_in + 3 * Cp) {
        const int k = i - n_in, j = k / Cp, rp = k - j * Cp, h = rp / dhp, f = rp - h * dhp;
        bin[k] = (f < dh && in_b) ? in_b[j * C + h * dh + f] : 0.f;
    } else if (i < n_in + 3 * Cp + C * Cp) {
        const int k = i - n_in - 3 * Cp, r = k / Cp, cp = k - r * Cp, h = cp / dhp, f = cp - h * dhp;
        wout[k] = f < dh ? out_w[(long long)r * C + h * dh + f] : 0.f;
    }
}

// ---- elementwise / resampling (NHWC) ------------------------------------------------------------------------------------
__global__ void add_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = a[i] + b[i];
}
// mode 0: MaxPool2d(2) (floor), 1: AvgPool2d(2,2), 2: Upsample(scale_factor=2, nearest); H, W are the INPUT sizes
__global__ void resample_kernel(const float* __restrict__ x, float* __restrict__ y, long long total, int H, int W, int C, int mode) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % C);
    long long r = i / C;
    if (mode == 2) {
        const int Wo = 2 * W, Ho = 2 * H;
        const int wo = (int)(r % Wo); r /= Wo;
        const int ho = (int)(r % Ho); const long long n = r / Ho;
        y[i] = x[((n * H + ho / 2) * W + wo / 2) * C + c];
    } else {
        const int Wo = W / 2, Ho = H / 2;
        const int wo = (int)(r % Wo); r /= Wo;
        const int ho = (int)(r % Ho); const long long n = r / Ho;
        const float* s = x + ((n * H + 2 * ho) * W + 2 * wo) * C + c;
        const float v0 = s[0], v1 = s[C], v2 = s[(long long)W * C], v3 = s[(long long)W * C + C];
        y[i] = mode == 0 ? fmaxf(fmaxf(v0, v1), fmaxf(v2, v3)) : 0.25f * ((v0 + v1) + (v2 + v3));
    }
}
// y[pix][0..Ca) = a[pix], y[pix][Ca..Ca+Cb) = b[pix]   (torch.cat along channels)
__global__ void concat_kernel(const float* __restrict__ a, int Ca, const float* __restrict__ b, int Cb, float* __restrict__ y, long long P) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int C = Ca + Cb;
    if (i >= P * C) return;
    const long long pix = i / C;
    const int c = (int)(i - pix * C);
    y[i] = c < Ca ? a[pix * Ca + c] : b[pix * Cb + c - Ca];
}

// ---- Dice loss (Metrics.py:5-22): channel 0 of NCHW pred / true; partial sums per block, fixed-order finish ------------------
__global__ __launch_bounds__(256) void dice_partial_kernel(const float* __restrict__ pred, const float* __restrict__ tru, float* __restrict__ part,
                                                           int N, int C, long long HW) {
    __shared__ float red[3][256];
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    const long long total = (long long)N * HW;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long n = i / HW, r = i - n * HW;
        const float p = pred[(n * C) * HW + r], t = tru[(n * C) * HW + r];
        s0 += p * t; s1 += p; s2 += t;
    }
    red[0][threadIdx.x] = s0; red[1][threadIdx.x] = s1; red[2][threadIdx.x] = s2;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o)
#pragma unroll
            for (int j = 0; j < 3; ++j) red[j][threadIdx.x] += red[j][threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x < 3) part[blockIdx.x * 3 + threadIdx.x] = red[threadIdx.x][0];
}
__global__ void dice_finish_kernel(const float* __restrict__ part, int blocks, float smooth, float* __restrict__ loss) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double s[3] = {0.0, 0.0, 0.0};
    for (int b = 0; b < blocks; ++b)
        for (int j = 0; j < 3; ++j) s[j] += (double)part[b * 3 + j];
    loss[0] = (float)(1.0 - (2.0 * s[0] + smooth) / (s[1] + s[2] + smooth));
}

inline int grid1(long long n) { return hyb_cdiv(n, 256); }
constexpr long long CONV_CHUNK_BYTES = 512ll << 20;     // im2col rows are produced in image chunks of at most this size

}  // namespace

// ---- general Conv2d forward (conv_geo.h): patch matrix in image chunks + exact-fp32 MFMA GEMM with bias / ReLU epilogue -------------
extern "C" size_t hyb_conv2d_workspace(int N, int H, int W, int Ci, int Co, int k, int stride, int pad, int dilation) {
    ConvGeo g;
    if (N < 1 || Co < 1 || !conv_geo_make(g, H, W, Ci, k, stride, pad, dilation)) return 0;
    const long long per_img = (long long)g.Ho * g.Wo * g.Kp * 4;
    long long nb = CONV_CHUNK_BYTES / per_img; if (nb < 1) nb = 1; if (nb > N) nb = N;
    return al256((size_t)Co * g.Kp * 4) + (conv_geo_identity(g) ? 0 : al256((size_t)nb * per_img));
}

extern "C" int hyb_conv2d_fwd(const float* x, const float* w, const float* b, float* y, float* z_out, int N, int H, int W, int Ci, int Co, int k,
                              int stride, int pad, int dilation, int act, void* workspace, size_t workspace_bytes, void* stream) {
    ConvGeo g;
    HYB_CHECK_ARG(x && w && y && workspace && N > 0 && Co > 0 && conv_geo_make(g, H, W, Ci, k, stride, pad, dilation));
    HYB_CHECK_ARG(act >= HYB_ACT_NONE && act <= HYB_ACT_SIGMOID);
    if (workspace_bytes < hyb_conv2d_workspace(N, H, W, Ci, Co, k, stride, pad, dilation)) return HYB_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int Kp = g.Kp;
    const bool ident = conv_geo_identity(g);
    float* wp = (float*)workspace;
    float* col = (float*)((char*)workspace + al256((size_t)Co * Kp * 4));
    hipLaunchKernelGGL(conv_pack_g_kernel, dim3(grid1((long long)Co * Kp)), dim3(256), 0, st, w, wp, Co, Ci, k * k, Kp);
    HYB_LAUNCH_CHECK();
    static const int implicit_env = getenv("HYB_CONV_IMPLICIT") ? atoi(getenv("HYB_CONV_IMPLICIT")) : 1;
    if (implicit_env && !ident && hyb_conv_implicit_ok(Ci, (long long)N * g.Ho * g.Wo) && (long long)N * g.Ho * g.Wo <= 0x7fffffff / 32 * 32 &&
        (long long)N * H * W <= 0x7fffffff / 32 * 32) {
        // no patch matrix: the GEMM gathers its A fragments from the image (all images in one launch)
        float* gemm_out = ((act == HYB_ACT_GELU || act == HYB_ACT_SIGMOID) && z_out) ? z_out : y;
        FCT_TRY(hyb_conv_implicit_gemm(x, wp, b, gemm_out, N, H, W, Ci, g.Ho, g.Wo, Co, Kp, k, stride, pad, dilation, Co, act == HYB_ACT_RELU, st));
        if (act == HYB_ACT_GELU || act == HYB_ACT_SIGMOID) {
            const long long n = (long long)N * g.Ho * g.Wo * Co;
            hipLaunchKernelGGL(act_kernel, dim3(grid1(n)), dim3(256), 0, st, z_out ? (const float*)z_out : (const float*)y, y, n, act);
            HYB_LAUNCH_CHECK();
        }
        return 0;
    }
    const long long per_img = (long long)g.Ho * g.Wo * Kp * 4;
    long long nb = CONV_CHUNK_BYTES / per_img; if (nb < 1) nb = 1; if (nb > N) nb = N;
    for (int n0 = 0; n0 < N; n0 += (int)nb) {
        const int nn = N - n0 < nb ? N - n0 : (int)nb;
        const long long P = (long long)nn * g.Ho * g.Wo, off = (long long)n0 * g.Ho * g.Wo;
        if (P > 0x7fffffff / 32 * 32 || (long long)nn * H * W > 0x7fffffff / 32 * 32) return HYB_E_ARG;
        const float* xin = x + (long long)n0 * H * W * Ci;
        if (!ident) { launch_im2col(xin, col, P, g, st); HYB_LAUNCH_CHECK(); }
        // GELU / sigmoid: the GEMM leaves the pre-activation (in z_out when the caller keeps it for backward, else in y)
        float* gemm_out = ((act == HYB_ACT_GELU || act == HYB_ACT_SIGMOID) && z_out) ? z_out : y;
        const void* A[1] = {ident ? xin : col}; const void* B[1] = {wp}; void* Cc[1] = {gemm_out + off * Co}; const float* bias[1] = {b};
        FCT_TRY(hyb_gemm_nt(HYB_F32, 1, A, B, Cc, bias, 0, (int)P, Co, Kp, Kp, Kp, Co, act == HYB_ACT_RELU, 0, st));
    }
    if (act == HYB_ACT_GELU || act == HYB_ACT_SIGMOID) {
        const long long n = (long long)N * g.Ho * g.Wo * Co;
        hipLaunchKernelGGL(act_kernel, dim3(grid1(n)), dim3(256), 0, st, z_out ? (const float*)z_out : (const float*)y, y, n, act);
        HYB_LAUNCH_CHECK();
    }
    return 0;
}

// FCT's nn.Conv2d(.., 3, 1, padding="same"[, dilation=d]) is the general convolution at k = 3, stride 1, pad = d
extern "C" size_t hyb_fct_conv_workspace(int N, int H, int W, int Ci, int Co) { return hyb_conv2d_workspace(N, H, W, Ci, Co, 3, 1, 1, 1); }
extern "C" int hyb_fct_conv_fwd(const float* x, const float* w, const float* b, float* y, float* z_out, int N, int H, int W, int Ci, int Co,
                                int dilation, int act, void* workspace, size_t workspace_bytes, void* stream) {
    HYB_CHECK_ARG(dilation >= 1 && dilation <= 8);
    return hyb_conv2d_fwd(x, w, b, y, z_out, N, H, W, Ci, Co, 3, 1, dilation, dilation, act, workspace, workspace_bytes, stream);
}

template <typename F>
static int dispatch_cpl(int C, int& LPP, F&& f) {
    if (C < 1 || C > 512 || (C & (C - 1)) != 0) return HYB_E_ARG;          // a power of two (the reference uses 8 .. 128)
    LPP = C < 64 ? C : 64;
    return f(C / LPP);
}

extern "C" int hyb_fct_qkv_proj_fwd(const float* x, const float* const* w3, const float* const* b3, const float* const* g3, const float* const* beta3,
                                    float* q, float* k, float* v, int N, int H, int W, int C, float eps, void* stream) {
    HYB_CHECK_ARG(x && w3 && b3 && g3 && beta3 && q && k && v && N > 0 && H > 0 && W > 0);
    ProjArgs a{};
    float* outs[3] = {q, k, v};
    for (int j = 0; j < 3; ++j) { HYB_CHECK_ARG(w3[j] && g3[j] && beta3[j]); a.w[j] = w3[j]; a.b[j] = b3[j]; a.g[j] = g3[j]; a.beta[j] = beta3[j]; a.out[j] = outs[j]; }
    const long long P = (long long)N * H * W;
    hipStream_t st = (hipStream_t)stream;
    int LPP = 0;
    const int rc = dispatch_cpl(C, LPP, [&](int cpl) {
        const dim3 grid(grid1(P * LPP));
        switch (cpl) {
            case 1: hipLaunchKernelGGL(qkv_proj_kernel<1>, grid, dim3(256), 0, st, x, a, P, H, W, C, LPP, eps); break;
            case 2: hipLaunchKernelGGL(qkv_proj_kernel<2>, grid, dim3(256), 0, st, x, a, P, H, W, C, LPP, eps); break;
            case 4: hipLaunchKernelGGL(qkv_proj_kernel<4>, grid, dim3(256), 0, st, x, a, P, H, W, C, LPP, eps); break;
            case 8: hipLaunchKernelGGL(qkv_proj_kernel<8>, grid, dim3(256), 0, st, x, a, P, H, W, C, LPP, eps); break;
            default: return HYB_E_ARG;
        }
        return 0;
    });
    if (rc) return rc;
    HYB_LAUNCH_CHECK();
    return 0;
}

extern "C" int hyb_fct_ln_fwd(const float* x, const float* g, const float* b, float* y, long long P, int C, float eps, void* stream) {
    HYB_CHECK_ARG(x && g && b && y && P > 0);
    hipStream_t st = (hipStream_t)stream;
    int LPP = 0;
    const int rc = dispatch_cpl(C, LPP, [&](int cpl) {
        const dim3 grid(grid1(P * LPP));
        switch (cpl) {
            case 1: hipLaunchKernelGGL(ln_c_kernel<1>, grid, dim3(256), 0, st, x, g, b, y, P, C, LPP, eps); break;
            case 2: hipLaunchKernelGGL(ln_c_kernel<2>, grid, dim3(256), 0, st, x, g, b, y, P, C, LPP, eps); break;
            case 4: hipLaunchKernelGGL(ln_c_kernel<4>, grid, dim3(256), 0, st, x, g, b, y, P, C, LPP, eps); break;
            case 8: hipLaunchKernelGGL(ln_c_kernel<8>, grid, dim3(256), 0, st, x, g, b, y, P, C, LPP, eps); break;
            default: return HYB_E_ARG;
        }
        return 0;
    });
    if (rc) return rc;
    HYB_LAUNCH_CHECK();
    return 0;
}

extern "C" size_t hyb_fct_mha_workspace(int N, int L, int C, int heads) {
    if (N < 1 || L < 1 || C < 1 || heads < 1 || C % heads != 0) return 0;
    const int dhp = up8(C / heads), Cp = heads * dhp;
    const size_t M = (size_t)N * L;
    return al256((size_t)3 * Cp * C * 4) + al256((size_t)3 * Cp * 4) + al256((size_t)C * Cp * 4) + 4 * al256(M * Cp * 4);
}

extern "C" size_t hyb_fct_mha_saved_bytes(int N, int L, int C, int heads) {
    if (N < 1 || L < 1 || C < 1 || heads < 1 || C % heads != 0) return 0;
    const int Cp = heads * up8(C / heads);
    return 4 * al256((size_t)N * L * Cp * 4) + al256((size_t)N * heads * L * 4);
}

extern "C" int hyb_fct_mha_fwd(const float* q, const float* k, const float* v, const float* in_w, const float* in_b, const float* out_w,
                               const float* out_b, float* out, void* saved, int N, int L, int C, int heads, void* workspace,
                               size_t workspace_bytes, void* stream) {
    HYB_CHECK_ARG(q && k && v && in_w && out_w && out && workspace && N > 0 && L > 0 && C > 0 && heads > 0 && C % heads == 0 && C % 8 == 0);
    if (workspace_bytes < hyb_fct_mha_workspace(N, L, C, heads)) return HYB_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int dh = C / heads, dhp = up8(dh), Cp = heads * dhp;
    const long long M = (long long)N * L;
    HYB_CHECK_ARG(M <= 0x7fffffff / 32 * 32 && dhp <= 128);
    char* ws = (char*)workspace;
    float* win = (float*)ws;   ws += al256((size_t)3 * Cp * C * 4);
    float* bin = (float*)ws;   ws += al256((size_t)3 * Cp * 4);
    float* wout = (float*)ws;  ws += al256((size_t)C * Cp * 4);
    const size_t act = al256((size_t)M * Cp * 4);
    if (saved) ws = (char*)saved;                              // kept for backward: projected q, k, v, the attention output and the row log-sum-exps
    float* Q = (float*)ws; float* K = (float*)(ws + act); float* V = (float*)(ws + 2 * act); float* A = (float*)(ws + 3 * act);
    float* lse = saved ? (float*)(ws + 4 * act) : nullptr;
    const int total = 3 * Cp * C + 3 * Cp + C * Cp;
    hipLaunchKernelGGL(mha_pack_kernel, dim3(grid1(total)), dim3(256), 0, st, in_w, in_b, out_w, win, bin, wout, C, heads, dh, dhp);
    HYB_LAUNCH_CHECK();
    {   // in-projection: three GEMMs in one launch (torch's _in_projection_packed)
        const void* As[3] = {q, k, v}; const void* Bs[3] = {win, win + (size_t)Cp * C, win + (size_t)2 * Cp * C};
        void* Cs[3] = {Q, K, V}; const float* bs[3] = {bin, bin + Cp, bin + 2 * Cp};
        FCT_TRY(hyb_gemm_nt(HYB_F32, 3, As, Bs, Cs, bs, 0, (int)M, Cp, C, C, C, Cp, 0, 0, st));
    }
    FCT_TRY(hyb_flash_attention_fwd(HYB_F32, Q, K, V, A, lse, N, L, heads, dhp, Cp, 1.0f / sqrtf((float)dh), st, dh));
    {   // out-projection
        const void* As[1] = {A}; const void* Bs[1] = {wout}; void* Cs[1] = {out}; const float* bs[1] = {out_b};
        FCT_TRY(hyb_gemm_nt(HYB_F32, 1, As, Bs, Cs, bs, 0, (int)M, C, Cp, Cp, Cp, C, 0, 0, st));
    }
    return 0;
}

extern "C" int hyb_fct_add(const float* a, const float* b, float* y, long long n, void* stream) {
    HYB_CHECK_ARG(a && b && y && n > 0);
    hipLaunchKernelGGL(add_kernel, dim3(grid1(n)), dim3(256), 0, (hipStream_t)stream, a, b, y, n);
    HYB_LAUNCH_CHECK();
    return 0;
}

extern "C" int hyb_fct_resample(int mode, const float* x, float* y, int N, int H, int W, int C, void* stream) {
    HYB_CHECK_ARG(x && y && N > 0 && H > 0 && W > 0 && C > 0 && mode >= 0 && mode <= 2 && (mode == 2 || (H >= 2 && W >= 2)));
    const long long total = mode == 2 ? (long long)N * 4 * H * W * C : (long long)N * (H / 2) * (W / 2) * C;
    hipLaunchKernelGGL(resample_kernel, dim3(grid1(total)), dim3(256), 0, (hipStream_t)stream, x, y, total, H, W, C, mode);
    HYB_LAUNCH_CHECK();
    return 0;
}

extern "C" int hyb_fct_concat(const float* a, int Ca, const float* b, int Cb, float* y, long long P, void* stream) {
    HYB_CHECK_ARG(a && b && y && Ca > 0 && Cb > 0 && P > 0);
    hipLaunchKernelGGL(concat_kernel, dim3(grid1(P * (Ca + Cb))), dim3(256), 0, (hipStream_t)stream, a, Ca, b, Cb, y, P);
    HYB_LAUNCH_CHECK();
    return 0;
}

extern "C" size_t hyb_dice_workspace(void) { return 256 * 3 * sizeof(float); }

extern "C" int hyb_dice_fwd(const float* pred, const float* tru, float* loss, int N, int C, long long HW, float smooth, void* workspace,
                            size_t workspace_bytes, void* stream) {
    HYB_CHECK_ARG(pred && tru && loss && workspace && N > 0 && C > 0 && HW > 0);
    if (workspace_bytes < hyb_dice_workspace()) return HYB_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    int blocks = hyb_cdiv((long long)N * HW, 256 * 16);
    if (blocks > 256) blocks = 256;
    hipLaunchKernelGGL(dice_partial_kernel, dim3(blocks), dim3(256), 0, st, pred, tru, (float*)workspace, N, C, HW);
    hipLaunchKernelGGL(dice_finish_kernel, dim3(1), dim3(64), 0, st, (const float*)workspace, blocks, smooth, loss);
    HYB_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------------
// Clip input pipeline: torchvision ToTensor on the GPU.  The reference decodes frames on the host and turns
// each into a float CHW tensor in [0,1] (transforms.ToTensor, Dataloader.py:19-23; dataset.pyc src L106-113 for clips); moving the
// frames over PCIe as uint8 HWC (1/4 of the bytes) and converting here gives the same values: out[f][c][h][w] = src[f][h][w][c] / 255.
// ---------------------------------------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void u8hwc_to_f32chw_kernel(const unsigned char* __restrict__ src, float* __restrict__ dst, long long frames,
                                                              int HW, int C) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;        // one pixel of one frame
    if (i >= frames * HW) return;
    const long long f = i / HW;
    const int pix = (int)(i - f * HW);
    const unsigned char* s = src + i * C;
    for (int c = 0; c < C; ++c) dst[(f * C + c) * HW + pix] = (float)s[c] / 255.0f;      // ToTensor divides by 255 (not a multiply by 1/255)
}
}  // namespace

extern "C" int hyb_frames_u8hwc_to_f32chw(const unsigned char* src, float* dst, long long frames, int H, int W, int C, void* stream) {
    HYB_CHECK_ARG(src && dst && frames > 0 && H > 0 && W > 0 && C > 0 && C <= 4);
    const long long n = frames * H * W;
    hipLaunchKernelGGL(u8hwc_to_f32chw_kernel, dim3(hyb_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, src, dst, frames, H * W, C);
    HYB_LAUNCH_CHECK();
    return 0;
}
