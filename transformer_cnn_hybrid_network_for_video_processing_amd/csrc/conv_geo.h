// Geometry of a general nn.Conv2d (square kernel k, stride, zero padding, dilation) for the exact-fp32 im2col convolutions shared by
// the FCT drop-in (3x3 "same", dilated: FCT.py:110-113,140-143) and the ResNet-bottleneck backbone (SURVEY.md section 8f-3:
// 7x7 stride 2, 3x3 stride 1/2, 1x1 stride 1/2; AE_256_32K bytecode src L23-31, L60-94, L96-106).  NHWC fp32, true channel counts.
// Column order of the patch matrix: kcol = (ky * k + kx) * Ci + ci, zero padded to Kp (a multiple of 8).
#pragma once
#include "hyb_common.h"

namespace {

struct ConvGeo { int H, W, Ci, Ho, Wo, k, stride, pad, dil, Kp; };

inline int conv_out_size(int in, int k, int stride, int pad, int dil) { return (in + 2 * pad - dil * (k - 1) - 1) / stride + 1; }
inline bool conv_geo_make(ConvGeo& g, int H, int W, int Ci, int k, int stride, int pad, int dil) {
    if (H < 1 || W < 1 || Ci < 1 || k < 1 || k > 7 || stride < 1 || stride > 4 || pad < 0 || pad > 24 || dil < 1 || dil > 8) return false;
    if (H + 2 * pad < dil * (k - 1) + 1 || W + 2 * pad < dil * (k - 1) + 1) return false;
    g.H = H; g.W = W; g.Ci = Ci; g.k = k; g.stride = stride; g.pad = pad; g.dil = dil;
    g.Ho = conv_out_size(H, k, stride, pad, dil); g.Wo = conv_out_size(W, k, stride, pad, dil);
    g.Kp = (k * k * Ci + 7) / 8 * 8;
    return true;
}
// a 1x1 stride-1 convolution over a channel count that needs no padding: the patch matrix IS the input
inline bool conv_geo_identity(const ConvGeo& g) { return g.k == 1 && g.stride == 1 && g.pad == 0 && g.Ci % 8 == 0; }

// col [Po][Kp] <- x [n][H][W][Ci]: V consecutive columns per thread (V = 4 needs Ci % 4 == 0 so that a quad stays inside one tap)
template <int V>
__global__ __launch_bounds__(256) void im2col_g_kernel(const float* __restrict__ x, float* __restrict__ col, long long Po, ConvGeo g) {
    const int KV = g.Kp / V;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Po * KV) return;
    const long long pix = i / KV;
    const int kc = (int)(i - pix * KV) * V;
    float v[V];
#pragma unroll
    for (int j = 0; j < V; ++j) v[j] = 0.f;
    if (kc < g.k * g.k * g.Ci) {
        const int tap = kc / g.Ci, ci = kc - tap * g.Ci;
        const int ky = tap / g.k, kx = tap - ky * g.k;
        const int wo = (int)(pix % g.Wo);
        const long long t = pix / g.Wo;
        const int ho = (int)(t % g.Ho);
        const long long n = t / g.Ho;
        const int hh = ho * g.stride - g.pad + ky * g.dil, ww = wo * g.stride - g.pad + kx * g.dil;
        if (hh >= 0 && hh < g.H && ww >= 0 && ww < g.W) {
            const float* src = x + ((n * g.H + hh) * g.W + ww) * g.Ci + ci;
            if (V == 4) { const float4 q = *(const float4*)src; v[0] = q.x; v[1 % V] = q.y; v[2 % V] = q.z; v[3 % V] = q.w; }
            else v[0] = *src;
        }
    }
    if (V == 4) *(float4*)(col + i * 4) = make_float4(v[0], v[1 % V], v[2 % V], v[3 % V]);
    else col[i] = v[0];
}

// dx [n][H][W][Ci] <- dcol [Po][Kp]: every input pixel gathers the taps of the output pixels that read it
template <int V>
__global__ __launch_bounds__(256) void col2im_g_kernel(const float* __restrict__ dcol, float* __restrict__ dx, long long Pin, ConvGeo g) {
    const int CV = g.Ci / V;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Pin * CV) return;
    const long long pix = i / CV;
    const int ci = (int)(i - pix * CV) * V;
    const int w0 = (int)(pix % g.W);
    const long long t = pix / g.W;
    const int h0 = (int)(t % g.H);
    const long long n = t / g.H;
    float s[V];
#pragma unroll
    for (int j = 0; j < V; ++j) s[j] = 0.f;
    for (int ky = 0; ky < g.k; ++ky) {
        const int th = h0 + g.pad - ky * g.dil;
        if (th < 0 || th % g.stride != 0) continue;
        const int ho = th / g.stride;
        if (ho >= g.Ho) continue;
        for (int kx = 0; kx < g.k; ++kx) {
            const int tw = w0 + g.pad - kx * g.dil;
            if (tw < 0 || tw % g.stride != 0) continue;
            const int wo = tw / g.stride;
            if (wo >= g.Wo) continue;
            const float* src = dcol + ((n * g.Ho + ho) * g.Wo + wo) * g.Kp + (ky * g.k + kx) * g.Ci + ci;
            if (V == 4) { const float4 q = *(const float4*)src; s[0] += q.x; s[1 % V] += q.y; s[2 % V] += q.z; s[3 % V] += q.w; }
            else s[0] += *src;
        }
    }
    if (V == 4) *(float4*)(dx + i * 4) = make_float4(s[0], s[1 % V], s[2 % V], s[3 % V]);
    else dx[i] = s[0];
}

// w [Co][Ci][k][k] -> wp [Co][Kp] (row-major, zero padded)
__global__ void conv_pack_g_kernel(const float* __restrict__ w, float* __restrict__ wp, int Co, int Ci, int kk, int Kp) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Co * Kp) return;
    const int co = i / Kp, kc = i - co * Kp;
    float v = 0.f;
    if (kc < kk * Ci) { const int tap = kc / Ci, ci = kc - tap * Ci; v = w[((long long)co * Ci + ci) * kk + tap]; }
    wp[i] = v;
}

inline void launch_im2col(const float* x, float* col, long long Po, const ConvGeo& g, hipStream_t st) {
    if (g.Ci % 4 == 0) hipLaunchKernelGGL(im2col_g_kernel<4>, dim3(hyb_cdiv(Po * (g.Kp / 4), 256)), dim3(256), 0, st, x, col, Po, g);
    else hipLaunchKernelGGL(im2col_g_kernel<1>, dim3(hyb_cdiv(Po * g.Kp, 256)), dim3(256), 0, st, x, col, Po, g);
}
inline void launch_col2im(const float* dcol, float* dx, long long Pin, const ConvGeo& g, hipStream_t st) {
    if (g.Ci % 4 == 0) hipLaunchKernelGGL(col2im_g_kernel<4>, dim3(hyb_cdiv(Pin * (g.Ci / 4), 256)), dim3(256), 0, st, dcol, dx, Pin, g);
    else hipLaunchKernelGGL(col2im_g_kernel<1>, dim3(hyb_cdiv(Pin * g.Ci, 256)), dim3(256), 0, st, dcol, dx, Pin, g);
}

}  // namespace
