// nn.BatchNorm2d (+ residual add + ReLU) and nn.Dropout2d for the ResNet-bottleneck backbone (AE_256_32K
// bytecode: Bottleneck.forward src L35-53 `out = relu(bn3(conv3(..)) + residual)`, Encoder_32K.forward src L108-137).
// NHWC fp32 with the true channel count C (C % 4 == 0 and (C/4) | 256: 8, 16, .., 1024).  All of it is HBM-bound streaming:
//   forward   pass 1: per-channel sum / sum of squares (double accumulators, per-workgroup partial rows, fixed-order finalize in double (a wave per channel):
//             mean, 1/sqrt(var_biased + eps), running statistics with the unbiased variance like torch);
//             pass 2: y = relu?( x * a_c + b_c (+ residual) ),  a = gamma * invstd, b = beta - mean * a
//   backward  pass 1: dz = dy * (y > 0), partial sums of dz and dz * xhat;  finalize: dgamma, dbeta and the three per-channel
//             coefficients;  pass 2: dx = a * (dz - mean(dz) - xhat * mean(dz xhat)),  dresidual = dz.
// A thread owns one channel quad for its whole run (the float4 stride is a multiple of C/4), so the partial sums need no index math.
#include <math.h>
#include "hyb_common.h"

namespace {

inline size_t al256(size_t v) { return (v + 255) & ~(size_t)255; }
constexpr int BN_MAX_BLOCKS = 1024;

inline bool bn_shape_ok(long long P, int C) { return P > 0 && C >= 4 && C <= 1024 && C % 4 == 0 && 256 % (C / 4) == 0 && P * (C / 4) < (1ll << 40); }
inline int bn_blocks(long long P, int C) {
    const long long quads = P * (C / 4);
    long long b = (quads + 256 * 16 - 1) / (256 * 16);
    if (b > BN_MAX_BLOCKS) b = BN_MAX_BLOCKS;
    return (int)(b < 1 ? 1 : b);
}

// part [block][2][C] doubles
__global__ __launch_bounds__(256) void bn2d_stats_kernel(const float* __restrict__ x, double* __restrict__ part, long long quads, int C) {
    __shared__ double red[256][8];
    const int CQ = C >> 2;
    const int tid = threadIdx.x;
    double s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + tid; i < quads; i += stride) {
        const float4 v = ((const float4*)x)[i];
        s[0] += v.x; s[1] += v.y; s[2] += v.z; s[3] += v.w;
        q[0] = fma((double)v.x, (double)v.x, q[0]); q[1] = fma((double)v.y, (double)v.y, q[1]);
        q[2] = fma((double)v.z, (double)v.z, q[2]); q[3] = fma((double)v.w, (double)v.w, q[3]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { red[tid][j] = s[j]; red[tid][4 + j] = q[j]; }
    __syncthreads();
    // threads tid, tid + CQ, tid + 2 CQ, .. own the same channel quad
    if (tid < CQ) {
        double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int t = tid; t < 256; t += CQ)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += red[t][j];
        double* out = part + (long long)blockIdx.x * 2 * C;
#pragma unroll
        for (int j = 0; j < 4; ++j) { out[tid * 4 + j] = acc[j]; out[C + tid * 4 + j] = acc[4 + j]; }
    }
}

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// one wave per channel: fixed-order sum of the partial rows (lane l takes rows l, l + 64, ..; butterfly across the lanes);
// coef [4][C]: a, b (apply), mean, invstd (saved for the backward)
__global__ __launch_bounds__(64) void bn2d_finalize_kernel(const double* __restrict__ part, int blocks, long long P, int C,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta, float eps, float momentum,
                                                           int training, float* __restrict__ running_mean, float* __restrict__ running_var,
                                                           float* __restrict__ coef) {
    const int c = blockIdx.x, lane = threadIdx.x;
    double mean, var;
    if (training) {
        double s = 0, q = 0;
        for (int b = lane; b < blocks; b += 64) { s += part[(long long)b * 2 * C + c]; q += part[(long long)b * 2 * C + C + c]; }
        s = wave_sum_f64(s); q = wave_sum_f64(q);
        mean = s / (double)P;
        var = q / (double)P - mean * mean;
        if (var < 0) var = 0;
        if (lane == 0 && running_mean && running_var) {
            const double unbiased = P > 1 ? var * (double)P / (double)(P - 1) : var;
            running_mean[c] = (float)((1.0 - (double)momentum) * (double)running_mean[c] + (double)momentum * mean);
            running_var[c] = (float)((1.0 - (double)momentum) * (double)running_var[c] + (double)momentum * unbiased);
        }
    } else { mean = running_mean[c]; var = running_var[c]; }
    if (lane != 0) return;
    const double invstd = 1.0 / sqrt(var + (double)eps);
    const double a = (double)gamma[c] * invstd;
    coef[c] = (float)a;
    coef[C + c] = (float)((double)beta[c] - mean * a);
    coef[2 * C + c] = (float)mean;
    coef[3 * C + c] = (float)invstd;
}

template <bool RES, bool RELU>
__global__ __launch_bounds__(256) void bn2d_apply_kernel(const float* __restrict__ x, const float* __restrict__ res, const float* __restrict__ coef,
                                                         float* __restrict__ y, long long quads, int C) {
    const int CQ = C >> 2;
    const int cq = threadIdx.x % CQ;                 // (gridDim.x * 256) % CQ == 0: fixed for the thread
    const float4 a = ((const float4*)coef)[cq], b = ((const float4*)(coef + C))[cq];
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < quads; i += stride) {
        const float4 v = ((const float4*)x)[i];
        float4 o = make_float4(fmaf(v.x, a.x, b.x), fmaf(v.y, a.y, b.y), fmaf(v.z, a.z, b.z), fmaf(v.w, a.w, b.w));
        if (RES) { const float4 r = ((const float4*)res)[i]; o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w; }
        if (RELU) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
        ((float4*)y)[i] = o;
    }
}

// backward pass 1: part [block][2][C] doubles = sum dz, sum dz * xhat
// RELU: 0 = none, 1 = mask from the forward output y (needed when a residual was added before the ReLU), 2 = mask recomputed from x
// with the forward's own fmaf(x, a, b) > 0 (bit-identical decision, and one array less to read in both backward passes)
template <int RELU>
__global__ __launch_bounds__(256) void bn2d_bwd_reduce_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ y,
                                                              const float* __restrict__ coef, double* __restrict__ part, long long quads, int C) {
    __shared__ double red[256][8];
    const int CQ = C >> 2;
    const int tid = threadIdx.x, cq = tid % CQ;
    const float4 mean = ((const float4*)(coef + 2 * C))[cq], inv = ((const float4*)(coef + 3 * C))[cq];
    const float4 ca = ((const float4*)coef)[cq], cb = ((const float4*)(coef + C))[cq];
    double s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + tid; i < quads; i += stride) {
        float4 g = ((const float4*)dy)[i];
        const float4 v = ((const float4*)x)[i];
        if (RELU == 1) { const float4 o = ((const float4*)y)[i]; g.x = o.x > 0.f ? g.x : 0.f; g.y = o.y > 0.f ? g.y : 0.f; g.z = o.z > 0.f ? g.z : 0.f; g.w = o.w > 0.f ? g.w : 0.f; }
        if (RELU == 2) {
            g.x = fmaf(v.x, ca.x, cb.x) > 0.f ? g.x : 0.f; g.y = fmaf(v.y, ca.y, cb.y) > 0.f ? g.y : 0.f;
            g.z = fmaf(v.z, ca.z, cb.z) > 0.f ? g.z : 0.f; g.w = fmaf(v.w, ca.w, cb.w) > 0.f ? g.w : 0.f;
        }
        s[0] += g.x; s[1] += g.y; s[2] += g.z; s[3] += g.w;
        q[0] = fma((double)g.x, (double)((v.x - mean.x) * inv.x), q[0]); q[1] = fma((double)g.y, (double)((v.y - mean.y) * inv.y), q[1]);
        q[2] = fma((double)g.z, (double)((v.z - mean.z) * inv.z), q[2]); q[3] = fma((double)g.w, (double)((v.w - mean.w) * inv.w), q[3]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { red[tid][j] = s[j]; red[tid][4 + j] = q[j]; }
    __syncthreads();
    if (tid < CQ) {
        double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int t = tid; t < 256; t += CQ)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += red[t][j];
        double* out = part + (long long)blockIdx.x * 2 * C;
#pragma unroll
        for (int j = 0; j < 4; ++j) { out[tid * 4 + j] = acc[j]; out[C + tid * 4 + j] = acc[4 + j]; }
    }
}

// bcoef [3][C]: k1 = gamma * invstd, k2 = mean(dz) (0 in eval mode), k3 = mean(dz * xhat) (0 in eval mode); one wave per channel
__global__ __launch_bounds__(64) void bn2d_bwd_finalize_kernel(const double* __restrict__ part, int blocks, long long P, int C,
                                                               const float* __restrict__ gamma, const float* __restrict__ coef, int training,
                                                               float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ bcoef) {
    const int c = blockIdx.x, lane = threadIdx.x;
    double s = 0, q = 0;
    for (int b = lane; b < blocks; b += 64) { s += part[(long long)b * 2 * C + c]; q += part[(long long)b * 2 * C + C + c]; }
    s = wave_sum_f64(s); q = wave_sum_f64(q);
    if (lane != 0) return;
    dgamma[c] = (float)q;
    dbeta[c] = (float)s;
    bcoef[c] = gamma[c] * coef[3 * C + c];
    bcoef[C + c] = training ? (float)(s / (double)P) : 0.f;
    bcoef[2 * C + c] = training ? (float)(q / (double)P) : 0.f;
}

template <int RELU, bool RES>
__global__ __launch_bounds__(256) void bn2d_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ y,
                                                             const float* __restrict__ coef, const float* __restrict__ bcoef, float* __restrict__ dx,
                                                             float* __restrict__ dres, long long quads, int C) {
    const int CQ = C >> 2;
    const int cq = threadIdx.x % CQ;
    const float4 mean = ((const float4*)(coef + 2 * C))[cq], inv = ((const float4*)(coef + 3 * C))[cq];
    const float4 k1 = ((const float4*)bcoef)[cq], k2 = ((const float4*)(bcoef + C))[cq], k3 = ((const float4*)(bcoef + 2 * C))[cq];
    const float4 ca = ((const float4*)coef)[cq], cb = ((const float4*)(coef + C))[cq];
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < quads; i += stride) {
        float4 g = ((const float4*)dy)[i];
        const float4 v = ((const float4*)x)[i];
        if (RELU == 1) { const float4 o = ((const float4*)y)[i]; g.x = o.x > 0.f ? g.x : 0.f; g.y = o.y > 0.f ? g.y : 0.f; g.z = o.z > 0.f ? g.z : 0.f; g.w = o.w > 0.f ? g.w : 0.f; }
        if (RELU == 2) {
            g.x = fmaf(v.x, ca.x, cb.x) > 0.f ? g.x : 0.f; g.y = fmaf(v.y, ca.y, cb.y) > 0.f ? g.y : 0.f;
            g.z = fmaf(v.z, ca.z, cb.z) > 0.f ? g.z : 0.f; g.w = fmaf(v.w, ca.w, cb.w) > 0.f ? g.w : 0.f;
        }
        if (RES) ((float4*)dres)[i] = g;
        float4 o;
        o.x = k1.x * (g.x - k2.x - (v.x - mean.x) * inv.x * k3.x);
        o.y = k1.y * (g.y - k2.y - (v.y - mean.y) * inv.y * k3.y);
        o.z = k1.z * (g.z - k2.z - (v.z - mean.z) * inv.z * k3.z);
        o.w = k1.w * (g.w - k2.w - (v.w - mean.w) * inv.w * k3.w);
        ((float4*)dx)[i] = o;
    }
}

// nn.Dropout2d: one keep / drop decision per (image, channel) plane
__global__ __launch_bounds__(256) void dropout2d_kernel(const float* __restrict__ x, float* __restrict__ y, long long quads, long long hw_quads,
                                                        int C, float p, float inv_keep, unsigned long long seed,
                                                        const unsigned long long* __restrict__ seed_inc) {
    const int CQ = C >> 2;
    const unsigned long long s = seed + (seed_inc ? *seed_inc : 0ull);
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < quads; i += stride) {
        const long long n = i / hw_quads;
        const int c0 = (int)(i % CQ) * 4;
        const unsigned long long plane = (unsigned long long)n * C + c0;
        const float4 v = ((const float4*)x)[i];
        ((float4*)y)[i] = make_float4(v.x * dropout_mult(s, plane, p, inv_keep), v.y * dropout_mult(s, plane + 1, p, inv_keep),
                                      v.z * dropout_mult(s, plane + 2, p, inv_keep), v.w * dropout_mult(s, plane + 3, p, inv_keep));
    }
}

}  // namespace

extern "C" size_t hyb_bn2d_workspace(long long P, int C) {
    if (!bn_shape_ok(P, C)) return 0;
    return al256((size_t)bn_blocks(P, C) * 2 * C * 8) + al256((size_t)3 * C * 4);
}

// y = relu?( BatchNorm2d(x) (+ residual) ).  coef [4][C] is written (a, b, mean, invstd) and kept by the caller for the backward.
// training != 0: batch statistics, running_mean / running_var (nullable) updated in place with `momentum` and the unbiased variance;
// training == 0: the running statistics normalise.
extern "C" int hyb_bn2d_fwd(const float* x, const float* gamma, const float* beta, const float* residual, float* y, float* coef,
                            float* running_mean, float* running_var, long long P, int C, float eps, float momentum, int training, int relu,
                            void* workspace, size_t workspace_bytes, void* stream) {
    HYB_CHECK_ARG(x && gamma && beta && y && coef && workspace && bn_shape_ok(P, C) && eps > 0.f);
    HYB_CHECK_ARG(training || (running_mean && running_var));
    if (workspace_bytes < hyb_bn2d_workspace(P, C)) return HYB_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const long long quads = P * (C / 4);
    const int blocks = bn_blocks(P, C);
    double* part = (double*)workspace;
    if (training) { hipLaunchKernelGGL(bn2d_stats_kernel, dim3(blocks), dim3(256), 0, st, x, part, quads, C); HYB_LAUNCH_CHECK(); }
    hipLaunchKernelGGL(bn2d_finalize_kernel, dim3(C), dim3(64), 0, st, (const double*)part, blocks, P, C, gamma, beta, eps, momentum,
                       training, running_mean, running_var, coef);
    HYB_LAUNCH_CHECK();
    const dim3 grid(blocks), blk(256);
    if (residual && relu) hipLaunchKernelGGL((bn2d_apply_kernel<true, true>), grid, blk, 0, st, x, residual, (const float*)coef, y, quads, C);
    else if (residual) hipLaunchKernelGGL((bn2d_apply_kernel<true, false>), grid, blk, 0, st, x, residual, (const float*)coef, y, quads, C);
    else if (relu) hipLaunchKernelGGL((bn2d_apply_kernel<false, true>), grid, blk, 0, st, x, residual, (const float*)coef, y, quads, C);
    else hipLaunchKernelGGL((bn2d_apply_kernel<false, false>), grid, blk, 0, st, x, residual, (const float*)coef, y, quads, C);
    HYB_LAUNCH_CHECK();
    return 0;
}

// dresidual (nullable) receives the gradient of the fused residual input; y is the forward OUTPUT (ReLU mask) -- pass NULL when no residual
// was fused: the mask is then recomputed from x (ignored when relu == 0)
extern "C" int hyb_bn2d_bwd(const float* dy, const float* x, const float* y, const float* gamma, const float* coef, float* dx, float* dresidual,
                            float* dgamma, float* dbeta, long long P, int C, int training, int relu, void* workspace, size_t workspace_bytes,
                            void* stream) {
    HYB_CHECK_ARG(dy && x && gamma && coef && dx && dgamma && dbeta && workspace && bn_shape_ok(P, C) && (!dresidual || !relu || y));
    if (workspace_bytes < hyb_bn2d_workspace(P, C)) return HYB_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const long long quads = P * (C / 4);
    const int blocks = bn_blocks(P, C);
    double* part = (double*)workspace;
    float* bcoef = (float*)((char*)workspace + al256((size_t)blocks * 2 * C * 8));
    const dim3 grid(blocks), blk(256);
    // a ReLU without a fused residual (the caller passes y = NULL): y > 0 <=> fmaf(x, a, b) > 0, the forward's own expression
    if (relu && y) hipLaunchKernelGGL(bn2d_bwd_reduce_kernel<1>, grid, blk, 0, st, dy, x, y, coef, part, quads, C);
    else if (relu) hipLaunchKernelGGL(bn2d_bwd_reduce_kernel<2>, grid, blk, 0, st, dy, x, y, coef, part, quads, C);
    else hipLaunchKernelGGL(bn2d_bwd_reduce_kernel<0>, grid, blk, 0, st, dy, x, y, coef, part, quads, C);
    HYB_LAUNCH_CHECK();
    hipLaunchKernelGGL(bn2d_bwd_finalize_kernel, dim3(C), dim3(64), 0, st, (const double*)part, blocks, P, C, gamma, coef, training,
                       dgamma, dbeta, bcoef);
    HYB_LAUNCH_CHECK();
    if (relu && y && dresidual) hipLaunchKernelGGL((bn2d_bwd_apply_kernel<1, true>), grid, blk, 0, st, dy, x, y, coef, (const float*)bcoef, dx, dresidual, quads, C);
    else if (relu && y) hipLaunchKernelGGL((bn2d_bwd_apply_kernel<1, false>), grid, blk, 0, st, dy, x, y, coef, (const float*)bcoef, dx, dresidual, quads, C);
    else if (relu) hipLaunchKernelGGL((bn2d_bwd_apply_kernel<2, false>), grid, blk, 0, st, dy, x, y, coef, (const float*)bcoef, dx, dresidual, quads, C);
    else if (dresidual) hipLaunchKernelGGL((bn2d_bwd_apply_kernel<0, true>), grid, blk, 0, st, dy, x, y, coef, (const float*)bcoef, dx, dresidual, quads, C);
    else hipLaunchKernelGGL((bn2d_bwd_apply_kernel<0, false>), grid, blk, 0, st, dy, x, y, coef, (const float*)bcoef, dx, dresidual, quads, C);
    HYB_LAUNCH_CHECK();
    return 0;
}

// nn.Dropout2d(p) in train mode on [N, H, W, C]: y = x * keep(n, c) / (1 - p); the backward is the same call on the gradient
extern "C" int hyb_dropout2d(const float* x, float* y, int N, long long HW, int C, float p, unsigned long long seed,
                             const unsigned long long* seed_inc, void* stream) {
    HYB_CHECK_ARG(x && y && N > 0 && HW > 0 && C >= 4 && C % 4 == 0 && p >= 0.f && p < 1.f);
    const long long quads = (long long)N * HW * (C / 4);
    long long blocks = (quads + 256 * 8 - 1) / (256 * 8); if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(dropout2d_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, x, y, quads, HW * (C / 4), C, p, 1.f / (1.f - p), seed,
                       seed_inc);
    HYB_LAUNCH_CHECK();
    return 0;
}
