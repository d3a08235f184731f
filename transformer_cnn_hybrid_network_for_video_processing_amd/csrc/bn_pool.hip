// BatchNorm2d (UNet.py:59) + ReLU (UNet.py:60) + MaxPool2d(2,2) (UNet.py:13) on NHWC activations,
// forward and backward, plus the global-average-pool frame token and layout/cast helpers.
// All of these are HBM-bound streaming kernels: 8 channels (16 B of bf16) per lane, channel-fastest.
#include <stdlib.h>
#include "hyb_common.h"

namespace {

// ---- BN statistics -> scale/shift -------------------------------------------------------
__global__ void bn_finalize_kernel(const float* __restrict__ stats, const float* __restrict__ gamma, const float* __restrict__ beta,
                                   float* running_mean, float* running_var, long long* __restrict__ nbt,
                                   int training, float momentum, float eps, long long count, int Co, int Cop,
                                   float* __restrict__ scale_shift, float* __restrict__ mean_invstd, float* running_out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c == 0 && training && nbt && !running_out) *nbt += 1;
    if (c >= Cop) return;
    float mean = 0.f, invstd = 0.f, g = 0.f, b = 0.f;
    if (c < Co) {
        g = gamma[c];
        b = beta[c];
        if (training) {
            const float inv_n = 1.0f / (float)count;
            mean = stats[c] * inv_n;
            float var = stats[Cop + c] * inv_n - mean * mean;      // biased variance (normalisation)
            var = fmaxf(var, 0.f);
            invstd = rsqrtf(var + eps);
            const float unbiased = count > 1 ? var * ((float)count / (float)(count - 1)) : var;
            // functional form (running_out != NULL): the updated statistics go to running_out[2][Co], the inputs stay untouched
            (running_out ? running_out : running_mean)[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
            (running_out ? running_out + Co : running_var)[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
        } else {
            mean = running_mean[c];
            invstd = rsqrtf(running_var[c] + eps);
        }
    }
    const float scale = g * invstd;
    scale_shift[c] = scale;
    scale_shift[Cop + c] = b - mean * scale;
    mean_invstd[c] = mean;
    mean_invstd[Cop + c] = invstd;
}

// partial rows [G][2][Cop] -> statistics -> finalize, one launch (1024 threads: 32 channels x 32 row groups per block)
__global__ __launch_bounds__(1024) void bn_stats_finalize_kernel(const float* __restrict__ part, int G, const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, float* running_mean,
                                                                 float* running_var, long long* __restrict__ nbt, float momentum,
                                                                 float eps, long long count, int Co, int Cop, float* __restrict__ scale_shift,
                                                                 float* __restrict__ mean_invstd, float* running_out) {
    long long i1, i2; float s1 = 0.f, s2 = 0.f;
    const bool ok1 = rows_reduce_1024(part, G, 2ll * Cop, i1, s1, 0, Cop);
    const bool ok2 = rows_reduce_1024(part, G, 2ll * Cop, i2, s2, Cop, Cop);
    if (blockIdx.x == 0 && threadIdx.x == 0 && nbt && !running_out) *nbt += 1;
    if (!(ok1 && ok2)) return;
    const int c = (int)i1;
    float mean = 0.f, invstd = 0.f, g = 0.f, b = 0.f;
    if (c < Co) {
        g = gamma[c]; b = beta[c];
        const float inv_n = 1.0f / (float)count;
        mean = s1 * inv_n;
        float var = fmaxf(s2 * inv_n - mean * mean, 0.f);
        invstd = rsqrtf(var + eps);
        const float unbiased = count > 1 ? var * ((float)count / (float)(count - 1)) : var;
        (running_out ? running_out : running_mean)[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
        (running_out ? running_out + Co : running_var)[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
    }
    const float scale = g * invstd;
    scale_shift[c] = scale;
    scale_shift[Cop + c] = b - mean * scale;
    mean_invstd[c] = mean;
    mean_invstd[Cop + c] = invstd;
}

// first maximum of the four transformed values in torch's window scan order (0,0),(0,1),(1,0),(1,1)
__device__ __forceinline__ int argmax4(float v0, float v1, float v2, float v3, float& vmax) {
    int a = 0;
    vmax = v0;
    if (v1 > vmax) { vmax = v1; a = 1; }
    if (v2 > vmax) { vmax = v2; a = 2; }
    if (v3 > vmax) { vmax = v3; a = 3; }
    return a;
}

// The three streaming kernels below walk pooled rows (n, ho): blockIdx strides over rows, threads over (wo, channel octet)
// inside the row -- 32-bit index math only, and a thread keeps one channel octet for the whole kernel.
__device__ __forceinline__ int row_threads(int oct) { return (256 / oct) * oct; }     // threads used per block (multiple of oct)

template <typename T>
__global__ __launch_bounds__(256) void bn_relu_pool_fwd_kernel(const T* __restrict__ y, const float* __restrict__ ss, T* __restrict__ out,
                                                               int N, int H, int W, int Cop) {
    const int OCT = Cop >> 3, Ho = H >> 1, Wo = W >> 1;
    const int nthr = row_threads(OCT);
    if ((int)threadIdx.x >= nthr) return;
    const int oc = threadIdx.x % OCT;
    float sc[8], sh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { sc[j] = ss[oc * 8 + j]; sh[j] = ss[Cop + oc * 8 + j]; }
    const int rows = N * Ho, rowlen = Wo * OCT;
    for (int row = blockIdx.x; row < rows; row += gridDim.x) {
        const int n = row / Ho, ho = row - n * Ho;
        const T* yrow = y + ((long long)(n * H + 2 * ho) * W) * Cop;
        T* orow = out + (long long)row * Wo * Cop;
        for (int idx = threadIdx.x; idx < rowlen; idx += nthr) {
            const int wo = idx / OCT;
            const T* src = yrow + (long long)(2 * wo) * Cop + oc * 8;
            Vec8<T> a, b, c, d, o;
            a.load(src); b.load(src + Cop); c.load(src + (long long)W * Cop); d.load(src + (long long)W * Cop + Cop);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float m = fmaxf(fmaxf(a.get(j) * sc[j] + sh[j], b.get(j) * sc[j] + sh[j]),
                                      fmaxf(c.get(j) * sc[j] + sh[j], d.get(j) * sc[j] + sh[j]));
                o.set(j, fmaxf(m, 0.f));
            }
            o.store(orow + (long long)wo * Cop + oc * 8);
        }
    }
}

// pass 1 of the backward: per-channel sum(dy) and sum(dy * xhat), dy routed through argmax and relu.
template <typename T>
__global__ __launch_bounds__(256) void bn_relu_pool_bwd_reduce_kernel(const T* __restrict__ dp, const T* __restrict__ y,
                                                                      const float* __restrict__ ss, const float* __restrict__ mi,
                                                                      float* __restrict__ sums, int N, int H, int W, int Cop) {
    extern __shared__ float red[];                 // [256 threads][16] per-thread partials, combined in a fixed order
    const int OCT = Cop >> 3, Ho = H >> 1, Wo = W >> 1;
    const int nthr = row_threads(OCT);
    if ((int)threadIdx.x < nthr) {
        const int oc = threadIdx.x % OCT;
        float sc[8], sh[8], mean[8], inv[8], a1[8], a2[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            sc[j] = ss[oc * 8 + j]; sh[j] = ss[Cop + oc * 8 + j];
            mean[j] = mi[oc * 8 + j]; inv[j] = mi[Cop + oc * 8 + j];
            a1[j] = 0.f; a2[j] = 0.f;
        }
        const int rows = N * Ho, rowlen = Wo * OCT;
        for (int row = blockIdx.x; row < rows; row += gridDim.x) {
            const int n = row / Ho, ho = row - n * Ho;
            const T* yrow = y + ((long long)(n * H + 2 * ho) * W) * Cop;
            const T* drow = dp + (long long)row * Wo * Cop;
            for (int idx = threadIdx.x; idx < rowlen; idx += nthr) {
                const int wo = idx / OCT;
                const T* src = yrow + (long long)(2 * wo) * Cop + oc * 8;
                Vec8<T> a, b, c, d, g;
                a.load(src); b.load(src + Cop); c.load(src + (long long)W * Cop); d.load(src + (long long)W * Cop + Cop);
                g.load(drow + (long long)wo * Cop + oc * 8);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float y0 = a.get(j), y1 = b.get(j), y2 = c.get(j), y3 = d.get(j);
                    float vmax;
                    const int am = argmax4(y0 * sc[j] + sh[j], y1 * sc[j] + sh[j], y2 * sc[j] + sh[j], y3 * sc[j] + sh[j], vmax);
                    const float ysel = am == 0 ? y0 : am == 1 ? y1 : am == 2 ? y2 : y3;
                    const float dy = vmax > 0.f ? g.get(j) : 0.f;
                    a1[j] += dy;
                    a2[j] += dy * (ysel - mean[j]) * inv[j];
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            red[threadIdx.x * 16 + j] = a1[j];
            red[threadIdx.x * 16 + 8 + j] = a2[j];
        }
    }
    __syncthreads();
    // one partial row per block (no float atomics anywhere): entry (which, ch) = sum over the threads that own octet ch/8
    for (int i = threadIdx.x; i < 2 * Cop; i += blockDim.x) {
        const int which = i / Cop, ch = i % Cop, o8 = ch >> 3, j = ch & 7;
        float acc = 0.f;
        for (int t = o8; t < nthr; t += OCT) acc += red[t * 16 + which * 8 + j];
        sums[(long long)blockIdx.x * 2 * Cop + i] = acc;
    }
}

// pass 1, fast form: the same two sums from the stage's OUTPUT instead of its full-resolution raw conv output.  The routed gradient is
// non-zero only at a window's arg-max and only if the ReLU passed, i.e. iff pooled > 0; there pooled = scale*y_sel + shift, so
// xhat_sel = (y_sel - mean)*invstd = (pooled - beta)/gamma with beta = shift + mean*scale, 1/gamma = invstd/scale: one quarter-resolution
// read (pooled) replaces four full-resolution ones (2.5x fewer bytes for the pass).  The inversion carries pooled's storage rounding
// u*|pooled| into xhat as u*|xhat + beta/gamma| (the raw-output path carries u*|xhat + mean/sigma|), so it is taken only where
// |beta/gamma| <= 4 (bf16 storage) / 1024 (fp32); an octet holding any other channel -- including gamma == 0, which cannot be inverted
// at all -- takes the raw-output path.
template <typename T>
__global__ __launch_bounds__(256) void bn_relu_pool_bwd_reduce_pooled_kernel(const T* __restrict__ dp, const T* __restrict__ pooled,
                                                                             const T* __restrict__ y, const float* __restrict__ ss,
                                                                             const float* __restrict__ mi, float* __restrict__ sums, int N, int H,
                                                                             int W, int Cop) {
    extern __shared__ float red[];
    const int OCT = Cop >> 3, Ho = H >> 1, Wo = W >> 1;
    const int nthr = row_threads(OCT);
    if ((int)threadIdx.x < nthr) {
        const int oc = threadIdx.x % OCT;
        float sc[8], sh[8], mean[8], inv[8], bta[8], rg[8], a1[8], a2[8];
        bool singular = false;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            sc[j] = ss[oc * 8 + j]; sh[j] = ss[Cop + oc * 8 + j];
            mean[j] = mi[oc * 8 + j]; inv[j] = mi[Cop + oc * 8 + j];
            bta[j] = fmaf(mean[j], sc[j], sh[j]);
            rg[j] = inv[j] / sc[j];
            singular = singular || !(fabsf(bta[j] * rg[j]) <= (sizeof(T) == 2 ? 4.f : 1024.f) && fabsf(rg[j]) <= 3.0e38f);
            a1[j] = 0.f; a2[j] = 0.f;
        }
        const int rows = N * Ho, rowlen = Wo * OCT;
        if (!singular) {
            // dp and pooled are flat streams of 8-channel groups; group i belongs to octet i % OCT, and the stride nthr * gridDim keeps a
            // thread on its octet.  Four groups per iteration: eight 16-byte loads in flight per lane.
            const long long total = (long long)rows * rowlen, stride = (long long)nthr * gridDim.x;
            long long i = (long long)blockIdx.x * nthr + threadIdx.x;
            auto accum = [&](const Vec8<T>& pv, const Vec8<T>& g) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float v = pv.get(j);
                    const float dy = v > 0.f ? g.get(j) : 0.f;
                    a1[j] += dy;
                    a2[j] += dy * (v - bta[j]) * rg[j];
                }
            };
            for (; i + 3 * stride < total; i += 4 * stride) {
                Vec8<T> pv[4], g[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { pv[u].load(pooled + (i + u * stride) * 8); g[u].load(dp + (i + u * stride) * 8); }
#pragma unroll
                for (int u = 0; u < 4; ++u) accum(pv[u], g[u]);
            }
            for (; i < total; i += stride) {
                Vec8<T> pv, g;
                pv.load(pooled + i * 8); g.load(dp + i * 8);
                accum(pv, g);
            }
        } else {
            for (int row = blockIdx.x; row < rows; row += gridDim.x) {
                const long long rbase = (long long)row * Wo * Cop;
                const int n = row / Ho, ho = row - n * Ho;
                const T* yrow = y + ((long long)(n * H + 2 * ho) * W) * Cop;
                for (int idx = threadIdx.x; idx < rowlen; idx += nthr) {
                    const int wo = idx / OCT;
                    const T* src = yrow + (long long)(2 * wo) * Cop + oc * 8;
                    Vec8<T> a, b, c, d, g;
                    a.load(src); b.load(src + Cop); c.load(src + (long long)W * Cop); d.load(src + (long long)W * Cop + Cop);
                    g.load(dp + rbase + (long long)wo * Cop + oc * 8);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float y0 = a.get(j), y1 = b.get(j), y2 = c.get(j), y3 = d.get(j);
                        float vmax;
                        const int am = argmax4(y0 * sc[j] + sh[j], y1 * sc[j] + sh[j], y2 * sc[j] + sh[j], y3 * sc[j] + sh[j], vmax);
                        const float ysel = am == 0 ? y0 : am == 1 ? y1 : am == 2 ? y2 : y3;
                        const float dy = vmax > 0.f ? g.get(j) : 0.f;
                        a1[j] += dy;
                        a2[j] += dy * (ysel - mean[j]) * inv[j];
                    }
                }
            }
        }
        // Block partial row without the old 16-float-stride LDS scatter (89 % of its LDS cycles were bank conflicts, and with three
        // groups per thread at stage 4 the epilogue WAS the kernel): lanes of a wave that own the same octet (lane = oc mod OCT) add up
        // through cross-lane steps first -- a fixed butterfly, so the sum order is fixed -- and only OCT lanes per wave touch LDS.
        if (OCT <= 32 && (OCT & (OCT - 1)) == 0 && nthr == 256) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                for (int o = 32; o >= OCT; o >>= 1) { a1[j] += __shfl_xor(a1[j], o, 64); a2[j] += __shfl_xor(a2[j], o, 64); }
            }
            const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
            if (lane < OCT) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {                   // red[wave][which][j][oc]: consecutive lanes -> consecutive banks
                    red[((wv * 2 + 0) * 8 + j) * OCT + lane] = a1[j];
                    red[((wv * 2 + 1) * 8 + j) * OCT + lane] = a2[j];
                }
            }
                    } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                red[threadIdx.x * 16 + j] = a1[j];
                red[threadIdx.x * 16 + 8 + j] = a2[j];
            }
        }
    }
    __syncthreads();
    if (OCT <= 32 && (OCT & (OCT - 1)) == 0 && nthr == 256) {
        for (int i = threadIdx.x; i < 2 * Cop; i += blockDim.x) {
            const int which = i / Cop, ch = i % Cop, o8 = ch >> 3, j = ch & 7;
            float acc = 0.f;
#pragma unroll
            for (int wv = 0; wv < 4; ++wv) acc += red[((wv * 2 + which) * 8 + j) * OCT + o8];
            sums[(long long)blockIdx.x * 2 * Cop + i] = acc;
        }
        return;
    }
    for (int i = threadIdx.x; i < 2 * Cop; i += blockDim.x) {
        const int which = i / Cop, ch = i % Cop, o8 = ch >> 3, j = ch & 7;
        float acc = 0.f;
        for (int t = o8; t < nthr; t += OCT) acc += red[t * 16 + which * 8 + j];
        sums[(long long)blockIdx.x * 2 * Cop + i] = acc;
    }
}

// pass 2: dense gradient w.r.t. the raw conv output.
template <typename T>
__global__ __launch_bounds__(256) void bn_relu_pool_bwd_dx_kernel(const T* __restrict__ dp, const T* __restrict__ y,
                                                                  const float* __restrict__ ss, const float* __restrict__ mi,
                                                                  const float* __restrict__ gamma, const float* __restrict__ sums,
                                                                  int training, float inv_count, T* __restrict__ dyraw,
                                                                  int N, int H, int W, int Co, int Cop) {
    const int OCT = Cop >> 3, Ho = H >> 1, Wo = W >> 1;
    const bool oddW = (W & 1) != 0, oddH = (H & 1) != 0;
    const int nthr = row_threads(OCT);
    if ((int)threadIdx.x >= nthr) return;
    const int oc = threadIdx.x % OCT;
    float sc[8], sh[8], mean[8], inv[8], kk[8], m1[8], m2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int ch = oc * 8 + j;
        sc[j] = ss[ch]; sh[j] = ss[Cop + ch]; mean[j] = mi[ch]; inv[j] = mi[Cop + ch];
        kk[j] = (ch < Co ? gamma[ch] : 0.f) * inv[j];
        m1[j] = training ? sums[ch] * inv_count : 0.f;
        m2[j] = training ? sums[Cop + ch] * inv_count : 0.f;
    }
    auto dense_only = [&](long long off) {       // pixels outside every pooling window (odd H/W): dy = 0
        Vec8<T> e, oe;
        e.load(y + off);
#pragma unroll
        for (int j = 0; j < 8; ++j) oe.set(j, kk[j] * (-m1[j] - (e.get(j) - mean[j]) * inv[j] * m2[j]));
        oe.store(dyraw + off);
    };
    const int rows = N * Ho, rowlen = Wo * OCT;
    for (int row = blockIdx.x; row < rows; row += gridDim.x) {
        const int n = row / Ho, ho = row - n * Ho;
        const long long rbase = ((long long)(n * H + 2 * ho) * W) * Cop;
        const T* drow = dp + (long long)row * Wo * Cop;
        for (int idx = threadIdx.x; idx < rowlen; idx += nthr) {
            const int wo = idx / OCT;
            const long long base = rbase + (long long)(2 * wo) * Cop + oc * 8;
            Vec8<T> a, b, c, d, g;
            a.load(y + base); b.load(y + base + Cop); c.load(y + base + (long long)W * Cop); d.load(y + base + (long long)W * Cop + Cop);
            g.load(drow + (long long)wo * Cop + oc * 8);
            Vec8<T> o0, o1, o2, o3;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float y0 = a.get(j), y1 = b.get(j), y2 = c.get(j), y3 = d.get(j);
                float vmax;
                const int am = argmax4(y0 * sc[j] + sh[j], y1 * sc[j] + sh[j], y2 * sc[j] + sh[j], y3 * sc[j] + sh[j], vmax);
                const float dy = vmax > 0.f ? g.get(j) : 0.f;
                o0.set(j, kk[j] * ((am == 0 ? dy : 0.f) - m1[j] - (y0 - mean[j]) * inv[j] * m2[j]));
                o1.set(j, kk[j] * ((am == 1 ? dy : 0.f) - m1[j] - (y1 - mean[j]) * inv[j] * m2[j]));
                o2.set(j, kk[j] * ((am == 2 ? dy : 0.f) - m1[j] - (y2 - mean[j]) * inv[j] * m2[j]));
                o3.set(j, kk[j] * ((am == 3 ? dy : 0.f) - m1[j] - (y3 - mean[j]) * inv[j] * m2[j]));
            }
            o0.store(dyraw + base); o1.store(dyraw + base + Cop);
            o2.store(dyraw + base + (long long)W * Cop); o3.store(dyraw + base + (long long)W * Cop + Cop);
            if (oddW && wo == Wo - 1) { dense_only(base + 2 * Cop); dense_only(base + 2 * Cop + (long long)W * Cop); }
            if (oddH && ho == Ho - 1) {
                dense_only(base + 2ll * W * Cop); dense_only(base + 2ll * W * Cop + Cop);
                if (oddW && wo == Wo - 1) dense_only(base + 2ll * W * Cop + 2 * Cop);
            }
        }
    }
}

// sums[2][Cop] = sum over G partial rows (fixed order); also dbeta[c] = sums[c], dgamma[c] = sums[Cop + c] for c < Co
__global__ __launch_bounds__(1024) void bn_rows_reduce_kernel(const float* __restrict__ part, float* __restrict__ out, int G, int n,
                                                              float* __restrict__ dgamma, float* __restrict__ dbeta, int Co, int Cop) {
    long long i; float v;
    if (!rows_reduce_1024(part, G, n, i, v)) return;
    out[i] = v;
    if (i < Cop) { if (dbeta && i < Co) dbeta[i] = v; }
    else if (dgamma && i - Cop < Co) dgamma[i - Cop] = v;
}

__global__ void bn_param_grad_kernel(const float* __restrict__ sums, float* __restrict__ dgamma, float* __restrict__ dbeta, int Co, int Cop) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= Co) return;
    if (dbeta) dbeta[c] = sums[c];
    if (dgamma) dgamma[c] = sums[Cop + c];
}

// ---- global average pool ------------------------------------------------------------------
// one block per (frame, 32-octet slab): 32 octets x 8 pixel groups, LDS combine
template <typename T, typename TO = T>
__global__ __launch_bounds__(256) void gap_fwd_kernel(const T* __restrict__ x, TO* __restrict__ feat, int HW, int Cp, int octBlocks) {
    __shared__ float red[8][32][9];
    const int OCT = Cp >> 3;
    const int n = blockIdx.x / octBlocks, ob = blockIdx.x % octBlocks;
    const int ocl = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const int oc = ob * 32 + ocl;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (oc < OCT) {
        const T* src = x + (long long)n * HW * Cp + oc * 8;
        int p = grp;
        for (; p + 3 * 8 < HW; p += 4 * 8) {              // four independent loads in flight, additions in pixel order
            Vec8<T> v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u].load(src + (long long)(p + u * 8) * Cp);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += v[u].get(j);
        }
        for (; p < HW; p += 8) {
            Vec8<T> v;
            v.load(src + (long long)p * Cp);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += v.get(j);
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) red[grp][ocl][j] = acc[j];
    __syncthreads();
    if (grp == 0 && oc < OCT) {
        Vec8<TO> o;
        const float inv = 1.0f / (float)HW;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) s += red[k][ocl][j];
            o.set(j, s * inv);
        }
        o.store(feat + (long long)n * Cp + oc * 8);
    }
}
template <typename T, typename TO = T>
__global__ void gap_bwd_kernel(const T* __restrict__ dfeat, TO* __restrict__ dx, int HW, int Cp, long long total) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int OCT = Cp >> 3;
    const int oc = (int)(i % OCT);
    const long long np = i / OCT;       // n*HW + p
    const long long n = np / HW;
    Vec8<T> v;
    Vec8<TO> o;
    v.load(dfeat + n * Cp + oc * 8);
    const float inv = 1.0f / (float)HW;
#pragma unroll
    for (int j = 0; j < 8; ++j) o.set(j, v.get(j) * inv);
    o.store(dx + np * Cp + oc * 8);
}

// ---- layout / cast helpers ------------------------------------------------------------------
template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ src, T* __restrict__ dst, int C, int H, int W, int Cp, long long total) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % Cp);
    const long long pix = i / Cp;
    const long long hw = (long long)H * W;
    const long long n = pix / hw, p = pix % hw;
    dst[i] = from_f32<T>(c < C ? src[(n * C + c) * hw + p] : 0.f);
}
template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* __restrict__ src, float* __restrict__ dst, int C, int H, int W, int Cp, long long total) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const long long hw = (long long)H * W;
    const long long p = i % hw;
    const int c = (int)((i / hw) % C);
    const long long n = i / (hw * C);
    dst[i] = to_f32<T>(src[(n * hw + p) * Cp + c]);
}
template <typename T>
__global__ void cast_to_f32_kernel(const T* __restrict__ s, float* __restrict__ d, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) d[i] = to_f32<T>(s[i]);
}
template <typename T>
__global__ void cast_from_f32_kernel(const float* __restrict__ s, T* __restrict__ d, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) d[i] = from_f32<T>(s[i]);
}

constexpr int BN_MAX_ROWBLOCKS = 1024;
inline int row_grid(int rows) { return rows < BN_MAX_ROWBLOCKS ? (rows < 1 ? 1 : rows) : BN_MAX_ROWBLOCKS; }

inline int stream_grid(long long total, int cap = 4096) {
    long long b = (total + 255) / 256;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (int)b;
}

}  // namespace

#define HYB_DISPATCH_T(dtype, CALL_F32, CALL_BF16) \
    do { if ((dtype) == HYB_F32) { CALL_F32; } else if ((dtype) == HYB_BF16) { CALL_BF16; } else return HYB_E_ARG; } while (0)

HybProfileHook g_hyb_hooks[16];
int g_hyb_hooks_active = 0;

extern "C" int hyb_profile_set(int slot, int kernel_id, int a, int b, void* ev_start, void* ev_stop) {
    HYB_CHECK_ARG(slot >= 0 && slot < 16 && ev_start && ev_stop && kernel_id > 0);
    g_hyb_hooks[slot] = HybProfileHook{kernel_id, a, b, (hipEvent_t)ev_start, (hipEvent_t)ev_stop};
    g_hyb_hooks_active = 1;
    return 0;
}
extern "C" int hyb_profile_clear(void) {
    for (int i = 0; i < 16; ++i) g_hyb_hooks[i] = HybProfileHook{0, 0, 0, nullptr, nullptr};
    g_hyb_hooks_active = 0;
    return 0;
}

extern "C" int hyb_abi_version(void) { return 9; }
extern "C" int hyb_dtype_size(int dtype) { return dtype == HYB_F32 ? 4 : dtype == HYB_BF16 ? 2 : HYB_E_ARG; }
extern "C" int hyb_pad_channels(int c) { return (c + 31) / 32 * 32; }

extern "C" int hyb_bn_finalize(const float* stats, const float* gamma, const float* beta, float* running_mean, float* running_var,
                               long long* nbt, int training, float momentum, float eps, long long count, int Co, int Cop,
                               float* scale_shift, float* mean_invstd, float* running_out, void* stream) {
    HYB_CHECK_ARG(gamma && beta && running_mean && running_var && scale_shift && mean_invstd && Co > 0 && Cop >= Co && count > 0);
    HYB_CHECK_ARG(!training || stats);
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(hyb_cdiv(Cop, 256)), dim3(256), 0, (hipStream_t)stream, stats, gamma, beta, running_mean,
                       running_var, nbt, training, momentum, eps, count, Co, Cop, scale_shift, mean_invstd, running_out);
    HYB_LAUNCH_CHECK();
    return 0;
}

extern "C" int hyb_bn_relu_pool_fwd(int dtype, const void* y, const float* ss, void* pooled, int N, int H, int W, int Cop, void* stream) {
    HYB_CHECK_ARG(y && ss && pooled && N > 0 && H >= 2 && W >= 2 && Cop % 32 == 0 && Cop > 0);
    HYB_CHECK_ARG(Cop / 8 <= 256);
    const int grid = row_grid(N * (H / 2));
    hipStream_t st = (hipStream_t)stream;
    HYB_DISPATCH_T(dtype,
        hipLaunchKernelGGL(bn_relu_pool_fwd_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)y, ss, (float*)pooled, N, H, W, Cop),
        hipLaunchKernelGGL(bn_relu_pool_fwd_kernel<bf16>, dim3(grid), dim3(256), 0, st, (const bf16*)y, ss, (bf16*)pooled, N, H, W, Cop));
    HYB_LAUNCH_CHECK();
    return 0;
}

extern "C" size_t hyb_bn_bwd_reduce_workspace(int Cop) { return Cop > 0 ? (size_t)BN_MAX_ROWBLOCKS * 2 * Cop * sizeof(float) : 0; }

extern "C" int hyb_bn_stats_finalize(const float* stats_partials, int G, const float* gamma, const float* beta, float* running_mean,
                                     float* running_var, long long* nbt, float momentum, float eps, long long count, int Co, int Cop,
                                     float* scale_shift, float* mean_invstd, float* running_out, void* stream) {
    HYB_CHECK_ARG(stats_partials && G > 0 && gamma && beta && running_mean && running_var && scale_shift && mean_invstd && Co > 0 && Cop >= Co && count > 0);
    hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3(hyb_cdiv(Cop, 32)), dim3(1024), 0, (hipStream_t)stream, stats_partials, G, gamma, beta,
                       running_mean, running_var, nbt, momentum, eps, count, Co, Cop, scale_shift, mean_invstd, running_out);
    HYB_LAUNCH_CHECK();
    return 0;
}

extern "C" int hyb_bn_relu_pool_bwd_reduce(int dtype, const void* dpooled, const void* y, const void* pooled, const float* ss, const float* mi,
                                           float* sums, float* partials, float* dgamma, float* dbeta, int N, int H, int W, int Co, int Cop,
                                           void* stream) {
    HYB_CHECK_ARG(dpooled && y && ss && mi && sums && partials && N > 0 && H >= 2 && W >= 2 && Cop % 32 == 0 && Cop > 0 && Cop / 8 <= 256);
    // (the pooled form streams flat groups, not rows: one block per CU (measured 26 / 17.5 / 15.8 us at 1024 / 512 / 256 blocks) with 4 x 2 loads in flight per lane cover the memory latency, and
    // fewer blocks mean fewer partial rows for the second launch)
    static const int pooled_grid = getenv("HYB_BN_REDUCE_WGS") ? atoi(getenv("HYB_BN_REDUCE_WGS")) : 256;
    const int grid = (pooled && pooled_grid >= 1 && pooled_grid <= BN_MAX_ROWBLOCKS) ? (N * (H / 2) < pooled_grid ? N * (H / 2) : pooled_grid) : row_grid(N * (H / 2));
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = 256 * 16 * sizeof(float);
    if (pooled) {
        HYB_DISPATCH_T(dtype,
            hipLaunchKernelGGL(bn_relu_pool_bwd_reduce_pooled_kernel<float>, dim3(grid), dim3(256), lds, st, (const float*)dpooled, (const float*)pooled, (const float*)y, ss, mi, partials, N, H, W, Cop),
            hipLaunchKernelGGL(bn_relu_pool_bwd_reduce_pooled_kernel<bf16>, dim3(grid), dim3(256), lds, st, (const bf16*)dpooled, (const bf16*)pooled, (const bf16*)y, ss, mi, partials, N, H, W, Cop));
    } else
    HYB_DISPATCH_T(dtype,
        hipLaunchKernelGGL(bn_relu_pool_bwd_reduce_kernel<float>, dim3(grid), dim3(256), lds, st, (const float*)dpooled, (const float*)y, ss, mi, partials, N, H, W, Cop),
        hipLaunchKernelGGL(bn_relu_pool_bwd_reduce_kernel<bf16>, dim3(grid), dim3(256), lds, st, (const bf16*)dpooled, (const bf16*)y, ss, mi, partials, N, H, W, Cop));
    HYB_LAUNCH_CHECK();
    hipLaunchKernelGGL(bn_rows_reduce_kernel, dim3(hyb_cdiv(2 * Cop, 32)), dim3(1024), 0, st, partials, sums, grid, 2 * Cop, dgamma, dbeta, Co, Cop);
    HYB_LAUNCH_CHECK();
    return 0;
}

extern "C" int hyb_bn_relu_pool_bwd_dx(int dtype, const void* dpooled, const void* y, const float* ss, const float* mi, const float* gamma,
                                       const float* sums, int training, long long count, void* dyraw, float* dgamma, float* dbeta,
                                       int N, int H, int W, int Co, int Cop, void* stream) {
    HYB_CHECK_ARG(dpooled && y && ss && mi && gamma && sums && dyraw && N > 0 && H >= 2 && W >= 2 && Cop % 32 == 0 && Co <= Cop && count > 0);
    HYB_CHECK_ARG(Cop / 8 <= 256);
    const int grid = row_grid(N * (H / 2));
    const float inv_count = 1.0f / (float)count;
    hipStream_t st = (hipStream_t)stream;
    HYB_DISPATCH_T(dtype,
        hipLaunchKernelGGL(bn_relu_pool_bwd_dx_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)dpooled, (const float*)y, ss, mi, gamma, sums, training, inv_count, (float*)dyraw, N, H, W, Co, Cop),
        hipLaunchKernelGGL(bn_relu_pool_bwd_dx_kernel<bf16>, dim3(grid), dim3(256), 0, st, (const bf16*)dpooled, (const bf16*)y, ss, mi, gamma, sums, training, inv_count, (bf16*)dyraw, N, H, W, Co, Cop));
    HYB_LAUNCH_CHECK();
    if (dgamma || dbeta) {
        hipLaunchKernelGGL(bn_param_grad_kernel, dim3(hyb_cdiv(Co, 256)), dim3(256), 0, st, sums, dgamma, dbeta, Co, Cop);
        HYB_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int hyb_gap_fwd(int dtype, const void* x, void* feat, int N, int HW, int Cp, void* stream) {
    HYB_CHECK_ARG(x && feat && N > 0 && HW > 0 && Cp % 32 == 0 && Cp > 0);
    const int octBlocks = hyb_cdiv(Cp / 8, 32);
    hipStream_t st = (hipStream_t)stream;
    HYB_DISPATCH_T(dtype,
        hipLaunchKernelGGL(gap_fwd_kernel<float>, dim3(N * octBlocks), dim3(256), 0, st, (const float*)x, (float*)feat, HW, Cp, octBlocks),
        hipLaunchKernelGGL(gap_fwd_kernel<bf16>, dim3(N * octBlocks), dim3(256), 0, st, (const bf16*)x, (bf16*)feat, HW, Cp, octBlocks));
    HYB_LAUNCH_CHECK();
    return 0;
}
extern "C" int hyb_gap_bwd(int dtype, const void* dfeat, void* dx, int N, int HW, int Cp, void* stream) {
    HYB_CHECK_ARG(dfeat && dx && N > 0 && HW > 0 && Cp % 32 == 0 && Cp > 0);
    const long long total = (long long)N * HW * (Cp / 8);
    hipStream_t st = (hipStream_t)stream;
    HYB_DISPATCH_T(dtype,
        hipLaunchKernelGGL(gap_bwd_kernel<float>, dim3(hyb_cdiv(total, 256)), dim3(256), 0, st, (const float*)dfeat, (float*)dx, HW, Cp, total),
        hipLaunchKernelGGL(gap_bwd_kernel<bf16>, dim3(hyb_cdiv(total, 256)), dim3(256), 0, st, (const bf16*)dfeat, (bf16*)dx, HW, Cp, total));
    HYB_LAUNCH_CHECK();
    return 0;
}

// Internal (hyb_temporal_* with HYB_H_BF16): bf16 pooled map -> fp32 frame features, fp32 feature gradient -> bf16 map gradient
int hyb_gap_fwd_h16(const void* x, float* feat, int N, int HW, int Cp, hipStream_t st) {
    if (!x || !feat || N <= 0 || HW <= 0 || Cp % 32 != 0 || Cp <= 0) return HYB_E_ARG;
    const int octBlocks = hyb_cdiv(Cp / 8, 32);
    hipLaunchKernelGGL((gap_fwd_kernel<bf16, float>), dim3(N * octBlocks), dim3(256), 0, st, (const bf16*)x, feat, HW, Cp, octBlocks);
    HYB_LAUNCH_CHECK();
    return 0;
}
int hyb_gap_bwd_h16(const float* dfeat, void* dx, int N, int HW, int Cp, hipStream_t st) {
    if (!dfeat || !dx || N <= 0 || HW <= 0 || Cp % 32 != 0 || Cp <= 0) return HYB_E_ARG;
    const long long total = (long long)N * HW * (Cp / 8);
    hipLaunchKernelGGL((gap_bwd_kernel<float, bf16>), dim3(hyb_cdiv(total, 256)), dim3(256), 0, st, dfeat, (bf16*)dx, HW, Cp, total);
    HYB_LAUNCH_CHECK();
    return 0;
}

extern "C" int hyb_nchw_to_nhwc(int dtype, const float* src, void* dst, int N, int C, int H, int W, int Cp, void* stream) {
    HYB_CHECK_ARG(src && dst && N > 0 && C > 0 && H > 0 && W > 0 && Cp >= C);
    const long long total = (long long)N * H * W * Cp;
    hipStream_t st = (hipStream_t)stream;
    HYB_DISPATCH_T(dtype,
        hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, dim3(hyb_cdiv(total, 256)), dim3(256), 0, st, src, (float*)dst, C, H, W, Cp, total),
        hipLaunchKernelGGL(nchw_to_nhwc_kernel<bf16>, dim3(hyb_cdiv(total, 256)), dim3(256), 0, st, src, (bf16*)dst, C, H, W, Cp, total));
    HYB_LAUNCH_CHECK();
    return 0;
}
extern "C" int hyb_nhwc_to_nchw(int dtype, const void* src, float* dst, int N, int C, int H, int W, int Cp, void* stream) {
    HYB_CHECK_ARG(src && dst && N > 0 && C > 0 && H > 0 && W > 0 && Cp >= C);
    const long long total = (long long)N * H * W * C;
    hipStream_t st = (hipStream_t)stream;
    HYB_DISPATCH_T(dtype,
        hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, dim3(hyb_cdiv(total, 256)), dim3(256), 0, st, (const float*)src, dst, C, H, W, Cp, total),
        hipLaunchKernelGGL(nhwc_to_nchw_kernel<bf16>, dim3(hyb_cdiv(total, 256)), dim3(256), 0, st, (const bf16*)src, dst, C, H, W, Cp, total));
    HYB_LAUNCH_CHECK();
    return 0;
}
extern "C" int hyb_cast_to_f32(int dtype, const void* src, float* dst, long long n, void* stream) {
    HYB_CHECK_ARG(src && dst && n > 0);
    hipStream_t st = (hipStream_t)stream;
    HYB_DISPATCH_T(dtype,
        hipLaunchKernelGGL(cast_to_f32_kernel<float>, dim3(stream_grid(n)), dim3(256), 0, st, (const float*)src, dst, n),
        hipLaunchKernelGGL(cast_to_f32_kernel<bf16>, dim3(stream_grid(n)), dim3(256), 0, st, (const bf16*)src, dst, n));
    HYB_LAUNCH_CHECK();
    return 0;
}
extern "C" int hyb_cast_from_f32(int dtype, const float* src, void* dst, long long n, void* stream) {
    HYB_CHECK_ARG(src && dst && n > 0);
    hipStream_t st = (hipStream_t)stream;
    HYB_DISPATCH_T(dtype,
        hipLaunchKernelGGL(cast_from_f32_kernel<float>, dim3(stream_grid(n)), dim3(256), 0, st, src, (float*)dst, n),
        hipLaunchKernelGGL(cast_from_f32_kernel<bf16>, dim3(stream_grid(n)), dim3(256), 0, st, src, (bf16*)dst, n));
    HYB_LAUNCH_CHECK();
    return 0;
}
