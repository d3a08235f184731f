// FCT backward kernels (autograd of FCT.py:24-254 and Metrics.py:14-22).  NHWC fp32, exact-fp32 MFMA for
// the contractions.  Every reduction over pixels (weight / bias / LayerNorm-affine gradients) goes through per-workgroup partial
// rows that are summed in a fixed order: no float atomics, results are reproducible run to run.
//
//   conv3x3 backward     dz = dy * act'(.) -> dgrad: dcol = dz Wp (MFMA GEMM) + col2im gather;  wgrad: dWp = dz^T col over pixel
//                        slices (MFMA, partial slabs) ; db = column sums of dz
//   q/k/v projection     pass 1 per pixel: LayerNorm backward + ReLU mask -> dr_j, partial sums of the 12 per-channel gradients;
//                        pass 2: dx = sum_j depthwise-conv-transpose(dr_j)
//   LayerNorm over C, MaxPool2d(2) (gradient to the first maximum in scan order, like torch), Upsample x2 (sum of the 2x2 block),
//   cat (split), DiceLoss, dropout (counter-based mask regenerated from the seed).
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
int hyb_flash_attention_bwd(int dtype, const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse, float* delta_ws,
                            void* dq, void* dk, void* dv, int N, int L, int H, int dhp, int ld, float scale, hipStream_t st, int dh_true);

namespace {

inline size_t al256(size_t v) { return (v + 255) & ~(size_t)255; }
inline int up8(int v) { return (v + 7) / 8 * 8; }
inline int grid1(long long n) { return hyb_cdiv(n, 256); }
#define FCT_TRY(call) do { int rc_ = (call); if (rc_ != 0) return rc_; } while (0)
constexpr long long CHUNK_BYTES = 512ll << 20;
constexpr int SLICE_ROWS = 4096;                  // pixel rows per weight-gradient slice (one partial slab each)

__device__ __forceinline__ float gelu_grad(float z) {            // d/dz [0.5 z (1 + erf(z / sqrt 2))]
    return 0.5f * (1.f + erff(z * 0.70710678118654752f)) + z * 0.3989422804014327f * expf(-0.5f * z * z);
}

// dz[p][c] (row stride ldz >= Co, padded columns zero) = dy[p][c] * act'(saved)
__global__ void act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ saved, float* __restrict__ dz, long long P, int Co, int ldz,
                               int act) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P * ldz) return;
    const long long p = i / ldz;
    const int c = (int)(i - p * ldz);
    float v = 0.f;
    if (c < Co) {
        v = dy[p * Co + c];
        if (act == HYB_ACT_RELU) v = saved[p * Co + c] > 0.f ? v : 0.f;
        else if (act == HYB_ACT_GELU) v *= gelu_grad(saved[p * Co + c]);
        else if (act == HYB_ACT_SIGMOID) { const float y = saved[p * Co + c]; v *= y * (1.f - y); }
    }
    dz[i] = v;
}

// weights of the input-gradient convolution (stride 1): wpd [Ci][k*k*Co8], column = (ky'*k + kx')*Co8 + co holds w[co][ci][k-1-ky'][k-1-kx']
// (taps flipped, channel roles swapped), zero for co >= Co
__global__ void conv_pack_flip_kernel(const float* __restrict__ w, float* __restrict__ wpd, int Co, int Ci, int k, int Co8) {
    const int Kd = k * k * Co8;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Ci * Kd) return;
    const int ci = i / Kd, r = i - ci * Kd;
    const int tap = r / Co8, co = r - tap * Co8;
    const int ky = tap / k, kx = tap - ky * k;
    wpd[i] = co < Co ? w[(((long long)co * Ci + ci) * k + (k - 1 - ky)) * k + (k - 1 - kx)] : 0.f;
}
// wpt [Kp][Co8] <- w [Co][Ci][k][k] (column = tap*Ci + ci, kk = k*k taps), zero padded
__global__ void conv_pack_t_kernel(const float* __restrict__ w, float* __restrict__ wpt, int Co, int Ci, int kk, int Kp, int Co8) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Kp * Co8) return;
    const int k = i / Co8, co = i - k * Co8;
    float v = 0.f;
    if (k < kk * Ci && co < Co) { const int tap = k / Ci, ci = k - tap * Ci; v = w[((long long)co * Ci + ci) * kk + tap]; }
    wpt[i] = v;
}

// ---- generic sliced weight gradient:  part[slice][n][k] = sum_{p in slice} dy[p][n] * x[p][k]   (n < Nn, k < K; 64 x 64 tiles) -------
// grid (tiles_k * tiles_n, slices); LDS tiles [64][32 + 8] of both operands, transposed while staging (rows of dy / x are pixels).
constexpr int WG_BK = 32, WG_LD = WG_BK + 8;
__global__ __launch_bounds__(256) void sliced_wgrad_kernel(const float* __restrict__ dy, int lddy, const float* __restrict__ x, int ldx,
                                                           float* __restrict__ part, long long P, int Nn, int K, int tiles_k, int slice_rows) {
    __shared__ __attribute__((aligned(16))) float As[64 * WG_LD];
    __shared__ __attribute__((aligned(16))) float Bs[64 * WG_LD];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, p = lane & 15, q = lane >> 4;
    const int tk = blockIdx.x % tiles_k, tn = blockIdx.x / tiles_k;
    const int n0 = tn * 64, k0 = tk * 64;
    const long long r_begin = (long long)blockIdx.y * slice_rows;
    const long long r_end = r_begin + slice_rows < P ? r_begin + slice_rows : P;
    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (long long r0 = r_begin; r0 < r_end; r0 += WG_BK) {
        __syncthreads();
        {   // 256 threads: pixel row r = tid / 8 (32 rows), 8 columns each of the 64-wide tile
            const int r = tid >> 3, seg = tid & 7;
            const long long pr = r0 + r;
            const bool ok = pr < r_end;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int n = n0 + seg * 8 + j, k = k0 + seg * 8 + j;
                As[(seg * 8 + j) * WG_LD + r] = (ok && n < Nn) ? dy[pr * lddy + n] : 0.f;
                Bs[(seg * 8 + j) * WG_LD + r] = (ok && k < K) ? x[pr * ldx + k] : 0.f;
            }
        }
        __syncthreads();
        Frag<float> a[2], b[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) frag_load(a[i], As + (wm * 32 + i * 16 + p) * WG_LD + 8 * q);
#pragma unroll
        for (int j = 0; j < 2; ++j) frag_load(b[j], Bs + (wn * 32 + j * 16 + p) * WG_LD + 8 * q);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = mma32(a[i], b[j], acc[i][j]);
    }
    float* out = part + (long long)blockIdx.y * Nn * K;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int k = k0 + wn * 32 + j * 16 + p;
            if (k >= K) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + wm * 32 + i * 16 + 4 * q + r;
                if (n < Nn) out[(long long)n * K + k] = acc[i][j][r];
            }
        }
}
// Second generation: no LDS staging and no transposition at all.  With the 4-deep fp32 MFMA both operands are read straight from
// the row-major [pixel][feature] matrices: lane (m = lane & 15, kq = lane >> 4) holds dy[p0 + kq][n0 + m] and x[p0 + kq][k0 + m] --
// 16 consecutive features of 4 consecutive pixels per wave-load, fully coalesced.  A block owns NTL x KTL 16x16 output tiles of
// one pixel slice, its 4 waves interleave over the slice's 4-pixel groups and are combined in a fixed order at the end.  The bias
// gradient (column sums of dy) is one more MFMA per n tile against a constant ones operand.
// IMPL: x is not the patch matrix but the NHWC image [n][H][W][Ci] (Ci % 16 == 0, so the 16 columns of a k tile are 16 channels of ONE tap);
// row `pix` is the output pixel (n, ho, wo) of the convolution `g` and column (tap, ci) is gathered from input pixel (ho*stride - pad +
// ky*dil, wo*stride - pad + kx*dil), zero outside the image: no patch matrix is written or read (a 3x3 conv's patch matrix is 9 x the
// image; the taps of neighbouring pixels now meet in L2).
template <int NTL, int KTL, bool IMPL>
__global__ __launch_bounds__(256) void direct_wgrad_kernel(const float* __restrict__ dy, int lddy, const float* __restrict__ x, int ldx,
                                                           float* __restrict__ part, float* __restrict__ cpart /* [S][Nn] or null */, long long P,
                                                           int Nn, int K, int kgroups, int slice_rows, ConvGeo g) {
    __shared__ __attribute__((aligned(16))) float red[NTL * KTL * 256 + NTL * 256];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 15, kq = lane >> 4;
    const int kg = blockIdx.x % kgroups, ng = blockIdx.x / kgroups;
    const int n0 = ng * (16 * NTL), k0 = kg * (16 * KTL);
    const long long r_begin = (long long)blockIdx.y * slice_rows;
    const long long r_end = r_begin + slice_rows < P ? r_begin + slice_rows : P;
    const bool want_cs = cpart != nullptr && kg == 0;
    bool nok[NTL], kok[KTL];
#pragma unroll
    for (int i = 0; i < NTL; ++i) nok[i] = n0 + 16 * i + m < Nn;
#pragma unroll
    for (int j = 0; j < KTL; ++j) kok[j] = k0 + 16 * j + m < K;
    f32x4 acc[NTL][KTL], accs[NTL];
#pragma unroll
    for (int i = 0; i < NTL; ++i) {
        accs[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < KTL; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const float* dyp = dy + n0 + m;
    const float* xp = x + k0 + m;
    // IMPL: per k tile the tap's input offset and channel; per lane the output pixel's coordinates, advanced by 16 pixels per iteration
    int tdy[KTL], tdx[KTL], tci[KTL];
    int wo = 0, ho = 0, nimg = 0;
    if (IMPL) {
#pragma unroll
        for (int j = 0; j < KTL; ++j) {
            const int kc = k0 + 16 * j, tap = kc / g.Ci;
            tdy[j] = (tap / g.k) * g.dil - g.pad; tdx[j] = (tap % g.k) * g.dil - g.pad;
            tci[j] = (tdy[j] * g.W + tdx[j]) * g.Ci + kc - tap * g.Ci + m;       // element offset of (tap, channel) from the pixel's own position
            kok[j] = kok[j] && tap < g.k * g.k;
        }
        const long long pix0 = r_begin + 4 * wave + kq;
        wo = (int)(pix0 % g.Wo);
        const long long t = pix0 / g.Wo;
        ho = (int)(t % g.Ho); nimg = (int)(t / g.Ho);
    }
#pragma unroll 2
    for (long long p0 = r_begin + 4 * wave; p0 < r_end; p0 += 16) {
        const long long pix = p0 + kq;
        const bool pok = pix < r_end;
        float a[NTL], b[KTL];
#pragma unroll
        for (int i = 0; i < NTL; ++i) a[i] = (pok && nok[i]) ? dyp[pix * lddy + 16 * i] : 0.f;
        if (IMPL) {
            const int iy0 = ho * g.stride, ix0 = wo * g.stride;
            const float* pb = x + (((long long)nimg * g.H + iy0) * g.W + ix0) * g.Ci;        // one 64-bit address per pixel, 32-bit offsets per tap
#pragma unroll
            for (int j = 0; j < KTL; ++j) {
                const bool ok = pok && kok[j] && (unsigned)(iy0 + tdy[j]) < (unsigned)g.H && (unsigned)(ix0 + tdx[j]) < (unsigned)g.W;
                b[j] = ok ? pb[tci[j]] : 0.f;
            }
            wo += 16;
            while (wo >= g.Wo) { wo -= g.Wo; if (++ho == g.Ho) { ho = 0; ++nimg; } }
        } else
#pragma unroll
        for (int j = 0; j < KTL; ++j) b[j] = (pok && kok[j]) ? xp[pix * ldx + 16 * j] : 0.f;
#pragma unroll
        for (int i = 0; i < NTL; ++i) {
#pragma unroll
            for (int j = 0; j < KTL; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
            if (want_cs) accs[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], 1.0f, accs[i], 0, 0, 0);
        }
    }
    // combine the four waves in a fixed order (wave 0 + 1 + 2 + 3): waves 1..3 park their tiles in LDS one after the other
    for (int w = 1; w < 4; ++w) {
        __syncthreads();
        if (wave == w) {
#pragma unroll
            for (int i = 0; i < NTL; ++i) {
#pragma unroll
                for (int j = 0; j < KTL; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) red[((i * KTL + j) * 4 + r) * 64 + lane] = acc[i][j][r];
#pragma unroll
                for (int r = 0; r < 4; ++r) red[NTL * KTL * 256 + (i * 4 + r) * 64 + lane] = accs[i][r];
            }
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int i = 0; i < NTL; ++i) {
#pragma unroll
                for (int j = 0; j < KTL; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[i][j][r] += red[((i * KTL + j) * 4 + r) * 64 + lane];
#pragma unroll
                for (int r = 0; r < 4; ++r) accs[i][r] += red[NTL * KTL * 256 + (i * 4 + r) * 64 + lane];
            }
        }
    }
    if (wave != 0) return;
    float* out = part + (long long)blockIdx.y * Nn * K;
#pragma unroll
    for (int i = 0; i < NTL; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = n0 + 16 * i + 4 * kq + r;                 // accumulator row = A's row index
            if (n >= Nn) continue;
#pragma unroll
            for (int j = 0; j < KTL; ++j) {
                const int k = k0 + 16 * j + m;
                if (k < K) out[(long long)n * K + k] = acc[i][j][r];
            }
            if (want_cs && m == 0) cpart[(long long)blockIdx.y * Nn + n] = accs[i][r];
        }
}
// out[i] (+)= sum_s part[s][i], fixed order: 256 threads = 32 columns x 8 slab groups (a column's slabs are split over 8 threads,
// four loads in flight each, then combined through LDS) -- with ~2000 slabs a thread per column was a 2000-deep serial chain
// A second, independent sum (the bias gradient's column partials) rides in the same launch: blocks >= nb1 work on (part2, out2, n2).
__global__ __launch_bounds__(256) void slab_sum_kernel(const float* __restrict__ part, float* __restrict__ out, int S, long long n, int accumulate,
                                                       int nb1, const float* __restrict__ part2, float* __restrict__ out2, long long n2) {
    __shared__ float red[8][33];
    const int col = threadIdx.x & 31, grp = threadIdx.x >> 5;
    int blk = blockIdx.x;
    if (blk >= nb1) { blk -= nb1; part = part2; out = out2; n = n2; }
    const long long i = (long long)blk * 32 + col;
    float a = 0.f;
    if (i < n) {
        int j = grp;
        for (; j + 24 < S; j += 32) {
            const float v0 = part[(long long)j * n + i], v1 = part[(long long)(j + 8) * n + i], v2 = part[(long long)(j + 16) * n + i],
                        v3 = part[(long long)(j + 24) * n + i];
            a += v0; a += v1; a += v2; a += v3;
        }
        for (; j < S; j += 8) a += part[(long long)j * n + i];
    }
    red[grp][col] = a;
    __syncthreads();
    if (grp == 0 && i < n) {
        const float t = ((red[0][col] + red[1][col]) + (red[2][col] + red[3][col])) + ((red[4][col] + red[5][col]) + (red[6][col] + red[7][col]));
        out[i] = accumulate ? out[i] + t : t;
    }
}
// column sums over pixel slices: part[slice][c] = sum_{p in slice} v[p][c]   (block = 256 threads = 256/C.. generic: thread per column chunk)
__global__ __launch_bounds__(256) void colsum_slice_kernel(const float* __restrict__ v, int ldv, float* __restrict__ part, long long P, int C,
                                                           int slice_rows) {
    __shared__ float red[256];
    const long long r_begin = (long long)blockIdx.x * slice_rows;
    const long long r_end = r_begin + slice_rows < P ? r_begin + slice_rows : P;
    for (int c0 = 0; c0 < C; c0 += 16) {                       // 16 columns x 16 row lanes per pass
        const int c = c0 + (threadIdx.x & 15), rl = threadIdx.x >> 4;
        float s = 0.f;
        if (c < C)
            for (long long r = r_begin + rl; r < r_end; r += 16) s += v[r * ldv + c];
        red[threadIdx.x] = s;
        __syncthreads();
        if (threadIdx.x < 16 && c0 + threadIdx.x < C) {
            float t = 0.f;
#pragma unroll
            for (int j = 0; j < 16; ++j) t += red[j * 16 + threadIdx.x];
            part[(long long)blockIdx.x * C + c0 + threadIdx.x] = t;
        }
        __syncthreads();
    }
}
// dw [Co][Ci][k][k] <- dWp [Co8 or Co][Kp]
__global__ void conv_unpack_kernel(const float* __restrict__ dwp, float* __restrict__ dw, int Co, int Ci, int kk, int Kp) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Co * Ci * kk) return;
    const int tap = i % kk, ci = (i / kk) % Ci, co = i / (kk * Ci);
    dw[i] = dwp[(long long)co * Kp + tap * Ci + ci];
}

// ---- LayerNorm over C backward + partial affine gradients ------------------------------------------------------------------
template <int CPL>
__global__ __launch_bounds__(256) void ln_c_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ g,
                                                       float* __restrict__ dx, float* __restrict__ part, long long P, int C, int LPP, float eps) {
    __shared__ float red[256 * 2 * CPL];
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long pix = t / LPP;
    const int sub = (int)(t - pix * LPP);
    const bool live = pix < P;
    const long long pc = live ? pix : P - 1;
    float r[CPL], d[CPL];
    float s1 = 0.f;
#pragma unroll
    for (int j = 0; j < CPL; ++j) { r[j] = x[pc * C + sub + j * LPP]; d[j] = live ? dy[pc * C + sub + j * LPP] : 0.f; s1 += r[j]; }
    for (int o = LPP >> 1; o > 0; o >>= 1) s1 += __shfl_xor(s1, o, 64);
    const float mean = s1 / (float)C;
    float s2 = 0.f;
#pragma unroll
    for (int j = 0; j < CPL; ++j) { r[j] -= mean; s2 += r[j] * r[j]; }
    for (int o = LPP >> 1; o > 0; o >>= 1) s2 += __shfl_xor(s2, o, 64);
    const float rstd = rsqrtf(s2 / (float)C + eps);
    float c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int j = 0; j < CPL; ++j) { r[j] *= rstd; const float gd = d[j] * g[sub + j * LPP]; c1 += gd; c2 += gd * r[j]; }
    for (int o = LPP >> 1; o > 0; o >>= 1) { c1 += __shfl_xor(c1, o, 64); c2 += __shfl_xor(c2, o, 64); }
    c1 /= (float)C; c2 /= (float)C;
    if (live)
#pragma unroll
        for (int j = 0; j < CPL; ++j) dx[pix * C + sub + j * LPP] = rstd * (d[j] * g[sub + j * LPP] - c1 - r[j] * c2);
    // per-workgroup partial sums of dgamma = sum dy * xhat and dbeta = sum dy: threads with the same `sub` own the same channels
#pragma unroll
    for (int j = 0; j < CPL; ++j) { red[(2 * j) * 256 + threadIdx.x] = d[j] * r[j]; red[(2 * j + 1) * 256 + threadIdx.x] = d[j]; }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * CPL * LPP; i += 256) {
        const int which = i / LPP, s = i - which * LPP;
        float acc = 0.f;
        for (int u = s; u < 256; u += LPP) acc += red[which * 256 + u];
        const int j = which >> 1, kind = which & 1;              // channel = s + j*LPP
        part[((long long)blockIdx.x * 2 + kind) * C + s + j * LPP] = acc;
    }
}

// ---- q/k/v projection backward, pass 1 (per pixel): dr_j = relu'(.) * LayerNorm_j backward(dq_j); partial per-channel sums of
//      dgamma_j, dbeta_j, dbias_j and the 9 depthwise weight gradients (12 values per channel and projection)
struct ProjBwdArgs { const float* w[3]; const float* b[3]; const float* g[3]; const float* dq[3]; float* dr[3]; };
template <int CPL>
__global__ __launch_bounds__(256) void qkv_proj_bwd1_kernel(const float* __restrict__ x, ProjBwdArgs a, float* __restrict__ part, long long P, int H,
                                                            int W, int C, int LPP, float eps) {
    __shared__ float red[256];
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
    for (int pj = 0; pj < 3; ++pj) {
        float r[CPL], d[CPL];
        float s1 = 0.f;
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
            const int c = sub + j * LPP;
            float acc = a.b[pj] ? a.b[pj][c] : 0.f;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) acc = fmaf(a.w[pj][c * 9 + tap], xin[tap][j], acc);
            r[j] = fmaxf(acc, 0.f);
            d[j] = live ? a.dq[pj][pc * C + c] : 0.f;
            s1 += r[j];
        }
        for (int o = LPP >> 1; o > 0; o >>= 1) s1 += __shfl_xor(s1, o, 64);
        const float mean = s1 / (float)C;
        float s2 = 0.f;
        float xh[CPL];
#pragma unroll
        for (int j = 0; j < CPL; ++j) { xh[j] = r[j] - mean; s2 += xh[j] * xh[j]; }
        for (int o = LPP >> 1; o > 0; o >>= 1) s2 += __shfl_xor(s2, o, 64);
        const float rstd = rsqrtf(s2 / (float)C + eps);
        float c1 = 0.f, c2 = 0.f;
#pragma unroll
        for (int j = 0; j < CPL; ++j) { xh[j] *= rstd; const float gd = d[j] * a.g[pj][sub + j * LPP]; c1 += gd; c2 += gd * xh[j]; }
        for (int o = LPP >> 1; o > 0; o >>= 1) { c1 += __shfl_xor(c1, o, 64); c2 += __shfl_xor(c2, o, 64); }
        c1 /= (float)C; c2 /= (float)C;
        float dr[CPL];
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
            const float v = rstd * (d[j] * a.g[pj][sub + j * LPP] - c1 - xh[j] * c2);
            dr[j] = (live && r[j] > 0.f) ? v : 0.f;
            if (live) a.dr[pj][pix * C + sub + j * LPP] = dr[j];
        }
        // 12 per-channel sums: [0] dgamma, [1] dbeta, [2] dbias, [3..11] dweight taps
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
            for (int v = 0; v < 12; ++v) {
                const float val = v == 0 ? d[j] * xh[j] : v == 1 ? d[j] : v == 2 ? dr[j] : dr[j] * xin[v - 3][j];
                __syncthreads();
                red[threadIdx.x] = val;
                __syncthreads();
                if (threadIdx.x < LPP) {
                    float acc = 0.f;
                    for (int u = threadIdx.x; u < 256; u += LPP) acc += red[u];
                    part[(((long long)blockIdx.x * 3 + pj) * 12 + v) * C + threadIdx.x + j * LPP] = acc;
                }
            }
        }
    }
}
// pass 2: dx[p][c] = sum_j sum_tap w_j[c][tap] * dr_j[p - delta_tap][c]
__global__ __launch_bounds__(256) void qkv_proj_bwd2_kernel(ProjBwdArgs a, float* __restrict__ dx, long long P, int H, int W, int C) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P * C) return;
    const long long pix = i / C;
    const int c = (int)(i - pix * C);
    const int w0 = (int)(pix % W), h0 = (int)((pix / W) % H);
    float s = 0.f;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int hh = h0 - (tap / 3 - 1), ww = w0 - (tap % 3 - 1);
        if (hh >= 0 && hh < H && ww >= 0 && ww < W) {
            const long long q = (pix + (long long)(hh - h0) * W + (ww - w0)) * C + c;
            s += a.w[0][c * 9 + tap] * a.dr[0][q] + a.w[1][c * 9 + tap] * a.dr[1][q] + a.w[2][c * 9 + tap] * a.dr[2][q];
        }
    }
    dx[i] = s;
}
// out[j] = sum_b part[b][j], fixed order (j < n)
__global__ void rows_sum_kernel(const float* __restrict__ part, float* __restrict__ out, int B, int n) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += part[(long long)b * n + j];
    out[j] = s;
}

// ---- resampling / concat / dice / dropout ------------------------------------------------------------------------------------
// mode 0: MaxPool2d(2) backward (x = pool input [N,H,W,C], dy [N,H/2,W/2,C]); mode 2: Upsample x2 backward (dy [N,2H,2W,C] -> dx [N,H,W,C])
__global__ void resample_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ dx, long long total, int H, int W,
                                    int C, int mode) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;        // over dx elements
    if (i >= total) return;
    const int c = (int)(i % C);
    long long r = i / C;
    const int w = (int)(r % W); r /= W;
    const int h = (int)(r % H); const long long n = r / H;
    if (mode == 2) {
        const float* s = dy + ((n * 2 * H + 2 * h) * (2 * W) + 2 * w) * C + c;
        dx[i] = (s[0] + s[C]) + (s[(long long)2 * W * C] + s[(long long)2 * W * C + C]);
    } else {
        const int Ho = H / 2, Wo = W / 2, ho = h / 2, wo = w / 2;
        float v = 0.f;
        if (ho < Ho && wo < Wo) {
            const float* s = x + ((n * H + 2 * ho) * W + 2 * wo) * C + c;
            const float v0 = s[0], v1 = s[C], v2 = s[(long long)W * C], v3 = s[(long long)W * C + C];
            int am = 0; float m = v0;                              // first maximum in torch's window scan order
            if (v1 > m) { m = v1; am = 1; }
            if (v2 > m) { m = v2; am = 2; }
            if (v3 > m) { m = v3; am = 3; }
            if (am == (h & 1) * 2 + (w & 1)) v = dy[((n * Ho + ho) * Wo + wo) * C + c];
        }
        dx[i] = v;
    }
}
__global__ void split_kernel(const float* __restrict__ dy, float* __restrict__ da, int Ca, float* __restrict__ db, int Cb, long long P) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int C = Ca + Cb;
    if (i >= P * C) return;
    const long long pix = i / C;
    const int c = (int)(i - pix * C);
    if (c < Ca) { if (da) da[pix * Ca + c] = dy[i]; } else if (db) db[pix * Cb + c - Ca] = dy[i];
}
// dpred (NCHW, all channels; only channel 0 is non-zero) from the three sums of the forward pass
__global__ void dice_bwd_kernel(const float* __restrict__ tru, const float* __restrict__ sums, const float* __restrict__ dloss, float smooth,
                                float* __restrict__ dpred, int N, int C, long long HW) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)N * C * HW) return;
    const long long r = i % HW;
    const int c = (int)((i / HW) % C);
    float v = 0.f;
    if (c == 0) {
        const float I = sums[0], den = sums[1] + sums[2] + smooth, num = 2.f * I + smooth;
        v = -dloss[0] * (2.f * tru[i] * den - num) / (den * den);
    }
    dpred[i] = v;
    (void)r;
}
__global__ void dropout_kernel(const float* __restrict__ x, float* __restrict__ y, long long n, float p, float inv_keep, unsigned long long seed,
                               const unsigned long long* __restrict__ seed_inc) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned long long s = seed + (seed_inc ? *seed_inc : 0ull);
    y[i] = x[i] * dropout_mult(s, (unsigned long long)i, p, inv_keep);
}

// out[j] = sum_b part[b][j] (j < n): one workgroup per column, 256 threads stride over the B rows, fixed-order tree
__global__ __launch_bounds__(256) void col_reduce_kernel(const float* __restrict__ part, int B, int n, float* __restrict__ out) {
    __shared__ float red[256];
    const int j = blockIdx.x;
    float s = 0.f;
    for (int b = threadIdx.x; b < B; b += 256) s += part[(long long)b * n + j];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[j] = red[0];
}
// tmp [3][12][C] -> the 12 gradient tensors of the three projections
struct ProjGradOut { float* dg[3]; float* dbeta[3]; float* dbias[3]; float* dw[3]; };
__global__ void proj_scatter_kernel(const float* __restrict__ tmp, ProjGradOut o, int C) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 36 * C) return;
    const int c = i % C, v = (i / C) % 12, pj = i / (12 * C);
    const float s = tmp[i];
    if (v == 0) o.dg[pj][c] = s;
    else if (v == 1) o.dbeta[pj][c] = s;
    else if (v == 2) { if (o.dbias[pj]) o.dbias[pj][c] = s; }
    else o.dw[pj][c * 9 + v - 3] = s;
}
__global__ __launch_bounds__(256) void dice_sums_kernel(const float* __restrict__ pred, const float* __restrict__ tru, float* __restrict__ part,
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

template <typename F>
int dispatch_cpl(int C, int& LPP, F&& f) {
    if (C < 1 || C > 512 || (C & (C - 1)) != 0) return HYB_E_ARG;
    LPP = C < 64 ? C : 64;
    return f(C / LPP);
}

}  // namespace

// Internal + used by fct MHA backward: out[Nn][K] (+)= dy^T x over P pixel rows, through SLICE_ROWS-row partial slabs in `ws`
// Slice length of the direct kernel: enough slices that (n groups) x (k groups) x slices is about 2048 workgroups -- the big feature
// maps (800 k pixels) have 8 .. 16 channels, i.e. ONE (n, k) group: with 4096-row slices that was 196 workgroups on 256 CUs.
namespace {
#include "wgrad_lds.h"
}
static void direct_geometry(long long P, int Nn, int K, int& ntl, int& ktl, int& rows, int& S) {
    ntl = Nn <= 16 ? 1 : 2;
    ktl = K <= 32 ? 2 : K <= 64 ? 4 : 8;
    const long long groups = (long long)hyb_cdiv(Nn, 16 * ntl) * hyb_cdiv(K, 16 * ktl);
    long long r = (P * groups + 2047) / 2048;
    r = (r + 15) / 16 * 16;
    if (r < 128) r = 128;
    if (r > SLICE_ROWS) r = SLICE_ROWS;
    rows = (int)r;
    S = (int)((P + r - 1) / r);
}
size_t hyb_sliced_wgrad_workspace(long long P, int Nn, int K) {
    int ntl, ktl, rows, S;
    direct_geometry(P, Nn, K, ntl, ktl, rows, S);
    int S1 = hyb_cdiv(P, SLICE_ROWS);                            // (first-generation kernel, HYB_FCT_WGRAD_V1)
    WlPlan pl;
    if (wl_plan(pl, P, Nn, K, 4, 4, nullptr, nullptr, nullptr) && pl.S > S1) S1 = pl.S;      // (wgrad_lds_kernel's slices)
    return al256((size_t)(S > S1 ? S : S1) * Nn * K * 4);
}
template <int NTL, int KTL>
static void launch_direct_wgrad(const float* dy, int lddy, const float* x, int ldx, float* part, float* cpart, long long P, int Nn, int K, int S,
                                int rows, hipStream_t st, const ConvGeo* geo = nullptr) {
    const int kgroups = hyb_cdiv(K, 16 * KTL), ngroups = hyb_cdiv(Nn, 16 * NTL);
    if (geo) hipLaunchKernelGGL((direct_wgrad_kernel<NTL, KTL, true>), dim3(kgroups * ngroups, S), dim3(256), 0, st, dy, lddy, x, ldx, part, cpart, P, Nn, K,
                                kgroups, rows, *geo);
    else hipLaunchKernelGGL((direct_wgrad_kernel<NTL, KTL, false>), dim3(kgroups * ngroups, S), dim3(256), 0, st, dy, lddy, x, ldx, part, cpart, P, Nn, K,
                            kgroups, rows, ConvGeo{});
}
// out[Nn][K] (+)= dy^T x over P rows; colsum (optional, with its own partial workspace cws of hyb_sliced_colsum_workspace bytes):
// colsum[Nn] (+)= column sums of dy, formed inside the same launch
// geo != NULL: x is the NHWC image of the convolution `geo` (K = k*k*Ci, Ci % 16 == 0), gathered implicitly (direct_wgrad_kernel<.., true>)
int hyb_sliced_wgrad_cs(const float* dy, int lddy, const float* x, int ldx, float* out, float* colsum, long long P, int Nn, int K, int accumulate,
                        void* ws, void* cws, hipStream_t st, const ConvGeo* geo = nullptr) {
    static const int legacy = getenv("HYB_FCT_WGRAD_V1") ? atoi(getenv("HYB_FCT_WGRAD_V1")) : 0;      // first-generation kernel (A/B)
    int ntl, ktl, rows, S;
    direct_geometry(P, Nn, K, ntl, ktl, rows, S);
    float* cpart = colsum ? (float*)cws : nullptr;
    WlPlan pl, cap;
    // (slices shortened for 32-bit addressing may need more slabs than hyb_sliced_wgrad_workspace promised: then the direct kernel runs)
    const int s_cap = wl_plan(cap, P, Nn, K, 4, 4, nullptr, nullptr, nullptr) ? (cap.S > S ? cap.S : S) : S;
    if (wl_plan(pl, P, Nn, K, lddy, ldx, dy, x, geo) && pl.S <= (s_cap > hyb_cdiv(P, SLICE_ROWS) ? s_cap : hyb_cdiv(P, SLICE_ROWS))) {
        S = pl.S;
        wl_launch(pl, dy, lddy, x, ldx, (float*)ws, cpart, P, Nn, K, st, geo);
    } else if (legacy && !geo) {
        S = hyb_cdiv(P, SLICE_ROWS);
        const int tk = hyb_cdiv(K, 64), tn = hyb_cdiv(Nn, 64);
        hipLaunchKernelGGL(sliced_wgrad_kernel, dim3(tk * tn, S), dim3(256), 0, st, dy, lddy, x, ldx, (float*)ws, P, Nn, K, tk, SLICE_ROWS);
        if (colsum) hipLaunchKernelGGL(colsum_slice_kernel, dim3(S), dim3(256), 0, st, dy, lddy, cpart, P, Nn, SLICE_ROWS);
    } else if (ntl == 1) {
        if (ktl == 2) launch_direct_wgrad<1, 2>(dy, lddy, x, ldx, (float*)ws, cpart, P, Nn, K, S, rows, st, geo);
        else if (ktl == 4) launch_direct_wgrad<1, 4>(dy, lddy, x, ldx, (float*)ws, cpart, P, Nn, K, S, rows, st, geo);
        else launch_direct_wgrad<1, 8>(dy, lddy, x, ldx, (float*)ws, cpart, P, Nn, K, S, rows, st, geo);
    } else {
        if (ktl == 2) launch_direct_wgrad<2, 2>(dy, lddy, x, ldx, (float*)ws, cpart, P, Nn, K, S, rows, st, geo);
        else if (ktl == 4) launch_direct_wgrad<2, 4>(dy, lddy, x, ldx, (float*)ws, cpart, P, Nn, K, S, rows, st, geo);
        else launch_direct_wgrad<2, 8>(dy, lddy, x, ldx, (float*)ws, cpart, P, Nn, K, S, rows, st, geo);
    }
    {   // the slab sums of dW and (when wanted) of the bias gradient: one launch
        const int nb1 = hyb_cdiv((long long)Nn * K, 32), nb2 = colsum ? hyb_cdiv(Nn, 32) : 0;
        hipLaunchKernelGGL(slab_sum_kernel, dim3(nb1 + nb2), dim3(256), 0, st, (const float*)ws, out, S, (long long)Nn * K, accumulate, nb1,
                           (const float*)cpart, colsum, (long long)Nn);
    }
    HYB_LAUNCH_CHECK();
    return 0;
}
int hyb_sliced_wgrad(const float* dy, int lddy, const float* x, int ldx, float* out, long long P, int Nn, int K, int accumulate, void* ws,
                     hipStream_t st) {
    return hyb_sliced_wgrad_cs(dy, lddy, x, ldx, out, nullptr, P, Nn, K, accumulate, ws, nullptr, st);
}
// (also the bias-gradient partials of hyb_sliced_wgrad_cs, whose slices are at least 128 rows and at most ~2048 + 1 in number)
size_t hyb_sliced_colsum_workspace(long long P, int C) {
    long long S = hyb_cdiv(P, 128);
    if (S > 2064) S = 2064;
    const long long S1 = hyb_cdiv(P, SLICE_ROWS);
    return al256((size_t)(S > S1 ? S : S1) * C * 4);
}
int hyb_sliced_colsum(const float* v, int ldv, float* out, long long P, int C, int accumulate, void* ws, hipStream_t st) {
    const int S = hyb_cdiv(P, SLICE_ROWS);
    hipLaunchKernelGGL(colsum_slice_kernel, dim3(S), dim3(256), 0, st, v, ldv, (float*)ws, P, C, SLICE_ROWS);
    hipLaunchKernelGGL(slab_sum_kernel, dim3(hyb_cdiv(C, 32)), dim3(256), 0, st, (const float*)ws, out, S, (long long)C, accumulate, hyb_cdiv(C, 32),
                       (const float*)nullptr, (float*)nullptr, 0ll);
    HYB_LAUNCH_CHECK();
    return 0;
}

// ---- general Conv2d backward (conv_geo.h) ----------------------------------------------------------------------------------------
extern "C" size_t hyb_conv2d_bwd_workspace(int N, int H, int W, int Ci, int Co, int k, int stride, int pad, int dilation) {
    ConvGeo g;
    if (N < 1 || Co < 1 || !conv_geo_make(g, H, W, Ci, k, stride, pad, dilation)) return 0;
    const int Kp = g.Kp, Co8 = up8(Co);
    const long long per_img = (long long)g.Ho * g.Wo * (Kp + Co8) * 4;
    long long nb = CHUNK_BYTES / per_img; if (nb < 1) nb = 1; if (nb > N) nb = N;
    const long long Pc = nb * g.Ho * g.Wo;
    return al256((size_t)Kp * Co8 * 4) + al256((size_t)Pc * Co8 * 4) + (conv_geo_identity(g) ? 0 : al256((size_t)Pc * Kp * 4)) +
           al256((size_t)Co8 * Kp * 4) + al256((size_t)Co8 * 4) + hyb_sliced_wgrad_workspace(Pc, Co8, Kp) + hyb_sliced_colsum_workspace(Pc, Co8);
}

extern "C" int hyb_conv2d_bwd(const float* dy, const float* x, const float* w, const float* saved, float* dx, float* dw, float* db, int N, int H,
                              int W, int Ci, int Co, int k, int stride, int pad, int dilation, int act, void* workspace, size_t workspace_bytes,
                              void* stream) {
    ConvGeo g;
    HYB_CHECK_ARG(dy && x && w && dw && workspace && N > 0 && Co > 0 && conv_geo_make(g, H, W, Ci, k, stride, pad, dilation));
    HYB_CHECK_ARG(act >= HYB_ACT_NONE && act <= HYB_ACT_SIGMOID && (act == HYB_ACT_NONE || saved));
    if (workspace_bytes < hyb_conv2d_bwd_workspace(N, H, W, Ci, Co, k, stride, pad, dilation)) return HYB_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int Kp = g.Kp, Co8 = up8(Co), kk = k * k;
    const bool ident = conv_geo_identity(g);
    const bool dz_is_dy = act == HYB_ACT_NONE && Co8 == Co;          // nothing to mask or pad: the GEMMs read dy in place
    const long long per_img = (long long)g.Ho * g.Wo * (Kp + Co8) * 4;
    long long nb = CHUNK_BYTES / per_img; if (nb < 1) nb = 1; if (nb > N) nb = N;
    const long long Pc = nb * g.Ho * g.Wo;
    char* ws = (char*)workspace;
    float* wpt = (float*)ws;   ws += al256((size_t)Kp * Co8 * 4);
    float* dzb = (float*)ws;   ws += al256((size_t)Pc * Co8 * 4);
    float* col = (float*)ws;   ws += ident ? 0 : al256((size_t)Pc * Kp * 4);
    float* dwp = (float*)ws;   ws += al256((size_t)Co8 * Kp * 4);
    float* dbp = (float*)ws;   ws += al256((size_t)Co8 * 4);
    void* ws_w = ws;           ws += hyb_sliced_wgrad_workspace(Pc, Co8, Kp);
    void* ws_c = ws;
    // stride-1 input gradient = a convolution of dz with the flipped kernel (padding dil*(k-1) - pad): the same implicit GEMM as the
    // forward, no dcol matrix and no col2im pass
    static const int implicit_env = getenv("HYB_CONV_IMPLICIT") ? atoi(getenv("HYB_CONV_IMPLICIT")) : 1;
    const bool dgrad_implicit = implicit_env && dx && !ident && stride == 1 && dilation * (k - 1) >= pad && hyb_conv_implicit_ok(Co8, (long long)N * H * W);
    // weight gradient straight from the image (no patch matrix) whenever a k tile of 16 columns stays inside one tap
    static const int wgrad_implicit_env = getenv("HYB_WGRAD_IMPLICIT") ? atoi(getenv("HYB_WGRAD_IMPLICIT")) : 1;
    const bool wgrad_implicit = wgrad_implicit_env && !ident && Ci % 16 == 0 && Kp == kk * Ci;
    // unpadded shapes need no repacking of the results: the slab sums land in dw (1x1: [Co][Ci] is the packed layout) and db directly
    float* dw_dst = (k == 1 && Kp == Ci && Co8 == Co) ? dw : dwp;
    float* db_dst = (Co8 == Co) ? db : dbp;
    if (dx && dgrad_implicit) {
        hipLaunchKernelGGL(conv_pack_flip_kernel, dim3(grid1((long long)Ci * kk * Co8)), dim3(256), 0, st, w, wpt, Co, Ci, k, Co8); HYB_LAUNCH_CHECK();
    } else if (dx) {
        hipLaunchKernelGGL(conv_pack_t_kernel, dim3(grid1((long long)Kp * Co8)), dim3(256), 0, st, w, wpt, Co, Ci, kk, Kp, Co8); HYB_LAUNCH_CHECK();
    }
    for (int n0 = 0, chunk = 0; n0 < N; n0 += (int)nb, ++chunk) {
        const int nn = N - n0 < nb ? N - n0 : (int)nb;
        const long long P = (long long)nn * g.Ho * g.Wo, off = (long long)n0 * g.Ho * g.Wo;
        const long long Pin = (long long)nn * H * W, off_in = (long long)n0 * H * W;
        if (P > 0x7fffffff / 32 * 32 || Pin > 0x7fffffff / 32 * 32) return HYB_E_ARG;
        const float* dz = dz_is_dy ? dy + off * Co : dzb;
        if (!dz_is_dy) {
            hipLaunchKernelGGL(act_bwd_kernel, dim3(grid1(P * Co8)), dim3(256), 0, st, dy + off * Co, saved ? saved + off * Co : nullptr, dzb, P, Co, Co8, act);
            HYB_LAUNCH_CHECK();
        }
        if (dx && dgrad_implicit) {
            FCT_TRY(hyb_conv_implicit_gemm(dz, wpt, nullptr, dx + off_in * Ci, nn, g.Ho, g.Wo, Co8, H, W, Ci, kk * Co8, k, 1, dilation * (k - 1) - pad,
                                           dilation, Ci, 0, st));
        } else if (dx) {
            const void* A[1] = {dz}; const void* B[1] = {wpt}; void* Cc[1] = {ident ? dx + off_in * Ci : col};
            FCT_TRY(hyb_gemm_nt(HYB_F32, 1, A, B, Cc, nullptr, 0, (int)P, Kp, Co8, Co8, Co8, Kp, 0, 0, st));
            if (!ident) { launch_col2im(col, dx + off_in * Ci, Pin, g, st); HYB_LAUNCH_CHECK(); }
        }
        if (wgrad_implicit) {
            FCT_TRY(hyb_sliced_wgrad_cs(dz, Co8, x + off_in * Ci, Ci, dw_dst, db ? db_dst : nullptr, P, Co8, Kp, chunk > 0, ws_w, ws_c, st, &g));
        } else {
            if (!ident) { launch_im2col(x + off_in * Ci, col, P, g, st); HYB_LAUNCH_CHECK(); }
            FCT_TRY(hyb_sliced_wgrad_cs(dz, Co8, ident ? x + off_in * Ci : col, Kp, dw_dst, db ? db_dst : nullptr, P, Co8, Kp, chunk > 0, ws_w, ws_c, st));
        }
    }
    if (dw_dst != dw) hipLaunchKernelGGL(conv_unpack_kernel, dim3(grid1((long long)Co * Ci * kk)), dim3(256), 0, st, (const float*)dwp, dw, Co, Ci, kk, Kp);
    if (db && db_dst != db) { hipError_t e = hipMemcpyAsync(db, dbp, (size_t)Co * 4, hipMemcpyDeviceToDevice, st); if (e != hipSuccess) return (int)e; }
    HYB_LAUNCH_CHECK();
    return 0;
}

extern "C" size_t hyb_fct_conv_bwd_workspace(int N, int H, int W, int Ci, int Co) { return hyb_conv2d_bwd_workspace(N, H, W, Ci, Co, 3, 1, 1, 1); }
extern "C" int hyb_fct_conv_bwd(const float* dy, const float* x, const float* w, const float* saved, float* dx, float* dw, float* db, int N, int H,
                                int W, int Ci, int Co, int dilation, int act, void* workspace, size_t workspace_bytes, void* stream) {
    HYB_CHECK_ARG(dilation >= 1 && dilation <= 8);
    return hyb_conv2d_bwd(dy, x, w, saved, dx, dw, db, N, H, W, Ci, Co, 3, 1, dilation, dilation, act, workspace, workspace_bytes, stream);
}

extern "C" size_t hyb_fct_ln_bwd_workspace(long long P, int C) {
    if (P < 1 || C < 1) return 0;
    const int LPP = C < 64 ? C : 64;
    return al256((size_t)hyb_cdiv(P * LPP, 256) * 2 * C * 4) + al256((size_t)2 * C * 4);
}
extern "C" int hyb_fct_ln_bwd(const float* dy, const float* x, const float* g, float* dx, float* dg, float* db, long long P, int C, float eps,
                              void* workspace, size_t workspace_bytes, void* stream) {
    HYB_CHECK_ARG(dy && x && g && dx && dg && db && workspace && P > 0);
    if (workspace_bytes < hyb_fct_ln_bwd_workspace(P, C)) return HYB_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    float* part = (float*)workspace;
    int LPP = 0, blocks = 0;
    const int rc = dispatch_cpl(C, LPP, [&](int cpl) {
        blocks = grid1(P * LPP);
        const dim3 grid(blocks);
        switch (cpl) {
            case 1: hipLaunchKernelGGL(ln_c_bwd_kernel<1>, grid, dim3(256), 0, st, dy, x, g, dx, part, P, C, LPP, eps); break;
            case 2: hipLaunchKernelGGL(ln_c_bwd_kernel<2>, grid, dim3(256), 0, st, dy, x, g, dx, part, P, C, LPP, eps); break;
            case 4: hipLaunchKernelGGL(ln_c_bwd_kernel<4>, grid, dim3(256), 0, st, dy, x, g, dx, part, P, C, LPP, eps); break;
            case 8: hipLaunchKernelGGL(ln_c_bwd_kernel<8>, grid, dim3(256), 0, st, dy, x, g, dx, part, P, C, LPP, eps); break;
            default: return HYB_E_ARG;
        }
        return 0;
    });
    if (rc) return rc;
    float* tmp = (float*)((char*)workspace + al256((size_t)blocks * 2 * C * 4));        // [2][C]: dgamma | dbeta
    hipLaunchKernelGGL(col_reduce_kernel, dim3(2 * C), dim3(256), 0, st, (const float*)part, blocks, 2 * C, tmp);
    hipError_t e = hipMemcpyAsync(dg, tmp, (size_t)C * 4, hipMemcpyDeviceToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(db, tmp + C, (size_t)C * 4, hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) return (int)e;
    HYB_LAUNCH_CHECK();
    return 0;
}

extern "C" size_t hyb_fct_qkv_proj_bwd_workspace(int N, int H, int W, int C) {
    if (N < 1 || H < 1 || W < 1 || C < 1) return 0;
    const long long P = (long long)N * H * W;
    const int LPP = C < 64 ? C : 64;
    return 3 * al256((size_t)P * C * 4) + al256((size_t)hyb_cdiv(P * LPP, 256) * 36 * C * 4) + al256((size_t)36 * C * 4);
}
extern "C" int hyb_fct_qkv_proj_bwd(const float* x, const float* const* w3, const float* const* b3, const float* const* g3, const float* const* dq3,
                                    float* dx, float* const* dw3, float* const* db3, float* const* dg3, float* const* dbeta3, int N, int H, int W,
                                    int C, float eps, void* workspace, size_t workspace_bytes, void* stream) {
    HYB_CHECK_ARG(x && w3 && b3 && g3 && dq3 && dx && dw3 && db3 && dg3 && dbeta3 && workspace && N > 0 && H > 0 && W > 0);
    if (workspace_bytes < hyb_fct_qkv_proj_bwd_workspace(N, H, W, C)) return HYB_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const long long P = (long long)N * H * W;
    char* ws = (char*)workspace;
    ProjBwdArgs a{};
    ProjGradOut o{};
    for (int j = 0; j < 3; ++j) {
        HYB_CHECK_ARG(w3[j] && g3[j] && dq3[j] && dw3[j] && dg3[j] && dbeta3[j]);
        a.w[j] = w3[j]; a.b[j] = b3[j]; a.g[j] = g3[j]; a.dq[j] = dq3[j];
        a.dr[j] = (float*)ws; ws += al256((size_t)P * C * 4);
        o.dg[j] = dg3[j]; o.dbeta[j] = dbeta3[j]; o.dbias[j] = db3[j]; o.dw[j] = dw3[j];
    }
    float* part = (float*)ws;
    int LPP = 0, blocks = 0;
    const int rc = dispatch_cpl(C, LPP, [&](int cpl) {
        blocks = grid1(P * LPP);
        const dim3 grid(blocks);
        switch (cpl) {
            case 1: hipLaunchKernelGGL(qkv_proj_bwd1_kernel<1>, grid, dim3(256), 0, st, x, a, part, P, H, W, C, LPP, eps); break;
            case 2: hipLaunchKernelGGL(qkv_proj_bwd1_kernel<2>, grid, dim3(256), 0, st, x, a, part, P, H, W, C, LPP, eps); break;
            case 4: hipLaunchKernelGGL(qkv_proj_bwd1_kernel<4>, grid, dim3(256), 0, st, x, a, part, P, H, W, C, LPP, eps); break;
            case 8: hipLaunchKernelGGL(qkv_proj_bwd1_kernel<8>, grid, dim3(256), 0, st, x, a, part, P, H, W, C, LPP, eps); break;
            default: return HYB_E_ARG;
        }
        return 0;
    });
    if (rc) return rc;
    float* tmp = (float*)((char*)part + al256((size_t)blocks * 36 * C * 4));
    hipLaunchKernelGGL(col_reduce_kernel, dim3(36 * C), dim3(256), 0, st, (const float*)part, blocks, 36 * C, tmp);
    hipLaunchKernelGGL(proj_scatter_kernel, dim3(grid1(36 * C)), dim3(256), 0, st, (const float*)tmp, o, C);
    hipLaunchKernelGGL(qkv_proj_bwd2_kernel, dim3(grid1(P * C)), dim3(256), 0, st, a, dx, P, H, W, C);
    HYB_LAUNCH_CHECK();
    return 0;
}

extern "C" int hyb_fct_resample_bwd(int mode, const float* dy, const float* x, float* dx, int N, int H, int W, int C, void* stream) {
    HYB_CHECK_ARG(dy && dx && N > 0 && H > 0 && W > 0 && C > 0 && (mode == 0 || mode == 2) && (mode == 2 || x));
    const long long total = (long long)N * H * W * C;
    hipLaunchKernelGGL(resample_bwd_kernel, dim3(grid1(total)), dim3(256), 0, (hipStream_t)stream, dy, x, dx, total, H, W, C, mode);
    HYB_LAUNCH_CHECK();
    return 0;
}

extern "C" int hyb_fct_concat_bwd(const float* dy, float* da, int Ca, float* db, int Cb, long long P, void* stream) {
    HYB_CHECK_ARG(dy && (da || db) && Ca > 0 && Cb > 0 && P > 0);
    hipLaunchKernelGGL(split_kernel, dim3(grid1(P * (Ca + Cb))), dim3(256), 0, (hipStream_t)stream, dy, da, Ca, db, Cb, P);
    HYB_LAUNCH_CHECK();
    return 0;
}

extern "C" int hyb_dice_bwd(const float* pred, const float* tru, const float* dloss, float* dpred, int N, int C, long long HW, float smooth,
                            void* workspace, size_t workspace_bytes, void* stream) {
    HYB_CHECK_ARG(pred && tru && dloss && dpred && workspace && N > 0 && C > 0 && HW > 0);
    if (workspace_bytes < 256 * 3 * sizeof(float) + 16) return HYB_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    int blocks = hyb_cdiv((long long)N * HW, 256 * 16);
    if (blocks > 255) blocks = 255;
    float* part = (float*)workspace;
    float* sums = part + 255 * 3;
    hipLaunchKernelGGL(dice_sums_kernel, dim3(blocks), dim3(256), 0, st, pred, tru, part, N, C, HW);
    hipLaunchKernelGGL(col_reduce_kernel, dim3(3), dim3(256), 0, st, (const float*)part, blocks, 3, sums);
    hipLaunchKernelGGL(dice_bwd_kernel, dim3(grid1((long long)N * C * HW)), dim3(256), 0, st, tru, (const float*)sums, dloss, smooth, dpred, N, C, HW);
    HYB_LAUNCH_CHECK();
    return 0;
}

/* y = x * mask / (1 - p): nn.Dropout in train mode (FCT.py:115,146,175); the same call with dy gives the backward */
extern "C" int hyb_fct_dropout(const float* x, float* y, long long n, float p, unsigned long long seed, const unsigned long long* seed_inc,
                               void* stream) {
    HYB_CHECK_ARG(x && y && n > 0 && p >= 0.f && p < 1.f);
    hipLaunchKernelGGL(dropout_kernel, dim3(grid1(n)), dim3(256), 0, (hipStream_t)stream, x, y, n, p, 1.f / (1.f - p), seed, seed_inc);
    HYB_LAUNCH_CHECK();
    return 0;
}

// ---- nn.MultiheadAttention backward (FCT.py:37,75) ----------------------------------------------------------------------------
namespace {
// padded packs of the projection weights and their transposes:
//   win [3][Cp][C], winT [3][C][Cp] <- in_proj_weight [3C][C];   wout [C][Cp], woutT [Cp][C] <- out_proj.weight [C][C]
__global__ void mha_pack_bwd_kernel(const float* __restrict__ in_w, const float* __restrict__ out_w, float* __restrict__ winT, float* __restrict__ woutT,
                                    int C, int Hh, int dh, int dhp) {
    const int Cp = Hh * dhp;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int n_in = 3 * C * Cp;
    if (i < n_in) {                                            // winT[j][c][rp] = in_w[j*C + h*dh + f][c]
        const int j = i / (C * Cp), rem = i - j * C * Cp, c = rem / Cp, rp = rem - c * Cp, h = rp / dhp, f = rp - h * dhp;
        winT[i] = f < dh ? in_w[((long long)j * C + h * dh + f) * C + c] : 0.f;
    } else if (i < n_in + Cp * C) {                            // woutT[cp][r] = out_w[r][h*dh + f]
        const int k = i - n_in, cp = k / C, r = k - cp * C, h = cp / dhp, f = cp - h * dhp;
        woutT[k] = f < dh ? out_w[(long long)r * C + h * dh + f] : 0.f;
    }
}
// unpack padded gradients: din_w [3C][C] <- dwin [3][Cp][C]; din_b [3C] <- dbin [3][Cp]; dout_w [C][C] <- dwout [C][Cp]
__global__ void mha_unpack_kernel(const float* __restrict__ dwin, const float* __restrict__ dbin, const float* __restrict__ dwout,
                                  float* __restrict__ din_w, float* __restrict__ din_b, float* __restrict__ dout_w, int C, int Hh, int dh, int dhp) {
    const int Cp = Hh * dhp;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 3 * C * C) {
        const int j = i / (C * C), rem = i - j * C * C, r = rem / C, c = rem - r * C, h = r / dh, f = r - h * dh;
        din_w[i] = dwin[((long long)j * Cp + h * dhp + f) * C + c];
    } else if (i < 3 * C * C + 3 * C) {
        const int k = i - 3 * C * C, j = k / C, r = k - j * C, h = r / dh, f = r - h * dh;
        if (din_b) din_b[k] = dbin[j * Cp + h * dhp + f];
    } else if (i < 3 * C * C + 3 * C + C * C) {
        const int k = i - 3 * C * C - 3 * C, r = k / C, c = k - r * C, h = c / dh, f = c - h * dh;
        dout_w[k] = dwout[(long long)r * Cp + h * dhp + f];
    }
}
}  // namespace

extern "C" size_t hyb_fct_mha_bwd_workspace(int N, int L, int C, int heads) {
    if (N < 1 || L < 1 || C < 1 || heads < 1 || C % heads != 0) return 0;
    const int Cp = heads * up8(C / heads);
    const long long M = (long long)N * L;
    return al256((size_t)3 * C * Cp * 4) + al256((size_t)Cp * C * 4) + 4 * al256((size_t)M * Cp * 4) + al256((size_t)N * heads * L * 4) +
           al256((size_t)3 * Cp * C * 4) + al256((size_t)3 * Cp * 4) + al256((size_t)C * Cp * 4) +
           hyb_sliced_wgrad_workspace(M, Cp > C ? Cp : C, Cp > C ? Cp : C) + hyb_sliced_colsum_workspace(M, Cp > C ? Cp : C);
}

extern "C" int hyb_fct_mha_bwd(const float* dout, const float* q, const float* k, const float* v, const float* in_w, const float* out_w,
                               const void* saved, float* dq, float* dk, float* dv, float* din_w, float* din_b, float* dout_w, float* dout_b,
                               int N, int L, int C, int heads, void* workspace, size_t workspace_bytes, void* stream) {
    HYB_CHECK_ARG(dout && q && k && v && in_w && out_w && saved && dq && dk && dv && din_w && dout_w && workspace);
    HYB_CHECK_ARG(N > 0 && L > 0 && C > 0 && heads > 0 && C % heads == 0 && C % 8 == 0);
    if (workspace_bytes < hyb_fct_mha_bwd_workspace(N, L, C, heads)) return HYB_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int dh = C / heads, dhp = up8(dh), Cp = heads * dhp;
    const long long M = (long long)N * L;
    HYB_CHECK_ARG(M <= 0x7fffffff / 32 * 32);
    const size_t act = al256((size_t)M * Cp * 4);
    const char* sv = (const char*)saved;
    const float* Q = (const float*)sv; const float* K = (const float*)(sv + act); const float* V = (const float*)(sv + 2 * act);
    const float* A = (const float*)(sv + 3 * act); const float* lse = (const float*)(sv + 4 * act);
    char* ws = (char*)workspace;
    float* winT = (float*)ws;   ws += al256((size_t)3 * C * Cp * 4);
    float* woutT = (float*)ws;  ws += al256((size_t)Cp * C * 4);
    float* dA = (float*)ws;     ws += act;
    float* dQ = (float*)ws;     ws += act;
    float* dK = (float*)ws;     ws += act;
    float* dV = (float*)ws;     ws += act;
    float* delta = (float*)ws;  ws += al256((size_t)N * heads * L * 4);
    float* dwin = (float*)ws;   ws += al256((size_t)3 * Cp * C * 4);
    float* dbin = (float*)ws;   ws += al256((size_t)3 * Cp * 4);
    float* dwout = (float*)ws;  ws += al256((size_t)C * Cp * 4);
    const int mx = Cp > C ? Cp : C;
    void* ws_w = ws;            ws += hyb_sliced_wgrad_workspace(M, mx, mx);
    void* ws_c = ws;
    hipLaunchKernelGGL(mha_pack_bwd_kernel, dim3(grid1((long long)3 * C * Cp + Cp * C)), dim3(256), 0, st, in_w, out_w, winT, woutT, C, heads, dh, dhp);
    HYB_LAUNCH_CHECK();
    {   // out-projection: dA = dout Wout (padded), dWout = dout^T A, dbout = column sums of dout
        const void* As[1] = {dout}; const void* Bs[1] = {woutT}; void* Cs[1] = {dA};
        FCT_TRY(hyb_gemm_nt(HYB_F32, 1, As, Bs, Cs, nullptr, 0, (int)M, Cp, C, C, C, Cp, 0, 0, st));
        FCT_TRY(hyb_sliced_wgrad_cs(dout, C, A, Cp, dwout, dout_b, M, C, Cp, 0, ws_w, ws_c, st));
    }
    FCT_TRY(hyb_flash_attention_bwd(HYB_F32, Q, K, V, A, dA, lse, delta, dQ, dK, dV, N, L, heads, dhp, Cp, 1.0f / sqrtf((float)dh), st, dh));
    {   // in-projection: d(inputs) = dQ Win (three GEMMs in one launch), dWin_j = dQ_j^T input_j, dbin_j = column sums
        const void* As[3] = {dQ, dK, dV}; const void* Bs[3] = {winT, winT + (size_t)C * Cp, winT + (size_t)2 * C * Cp}; void* Cs[3] = {dq, dk, dv};
        FCT_TRY(hyb_gemm_nt(HYB_F32, 3, As, Bs, Cs, nullptr, 0, (int)M, C, Cp, Cp, Cp, C, 0, 0, st));
        const float* ins[3] = {q, k, v}; const float* ds[3] = {dQ, dK, dV};
        for (int j = 0; j < 3; ++j) {
            FCT_TRY(hyb_sliced_wgrad_cs(ds[j], Cp, ins[j], C, dwin + (size_t)j * Cp * C, dbin + (size_t)j * Cp, M, Cp, C, 0, ws_w, ws_c, st));
        }
    }
    hipLaunchKernelGGL(mha_unpack_kernel, dim3(grid1((long long)3 * C * C + 3 * C + C * C)), dim3(256), 0, st, (const float*)dwin, (const float*)dbin,
                       (const float*)dwout, din_w, din_b, dout_w, C, heads, dh, dhp);
    HYB_LAUNCH_CHECK();
    return 0;
}
