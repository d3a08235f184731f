// Stage 1 (Conv3x3 -> BatchNorm -> ReLU -> MaxPool on the NCHW fp32 clip frames, UNet.py:58-60 + UNet.py:13) WITHOUT ever
// writing the full-resolution conv output: with Ci <= 4 the conv is K = 9*4 = 36 deep, far cheaper to recompute on the
// matrix cores than to store and re-read (411 MB of bf16 per step at config 2).  One kernel template, three modes:
//   MODE 0  conv -> per-channel sum / sum of squares (BN batch statistics)                      reads x
//   MODE 1  conv -> scale/shift -> ReLU -> 2x2 max (lane shuffles) -> pooled NHWC output          reads x, writes pooled
//   MODE 4  the whole backward in ONE pass over x and dpooled (training and eval): BatchNorm backward is linear in the routed
//           gradient dz (= dpooled at the window's arg-max if the ReLU let it through, else 0), so with the im2col matrix P
//             S1 = sum_pix dz (x) P,   G = sum_pix P (x) P  (P carries a ones column: G[:,36] = sum P, S1[:,36] = sum dz)
//           everything else follows from tiny matrices in a finalize kernel:  sum dz*y = rowdot(W, S1),
//             dW = gamma*inv * [ S1 - m1*SP - m2*inv*(W G - mean*SP) ],  m1 = sum dz / n,  m2 = sum dz*xhat / n.
// A workgroup walks 8x32-pixel tiles.  The fp32 halo (10x34 pixels, prefetched one tile ahead into registers) is converted
// to T and expanded into an im2col matrix P[pixel][k = tap*4 + c] in LDS; P rows feed the conv MFMA (k contiguous) and
// P columns feed the wgrad MFMA through transposed LDS reads, so no scalar gathers are needed.
#include <stdlib.h>
#include "conv_first.h"

namespace {

constexpr int S1_TH = 8, S1_TW = 32, S1_HW = S1_TW + 2, S1_HH = S1_TH + 2, S1_HP = S1_HH * S1_HW;   // halo 10 x 34 = 340 pixels
constexpr int S1_NPIX = S1_TH * S1_TW;         // 256 pixels per tile
constexpr int S1_PS = S1_KP + 8;               // P row stride (elements); S1_KP (padded K) is in conv_first.h
constexpr int S1_MAXPART = 2048;               // most workgroups (= partial statistics rows) of the statistics pass (workspace size)
constexpr int S1_FWD_WGS = 1024;               // its default grid: four workgroups per CU (512 measured 40% slower)
constexpr int S1_BWD_PART = 1024;              // most workgroups (= partial rows) of the one-pass backward and of the Gram pass (workspace size)
constexpr int S1_BWD_WGS = 512;                // default grid of the block-level backward (two workgroups per CU)

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4_s1;


__device__ __forceinline__ void tr_frag_s1(Frag<bf16>& f, const bf16* base, int stride, int second, int lane) {
    const int qq = (lane & 15) >> 2, pp = lane & 3;
    const bf16* a0 = base + qq * stride + 4 * pp;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_s1*)(a0));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_s1*)(a0 + second * stride));
    f.v[0] = lo[0]; f.v[1] = lo[1]; f.v[2] = lo[2]; f.v[3] = lo[3];
    f.v[4] = hi[0]; f.v[5] = hi[1]; f.v[6] = hi[2]; f.v[7] = hi[3];
}
__device__ __forceinline__ void tr_frag_s1(Frag<float>& f, const float* base, int stride, int second, int lane) {
    const float* a0 = base + (lane & 15);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        f.v[j] = a0[j * stride];
        f.v[4 + j] = a0[(second + j) * stride];
    }
}


constexpr int S1_IW = S1_TW + 8;               // image row in LDS: columns tx0-4 .. tx0+35 (aligned float4 segments)
constexpr int S1_IMG = S1_HH * S1_IW;          // 400 pixels x 4 channels

// tile -> (image, tile row, tile col) with float reciprocals: exact while numTiles < 2^20 and the divisors are <= 4096
__device__ __forceinline__ void s1_decode(const S1Args& a, int tile, int& n, int& ty0, int& tx0) {
    n = (int)(((float)tile + 0.5f) * a.inv_tpi);
    const int trem = tile - n * (a.tilesX * a.tilesY);
    const int tr = (int)(((float)trem + 0.5f) * a.inv_tx);
    ty0 = tr * S1_TH;
    tx0 = (trem - tr * a.tilesX) * S1_TW;
}

// 8 K-elements of one pixel's im2col row for k-step 0 (taps 2q, 2q+1) or k-step 1 (tap 8 for q == 0, else zero),
// read straight from the halo image [row][col][4 channels]
__device__ __forceinline__ void s1_bfrag(Frag<bf16>& f, const bf16* img, int base_el, int q, int step) {
    if (step == 0) {
        const int t0 = 2 * q, t1 = 2 * q + 1;
        const Quad<bf16> a = *reinterpret_cast<const Quad<bf16>*>(img + base_el + ((t0 / 3) * S1_IW + t0 % 3) * 4);
        const Quad<bf16> b = *reinterpret_cast<const Quad<bf16>*>(img + base_el + ((t1 / 3) * S1_IW + t1 % 3) * 4);
#pragma unroll
        for (int c = 0; c < 4; ++c) { f.v[c] = a.v[c]; f.v[4 + c] = b.v[c]; }
    } else {
        Quad<bf16> a = *reinterpret_cast<const Quad<bf16>*>(img + base_el + (2 * S1_IW + 2) * 4);
#pragma unroll
        for (int c = 0; c < 4; ++c) { f.v[c] = q == 0 ? a.v[c] : (bf16)0.0f; f.v[4 + c] = (bf16)0.0f; }
    }
}
__device__ __forceinline__ void s1_bfrag(Frag<float>& f, const float* img, int base_el, int q, int step) {
    if (step == 0) {
        const int t0 = 2 * q, t1 = 2 * q + 1;
        const Quad<float> a = *reinterpret_cast<const Quad<float>*>(img + base_el + ((t0 / 3) * S1_IW + t0 % 3) * 4);
        const Quad<float> b = *reinterpret_cast<const Quad<float>*>(img + base_el + ((t1 / 3) * S1_IW + t1 % 3) * 4);
#pragma unroll
        for (int c = 0; c < 4; ++c) { f.v[c] = a.v[c]; f.v[4 + c] = b.v[c]; }
    } else {
        const Quad<float> a = *reinterpret_cast<const Quad<float>*>(img + base_el + (2 * S1_IW + 2) * 4);
#pragma unroll
        for (int c = 0; c < 4; ++c) { f.v[c] = q == 0 ? a.v[c] : 0.f; f.v[4 + c] = 0.f; }
    }
}

// A wave owns one 8x8-pixel block = 16 pooling windows.  MFMA column (lane & 15) = window (wy, wx); MFMA j = window position
// (jy, jx), so a lane holds all four values of its window for NT*4 channels: max / first-argmax are register-local.
template <typename T, int NT, int MODE>
__global__ __launch_bounds__(256) void stage1_kernel(S1Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* img = reinterpret_cast<T*>(smem_raw);                       // [10][40][4]
    T* P = img + S1_IMG * 4;                                       // MODE 3: im2col [256][S1_PS]
    constexpr int DS = NT * 16 + 8;
    T* dyt = P + S1_NPIX * S1_PS;                                  // MODE 3: [256][DS]
    float* wgstat = reinterpret_cast<float*>(MODE == 4 ? (dyt + S1_NPIX * DS) : P);   // [2][NT*16] (mode 0)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int p = lane & 15, q = lane >> 4, wy = p >> 2, wx = p & 3;
    const int co_base = blockIdx.y * (NT * 16);
    const int H = a.H, W = a.W, Ci = a.Ci, Cop = a.Cop;
    const int Ho = H >> 1, Wo = W >> 1;
    const T* wp = (const T*)a.wp;

    if (MODE == 4) {   // zero the im2col padding columns (k >= 36) once; they are never written again (k = 36 is the ones column)
        for (int i = tid; i < S1_NPIX * (S1_PS - 36); i += 256) {
            const int pix = i / (S1_PS - 36), k = 36 + i % (S1_PS - 36);
            P[pix * S1_PS + k] = from_f32<T>(0.f);
        }
    }
    // (mode 0) per-wave slots [4][2][NT*16]: no atomics, fixed-order combine at the end (bit-reproducible)

    Frag<T> w0[NT], w1[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const T* row = wp + (long long)(co_base + (p >> 2) * (NT * 4) + t * 4 + (p & 3)) * S1_KP + 8 * q;
        frag_load(w0[t], row);
        frag_load(w1[t], row + 32);
#pragma unroll
        for (int j = 0; j < 8; ++j) { w0[t].v[j] = hyb_epack(w0[t].v[j]); w1[t].v[j] = hyb_epack(w1[t].v[j]); }
    }
    float c_sc[NT][4], c_sh[NT][4];
    if (MODE >= 1) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ch = co_base + q * (NT * 4) + t * 4 + r;
                c_sc[t][r] = a.ss[ch]; c_sh[t][r] = a.ss[Cop + ch];
            }
    }
    float acc1[NT][4], acc2[NT][4];           // MODE 0: sum / sum of squares (per lane, all tiles)
    if (MODE == 0) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) { acc1[t][r] = 0.f; acc2[t][r] = 0.f; }
    }
    f32x4 wacc[3][NT];                        // MODE 4: S1 [co tile][k tile] partials of this wave
    f32x4 gacc[6];                            // MODE 4: upper triangle of the Gram matrix G[k tile][k tile]
    if (MODE == 4) {
#pragma unroll
        for (int i = 0; i < 6; ++i) gacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (MODE == 4) {
#pragma unroll
        for (int kt = 0; kt < 3; ++kt)
#pragma unroll
            for (int c = 0; c < NT; ++c) wacc[kt][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    // halo prefetch: thread u < 100 owns (row = u / 10, segment = u % 10): 4 consecutive pixels x up to 4 channels
    f32x4 pf[4];
    auto prefetch = [&](int tile) {
        if (tid >= 100) return;
        int n, ty0, tx0;
        s1_decode(a, tile, n, ty0, tx0);
        const int row = tid / 10, seg = tid - row * 10;
        const int gy = ty0 + row - 1, gx0 = tx0 - 4 + 4 * seg;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
            if (c < Ci && gy >= 0 && gy < H) {
                const float* src = a.x + ((long long)(n * Ci + c) * H + gy) * W + gx0;
                if (a.vec_ok) {
                    if (gx0 >= 0 && gx0 + 3 < W) v = *reinterpret_cast<const f32x4*>(src);
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (gx0 + i >= 0 && gx0 + i < W) v[i] = src[i];
                }
            }
            pf[c] = v;
        }
    };
    // MODE == 4: this lane's dpooled values (window (wy, wx) of the wave's 8x8 block) are prefetched one tile ahead too
    Vec8<T> gpf[NT / 2];
    auto prefetch_dp = [&](int tile) {
        int n, ty0, tx0;
        s1_decode(a, tile, n, ty0, tx0);
        const int oy = (ty0 + 2 * wy) >> 1, ox = (tx0 + wave * 8 + 2 * wx) >> 1;
        const T* dsrc = (const T*)a.dp + ((long long)(n * Ho + oy) * Wo + ox) * Cop + co_base + q * (NT * 4);
#pragma unroll
        for (int h8 = 0; h8 < NT / 2; ++h8) {
            if (oy < Ho && ox < Wo) gpf[h8].load(dsrc + h8 * 8); else gpf[h8].zero();
        }
    };
    // a workgroup walks a CONTIGUOUS run of tiles (row-major inside an image): the halo rows/columns that neighbouring tiles share
    // are then re-read from this XCD's L2 instead of being fetched once per XCD
    const int chunk = (a.numTiles + (int)gridDim.x - 1) / (int)gridDim.x;
    const int tile_begin = blockIdx.x * chunk;
    const int tile_end = tile_begin + chunk < a.numTiles ? tile_begin + chunk : a.numTiles;
    if (tile_begin < tile_end) { prefetch(tile_begin); if (MODE == 4) prefetch_dp(tile_begin); }

    for (int tile = tile_begin; tile < tile_end; ++tile) {
        int n, ty0, tx0;
        s1_decode(a, tile, n, ty0, tx0);
        __syncthreads();                                   // previous tile done with img / P / dyt
        if (tid < 100) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                Quad<T> qv;
#pragma unroll
                for (int c = 0; c < 4; ++c) qv.v[c] = hyb_epack(from_f32<T>(pf[c][i]));      // (split-bf16 build: element-packed LDS, hyb_common.h)
                *reinterpret_cast<Quad<T>*>(img + (tid * 4 + i) * 4) = qv;       // pixel (row, 4*seg + i) = tid*4 + i
            }
        }
        __syncthreads();
        Vec8<T> gcur[NT / 2];
        if (MODE == 4) {
#pragma unroll
            for (int h8 = 0; h8 < NT / 2; ++h8) gcur[h8] = gpf[h8];
        }
        if (tile + 1 < tile_end) {                                               // in flight under this tile's work
            prefetch(tile + 1);
            if (MODE == 4) prefetch_dp(tile + 1);
        }
        if (MODE == 4) {   // im2col (pixel tid, 9 taps x 4 channels) for the pixel contractions; rows of pixels outside the image are zero
                           // (they must not enter G) and column 36 flags validity
            const int ty = tid >> 5, tx = tid & 31;
            const bool pvalid = (ty0 + ty < H) && (tx0 + tx < W);
            Quad<T> zq;
#pragma unroll
            for (int c = 0; c < 4; ++c) zq.v[c] = from_f32<T>(0.f);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const Quad<T> v = *reinterpret_cast<const Quad<T>*>(img + ((ty + tap / 3) * S1_IW + tx + 3 + tap % 3) * 4);
                *reinterpret_cast<Quad<T>*>(P + tid * S1_PS + tap * 4) = pvalid ? v : zq;
            }
            Quad<T> one = zq;
            one.v[0] = from_f32<T>(pvalid ? 1.f : 0.f);
            *reinterpret_cast<Quad<T>*>(P + tid * S1_PS + 36) = one;
        }

        // ---- conv of this wave's 8x8 block: 4 window positions x NT channel tiles
        f32x4 acc[4][NT];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ly = 2 * wy + (j >> 1), lx = wave * 8 + 2 * wx + (j & 1);
            const int base_el = (ly * S1_IW + lx + 3) * 4;                     // tap (0,0) of this pixel
            Frag<T> b0, b1;
            s1_bfrag(b0, img, base_el, q, 0);
            s1_bfrag(b1, img, base_el, q, 1);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                acc[j][t] = mma32_e(w0[t], b0, f32x4{0.f, 0.f, 0.f, 0.f});
                acc[j][t] = mma32_e(w1[t], b1, acc[j][t]);
            }
        }
        const int gy0 = ty0 + 2 * wy, gx0 = tx0 + wave * 8 + 2 * wx;          // window origin
        const int oy = gy0 >> 1, ox = gx0 >> 1;
        const bool win_ok = oy < Ho && ox < Wo;

        if (MODE == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool valid = (gy0 + (j >> 1)) < H && (gx0 + (j & 1)) < W;
                if (valid) {
#pragma unroll
                    for (int t = 0; t < NT; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r) { acc1[t][r] += acc[j][t][r]; acc2[t][r] = fmaf(acc[j][t][r], acc[j][t][r], acc2[t][r]); }
                }
            }
        }
        if (MODE == 1) {
            if (win_ok) {
                T* dst = (T*)a.pooled + ((long long)(n * Ho + oy) * Wo + ox) * Cop + co_base + q * (NT * 4);
#pragma unroll
                for (int h8 = 0; h8 < NT / 2; ++h8) {
                    Vec8<T> o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int t = h8 * 2 + (e >> 2), r = e & 3;
                        const float sc = c_sc[t][r], sh = c_sh[t][r];
                        const float m = fmaxf(fmaxf(acc[0][t][r] * sc + sh, acc[1][t][r] * sc + sh),
                                              fmaxf(acc[2][t][r] * sc + sh, acc[3][t][r] * sc + sh));
                        o.set(e, fmaxf(m, 0.f));
                    }
                    o.store(dst + h8 * 8);
                }
            }
        }
        if (MODE == 4) {
#pragma unroll
            for (int h8 = 0; h8 < NT / 2; ++h8) {
                const Vec8<T> g = gcur[h8];
                Vec8<T> o[4];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int t = h8 * 2 + (e >> 2), r = e & 3;
                    const float sc = c_sc[t][r], sh = c_sh[t][r];
                    // first maximum in torch's window scan order (0,0),(0,1),(1,0),(1,1); ReLU gate on the maximum
                    float vmax = acc[0][t][r] * sc + sh;
                    int am = 0;
#pragma unroll
                    for (int j = 1; j < 4; ++j) {
                        const float v = acc[j][t][r] * sc + sh;
                        if (v > vmax) { vmax = v; am = j; }
                    }
                    const float dy = (vmax > 0.f && win_ok) ? g.get(e) : 0.f;
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j].set(e, am == j ? dy : 0.f);          // routed gradient dz (win_ok implies the pixel is valid)
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int pix = (2 * wy + (j >> 1)) * S1_TW + wave * 8 + 2 * wx + (j & 1);
                    o[j].epack();
                    o[j].store(dyt + pix * DS + q * (NT * 4) + h8 * 8);
                }
            }
            __syncthreads();
            // S1[co][k] (and G) += sum over this wave's 2 tile rows (32 pixels = one k32 step each)
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                const int row = wave * 2 + rr;
                Frag<T> af[NT];
#pragma unroll
                for (int c = 0; c < NT; ++c) tr_frag_s1(af[c], dyt + (row * S1_TW + 4 * q) * DS + c * 16, DS, 16, lane);
                Frag<T> bf[3];
#pragma unroll
                for (int kt = 0; kt < 3; ++kt) {
                    tr_frag_s1(bf[kt], P + (row * S1_TW + 4 * q) * S1_PS + kt * 16, S1_PS, 16, lane);
#pragma unroll
                    for (int c = 0; c < NT; ++c) wacc[kt][c] = mma32_e(af[c], bf[kt], wacc[kt][c]);
                }
                if (MODE == 4 && blockIdx.y == 0) {       // Gram matrix of the patches: A and B fragments have the same lane layout
                    gacc[0] = mma32_e(bf[0], bf[0], gacc[0]); gacc[1] = mma32_e(bf[0], bf[1], gacc[1]); gacc[2] = mma32_e(bf[0], bf[2], gacc[2]);
                    gacc[3] = mma32_e(bf[1], bf[1], gacc[3]); gacc[4] = mma32_e(bf[1], bf[2], gacc[4]); gacc[5] = mma32_e(bf[2], bf[2], gacc[5]);
                }
            }
        }
    }

    if (MODE == 0) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float s1 = group16_sum(acc1[t][r]), s2 = group16_sum(acc2[t][r]);
                if (p == 0) {
                    wgstat[(wave * 2 + 0) * (NT * 16) + q * (NT * 4) + t * 4 + r] = s1;
                    wgstat[(wave * 2 + 1) * (NT * 16) + q * (NT * 4) + t * 4 + r] = s2;
                }
            }
        __syncthreads();
        for (int i = tid; i < 2 * NT * 16; i += 256) {
            const int which = i / (NT * 16), cl = i % (NT * 16);
            const float v = (wgstat[(0 * 2 + which) * (NT * 16) + cl] + wgstat[(1 * 2 + which) * (NT * 16) + cl]) +
                            (wgstat[(2 * 2 + which) * (NT * 16) + cl] + wgstat[(3 * 2 + which) * (NT * 16) + cl]);
            a.part[((long long)blockIdx.x * 2 + which) * Cop + co_base + cl] = v;
        }
    }
    if (MODE == 4) {
        // combine the 4 waves through LDS (reusing P/dyt): row = [S1: Cop x 48][G: 48 x 48 (upper triangle tiles filled)]
        __syncthreads();
        float* red = reinterpret_cast<float*>(P);                  // [4][NT*16*48 + 2304]
        constexpr int RW = NT * 16 * 48 + 2304;
#pragma unroll
        for (int kt = 0; kt < 3; ++kt)
#pragma unroll
            for (int c = 0; c < NT; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[wave * RW + (c * 16 + 4 * q + r) * 48 + kt * 16 + p] = wacc[kt][c][r];
        auto put_g = [&](int i, int gi, int gj) {
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wave * RW + NT * 16 * 48 + (gi * 16 + 4 * q + r) * 48 + gj * 16 + p] = gacc[i][r];
        };
        put_g(0, 0, 0); put_g(1, 0, 1); put_g(2, 0, 2); put_g(3, 1, 1); put_g(4, 1, 2); put_g(5, 2, 2);
        __syncthreads();
        const long long roww = (long long)Cop * 48 + 2304;
        float* out = a.part + (long long)blockIdx.x * roww;
        for (int i = tid; i < NT * 16 * 48; i += 256)
            out[(long long)co_base * 48 + i] = (red[i] + red[RW + i]) + (red[2 * RW + i] + red[3 * RW + i]);
        if (blockIdx.y == 0) {
            for (int i = tid; i < 2304; i += 256) {
                const int gr = i / 48, gc = i % 48;
                const int o = NT * 16 * 48 + i;
                out[(long long)Cop * 48 + i] = (gr / 16 <= gc / 16) ? (red[o] + red[RW + o]) + (red[2 * RW + o] + red[3 * RW + o]) : 0.f;
            }
        }
    }
}

// MODE 4 finalize.  red = [S1: Cop x 48][G: 48 x 48, only tiles (i <= j) valid]; one thread per (co, k).
// The finalize subtracts sums over all N*H*W positions that nearly cancel (S1 - m1*sum(P) is n*cov(dz, P) left over from terms of size
// n*mean(dz)*mean(P)), so the workgroup partial rows are summed in DOUBLE and the tiny-matrix algebra below runs in double:
// 36*Co threads, free.  (The fp32 partial rows themselves carry ~1e-6 relative error each, uncorrelated across the 512 of them.)
template <typename T>
__global__ __launch_bounds__(256) void s1_bwd_finalize_kernel(const double* __restrict__ red, const double* __restrict__ G, const T* __restrict__ wp,
                                                              const float* __restrict__ mi,
                                                              const float* __restrict__ gamma, int training, double inv_count, int Co, int Ci,
                                                              int Cop, float* __restrict__ dw, float* __restrict__ dgamma,
                                                              float* __restrict__ dbeta) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Co * 36) return;
    const int co = i / 36, k = i - co * 36;
    const double* S1 = red + (long long)co * 48;
    auto g_at = [&](int r, int c) { return (r / 16 <= c / 16) ? G[r * 48 + c] : G[c * 48 + r]; };      // symmetric
    const double mean = mi[co], inv = mi[Cop + co];
    const double sdz = S1[36];
    double sdzy = 0.0, wg = 0.0;
    for (int kk = 0; kk < 36; ++kk) {
        const double w = (double)to_f32<T>(wp[(long long)co * 64 + kk]);
        sdzy = fma(w, S1[kk], sdzy);
        wg = fma(w, g_at(kk, k), wg);
    }
    const double sdzx = (sdzy - mean * sdz) * inv;             // sum dz * xhat
    const double m1 = training ? sdz * inv_count : 0.0, m2 = training ? sdzx * inv_count : 0.0;
    const double sp = g_at(k, 36);
    const double v = (double)gamma[co] * inv * (S1[k] - m1 * sp - m2 * inv * (wg - mean * sp));
    const int tap = k >> 2, ci = k & 3;
    if (ci < Ci) dw[((long long)co * Ci + ci) * 9 + tap] = (float)v;
    if (k == 0) {
        if (dbeta) dbeta[co] = (float)sdz;
        if (dgamma) dgamma[co] = (float)sdzx;
    }
}

// BatchNorm batch statistics of stage 1 from the Gram matrix of the patches (stage1w_gram_kernel): one thread per channel, in double.
//   sum y = w . G[:,36],  sum y^2 = w^T G w   (w = the storage-rounded weights the conv kernels multiply with)
// followed by what bn_stats_finalize_kernel does (UNet.py:59 semantics: biased variance for the normalisation, unbiased for the running
// estimate, momentum update, num_batches_tracked).
template <typename T>
__global__ __launch_bounds__(64) void s1_gram_stats_kernel(double* __restrict__ G /* [2304] read, [2304..2305] padding zeroed */, const T* __restrict__ wp, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float* running_mean, float* running_var,
                                                           long long* __restrict__ nbt, float momentum, float eps, long long count, int Co, int Cop,
                                                           float* __restrict__ scale_shift, float* __restrict__ mean_invstd, float* running_out) {
    // one wave per channel: lane k < 36 forms row k of G w (36 independent loads), the two sums are fixed-order wave reductions
    __shared__ double wl[36];
    const int c = blockIdx.x, k = threadIdx.x;
    if (c == 0 && k == 0 && nbt && !running_out) *nbt += 1;
    if (c == 0 && k < S1_GRAM_DOUBLES - 2304) G[2304 + k] = 0.0;
    auto g_at = [&](int r, int cc) { return (r / 16 <= cc / 16) ? G[r * 48 + cc] : G[cc * 48 + r]; };
    if (k < 36) wl[k] = c < Co ? (double)to_f32<T>(wp[(long long)c * 64 + k]) : 0.0;
    __syncthreads();
    double s1 = 0.0, s2 = 0.0;
    if (k < 36 && c < Co) {
        double row = 0.0;
#pragma unroll 6
        for (int kk = 0; kk < 36; ++kk) row = fma(wl[kk], g_at(k, kk), row);
        s2 = wl[k] * row;
        s1 = wl[k] * g_at(k, 36);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
    if (k != 0) return;
    float mean = 0.f, invstd = 0.f, g = 0.f, b = 0.f;
    if (c < Co) {
        const double inv_n = 1.0 / (double)count;
        const double m = s1 * inv_n;
        double var = s2 * inv_n - m * m;
        if (var < 0.0) var = 0.0;
        g = gamma[c]; b = beta[c];
        mean = (float)m;
        invstd = (float)(1.0 / sqrt(var + (double)eps));
        const float unbiased = (float)(count > 1 ? var * ((double)count / (double)(count - 1)) : var);
        (running_out ? running_out : running_mean)[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
        (running_out ? running_out + Co : running_var)[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
    }
    const float scale = g * invstd;
    scale_shift[c] = scale;
    scale_shift[Cop + c] = b - mean * scale;
    mean_invstd[c] = mean;
    mean_invstd[Cop + c] = invstd;
}

// fixed-order DOUBLE sum of the MODE 4 partial rows: column i = sum_g part[g*n + i]; 256 threads = 32 columns x 8 row groups
__global__ __launch_bounds__(256) void s1_rows_sum_kernel(const float* __restrict__ part, double* __restrict__ out, int G, long long n) {
    __shared__ double red[8][33];
    const int col = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const long long i = (long long)blockIdx.x * 32 + col;
    double a = 0.0;
    if (i < n) {
        int g = grp;
        for (; g + 7 * 8 < G; g += 64) {                      // eight independent loads in flight (pure latency kernel), fixed add order
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = part[(long long)(g + j * 8) * n + i];
#pragma unroll
            for (int j = 0; j < 8; ++j) a += (double)v[j];
        }
        for (; g < G; g += 8) a += (double)part[(long long)g * n + i];
    }
    red[grp][col] = a;
    __syncthreads();
    if (grp == 0 && i < n) out[i] = ((red[0][col] + red[1][col]) + (red[2][col] + red[3][col])) + ((red[4][col] + red[5][col]) + (red[6][col] + red[7][col]));
}

// packed first-layer weights for this path: T [Cop][64], k = tap*4 + c
template <typename T>
__global__ void s1_pack_kernel(const float* __restrict__ w, T* __restrict__ wp, int Co, int Ci, long long total) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int k = (int)(i % 64), co = (int)(i / 64);
    const int tap = k >> 2, c = k & 3;
    float v = 0.f;
    if (co < Co && tap < 9 && c < Ci) v = w[((long long)co * Ci + c) * 9 + tap];
    wp[i] = from_f32<T>(v);
}

template <typename T, int NT, int MODE>
size_t s1_lds_bytes() {
    size_t el = (size_t)S1_IMG * 4;
    if (MODE == 4) el += (size_t)S1_NPIX * S1_PS + (size_t)S1_NPIX * (NT * 16 + 8);
    size_t bytes = el * sizeof(T) + 4 * 2 * NT * 16 * sizeof(float) + 64;
    if (MODE == 4) {                                       // the end-of-kernel combine reuses P/dyt as [4][NT*16*48 + 2304] floats
        const size_t need = (size_t)S1_IMG * 4 * sizeof(T) + (size_t)4 * (NT * 16 * 48 + 2304) * sizeof(float);
        if (need > bytes) bytes = need;
    }
    return bytes;
}

template <typename T, int NT, int MODE>
int s1_launch(S1Args a, int grid_x, hipStream_t st) {
    const size_t lds = s1_lds_bytes<T, NT, MODE>();
    if (lds > 64 * 1024) {
        static HybAttrOnce once;                               // per template instantiation (lds is a compile-time size), per device
        if (int e = hyb_set_lds_attr(once, (const void*)stage1_kernel<T, NT, MODE>, (int)lds)) return e;
    }
    dim3 grid(grid_x, a.Cop / (NT * 16));
    hipLaunchKernelGGL((stage1_kernel<T, NT, MODE>), grid, dim3(256), lds, st, a);
    HYB_LAUNCH_CHECK();
    return 0;
}

template <typename T, int MODE>
int s1_dispatch(const S1Args& a, int grid_x, hipStream_t st) {
    if (a.Cop % 64 == 0) return s1_launch<T, 4, MODE>(a, grid_x, st);
    return s1_launch<T, 2, MODE>(a, grid_x, st);
}

inline size_t al256(size_t v) { return (v + 255) & ~(size_t)255; }

}  // namespace

// ---- internal entry points used by hyb_convstage_{fwd,bwd} when first = 1 ----------------------------------------------
size_t hyb_stage1_fwd_workspace(int dtype, int Cop) {
    const size_t es = dtype == HYB_F32 ? 4 : 2;
    size_t part = (size_t)S1_MAXPART * 2 * Cop * 4;
    if (part < (size_t)S1_BWD_PART * 2304 * 4) part = (size_t)S1_BWD_PART * 2304 * 4;            // Gram partial rows (statistics from G)
    return 2 * al256((size_t)Cop * 64 * es) + al256(2 * (size_t)Cop * 4) + al256(part) + al256((size_t)S1_GRAM_DOUBLES * 8);
}
size_t hyb_stage1_bwd_workspace(int dtype, int Cop) {
    const size_t es = dtype == HYB_F32 ? 4 : 2;
    return 2 * al256((size_t)Cop * 64 * es) + al256(((size_t)Cop * 48 + 2304) * 8) + al256((size_t)S1_BWD_PART * ((size_t)Cop * 48 + 2304) * 4);
}

// Routing codes (S1Args::route): the forward's wave-private apply pass writes them, the backward's wave-private pass reads them instead of
// recomputing the conv.  ONE rule for both sides (same frames pointer, same shape, same switches), so a backward never reads codes no
// forward wrote: 16-bit storage, no ragged 8x16 blocks, aligned float4 rows, 32-bit buffer offsets, both wave-private generations enabled.
static bool s1_route_ok(size_t es, const float* x, int N, int H, int W, int Ci, int Cop) {
    static const int fwd_env = getenv("HYB_S1_WAVE") ? atoi(getenv("HYB_S1_WAVE")) : 1;
    static const int bwd_env = getenv("HYB_S1_WAVE_BWD") ? atoi(getenv("HYB_S1_WAVE_BWD")) : 1;
    static const int route_env = getenv("HYB_S1_ROUTE") ? atoi(getenv("HYB_S1_ROUTE")) : 1;          // (=0: A/B, the backward recomputes the conv)
    return route_env && fwd_env && bwd_env && es == 2 && !HYB_X3 && (W % 16 == 0) && (H % 8 == 0) && (((uintptr_t)x & 15) == 0) && (Cop % 8 == 0) &&
           (long long)N * Ci * H * W * 4 < (1ll << 32) && (long long)N * (H / 2) * (W / 2) * Cop * (long long)es < (1ll << 32);
}
// elements of type T the caller provides for the codes: 4 bits per pooled element
long long hyb_stage1_route_elems(int dtype, int N, int H, int W, int Cop) {
    if (dtype != HYB_BF16 || HYB_X3 || W % 16 != 0 || H % 8 != 0 || Cop % 8 != 0 || N < 1) return 0;
    return (long long)N * (H / 2) * (W / 2) * Cop / 4;          // bytes / 2
}

static int s1_grid(long long numTiles) {
    static const int fwd_wgs = getenv("HYB_S1_FWD_WGS") ? atoi(getenv("HYB_S1_FWD_WGS")) : S1_FWD_WGS;
    long long g = numTiles < fwd_wgs ? numTiles : fwd_wgs;
    if (g > S1_MAXPART) g = S1_MAXPART;
    return (int)(g < 1 ? 1 : g);
}

template <typename T>
static int stage1_fwd_t(int dtype, const float* x, const float* weight, const float* gamma, const float* beta, float* running_mean,
                        float* running_var, long long* nbt, int training, float momentum, float eps, int N, int H, int W, int Ci, int Co,
                        int Cop, void* pooled, float* scale_shift, float* mean_invstd, void* packed_out, void* workspace, float* running_out,
                        int prepacked, void* route, hipStream_t st) {
    const size_t es = sizeof(T);
    char* ws = (char*)workspace;
    // packed weights in both K orders; kept for backward when asked (packed_out = [2][Cop][64])
    T* wp = packed_out ? (T*)packed_out : (T*)ws;                       ws += al256((size_t)Cop * 64 * es);
    T* wp2 = packed_out ? (T*)packed_out + (size_t)Cop * 64 : (T*)ws;   ws += al256((size_t)Cop * 64 * es);
    float* stats = (float*)ws;                   ws += al256(2 * (size_t)Cop * 4);
    float* part = (float*)ws;
    {
        size_t pb = (size_t)S1_MAXPART * 2 * Cop * 4;
        if (pb < (size_t)S1_BWD_PART * 2304 * 4) pb = (size_t)S1_BWD_PART * 2304 * 4;
        ws += al256(pb);
    }
    // Gram matrix (double): kept for backward behind the packed weights when asked, else scratch
    double* gram = packed_out ? (double*)((T*)packed_out + (size_t)Cop * 128) : (double*)ws;
    const long long total = (long long)Cop * 64;
    static const int wave_env = getenv("HYB_S1_WAVE") ? atoi(getenv("HYB_S1_WAVE")) : 1;          // second-generation forward passes (A/B)
    // they address x and pooled with 32-bit buffer offsets and load aligned float4 row segments
    const bool vec_ok = (W % 4 == 0) && (((uintptr_t)x & 15) == 0);
    const bool wave_private = wave_env && vec_ok && (long long)N * Ci * H * W * 4 < (1ll << 32) &&
                              (long long)N * (H / 2) * (W / 2) * Cop * (long long)es < (1ll << 32);
    // statistics from the Gram matrix: 16-bit storage, training mode, no ragged 8x16 blocks (same test as the backward's wave-private kernel)
    static const int gram_env = getenv("HYB_S1_GRAM") ? atoi(getenv("HYB_S1_GRAM")) : 1;
    const bool use_gram = gram_env && wave_private && training && sizeof(T) == 2 && (W % 16 == 0) && (H % 8 == 0);
    const bool need_a = packed_out || !wave_private || use_gram;     // k = tap*4 + c: the block-level kernels, the Gram statistics, the backward pass
    const bool need_b = wave_private || packed_out;
    if (prepacked && packed_out) {
        // the caller (hyb_backbone_fwd) has written both layouts into packed_out already, in its one pack launch
    } else if (need_a && need_b && wp2 == wp + (size_t)Cop * 64) {          // the two layouts are adjacent (always: Cop*64*es is a multiple of 256): one launch
        if (int e = hyb_stage1w_pack(dtype, weight, wp, Co, Ci, Cop, 1, st)) return e;
    } else {
        if (need_a) {
            hipLaunchKernelGGL(s1_pack_kernel<T>, dim3(hyb_cdiv(total, 256)), dim3(256), 0, st, weight, wp, Co, Ci, total);
            HYB_LAUNCH_CHECK();
        }
        if (need_b) { if (int e = hyb_stage1w_pack(dtype, weight, wp2, Co, Ci, Cop, 0, st)) return e; }
    }
    S1Args a{};
    a.x = x; a.wp = wp; a.wp2 = wp2; a.ss = scale_shift; a.mi = mean_invstd; a.gamma = gamma; a.pooled = pooled; a.part = part;
    a.N = N; a.H = H; a.W = W; a.Ci = Ci; a.Co = Co; a.Cop = Cop; a.training = training;
    a.tilesX = hyb_cdiv(W, S1_TW); a.tilesY = hyb_cdiv(H, S1_TH);
    a.inv_tpi = 1.0f / (float)(a.tilesX * a.tilesY); a.inv_tx = 1.0f / (float)a.tilesX;
    a.vec_ok = (W % 4 == 0) && (((uintptr_t)x & 15) == 0);
    a.route = (route && wave_private && s1_route_ok(es, x, N, H, W, Ci, Cop)) ? (unsigned*)route : nullptr;
    const long long numTiles = (long long)N * a.tilesX * a.tilesY;
    a.numTiles = (int)numTiles;
    const int gx = s1_grid(numTiles);
    int rc;
    int gx_rows = gx;                                      // partial statistics rows actually written
    if (packed_out) {
        // defined contents for the whole saved buffer (its size is counted in 2-byte elements: twice the need with fp32 storage): the
        // Gram slot's padding always, the slot itself when this path does not produce it
        // (when it is produced -- 16-bit storage -- the statistics kernel below zeroes the 16 bytes of padding itself: no extra launch)
        const size_t tail = (size_t)S1_GRAM_DOUBLES * 4 * es;
        if (!(training && use_gram)) { if (hipError_t e = hipMemsetAsync(gram, 0, tail, st)) return (int)e; }
    }
    if (training && use_gram) {
        static const int gram_wgs = getenv("HYB_S1_GRAM_WGS") ? atoi(getenv("HYB_S1_GRAM_WGS")) : 512;          // 1024 measured 0.5 % slower
        int grows = gram_wgs < 1 ? 1 : (gram_wgs > S1_BWD_PART ? S1_BWD_PART : gram_wgs);
        rc = hyb_stage1w_gram(dtype, a, grows, st);
        if (rc) return rc;
        hipLaunchKernelGGL(s1_rows_sum_kernel, dim3(hyb_cdiv(2304, 32)), dim3(256), 0, st, part, gram, grows, 2304ll);
        HYB_LAUNCH_CHECK();
        hipLaunchKernelGGL(s1_gram_stats_kernel<T>, dim3(Cop), dim3(64), 0, st, gram, (const T*)wp, gamma, beta, running_mean,
                           running_var, nbt, momentum, eps, (long long)N * H * W, Co, Cop, scale_shift, mean_invstd, running_out);
        HYB_LAUNCH_CHECK();
        rc = 0;
    } else if (training) {
        rc = wave_private ? hyb_stage1w_pass(dtype, 0, a, gx_rows, st) : s1_dispatch<T, 0>(a, gx, st);
        if (rc) return rc;
        rc = hyb_bn_stats_finalize(part, gx_rows, gamma, beta, running_mean, running_var, nbt, momentum, eps, (long long)N * H * W, Co, Cop,
                                   scale_shift, mean_invstd, running_out, (void*)st);
    } else {
        rc = hyb_bn_finalize(stats, gamma, beta, running_mean, running_var, nbt, 0, momentum, eps, (long long)N * H * W, Co, Cop,
                             scale_shift, mean_invstd, nullptr, (void*)st);
    }
    if (rc) return rc;
    // the apply+pool pass keeps no partial rows, so its grid is free: 2048 workgroups (shorter tile runs, 166 VGPRs = 3 workgroups
    // per CU resident) measured +1 % of the step over 1024.  HYB_S1_APPLY_WGS overrides (A/B).
    static const int apply_wgs = getenv("HYB_S1_APPLY_WGS") ? atoi(getenv("HYB_S1_APPLY_WGS")) : 2048;
    int gx1 = apply_wgs > 0 ? apply_wgs : gx;
    if (gx1 > numTiles) gx1 = (int)numTiles;
    return wave_private ? hyb_stage1w_pass(dtype, 1, a, gx1, st) : s1_dispatch<T, 1>(a, gx1, st);
}

template <typename T>
static int stage1_bwd_t(const void* dpooled, const float* x, const float* weight, const float* gamma, const float* scale_shift,
                        const float* mean_invstd, int training, int N, int H, int W, int Ci, int Co, int Cop, float* dweight, float* dgamma,
                        float* dbeta, const void* packed_in, void* workspace, const void* route, hipStream_t st) {
    const size_t es = sizeof(T);
    char* ws = (char*)workspace;
    T* wp = (T*)ws;                              ws += al256((size_t)Cop * 64 * es);
    T* wp2 = (T*)ws;                             ws += al256((size_t)Cop * 64 * es);
    double* sums = (double*)ws;                  ws += al256(((size_t)Cop * 48 + 2304) * 8);      // reduced row [S1][G], double
    float* part = (float*)ws;
    // the wave-private kernel: 16-bit storage, aligned float4 row segments, no ragged 8x16 blocks, 32-bit buffer offsets
    static const int wave_env = getenv("HYB_S1_WAVE_BWD") ? atoi(getenv("HYB_S1_WAVE_BWD")) : 1;
    // (fp32 storage: only the split-bf16 build has a wave-private kernel, conv_first_wave.hip)
    const bool wave_private = wave_env && (sizeof(T) == 2 || HYB_X3) && (W % 16 == 0) && (H % 8 == 0) && (((uintptr_t)x & 15) == 0) &&
                              (((uintptr_t)dpooled & 15) == 0) && (Cop % 4 == 0) &&
                              (long long)N * Ci * H * W * 4 < (1ll << 32) && (long long)N * (H / 2) * (W / 2) * Cop * (long long)es < (1ll << 32);
    const int dtype = sizeof(T) == 2 ? HYB_BF16 : HYB_F32;
    if (packed_in) {
        wp = (T*)packed_in;                      // packed by the forward pass: [2][Cop][64]
        wp2 = (T*)packed_in + (size_t)Cop * 64;
    } else {
        const long long total = (long long)Cop * 64;
        hipLaunchKernelGGL(s1_pack_kernel<T>, dim3(hyb_cdiv(total, 256)), dim3(256), 0, st, weight, wp, Co, Ci, total);
        HYB_LAUNCH_CHECK();
        if (wave_private) { if (int e = hyb_stage1w_pack(dtype, weight, wp2, Co, Ci, Cop, 0, st)) return e; }
    }
    S1Args a{};
    a.x = x; a.wp = wp; a.wp2 = wp2; a.ss = scale_shift; a.mi = mean_invstd; a.gamma = gamma; a.sums = nullptr; a.dp = dpooled; a.part = part;
    a.N = N; a.H = H; a.W = W; a.Ci = Ci; a.Co = Co; a.Cop = Cop; a.training = training;
    a.inv_count = 1.0f / (float)((long long)N * H * W);
    a.tilesX = hyb_cdiv(W, S1_TW); a.tilesY = hyb_cdiv(H, S1_TH);
    a.inv_tpi = 1.0f / (float)(a.tilesX * a.tilesY); a.inv_tx = 1.0f / (float)a.tilesX;
    a.vec_ok = (W % 4 == 0) && (((uintptr_t)x & 15) == 0);
    // (the forward's apply pass was wave-private under the same rule: s1_route_ok covers its conditions)
    a.route = (route && wave_private && s1_route_ok(es, x, N, H, W, Ci, Cop)) ? (unsigned*)const_cast<void*>(route) : nullptr;
    const long long numTiles = (long long)N * a.tilesX * a.tilesY;
    a.numTiles = (int)numTiles;
    // one pass over x and dpooled (MODE 4), a fixed-order sum of the partial rows, and a finalize on tiny matrices
    static const int bwd_wgs = getenv("HYB_S1_BWD_WGS") ? atoi(getenv("HYB_S1_BWD_WGS")) : 0;                // 0: 512 (768 = three workgroups per CU for the kernel without G measured 1 % slower)
    // (saved_g is decided below from the same inputs; the grid only needs the bound)
    int want = bwd_wgs > 0 ? bwd_wgs : S1_BWD_WGS;
    if (want > S1_BWD_PART) want = S1_BWD_PART;
    int gx = (int)(numTiles < want ? numTiles : want);
    if (gx < 1) gx = 1;
    const long long roww = (long long)Cop * 48 + 2304;
    // the forward pass of a training step (same conditions) left the Gram matrix behind the packed weights: accumulate S1 only
    static const int gram_env = getenv("HYB_S1_GRAM") ? atoi(getenv("HYB_S1_GRAM")) : 1;
    const bool saved_g = gram_env && wave_private && training && packed_in != nullptr && sizeof(T) == 2;    // (the Gram pass is 16-bit only)
    const long long rw = saved_g ? (long long)Cop * 48 : roww;
    int rc = wave_private ? hyb_stage1w_bwd(dtype, a, saved_g ? 0 : 1, gx, st) : s1_dispatch<T, 4>(a, gx, st);
    if (rc) return rc;
    hipLaunchKernelGGL(s1_rows_sum_kernel, dim3(hyb_cdiv(rw, 32)), dim3(256), 0, st, part, sums, gx, rw);
    HYB_LAUNCH_CHECK();
    const double* Gp = saved_g ? (const double*)((const T*)packed_in + (size_t)Cop * 128) : sums + (size_t)Cop * 48;
    hipLaunchKernelGGL(s1_bwd_finalize_kernel<T>, dim3(hyb_cdiv(Co * 36, 256)), dim3(256), 0, st, sums, Gp, wp, mean_invstd, gamma, training,
                       1.0 / (double)((long long)N * H * W), Co, Ci, Cop, dweight, dgamma, dbeta);
    HYB_LAUNCH_CHECK();
    return 0;
}

int hyb_stage1_fwd(int dtype, const float* x, const float* weight, const float* gamma, const float* beta, float* running_mean,
                   float* running_var, long long* nbt, int training, float momentum, float eps, int N, int H, int W, int Ci, int Co, int Cop,
                   void* pooled, float* scale_shift, float* mean_invstd, void* packed_out, void* workspace, float* running_out, int prepacked,
                   void* route, hipStream_t st) {
    if (Ci < 1 || Ci > 4) return HYB_E_ARG;
    if (dtype == HYB_F32) return stage1_fwd_t<float>(dtype, x, weight, gamma, beta, running_mean, running_var, nbt, training, momentum, eps, N, H, W, Ci, Co, Cop, pooled, scale_shift, mean_invstd, packed_out, workspace, running_out, prepacked, route, st);
    if (dtype == HYB_BF16) return stage1_fwd_t<bf16>(dtype, x, weight, gamma, beta, running_mean, running_var, nbt, training, momentum, eps, N, H, W, Ci, Co, Cop, pooled, scale_shift, mean_invstd, packed_out, workspace, running_out, prepacked, route, st);
    return HYB_E_ARG;
}

int hyb_stage1_bwd(int dtype, const void* dpooled, const float* x, const float* weight, const float* gamma, const float* scale_shift,
                   const float* mean_invstd, int training, int N, int H, int W, int Ci, int Co, int Cop, float* dweight, float* dgamma,
                   float* dbeta, const void* packed_in, void* workspace, const void* route, hipStream_t st) {
    if (Ci < 1 || Ci > 4) return HYB_E_ARG;
    if (dtype == HYB_F32) return stage1_bwd_t<float>(dpooled, x, weight, gamma, scale_shift, mean_invstd, training, N, H, W, Ci, Co, Cop, dweight, dgamma, dbeta, packed_in, workspace, route, st);
    if (dtype == HYB_BF16) return stage1_bwd_t<bf16>(dpooled, x, weight, gamma, scale_shift, mean_invstd, training, N, H, W, Ci, Co, Cop, dweight, dgamma, dbeta, packed_in, workspace, route, st);
    return HYB_E_ARG;
}
