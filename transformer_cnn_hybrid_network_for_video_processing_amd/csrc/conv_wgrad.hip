// Weight gradient of the 3x3 conv (autograd of nn.Conv2d, UNet.py:58) on the matrix cores.
//
//   dW[co][tap][ci] = sum over pixels  dy[pixel][co] * x[pixel + tap][ci]
//
// GEMM view: M = co, N = ci (per tap), K = pixels.  Both operands are channel-fastest (NHWC) in HBM,
// but the contraction runs over PIXELS, so every MFMA fragment is a transposed read of an LDS tile:
//   bf16: ds_read_b64_tr_b16 (4 pixels x 16 channels per 16-lane group, delivered channel-major)
//   fp32: eight ds_read_b32 (parity mode only)
// A workgroup owns one output block (64 co x CIT*16 ci x 9 taps, fp32 accumulators in registers) and
// walks a strided subset of 8x16-pixel tiles; partial blocks go to fp32 slabs [S][Cop][9][Cip] that a
// second kernel sums in a fixed order (bitwise reproducible; no float atomics).
#include <stdlib.h>
#include <type_traits>
#include "hyb_common.h"

int hyb_wgrad_reduce_multi(int n, const HybSlabInfo* infos, hipStream_t st);

namespace {

constexpr int WG_TH = 8, WG_TW = 16, WG_HW = WG_TW + 2, WG_HH = WG_TH + 2, WG_HP = WG_HH * WG_HW;   // halo 10 x 18
constexpr int DY_COLS = 64, DY_STRIDE = DY_COLS + 16;          // row stride 80 elements: conflict-free tr reads

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

// 8 pixel values of one channel column: pixels (row0, 4g + 0..3) and (row0 + 1, 4g + 0..3) of a tile image with
// `stride` elements per pixel row-entry and `pitch` pixels per image row.  `base` points at (row0, 4g, channel 0).
__device__ __forceinline__ void tr_frag(Frag<bf16>& f, const bf16* base, int stride, int pitch, int lane) {
    const int qq = (lane & 15) >> 2, pp = lane & 3;
    const bf16* a0 = base + qq * stride + 4 * pp;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(a0));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(a0 + pitch * stride));
    f.v[0] = lo[0]; f.v[1] = lo[1]; f.v[2] = lo[2]; f.v[3] = lo[3];
    f.v[4] = hi[0]; f.v[5] = hi[1]; f.v[6] = hi[2]; f.v[7] = hi[3];
}
__device__ __forceinline__ void tr_frag(Frag<float>& f, const float* base, int stride, int pitch, int lane) {
    const float* a0 = base + (lane & 15);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        f.v[j] = a0[j * stride];
        f.v[4 + j] = a0[(pitch + j) * stride];
    }
}

// Inputs of the fused BatchNorm/ReLU/MaxPool backward (FUSE = true): the gradient tile is computed on the fly from the raw conv
// output y, the pooled-output gradient dp and per-channel constants, instead of being read from a materialised dyraw tensor.
struct WgradFuse {
    const void* y;          // raw conv output NHWC T [N,H,W,Cop]
    const void* dp;         // dpooled NHWC T [N,H/2,W/2,Cop]
    const float* ss;        // scale/shift [2][Cop]
    const float* mi;        // mean/invstd [2][Cop]
    const float* gamma;     // [Co]
    const float* sums;      // [2][Cop]: sum dy, sum dy*xhat
    void* dyraw_out;        // optional: dense gradient written once (by the ci-block 0 workgroups) for the dgrad conv
    long long dyraw_blk;    // 0: NHWC [N][H][W][Cop]; else block-planar [Cop/32][N][H][W][32] with this block stride in elements (v2 kernel only)
    int Co, training;
    float inv_count;
};

template <typename T, int CIT, bool FUSE>
__global__ __launch_bounds__(256) void conv3x3_wgrad_kernel(const T* __restrict__ x, const T* __restrict__ dy, float* __restrict__ slab,
                                                            int N, int H, int W, int Cip, int Cop, int tilesX, int tilesY, int numTiles,
                                                            WgradFuse fz) {
    constexpr int NCT = CIT;                       // co tiles per wave (waves = CIT ci tiles x 4/CIT co groups)
    constexpr int XCOLS = CIT * 16, X_STRIDE = XCOLS + 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* dyt = reinterpret_cast<T*>(smem_raw);                       // [128][DY_STRIDE]
    T* xh = dyt + WG_TH * WG_TW * DY_STRIDE;                       // [180][X_STRIDE]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cit = wave % CIT, cog = wave / CIT;
    const int g = lane >> 4;
    const int nCiBlk = Cip / XCOLS;
    const int co0 = (blockIdx.y / nCiBlk) * 64, ci0 = (blockIdx.y % nCiBlk) * XCOLS;
    // FUSE: per-channel constants of this workgroup's 64 output channels, in LDS behind the two tiles:
    //   v = sc*y + sh (argmax / ReLU gate);  dyraw = A1*y + A0 + (argmax ? k*dy : 0)
    float* cst = reinterpret_cast<float*>(xh + WG_HP * X_STRIDE);          // [5][64]: sc, sh, k, A1, A0
    if (FUSE) {
        if (tid < 64) {
            const int ch = co0 + tid;
            float sc = 0.f, sh = 0.f, k = 0.f, a1 = 0.f, a0 = 0.f;
            if (ch < Cop) {
                sc = fz.ss[ch]; sh = fz.ss[Cop + ch];
                const float mean = fz.mi[ch], inv = fz.mi[Cop + ch];
                k = (ch < fz.Co ? fz.gamma[ch] : 0.f) * inv;
                const float m1 = fz.training ? fz.sums[ch] * fz.inv_count : 0.f, m2 = fz.training ? fz.sums[Cop + ch] * fz.inv_count : 0.f;
                a1 = -k * m2 * inv;
                a0 = -k * m1 + k * m2 * inv * mean;
            }
            cst[tid] = sc; cst[64 + tid] = sh; cst[128 + tid] = k; cst[192 + tid] = a1; cst[256 + tid] = a0;
        }
    }

    f32x4 acc[9][NCT];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int c = 0; c < NCT; ++c) acc[t][c] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int tchunk = (numTiles + (int)gridDim.x - 1) / (int)gridDim.x;       // contiguous run of tiles per workgroup (L2 reuse of shared halos)
    const int tile_end = ((int)blockIdx.x + 1) * tchunk < numTiles ? ((int)blockIdx.x + 1) * tchunk : numTiles;
    for (int tile = blockIdx.x * tchunk; tile < tile_end; ++tile) {
        const int n = tile / (tilesX * tilesY);
        const int trem = tile - n * (tilesX * tilesY);
        const int ty0 = (trem / tilesX) * WG_TH, tx0 = (trem % tilesX) * WG_TW;
        __syncthreads();
        if (!FUSE) {
            // stage dy tile: 128 pixels x 64 channels
            for (int u = tid; u < WG_TH * WG_TW * (DY_COLS / 8); u += 256) {
                const int pix = u >> 3, s = u & 7;
                const int gy = ty0 + (pix >> 4), gx = tx0 + (pix & 15);
                Vec8<T> v;
                if (gy < H && gx < W && co0 + 8 * s < Cop) v.load(dy + ((long long)(n * H + gy) * W + gx) * Cop + co0 + 8 * s);
                else v.zero();
                v.epack();
                v.store(dyt + pix * DY_STRIDE + 8 * s);
            }
        } else {
            // one thread per (2x2 pooling window, 8-channel octet): 32 windows x 8 octets = 256 units
            const int oct = tid & 7, win = tid >> 3, wyy = win >> 3, wxx = win & 7;
            const int gy0 = ty0 + 2 * wyy, gx0 = tx0 + 2 * wxx;
            const int Ho = H >> 1, Wo = W >> 1;
            const bool chok = co0 + 8 * oct < Cop;
            const bool win_ok = (gy0 >> 1) < Ho && (gx0 >> 1) < Wo;
            const T* yb = (const T*)fz.y + co0 + 8 * oct;
            Vec8<T> yv[4], g;
            bool pv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int gy = gy0 + (j >> 1), gx = gx0 + (j & 1);
                pv[j] = chok && gy < H && gx < W;
                if (pv[j]) yv[j].load(yb + ((long long)(n * H + gy) * W + gx) * Cop); else yv[j].zero();
            }
            if (chok && win_ok) g.load((const T*)fz.dp + ((long long)(n * Ho + (gy0 >> 1)) * Wo + (gx0 >> 1)) * Cop + co0 + 8 * oct);
            else g.zero();
            Vec8<T> o[4];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int c = oct * 8 + e;
                const float sc = cst[c], sh = cst[64 + c], kk = cst[128 + c], a1 = cst[192 + c], a0 = cst[256 + c];
                float vmax = yv[0].get(e) * sc + sh;
                int am = 0;
#pragma unroll
                for (int j = 1; j < 4; ++j) {
                    const float v = yv[j].get(e) * sc + sh;
                    if (v > vmax) { vmax = v; am = j; }
                }
                const float kdy = (vmax > 0.f && win_ok) ? kk * g.get(e) : 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float base = fmaf(yv[j].get(e), a1, a0);
                    o[j].set(e, pv[j] ? (am == j ? base + kdy : base) : 0.f);
                }
            }
            const bool writer = fz.dyraw_out && (blockIdx.y % nCiBlk) == 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int ly = 2 * wyy + (j >> 1), lx = 2 * wxx + (j & 1);
                if (writer && pv[j])
                    o[j].store((T*)fz.dyraw_out + ((long long)(n * H + ty0 + ly) * W + tx0 + lx) * Cop + co0 + 8 * oct);
                o[j].epack();                       // (split-bf16 build: the LDS tiles are element-packed, hyb_common.h; else a no-op)
                o[j].store(dyt + (ly * WG_TW + lx) * DY_STRIDE + 8 * oct);
            }
        }
        // stage x halo: 180 pixels x XCOLS channels
        for (int u = tid; u < WG_HP * (XCOLS / 8); u += 256) {
            const int hp = u / (XCOLS / 8), s = u - hp * (XCOLS / 8);
            const int hy = hp / WG_HW, hx = hp - hy * WG_HW;
            const int gy = ty0 + hy - 1, gx = tx0 + hx - 1;
            Vec8<T> v;
            if (gy >= 0 && gy < H && gx >= 0 && gx < W) v.load(x + ((long long)(n * H + gy) * W + gx) * Cip + ci0 + 8 * s);
            else v.zero();
            v.epack();
            v.store(xh + hp * X_STRIDE + 8 * s);
        }
        __syncthreads();
#pragma unroll 1
        for (int ks = 0; ks < 4; ++ks) {           // 32 pixels per step: tile rows 2ks, 2ks+1
            Frag<T> a[NCT];
#pragma unroll
            for (int c = 0; c < NCT; ++c)
                tr_frag(a[c], dyt + ((2 * ks) * WG_TW + 4 * g) * DY_STRIDE + (cog * NCT + c) * 16, DY_STRIDE, WG_TW, lane);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int kh = tap / 3, kw = tap % 3;
                Frag<T> b;
                tr_frag(b, xh + ((2 * ks + kh) * WG_HW + 4 * g + kw) * X_STRIDE + cit * 16, X_STRIDE, WG_HW, lane);
#pragma unroll
                for (int c = 0; c < NCT; ++c) acc[tap][c] = mma32_e(a[c], b, acc[tap][c]);
            }
        }
    }
    // D[row = co][col = ci]: lane holds ci = lane&15, co rows 4g + r
    float* out = slab + (long long)blockIdx.x * Cop * 9 * Cip;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int c = 0; c < NCT; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = co0 + (cog * NCT + c) * 16 + 4 * g + r;
                if (co < Cop) out[((long long)co * 9 + tap) * Cip + ci0 + cit * 16 + (lane & 15)] = acc[tap][c][r];
            }
}

// First stage: x is NCHW fp32 [N,Ci,H,W], Ci <= 3; N-dimension of the GEMM is k = tap*Ci + ci (27 -> 32).
template <typename T>
__global__ __launch_bounds__(256) void conv3x3_wgrad_first_kernel(const float* __restrict__ x, const T* __restrict__ dy,
                                                                  float* __restrict__ slab, int N, int H, int W, int Ci, int Cop,
                                                                  int tilesX, int tilesY, int numTiles) {
    constexpr int DYC = 32, DYS = DYC + 16;        // 32 channels per workgroup, stride 48
    __shared__ __attribute__((aligned(16))) unsigned char smem_raw[WG_TH * WG_TW * DYS * sizeof(float) + (3 * WG_HP + 4) * sizeof(float)];
    T* dyt = reinterpret_cast<T*>(smem_raw);
    float* xh = reinterpret_cast<float*>(smem_raw + WG_TH * WG_TW * DYS * sizeof(float));
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kt = wave & 1, ct = wave >> 1;
    const int g = lane >> 4;
    const int co0 = blockIdx.y * DYC;
    const int zero_idx = 3 * WG_HP;
    if (tid == 0) xh[zero_idx] = 0.f;
    // this lane's k column
    const int k = kt * 16 + (lane & 15);
    int koff = -1;
    if (k < 9 * Ci) { const int tap = k / Ci, ci = k - tap * Ci; koff = ci * WG_HP + (tap / 3) * WG_HW + (tap % 3); }

    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    const int tchunk = (numTiles + (int)gridDim.x - 1) / (int)gridDim.x;       // contiguous run of tiles per workgroup (L2 reuse of shared halos)
    const int tile_end = ((int)blockIdx.x + 1) * tchunk < numTiles ? ((int)blockIdx.x + 1) * tchunk : numTiles;
    for (int tile = blockIdx.x * tchunk; tile < tile_end; ++tile) {
        const int n = tile / (tilesX * tilesY);
        const int trem = tile - n * (tilesX * tilesY);
        const int ty0 = (trem / tilesX) * WG_TH, tx0 = (trem % tilesX) * WG_TW;
        __syncthreads();
        for (int u = tid; u < WG_TH * WG_TW * (DYC / 8); u += 256) {
            const int pix = u >> 2, s = u & 3;
            const int gy = ty0 + (pix >> 4), gx = tx0 + (pix & 15);
            Vec8<T> v;
            if (gy < H && gx < W) v.load(dy + ((long long)(n * H + gy) * W + gx) * Cop + co0 + 8 * s);
            else v.zero();
            v.store(dyt + pix * DYS + 8 * s);
        }
        for (int u = tid; u < Ci * WG_HP; u += 256) {
            const int ci = u / WG_HP, hp = u - ci * WG_HP;
            const int hy = hp / WG_HW, hx = hp - hy * WG_HW;
            const int gy = ty0 + hy - 1, gx = tx0 + hx - 1;
            float v = 0.f;
            if (gy >= 0 && gy < H && gx >= 0 && gx < W) v = x[((long long)(n * Ci + ci) * H + gy) * W + gx];
            xh[u] = v;
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            Frag<T> a, b;
            tr_frag(a, dyt + ((2 * ks) * WG_TW + 4 * g) * DYS + ct * 16, DYS, WG_TW, lane);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int hidx = (2 * ks + (j >> 2)) * WG_HW + 4 * g + (j & 3);
                frag_set<T>(b, j, xh[koff >= 0 ? koff + hidx : zero_idx]);
            }
            acc = mma32(a, b, acc);
        }
    }
    float* out = slab + (long long)blockIdx.x * Cop * 32;
#pragma unroll
    for (int r = 0; r < 4; ++r) out[(long long)(co0 + ct * 16 + 4 * g + r) * 32 + k] = acc[r];
}

// dw[co][ci][tap] = sum_s slab[s][co][tap][ci]  (first: slab[s][co][k], k = tap*Ci + ci); fixed summation order.
__global__ __launch_bounds__(1024) void wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, int S, int first, int Co,
                                                            int Ci, int Cop, int Cip, long long per_slab) {
    long long i; float s;
    if (!rows_reduce_1024(slab, S, per_slab, i, s)) return;
    int co, ci, tap;
    if (first) {
        const int kk = (int)(i % 32);
        co = (int)(i / 32);
        if (kk >= 9 * Ci || co >= Co) return;
        tap = kk / Ci; ci = kk - tap * Ci;
    } else {
        ci = (int)(i % Cip);
        tap = (int)((i / Cip) % 9);
        co = (int)(i / ((long long)9 * Cip));
        if (ci >= Ci || co >= Co) return;
    }
    dw[((long long)co * Ci + ci) * 9 + tap] = s;
}

// -----------------------------------------------------------------------------------------------------------------
// Second generation (bf16, Cip % 64 == 0, Cop % 64 == 0): one 8-wave workgroup per CU owns a 64 co x 64 ci x 9 tap block and
// walks 8 x 28-pixel tiles (7 k-steps of 32 pixels; 28 | 224/2^k).  The waves are specialised:
//   * CW consumer waves (4 = one per SIMD, or 8): the contraction.  At CW = 4, wave = 64 co x 16 ci x 9 taps (36 accumulator tiles); fragments are
//     transposed reads (ds_read_b64_tr_b16) of the two LDS tile images;
//   * PW producer waves (4 or 8, sharing the SIMDs with the consumers): fill the OTHER pair of tile images meanwhile.  The x halo
//     is written by buffer_load ... lds (zero padding = out-of-range lanes, see conv_v2.hip).  FUSE: the gradient tile
//     (BatchNorm/ReLU/MaxPool backward from y, dpooled and per-channel constants) is computed on the vector units, which are
//     otherwise idle under the consumers' MFMAs, and its dense copy dyraw is written once for the dgrad conv;  plain: the
//     gradient tile is DMA'd as well.
//   * both tile images are double buffered, unpadded and XOR-swizzled per 16-byte chunk (DMA needs lane-linear images; the
//     swizzle keeps the transposed reads of 4 consecutive pixels x 16 channels bank-conflict free);
//   * one barrier per tile.
// -----------------------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
constexpr int W2_TH = 8, W2_TW = 28, W2_HW = 30, W2_HP = 10 * 30, W2_PX = 8 * 28;
template <int CI> struct W2X {                       // x-halo image of a CI-channel block (CI = 64 or 32)
    static constexpr int XC = CI / 8;                 // 16-byte chunks per pixel
    static constexpr int XW = (W2_HP * XC + 63) / 64; // DMA wave-instructions (38 / 19)
    static constexpr int XBUF = XW * 512;             // bf16 elements per buffer
    static constexpr size_t LDS = (size_t)2 * (XBUF + W2_PX * 64) * 2 + 5 * 64 * sizeof(float);
    // swizzle of the chunk index: CI = 64 -> see w2_swz; CI = 32 (64-byte pixels: 4 consecutive pixels already cover all banks)
    // only separates the two pixel rows a 32-lane read touches
};
template <int CI> __device__ __forceinline__ int w2x_swz(int row, int col) { return CI == 64 ? ((col & 2) | (((row >> 1) & 1) << 2)) : (((row >> 1) & 1) << 1); }
constexpr int W2_DW = W2_PX * 8 / 64;                 // dy-tile DMA wave-instructions (28)
constexpr int W2_DBUF = W2_PX * 64;                   // bf16 elements per gradient-tile buffer
constexpr unsigned W2_OOB = 0xfffffff0u, W2_RECORDS = 0x80000000u;
constexpr int W2_UNITS = 56 * 8;                      // FUSE work items per tile: (2x2 pooling window, 8-channel octet)

__device__ __forceinline__ int w2_swz(int row, int col) { return (col & 2) | (((row >> 1) & 1) << 2); }

__device__ __forceinline__ void w2_tr(Frag<bf16>& f, const bf16* lo_ptr, const bf16* hi_ptr) {
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)lo_ptr);
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)hi_ptr);
    f.v[0] = lo[0]; f.v[1] = lo[1]; f.v[2] = lo[2]; f.v[3] = lo[3];
    f.v[4] = hi[0]; f.v[5] = hi[1]; f.v[6] = hi[2]; f.v[7] = hi[3];
}

struct W2Tile { int n, ty0, tx0; };
__device__ __forceinline__ W2Tile w2_tile(int tile, int tilesX, int tilesY) {
    W2Tile t;
    t.n = tile / (tilesX * tilesY);
    const int trem = tile - t.n * (tilesX * tilesY);
    t.ty0 = (trem / tilesX) * W2_TH;
    t.tx0 = (trem % tilesX) * W2_TW;
    return t;
}

// One FUSE work item of a producer thread: registers of the loads in flight plus its validity flags.
struct W2Unit {
    union { u32x4 u; bf16x8 v; } y[4], g;
    bool pv[4], win_ok;
};

// the prefetch loads of the tile after next must be ISSUED before the vector work on the next tile (the compiler otherwise sinks
// them behind it, next to their first use, and their HBM latency lands on the critical path of the following iteration)
#ifdef HYB_NO_KEEP_EARLY
#define W2_KEEP_EARLY
#else
#define W2_KEEP_EARLY asm volatile("" ::: "memory")
#endif
template <bool FUSE, int CI, int CW, int PW>
__global__ __launch_bounds__((CW + PW) * 64) void wgrad_v2_kernel(const bf16* __restrict__ x, const bf16* __restrict__ dy, float* __restrict__ slab,
                                                       int N, int H, int W, int Cip, int Cop, int tilesX, int tilesY, int numTiles,
                                                       WgradFuse fz) {
    constexpr int XC = CI / 8, W2_XW = (W2_HP * XC + 63) / 64, W2_XBUF = W2_XW * 512;      // = W2X<CI>
    // CW consumer waves (4: one per SIMD, 8: two per SIMD) share the (CI / 16) x 4 output tiles of the block: NCT co tiles per wave
    constexpr int NCT = 4 * (CI / 16) / CW, NCIT = CI / 16;
    static_assert(NCT >= 1 && NCT * CW == 4 * NCIT, "consumer tiling");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16* const xbuf = reinterpret_cast<bf16*>(smem_raw);                 // [2][W2_XBUF]
    bf16* const dbuf = xbuf + 2 * W2_XBUF;                                // [2][W2_DBUF]
    float* const cst = reinterpret_cast<float*>(dbuf + 2 * W2_DBUF);      // [5][64]: sc, sh, k, A1, A0

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nCiBlk = Cip / CI;
    const int co0 = (blockIdx.y / nCiBlk) * 64, ci0 = (blockIdx.y % nCiBlk) * CI;
    // contiguous run of tiles per workgroup (the host sizes the grid so that no run is empty)
    const int tchunk = (numTiles + (int)gridDim.x - 1) / (int)gridDim.x;
    const int tbegin = blockIdx.x * tchunk;
    const int tcount = (tbegin + tchunk < numTiles ? tbegin + tchunk : numTiles) - tbegin;

    if (FUSE) {
        if (tid < 64) {
            const int ch = co0 + tid;
            const float sc = fz.ss[ch], sh = fz.ss[Cop + ch];
            const float mean = fz.mi[ch], inv = fz.mi[Cop + ch];
            const float k = (ch < fz.Co ? fz.gamma[ch] : 0.f) * inv;
            const float m1 = fz.training ? fz.sums[ch] * fz.inv_count : 0.f, m2 = fz.training ? fz.sums[Cop + ch] * fz.inv_count : 0.f;
            cst[tid] = sc; cst[64 + tid] = sh; cst[128 + tid] = k; cst[192 + tid] = -k * m2 * inv; cst[256 + tid] = -k * m1 + k * m2 * inv * mean;
        }
    }
    __syncthreads();                                   // barrier 0: constants visible

    if (wave >= CW) {
        // ================================================= producers =================================================
        const int pw = wave - CW, ptid = tid - CW * 64;
        constexpr int XT = (W2_XW + PW - 1) / PW, DT = (W2_DW + PW - 1) / PW;
        constexpr int NU = 512 / (PW * 64);            // FUSE work items per producer thread (448 units over PW * 64 threads)
        unsigned xoff[10];                           // XT <= 10 entries used (a CI-dependent array bound captured by the lambdas below
        int xyx[10];                                 // breaks the host-side stub instantiation in ROCm 7.2)
#pragma unroll
        for (int k = 0; k < XT; ++k) {
            int wi = k * PW + pw;
            if (wi > W2_XW - 1) wi = W2_XW - 1;                                     // duplicate the last piece
            const int u = wi * 64 + lane, hp = u / XC, cp = u % XC;
            const int hy = hp / W2_HW, hx = hp - hy * W2_HW;
            xoff[k] = (unsigned)(((hy * W + hx) * Cip + ((cp ^ w2x_swz<CI>(hy, hx)) << 3)) * 2);
            xyx[k] = hp < W2_HP ? ((hy << 16) | hx) : (0x7fff << 16);
        }
        auto x_dma = [&](const W2Tile& t, bf16* xb) {
            const long long base = ((long long)(t.n * H + t.ty0 - 1) * W + (t.tx0 - 1)) * Cip + ci0;
            const __amdgpu_buffer_rsrc_t rs = hyb_rsrc(x + base, W2_RECORDS);
#pragma unroll
            for (int k = 0; k < XT; ++k) {
                const int gy = t.ty0 - 1 + (xyx[k] >> 16), gx = t.tx0 - 1 + (xyx[k] & 0xffff);
                const bool valid = ((unsigned)gy < (unsigned)H) && ((unsigned)gx < (unsigned)W);
                int wi = k * PW + pw;
                if (wi > W2_XW - 1) wi = W2_XW - 1;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)(xb + wi * 512), 16, valid ? xoff[k] : W2_OOB, 0, 0, 0);
            }
        };
        auto dy_dma = [&](const W2Tile& t, bf16* db) {
            const long long base = ((long long)(t.n * H + t.ty0) * W + t.tx0) * Cop + co0;
            const __amdgpu_buffer_rsrc_t rs = hyb_rsrc(dy + base, W2_RECORDS);
#pragma unroll
            for (int k = 0; k < DT; ++k) {
                int wi = k * PW + pw;
                if (wi > W2_DW - 1) wi = W2_DW - 1;
                const int u = wi * 64 + lane, px = u >> 3, cp = u & 7;
                const int ly = px / W2_TW, lx = px - ly * W2_TW;
                const bool valid = (t.ty0 + ly < H) && (t.tx0 + lx < W);
                const unsigned off = (unsigned)(((ly * W + lx) * Cop + ((cp ^ w2_swz(ly, lx)) << 3)) * 2);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)(db + wi * 512), 16, valid ? off : W2_OOB, 0, 0, 0);
            }
        };
        // FUSE work items of this thread: units ptid + i * PW * 64 (indices past 447 repeat unit 447: same
        // values to the same addresses).  Branch-free: invalid lanes go out of the descriptors' range.
        int uoct[NU], uwy[NU], uwx[NU];
        unsigned yoff[NU][4], goff[NU];
#pragma unroll
        for (int i = 0; i < NU; ++i) {
            int u = ptid + i * PW * 64;
            if (u > W2_UNITS - 1) u = W2_UNITS - 1;
            uoct[i] = u & 7;
            const int win = u >> 3;
            uwy[i] = win / 14; uwx[i] = win - uwy[i] * 14;
#pragma unroll
            for (int j = 0; j < 4; ++j) yoff[i][j] = (unsigned)((((2 * uwy[i] + (j >> 1)) * W + 2 * uwx[i] + (j & 1)) * Cop + 8 * uoct[i]) * 2);
            goff[i] = (unsigned)(((uwy[i] * (W >> 1) + uwx[i]) * Cop + 8 * uoct[i]) * 2);
        }
        // dyraw store offsets: NHWC = the y load offsets; block-planar: pixel stride 32 channels, this lane's 32-channel half of the
        // workgroup's 64 output channels at + dyraw_blk
        const bool planar = FUSE && fz.dyraw_blk != 0;
        unsigned ooff[NU][4];
#pragma unroll
        for (int i = 0; i < NU; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                ooff[i][j] = planar ? (unsigned)((((2 * uwy[i] + (j >> 1)) * W + 2 * uwx[i] + (j & 1)) * 32 + 8 * (uoct[i] & 3)) * 2 +
                                                 (uoct[i] >> 2) * fz.dyraw_blk * 2)
                                    : yoff[i][j];
        const bool writer = FUSE && fz.dyraw_out && ci0 == 0;
        auto fuse_load = [&](const W2Tile& t, W2Unit (&un)[NU]) {
            const int Ho = H >> 1, Wo = W >> 1;
            const long long ybase = ((long long)(t.n * H + t.ty0) * W + t.tx0) * Cop + co0;
            const __amdgpu_buffer_rsrc_t y_rs = hyb_rsrc((const bf16*)fz.y + ybase, W2_RECORDS);
            const __amdgpu_buffer_rsrc_t g_rs = hyb_rsrc((const bf16*)fz.dp + ((long long)(t.n * Ho + (t.ty0 >> 1)) * Wo + (t.tx0 >> 1)) * Cop + co0, W2_RECORDS);
#pragma unroll
            for (int i = 0; i < NU; ++i) {
                un[i].win_ok = ((t.ty0 >> 1) + uwy[i]) < Ho && ((t.tx0 >> 1) + uwx[i]) < Wo;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    un[i].pv[j] = (t.ty0 + 2 * uwy[i] + (j >> 1) < H) && (t.tx0 + 2 * uwx[i] + (j & 1) < W);
                    un[i].y[j].u = __builtin_amdgcn_raw_buffer_load_b128(y_rs, un[i].pv[j] ? yoff[i][j] : W2_OOB, 0, 0);
                }
                un[i].g.u = __builtin_amdgcn_raw_buffer_load_b128(g_rs, un[i].win_ok ? goff[i] : W2_OOB, 0, 0);
            }
        };
        auto fuse_compute = [&](const W2Tile& t, W2Unit (&un)[NU], bf16* db) {
            const long long ybase = planar ? ((long long)(t.n * H + t.ty0) * W + t.tx0) * 32 + (long long)(co0 / 32) * fz.dyraw_blk
                                           : ((long long)(t.n * H + t.ty0) * W + t.tx0) * Cop + co0;
            const __amdgpu_buffer_rsrc_t o_rs = hyb_rsrc((bf16*)fz.dyraw_out + ybase, writer ? W2_RECORDS : 0u);
#pragma unroll
            for (int i = 0; i < NU; ++i) {
                union { u32x4 u; bf16x8 v; } o[4];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int c = uoct[i] * 8 + e;
                    const float sc = cst[c], sh = cst[64 + c], kk = cst[128 + c], a1 = cst[192 + c], a0 = cst[256 + c];
                    float yf[4], v[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) { yf[j] = (float)un[i].y[j].v[e]; v[j] = fmaf(yf[j], sc, sh); }
                    // the window maximum and its FIRST position in torch's scan order (0,0),(0,1),(1,0),(1,1); the flags live in scalar
                    // masks, so the routing costs one select per position:  dyraw_j = y_j*A1 + (A0 [+ k*dy at the arg-max])
                    const float vmax = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
                    const bool f0 = v[0] == vmax, f1 = !f0 && v[1] == vmax, f2 = !f0 && !f1 && v[2] == vmax;
                    const bool fl[4] = {f0, f1, f2, !(f0 || f1 || f2)};
                    const float kdy = (vmax > 0.f && un[i].win_ok) ? kk * (float)un[i].g.v[e] : 0.f;
                    const float a0k = a0 + kdy;
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j].v[e] = (bf16)fmaf(yf[j], a1, fl[j] ? a0k : a0);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (!un[i].pv[j]) o[j].u = u32x4{0u, 0u, 0u, 0u};          // pixels outside the image (a select per register)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int ly = 2 * uwy[i] + (j >> 1), lx = 2 * uwx[i] + (j & 1);
                    *reinterpret_cast<bf16x8*>(db + (ly * W2_TW + lx) * 64 + ((uoct[i] ^ w2_swz(ly, lx)) << 3)) = o[j].v;
                    __builtin_amdgcn_raw_buffer_store_b128(o[j].u, o_rs, un[i].pv[j] ? ooff[i][j] : W2_OOB, 0, 0);
                }
            }
        };

        // End of a producer iteration.  Its vector-memory operations were issued in the order [XT halo DMAs][10 prefetch loads of
        // the tile after][8 dyraw stores]; the images are complete once the DMAs have landed and the LDS writes are done, so the
        // counted wait leaves the prefetch loads and the stores in flight across the barrier (a __syncthreads() would drain
        // them and add a store round trip to every tile).
        auto publish = [&]() {
            if (FUSE && NU == 2) asm volatile("s_waitcnt vmcnt(18) lgkmcnt(0)" ::: "memory");         // 2 x (5 loads + 4 stores)
            else if (FUSE) asm volatile("s_waitcnt vmcnt(9) lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        };
        // tile i of this workgroup; the tiles "after the last" are the last one again (harmless repeats)
        auto tl = [&](int i) { return w2_tile(tbegin + (i < tcount ? i : tcount - 1), tilesX, tilesY); };
        W2Unit ua[NU], ub[NU];
        W2Tile t0 = tl(0);
        x_dma(t0, xbuf);
        if (FUSE) { fuse_load(t0, ua); fuse_compute(t0, ua, dbuf); } else dy_dma(t0, dbuf);
        W2Tile t1 = tl(1);
        if (FUSE) fuse_load(t1, ua);                   // in flight across the barrier
        __syncthreads();                               // barrier 1: tile 0 staged
        for (int i = 0; i < tcount; i += 2) {
            // even iteration: consumers read buffers 0; fill buffers 1 with t1 (registers ua), prefetch t2 into ub
            {
                const W2Tile t2 = tl(i + 2);
                x_dma(t1, xbuf + W2_XBUF);
                asm volatile("" ::: "memory");         // the DMAs stay the OLDEST vector-memory operations of the iteration (see publish)
                if (FUSE) { fuse_load(t2, ub); W2_KEEP_EARLY; fuse_compute(t1, ua, dbuf + W2_DBUF); } else dy_dma(t1, dbuf + W2_DBUF);
                publish();
                t1 = t2;
            }
            if (i + 1 >= tcount) break;
            // odd iteration: consumers read buffers 1; fill buffers 0 with t1 (registers ub), prefetch the next into ua
            {
                const W2Tile t2 = tl(i + 3);
                x_dma(t1, xbuf);
                asm volatile("" ::: "memory");
                if (FUSE) { fuse_load(t2, ua); W2_KEEP_EARLY; fuse_compute(t1, ub, dbuf); } else dy_dma(t1, dbuf);
                publish();
                t1 = t2;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }

    // =================================================== consumers ===================================================
    const int cit = wave % NCIT, coh = wave / NCIT;           // this wave: ci tile cit, co tiles coh*NCT .. coh*NCT + NCT-1
    const int g = lane >> 4, qq = (lane & 15) >> 2, pp = lane & 3;
    // fragment addresses (elements).  dy tile pixel (row, col) -> (row*28 + col)*64; x halo pixel -> (row*30 + col)*64.
    int aoff[NCT];
#pragma unroll
    for (int c = 0; c < NCT; ++c) {
        const int chunk = (coh * NCT + c) * 2 + (pp >> 1);
        aoff[c] = (2 * g * W2_TW + qq) * 64 + (((chunk ^ ((qq & 2) | ((g & 1) << 2))) << 3) | ((pp & 1) << 2));
    }
    int boff[3][2];      // [kw][row parity class]: swizzled chunk offset inside the pixel
    {
        const int chunk = cit * 2 + (pp >> 1);
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
#pragma unroll
            for (int sy = 0; sy < 2; ++sy)
                boff[kw][sy] = ((chunk ^ (CI == 64 ? (((kw + qq) & 2) | ((((g & 1) ^ sy)) << 2)) : (((g & 1) ^ sy) << 1))) << 3) | ((pp & 1) << 2);
    }
    const int bpix = (2 * g * W2_HW + qq) * CI;

    f32x4 acc[9][NCT];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int c = 0; c < NCT; ++c) acc[t][c] = f32x4{0.f, 0.f, 0.f, 0.f};

    __syncthreads();                                   // barrier 1: tile 0 staged
    for (int it = 0; it < tcount; ++it) {
        const bf16* xb = xbuf + (it & 1) * W2_XBUF;
        const bf16* db = dbuf + (it & 1) * W2_DBUF;
        // software pipeline over the 63 (k-step, tap) steps of the tile: the x fragment is read two steps ahead (ring of 3), the
        // four gradient fragments of k-step j+1 are read during taps 2..5 of k-step j; the only exposed LDS latency is the
        // first step after the tile barrier
        auto load_a = [&](Frag<bf16>& f, int j, int c) { w2_tr(f, db + aoff[c] + j * 256, db + aoff[c] + j * 256 + W2_TW * 64); };
        auto load_b = [&](Frag<bf16>& f, int j, int tap) {
            const int kh = tap / 3, kw = tap % 3;
            // rows 2g+kh (lo) and 2g+kh+1 (hi): swizzle class = ((row >> 1) & 1) ^ (g & 1)
            w2_tr(f, xb + bpix + (kh * W2_HW + 4 * j + kw) * CI + boff[kw][(kh >> 1) & 1],
                  xb + bpix + ((kh + 1) * W2_HW + 4 * j + kw) * CI + boff[kw][((kh + 1) >> 1) & 1]);
        };
        Frag<bf16> a[2][NCT], b[3];
#pragma unroll
        for (int c = 0; c < NCT; ++c) load_a(a[0][c], 0, c);
        load_b(b[0], 0, 0);
        load_b(b[1], 0, 1);
#pragma unroll
        for (int st = 0; st < 63; ++st) {
            const int j = st / 9, tap = st % 9;
            if (st + 2 < 63) load_b(b[(st + 2) % 3], (st + 2) / 9, (st + 2) % 9);
            if (j < 6 && tap >= 2 && tap < 2 + NCT) load_a(a[(j + 1) & 1][tap - 2], j + 1, tap - 2);
#pragma unroll
            for (int c = 0; c < NCT; ++c) acc[tap][c] = mma32(a[j & 1][c], b[st % 3], acc[tap][c]);
        }
        // pin the interleave: after each group of 4 MFMAs the reads of two steps ahead (2 or 4 transposed reads)
#pragma unroll
        for (int st = 0; st < 63; ++st) {
            const int j = st / 9, tap = st % 9;
            const bool rb = st + 2 < 63, ra = j < 6 && tap >= 2 && tap < 2 + NCT;
            if (ra && rb) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
            else if (ra || rb) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, NCT, 0);
        }
        __syncthreads();                               // the other pair of images is complete, this pair may be overwritten
    }

    // D[row = co][col = ci]: lane holds ci = lane&15, co rows 4g + r
    float* out = slab + (long long)blockIdx.x * Cop * 9 * Cip;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int c = 0; c < NCT; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = co0 + (coh * NCT + c) * 16 + 4 * g + r;
                out[((long long)co * 9 + tap) * Cip + ci0 + cit * 16 + (lane & 15)] = acc[tap][c][r];
            }
}

template <bool FUSE, int CI, int CW, int PW>
int w2_launch_cw(dim3 grid, HybProfileHook* hook, hipStream_t st, const bf16* x, const bf16* dy, float* slab, int N, int H, int W, int Cip, int Cop,
                 int tX, int tY, int nT, const WgradFuse& fz) {
    static HybAttrOnce once;                                   // per template instantiation, per device
    if (int e = hyb_set_lds_attr(once, (const void*)wgrad_v2_kernel<FUSE, CI, CW, PW>, (int)W2X<CI>::LDS)) return e;
    if (hook) hipEventRecord(hook->ev0, st);
    hipLaunchKernelGGL((wgrad_v2_kernel<FUSE, CI, CW, PW>), grid, dim3((CW + PW) * 64), W2X<CI>::LDS, st, x, dy, slab, N, H, W, Cip, Cop, tX, tY, nT, fz);
    if (hook) hipEventRecord(hook->ev1, st);
    return 0;
}
template <bool FUSE, int CI>
int w2_launch(dim3 grid, HybProfileHook* hook, hipStream_t st, const bf16* x, const bf16* dy, float* slab, int N, int H, int W, int Cip, int Cop,
              int tX, int tY, int nT, const WgradFuse& fz) {
    // consumer x producer waves per workgroup: 4x4 (one of each per SIMD) or 8x8 (two of each per SIMD).  Measured in one call
    // (fused, config 2): the HBM-bound 32-channel-block stage gains from 8x8 (127 -> 110 us), the 64-channel-block stages lose
    // (94 -> 110 us: 128 VGPRs spill the 36-tile accumulators' helpers); 8 consumers + 4 producers is never best when fused.
    static const int env = getenv("HYB_WGRAD_WAVES") ? atoi(getenv("HYB_WGRAD_WAVES")) : 0;
    const int cfg = env ? env : (CI == 32 ? 88 : 44);
    if (cfg == 88) return w2_launch_cw<FUSE, CI, 8, 8>(grid, hook, st, x, dy, slab, N, H, W, Cip, Cop, tX, tY, nT, fz);
    return w2_launch_cw<FUSE, CI, 4, 4>(grid, hook, st, x, dy, slab, N, H, W, Cip, Cop, tX, tY, nT, fz);
}

#include "conv_wgrad_v3.h"
#ifdef HYB_WGRAD_EXPERIMENTS      // scripts/micro/wgrad_variants: the 16-wave and the pipelined 12-wave forms measured in round 3 (not faster)
#include "conv_wgrad_v4.h"
#include "conv_wgrad_v5.h"
#endif

// few slabs (second-generation kernel: S = 256 / blocks): one thread per output element walks the S slabs; loads are issued
// eight at a time, the additions keep the slab order
// The slab sums of up to four stages in ONE launch (they feed nothing but the optimizer, so the backbone backward defers them to its end:
// three dependent launches of 5.6-11 us become one).  Flat grid; a block is 64 columns x 4 row groups: thread (c, g) sums slabs g, g + 4, ...
// of its column with eight loads in flight, the four groups are combined through LDS as (g0 + g1) + (g2 + g3).  Every path that sums
// second / third generation slabs uses this kernel (a single stage = n = 1), so deferred and immediate sums are the same bits.
struct SlabMulti { HybSlabInfo g[4]; int begin[5]; int n; };
__global__ __launch_bounds__(256) void wgrad_reduce_multi_kernel(SlabMulti a) {
    __shared__ f32x4 red[4][64];
    int gi = 0;
#pragma unroll
    for (int i = 1; i < 4; ++i)
        if (i < a.n && (int)blockIdx.x >= a.begin[i]) gi = i;
    const HybSlabInfo s = a.g[gi];
    const int c4 = threadIdx.x & 63, grp = threadIdx.x >> 6;        // thread: 4 consecutive columns (16-byte loads: 1 KB per slab row and block), row group grp
    const long long i = ((long long)((int)blockIdx.x - a.begin[gi]) * 64 + c4) * 4;
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    if (i < s.per_slab) {
        const float* p = s.slab + i;
        int k = grp;
        for (; k + 28 < s.S; k += 32) {
            f32x4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const f32x4*>(p + (long long)(k + 4 * j) * s.per_slab);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc += v[j];
        }
        for (; k < s.S; k += 4) acc += *reinterpret_cast<const f32x4*>(p + (long long)k * s.per_slab);
    }
    red[grp][c4] = acc;
    __syncthreads();
    if (grp != 0 || i >= s.per_slab) return;
    const f32x4 v = (red[0][c4] + red[1][c4]) + (red[2][c4] + red[3][c4]);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const long long ie = i + e;
        const int ci = (int)(ie % s.Cip), tap = (int)((ie / s.Cip) % 9), co = (int)(ie / ((long long)9 * s.Cip));
        if (ci < s.Ci && co < s.Co) s.dw[((long long)co * s.Ci + ci) * 9 + tap] = v[e];
    }
}

}  // namespace
int hyb_wgrad_reduce_multi(int n, const HybSlabInfo* infos, hipStream_t st) {
    if (n < 1 || n > 4 || !infos) return HYB_E_ARG;
    SlabMulti a{};
    int blocks = 0;
    for (int i = 0; i < n; ++i) {
        if (!infos[i].slab || !infos[i].dw || infos[i].S < 1 || infos[i].per_slab < 1) return HYB_E_ARG;
        a.g[i] = infos[i];
        a.begin[i] = blocks;
        if (infos[i].per_slab % 4 != 0 || ((uintptr_t)infos[i].slab & 15) != 0) return HYB_E_ARG;      // 16-byte loads
        blocks += hyb_cdiv(infos[i].per_slab, 256);
    }
    a.begin[n] = blocks; a.n = n;
    hipLaunchKernelGGL(wgrad_reduce_multi_kernel, dim3(blocks), dim3(256), 0, st, a);
    HYB_LAUNCH_CHECK();
    return 0;
}
namespace {

struct WgradPlan { int S, gy, cit; long long per_slab; };

inline WgradPlan wgrad_plan(int first, int N, int H, int W, int Cip, int Cop) {
    WgradPlan p;
    const long long numTiles = (long long)N * hyb_cdiv(H, WG_TH) * hyb_cdiv(W, WG_TW);
    if (first) { p.cit = 0; p.gy = Cop / 32; p.per_slab = (long long)Cop * 32; }
    else { p.cit = (Cip % 64 == 0) ? 4 : 2; p.gy = ((Cop + 63) / 64) * (Cip / (p.cit * 16)); p.per_slab = (long long)Cop * 9 * Cip; }
    static const int wgs = getenv("HYB_WGRAD1_WGS") ? atoi(getenv("HYB_WGRAD1_WGS")) : 512;       // first-generation kernel: workgroups in total (A/B)
    long long s = (wgs > 0 ? wgs : 512) / p.gy;
    if (s < 1) s = 1;
    if (s > numTiles) s = numTiles;
    p.S = (int)s;
    return p;
}

template <typename T>
int wgrad_t(int first, const void* x, const void* dy, float* dw, int N, int H, int W, int Ci, int Cip, int Co, int Cop, void* ws,
            size_t ws_bytes, hipStream_t st, const WgradFuse* fz = nullptr, HybSlabInfo* defer = nullptr) {
    if (defer) defer->S = 0;
    const WgradPlan p = wgrad_plan(first, N, H, W, Cip, Cop);
    if (ws_bytes < (size_t)p.S * p.per_slab * sizeof(float)) return HYB_E_WORKSPACE;
    float* slab = (float*)ws;
    if constexpr (sizeof(T) == 2) {
        static const int v2 = getenv("HYB_WGRAD_V2") ? atoi(getenv("HYB_WGRAD_V2")) : 1;
        const int ci_blk = Cip % 64 == 0 ? 64 : 32;
        const int blocks = (Cop / 64) * (Cip / ci_blk);
        if (v2 && !first && Cip % 32 == 0 && Cop % 64 == 0 && blocks <= 256 && (long long)12 * W * (Cip > Cop ? Cip : Cop) < (1ll << 29)) {
            // generation of the fused kernel for this shape: 0 second (below), 1 third (conv_wgrad_v3.h); 2 / 3 only in experiment builds
            const int gen3 = (fz && ci_blk == 64) ? w3_supported(H, W, Cip, Cop) : 0;
            const int tX = hyb_cdiv(W, W2_TW), tY = gen3 == 3 ? H / 4 : hyb_cdiv(H, W2_TH);
            const long long nT = (long long)N * tX * tY;
            int S = 256 / blocks;
            if (S > p.S) S = p.S;                          // never more slabs than the workspace query promised
            if (S > nT) S = (int)nT;
            S = hyb_cdiv(nT, hyb_cdiv(nT, S));                // contiguous runs of ceil(nT / S) tiles: drop the empty ones
            HybProfileHook* hook2 = hyb_find_hook(2, Cip, Cop);
            const WgradFuse fzv = fz ? *fz : WgradFuse{};
            const dim3 grid2(S, blocks);
            int lrc = 0;
#ifdef HYB_WGRAD_EXPERIMENTS
            if (gen3 == 3) lrc = w5_launch(grid2, hook2, st, (const bf16*)x, slab, N, H, W, Cip, Cop, (int)nT, fzv);
            else if (gen3 == 2) lrc = w4_launch(grid2, hook2, st, (const bf16*)x, slab, N, H, W, Cip, Cop, tX, tY, (int)nT, fzv);
            else
#endif
            if (gen3) lrc = w3_launch(grid2, hook2, st, (const bf16*)x, slab, N, H, W, Cip, Cop, tX, tY, (int)nT, fzv);
            else if (ci_blk == 64) lrc = fz ? w2_launch<true, 64>(grid2, hook2, st, (const bf16*)x, (const bf16*)dy, slab, N, H, W, Cip, Cop, tX, tY, (int)nT, fzv)
                                       : w2_launch<false, 64>(grid2, hook2, st, (const bf16*)x, (const bf16*)dy, slab, N, H, W, Cip, Cop, tX, tY, (int)nT, fzv);
            else lrc = fz ? w2_launch<true, 32>(grid2, hook2, st, (const bf16*)x, (const bf16*)dy, slab, N, H, W, Cip, Cop, tX, tY, (int)nT, fzv)
                          : w2_launch<false, 32>(grid2, hook2, st, (const bf16*)x, (const bf16*)dy, slab, N, H, W, Cip, Cop, tX, tY, (int)nT, fzv);
            if (lrc) return lrc;
            HYB_LAUNCH_CHECK();
            if (!dw) return 0;
            const HybSlabInfo info{slab, dw, S, Co, Ci, Cip, p.per_slab};
            if (defer) { *defer = info; return 0; }              // the caller sums the slabs later (hyb_wgrad_reduce_multi), with other stages'
            return hyb_wgrad_reduce_multi(1, &info, st);
        }
    }
    const int tilesX = hyb_cdiv(W, WG_TW), tilesY = hyb_cdiv(H, WG_TH);
    const int numTiles = N * tilesX * tilesY;
    dim3 grid(p.S, p.gy);
    HybProfileHook* hook = first ? nullptr : hyb_find_hook(2, Cip, Cop);
    if (hook) hipEventRecord(hook->ev0, st);
    if (first) {
        hipLaunchKernelGGL(conv3x3_wgrad_first_kernel<T>, grid, dim3(256), 0, st, (const float*)x, (const T*)dy, slab, N, H, W, Ci, Cop, tilesX,
                           tilesY, numTiles);
    } else if (p.cit == 4) {
        const size_t lds = (size_t)(WG_TH * WG_TW * DY_STRIDE + WG_HP * (64 + 16)) * sizeof(T) + 5 * 64 * sizeof(float);
        if (fz) hipLaunchKernelGGL((conv3x3_wgrad_kernel<T, 4, true>), grid, dim3(256), lds, st, (const T*)x, (const T*)dy, slab, N, H, W, Cip, Cop,
                                   tilesX, tilesY, numTiles, *fz);
        else hipLaunchKernelGGL((conv3x3_wgrad_kernel<T, 4, false>), grid, dim3(256), lds, st, (const T*)x, (const T*)dy, slab, N, H, W, Cip, Cop,
                                tilesX, tilesY, numTiles, WgradFuse{});
    } else {
        const size_t lds = (size_t)(WG_TH * WG_TW * DY_STRIDE + WG_HP * (32 + 16)) * sizeof(T) + 5 * 64 * sizeof(float);
        if (fz) hipLaunchKernelGGL((conv3x3_wgrad_kernel<T, 2, true>), grid, dim3(256), lds, st, (const T*)x, (const T*)dy, slab, N, H, W, Cip, Cop,
                                   tilesX, tilesY, numTiles, *fz);
        else hipLaunchKernelGGL((conv3x3_wgrad_kernel<T, 2, false>), grid, dim3(256), lds, st, (const T*)x, (const T*)dy, slab, N, H, W, Cip, Cop,
                                tilesX, tilesY, numTiles, WgradFuse{});
    }
    if (hook) hipEventRecord(hook->ev1, st);
    HYB_LAUNCH_CHECK();
    if (!dw) return 0;            // caller only wants the partial slabs (used to time the contraction kernel alone)
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(hyb_cdiv(p.per_slab, 32)), dim3(1024), 0, st, slab, dw, p.S, first, Co, Ci, Cop, Cip, p.per_slab);
    HYB_LAUNCH_CHECK();
    return 0;
}

}  // namespace

// Internal: weight gradient with the BatchNorm/ReLU/MaxPool backward fused into the tile staging (non-first stages)
// which shapes take the warp-specialised kernel (the only one that can write a block-planar dyraw): the tests of wgrad_t
int hyb_wgrad_v2_supported(int dtype, int W, int Cip, int Cop) {
    static const int v2 = getenv("HYB_WGRAD_V2") ? atoi(getenv("HYB_WGRAD_V2")) : 1;
    const int ci_blk = Cip % 64 == 0 ? 64 : 32;
    return dtype == HYB_BF16 && v2 && Cip % 32 == 0 && Cop % 64 == 0 && (Cop / 64) * (Cip / ci_blk) <= 256 &&
           (long long)12 * W * (Cip > Cop ? Cip : Cop) < (1ll << 29);
}

int hyb_conv3x3_wgrad_fused(int dtype, const void* x, const void* y, const void* dp, const float* ss, const float* mi, const float* gamma,
                            const float* sums, int training, long long count, void* dyraw_out, long long dyraw_blk, float* dw, int N, int H, int W,
                            int Ci, int Cip, int Co, int Cop, void* workspace, size_t workspace_bytes, hipStream_t st, HybSlabInfo* defer) {
    WgradFuse fz{y, dp, ss, mi, gamma, sums, dyraw_out, dyraw_blk, Co, training, 1.0f / (float)count};
    if (dtype == HYB_F32) return wgrad_t<float>(0, x, nullptr, dw, N, H, W, Ci, Cip, Co, Cop, workspace, workspace_bytes, st, &fz, defer);
    if (dtype == HYB_BF16) return wgrad_t<bf16>(0, x, nullptr, dw, N, H, W, Ci, Cip, Co, Cop, workspace, workspace_bytes, st, &fz, defer);
    return HYB_E_ARG;
}

extern "C" size_t hyb_conv3x3_wgrad_workspace(int first, int N, int H, int W, int Cip, int Cop) {
    if (N <= 0 || H <= 0 || W <= 0 || Cop <= 0 || Cop % 32 != 0 || (!first && (Cip <= 0 || Cip % 32 != 0))) return 0;
    const WgradPlan p = wgrad_plan(first, N, H, W, Cip, Cop);
    return (size_t)p.S * p.per_slab * sizeof(float);
}

extern "C" int hyb_conv3x3_wgrad(int dtype, int first, const void* x, const void* dy, float* dw, int N, int H, int W, int Ci, int Cip,
                                 int Co, int Cop, void* workspace, size_t workspace_bytes, void* stream) {
    HYB_CHECK_ARG(x && dy && workspace && N > 0 && H > 0 && W > 0 && Co > 0 && Ci > 0 && Cop % 32 == 0 && Cop >= Co);
    if (first) HYB_CHECK_ARG(Ci <= 3);
    else HYB_CHECK_ARG(Cip % 32 == 0 && Cip >= Ci);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == HYB_F32) return wgrad_t<float>(first, x, dy, dw, N, H, W, Ci, Cip, Co, Cop, workspace, workspace_bytes, st);
    if (dtype == HYB_BF16) return wgrad_t<bf16>(first, x, dy, dw, N, H, W, Ci, Cip, Co, Cop, workspace, workspace_bytes, st);
    return HYB_E_ARG;
}
