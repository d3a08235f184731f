// conv3x3 (stride 1, zero pad 1) as an implicit GEMM on the matrix cores, NHWC activations.
// Replaces nn.Conv2d of UNet.py:58 (forward) and its autograd dgrad (with mode-1 packed weights).
//
// GEMM view per workgroup:  D[co][pixel] = sum_k  Wp[co][k] * patch[k][pixel],  k = (tap, ci)
//   A operand = packed weights, held in REGISTERS for one (tap, 32-channel chunk) step
//   B operand = 8 consecutive input channels of one pixel of a 4x4 pixel patch, read with one
//               ds_read_b128 from an XOR-swizzled halo tile of the input staged in LDS
//   a wave owns 8 patches (128 pixels) x NT*16 output channels (acc = 8*NT f32x4)
// Workgroup = 4 waves = CB channel blocks x PG pixel groups; the grid walks image tiles
// (persistent, grid-stride) and folds the BatchNorm batch statistics (sum, sum of squares per
// output channel, UNet.py:59) into the epilogue from the fp32 accumulators.
#include <stdlib.h>
#include "hyb_common.h"
#include "conv_first.h"

namespace {

template <int PG> struct TileGeom;
template <> struct TileGeom<4> { static constexpr int PHP = 4, PWP = 8; };   // 16 x 32 pixels
template <> struct TileGeom<2> { static constexpr int PHP = 4, PWP = 4; };   // 16 x 16 pixels
template <> struct TileGeom<1> { static constexpr int PHP = 2, PWP = 4; };   //  8 x 16 pixels

// swizzle of the 8-channel fragment slot inside a halo pixel (see DESIGN.md "conv LDS image"):
// conflict-free ds_read_b128 for a 4x4-pixel patch at CK = 32, 64, >=128 (bf16).
__device__ __forceinline__ int halo_swz(int hp, int hy, int spf) {
    int sw = (hy & 1) << 1;
    if (spf == 8) sw |= ((hp >> 1) & 1) << 2;
    else if (spf >= 16) sw |= (hp & 3) << 2;
    return sw & (spf - 1);
}

// origin (in halo pixels) of patch m of pixel-group pg, relative to the tile origin
template <int PG> __device__ __host__ constexpr int patch_row(int pg, int m) { return PG == 4 ? pg : (PG == 2 ? pg * 2 + m / 4 : m / 4); }
template <int PG> __device__ __host__ constexpr int patch_col(int m) { return PG == 4 ? m : m % 4; }

template <typename T, int NT, int CB, int PG, int CK, bool STATS, bool WLDS = false>
__global__ __launch_bounds__(256, 2) void conv3x3_nhwc_kernel(const T* __restrict__ x, const T* __restrict__ wp,
                                                           T* __restrict__ y, float* __restrict__ stats,
                                                           int N, int H, int W, int Cip, int Cop,
                                                           int tilesX, int tilesY, int numTiles) {
    constexpr int PHP = TileGeom<PG>::PHP, PWP = TileGeom<PG>::PWP;
    constexpr int TH = 4 * PHP, TW = 4 * PWP, HH = TH + 2, HW_ = TW + 2, HP = HH * HW_;
    constexpr int MT = 8;
    // split-bf16 build, fp32 storage: the halo image and the packed weights hold pre-split fragments (hyb_common.h) -- the image is split
    // once when staged instead of once per tap and wave, the weights once per step by the pack kernels
    constexpr bool PRESPLIT = HYB_X3 && sizeof(T) == 4;
    constexpr int SPF = CK / 8;                            // 8-channel fragment slots per halo pixel
    constexpr int NCHUNK = CK / 32;
    static_assert(HW_ % 4 == 2, "swizzle derivation assumes halo width = 2 mod 4");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* halo = reinterpret_cast<T*>(smem_raw);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cb = wave % CB, pg = wave / CB;
    const int p = lane & 15, q = lane >> 4, py = p >> 2, px = p & 3;
    const int co_base = blockIdx.y * (CB * NT * 16) + cb * (NT * 16);
    const int nCblk = Cip / CK;
    const long long wrow = (long long)9 * Cip;            // packed row: [Cip/32][9][32]

    // this lane's A rows (one per co tile): permuted so that a lane ends up with NT*4 consecutive channels
    const T* wlane[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
        wlane[t] = wp + (long long)(co_base + (p >> 2) * (NT * 4) + t * 4 + (p & 3)) * wrow + 8 * q;

    // LDS element offset of this lane's pixel in patch 0 of its wave (tap (0,0)); patches m and taps add constants
    const int lane_el = ((patch_row<PG>(pg, 0) * 4 + py) * HW_ + px) * CK;
    // swizzle of the fragment slot: depends on the lane and the tap only (patch origins are multiples of 4 pixels)
    const int sw_row = py, sw_col = 2 * py + px;

    // BN sums live in LDS behind the halo image: one private slot per wave [4][2][NT*16] (single owner lane per entry, so the
    // per-tile updates need no atomics and the result is bit-reproducible); combined in a fixed order at the end
    float* wgstat = reinterpret_cast<float*>(smem_raw + (size_t)HP * CK * sizeof(T));
    if (STATS) {
        for (int i = tid; i < 4 * 2 * NT * 16; i += 256) wgstat[i] = 0.f;
    }
    // WLDS: per-tap weight slices [CB*NT*16 co][32 ci] shared by the workgroup, ring of 3 slots behind the stat slots
    constexpr int CBW = CB * NT * 16;                       // output channels per workgroup
    constexpr int WSLOT = CBW * 32;                         // elements per ring slot
    constexpr int WU = (CBW * 4 + 255) / 256;               // 16-byte units per thread per slice
    T* wring = reinterpret_cast<T*>(smem_raw + (size_t)HP * CK * sizeof(T) + 4 * 2 * NT * 16 * sizeof(float));

    auto stage = [&](int n, int ty0, int tx0, int cb0) {
        for (int u = tid; u < HP * SPF; u += 256) {
            const int hp = u / SPF, s = u - hp * SPF;
            const int hy = hp / HW_, hx = hp - hy * HW_;
            const int gy = ty0 + hy - 1, gx = tx0 + hx - 1;
            Vec8<T> v;
            if (gy >= 0 && gy < H && gx >= 0 && gx < W)
                v.load(x + ((long long)(n * H + gy) * W + gx) * Cip + cb0 + 8 * s);
            else
                v.zero();
            if constexpr (PRESPLIT) v.presplit();
            v.store(halo + hp * CK + ((s ^ halo_swz(hp, hy, SPF)) << 3));
        }
    };

    for (int tile = blockIdx.x; tile < numTiles; tile += gridDim.x) {
        const int n = tile / (tilesX * tilesY);
        const int trem = tile - n * (tilesX * tilesY);
        const int ty0 = (trem / tilesX) * TH, tx0 = (trem % tilesX) * TW;

        // first channel block: staged while the accumulators are dead
        __syncthreads();
        stage(n, ty0, tx0, 0);
        __syncthreads();

        f32x4 acc[MT][NT];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};

        for (int cblk = 0; cblk < nCblk; ++cblk) {
            if (cblk > 0) {
                __syncthreads();
                stage(n, ty0, tx0, cblk * CK);
                __syncthreads();
            }
            if constexpr (WLDS) {
                constexpr int NSTEP = NCHUNK * 9;
                const T* wsrc = wp + (long long)(blockIdx.y * CBW) * wrow + (long long)(cblk * NCHUNK) * 288;   // step s: + s*32
                Vec8<T> wreg[WU];
                auto wload = [&](int sidx) {
#pragma unroll
                    for (int k = 0; k < WU; ++k) {
                        const int u = tid + k * 256;
                        if (u < CBW * 4) wreg[k].load(wsrc + (long long)(u >> 2) * wrow + sidx * 32 + (u & 3) * 8);
                    }
                };
                auto wstore = [&](int slot) {
#pragma unroll
                    for (int k = 0; k < WU; ++k) {
                        const int u = tid + k * 256;
                        if (u < CBW * 4) {
                            const int col = u >> 2, seg = u & 3, key = (col / (NT * 4)) & 3;
                            const int f = (4 - key) & 3;                                    // F = [0, 3, 2, 1]
                            wreg[k].store(wring + slot * WSLOT + col * 32 + ((seg ^ f) << 3));
                        }
                    }
                };
                wload(0);
                wstore(0);
                if (NSTEP > 1) wload(1);
                const int akey = (4 - (p >> 2)) & 3;
                int aoff[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) aoff[t] = (cb * (NT * 16) + (p >> 2) * (NT * 4) + t * 4 + (p & 3)) * 32 + ((q ^ akey) << 3);
#pragma unroll 1
                for (int chunk = 0; chunk < NCHUNK; ++chunk) {
#pragma unroll
                    for (int tap = 0; tap < 9; ++tap) {
                        const int sidx = chunk * 9 + tap;
                        const int kh = tap / 3, kw = tap % 3;
                        if (sidx + 1 < NSTEP) wstore((sidx + 1) % 3);
                        if (sidx + 2 < NSTEP) wload(sidx + 2);
                        int sw = ((sw_row + kh) & 1) << 1;
                        if (SPF == 8) sw |= (((sw_col + 2 * kh + kw) >> 1) & 1) << 2;
                        if (SPF >= 16) sw |= ((sw_col + 2 * kh + kw) & 3) << 2;
                        const T* bptr = halo + lane_el + (kh * HW_ + kw) * CK + (((chunk * 4 + q) ^ sw) << 3);
                        // the halo image is static: the first half of this tap's patch fragments is requested before the
                        // barrier that publishes the weight slice, so the LDS latency overlaps the barrier wait
                        constexpr int PRE = CK == 64 ? 0 : MT / 2;         // the CK=64 variants have no registers to spare
                        Frag<T> bpre[PRE > 0 ? PRE : 1];
#pragma unroll
                        for (int m = 0; m < PRE; ++m) frag_load(bpre[m], bptr + ((patch_row<PG>(0, m) * 4) * HW_ + patch_col<PG>(m) * 4) * CK);
                        __syncthreads();
                        Frag<T> a[NT];
                        const T* aslot = wring + (sidx % 3) * WSLOT;
#pragma unroll
                        for (int t = 0; t < NT; ++t) frag_load(a[t], aslot + aoff[t]);
#pragma unroll
                        for (int m = 0; m < MT; ++m) {
                            Frag<T> b;
                            if (m < PRE) b = bpre[m];
                            else frag_load(b, bptr + ((patch_row<PG>(0, m) * 4) * HW_ + patch_col<PG>(m) * 4) * CK);
#pragma unroll
                            for (int t = 0; t < NT; ++t) acc[m][t] = mma32(a[t], b, acc[m][t]);
                        }
                    }
                }
            } else
#pragma unroll 1
            for (int chunk = 0; chunk < NCHUNK; ++chunk) {
                const T* wbase[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) wbase[t] = wlane[t] + (long long)(cblk * NCHUNK + chunk) * (9 * 32);
                constexpr int NBUF = (sizeof(T) == 2 || PRESPLIT) ? 2 : 1;      // bf16 / pre-split: prefetch the next tap's weights; exact fp32: single buffer
                Frag<T> a[NBUF][NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) frag_load(a[0][t], wbase[t]);
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const int kh = tap / 3, kw = tap % 3;
                    if (NBUF == 2 && tap + 1 < 9) {
#pragma unroll
                        for (int t = 0; t < NT; ++t) frag_load(a[(tap + 1) % NBUF][t], wbase[t] + (tap + 1) * 32);
                    }
                    if (NBUF == 1 && tap > 0) {
#pragma unroll
                        for (int t = 0; t < NT; ++t) frag_load(a[0][t], wbase[t] + tap * 32);
                    }
                    int sw = ((sw_row + kh) & 1) << 1;
                    if (SPF == 8) sw |= (((sw_col + 2 * kh + kw) >> 1) & 1) << 2;
                    if (SPF >= 16) sw |= ((sw_col + 2 * kh + kw) & 3) << 2;
                    const T* bptr = halo + lane_el + (kh * HW_ + kw) * CK + (((chunk * 4 + q) ^ sw) << 3);
#pragma unroll
                    for (int m = 0; m < MT; ++m) {
                        Frag<T> b;
                        frag_load(b, bptr + ((patch_row<PG>(0, m) * 4) * HW_ + patch_col<PG>(m) * 4) * CK);
#pragma unroll
                        for (int t = 0; t < NT; ++t) acc[m][t] = PRESPLIT ? mma32_pre(a[tap % NBUF][t], b, acc[m][t]) : mma32(a[tap % NBUF][t], b, acc[m][t]);
                    }
                }
            }
        }

        // ---- epilogue: lane holds NT*4 consecutive channels of one pixel per patch
        const bool full = (ty0 + TH <= H) && (tx0 + TW <= W);
        float s1[NT][4], s2[NT][4];
        if (STATS) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) { s1[t][r] = 0.f; s2[t][r] = 0.f; }
        }
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int gy = ty0 + patch_row<PG>(pg, m) * 4 + py, gx = tx0 + patch_col<PG>(m) * 4 + px;
            const bool valid = full || ((gy < H) && (gx < W));
            if (valid) {
                T* dst = y + ((long long)(n * H + gy) * W + gx) * Cop + co_base + q * (NT * 4);
#pragma unroll
                for (int h8 = 0; h8 < NT / 2; ++h8) {
                    Vec8<T> v;
#pragma unroll
                    for (int j = 0; j < 8; ++j) v.set(j, acc[m][h8 * 2 + (j >> 2)][j & 3]);
                    v.store(dst + h8 * 8);
                }
                if (STATS) {
#pragma unroll
                    for (int t = 0; t < NT; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float v = acc[m][t][r];
                            s1[t][r] += v;
                            s2[t][r] = fmaf(v, v, s2[t][r]);
                        }
                }
            }
        }
        if (STATS) {
            // fold the 16 pixel lanes, then one LDS add per channel per wave (fixed wave order is not needed: fp32 adds of
            // per-wave partials commute up to rounding; the cross-workgroup sum below is in a fixed order)
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float a = group16_sum(s1[t][r]), b = group16_sum(s2[t][r]);
                    if (p == 0) {
                        const int cl = q * (NT * 4) + t * 4 + r;
                        wgstat[(wave * 2 + 0) * (NT * 16) + cl] += a;
                        wgstat[(wave * 2 + 1) * (NT * 16) + cl] += b;
                    }
                }
        }
    }

    if (STATS) {
        __syncthreads();
        for (int i = tid; i < 2 * CB * NT * 16; i += 256) {
            const int which = i / (CB * NT * 16), cl = i % (CB * NT * 16);
            const int cbi = cl / (NT * 16), c16 = cl % (NT * 16);
            float acc = 0.f;
#pragma unroll
            for (int g = 0; g < PG; ++g) acc += wgstat[((g * CB + cbi) * 2 + which) * (NT * 16) + c16];     // wave = pg*CB + cb
            stats[((long long)blockIdx.x * 2 + which) * Cop + blockIdx.y * (CB * NT * 16) + cl] = acc;
        }
    }
}

// First stage: x is the user's NCHW fp32 clip tensor [N,Ci,H,W], Ci <= 3, K = 9*Ci <= 27 padded to 32.
// Bandwidth-shaped: one MFMA k-step per patch; the halo tile is staged as fp32 planes (coalesced along W)
// and the patch fragment is gathered from LDS (k = tap*Ci + ci), converted to T in registers.
template <typename T, int NT, bool STATS>
__global__ __launch_bounds__(256) void conv3x3_first_kernel(const float* __restrict__ x, const T* __restrict__ wp,
                                                            T* __restrict__ y, float* __restrict__ stats,
                                                            int N, int H, int W, int Ci, int Cop,
                                                            int tilesX, int tilesY, int numTiles) {
    constexpr int PG = 4;
    constexpr int PHP = TileGeom<PG>::PHP, PWP = TileGeom<PG>::PWP;
    constexpr int TH = 4 * PHP, TW = 4 * PWP, HH = TH + 2, HW_ = TW + 2, HP = HH * HW_;
    constexpr int MT = 8;
    __shared__ float halo[3 * HP + 4];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int pg = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int p = lane & 15, q = lane >> 4, py = p >> 2, px = p & 3;
    const int co_base = blockIdx.y * (NT * 16);
    const int zero_idx = 3 * HP;
    if (tid == 0) halo[zero_idx] = 0.f;

    Frag<T> a[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
        frag_load(a[t], wp + (long long)(co_base + (p >> 2) * (NT * 4) + t * 4 + (p & 3)) * 32 + 8 * q);

    int koff[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = 8 * q + j;
        if (k < 9 * Ci) {
            const int tap = k / Ci, ci = k - tap * Ci;
            koff[j] = ci * HP + (tap / 3) * HW_ + (tap % 3);
        } else {
            koff[j] = -1;
        }
    }
    int hbase[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int pidx = pg * MT + m;
        hbase[m] = ((pidx / PWP) * 4 + py) * HW_ + (pidx % PWP) * 4 + px;
    }
    float s1[NT][4], s2[NT][4];
    if (STATS) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) { s1[t][r] = 0.f; s2[t][r] = 0.f; }
    }

    for (int tile = blockIdx.x; tile < numTiles; tile += gridDim.x) {
        const int n = tile / (tilesX * tilesY);
        const int trem = tile - n * (tilesX * tilesY);
        const int ty0 = (trem / tilesX) * TH, tx0 = (trem % tilesX) * TW;
        __syncthreads();
        for (int u = tid; u < Ci * HP; u += 256) {
            const int ci = u / HP, hp = u - ci * HP;
            const int hy = hp / HW_, hx = hp - hy * HW_;
            const int gy = ty0 + hy - 1, gx = tx0 + hx - 1;
            float v = 0.f;
            if (gy >= 0 && gy < H && gx >= 0 && gx < W) v = x[((long long)(n * Ci + ci) * H + gy) * W + gx];
            halo[u] = v;
        }
        __syncthreads();
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            Frag<T> b;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int idx = koff[j] >= 0 ? koff[j] + hbase[m] : zero_idx;
                frag_set<T>(b, j, halo[idx]);
            }
            f32x4 acc[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = mma32(a[t], b, f32x4{0.f, 0.f, 0.f, 0.f});

            const int pidx = pg * MT + m;
            const int gy = ty0 + (pidx / PWP) * 4 + py, gx = tx0 + (pidx % PWP) * 4 + px;
            const bool valid = (gy < H) && (gx < W);
            if (valid) {
                T* dst = y + ((long long)(n * H + gy) * W + gx) * Cop + co_base + q * (NT * 4);
#pragma unroll
                for (int h8 = 0; h8 < NT / 2; ++h8) {
                    Vec8<T> v;
#pragma unroll
                    for (int j = 0; j < 8; ++j) v.set(j, acc[h8 * 2 + (j >> 2)][j & 3]);
                    v.store(dst + h8 * 8);
                }
            }
            if (STATS) {
                const float vm = valid ? 1.f : 0.f;
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float v = acc[t][r] * vm;
                        s1[t][r] += v;
                        s2[t][r] += v * v;
                    }
            }
        }
    }
    if (STATS) {
        __syncthreads();
        float* red = halo;                                         // [4 waves][2][NT*16]
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float sa = group16_sum(s1[t][r]), sb = group16_sum(s2[t][r]);
                if (p == 0) {
                    const int cl = q * (NT * 4) + t * 4 + r;
                    red[(pg * 2 + 0) * (NT * 16) + cl] = sa;
                    red[(pg * 2 + 1) * (NT * 16) + cl] = sb;
                }
            }
        __syncthreads();
        for (int i = tid; i < 2 * NT * 16; i += 256) {
            const int cl = i % (NT * 16), which = i / (NT * 16);
            float acc = 0.f;
#pragma unroll
            for (int g = 0; g < PG; ++g) acc += red[(g * 2 + which) * (NT * 16) + cl];
            stats[((long long)blockIdx.x * 2 + which) * Cop + co_base + cl] = acc;
        }
    }
}

// one packed element; split-bf16 build with fp32 storage: the 3x3 layouts (not the first layer's) are stored pre-split for conv3x3_nhwc_kernel
__device__ __forceinline__ void pack_store(bf16* wp, long long i, float v, bool) { wp[i] = (bf16)v; }
__device__ __forceinline__ void pack_store(float* wp, long long i, float v, bool presplit) {
    if (HYB_X3 && presplit) hyb_presplit_store(wp, i, v);
    else wp[i] = v;
}

// w fp32 [Co,Ci,3,3] -> packed T (see hybrid_hip.h for the three modes); padded rows/cols are zero.
template <typename T>
__global__ void pack_weight_kernel(int mode, const float* __restrict__ w, T* __restrict__ wp, int Co, int Ci, int Cop, int Cip,
                                   long long total) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    float v = 0.f;
    if (mode == 0) {            // [Cop][Cip/32][9][32]
        const int c32 = (int)(i % 32);
        const int tap = (int)((i / 32) % 9);
        const int chunk = (int)((i / 288) % (Cip / 32));
        const int co = (int)(i / ((long long)9 * Cip));
        const int ci = chunk * 32 + c32;
        if (co < Co && ci < Ci) v = w[((long long)co * Ci + ci) * 9 + tap];
    } else if (mode == 1) {     // dgrad: rows = input channels: [Cip][Cop/32][9][32], tap flipped
        const int c32 = (int)(i % 32);
        const int tap = (int)((i / 32) % 9);
        const int chunk = (int)((i / 288) % (Cop / 32));
        const int ci = (int)(i / ((long long)9 * Cop));
        const int co = chunk * 32 + c32;
        if (co < Co && ci < Ci) v = w[((long long)co * Ci + ci) * 9 + (8 - tap)];
    } else {                    // first layer: [Cop][32], k = tap*Ci + ci
        const int k = (int)(i % 32);
        const int co = (int)(i / 32);
        if (co < Co && k < 9 * Ci) { const int tap = k / Ci, ci = k - tap * Ci; v = w[((long long)co * Ci + ci) * 9 + tap]; }
    }
    pack_store(wp, i, v, mode != 2);
}

// stats[2][Cop] = sum over the G per-workgroup partial rows, fixed order
__global__ __launch_bounds__(1024) void stats_reduce_kernel(const float* __restrict__ part, float* __restrict__ stats, int G, int n) {
    long long i; float v;
    if (rows_reduce_1024(part, G, n, i, v)) stats[i] = v;
}

constexpr int MAX_STAT_PARTIALS = 512;
}  // namespace
int hyb_conv_v2(const void* x, const void* wp, void* y, float* part, int N, int H, int W, int Cip, int Cop, int stat_rows, hipStream_t st,
                long long xblk = 0);   // conv_v2.hip
int hyb_conv_v2_supported(int W, int Cip, int Cop);
namespace {

template <typename T, int NT, int CB, int PG, int CK, bool WLDS = false>
int launch_conv_ck(const T* x, const T* wp, T* y, float* stats, float* part, int N, int H, int W, int Cip, int Cop, hipStream_t st) {
    constexpr int TH = 4 * TileGeom<PG>::PHP, TW = 4 * TileGeom<PG>::PWP, HP = (TH + 2) * (TW + 2);
    const size_t lds = (size_t)HP * CK * sizeof(T) + 4 * 2 * NT * 16 * sizeof(float) + (WLDS ? 3 * (size_t)CB * NT * 16 * 32 * sizeof(T) : 0);
    const int tilesX = hyb_cdiv(W, TW), tilesY = hyb_cdiv(H, TH);
    const long long numTiles = (long long)N * tilesX * tilesY;
    const int gy = Cop / (CB * NT * 16);
    int gx = (int)(numTiles < MAX_STAT_PARTIALS ? numTiles : MAX_STAT_PARTIALS);
    if (gx < 1) gx = 1;
    dim3 grid(gx, gy);
    if (lds > 64 * 1024) {
        static HybAttrOnce once_stats, once_plain;             // per template instantiation (lds is fixed by it), per device
        if (int e = part ? hyb_set_lds_attr(once_stats, (const void*)conv3x3_nhwc_kernel<T, NT, CB, PG, CK, true, WLDS>, (int)lds)
                         : hyb_set_lds_attr(once_plain, (const void*)conv3x3_nhwc_kernel<T, NT, CB, PG, CK, false, WLDS>, (int)lds)) return e;
    }
    HybProfileHook* hook = hyb_find_hook(1, Cip, Cop);
    if (hook) hipEventRecord(hook->ev0, st);
    if (part) {
        hipLaunchKernelGGL((conv3x3_nhwc_kernel<T, NT, CB, PG, CK, true, WLDS>), grid, dim3(256), lds, st, x, wp, y, part, N, H, W, Cip, Cop,
                           tilesX, tilesY, (int)numTiles);
        if (hook) hipEventRecord(hook->ev1, st);
        HYB_LAUNCH_CHECK();
        if (stats) hipLaunchKernelGGL(stats_reduce_kernel, dim3(hyb_cdiv(2 * Cop, 32)), dim3(1024), 0, st, part, stats, gx, 2 * Cop);
    } else {
        hipLaunchKernelGGL((conv3x3_nhwc_kernel<T, NT, CB, PG, CK, false, WLDS>), grid, dim3(256), lds, st, x, wp, y, stats, N, H, W, Cip, Cop,
                           tilesX, tilesY, (int)numTiles);
        if (hook) hipEventRecord(hook->ev1, st);
    }
    HYB_LAUNCH_CHECK();
    return 0;
}

// CK = channels staged per LDS halo image: the largest of {128, 64, 32} that divides Cip and keeps the image <= ~48 KB (bf16)
template <typename T, int NT, int CB, int PG>
int launch_conv(const T* x, const T* wp, T* y, float* stats, float* part, int N, int H, int W, int Cip, int Cop, hipStream_t st) {
    constexpr int ES = (int)sizeof(T);
    if constexpr (ES == 2) {
        static const int wlds = getenv("HYB_CONV_WLDS") ? atoi(getenv("HYB_CONV_WLDS")) : 1;
        if (wlds) {     // default: weight fragments come from an LDS ring shared by the workgroup (HYB_CONV_WLDS=0: per-wave global loads)
            if constexpr (PG == 1) { if (Cip % 64 == 0) return launch_conv_ck<T, NT, CB, PG, 64, true>(x, wp, y, stats, part, N, H, W, Cip, Cop, st); }
            if constexpr (PG == 2) { if (Cip % 64 == 0) return launch_conv_ck<T, NT, CB, PG, 64, true>(x, wp, y, stats, part, N, H, W, Cip, Cop, st); }
            return launch_conv_ck<T, NT, CB, PG, 32, true>(x, wp, y, stats, part, N, H, W, Cip, Cop, st);
        }
    }
    if constexpr (PG == 1) {
        if (Cip % 128 == 0 && ES == 2) return launch_conv_ck<T, NT, CB, PG, 128>(x, wp, y, stats, part, N, H, W, Cip, Cop, st);
        if (Cip % 64 == 0) return launch_conv_ck<T, NT, CB, PG, 64>(x, wp, y, stats, part, N, H, W, Cip, Cop, st);
    }
    if constexpr (PG == 2) {
        if (Cip % 64 == 0 && ES == 2) return launch_conv_ck<T, NT, CB, PG, 64>(x, wp, y, stats, part, N, H, W, Cip, Cop, st);
    }
    return launch_conv_ck<T, NT, CB, PG, 32>(x, wp, y, stats, part, N, H, W, Cip, Cop, st);
}

template <typename T>
int conv_fwd_t(int first, const void* x, const void* wp, void* y, float* stats, float* part, int N, int H, int W, int Ci, int Cip, int Cop,
               hipStream_t st) {
    if (first) {
        HYB_CHECK_ARG(Ci >= 1 && Ci <= 3);
        constexpr int TH = 16, TW = 32;
        const int tilesX = hyb_cdiv(W, TW), tilesY = hyb_cdiv(H, TH);
        const long long numTiles = (long long)N * tilesX * tilesY;
        const int nt = (Cop % 64 == 0) ? 4 : 2;
        const int gy = Cop / (nt * 16);
        int gx = (int)(numTiles < MAX_STAT_PARTIALS ? numTiles : MAX_STAT_PARTIALS);
        if (gx < 1) gx = 1;
            dim3 grid(gx, gy);
#define HYB_FIRST(NT_, ST_) hipLaunchKernelGGL((conv3x3_first_kernel<T, NT_, ST_>), grid, dim3(256), 0, st, (const float*)x, (const T*)wp, \
                                               (T*)y, part, N, H, W, Ci, Cop, tilesX, tilesY, (int)numTiles)
        if (nt == 4) { if (part) HYB_FIRST(4, true); else HYB_FIRST(4, false); }
        else         { if (part) HYB_FIRST(2, true); else HYB_FIRST(2, false); }
#undef HYB_FIRST
        HYB_LAUNCH_CHECK();
        if (stats) {
            hipLaunchKernelGGL(stats_reduce_kernel, dim3(hyb_cdiv(2 * Cop, 32)), dim3(1024), 0, st, part, stats, gx, 2 * Cop);
            HYB_LAUNCH_CHECK();
        }
        return 0;
    }
    HYB_CHECK_ARG(Cip % 32 == 0);
    if constexpr (sizeof(T) == 2) {
        // bf16: the asynchronous kernel (conv_v2.hip) takes every shape it has a variant for; HYB_CONV_V2=0 keeps the first-generation kernel
        static const int v2 = getenv("HYB_CONV_V2") ? atoi(getenv("HYB_CONV_V2")) : 1;
        if (v2) {
            const int rows = part ? hyb_conv_stats_rows(0, N, H, W, Cop) : 0;
            HybProfileHook* hook = hyb_find_hook(1, Cip, Cop);
            if (hook) hipEventRecord(hook->ev0, st);
            const int rc = hyb_conv_v2(x, wp, y, part, N, H, W, Cip, Cop, rows, st);
            if (rc != -100) {
                if (hook) hipEventRecord(hook->ev1, st);
                if (rc) return rc;
                if (part && stats) {
                    hipLaunchKernelGGL(stats_reduce_kernel, dim3(hyb_cdiv(2 * Cop, 32)), dim3(1024), 0, st, part, stats, rows, 2 * Cop);
                    HYB_LAUNCH_CHECK();
                }
                return 0;
            }
        }
    }
    static const int cfg = getenv("HYB_CONV_CFG") ? atoi(getenv("HYB_CONV_CFG")) : 0;
    if (cfg != 2 && Cop % 256 == 0) return launch_conv<T, 4, 4, 1>((const T*)x, (const T*)wp, (T*)y, stats, part, N, H, W, Cip, Cop, st);
    if (Cop % 128 == 0) return launch_conv<T, 4, 2, 2>((const T*)x, (const T*)wp, (T*)y, stats, part, N, H, W, Cip, Cop, st);
    if (Cop % 64 == 0) return launch_conv<T, 4, 1, 4>((const T*)x, (const T*)wp, (T*)y, stats, part, N, H, W, Cip, Cop, st);
    return launch_conv<T, 2, 1, 4>((const T*)x, (const T*)wp, (T*)y, stats, part, N, H, W, Cip, Cop, st);
}

}  // namespace

// Internal (hyb_convstage_bwd): does the bf16 dgrad conv of this shape run on the asynchronous kernel (which can read a block-planar input)?
int hyb_conv_dgrad_planar_ok(int dtype, int W, int Cin_p, int Cout_p) {
    static const int v2 = getenv("HYB_CONV_V2") ? atoi(getenv("HYB_CONV_V2")) : 1;
    return dtype == HYB_BF16 && v2 && hyb_conv_v2_supported(W, Cin_p, Cout_p);
}
// Internal: conv3x3 (dgrad weights) of a block-planar bf16 input [Cin_p/32][N][H][W][32] -> NHWC output
int hyb_conv3x3_planar_in(const void* x, const void* wp, void* y, int N, int H, int W, int Cin_p, int Cout_p, hipStream_t st) {
    HybProfileHook* hook = hyb_find_hook(1, Cin_p, Cout_p);
    if (hook) hipEventRecord(hook->ev0, st);
    const int rc = hyb_conv_v2(x, wp, y, nullptr, N, H, W, Cin_p, Cout_p, 0, st, (long long)N * H * W * 32);
    if (hook) hipEventRecord(hook->ev1, st);
    return rc;
}

extern "C" long long hyb_conv_packed_elems(int first, int Cip, int Cop) {
    return first ? (long long)Cop * 32 : (long long)Cop * 9 * Cip;
}

// forward (mode 0) and dgrad (mode 1) layouts in ONE launch (internal; used by hyb_convstage_fwd)
template <typename T>
__global__ void pack_weight_dual_kernel(const float* __restrict__ w, T* __restrict__ wp0, T* __restrict__ wp1, int Co, int Ci, int Cop, int Cip,
                                        long long count) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 2 * count) return;
    const bool second = i >= count;
    const long long k = second ? i - count : i;
    const int c32 = (int)(k % 32);
    const int tap = (int)((k / 32) % 9);
    float v = 0.f;
    if (!second) {
        const int chunk = (int)((k / 288) % (Cip / 32));
        const int co = (int)(k / ((long long)9 * Cip));
        const int ci = chunk * 32 + c32;
        if (co < Co && ci < Ci) v = w[((long long)co * Ci + ci) * 9 + tap];
        pack_store(wp0, k, v, true);
    } else {
        const int chunk = (int)((k / 288) % (Cop / 32));
        const int ci = (int)(k / ((long long)9 * Cop));
        const int co = chunk * 32 + c32;
        if (co < Co && ci < Ci) v = w[((long long)co * Ci + ci) * 9 + (8 - tap)];
        pack_store(wp1, k, v, true);
    }
}
int hyb_conv_pack_weight_dual(int dtype, const float* w, void* wp0, void* wp1, int Co, int Ci, int Cop, int Cip, hipStream_t st) {
    const long long count = (long long)Cop * 9 * Cip;
    const int blocks = hyb_cdiv(2 * count, 256);
    if (dtype == HYB_F32) hipLaunchKernelGGL(pack_weight_dual_kernel<float>, dim3(blocks), dim3(256), 0, st, w, (float*)wp0, (float*)wp1, Co, Ci, Cop, Cip, count);
    else if (dtype == HYB_BF16) hipLaunchKernelGGL(pack_weight_dual_kernel<bf16>, dim3(blocks), dim3(256), 0, st, w, (bf16*)wp0, (bf16*)wp1, Co, Ci, Cop, Cip, count);
    else return HYB_E_ARG;
    HYB_LAUNCH_CHECK();
    return 0;
}

// The same for SEVERAL stages in one launch (internal; hyb_backbone_fwd): blockIdx.y = stage.  The weights do not depend on the
// activations, so the whole backbone's packs can run before its first convolution (three 5 us launches fewer per step at config 2).
struct PackMany {
    const float* w[16]; void* wp0[16]; void* wp1[16]; int Co[16], Ci[16], Cop[16], Cip[16];
    // + the first stage's [2][Cop][64] pack (conv_first.h) as row blockIdx.y = n when s1_wp != NULL
    const float* s1_w; void* s1_wp; int s1_Co, s1_Ci, s1_Cop; int n;
};
template <typename T>
__global__ void pack_weight_many_kernel(PackMany a) {
    const int s = blockIdx.y;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (s == a.n) {                                       // first stage (uniform per workgroup)
        const long long total = (long long)a.s1_Cop * 64;
        if (i >= 2 * total) return;
        const bool second = i >= total;
        ((T*)a.s1_wp)[i] = from_f32<T>(s1w_pack_value(a.s1_w, second ? i - total : i, second, a.s1_Co, a.s1_Ci));
        return;
    }
    const long long count = (long long)a.Cop[s] * 9 * a.Cip[s];
    if (i >= 2 * count) return;
    const float* __restrict__ w = a.w[s];
    const int Co = a.Co[s], Ci = a.Ci[s], Cop = a.Cop[s], Cip = a.Cip[s];
    const bool second = i >= count;
    const long long k = second ? i - count : i;
    const int c32 = (int)(k % 32);
    const int tap = (int)((k / 32) % 9);
    float v = 0.f;
    if (!second) {
        const int chunk = (int)((k / 288) % (Cip / 32));
        const int co = (int)(k / ((long long)9 * Cip));
        const int ci = chunk * 32 + c32;
        if (co < Co && ci < Ci) v = w[((long long)co * Ci + ci) * 9 + tap];
        pack_store((T*)a.wp0[s], k, v, true);
    } else {
        const int chunk = (int)((k / 288) % (Cop / 32));
        const int ci = (int)(k / ((long long)9 * Cop));
        const int co = chunk * 32 + c32;
        if (co < Co && ci < Ci) v = w[((long long)co * Ci + ci) * 9 + (8 - tap)];
        pack_store((T*)a.wp1[s], k, v, true);
    }
}
// s1_wp != NULL: also the first stage's two layouts (s1_w [s1_Co][s1_Ci][3][3] -> s1_wp [2][s1_Cop][64])
int hyb_conv_pack_weight_many(int dtype, int n, const float* const* w, void* const* wp0, void* const* wp1, const int* Co, const int* Ci, const int* Cop,
                              const int* Cip, const float* s1_w, void* s1_wp, int s1_Co, int s1_Ci, int s1_Cop, hipStream_t st) {
    if (n < 0 || n > 16 || (n == 0 && !s1_wp)) return HYB_E_ARG;
    PackMany a{};
    long long maxc = 0;
    for (int i = 0; i < n; ++i) {
        a.w[i] = w[i]; a.wp0[i] = wp0[i]; a.wp1[i] = wp1[i]; a.Co[i] = Co[i]; a.Ci[i] = Ci[i]; a.Cop[i] = Cop[i]; a.Cip[i] = Cip[i];
        const long long c = (long long)Cop[i] * 9 * Cip[i];
        if (c > maxc) maxc = c;
    }
    a.n = n; a.s1_w = s1_w; a.s1_wp = s1_wp; a.s1_Co = s1_Co; a.s1_Ci = s1_Ci; a.s1_Cop = s1_Cop;
    if (s1_wp && (long long)s1_Cop * 64 > maxc) maxc = (long long)s1_Cop * 64;
    const dim3 grid(hyb_cdiv(2 * maxc, 256), n + (s1_wp ? 1 : 0));
    if (dtype == HYB_F32) hipLaunchKernelGGL(pack_weight_many_kernel<float>, grid, dim3(256), 0, st, a);
    else if (dtype == HYB_BF16) hipLaunchKernelGGL(pack_weight_many_kernel<bf16>, grid, dim3(256), 0, st, a);
    else return HYB_E_ARG;
    HYB_LAUNCH_CHECK();
    return 0;
}

extern "C" int hyb_conv_pack_weight(int dtype, int mode, const float* w, void* wp, int Co, int Ci, int Cop, int Cip, void* stream) {
    HYB_CHECK_ARG(w && wp && Co > 0 && Ci > 0 && Cop % 32 == 0 && Cop >= Co && mode >= 0 && mode <= 2);
    if (mode == 2) HYB_CHECK_ARG(Ci <= 3);
    else HYB_CHECK_ARG(Cip % 32 == 0 && Cip >= Ci);
    const long long total = mode == 2 ? (long long)Cop * 32 : (long long)Cop * 9 * Cip;
    const int blocks = hyb_cdiv(total, 256);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == HYB_F32) hipLaunchKernelGGL(pack_weight_kernel<float>, dim3(blocks), dim3(256), 0, st, mode, w, (float*)wp, Co, Ci, Cop, Cip, total);
    else if (dtype == HYB_BF16) hipLaunchKernelGGL(pack_weight_kernel<bf16>, dim3(blocks), dim3(256), 0, st, mode, w, (bf16*)wp, Co, Ci, Cop, Cip, total);
    else return HYB_E_ARG;
    HYB_LAUNCH_CHECK();
    return 0;
}

extern "C" int hyb_conv_stats_rows(int first, int N, int H, int W, int Cop) {
    if (N <= 0 || H <= 0 || W <= 0 || Cop <= 0) return HYB_E_ARG;
    int th, tw;
    if (first || Cop % 128 != 0) { th = 16; tw = 32; }            // PG = 4 tiles
    else if (Cop % 256 == 0) { th = 8; tw = 16; }                   // PG = 1
    else { th = 16; tw = 16; }                                      // PG = 2
    const long long numTiles = (long long)N * hyb_cdiv(W, tw) * hyb_cdiv(H, th);
    return (int)(numTiles < MAX_STAT_PARTIALS ? numTiles : MAX_STAT_PARTIALS);
}

extern "C" size_t hyb_conv_stats_workspace(int Cop) { return Cop > 0 ? (size_t)MAX_STAT_PARTIALS * 2 * Cop * sizeof(float) : 0; }

extern "C" int hyb_conv3x3_fwd(int dtype, int first, const void* x, const void* wp, void* y, float* stats, float* stats_partials, int N,
                               int H, int W, int Ci, int Cip, int Cop, void* stream) {
    HYB_CHECK_ARG(x && wp && y && N > 0 && H > 0 && W > 0 && Cop > 0 && Cop % 32 == 0);
    HYB_CHECK_ARG(!stats || stats_partials);
    float* part = stats_partials;          // partial rows are produced whenever a buffer is given; `stats` adds the final reduce
    HYB_CHECK_ARG((long long)N * H * W * (Cop > Cip ? Cop : Cip) < (1ll << 40));
    hipStream_t st = (hipStream_t)stream;
    if (dtype == HYB_F32) return conv_fwd_t<float>(first, x, wp, y, stats, part, N, H, W, Ci, Cip, Cop, st);
    if (dtype == HYB_BF16) return conv_fwd_t<bf16>(first, x, wp, y, stats, part, N, H, W, Ci, Cip, Cop, st);
    return HYB_E_ARG;
}
