// Stage 1, wave-private forward passes (see conv_first.hip for the stage's overall scheme and the block-level kernels).
// This file is compiled with -mllvm -amdgpu-mfma-vgpr-form (build.py): every accumulator value is consumed by vector instructions,
// which cannot read AGPRs, and the kernels are bound by VALU instruction issue -- one v_accvgpr_read per value otherwise.
#include <stdlib.h>
#include "conv_first.h"

namespace {

// ---- wave-private forward passes (second generation of MODE 0 / MODE 1) ---------------------------------------------------------
// Measured on the block-level kernel above: it is bound by VALU *instruction issue*, not by memory, LDS or the matrix cores -- a timing
// build with loads, LDS traffic, MFMAs and epilogue all removed still took 2/3 of the time (tile decode with float reciprocals, 64-bit
// address arithmetic and bounds tests per channel, accumulator copies).  This version is built around the instruction count:
//   * every WAVE is independent (no barrier in the loop): it owns a run of 8x16-pixel blocks, stages its own 10x24-pixel halo image
//     (double-buffered in its private LDS slice; the next block's fp32 rows are in flight in registers meanwhile);
//   * block coordinates advance by scalar increments; global addresses are "uniform block offset + per-lane constant" through buffer
//     loads/stores (1 VALU add per block; lanes outside the image get an out-of-range offset and the hardware returns zeros / drops the
//     store), so padding and ragged edges cost no per-channel tests;
//   * LDS fragment addresses are per-lane constants (+ immediates); the k-step-1 fragment reuses the registers of the k-step-0 one
//     (its extra K slots meet zero weights); the 2x2 max + ReLU is two v_max3, statistics use packed fp32 adds/fmas.
// Same fragment scheme, K order, packed weights, partial-row layout and results as the block-level kernel, which remains the path for
// row strides that are not a multiple of 4 floats and for tensors of 4 GiB and more (32-bit buffer offsets).
constexpr int S1W_BW = 16;                       // block width: two 8x8 sub-blocks side by side
// LDS image layout: rows come in PAIRS -- row 2i at pixel PAIR * i, row 2i+1 at PAIR * i + ODD (24 pixels of 8 / 16 bytes used per
// row; both offsets even so that rows stay 16-byte aligned for the staging stores).  With 8-byte pixel quads no linear row stride keeps
// the 2-pixel-strided fragment reads, the transposing reads and the staging stores off each other's banks; a bank model of the access
// patterns prefers 88 / 42 to the linear 28-pixel stride (= 56 / 28) by 13 %.  Measured (same box, alternating builds): 1.4549 vs
// 1.4580 ms per step -- inside the noise: LDS conflicts are not what these kernels wait for.  A row index that is "lane part + even
// compile-time part" still splits into a per-lane constant plus an immediate: rowoff(a + 2e) = rowoff(a) + PAIR * e.
#ifndef S1W_PAIR_PX
#define S1W_PAIR_PX 88          /* -DS1W_PAIR_PX=56 -DS1W_ODD_PX=28 is the linear 28-pixel stride (A/B: scripts/s1_layout_ab.sh) */
#define S1W_ODD_PX 42
#endif
constexpr int S1W_PAIR = S1W_PAIR_PX, S1W_ODD = S1W_ODD_PX;
__host__ __device__ constexpr int s1w_rowoff(int r) { return (r >> 1) * S1W_PAIR + (r & 1) * S1W_ODD; }
constexpr int S1W_ROWS = 10;
constexpr int S1W_IMG = (S1W_ROWS / 2) * S1W_PAIR * 4;   // elements per buffer
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void s1w_store8(__amdgpu_buffer_rsrc_t rs, unsigned voff, const Vec8<bf16>& o) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o.v), rs, voff, 0, 0);
}
__device__ __forceinline__ void s1w_store8(__amdgpu_buffer_rsrc_t rs, unsigned voff, const Vec8<float>& o) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o.a), rs, voff, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o.b), rs, voff, 16, 0);      // soffset is outside the range check: an out-of-range voff stays out of range
}

// pre-split forms (split-bf16 build, hyb_common.h): a whole fragment in place; two pre-split pixel quads -> one pre-split fragment
__device__ __forceinline__ void s1w_presplit_frag(Frag<float>& f) {
    f32x4 a = {f.v[0], f.v[1], f.v[2], f.v[3]}, b = {f.v[4], f.v[5], f.v[6], f.v[7]};
    hyb_presplit8(a, b);
#pragma unroll
    for (int j = 0; j < 4; ++j) { f.v[j] = a[j]; f.v[4 + j] = b[j]; }
}
__device__ __forceinline__ void s1w_presplit_frag(Frag<bf16>&) {}
__device__ __forceinline__ void s1w_quads_to_frag(Frag<float>& f, const Quad<float>& q0, const Quad<float>& q1) {
    f.v[0] = q0.v[0]; f.v[1] = q0.v[1]; f.v[2] = q1.v[0]; f.v[3] = q1.v[1];      // hi halves of the two pixels
    f.v[4] = q0.v[2]; f.v[5] = q0.v[3]; f.v[6] = q1.v[2]; f.v[7] = q1.v[3];      // lo halves
}
__device__ __forceinline__ void s1w_quads_to_frag(Frag<bf16>&, const Quad<bf16>&, const Quad<bf16>&) {}

#ifndef S1W_SB_UNROLL
#define S1W_SB_UNROLL 2
#endif
#ifndef S1W_MIN_WAVES
#define S1W_MIN_WAVES 1
#endif
template <typename T, int NT, int MODE, int CI /* 3: RGB frames (no per-channel tests); 0: any Ci <= 4 */, bool ROUTE = false /* MODE 1: also write a.route */>
__global__ __launch_bounds__(256, (NT == 2 && sizeof(T) == 2) ? S1W_MIN_WAVES : 1) void stage1w_kernel(S1Args a) {
    static_assert(MODE == 0 || MODE == 1, "forward passes only");
    static_assert(!ROUTE || (MODE == 1 && sizeof(T) == 2), "routing codes: apply pass, 16-bit storage");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    T* img0 = reinterpret_cast<T*>(smem_raw) + wave * (2 * S1W_IMG);           // this wave's two image buffers
    float* wgstat = reinterpret_cast<float*>(reinterpret_cast<T*>(smem_raw) + 4 * 2 * S1W_IMG);      // [4][2][NT*16] (mode 0)
    const int p = lane & 15, q = lane >> 4, wy = p >> 2, wx = p & 3;
    const int co_base = blockIdx.y * (NT * 16);
    const int H = a.H, W = a.W, Ci = a.Ci, Cop = a.Cop;
    const int Ho = H >> 1, Wo = W >> 1;
    const T* wp = (const T*)a.wp2;                                             // the wave kernels' own K order (s1w_pack_kernel)
    // split-bf16 build, fp32 storage: the image holds pre-split pixel quads (hyb_presplit4, written once when a pixel is staged; every
    // pixel feeds ~6 fragments) and the weight fragments are split once per kernel: no conversions in the block loop
    constexpr bool PRESPLIT = HYB_X3 && sizeof(T) == 4;

    Frag<T> w0[NT], w1[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const T* row = wp + (long long)(co_base + (p >> 2) * (NT * 4) + t * 4 + (p & 3)) * S1_KP + 8 * q;
        frag_load(w0[t], row);
        frag_load(w1[t], row + 32);
        if constexpr (PRESPLIT) { s1w_presplit_frag(w0[t]); s1w_presplit_frag(w1[t]); }
    }
    f32x2 c_sc[NT][2], c_sh[NT][2];
    if (MODE == 1) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ch = co_base + q * (NT * 4) + t * 4 + r;
                c_sc[t][r >> 1][r & 1] = a.ss[ch]; c_sh[t][r >> 1][r & 1] = a.ss[Cop + ch];
            }
    }
    f32x2 acc1[NT][2], acc2[NT][2];
    if (MODE == 0) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int h = 0; h < 2; ++h) { acc1[t][h] = f32x2{0.f, 0.f}; acc2[t][h] = f32x2{0.f, 0.f}; }
    }

    // this wave's run of blocks (row-major inside an image: neighbours share halo columns in L2)
    const int bx_n = a.tilesX, by_n = a.tilesY, bpi = bx_n * by_n;
    const int nwaves = (int)gridDim.x * 4, gw = (int)blockIdx.x * 4 + wave;
    const int chunk = (a.numTiles + nwaves - 1) / nwaves;
    const int blk_begin = gw * chunk < a.numTiles ? gw * chunk : a.numTiles;
    const int blk_end = blk_begin + chunk < a.numTiles ? blk_begin + chunk : a.numTiles;
    const int nblk = blk_end - blk_begin;

    // ---- per-lane constants ----
    // halo: lane u < 60 owns (row = u / 6, segment = u % 6) = 4 consecutive pixels x up to 4 channels
    const int hrow = lane / 6, hseg = lane - hrow * 6;
    const bool hlane = lane < 60;
    const unsigned ld_lane = (unsigned)((hrow * W + 4 * hseg) * 4);                         // bytes from the block's halo origin
    const unsigned st_lds = (unsigned)((s1w_rowoff(hrow) + 4 * hseg) * 4);                      // element offset of this lane's 4 pixels in an image buffer
    const unsigned plane = (unsigned)(H * W) * 4u;                                          // channel plane in bytes
    const __amdgpu_buffer_rsrc_t xrs = hyb_rsrc(a.x, (unsigned)((long long)a.N * Ci * H * W * 4));
    const unsigned OOB = 0xFFFFFFF0u;
    // fragment reads (K order of s1w_pack_kernel): a lane's 8 K elements are TWO HORIZONTALLY ADJACENT pixels x 4 channels, i.e. one
    // 16-byte run of the image -- a single ds_read2_b64 straight into the MFMA operand registers, no assembly moves.
    //   k-step 0: q = 0,1,2 -> taps (q,0),(q,1);  q = 3 -> tap (0,2) + the pixel right of the window (zero weights)
    //   k-step 1: q = 0 -> tap (1,2), q = 1 -> tap (2,2) (+ zero-weight neighbours); q = 2,3 read q = 1's address (zero weights)
    int f_k0[2], f_k1[2];                                                      // [jy]: window rows 2 wy + jy
#pragma unroll
    for (int jy = 0; jy < 2; ++jy) {
        f_k0[jy] = (s1w_rowoff(2 * wy + jy + (q < 3 ? q : 0)) + 2 * wx + 3 + (q < 3 ? 0 : 2)) * 4;
        f_k1[jy] = (s1w_rowoff(2 * wy + jy + (q == 0 ? 1 : 2)) + 2 * wx + 3 + 2) * 4;
    }
    // pooled store (MODE 1): bytes from the block's first window, for sub-block 0
    const unsigned po_lane = (unsigned)(((wy * Wo + wx) * Cop + co_base + q * (NT * 4)) * (int)sizeof(T));
    const __amdgpu_buffer_rsrc_t prs = hyb_rsrc(MODE == 1 ? a.pooled : (void*)a.x, MODE == 1 ? (unsigned)((long long)a.N * Ho * Wo * Cop * (int)sizeof(T)) : 16u);
    // routing codes: one word per 8 pooled channels = the pooled tensor's byte offset / 4 (16-bit storage)
    const __amdgpu_buffer_rsrc_t rrs = hyb_rsrc(ROUTE ? (void*)a.route : (void*)a.x, ROUTE ? (unsigned)((long long)a.N * Ho * Wo * Cop / 2) : 16u);

    // block coordinates: scalar counters, advanced by increments (one division per wave, here)
    int n_c = blk_begin / bpi, by_c, bx_c;
    { const int rem = blk_begin - n_c * bpi; by_c = rem / bx_n; bx_c = rem - by_c * bx_n; }
    n_c = __builtin_amdgcn_readfirstlane(n_c); by_c = __builtin_amdgcn_readfirstlane(by_c); bx_c = __builtin_amdgcn_readfirstlane(bx_c);
    int n_n = n_c, by_n_ = by_c, bx_n_ = bx_c;                                              // coordinates of the block being prefetched
    auto advance = [&](int& n, int& by, int& bx) {
        ++bx;
        if (bx == bx_n) { bx = 0; ++by; if (by == by_n) { by = 0; ++n; } }
    };

    f32x4 pf[4];
    auto prefetch = [&](int n, int by, int bx) {
        const int row0 = by * 8 - 1, col0 = bx * S1W_BW - 4;
        const unsigned boff = (unsigned)((((long long)n * Ci * H + row0) * W + col0) * 4);  // wraps for the (masked) negative corner
        const bool ok = hlane && (unsigned)(row0 + hrow) < (unsigned)H && (unsigned)(col0 + 4 * hseg) <= (unsigned)(W - 4);
        const unsigned voff = ok ? ld_lane + boff : OOB;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (CI ? c < CI : c < Ci) pf[c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, voff, c * plane, 0));
            else pf[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto stage = [&](T* img) {
        if (hlane) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                Quad<T> qv;
#pragma unroll
                for (int c = 0; c < 4; ++c) qv.v[c] = from_f32<T>(pf[c][i]);
                if constexpr (PRESPLIT) hyb_presplit4(qv.v);
                *reinterpret_cast<Quad<T>*>(img + st_lds + i * 4) = qv;
            }
        }
    };
    int cur = 0;
    if (nblk > 0) { prefetch(n_c, by_c, bx_c); stage(img0); advance(n_n, by_n_, bx_n_); }

    for (int it = 0; it < nblk; ++it) {
        const T* img = img0 + cur * S1W_IMG;
        const bool more = it + 1 < nblk;
        if (more) prefetch(n_n, by_n_, bx_n_);                        // in flight under this block's work
        const bool inside = (by_c * 8 + 8 <= H) && (bx_c * S1W_BW + S1W_BW <= W);          // uniform: no ragged edge in this block
        // the fragments of window positions jx = 0 and jx = 1 overlap by one pixel; left alone the compiler reads every pixel once and
        // assembles the second fragment with register moves -- VALU work on the critical resource to save LDS bandwidth that is idle.
        // An opaque copy of the offset keeps the two reads separate: each fragment is one ds_read2_b64 into its operand registers.
        int fx0[4] = {f_k0[0], f_k0[0] + 4, f_k0[1], f_k0[1] + 4}, fx1[4] = {f_k1[0], f_k1[0] + 4, f_k1[1], f_k1[1] + 4};      // [j = 2 jy + jx]
        asm volatile("" : "+v"(fx0[1]), "+v"(fx1[1]), "+v"(fx0[3]), "+v"(fx1[3]));
#pragma unroll S1W_SB_UNROLL
        for (int sb = 0; sb < 2; ++sb) {
            f32x4 acc[4][NT];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int joff = 8 * sb * 4;
                const int f_k0 = fx0[j], f_k1 = fx1[j];
                Frag<T> b0, b1;
                {
                    const Quad<T> lo = *reinterpret_cast<const Quad<T>*>(img + f_k0 + joff);
                    const Quad<T> hi = *reinterpret_cast<const Quad<T>*>(img + f_k0 + joff + 4);
#pragma unroll
                    for (int c = 0; c < 4; ++c) { b0.v[c] = lo.v[c]; b0.v[4 + c] = hi.v[c]; }
                    if constexpr (PRESPLIT) s1w_quads_to_frag(b0, lo, hi);
                }
                {
                    const Quad<T> lo = *reinterpret_cast<const Quad<T>*>(img + f_k1 + joff);
                    const Quad<T> hi = *reinterpret_cast<const Quad<T>*>(img + f_k1 + joff + 4);
#pragma unroll
                    for (int c = 0; c < 4; ++c) { b1.v[c] = lo.v[c]; b1.v[4 + c] = hi.v[c]; }
                    if constexpr (PRESPLIT) s1w_quads_to_frag(b1, lo, hi);
                }
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    acc[j][t] = PRESPLIT ? mma32_pre(w0[t], b0, f32x4{0.f, 0.f, 0.f, 0.f}) : mma32(w0[t], b0, f32x4{0.f, 0.f, 0.f, 0.f});
                    acc[j][t] = PRESPLIT ? mma32_pre(w1[t], b1, acc[j][t]) : mma32(w1[t], b1, acc[j][t]);
                }
            }
            if (MODE == 0) {
                if (inside) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int t = 0; t < NT; ++t)
#pragma unroll
                            for (int h = 0; h < 2; ++h) {
                                const f32x2 v = f32x2{acc[j][t][2 * h], acc[j][t][2 * h + 1]};
                                acc1[t][h] += v;
                                acc2[t][h] = v * v + acc2[t][h];
                            }
                } else {
                    const int gy0 = by_c * 8 + 2 * wy, gx0 = bx_c * S1W_BW + 8 * sb + 2 * wx;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float m = ((gy0 + (j >> 1)) < H && (gx0 + (j & 1)) < W) ? 1.f : 0.f;
#pragma unroll
                        for (int t = 0; t < NT; ++t)
#pragma unroll
                            for (int h = 0; h < 2; ++h) {
                                const f32x2 v = f32x2{acc[j][t][2 * h] * m, acc[j][t][2 * h + 1] * m};
                                acc1[t][h] += v;
                                acc2[t][h] = v * v + acc2[t][h];
                            }
                    }
                }
            }
            if (MODE == 1) {
                const int oy0 = by_c * 4, ox0 = bx_c * (S1W_BW / 2) + 4 * sb;              // first window of this sub-block
                const unsigned pbase = (unsigned)((((long long)n_c * Ho + oy0) * Wo + ox0) * Cop * (int)sizeof(T));
                const bool win_ok = inside || ((oy0 + wy) < Ho && (ox0 + wx) < Wo);
                const unsigned pvoff = win_ok ? po_lane + pbase : OOB;
#pragma unroll
                for (int h8 = 0; h8 < NT / 2; ++h8) {
                    Vec8<T> o;
                    unsigned code = 0;
#pragma unroll
                    for (int e2 = 0; e2 < 4; ++e2) {
                        const int t = h8 * 2 + (e2 >> 1), h = e2 & 1;
                        const f32x2 sc = c_sc[t][h], sh = c_sh[t][h];
                        f32x2 v[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = f32x2{acc[j][t][2 * h], acc[j][t][2 * h + 1]} * sc + sh;
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            const float m = __builtin_fmaxf(__builtin_fmaxf(v[0][u], v[1][u]), v[2][u]);                 // v_max3
                            const float om = __builtin_fmaxf(__builtin_fmaxf(m, v[3][u]), 0.f);                          // v_max3 with the ReLU's zero
                            o.set(e2 * 2 + u, om);
                            if constexpr (ROUTE) {
                                // where the backward pass must send this window's gradient: the first maximum in torch's scan order
                                // (0,0),(0,1),(1,0),(1,1), behind the ReLU -- the decision stage1w_bwd_kernel would otherwise re-derive
                                // from a recomputed conv (16 MFMAs and ~100 vector instructions per 8x8 pixels there).  Where the ReLU
                                // passed, the pooled value IS the window maximum: no separate four-way maximum is needed.
                                unsigned k = v[2][u] == om ? 6u : 7u;
                                k = v[1][u] == om ? 5u : k;
                                k = v[0][u] == om ? 4u : k;
                                k = om > 0.f ? k : 0u;
                                code |= k << (4 * (e2 * 2 + u));
                            }
                        }
                    }
                    s1w_store8(prs, win_ok ? pvoff + h8 * 8 * (unsigned)sizeof(T) : OOB, o);
                    if constexpr (ROUTE)
                        __builtin_amdgcn_raw_buffer_store_b32(code, rrs, win_ok ? (pvoff + h8 * 8 * (unsigned)sizeof(T)) >> 2 : OOB, 0, 0);
                }
            }
        }
        if (more) { stage(img0 + (cur ^ 1) * S1W_IMG); advance(n_n, by_n_, bx_n_); }      // the other buffer: this block's reads are all issued
        advance(n_c, by_c, bx_c);
        cur ^= 1;
    }

    if (MODE == 0) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float s1 = group16_sum(acc1[t][r >> 1][r & 1]), s2 = group16_sum(acc2[t][r >> 1][r & 1]);
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
}

// ---- wave-private backward pass (MODE 4 of conv_first.hip, same mathematics and the same partial-row layout) ----------------------
// Per 8x8 sub-block a wave: recomputes the conv (16 MFMAs), routes dpooled to the window arg-max behind the ReLU (dz), parks dz in its
// LDS slice [pixel][channel] and accumulates  S1 += dz^T P  and the Gram matrix  G += P^T P  with pixels as the MFMA K dimension.
// The im2col matrix P is never built: a transposing LDS read (ds_read_b64_tr_b16) whose per-lane address is "pixel + tap offset"
// takes the 4 channels x 4 consecutive pixels block straight out of the halo IMAGE (k = tap*4 + c: the four K columns of one tap are
// the pixel's channel quad); the ones column and the zero padding of K come from a 3-quad constant table.  Only for frames whose
// height is a multiple of 8 and width a multiple of 16 (no ragged blocks: G must not see pixels outside the image); everything else
// takes the block-level kernel.
constexpr int S1B_CT = 16;                       // constant table elements: quad {1,0,0,0} (K column 36 = ones), two zero quads, pad
#ifndef S1B_SCATTER_DEFAULT
#define S1B_SCATTER_DEFAULT 1                    /* 0: the first form of the gradient routing (A/B: a -DS1B_SCATTER_DEFAULT=0 build) */
#endif
constexpr bool S1B_SCATTER = S1B_SCATTER_DEFAULT != 0;

template <typename T> struct S1BTr;
template <> struct S1BTr<bf16> {
    // lo/hi: byte-exact 8-byte rows supplied by this lane (see tr_frag in conv_first.hip): the hardware hands lane L column L & 15
    static __device__ __forceinline__ void read(Frag<bf16>& f, const bf16* lo, const bf16* hi) {
        typedef __attribute__((address_space(3))) bf16x4 lds_q;
        const bf16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_q*)lo);
        const bf16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_q*)hi);
        f.v[0] = a[0]; f.v[1] = a[1]; f.v[2] = a[2]; f.v[3] = a[3];
        f.v[4] = b[0]; f.v[5] = b[1]; f.v[6] = b[2]; f.v[7] = b[3];
    }
};

template <typename T, int NT, int CI, bool WITH_G /* false: the forward pass saved G (stage1w_gram_kernel), only S1 is accumulated */,
          bool ROUTED = false /* the forward pass saved its routing codes (a.route): no conv recompute */>
__global__ __launch_bounds__(256) void stage1w_bwd_kernel(S1Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int DS = NT * 16 + 8;                                            // dz row stride (elements): odd multiple of 16 bytes
    constexpr int WAVE_EL = S1W_IMG + 64 * DS + S1B_CT;                        // ONE image: a wave's LDS operations execute in order, so the next
                                                                               // block's image can be written once this block's reads are issued
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    T* img0 = reinterpret_cast<T*>(smem_raw) + wave * WAVE_EL;
    T* dyt = img0 + S1W_IMG;                                                   // [64 pixels][DS]
    T* ctab = dyt + 64 * DS;
    const int p = lane & 15, q = lane >> 4, wy = p >> 2, wx = p & 3;
    const int pp = lane & 3, qq = (lane & 15) >> 2;
    const int co_base = blockIdx.y * (NT * 16);
    const int H = a.H, W = a.W, Ci = a.Ci, Cop = a.Cop;
    const int Ho = H >> 1, Wo = W >> 1;
    const T* wp = (const T*)a.wp2;

    if (lane < S1B_CT) ctab[lane] = from_f32<T>(lane == 0 ? 1.f : 0.f);
    Frag<T> w0[NT], w1[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const T* row = wp + (long long)(co_base + (p >> 2) * (NT * 4) + t * 4 + (p & 3)) * S1_KP + 8 * q;
        frag_load(w0[t], row);
        frag_load(w1[t], row + 32);
    }
    f32x2 c_sc[NT][2], c_sh[NT][2];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ch = co_base + q * (NT * 4) + t * 4 + r;
            c_sc[t][r >> 1][r & 1] = a.ss[ch]; c_sh[t][r >> 1][r & 1] = a.ss[Cop + ch];
        }
    f32x4 wacc[3][NT], gacc[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) gacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < 3; ++kt)
#pragma unroll
        for (int c = 0; c < NT; ++c) wacc[kt][c] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int bx_n = a.tilesX, by_n = a.tilesY, bpi = bx_n * by_n;
    const int nwaves = (int)gridDim.x * 4, gw = (int)blockIdx.x * 4 + wave;
    const int chunk = (a.numTiles + nwaves - 1) / nwaves;
    const int blk_begin = gw * chunk < a.numTiles ? gw * chunk : a.numTiles;
    const int blk_end = blk_begin + chunk < a.numTiles ? blk_begin + chunk : a.numTiles;
    const int nblk = blk_end - blk_begin;

    // ---- per-lane constants (see stage1w_kernel) ----
    const int hrow = lane / 6, hseg = lane - hrow * 6;
    const bool hlane = lane < 60;
    const unsigned ld_lane = (unsigned)((hrow * W + 4 * hseg) * 4);
    const unsigned st_lds = (unsigned)((s1w_rowoff(hrow) + 4 * hseg) * 4);
    const unsigned plane = (unsigned)(H * W) * 4u;
    const __amdgpu_buffer_rsrc_t xrs = hyb_rsrc(a.x, (unsigned)((long long)a.N * Ci * H * W * 4));
    const __amdgpu_buffer_rsrc_t drs = hyb_rsrc(a.dp, (unsigned)((long long)a.N * Ho * Wo * Cop * (int)sizeof(T)));
    const unsigned OOB = 0xFFFFFFF0u;
    int f_k0[2], f_k1[2];                                                      // [jy]: window rows 2 wy + jy
#pragma unroll
    for (int jy = 0; jy < 2; ++jy) {
        f_k0[jy] = (s1w_rowoff(2 * wy + jy + (q < 3 ? q : 0)) + 2 * wx + 3 + (q < 3 ? 0 : 2)) * 4;
        f_k1[jy] = (s1w_rowoff(2 * wy + jy + (q == 0 ? 1 : 2)) + 2 * wx + 3 + 2) * 4;
    }
    const unsigned dp_lane = (unsigned)(((wy * Wo + wx) * Cop + co_base + q * (NT * 4)) * (int)sizeof(T));
    const __amdgpu_buffer_rsrc_t rrs = hyb_rsrc(ROUTED ? (const void*)a.route : (const void*)a.x, ROUTED ? (unsigned)((long long)a.N * Ho * Wo * Cop / 2) : 16u);
    // pixels-as-K operands.  K index i of a 32-pixel step = (row i / 8, column i % 8) of 4 rows; this lane group (q) holds i = 4q .. 4q+3
    // (lo) and 16 + 4q .. (hi = two rows further down); within a 16-lane group lane (pp, qq) supplies pixel i + qq, K columns 4pp .. 4pp+3.
    const int pix_col = 4 * (q & 1) + qq + 3;                                   // image column of pixel (4q + qq), tap column 0
    int boff[3];                                                               // + tap offset of K tile kt (tap = 4 kt + pp), or the constant table
    bool bconst[3];
#pragma unroll
    for (int kt = 0; kt < 3; ++kt) {
        const int tap = 4 * kt + pp;
        bconst[kt] = tap >= 9;
        boff[kt] = tap < 9 ? (s1w_rowoff((q >> 1) + tap / 3) + pix_col + tap % 3) * 4 : (tap - 9) * 4;
    }
    const int a_lo = (4 * q + qq) * DS + 4 * pp;                               // dz tile: pixel 4q + qq, channels 4pp .. (+ 16 c)

    int n_c = blk_begin / bpi, by_c, bx_c;
    { const int rem = blk_begin - n_c * bpi; by_c = rem / bx_n; bx_c = rem - by_c * bx_n; }
    n_c = __builtin_amdgcn_readfirstlane(n_c); by_c = __builtin_amdgcn_readfirstlane(by_c); bx_c = __builtin_amdgcn_readfirstlane(bx_c);
    int n_n = n_c, by_n_ = by_c, bx_n_ = bx_c;
    auto advance = [&](int& n, int& by, int& bx) {
        ++bx;
        if (bx == bx_n) { bx = 0; ++by; if (by == by_n) { by = 0; ++n; } }
    };
    f32x4 pf[4];
    Vec8<T> gpf[2][NT / 2];                                                    // dpooled of both sub-blocks, one block ahead
    unsigned cpf[2][NT / 2];                                                   // .. and their routing codes (ROUTED)
    auto prefetch = [&](int n, int by, int bx) {
        const int row0 = by * 8 - 1, col0 = bx * S1W_BW - 4;
        const unsigned bo = (unsigned)((((long long)n * Ci * H + row0) * W + col0) * 4);
        const bool ok = hlane && (unsigned)(row0 + hrow) < (unsigned)H && (unsigned)(col0 + 4 * hseg) <= (unsigned)(W - 4);
        const unsigned voff = ok ? ld_lane + bo : OOB;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (CI ? c < CI : c < Ci) pf[c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, voff, c * plane, 0));
            else pf[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        const unsigned dbase = (unsigned)((((long long)n * Ho + by * 4) * Wo + bx * (S1W_BW / 2)) * Cop * (int)sizeof(T)) + dp_lane;
#pragma unroll
        for (int sb = 0; sb < 2; ++sb)
#pragma unroll
            for (int h8 = 0; h8 < NT / 2; ++h8) {
                const unsigned off = dbase + (unsigned)((sb * 4 * Cop + h8 * 8) * (int)sizeof(T));
                static_assert(sizeof(T) == 2, "16-bit storage only (transposing LDS reads)");
                gpf[sb][h8] = __builtin_bit_cast(Vec8<T>, __builtin_amdgcn_raw_buffer_load_b128(drs, off, 0, 0));
                if constexpr (ROUTED) cpf[sb][h8] = __builtin_amdgcn_raw_buffer_load_b32(rrs, off >> 2, 0, 0);
            }
    };
    auto stage = [&](T* img) {
        if (hlane) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                Quad<T> qv;
#pragma unroll
                for (int c = 0; c < 4; ++c) qv.v[c] = from_f32<T>(pf[c][i]);
                *reinterpret_cast<Quad<T>*>(img + st_lds + i * 4) = qv;
            }
        }
    };
    if (nblk > 0) { prefetch(n_c, by_c, bx_c); stage(img0); advance(n_n, by_n_, bx_n_); }

    for (int it = 0; it < nblk; ++it) {
        const T* img = img0;
        const bool more = it + 1 < nblk;
        Vec8<T> gcur[2][NT / 2];
        unsigned ccur[2][NT / 2];
#pragma unroll
        for (int sb = 0; sb < 2; ++sb)
#pragma unroll
            for (int h8 = 0; h8 < NT / 2; ++h8) { gcur[sb][h8] = gpf[sb][h8]; if constexpr (ROUTED) ccur[sb][h8] = cpf[sb][h8]; }
        if (more) prefetch(n_n, by_n_, bx_n_);
        int fx0[4] = {f_k0[0], f_k0[0] + 4, f_k0[1], f_k0[1] + 4}, fx1[4] = {f_k1[0], f_k1[0] + 4, f_k1[1], f_k1[1] + 4};      // [j = 2 jy + jx]
        asm volatile("" : "+v"(fx0[1]), "+v"(fx1[1]), "+v"(fx0[3]), "+v"(fx1[3]));
#pragma unroll 1
        for (int sb = 0; sb < 2; ++sb) {
            // ---- conv of the 8x8 sub-block (as in the forward kernels) -- only to re-derive the routing; skipped when the forward pass saved it
            f32x4 acc[4][NT];
            if constexpr (!ROUTED) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int joff = 8 * sb * 4;
                Frag<T> b0, b1;
                {
                    const Quad<T> lo = *reinterpret_cast<const Quad<T>*>(img + fx0[j] + joff);
                    const Quad<T> hi = *reinterpret_cast<const Quad<T>*>(img + fx0[j] + joff + 4);
#pragma unroll
                    for (int c = 0; c < 4; ++c) { b0.v[c] = lo.v[c]; b0.v[4 + c] = hi.v[c]; }
                }
                {
                    const Quad<T> lo = *reinterpret_cast<const Quad<T>*>(img + fx1[j] + joff);
                    const Quad<T> hi = *reinterpret_cast<const Quad<T>*>(img + fx1[j] + joff + 4);
#pragma unroll
                    for (int c = 0; c < 4; ++c) { b1.v[c] = lo.v[c]; b1.v[4 + c] = hi.v[c]; }
                }
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    acc[j][t] = mma32(w0[t], b0, f32x4{0.f, 0.f, 0.f, 0.f});
                    acc[j][t] = mma32(w1[t], b1, acc[j][t]);
                }
            }
            }
            // ---- route dpooled to the first maximum of each window (torch's scan order) behind the ReLU; park dz [pixel][channel].
            // dz has ONE non-zero per (window, channel): the tile is zeroed (four 16-byte LDS stores, no vector arithmetic) and every
            // gradient element is then written ONCE, as the 16 bits it arrived as, to the slot of its window's arg-max pixel -- a 2-byte
            // LDS store whose address is picked by three selects (an element whose ReLU was off goes to a pad column).  The first form
            // built all four pixels' values for every element (4 selects + bf16 repacking per pixel): 195 vector instructions per sub-block
            // against 28 MFMAs, and the kernel is bound by vector issue.  A wave's LDS operations execute in order.
            if constexpr (ROUTED) {
                // the forward pass's decisions (stage1w_kernel<.., ROUTE>): 4 bits per channel, bits 0-1 = arg-max pixel, bit 2 = ReLU passed.
                // The pixel's byte offset from pixel 0 -- {0, 1, 8, 9} x DS x 2 -- comes out of a four-entry byte table in one register.
                const f32x4 z4 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int i = 0; i < (64 * NT * 32) / (64 * 16); ++i) {
                    const int idx = i * 64 + lane, row = idx / (NT * 2), seg = idx % (NT * 2);
                    *reinterpret_cast<f32x4*>(dyt + row * DS + seg * 8) = z4;
                }
                static_assert((DS * 2) % 16 == 0 && 9 * DS * 2 / 16 < 256, "pixel offsets as bytes, in 16-byte units");
                constexpr unsigned LUT = (0u) | ((DS * 2 / 16) << 8) | ((8 * DS * 2 / 16) << 16) | ((unsigned)(9 * DS * 2 / 16) << 24);
#pragma unroll
                for (int h8 = 0; h8 < NT / 2; ++h8) {
                    const Vec8<T> g = sb == 0 ? gcur[0][h8] : gcur[1][h8];
                    const unsigned code = sb == 0 ? ccur[0][h8] : ccur[1][h8];
                    const u32x4 gw = __builtin_bit_cast(u32x4, g.v);
                    unsigned char* const d0 = reinterpret_cast<unsigned char*>(dyt + ((2 * wy) * 8 + 2 * wx) * DS + q * (NT * 4) + h8 * 8);      // pixel j = 0
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const unsigned k8 = (e == 0 ? code << 3 : code >> (4 * e - 3)) & 0x18u;       // 8 x (arg-max pixel)
                        const unsigned off16 = __builtin_amdgcn_ubfe(LUT, k8, 8u);
                        if (code & (4u << (4 * e)))
                            *reinterpret_cast<unsigned short*>(d0 + off16 * 16 + e * 2) = (unsigned short)((e & 1) ? (gw[e >> 1] >> 16) : gw[e >> 1]);
                    }
                }
            } else if (S1B_SCATTER) {
                const f32x4 z4 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int i = 0; i < (64 * NT * 32) / (64 * 16); ++i) {                   // 64 pixels x NT*16 channels x 2 bytes, 16 bytes per lane and store
                    const int idx = i * 64 + lane, row = idx / (NT * 2), seg = idx % (NT * 2);
                    *reinterpret_cast<f32x4*>(dyt + row * DS + seg * 8) = z4;
                }
#pragma unroll
                for (int h8 = 0; h8 < NT / 2; ++h8) {
                    const Vec8<T> g = sb == 0 ? gcur[0][h8] : gcur[1][h8];
                    const u32x4 gw = __builtin_bit_cast(u32x4, g.v);
                    unsigned short* const d0 = reinterpret_cast<unsigned short*>(dyt) + ((2 * wy) * 8 + 2 * wx) * DS + q * (NT * 4) + h8 * 8;      // pixel j = 0
                    unsigned short* const pad = reinterpret_cast<unsigned short*>(dyt) + NT * 16 + (lane & 7);     // pad columns of pixel row 0: never read
#pragma unroll
                    for (int e2 = 0; e2 < 4; ++e2) {
                        const int t = h8 * 2 + (e2 >> 1), h = e2 & 1;
                        const f32x2 sc = c_sc[t][h], sh = c_sh[t][h];
                        f32x2 v[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = f32x2{acc[j][t][2 * h], acc[j][t][2 * h + 1]} * sc + sh;
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            const int e = e2 * 2 + u;
                            const float m = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(v[0][u], v[1][u]), v[2][u]), v[3][u]);
                            int off = v[2][u] == m ? 8 * DS : 9 * DS;                    // pixel j = (j >> 1) rows, (j & 1) columns from pixel 0
                            off = v[1][u] == m ? DS : off;
                            off = v[0][u] == m ? 0 : off;
                            unsigned short* dst = m > 0.f ? d0 + off + e : pad;
                            *dst = (unsigned short)((e & 1) ? (gw[e >> 1] >> 16) : gw[e >> 1]);
                        }
                    }
                }
            } else {
#pragma unroll
            for (int h8 = 0; h8 < NT / 2; ++h8) {
                const Vec8<T> g = sb == 0 ? gcur[0][h8] : gcur[1][h8];
                Vec8<T> o[4];
#pragma unroll
                for (int e2 = 0; e2 < 4; ++e2) {
                    const int t = h8 * 2 + (e2 >> 1), h = e2 & 1;
                    const f32x2 sc = c_sc[t][h], sh = c_sh[t][h];
                    f32x2 v[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = f32x2{acc[j][t][2 * h], acc[j][t][2 * h + 1]} * sc + sh;
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int e = e2 * 2 + u;
                        const float m = __builtin_fmaxf(__builtin_fmaxf(v[0][u], v[1][u]), __builtin_fmaxf(v[2][u], v[3][u]));
                        const float dy = m > 0.f ? g.get(e) : 0.f;
                        const bool e0 = v[0][u] == m, e1 = v[1][u] == m, e2b = v[2][u] == m;
                        o[0].set(e, e0 ? dy : 0.f);
                        o[1].set(e, (!e0 && e1) ? dy : 0.f);
                        o[2].set(e, (!e0 && !e1 && e2b) ? dy : 0.f);
                        o[3].set(e, (!e0 && !e1 && !e2b) ? dy : 0.f);
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int pix = (2 * wy + (j >> 1)) * 8 + 2 * wx + (j & 1);
                    o[j].store(dyt + pix * DS + q * (NT * 4) + h8 * 8);
                }
            }
            }
            // ---- S1[co][k] += dz^T P and G += P^T P over the sub-block's 64 pixels (two K steps of 32 pixels = 4 rows x 8)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                Frag<T> af[NT], bf[3];
#pragma unroll
                for (int c = 0; c < NT; ++c) {
                    const T* lo = dyt + ks * 32 * DS + a_lo + c * 16;
                    S1BTr<T>::read(af[c], lo, lo + 16 * DS);
                }
                const int pstep = (2 * ks * S1W_PAIR + 8 * sb) * 4;                // rows 4 ks .., columns 8 sb .. of the image
#pragma unroll
                for (int kt = 0; kt < 3; ++kt) {
                    const T* lo = bconst[kt] ? ctab + boff[kt] : img + boff[kt] + pstep;
                    const T* hi = bconst[kt] ? lo : lo + S1W_PAIR * 4;
                    S1BTr<T>::read(bf[kt], lo, hi);
#pragma unroll
                    for (int c = 0; c < NT; ++c) wacc[kt][c] = mma32(af[c], bf[kt], wacc[kt][c]);
                }
                if (WITH_G && blockIdx.y == 0) {
                    gacc[0] = mma32(bf[0], bf[0], gacc[0]); gacc[1] = mma32(bf[0], bf[1], gacc[1]); gacc[2] = mma32(bf[0], bf[2], gacc[2]);
                    gacc[3] = mma32(bf[1], bf[1], gacc[3]); gacc[4] = mma32(bf[1], bf[2], gacc[4]); gacc[5] = mma32(bf[2], bf[2], gacc[5]);
                }
            }
        }
        if (more) { stage(img0); advance(n_n, by_n_, bx_n_); }
        advance(n_c, by_c, bx_c);
    }

    // combine the 4 waves through LDS: row = [S1: Cop x 48][G: 48 x 48 (upper triangle tiles filled)] -- the block-level kernel's layout
    // (without G the rows are Cop x 48 long)
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem_raw);                           // [4][NT*16*48 + 2304]
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
    const long long roww = (long long)Cop * 48 + (WITH_G ? 2304 : 0);
    float* out = a.part + (long long)blockIdx.x * roww;
    for (int i = tid; i < NT * 16 * 48; i += 256)
        out[(long long)co_base * 48 + i] = (red[i] + red[RW + i]) + (red[2 * RW + i] + red[3 * RW + i]);
    if (WITH_G && blockIdx.y == 0) {
        for (int i = tid; i < 2304; i += 256) {
            const int gr = i / 48, gc = i % 48;
            const int o = NT * 16 * 48 + i;
            out[(long long)Cop * 48 + i] = (gr / 16 <= gc / 16) ? (red[o] + red[RW + o]) + (red[2 * RW + o] + red[3 * RW + o]) : 0.f;
        }
    }
}

#ifdef HYB_F32_X3
// ---- the same backward pass for the split-bf16 build (fp32 storage outside, "bf16x3" mode): everything a wave parks in LDS exists as
// TWO bf16 planes, hi = bf16(v) and lo = bf16(v - hi) -- the halo image (split once when a pixel is staged), the routed gradient dz
// (split once when parked), the constant table -- so the transposing 16-bit reads work unchanged on either plane and every product is the
// three MFMAs lo*hi + hi*lo + hi*hi of hyb_common.h's mma32, with no conversion in any inner loop.  The lo plane sits S1X_PLS elements
// behind the hi plane.  No saved Gram matrix in this mode (the forward pass takes the statistics pass): always accumulates S1 and G.
__device__ __forceinline__ f32x4 s1x_mma3(const Frag<bf16>& ah, const Frag<bf16>& al, const Frag<bf16>& bh, const Frag<bf16>& bl, f32x4 c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al.v, bh.v, c, 0, 0, 0);      // small terms first (as mma32)
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah.v, bl.v, c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah.v, bh.v, c, 0, 0, 0);
}
__device__ __forceinline__ void s1x_quads(Frag<bf16>& f, const bf16* p) {   // two horizontally adjacent pixel quads = one fragment
    const Quad<bf16> lo = *reinterpret_cast<const Quad<bf16>*>(p);
    const Quad<bf16> hi = *reinterpret_cast<const Quad<bf16>*>(p + 4);
#pragma unroll
    for (int c = 0; c < 4; ++c) { f.v[c] = lo.v[c]; f.v[4 + c] = hi.v[c]; }
}

template <int NT, int CI>
__global__ __launch_bounds__(256) void stage1w_bwd_x3_kernel(S1Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int DS = NT * 16 + 8;
    constexpr int PLS = S1W_IMG + 64 * DS + S1B_CT;                            // one plane of a wave's slice (elements)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    bf16* img0 = reinterpret_cast<bf16*>(smem_raw) + wave * (2 * PLS);         // hi plane; + PLS = lo plane
    bf16* dyt = img0 + S1W_IMG;
    bf16* ctab = dyt + 64 * DS;
    const int p = lane & 15, q = lane >> 4, wy = p >> 2, wx = p & 3;
    const int pp = lane & 3, qq = (lane & 15) >> 2;
    const int co_base = blockIdx.y * (NT * 16);
    const int H = a.H, W = a.W, Ci = a.Ci, Cop = a.Cop;
    const int Ho = H >> 1, Wo = W >> 1;
    const float* wp = (const float*)a.wp2;

    if (lane < S1B_CT) { ctab[lane] = (bf16)(lane == 0 ? 1.f : 0.f); ctab[PLS + lane] = (bf16)0.f; }
    Frag<bf16> w0h[NT], w0l[NT], w1h[NT], w1l[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const float* row = wp + (long long)(co_base + (p >> 2) * (NT * 4) + t * 4 + (p & 3)) * S1_KP + 8 * q;
        Frag<float> f;
        frag_load(f, row);      hyb_split_bf16(f, w0h[t].v, w0l[t].v);
        frag_load(f, row + 32); hyb_split_bf16(f, w1h[t].v, w1l[t].v);
    }
    f32x2 c_sc[NT][2], c_sh[NT][2];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ch = co_base + q * (NT * 4) + t * 4 + r;
            c_sc[t][r >> 1][r & 1] = a.ss[ch]; c_sh[t][r >> 1][r & 1] = a.ss[Cop + ch];
        }
    f32x4 wacc[3][NT], gacc[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) gacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < 3; ++kt)
#pragma unroll
        for (int c = 0; c < NT; ++c) wacc[kt][c] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int bx_n = a.tilesX, by_n = a.tilesY, bpi = bx_n * by_n;
    const int nwaves = (int)gridDim.x * 4, gw = (int)blockIdx.x * 4 + wave;
    const int chunk = (a.numTiles + nwaves - 1) / nwaves;
    const int blk_begin = gw * chunk < a.numTiles ? gw * chunk : a.numTiles;
    const int blk_end = blk_begin + chunk < a.numTiles ? blk_begin + chunk : a.numTiles;
    const int nblk = blk_end - blk_begin;

    const int hrow = lane / 6, hseg = lane - hrow * 6;
    const bool hlane = lane < 60;
    const unsigned ld_lane = (unsigned)((hrow * W + 4 * hseg) * 4);
    const unsigned st_lds = (unsigned)((s1w_rowoff(hrow) + 4 * hseg) * 4);
    const unsigned plane = (unsigned)(H * W) * 4u;
    const __amdgpu_buffer_rsrc_t xrs = hyb_rsrc(a.x, (unsigned)((long long)a.N * Ci * H * W * 4));
    const __amdgpu_buffer_rsrc_t drs = hyb_rsrc(a.dp, (unsigned)((long long)a.N * Ho * Wo * Cop * 4));
    const unsigned OOB = 0xFFFFFFF0u;
    int f_k0[2], f_k1[2];
#pragma unroll
    for (int jy = 0; jy < 2; ++jy) {
        f_k0[jy] = (s1w_rowoff(2 * wy + jy + (q < 3 ? q : 0)) + 2 * wx + 3 + (q < 3 ? 0 : 2)) * 4;
        f_k1[jy] = (s1w_rowoff(2 * wy + jy + (q == 0 ? 1 : 2)) + 2 * wx + 3 + 2) * 4;
    }
    const unsigned dp_lane = (unsigned)(((wy * Wo + wx) * Cop + co_base + q * (NT * 4)) * 4);
    const int pix_col = 4 * (q & 1) + qq + 3;
    int boff[3];
    bool bconst[3];
#pragma unroll
    for (int kt = 0; kt < 3; ++kt) {
        const int tap = 4 * kt + pp;
        bconst[kt] = tap >= 9;
        boff[kt] = tap < 9 ? (s1w_rowoff((q >> 1) + tap / 3) + pix_col + tap % 3) * 4 : (tap - 9) * 4;
    }
    const int a_lo = (4 * q + qq) * DS + 4 * pp;

    int n_c = blk_begin / bpi, by_c, bx_c;
    { const int rem = blk_begin - n_c * bpi; by_c = rem / bx_n; bx_c = rem - by_c * bx_n; }
    n_c = __builtin_amdgcn_readfirstlane(n_c); by_c = __builtin_amdgcn_readfirstlane(by_c); bx_c = __builtin_amdgcn_readfirstlane(bx_c);
    int n_n = n_c, by_n_ = by_c, bx_n_ = bx_c;
    auto advance = [&](int& n, int& by, int& bx) {
        ++bx;
        if (bx == bx_n) { bx = 0; ++by; if (by == by_n) { by = 0; ++n; } }
    };
    f32x4 pf[4];
    Vec8<float> gpf[2][NT / 2];
    auto prefetch = [&](int n, int by, int bx) {
        const int row0 = by * 8 - 1, col0 = bx * S1W_BW - 4;
        const unsigned bo = (unsigned)((((long long)n * Ci * H + row0) * W + col0) * 4);
        const bool ok = hlane && (unsigned)(row0 + hrow) < (unsigned)H && (unsigned)(col0 + 4 * hseg) <= (unsigned)(W - 4);
        const unsigned voff = ok ? ld_lane + bo : OOB;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (CI ? c < CI : c < Ci) pf[c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, voff, c * plane, 0));
            else pf[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        const unsigned dbase = (unsigned)((((long long)n * Ho + by * 4) * Wo + bx * (S1W_BW / 2)) * Cop * 4) + dp_lane;
#pragma unroll
        for (int sb = 0; sb < 2; ++sb)
#pragma unroll
            for (int h8 = 0; h8 < NT / 2; ++h8) {
                const unsigned off = dbase + (unsigned)((sb * 4 * Cop + h8 * 8) * 4);
                gpf[sb][h8].a = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(drs, off, 0, 0));
                gpf[sb][h8].b = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(drs, off, 16, 0));
            }
    };
    auto stage = [&](bf16* img) {
        if (hlane) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                Quad<bf16> qh, ql;
#pragma unroll
                for (int c = 0; c < 4; ++c) { const float x = pf[c][i]; qh.v[c] = (bf16)x; ql.v[c] = (bf16)(x - (float)qh.v[c]); }
                *reinterpret_cast<Quad<bf16>*>(img + st_lds + i * 4) = qh;
                *reinterpret_cast<Quad<bf16>*>(img + PLS + st_lds + i * 4) = ql;
            }
        }
    };
    if (nblk > 0) { prefetch(n_c, by_c, bx_c); stage(img0); advance(n_n, by_n_, bx_n_); }

    for (int it = 0; it < nblk; ++it) {
        const bf16* img = img0;
        const bool more = it + 1 < nblk;
        Vec8<float> gcur[2][NT / 2];
#pragma unroll
        for (int sb = 0; sb < 2; ++sb)
#pragma unroll
            for (int h8 = 0; h8 < NT / 2; ++h8) gcur[sb][h8] = gpf[sb][h8];
        if (more) prefetch(n_n, by_n_, bx_n_);
        int fx0[4] = {f_k0[0], f_k0[0] + 4, f_k0[1], f_k0[1] + 4}, fx1[4] = {f_k1[0], f_k1[0] + 4, f_k1[1], f_k1[1] + 4};
        asm volatile("" : "+v"(fx0[1]), "+v"(fx1[1]), "+v"(fx0[3]), "+v"(fx1[3]));
#pragma unroll 1
        for (int sb = 0; sb < 2; ++sb) {
            f32x4 acc[4][NT];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int joff = 8 * sb * 4;
                Frag<bf16> b0h, b0l, b1h, b1l;
                s1x_quads(b0h, img + fx0[j] + joff); s1x_quads(b0l, img + PLS + fx0[j] + joff);
                s1x_quads(b1h, img + fx1[j] + joff); s1x_quads(b1l, img + PLS + fx1[j] + joff);
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    acc[j][t] = s1x_mma3(w0h[t], w0l[t], b0h, b0l, f32x4{0.f, 0.f, 0.f, 0.f});
                    acc[j][t] = s1x_mma3(w1h[t], w1l[t], b1h, b1l, acc[j][t]);
                }
            }
#pragma unroll
            for (int h8 = 0; h8 < NT / 2; ++h8) {
                const Vec8<float> g = sb == 0 ? gcur[0][h8] : gcur[1][h8];
                Vec8<bf16> oh[4], ol[4];
#pragma unroll
                for (int e2 = 0; e2 < 4; ++e2) {
                    const int t = h8 * 2 + (e2 >> 1), h = e2 & 1;
                    const f32x2 sc = c_sc[t][h], sh = c_sh[t][h];
                    f32x2 v[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = f32x2{acc[j][t][2 * h], acc[j][t][2 * h + 1]} * sc + sh;
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int e = e2 * 2 + u;
                        const float m = __builtin_fmaxf(__builtin_fmaxf(v[0][u], v[1][u]), __builtin_fmaxf(v[2][u], v[3][u]));
                        const float dy = m > 0.f ? g.get(e) : 0.f;
                        const bf16 dh = (bf16)dy, dl = (bf16)(dy - (float)dh);      // split once; routing only selects
                        const bf16 z = (bf16)0.f;
                        const bool e0 = v[0][u] == m, e1 = v[1][u] == m, e2b = v[2][u] == m;
                        const bool s0 = e0, s1 = !e0 && e1, s2 = !e0 && !e1 && e2b, s3 = !e0 && !e1 && !e2b;
                        oh[0].v[e] = s0 ? dh : z; ol[0].v[e] = s0 ? dl : z;
                        oh[1].v[e] = s1 ? dh : z; ol[1].v[e] = s1 ? dl : z;
                        oh[2].v[e] = s2 ? dh : z; ol[2].v[e] = s2 ? dl : z;
                        oh[3].v[e] = s3 ? dh : z; ol[3].v[e] = s3 ? dl : z;
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int pix = (2 * wy + (j >> 1)) * 8 + 2 * wx + (j & 1);
                    oh[j].store(dyt + pix * DS + q * (NT * 4) + h8 * 8);
                    ol[j].store(dyt + PLS + pix * DS + q * (NT * 4) + h8 * 8);
                }
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                Frag<bf16> afh[NT], afl[NT], bfh[3], bfl[3];
#pragma unroll
                for (int c = 0; c < NT; ++c) {
                    const bf16* lo = dyt + ks * 32 * DS + a_lo + c * 16;
                    S1BTr<bf16>::read(afh[c], lo, lo + 16 * DS);
                    S1BTr<bf16>::read(afl[c], lo + PLS, lo + PLS + 16 * DS);
                }
                const int pstep = (2 * ks * S1W_PAIR + 8 * sb) * 4;
#pragma unroll
                for (int kt = 0; kt < 3; ++kt) {
                    const bf16* lo = bconst[kt] ? ctab + boff[kt] : img + boff[kt] + pstep;
                    const bf16* hi = bconst[kt] ? lo : lo + S1W_PAIR * 4;
                    S1BTr<bf16>::read(bfh[kt], lo, hi);
                    S1BTr<bf16>::read(bfl[kt], lo + PLS, hi + PLS);
#pragma unroll
                    for (int c = 0; c < NT; ++c) wacc[kt][c] = s1x_mma3(afh[c], afl[c], bfh[kt], bfl[kt], wacc[kt][c]);
                }
                if (blockIdx.y == 0) {
                    gacc[0] = s1x_mma3(bfh[0], bfl[0], bfh[0], bfl[0], gacc[0]); gacc[1] = s1x_mma3(bfh[0], bfl[0], bfh[1], bfl[1], gacc[1]);
                    gacc[2] = s1x_mma3(bfh[0], bfl[0], bfh[2], bfl[2], gacc[2]); gacc[3] = s1x_mma3(bfh[1], bfl[1], bfh[1], bfl[1], gacc[3]);
                    gacc[4] = s1x_mma3(bfh[1], bfl[1], bfh[2], bfl[2], gacc[4]); gacc[5] = s1x_mma3(bfh[2], bfl[2], bfh[2], bfl[2], gacc[5]);
                }
            }
        }
        if (more) { stage(img0); advance(n_n, by_n_, bx_n_); }
        advance(n_c, by_c, bx_c);
    }

    __syncthreads();
    float* red = reinterpret_cast<float*>(smem_raw);                           // [4][NT*16*48 + 2304]
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

template <int NT>
int s1x_bwd_launch(S1Args a, int grid_x, hipStream_t st) {
    constexpr int DS = NT * 16 + 8;
    size_t lds = (size_t)4 * 2 * (S1W_IMG + 64 * DS + S1B_CT) * sizeof(bf16);
    const size_t red = (size_t)4 * (NT * 16 * 48 + 2304) * sizeof(float);
    if (red > lds) lds = red;
    lds += 64;
    dim3 grid(grid_x, a.Cop / (NT * 16));
    if (lds > 64 * 1024) {
        static HybAttrOnce once3, once0;
        if (int e = hyb_set_lds_attr(a.Ci == 3 ? once3 : once0, a.Ci == 3 ? (const void*)stage1w_bwd_x3_kernel<NT, 3> : (const void*)stage1w_bwd_x3_kernel<NT, 0>, (int)lds)) return e;
    }
    if (a.Ci == 3) hipLaunchKernelGGL((stage1w_bwd_x3_kernel<NT, 3>), grid, dim3(256), lds, st, a);
    else hipLaunchKernelGGL((stage1w_bwd_x3_kernel<NT, 0>), grid, dim3(256), lds, st, a);
    HYB_LAUNCH_CHECK();
    return 0;
}
#endif  // HYB_F32_X3

// ---- Gram pass: G = P^T P over all pixels (48 x 48, upper-triangle tiles), the im2col patches read as in the backward kernel.
// G depends on the frames only.  It serves the BatchNorm batch statistics (column 36 of P is all ones: G[k][36] = sum P[k], and
//   sum_pix y_co = w_co . G[:,36],   sum_pix y_co^2 = w_co^T G w_co      -- no conv, no per-pixel squares: 12 MFMAs and no vector
// epilogue per 64 pixels instead of 16 MFMAs + 64 packed adds/fmas) and is kept for the backward pass, which then accumulates S1 only.
template <typename T, int CI>
__global__ __launch_bounds__(256) void stage1w_gram_kernel(S1Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int WAVE_EL = S1W_IMG + S1B_CT;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    T* img = reinterpret_cast<T*>(smem_raw) + wave * WAVE_EL;                  // one image: the next block's rows wait in registers
    T* ctab = img + S1W_IMG;
    const int q = lane >> 4, p = lane & 15, pp = lane & 3, qq = (lane & 15) >> 2;
    const int H = a.H, W = a.W, Ci = a.Ci;
    if (lane < S1B_CT) ctab[lane] = from_f32<T>(lane == 0 ? 1.f : 0.f);
    f32x4 gacc[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) gacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int bx_n = a.tilesX, by_n = a.tilesY, bpi = bx_n * by_n;
    const int nwaves = (int)gridDim.x * 4, gw = (int)blockIdx.x * 4 + wave;
    const int chunk = (a.numTiles + nwaves - 1) / nwaves;
    const int blk_begin = gw * chunk < a.numTiles ? gw * chunk : a.numTiles;
    const int blk_end = blk_begin + chunk < a.numTiles ? blk_begin + chunk : a.numTiles;
    const int nblk = blk_end - blk_begin;
    const int hrow = lane / 6, hseg = lane - hrow * 6;
    const bool hlane = lane < 60;
    const unsigned ld_lane = (unsigned)((hrow * W + 4 * hseg) * 4);
    const unsigned st_lds = (unsigned)((s1w_rowoff(hrow) + 4 * hseg) * 4);
    const unsigned plane = (unsigned)(H * W) * 4u;
    const __amdgpu_buffer_rsrc_t xrs = hyb_rsrc(a.x, (unsigned)((long long)a.N * Ci * H * W * 4));
    const unsigned OOB = 0xFFFFFFF0u;
    const int pix_col = 4 * (q & 1) + qq + 3;
    int boff[3];
    bool bconst[3];
#pragma unroll
    for (int kt = 0; kt < 3; ++kt) {
        const int tap = 4 * kt + pp;
        bconst[kt] = tap >= 9;
        boff[kt] = tap < 9 ? (s1w_rowoff((q >> 1) + tap / 3) + pix_col + tap % 3) * 4 : (tap - 9) * 4;
    }
    int n_c = blk_begin / bpi, by_c, bx_c;
    { const int rem = blk_begin - n_c * bpi; by_c = rem / bx_n; bx_c = rem - by_c * bx_n; }
    n_c = __builtin_amdgcn_readfirstlane(n_c); by_c = __builtin_amdgcn_readfirstlane(by_c); bx_c = __builtin_amdgcn_readfirstlane(bx_c);
    auto advance = [&](int& n, int& by, int& bx) {
        ++bx;
        if (bx == bx_n) { bx = 0; ++by; if (by == by_n) { by = 0; ++n; } }
    };
    f32x4 pf[4];
    auto prefetch = [&](int n, int by, int bx) {
        const int row0 = by * 8 - 1, col0 = bx * S1W_BW - 4;
        const unsigned bo = (unsigned)((((long long)n * Ci * H + row0) * W + col0) * 4);
        const bool ok = hlane && (unsigned)(row0 + hrow) < (unsigned)H && (unsigned)(col0 + 4 * hseg) <= (unsigned)(W - 4);
        const unsigned voff = ok ? ld_lane + bo : OOB;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (CI ? c < CI : c < Ci) pf[c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, voff, c * plane, 0));
            else pf[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto stage = [&]() {
        if (hlane) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                Quad<T> qv;
#pragma unroll
                for (int c = 0; c < 4; ++c) qv.v[c] = from_f32<T>(pf[c][i]);
                *reinterpret_cast<Quad<T>*>(img + st_lds + i * 4) = qv;
            }
        }
    };
    if (nblk > 0) { prefetch(n_c, by_c, bx_c); stage(); advance(n_c, by_c, bx_c); }
    for (int it = 0; it < nblk; ++it) {
        const bool more = it + 1 < nblk;
        if (more) prefetch(n_c, by_c, bx_c);                          // in flight under this block's work
#pragma unroll
        for (int sb = 0; sb < 2; ++sb)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                Frag<T> bf[3];
                const int pstep = (2 * ks * S1W_PAIR + 8 * sb) * 4;
#pragma unroll
                for (int kt = 0; kt < 3; ++kt) {
                    const T* lo = bconst[kt] ? ctab + boff[kt] : img + boff[kt] + pstep;
                    const T* hi = bconst[kt] ? lo : lo + S1W_PAIR * 4;
                    S1BTr<T>::read(bf[kt], lo, hi);
                }
                gacc[0] = mma32(bf[0], bf[0], gacc[0]); gacc[1] = mma32(bf[0], bf[1], gacc[1]); gacc[2] = mma32(bf[0], bf[2], gacc[2]);
                gacc[3] = mma32(bf[1], bf[1], gacc[3]); gacc[4] = mma32(bf[1], bf[2], gacc[4]); gacc[5] = mma32(bf[2], bf[2], gacc[5]);
            }
        if (more) { stage(); advance(n_c, by_c, bx_c); }              // same buffer: this wave's reads of it are all issued (in-order LDS)
    }
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem_raw);                           // [4][2304]
    auto put_g = [&](int i, int gi, int gj) {
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave * 2304 + (gi * 16 + 4 * q + r) * 48 + gj * 16 + p] = gacc[i][r];
    };
    put_g(0, 0, 0); put_g(1, 0, 1); put_g(2, 0, 2); put_g(3, 1, 1); put_g(4, 1, 2); put_g(5, 2, 2);
    __syncthreads();
    float* out = a.part + (long long)blockIdx.x * 2304;
    for (int i = tid; i < 2304; i += 256) {
        const int gr = i / 48, gc = i % 48;
        out[i] = (gr / 16 <= gc / 16) ? (red[i] + red[2304 + i]) + (red[2 * 2304 + i] + red[3 * 2304 + i]) : 0.f;
    }
}

template <typename T, int NT, bool WITH_G, bool ROUTED>
int s1w_bwd_launch_r(S1Args a, int grid_x, hipStream_t st) {
    constexpr int DS = NT * 16 + 8;
    size_t lds = (size_t)4 * (S1W_IMG + 64 * DS + S1B_CT) * sizeof(T);
    const size_t red = (size_t)4 * (NT * 16 * 48 + 2304) * sizeof(float);
    if (red > lds) lds = red;
    lds += 64;
    dim3 grid(grid_x, a.Cop / (NT * 16));
    if (lds > 64 * 1024) {
        static HybAttrOnce once3, once0;
        if (int e = hyb_set_lds_attr(a.Ci == 3 ? once3 : once0, a.Ci == 3 ? (const void*)stage1w_bwd_kernel<T, NT, 3, WITH_G, ROUTED> : (const void*)stage1w_bwd_kernel<T, NT, 0, WITH_G, ROUTED>, (int)lds)) return e;
    }
    if (a.Ci == 3) hipLaunchKernelGGL((stage1w_bwd_kernel<T, NT, 3, WITH_G, ROUTED>), grid, dim3(256), lds, st, a);
    else hipLaunchKernelGGL((stage1w_bwd_kernel<T, NT, 0, WITH_G, ROUTED>), grid, dim3(256), lds, st, a);
    HYB_LAUNCH_CHECK();
    return 0;
}
template <typename T, int NT, bool WITH_G>
int s1w_bwd_launch(S1Args a, int grid_x, hipStream_t st) {
    return a.route ? s1w_bwd_launch_r<T, NT, WITH_G, true>(a, grid_x, st) : s1w_bwd_launch_r<T, NT, WITH_G, false>(a, grid_x, st);
}

// packed first-layer weights for the wave-private forward kernels: T [Cop][64], k = slot*4 + c with the slot -> tap map below
// (slot pairs (2q, 2q+1) of a k-step are horizontally adjacent pixels; -1 = zero weights)
template <typename T>
__global__ void s1w_pack_kernel(const float* __restrict__ w, T* __restrict__ wp2, int Co, int Ci, long long total, int both) {
    // `both`: wp2 = [2][Cop][64], the first half in the block-level kernels' order (k = tap*4 + c), the second in this file's order
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (both ? 2 * total : total)) return;
    const bool second = !both || i >= total;
    T* dst = wp2 + i;
    if (both && second) i -= total;
    const float v = s1w_pack_value(w, i, second, Co, Ci);
    *dst = from_f32<T>(v);
}

template <typename T, int NT, int MODE>
int s1w_launch(S1Args a, int grid_x, hipStream_t st) {
    const size_t lds = (size_t)4 * 2 * S1W_IMG * sizeof(T) + 4 * 2 * NT * 16 * sizeof(float) + 64;
    dim3 grid(grid_x, a.Cop / (NT * 16));
    if constexpr (MODE == 1 && sizeof(T) == 2) {
        if (a.route) {
            if (a.Ci == 3) hipLaunchKernelGGL((stage1w_kernel<T, NT, MODE, 3, true>), grid, dim3(256), lds, st, a);
            else hipLaunchKernelGGL((stage1w_kernel<T, NT, MODE, 0, true>), grid, dim3(256), lds, st, a);
            HYB_LAUNCH_CHECK();
            return 0;
        }
    }
    if (a.Ci == 3) hipLaunchKernelGGL((stage1w_kernel<T, NT, MODE, 3>), grid, dim3(256), lds, st, a);
    else hipLaunchKernelGGL((stage1w_kernel<T, NT, MODE, 0>), grid, dim3(256), lds, st, a);
    HYB_LAUNCH_CHECK();
    return 0;
}
// the wave-private forward passes work on 8x8 blocks: tilesX/tilesY/numTiles of `a` are re-derived for that block size
template <typename T, int MODE>
int s1w_dispatch(S1Args a, int& grid_x /* in: wanted workgroups; out: launched (= partial rows of MODE 0) */, hipStream_t st) {
    a.tilesX = hyb_cdiv(a.W, S1W_BW); a.tilesY = hyb_cdiv(a.H, 8);
    const long long nb = (long long)a.N * a.tilesX * a.tilesY;
    if (nb >= (1ll << 30)) return HYB_E_ARG;
    a.numTiles = (int)nb;
    if ((long long)grid_x * 4 > nb) grid_x = (int)((nb + 3) / 4);
    if (a.Cop % 64 == 0) return s1w_launch<T, 4, MODE>(a, grid_x, st);
    return s1w_launch<T, 2, MODE>(a, grid_x, st);
}

}  // namespace

// both = 0: wp2 [Cop][64] in this file's K order; both = 1: wp2 is [2][Cop][64], block-level order first (one launch for the two layouts)
int hyb_stage1w_pack(int dtype, const float* weight, void* wp2, int Co, int Ci, int Cop, int both, hipStream_t st) {
    const long long total = (long long)Cop * 64, n = both ? 2 * total : total;
    if (dtype == HYB_F32) hipLaunchKernelGGL(s1w_pack_kernel<float>, dim3(hyb_cdiv(n, 256)), dim3(256), 0, st, weight, (float*)wp2, Co, Ci, total, both);
    else if (dtype == HYB_BF16) hipLaunchKernelGGL(s1w_pack_kernel<bf16>, dim3(hyb_cdiv(n, 256)), dim3(256), 0, st, weight, (bf16*)wp2, Co, Ci, total, both);
    else return HYB_E_ARG;
    HYB_LAUNCH_CHECK();
    return 0;
}

// backward pass over 8x16 blocks; `a` as for the block-level MODE 4 launch (wp2 set); bf16 only (the transposing LDS read is a 16-bit
// instruction): fp32 parity mode keeps the block-level kernel
int hyb_stage1w_bwd(int dtype, S1Args a, int with_g, int& grid_x, hipStream_t st) {
    if (dtype != HYB_BF16 && !(HYB_X3 && dtype == HYB_F32 && with_g)) return HYB_E_ARG;
    a.tilesX = hyb_cdiv(a.W, S1W_BW); a.tilesY = hyb_cdiv(a.H, 8);
    const long long nb = (long long)a.N * a.tilesX * a.tilesY;
    if (nb >= (1ll << 30)) return HYB_E_ARG;
    a.numTiles = (int)nb;
    if ((long long)grid_x * 4 > nb) grid_x = (int)((nb + 3) / 4);
#ifdef HYB_F32_X3
    if (dtype == HYB_F32) return a.Cop % 64 == 0 ? s1x_bwd_launch<4>(a, grid_x, st) : s1x_bwd_launch<2>(a, grid_x, st);
#endif
    if (a.Cop % 64 == 0) return with_g ? s1w_bwd_launch<bf16, 4, true>(a, grid_x, st) : s1w_bwd_launch<bf16, 4, false>(a, grid_x, st);
    return with_g ? s1w_bwd_launch<bf16, 2, true>(a, grid_x, st) : s1w_bwd_launch<bf16, 2, false>(a, grid_x, st);
}

// Gram pass over 8x16 blocks (no ragged blocks: H % 8 == 0, W % 16 == 0); writes grid_x partial rows of 2304 floats to a.part
int hyb_stage1w_gram(int dtype, S1Args a, int& grid_x, hipStream_t st) {
    if (dtype != HYB_BF16) return HYB_E_ARG;
    a.tilesX = hyb_cdiv(a.W, S1W_BW); a.tilesY = hyb_cdiv(a.H, 8);
    const long long nb = (long long)a.N * a.tilesX * a.tilesY;
    if (nb >= (1ll << 30)) return HYB_E_ARG;
    a.numTiles = (int)nb;
    if ((long long)grid_x * 4 > nb) grid_x = (int)((nb + 3) / 4);
    size_t lds = (size_t)4 * (S1W_IMG + S1B_CT) * sizeof(bf16);
    if (lds < (size_t)4 * 2304 * sizeof(float)) lds = (size_t)4 * 2304 * sizeof(float);
    lds += 64;
    if (a.Ci == 3) hipLaunchKernelGGL((stage1w_gram_kernel<bf16, 3>), dim3(grid_x), dim3(256), lds, st, a);
    else hipLaunchKernelGGL((stage1w_gram_kernel<bf16, 0>), dim3(grid_x), dim3(256), lds, st, a);
    HYB_LAUNCH_CHECK();
    return 0;
}

int hyb_stage1w_pass(int dtype, int mode, const S1Args& a, int& grid_x, hipStream_t st) {
    if (dtype == HYB_F32) return mode == 0 ? s1w_dispatch<float, 0>(a, grid_x, st) : s1w_dispatch<float, 1>(a, grid_x, st);
    if (dtype == HYB_BF16) return mode == 0 ? s1w_dispatch<bf16, 0>(a, grid_x, st) : s1w_dispatch<bf16, 1>(a, grid_x, st);
    return HYB_E_ARG;
}
