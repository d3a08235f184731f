// conv3x3 (stride 1, zero pad 1), bf16, NHWC: the asynchronous implicit-GEMM kernel used for forward (nn.Conv2d of
// UNet.py:58) and dgrad (mode-1 packed weights) whenever the channel counts allow it.
//
// A workgroup is NW = CB x PGR x PGC waves (CB channel blocks x pixel groups), persistent over image tiles: either one
// 8-wave workgroup per CU (all 160 KiB of LDS: deep weight ring for 256-channel blocks) or two independent 4-wave workgroups
// per CU, whose epilogues (vector work) and MFMA phases interleave on the SIMDs.
//   * a wave owns a strip of 7 patches (4 x 28 pixels) x NT*16 output channels: 28 | 224/2^k, so the stages of a 224 x 224
//     clip tile without remainder in x;
//   * the input halo of a 32-channel block and the per-tap weight slices never pass through registers: both are written
//     into LDS by buffer_load_dwordx4 ... lds (LDS-DMA).  Out-of-image halo lanes are sent out of the descriptor's range and
//     the hardware writes zeros for them (the zero padding costs no branch).  Halo images are double buffered, weight slices
//     live in a ring of R slots, so the loads of block j+1 / step s+R-1 are in flight while block j / step s is computed;
//   * every wave issues the same, compile-time-known number of DMA instructions per step, which makes the only
//     synchronisation of a step one counted `s_waitcnt vmcnt(N)` + one raw `s_barrier` (no vmcnt(0) in the loop); after a
//     full-tile epilogue the count also steps over that tile's output stores, which stay in flight;
//   * the A fragments and the patch fragments of step s+1 are read from LDS between the MFMA groups of step s;
//   * output channels are permuted inside the MFMA tiles so that a lane ends up with 8 consecutive channels per 32-channel
//     half: an epilogue store instruction writes 64 contiguous bytes per pixel;
//   * the BatchNorm batch statistics (UNet.py:59) are folded into the epilogue from the fp32 accumulators, reduced in a fixed
//     order (bit-reproducible).
#include <stdlib.h>
#include <type_traits>
#include "hyb_common.h"

namespace {

typedef __attribute__((address_space(3))) void lds_void_t;

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// LDS-DMA of 16 bytes per lane through a buffer descriptor: LDS destination = wave-uniform base + lane * 16, source =
// descriptor base + voff + soff.  A lane whose voff is outside the descriptor's range writes ZEROS (measured on gfx950):
// that is how the out-of-image part of a halo gets its zero padding, without a branch or a zero page.
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff, bf16* lds_dst_wave_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_t*)lds_dst_wave_base, 16, voff, soff, 0, 0);
}
constexpr unsigned V2_OOB = 0xfffffff0u;        // beyond every descriptor below (num_records = 2^31)
constexpr unsigned V2_RECORDS = 0x80000000u;

template <int S> using step_c = std::integral_constant<int, S>;

template <int NT, int CB, int PGR, int PGC, int R>
struct V2Geom {
    static constexpr int PG = PGR * PGC, MT = 7;
    static constexpr int NW = CB * PG;                       // waves per workgroup: 8 (one workgroup per CU) or 4 (two per CU)
    static constexpr int TH = 4 * PGR, TW = 28 * PGC, HH = TH + 2, HW_ = TW + 2, HP = HH * HW_;
    static constexpr int NHW = (HP * 4 + 63) / 64;          // halo wave-instructions (64 x 16 B each)
    static constexpr int HT = (NHW + NW - 1) / NW;          // ... per wave
    static constexpr int HBUF = NHW * 512;                  // bf16 elements per halo buffer
    static constexpr int CBW = CB * NT * 16;                // output channels per workgroup
    static constexpr int WSLOT = CBW * 32;                  // bf16 elements per ring slot
    static constexpr int NWW = CBW / 16;                    // weight wave-instructions per slice
    static constexpr int WI = NWW >= NW ? NWW / NW : 1;     // ... per wave (duplicated when the slice has fewer pieces than waves)
    static constexpr int NHS = 10 - R;                      // halo pieces are issued in steps 0 .. NHS-1 (see wait rule)
    static constexpr int STAT_FLOATS = NW * 2 * NT * 16;
    static constexpr size_t LDS_BYTES = (size_t)(2 * HBUF + R * WSLOT) * 2 + STAT_FLOATS * 4;
    static_assert(NW == 8 || NW == 4, "four or eight waves");
    static_assert(LDS_BYTES <= (NW == 8 ? 160 : 80) * 1024, "160 KiB of LDS per CU");
    static_assert(HW_ % 4 == 2, "halo swizzle assumes halo width = 2 mod 4");
    static_assert(R >= 3 && R <= 6, "ring depth");
    // halo pieces of step s
    static constexpr int hi(int s) { s = ((s % 9) + 9) % 9; return s < NHS ? (HT + NHS - 1 - s) / NHS : 0; }
    static constexpr int hstart(int s) { int a = 0; for (int j = 0; j < s; ++j) a += hi(j); return a; }
    // DMA instructions younger than the weights of step s+2 when step s ends (= the vmcnt to wait for)
    static constexpr int wait_n(int s) {
        int n = hi(s - R + 3);
        for (int j = s - R + 4; j <= s; ++j) n += WI + hi(j);
        return n;
    }
};

template <int NT, int CB, int PGR, int PGC, int R, bool STATS>
__global__ __launch_bounds__(CB * PGR * PGC * 64, CB * PGR * PGC == 4 ? 2 : 1) void conv3x3_v2_kernel(const bf16* __restrict__ x, const bf16* __restrict__ wp,
                                                            bf16* __restrict__ y, float* __restrict__ stats,
                                                            int N, int H, int W, int Cip, int Cop,
                                                            int tilesX, int tilesY, int numTiles, int stat_rows, int xpix, long long xblk) {
    // input addressing: element stride between pixels (Cip for NHWC) and between 32-channel blocks (32 for NHWC); a "block-planar"
    // input [Cip/32][N][H][W][32] has xpix = 32, xblk = N*H*W*32: a block's halo then uses whole 128-byte lines (see hyb_convstage_bwd)
    using G = V2Geom<NT, CB, PGR, PGC, R>;
    constexpr int MT = G::MT, TH = G::TH, TW = G::TW, HW_ = G::HW_, HP = G::HP, HT = G::HT, NHW = G::NHW;
    constexpr int CBW = G::CBW, WSLOT = G::WSLOT, NWW = G::NWW, WI = G::WI, PG = G::PG, NW = G::NW, NTHR = NW * 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16* const hbuf = reinterpret_cast<bf16*>(smem_raw);                       // [2][HBUF]
    bf16* const wring = hbuf + 2 * G::HBUF;                                     // [R][WSLOT]
    float* const wgstat = reinterpret_cast<float*>(wring + R * WSLOT);          // [NW waves][2][NT*16]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cb = wave % CB, pg = wave / CB;
    const int prow = pg / PGC, pstrip = pg % PGC;
    const int p = lane & 15, q = lane >> 4, py = p >> 2, px = p & 3;
    const int co_wg = blockIdx.y * CBW;
    const int co_base = co_wg + cb * (NT * 16);
    const int nCblk = Cip / 32;
    const long long wrow = (long long)9 * Cip;                                  // packed row: [Cip/32][9][32]

    if (STATS) {
        for (int i = tid; i < G::STAT_FLOATS; i += NTHR) wgstat[i] = 0.f;
    }

    // ---- halo DMA pieces of this lane: byte offset from the halo origin pixel, and (hy, hx) for the bounds test
    constexpr bool HOFF_REG = true;              // large halos: recompute the offset per piece instead of holding it (registers)
    unsigned hoff[HOFF_REG ? HT : 1];
    int hyx[HT];
#pragma unroll
    for (int k = 0; k < HT; ++k) {
        int wi = k * NW + wave;
        if (wi > NHW - 1) wi = NHW - 1;                                         // duplicate the last piece: equal counts per wave
        const int u = wi * 64 + lane, hp = u >> 2, sp = u & 3;
        const int hy = hp / HW_, hx = hp - hy * HW_;
        const int s = sp ^ ((hy & 1) << 1);                                     // halo swizzle for 32-channel pixels
        if (HOFF_REG) hoff[k] = (unsigned)(((hy * W + hx) * xpix + s * 8) * 2);
        hyx[k] = hp < HP ? ((hy << 20) | (hx << 4) | s) : (0x7ff << 20);
    }
    // ---- weight DMA pieces of this lane: byte offset inside this workgroup's CBW packed rows
    unsigned woff[WI];
    int wdst[WI];
#pragma unroll
    for (int i = 0; i < WI; ++i) {
        const int wi = (i * NW + wave) % NWW;
        const int u = wi * 64 + lane, row = u >> 2, sp = u & 3;
        const int f = (4 - ((row >> 3) & 3)) & 3;
        woff[i] = (unsigned)(((long long)row * wrow + ((sp ^ f) << 3)) * 2);
        wdst[i] = wi * 512;
    }
    const __amdgpu_buffer_rsrc_t wrsrc = hyb_rsrc((wp + (long long)co_wg * wrow), V2_RECORDS);
    // ---- fragment addresses.  Row (t, p) of the MFMA tile holds channel (t>>1)*32 + (p>>2)*8 + (t&1)*4 + (p&3).
    const int aoff0 = (cb * (NT * 16) + (p >> 2) * 8 + (p & 3)) * 32 + ((q ^ ((4 - (p >> 2)) & 3)) << 3);
    auto aoff = [&](int t) { return aoff0 + ((t >> 1) * 32 + (t & 1) * 4) * 32; };
    const int lane_el = ((prow * 4 + py) * HW_ + pstrip * 28 + px) * 32;

    // block descriptors (wave-uniform)
    struct Blk { int n, ty0, tx0, cblk; };
    auto decode = [&](int tile, int cblk) {
        Blk b;
        b.n = tile / (tilesX * tilesY);
        const int trem = tile - b.n * (tilesX * tilesY);
        b.ty0 = (trem / tilesX) * TH;
        b.tx0 = (trem % tilesX) * TW;
        b.cblk = cblk;
        return b;
    };
    // descriptor of a block's halo: base = the halo origin pixel (may lie before the tensor for border tiles: such lanes are
    // sent out of range and never dereference it)
    auto halo_rsrc = [&](const Blk& b) {
        const long long base = ((long long)(b.n * H + b.ty0 - 1) * W + (b.tx0 - 1)) * xpix + b.cblk * xblk;
        return hyb_rsrc((x + base), V2_RECORDS);
    };
    auto halo_piece = [&](const Blk& b, __amdgpu_buffer_rsrc_t rs, bf16* hb, int k) {
        const int hy = hyx[k] >> 20, hx = (hyx[k] >> 4) & 0xffff;
        const int gy = b.ty0 - 1 + hy, gx = b.tx0 - 1 + hx;
        const bool valid = ((unsigned)gy < (unsigned)H) && ((unsigned)gx < (unsigned)W);
        const unsigned off = HOFF_REG ? hoff[HOFF_REG ? k : 0] : (unsigned)(((hy * W + hx) * xpix + (hyx[k] & 15) * 8) * 2);
        int wi = k * NW + wave;
        if (wi > NHW - 1) wi = NHW - 1;
        dma16(rs, valid ? off : V2_OOB, 0, hb + wi * 512);
    };
    auto weight_pieces = [&](int cblk, int tap, int slot) {
        const unsigned so = (unsigned)((cblk * 9 + tap) * 64);
#pragma unroll
        for (int i = 0; i < WI; ++i) dma16(wrsrc, woff[i], so, wring + slot * WSLOT + wdst[i]);
    };
    auto bptr_of = [&](const bf16* hb, int tap) {
        const int kh = tap / 3, kw = tap % 3;
        const int sw = ((py + kh) & 1) << 1;
        return hb + lane_el + (kh * HW_ + kw) * 32 + ((q ^ sw) << 3);
    };

    // a workgroup walks a contiguous run of tiles (row-major inside an image): the halo rows/columns neighbouring tiles share are
    // re-read from this XCD's L2.  The host sizes the grid so that no run is empty.
    const int tchunk = (numTiles + (int)gridDim.x - 1) / (int)gridDim.x;
    int tile = blockIdx.x * tchunk;
    const int tile_end = tile + tchunk < numTiles ? tile + tchunk : numTiles;
    Blk cur = decode(tile, 0);
    int hsel = 0, slot_cur = 0;
    bool after_store = false;                       // the previous block ended with a full-tile epilogue (NS stores per wave)
    constexpr int NS = MT * ((NT + 1) / 2);

    // ---- prologue: halo of the first block, weights of steps 0 .. R-2
    {
        const __amdgpu_buffer_rsrc_t rs0 = halo_rsrc(cur);
#pragma unroll
        for (int k = 0; k < HT; ++k) halo_piece(cur, rs0, hbuf, k);
    }
#pragma unroll
    for (int s = 0; s < R - 1; ++s) weight_pieces(0, s, s);
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();

    Frag<bf16> a_cur[NT], bfr[MT];
#pragma unroll
    for (int t = 0; t < NT; ++t) frag_load(a_cur[t], wring + aoff(t));
    {
        const bf16* bp = bptr_of(hbuf, 0);
#pragma unroll
        for (int m = 0; m < MT; ++m) frag_load(bfr[m], bp + m * 128);
    }

    f32x4 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    while (true) {
        // next block: the following channel block of this tile, else the first block of this workgroup's next tile
        const bool last_cblk = cur.cblk + 1 >= nCblk;
        const int ntile = last_cblk ? (tile + 1 < tile_end ? tile + 1 : tile) : tile;
        const Blk nxt = decode(ntile, last_cblk ? 0 : cur.cblk + 1);
        bf16* const hb_cur = hbuf + hsel * G::HBUF;
        bf16* const hb_nxt = hbuf + (hsel ^ 1) * G::HBUF;
        const __amdgpu_buffer_rsrc_t nrs = halo_rsrc(nxt);

        auto step = [&](auto S_) __attribute__((always_inline)) {
            constexpr int S = decltype(S_)::value;
            // (1) DMA: weights of step S+R-1 into the slot freed by step S-1, then this step's share of the next halo
            {
                constexpr int FT = (S + R - 1) % 9;
                const int fc = (S + R - 1 >= 9) ? nxt.cblk : cur.cblk;
                int slot_fill = slot_cur - 1;
                if (slot_fill < 0) slot_fill += R;
                weight_pieces(fc, FT, slot_fill);
#pragma unroll
                for (int k = 0; k < G::hi(S); ++k) halo_piece(nxt, nrs, hb_nxt, G::hstart(S) + k);
            }
            // (2) A fragments of step S+1 (their slot was published by the barrier that ended step S-1)
            int slot_next = slot_cur + 1;
            if (slot_next >= R) slot_next -= R;
            Frag<bf16> a_nxt[NT];
            {
                const bf16* as = wring + slot_next * WSLOT;
#pragma unroll
                for (int t = 0; t < NT; ++t) frag_load(a_nxt[t], as + aoff(t));
            }
            // (3) MFMAs of step S; each patch fragment is replaced by the one step S+1 needs as soon as it has been used
            const bf16* bp = bptr_of(S == 8 ? hb_nxt : hb_cur, (S + 1) % 9);
#pragma unroll
            for (int m = 0; m < MT; ++m) {
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[m][t] = mma32(a_cur[t], bfr[m], acc[m][t]);
                frag_load(bfr[m], bp + m * 128);
            }
            // pin the interleave: the LDS reads of step S+1 trickle out between the MFMA groups of step S
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                __builtin_amdgcn_sched_group_barrier(0x008, NT, 0);                      // MFMA
                if (m < NT) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);           // DS read (A of S+1 first, then patches)
                else __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) a_cur[t] = a_nxt[t];
            slot_cur = slot_next;
            // (4) the weights of step S+2 (and everything older) have landed for this wave; publish
            //     After a full-tile epilogue the NS output stores are younger than the weights waited for in the first R-3 steps
            //     (those were issued before the epilogue): counting them keeps the stores in flight instead of draining them.
            if constexpr (S < R - 3) {
                if (after_store) wait_vmcnt<G::wait_n(S) + NS>(); else wait_vmcnt<G::wait_n(S)>();
            } else {
                wait_vmcnt<G::wait_n(S)>();
            }
            __builtin_amdgcn_s_barrier();
        };
        step(step_c<0>{}); step(step_c<1>{}); step(step_c<2>{}); step(step_c<3>{}); step(step_c<4>{});
        step(step_c<5>{}); step(step_c<6>{}); step(step_c<7>{}); step(step_c<8>{});

        if (last_cblk) {
            // ---- epilogue: the lane holds 8 consecutive channels per 32-channel half of one pixel per patch
            const int n = cur.n, ty0 = cur.ty0, tx0 = cur.tx0;
            const bool full = (ty0 + TH <= H) && (tx0 + TW <= W);
            float s1[NT][4], s2[NT][4];
            if (STATS) {
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) { s1[t][r] = 0.f; s2[t][r] = 0.f; }
            }
            const int gy = ty0 + prow * 4 + py;
            auto emit = [&](auto FULL_) __attribute__((always_inline)) {
                constexpr bool FULL = decltype(FULL_)::value;
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const int gx = tx0 + pstrip * 28 + m * 4 + px;
                    const bool valid = FULL || ((gy < H) && (gx < W));
                    if (valid) {
                        bf16* dst = y + ((long long)(n * H + gy) * W + gx) * Cop + co_base + q * 8;
#pragma unroll
                        for (int h = 0; h < (NT + 1) / 2; ++h) {
                            Vec8<bf16> v;
#pragma unroll
                            for (int j = 0; j < 8; ++j) v.set(j, acc[m][NT == 1 ? 0 : h * 2 + (j >> 2)][j & 3]);
                            v.store(dst + h * 32);
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
#pragma unroll
                    for (int t = 0; t < NT; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            };
            if (full) {
                emit(std::true_type{});          // exactly NS store instructions per wave: the next waits step over them by count
                after_store = true;
            } else {
                emit(std::false_type{});         // edge tile: the store count depends on the lane masks, so drain instead
                wait_vmcnt<0>();
                after_store = false;
            }
            if (STATS) {
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float a = row16_sum(s1[t][r]), b = row16_sum(s2[t][r]);
                        if (p == 0) {
                            const int cl = (t >> 1) * 32 + q * 8 + (t & 1) * 4 + r;
                            wgstat[(wave * 2 + 0) * (NT * 16) + cl] += a;
                            wgstat[(wave * 2 + 1) * (NT * 16) + cl] += b;
                        }
                    }
            }
            if (tile + 1 >= tile_end) break;
            ++tile;
        }
        else after_store = false;
        cur = nxt;
        hsel ^= 1;
    }

    wait_vmcnt<0>();                 // the run-ahead DMA of the (non-existent) next block must not outlive the workgroup
    if (STATS) {
        __syncthreads();
        for (int i = tid; i < 2 * CBW; i += NTHR) {
            const int which = i / CBW, cl = i % CBW;
            const int cbi = cl / (NT * 16), c16 = cl % (NT * 16);
            float a = 0.f;
#pragma unroll
            for (int g = 0; g < PG; ++g) a += wgstat[((g * CB + cbi) * 2 + which) * (NT * 16) + c16];     // wave = pg*CB + cb
            stats[((long long)blockIdx.x * 2 + which) * Cop + co_wg + cl] = a;
            // the caller sums a fixed number of partial rows: rows no workgroup owns are zero
            for (int rrow = blockIdx.x + gridDim.x; rrow < stat_rows; rrow += gridDim.x) stats[((long long)rrow * 2 + which) * Cop + co_wg + cl] = 0.f;
        }
    }
}

// ---- 32 input channels (one channel block per tile): the weights live in REGISTERS ------------------------------------------------------
// conv3x3_v2_kernel synchronises once per (tap, channel block) step -- the price of streaming weight slices through an LDS ring.  With
// Cip = 32 a step is only MT x NT = 14 MFMAs per wave (224 matrix-pipe cycles) and the barrier, the counted wait and the ring traffic cost as
// much as the arithmetic: stage 2's forward conv (32 -> 64 channels, 308 MB) ran at 3.6 TB/s, memory-bound on paper.  Here a wave keeps all
// nine taps' fragments of its 32 output channels (72 registers) for the whole kernel, the halo images are a ring of THREE filled two tiles
// ahead by LDS-DMA, and a tile costs ONE counted wait + ONE barrier.  Same tiles, fragment layout, swizzle, epilogue and statistics order
// per tile as conv3x3_v2_kernel<2, CB, PGR, PGC, .>; the per-lane statistics are folded across lanes once, at the end.
template <int CB, int PGR, int PGC, bool STATS>
__global__ __launch_bounds__(CB * PGR * PGC * 64, 2) void conv3x3_k32_kernel(const bf16* __restrict__ x, const bf16* __restrict__ wp, bf16* __restrict__ y,
                                                                            float* __restrict__ stats, int N, int H, int W, int Cip, int Cop, int tilesX,
                                                                            int tilesY, int numTiles, int stat_rows, int xpix, long long xblk) {
    constexpr int NT = 2, HBN = 3;
    using G = V2Geom<NT, CB, PGR, PGC, 3>;
    constexpr int MT = G::MT, TH = G::TH, TW = G::TW, HW_ = G::HW_, HP = G::HP, HT = G::HT, NHW = G::NHW, CBW = G::CBW, PG = G::PG, NW = G::NW,
                  NTHR = NW * 64;
    static_assert(NW == 4, "two four-wave workgroups per CU");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16* const hbuf = reinterpret_cast<bf16*>(smem_raw);                       // [3][HBUF]
    float* const wgstat = reinterpret_cast<float*>(hbuf + HBN * G::HBUF);       // [NW waves][2][NT*16]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cb = wave % CB, pg = wave / CB;
    const int prow = pg / PGC, pstrip = pg % PGC;
    const int p = lane & 15, q = lane >> 4, py = p >> 2, px = p & 3;
    const int co_wg = blockIdx.y * CBW;
    const int co_base = co_wg + cb * (NT * 16);
    const long long wrow = (long long)9 * Cip;
    (void)xblk;

    // the wave's weights: row (t, p) of the MFMA tile holds channel (t >> 1) * 32 + (p >> 2) * 8 + (t & 1) * 4 + (p & 3) (conv3x3_v2_kernel)
    Frag<bf16> aw[9][NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const bf16* row = wp + (long long)(co_base + (t >> 1) * 32 + (p >> 2) * 8 + (t & 1) * 4 + (p & 3)) * wrow + 8 * q;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) frag_load(aw[tap][t], row + tap * 32);
    }
    unsigned hoff[HT];
    int hyx[HT];
#pragma unroll
    for (int k = 0; k < HT; ++k) {
        int wi = k * NW + wave;
        if (wi > NHW - 1) wi = NHW - 1;
        const int u = wi * 64 + lane, hp = u >> 2, sp = u & 3;
        const int hy = hp / HW_, hx = hp - hy * HW_;
        const int s = sp ^ ((hy & 1) << 1);
        hoff[k] = (unsigned)(((hy * W + hx) * xpix + s * 8) * 2);
        hyx[k] = hp < HP ? ((hy << 20) | (hx << 4) | s) : (0x7ff << 20);
    }
    const int lane_el = ((prow * 4 + py) * HW_ + pstrip * 28 + px) * 32;
    struct Tl { int n, ty0, tx0; };
    auto decode = [&](int tile) {
        Tl b;
        b.n = tile / (tilesX * tilesY);
        const int trem = tile - b.n * (tilesX * tilesY);
        b.ty0 = (trem / tilesX) * TH;
        b.tx0 = (trem % tilesX) * TW;
        return b;
    };
    auto halo_dma = [&](const Tl& b, bf16* hb) {
        const long long base = ((long long)(b.n * H + b.ty0 - 1) * W + (b.tx0 - 1)) * xpix;
        const __amdgpu_buffer_rsrc_t rs = hyb_rsrc((x + base), V2_RECORDS);
#pragma unroll
        for (int k = 0; k < HT; ++k) {
            const int hy = hyx[k] >> 20, hx = (hyx[k] >> 4) & 0xffff;
            const int gy = b.ty0 - 1 + hy, gx = b.tx0 - 1 + hx;
            const bool valid = ((unsigned)gy < (unsigned)H) && ((unsigned)gx < (unsigned)W);
            int wi = k * NW + wave;
            if (wi > NHW - 1) wi = NHW - 1;
            dma16(rs, valid ? hoff[k] : V2_OOB, 0, hb + wi * 512);
        }
    };
    auto bptr_of = [&](const bf16* hb, int tap) {
        const int kh = tap / 3, kw = tap % 3;
        const int sw = ((py + kh) & 1) << 1;
        return hb + lane_el + (kh * HW_ + kw) * 32 + ((q ^ sw) << 3);
    };

    const int tchunk = (numTiles + (int)gridDim.x - 1) / (int)gridDim.x;
    const int tile0 = blockIdx.x * tchunk;
    const int tile_end = tile0 + tchunk < numTiles ? tile0 + tchunk : numTiles;
    const int ntiles = tile_end - tile0;
    auto tl = [&](int i) { return decode(tile0 + (i < ntiles ? i : ntiles - 1)); };
    constexpr int NS = MT * ((NT + 1) / 2);

    halo_dma(tl(0), hbuf);
    halo_dma(tl(1), hbuf + G::HBUF);
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();

    Frag<bf16> bfr[MT];
    {
        const bf16* bp = bptr_of(hbuf, 0);
#pragma unroll
        for (int m = 0; m < MT; ++m) frag_load(bfr[m], bp + m * 128);
    }
    f32x4 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float s1[NT][4], s2[NT][4];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) { s1[t][r] = 0.f; s2[t][r] = 0.f; }

    int hsel = 0;
    bool prev_full = false;
    for (int it = 0; it < ntiles; ++it) {
        const Tl cur = tl(it);
        const int h1 = hsel + 1 >= HBN ? 0 : hsel + 1, h2 = h1 + 1 >= HBN ? 0 : h1 + 1;
        const bf16* hb_cur = hbuf + hsel * G::HBUF;
        const bf16* hb_nxt = hbuf + h1 * G::HBUF;
        halo_dma(tl(it + 2), hbuf + h2 * G::HBUF);            // two tiles ahead; its image was tile it - 1's: every wave left it at the last barrier
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const bf16* bp = bptr_of(tap == 8 ? hb_nxt : hb_cur, (tap + 1) % 9);
#pragma unroll
            for (int m = 0; m < MT; ++m) {
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[m][t] = mma32(aw[tap][t], bfr[m], acc[m][t]);
                frag_load(bfr[m], bp + m * 128);
            }
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                __builtin_amdgcn_sched_group_barrier(0x008, NT, 0);                      // MFMA
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                       // DS read
            }
        }
        // ---- epilogue (conv3x3_v2_kernel's): the lane holds 8 consecutive channels of one pixel per patch
        const int n = cur.n, ty0 = cur.ty0, tx0 = cur.tx0;
        const bool full = (ty0 + TH <= H) && (tx0 + TW <= W);
        const int gy = ty0 + prow * 4 + py;
        auto emit = [&](auto FULL_) __attribute__((always_inline)) {
            constexpr bool FULL = decltype(FULL_)::value;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int gx = tx0 + pstrip * 28 + m * 4 + px;
                const bool valid = FULL || ((gy < H) && (gx < W));
                if (valid) {
                    bf16* dst = y + ((long long)(n * H + gy) * W + gx) * Cop + co_base + q * 8;
                    Vec8<bf16> v;
#pragma unroll
                    for (int j = 0; j < 8; ++j) v.set(j, acc[m][j >> 2][j & 3]);
                    v.store(dst);
                    if (STATS) {
#pragma unroll
                        for (int t = 0; t < NT; ++t)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const float a = acc[m][t][r];
                                s1[t][r] += a;
                                s2[t][r] = fmaf(a, a, s2[t][r]);
                            }
                    }
                }
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        };
        // the next tile's halo (issued one iteration ago) must have landed: in issue order the outstanding operations are [this wave's DMA pieces of
        // iteration it - 1] [its NS stores, if that tile was full] [HT pieces of this iteration] [NS stores]: leave the last three groups in flight
        if (full) {
            emit(std::true_type{});
            if (prev_full) wait_vmcnt<HT + 2 * NS>(); else wait_vmcnt<HT + NS>();
        } else {
            emit(std::false_type{});         // edge tile: the store count depends on the lane masks, so drain
            wait_vmcnt<0>();
        }
        prev_full = full;
        __builtin_amdgcn_s_barrier();
        hsel = h1;
    }
    wait_vmcnt<0>();                 // the run-ahead DMA must not outlive the workgroup
    if (STATS) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float a = row16_sum(s1[t][r]), b = row16_sum(s2[t][r]);
                if (p == 0) {
                    const int cl = (t >> 1) * 32 + q * 8 + (t & 1) * 4 + r;
                    wgstat[(wave * 2 + 0) * (NT * 16) + cl] = a;
                    wgstat[(wave * 2 + 1) * (NT * 16) + cl] = b;
                }
            }
        __syncthreads();
        for (int i = tid; i < 2 * CBW; i += NTHR) {
            const int which = i / CBW, cl = i % CBW;
            const int cbi = cl / (NT * 16), c16 = cl % (NT * 16);
            float a = 0.f;
#pragma unroll
            for (int g = 0; g < PG; ++g) a += wgstat[((g * CB + cbi) * 2 + which) * (NT * 16) + c16];     // wave = pg*CB + cb
            stats[((long long)blockIdx.x * 2 + which) * Cop + co_wg + cl] = a;
            for (int rrow = blockIdx.x + gridDim.x; rrow < stat_rows; rrow += gridDim.x) stats[((long long)rrow * 2 + which) * Cop + co_wg + cl] = 0.f;
        }
    }
}

template <int CB, int PGR, int PGC>
int launch_k32(const bf16* x, const bf16* wp, bf16* y, float* part, int N, int H, int W, int Cip, int Cop, int stat_rows, hipStream_t st, int xpix,
               long long xblk) {
    using G = V2Geom<2, CB, PGR, PGC, 3>;
    constexpr size_t LDS = (size_t)3 * G::HBUF * 2 + G::STAT_FLOATS * 4;
    static_assert(LDS <= 80 * 1024, "two workgroups per CU");
    const int tilesX = hyb_cdiv(W, G::TW), tilesY = hyb_cdiv(H, G::TH);
    const long long numTiles = (long long)N * tilesX * tilesY;
    int gx = (int)(numTiles < 512 ? numTiles : 512);
    if (part && gx > stat_rows) gx = stat_rows;
    if (gx < 1) gx = 1;
    gx = hyb_cdiv(numTiles, hyb_cdiv(numTiles, gx));
    const dim3 grid(gx, Cop / G::CBW);
    static HybAttrOnce once_stats, once_plain;
    if (int e = hyb_set_lds_attr(once_stats, (const void*)conv3x3_k32_kernel<CB, PGR, PGC, true>, (int)LDS)) return e;
    if (int e = hyb_set_lds_attr(once_plain, (const void*)conv3x3_k32_kernel<CB, PGR, PGC, false>, (int)LDS)) return e;
    if (part)
        hipLaunchKernelGGL((conv3x3_k32_kernel<CB, PGR, PGC, true>), grid, dim3(256), LDS, st, x, wp, y, part, N, H, W, Cip, Cop, tilesX, tilesY, (int)numTiles,
                           stat_rows, xpix, xblk);
    else
        hipLaunchKernelGGL((conv3x3_k32_kernel<CB, PGR, PGC, false>), grid, dim3(256), LDS, st, x, wp, y, (float*)nullptr, N, H, W, Cip, Cop, tilesX, tilesY,
                           (int)numTiles, 0, xpix, xblk);
    HYB_LAUNCH_CHECK();
    return 0;
}

template <int NT, int CB, int PGR, int PGC, int R>
int launch_v2(const bf16* x, const bf16* wp, bf16* y, float* part, int N, int H, int W, int Cip, int Cop, int stat_rows, hipStream_t st, int xpix,
              long long xblk) {
    using G = V2Geom<NT, CB, PGR, PGC, R>;
    const int tilesX = hyb_cdiv(W, G::TW), tilesY = hyb_cdiv(H, G::TH);
    const long long numTiles = (long long)N * tilesX * tilesY;
    constexpr int SLOTS = G::NW == 8 ? 256 : 512;           // resident workgroups on 256 CUs
    int gx = (int)(numTiles < SLOTS ? numTiles : SLOTS);
    if (part && gx > stat_rows) gx = stat_rows;
    if (gx < 1) gx = 1;
    gx = hyb_cdiv(numTiles, hyb_cdiv(numTiles, gx));          // contiguous runs of ceil(numTiles / gx) tiles: drop the empty ones
    const dim3 grid(gx, Cop / G::CBW);
    static HybAttrOnce once_stats, once_plain;                 // per template instantiation, per device
    if (int e = hyb_set_lds_attr(once_stats, (const void*)conv3x3_v2_kernel<NT, CB, PGR, PGC, R, true>, (int)G::LDS_BYTES)) return e;
    if (int e = hyb_set_lds_attr(once_plain, (const void*)conv3x3_v2_kernel<NT, CB, PGR, PGC, R, false>, (int)G::LDS_BYTES)) return e;
    if (part)
        hipLaunchKernelGGL((conv3x3_v2_kernel<NT, CB, PGR, PGC, R, true>), grid, dim3(G::NW * 64), G::LDS_BYTES, st, x, wp, y, part, N, H, W, Cip, Cop,
                           tilesX, tilesY, (int)numTiles, stat_rows, xpix, xblk);
    else
        hipLaunchKernelGGL((conv3x3_v2_kernel<NT, CB, PGR, PGC, R, false>), grid, dim3(G::NW * 64), G::LDS_BYTES, st, x, wp, y, (float*)nullptr, N, H, W,
                           Cip, Cop, tilesX, tilesY, (int)numTiles, 0, xpix, xblk);
    HYB_LAUNCH_CHECK();
    return 0;
}

// relative cost of covering an H x W image with TH x TW tiles on 256 persistent workgroups
double v2_cost(int N, int H, int W, int TH, int TW, int gy) {
    const long long tiles = (long long)N * hyb_cdiv(W, TW) * hyb_cdiv(H, TH) * gy;
    const long long rounds = (tiles + 255) / 256;
    return (double)rounds * TH * TW;
}

}  // namespace

// Internal (conv_fwd.hip): returns -100 when no asynchronous variant fits this shape.  part: partial-statistics rows
// [stat_rows][2][Cop] (may be NULL), all of them written.
// which shapes hyb_conv_v2 takes (the same tests as below)
int hyb_conv_v2_supported(int W, int Cip, int Cop) {
    return !(Cip % 32 != 0 || (long long)40 * W * Cip >= (1ll << 29) || (long long)256 * 9 * Cip >= (1ll << 29)) && Cop % 32 == 0;
}

// xblk = 0: NHWC input; else the block-planar input's block stride in elements (see conv3x3_v2_kernel)
int hyb_conv_v2(const void* x, const void* wp, void* y, float* part, int N, int H, int W, int Cip, int Cop, int stat_rows, hipStream_t st, long long xblk) {
    const int xpix = xblk ? 32 : Cip;
    if (!xblk) xblk = 32;
    if (Cip % 32 != 0 || (long long)40 * W * Cip >= (1ll << 29) || (long long)256 * 9 * Cip >= (1ll << 29)) return -100;   // 32-bit buffer offsets
    const bf16* xb = (const bf16*)x; const bf16* wb = (const bf16*)wp; bf16* yb = (bf16*)y;
#define V2(NT_, CB_, PGR_, PGC_, R_) launch_v2<NT_, CB_, PGR_, PGC_, R_>(xb, wb, yb, part, N, H, W, Cip, Cop, stat_rows, st, xpix, xblk)
    // Measured on the 224 x 224 clip stages: two four-wave workgroups per CU (their epilogues and MFMA phases interleave) win for
    // Cop <= 128; 256-channel blocks need the whole CU's LDS for a deep weight ring.  HYB_V2_NW=4|8 forces one family.
    static const int nw_env = getenv("HYB_V2_NW") ? atoi(getenv("HYB_V2_NW")) : 0;
    const int nw = nw_env ? nw_env : (Cop % 256 == 0 ? 8 : 4);
    static const int k32_env = getenv("HYB_CONV_K32") ? atoi(getenv("HYB_CONV_K32")) : 1;      // (=0: A/B, the ring kernel for 32 input channels too)
    if (k32_env && nw == 4 && Cip == 32 && Cop % 64 == 0) {
        // one channel block per tile: weights in registers, one barrier per tile (conv3x3_k32_kernel)
        return v2_cost(N, H, W, 8, 28, Cop / 64) <= v2_cost(N, H, W, 4, 56, Cop / 64)
                   ? launch_k32<2, 2, 1>(xb, wb, yb, part, N, H, W, Cip, Cop, stat_rows, st, xpix, xblk)
                   : launch_k32<2, 1, 2>(xb, wb, yb, part, N, H, W, Cip, Cop, stat_rows, st, xpix, xblk);
    }
    if (nw == 4) {
        if (Cop % 256 == 0) return V2(4, 4, 1, 1, 3);
        if (Cop % 128 == 0) return v2_cost(N, H, W, 8, 28, Cop / 128) <= v2_cost(N, H, W, 4, 56, Cop / 128) ? V2(4, 2, 2, 1, 4) : V2(4, 2, 1, 2, 4);
        if (Cop % 64 == 0) return v2_cost(N, H, W, 8, 28, Cop / 64) <= v2_cost(N, H, W, 4, 56, Cop / 64) ? V2(2, 2, 2, 1, 4) : V2(2, 2, 1, 2, 4);
        if (Cop % 32 == 0) return V2(2, 1, 4, 1, 4);
        return -100;
    }
    if (Cop % 256 == 0) {
        return v2_cost(N, H, W, 8, 28, Cop / 256) <= v2_cost(N, H, W, 4, 56, Cop / 256) ? V2(4, 4, 2, 1, 6) : V2(4, 4, 1, 2, 6);
    }
    if (Cop % 128 == 0) {
        return v2_cost(N, H, W, 16, 28, Cop / 128) <= v2_cost(N, H, W, 8, 56, Cop / 128) ? V2(4, 2, 4, 1, 6) : V2(4, 2, 2, 2, 6);
    }
    if (Cop % 64 == 0) return V2(4, 1, 4, 2, 5);
    if (Cop % 32 == 0) return V2(2, 1, 4, 2, 6);
#undef V2
    return -100;
}
