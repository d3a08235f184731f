// Third generation of the fused weight-gradient kernel (bf16, Cip % 64 == 0, Cop % 64 == 0, W % 28 == 0, H % 4 == 0).
// Included by conv_wgrad.hip inside its anonymous namespace (uses WgradFuse, hyb_rsrc, the W2 tile constants' conventions).
//
// What the round-3 ablations of the second generation showed (scripts/micro/wgrad_bench, profiles/r03_wgrad_ablation.txt): with the
// consumer waves idle the producers alone took 66 / 52 us (stages 3 / 4), with the producers idle the consumers alone 58 / 61 us,
// together 108 / 97 us -- the two halves overlapped badly, and the consumer loop reloaded spilled registers from scratch at its
// 256-register limit.  Changes:
//   * v_mfma_f32_32x32x16_bf16 instead of 16x16x32: the same FLOP per cycle, but an MFMA then blocks the SIMD's vector issue for
//     8 of 32 cycles instead of 8 of 16, which leaves the producer wave that shares the SIMD three times the issue slots; half the
//     matrix instructions, 9 accumulator tiles (144 registers) per consumer wave, no spills;
//   * the producers keep their eight channels' BatchNorm constants in registers (a thread's channel octet never changes) instead of
//     re-reading five LDS words per element; offsets of the second window row pair are uniform increments; no per-register
//     validity selects (the shapes taken here have no ragged tiles);
//   * H = 28 (8 does not divide it): the last tile of an image is a HALF tile -- the producers skip its second unit and the
//     consumers its second half instead of contracting four rows of zeros (12.5 % of stage 4);
//   * the last iteration no longer re-stages the last tile (1/14 of the producers' traffic and stores).
// Work split as before: 4 consumer waves (one per SIMD; wave = 32 co x 32 ci x 9 taps) + 4 producer waves, two tile-image pairs in
// LDS, one barrier per tile.  Images: 128-byte pixels, 16-byte chunks XOR-swizzled by bit 1 of the pixel column, which makes both the
// transposed fragment reads (4 consecutive pixels x 64 bytes per 32-lane group) and the producers' 128-byte pixel stores
// bank-conflict free.
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;

constexpr int W3_TW = 28, W3_HW = 30, W3_HP = 10 * 30, W3_PX = 8 * 28;
constexpr int W3_XW = (W3_HP * 8 + 63) / 64;          // x-halo DMA wave-instructions (38)
constexpr int W3_XBUF = W3_XW * 512;                  // bf16 elements per x-halo buffer
constexpr int W3_DBUF = W3_PX * 64;                   // bf16 elements per gradient-tile buffer
#ifndef W3_DEFAULT
#define W3_DEFAULT 1    // generation of the fused kernel: 0 second (conv_wgrad.hip), 1 third (this file); 2, 3: experiment builds only
#endif
#ifndef W3_RING
#define W3_RING 6
#endif
constexpr size_t W3_LDS = (size_t)2 * (W3_XBUF + W3_DBUF) * 2;

struct W3Tile { int n, ty0, tx0, rows; };
__device__ __forceinline__ W3Tile w3_tile(int tile, int tilesX, int tilesY, int H) {
    W3Tile t;
    t.n = tile / (tilesX * tilesY);
    const int trem = tile - t.n * (tilesX * tilesY);
    const int ty = trem / tilesX;
    t.ty0 = ty * 8;
    t.tx0 = (trem - ty * tilesX) * W3_TW;
    t.rows = H - t.ty0 < 8 ? H - t.ty0 : 8;           // 8, or 4 for the half tile at the bottom of an image
    return t;
}

struct W3Unit { union { u32x4 u; bf16x8 v; } y[4], g; };

__global__ __launch_bounds__(512) void wgrad_v3_kernel(const bf16* __restrict__ x, float* __restrict__ slab, int N, int H, int W, int Cip, int Cop,
                                                       int tilesX, int tilesY, int numTiles, WgradFuse fz) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16* const xbuf = reinterpret_cast<bf16*>(smem_raw);                 // [2][W3_XBUF]
    bf16* const dbuf = xbuf + 2 * W3_XBUF;                                // [2][W3_DBUF]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nCiBlk = Cip >> 6;
    const int co0 = (blockIdx.y / nCiBlk) * 64, ci0 = (blockIdx.y % nCiBlk) * 64;
    const int tchunk = (numTiles + (int)gridDim.x - 1) / (int)gridDim.x;   // contiguous run of tiles (the host leaves no run empty)
    const int tbegin = blockIdx.x * tchunk;
    const int tcount = (tbegin + tchunk < numTiles ? tbegin + tchunk : numTiles) - tbegin;

    const int pw = wave & 3;
// x-halo DMA pieces of this wave: k * 4 + pw, k = 0..9 (pieces past 37 repeat piece 37)
    unsigned xoff[10];
    int xyx[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) {
        int wi = k * 4 + pw;
        if (wi > W3_XW - 1) wi = W3_XW - 1;
        const int u = wi * 64 + lane, hp = u >> 3, cp = u & 7;
        const int hy = hp / W3_HW, hx = hp - hy * W3_HW;
        xoff[k] = (unsigned)(((hy * W + hx) * Cip + ((cp ^ (((hx >> 1) & 1) << 2)) << 3)) * 2);
        xyx[k] = hp < W3_HP ? ((hy << 16) | hx) : (0x7fff << 16);
    }
    auto x_dma = [&](const W3Tile& t, bf16* xb) {
        const long long base = ((long long)(t.n * H + t.ty0 - 1) * W + (t.tx0 - 1)) * Cip + ci0;
        const __amdgpu_buffer_rsrc_t rs = hyb_rsrc(x + base, W2_RECORDS);
#pragma unroll
        for (int k = 0; k < 10; ++k) {
            const int gy = t.ty0 - 1 + (xyx[k] >> 16), gx = t.tx0 - 1 + (xyx[k] & 0xffff);
            const bool valid = ((unsigned)gy < (unsigned)H) && ((unsigned)gx < (unsigned)W);
            int wi = k * 4 + pw;
            if (wi > W3_XW - 1) wi = W3_XW - 1;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)(xb + wi * 512), 16, valid ? xoff[k] : W2_OOB, 0, 0, 0);
        }
    };

    if (wave >= 4) {
        // ================================================= producers =================================================
        const int ptid = tid - 256;
        const int oct = ptid & 7, wslot = ptid >> 3;
        const bool dup = wslot >= 28;                     // 28 windows per window-row pair: slots 28..31 repeat window 27 (same values, same LDS
        const int w0 = dup ? 27 : wslot;                  // addresses); their global stores are dropped
        const int wy = w0 / 14, wx = w0 - wy * 14;
        // per-channel constants of this thread's octet:  v = sc*y + sh (arg-max / ReLU gate);  dyraw = A1*y + A0 + (arg-max ? k*dy : 0)
        float sc[8], sh[8], kk[8], a1[8], a0[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int ch = co0 + oct * 8 + e;
            sc[e] = fz.ss[ch]; sh[e] = fz.ss[Cop + ch];
            const float mean = fz.mi[ch], inv = fz.mi[Cop + ch];
            const float k = (ch < fz.Co ? fz.gamma[ch] : 0.f) * inv;
            const float m1 = fz.training ? fz.sums[ch] * fz.inv_count : 0.f, m2 = fz.training ? fz.sums[Cop + ch] * fz.inv_count : 0.f;
            kk[e] = k; a1[e] = -k * m2 * inv; a0[e] = -k * m1 + k * m2 * inv * mean;
        }
        // unit i of this thread = window (wy + 2 i, wx): byte offsets of its four pixels / its pooled element from the tile's base
        unsigned yoff[4], ooff[4];
        const bool planar = fz.dyraw_blk != 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int pix = (2 * wy + (j >> 1)) * W + 2 * wx + (j & 1);
            yoff[j] = (unsigned)((pix * Cop + 8 * oct) * 2);
            ooff[j] = planar ? (unsigned)((pix * 32 + 8 * (oct & 3)) * 2 + (oct >> 2) * fz.dyraw_blk * 2) : yoff[j];
            if (dup) ooff[j] = W2_OOB;
        }
        const unsigned goff = (unsigned)(((wy * (W >> 1) + wx) * Cop + 8 * oct) * 2);
        const unsigned y_i = (unsigned)(4 * W * Cop * 2), g_i = (unsigned)(2 * (W >> 1) * Cop * 2);
        const unsigned o_i = dup ? 0u : (planar ? (unsigned)(4 * W * 32 * 2) : y_i);      // (a dropped store stays out of range)
        const int lds0 = ((2 * wy) * W3_TW + 2 * wx) * 64 + ((oct ^ ((wx & 1) << 2)) << 3);      // element offset of pixel j = 0 of unit 0
        const bool writer = fz.dyraw_out && ci0 == 0;

        // Every iteration issues the SAME number of vector-memory operations (10 prefetch loads, 10 DMAs, 8 stores): what a half tile
        // or the end of the run does not need goes through a descriptor of zero records (loads return zeros, stores are dropped, no
        // memory traffic).  With conditional loads the compiler's wait-count bookkeeping has to assume the shortest path, and its
        // wait in front of the first use of a prefetched register then also waited for loads issued a few cycles earlier -- a full
        // memory round trip per tile on the producers' critical path.
        auto fuse_load = [&](const W3Tile& t, bool live, W3Unit (&un)[2]) {
            const int Ho = H >> 1, Wo = W >> 1;
            const bf16* yp = (const bf16*)fz.y + ((long long)(t.n * H + t.ty0) * W + t.tx0) * Cop + co0;
            const bf16* gp = (const bf16*)fz.dp + ((long long)(t.n * Ho + (t.ty0 >> 1)) * Wo + (t.tx0 >> 1)) * Cop + co0;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const unsigned rec = (live && (i == 0 || t.rows > 4)) ? W2_RECORDS : 0u;
                const __amdgpu_buffer_rsrc_t y_rs = hyb_rsrc(yp, rec), g_rs = hyb_rsrc(gp, rec);
#pragma unroll
                for (int j = 0; j < 4; ++j) un[i].y[j].u = __builtin_amdgcn_raw_buffer_load_b128(y_rs, yoff[j] + i * y_i, 0, 0);
                un[i].g.u = __builtin_amdgcn_raw_buffer_load_b128(g_rs, goff + i * g_i, 0, 0);
            }
        };
        auto fuse_compute = [&](const W3Tile& t, W3Unit (&un)[2], bf16* db) {
            const long long obase = planar ? ((long long)(t.n * H + t.ty0) * W + t.tx0) * 32 + (long long)(co0 / 32) * fz.dyraw_blk
                                           : ((long long)(t.n * H + t.ty0) * W + t.tx0) * Cop + co0;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const __amdgpu_buffer_rsrc_t o_rs = hyb_rsrc((bf16*)fz.dyraw_out + obase, (writer && (i == 0 || t.rows > 4)) ? W2_RECORDS : 0u);
                union { u32x4 u; bf16x8 v; } o[4];
                if (i == 0 || t.rows > 4) {
                    if (HYB_ABL & 1) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) o[j].u = un[i].y[j].u ^ un[i].g.u;
                    } else
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        float yf[4], v[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) { yf[j] = (float)un[i].y[j].v[e]; v[j] = fmaf(yf[j], sc[e], sh[e]); }
                        // the window maximum and its FIRST position in torch's scan order (0,0),(0,1),(1,0),(1,1); flags live in scalar masks
                        const float vmax = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
                        const bool f0 = v[0] == vmax, f1 = !f0 && v[1] == vmax, f2 = !f0 && !f1 && v[2] == vmax;
                        const bool fl[4] = {f0, f1, f2, !(f0 || f1 || f2)};
                        const float kdy = vmax > 0.f ? kk[e] * (float)un[i].g.v[e] : 0.f;
                        const float a0k = a0[e] + kdy;
#pragma unroll
                        for (int j = 0; j < 4; ++j) o[j].v[e] = (bf16)fmaf(yf[j], a1[e], fl[j] ? a0k : a0[e]);
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) *reinterpret_cast<bf16x8*>(db + lds0 + ((4 * i + (j >> 1)) * W3_TW + (j & 1)) * 64) = o[j].v;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j].u = un[i].y[j].u;      // (dropped stores of a half tile: any defined value)
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (!(HYB_ABL & 8)) __builtin_amdgcn_raw_buffer_store_b128(o[j].u, o_rs, ooff[j] + i * o_i, 0, 0);
            }
        };
        // End of a producer iteration.  Vector-memory operations of the iteration in issue order: [10 prefetch loads of the tile after
        // next] [10 x-halo DMAs] [8 dyraw stores]: the images are complete once the DMAs have landed and the LDS writes are done, so
        // the counted wait leaves the stores in flight across the barrier.  (The builtin, not inline assembly: the compiler's own
        // wait-count pass then knows that the prefetched registers are valid and adds no wait of its own in front of their use.)
        auto publish = [&]() {
            if (HYB_ABL & 16) __builtin_amdgcn_s_waitcnt(0x0070 | 0xc00f);                    // (ablation: lgkmcnt(0) only)
            else if (HYB_ABL & 8) __builtin_amdgcn_s_waitcnt(0x0070);                          // vmcnt(0) lgkmcnt(0)
            else __builtin_amdgcn_s_waitcnt(0x0078);                                           // vmcnt(8) lgkmcnt(0)
            __builtin_amdgcn_s_barrier();
        };
        auto tl = [&](int i) { return w3_tile(tbegin + (i < tcount ? i : tcount - 1), tilesX, tilesY, H); };
        W3Unit ua[2], ub[2];
        {
            const W3Tile t0 = tl(0);
            fuse_load(t0, true, ua);
            x_dma(t0, xbuf);
            fuse_load(tl(1), tcount > 1, ub);
            W2_KEEP_EARLY;
            fuse_compute(t0, ua, dbuf);
            publish();                                    // barrier 1: tile 0 staged
        }
        // iteration i: the consumers contract tile i (buffers i & 1); stage tile i + 1 (registers loaded one iteration earlier) into the
        // other pair and prefetch tile i + 2.  The last tile's iteration has nothing to stage.
        for (int i = 0; i + 1 < tcount; i += 2) {
            {
                const W3Tile t1 = tl(i + 1);
                fuse_load(tl(i + 2), i + 2 < tcount && !(HYB_ABL & 32), ua);
                if (!(HYB_ABL & 16)) x_dma(t1, xbuf + W3_XBUF);
                W2_KEEP_EARLY;
                fuse_compute(t1, ub, dbuf + W3_DBUF);
                publish();
            }
            if (i + 2 >= tcount) break;
            {
                const W3Tile t2 = tl(i + 2);
                fuse_load(tl(i + 3), i + 3 < tcount && !(HYB_ABL & 32), ub);
                if (!(HYB_ABL & 16)) x_dma(t2, xbuf);
                W2_KEEP_EARLY;
                fuse_compute(t2, ua, dbuf);
                publish();
            }
        }
        __builtin_amdgcn_s_barrier();                     // the last tile's barrier
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }

    // =================================================== consumers ===================================================
    const int cot = wave >> 1, cit = wave & 1;                // this wave: output channels 32 cot .., input channels 32 cit ..
    const int g = lane >> 4, h = lane >> 5, qq = (lane & 15) >> 2, pp = lane & 3;
    // fragment addresses (elements): lane = (k half h -> rows 2h, 2h + 1; 16-channel half g & 1; pixel qq of 4; channel quad pp)
    const int aoff = ((2 * h) * W3_TW + qq) * 64 + (((4 * cot + 2 * (g & 1) + (pp >> 1)) ^ (((qq >> 1) & 1) << 2)) << 3) + (pp & 1) * 4;
    int boff[3];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
        boff[kw] = ((2 * h) * W3_HW + qq + kw) * 64 + (((4 * cit + 2 * (g & 1) + (pp >> 1)) ^ ((((qq + kw) >> 1) & 1) << 2)) << 3) + (pp & 1) * 4;

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    __builtin_amdgcn_s_barrier();                      // barrier 1: tile 0 staged
    for (int it = 0; it < tcount; ++it) {
        const bf16* xb = xbuf + (it & 1) * W3_XBUF;
        const bf16* db = dbuf + (it & 1) * W3_DBUF;
        const int rows = w3_tile(tbegin + it, tilesX, tilesY, H).rows;
        // step s = (half, column group j, tap): k-step (half, j) = rows 4 half .. 4 half + 3 x columns 4 j .. 4 j + 3; the x fragment is read
        // three steps ahead (ring of 4), the gradient fragment of the next k-step during taps 2..3
        auto load_a = [&](Frag<bf16>& f, int ks) {
            const int half = ks / 7, j = ks % 7;
            const bf16* p = db + aoff + ((4 * half) * W3_TW + 4 * j) * 64;
            w2_tr(f, p, p + W3_TW * 64);
        };
        auto load_b = [&](Frag<bf16>& f, int s) {
            const int ks = s / 9, tap = s % 9, half = ks / 7, j = ks % 7, kh = tap / 3, kw = tap % 3;
            const bf16* p = xb + boff[kw] + ((4 * half + kh) * W3_HW + 4 * j) * 64;
            w2_tr(f, p, p + W3_HW * 64);
        };
        if (HYB_ABL & 2) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); continue; }
        constexpr int RB = W3_RING;                    // x fragments are read RB - 1 steps ahead
        Frag<bf16> a[2], b[RB];
        load_a(a[0], 0);
#pragma unroll
        for (int s = 0; s < RB - 1; ++s) load_b(b[s], s);
        // (the ring keeps running across the two halves: the reads issued for steps 63.. at the end of the first half are simply not
        // used after a half tile)
#pragma unroll
        for (int s = 0; s < 63; ++s) {
            const int ks = s / 9, tap = s % 9;
            load_b(b[(s + RB - 1) % RB], s + RB - 1);
            if (tap == 2) load_a(a[(ks + 1) & 1], ks + 1);
            acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks & 1].v, b[s % RB].v, acc[tap], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);      // keep the reads where they are: hoisted further ahead they cost accumulator spills
        }
        if (rows > 4) {
#pragma unroll
            for (int s = 63; s < 126; ++s) {
                const int ks = s / 9, tap = s % 9;
                if (s + RB - 1 < 126) load_b(b[(s + RB - 1) % RB], s + RB - 1);
                if (tap == 2 && ks + 1 < 14) load_a(a[(ks + 1) & 1], ks + 1);
                acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks & 1].v, b[s % RB].v, acc[tap], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_sched_barrier(0);      // keep the reads where they are: hoisted further ahead they cost accumulator spills
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                  // the other pair of images is complete, this pair may be overwritten
    }

    // D[row = co][col = ci]: lane holds ci = lane & 31, co rows 8 (r >> 2) + 4 (lane >> 5) + (r & 3)
    float* out = slab + (long long)blockIdx.x * Cop * 9 * Cip;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + 32 * cot + 8 * (r >> 2) + 4 * h + (r & 3);
            out[((long long)co * 9 + tap) * Cip + ci0 + 32 * cit + (lane & 31)] = acc[tap][r];
        }
}

inline int w3_supported(int H, int W, int Cip, int Cop) {
#ifdef HYB_NO_V3
    static const int v3 = 0;
#else
    static const int v3_env = getenv("HYB_WGRAD_V3") ? atoi(getenv("HYB_WGRAD_V3")) : W3_DEFAULT;
#ifdef HYB_WGRAD_EXPERIMENTS
    static const int v3 = v3_env;                       // 2 / 3: the experiment kernels of scripts/micro/wgrad_variants
#else
    static const int v3 = v3_env != 0;                  // the product build knows generations 0 and 1 only: any other value means "third"
#endif
#endif
    return (Cip % 64 == 0 && Cop % 64 == 0 && W % W3_TW == 0 && H % 4 == 0 && H >= 8) ? v3 : 0;
}

inline int w3_launch(dim3 grid, HybProfileHook* hook, hipStream_t st, const bf16* x, float* slab, int N, int H, int W, int Cip, int Cop, int tX, int tY,
                     int nT, const WgradFuse& fz) {
    static HybAttrOnce once;
    if (int e = hyb_set_lds_attr(once, (const void*)wgrad_v3_kernel, (int)W3_LDS)) return e;
    if (hook) hipEventRecord(hook->ev0, st);
    hipLaunchKernelGGL(wgrad_v3_kernel, grid, dim3(512), W3_LDS, st, x, slab, N, H, W, Cip, Cop, tX, tY, nT, fz);
    if (hook) hipEventRecord(hook->ev1, st);
    return 0;
}
