// Fused weight-gradient kernel, pipelined form (bf16, Cip % 64 == 0, Cop % 64 == 0, W % 28 == 0, H % 4 == 0).  Included by
// conv_wgrad.hip after conv_wgrad_v3.h (same 128-byte-pixel tile images and swizzle, same slabs, same arithmetic).
//
// What the SQ counters of the 4 + 4- and 8 + 8-wave kernels showed (profiles/r03_wgrad_pmc.txt):
//   * consumers alone: 4 waves 84 k cycles per workgroup, 8 waves (two per SIMD covering each other's LDS latency) 78 k at 1.9 GHz = 41 us;
//   * producers alone: 113 k / 100 k cycles with 4 / 8 waves -- NOT an issue problem: with two image pairs the x-halo DMA of tile
//     t + 1 is issued at the start of iteration t and must have landed at its end, so an iteration cannot be shorter than one
//     memory round trip under load, however little the producers compute;
//   * both together run at 1.44 GHz instead of 1.9 (the chip lowers its clock with matrix, vector and memory pipes all busy), so
//     every cycle saved on the critical path counts 1.3 times.
// Here the tile is 4 x 28 pixels (halo image 23 KB, gradient image 14 KB), which leaves room for FOUR x-halo images: the DMA of
// tile t + 3 is issued in iteration t and has three iterations to land; the y / dpooled registers are prefetched two tiles ahead
// (three register sets).  4 divides every height taken here, so there are no half tiles.  12 waves: 8 consumers (4 tiles of
// 32 co x 32 ci, taps 0..4 on waves 0-3 and 5..8 on waves 4-7: waves w and w + 4 share a SIMD) + 4 producers (one per SIMD, one
// pooling-window unit per thread and tile, BatchNorm constants in registers), 168 registers per lane.
constexpr int W5_HP = 6 * 30, W5_PX = 4 * 28;
constexpr int W5_XW = (W5_HP * 8 + 63) / 64;          // x-halo DMA wave-instructions per tile (23)
constexpr int W5_XBUF = W5_XW * 512;                  // bf16 elements per x-halo image
constexpr int W5_DBUF = W5_PX * 64;                   // bf16 elements per gradient image
constexpr int W5_XS = 4, W5_DS = 2;                   // images in flight
constexpr size_t W5_LDS = (size_t)(W5_XS * W5_XBUF + W5_DS * W5_DBUF) * 2;

struct W5Tile { int n, ty0, tx0; };
__device__ __forceinline__ W5Tile w5_tile(int tile, int tilesX, int tilesY) {
    W5Tile t;
    t.n = tile / (tilesX * tilesY);
    const int trem = tile - t.n * (tilesX * tilesY);
    const int ty = trem / tilesX;
    t.ty0 = ty * 4;
    t.tx0 = (trem - ty * tilesX) * W3_TW;
    return t;
}

template <int T0, int NT>
__device__ __forceinline__ void w5_consume(const bf16* xbuf, const bf16* dbuf, float* __restrict__ slab, int pair, int lane, int tcount, int Cip, int Cop,
                                           int co0, int ci0) {
    const int cot = pair >> 1, cit = pair & 1;
    const int g = lane >> 4, h = lane >> 5, qq = (lane & 15) >> 2, pp = lane & 3;
    // fragment addresses (elements): lane = (k half h -> rows 2h, 2h + 1; 16-channel half g & 1; pixel qq of 4; channel quad pp)
    const int aoff = ((2 * h) * W3_TW + qq) * 64 + (((4 * cot + 2 * (g & 1) + (pp >> 1)) ^ (((qq >> 1) & 1) << 2)) << 3) + (pp & 1) * 4;
    int boff[3];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
        boff[kw] = ((2 * h) * W3_HW + qq + kw) * 64 + (((4 * cit + 2 * (g & 1) + (pp >> 1)) ^ ((((qq + kw) >> 1) & 1) << 2)) << 3) + (pp & 1) * 4;
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    __builtin_amdgcn_s_barrier();                      // barrier 1: tile 0 staged
    for (int it = 0; it < tcount; ++it) {
        const bf16* xb = xbuf + (it & (W5_XS - 1)) * W5_XBUF;
        const bf16* db = dbuf + (it & (W5_DS - 1)) * W5_DBUF;
        if (HYB_ABL & 2) { __builtin_amdgcn_s_barrier(); continue; }
        // step s = (k-step j = columns 4j .. 4j + 3 of the four rows, tap); x fragments are read RB - 1 steps ahead, the gradient
        // fragment of the next k-step during tap 1
        constexpr int RB = W3_RING, NS = 7 * NT;
        auto load_a = [&](Frag<bf16>& f, int j) {
            const bf16* p = db + aoff + (4 * j) * 64;
            w2_tr(f, p, p + W3_TW * 64);
        };
        auto load_b = [&](Frag<bf16>& f, int s) {
            const int j = s / NT, tap = T0 + s % NT, kh = tap / 3, kw = tap % 3;
            const bf16* p = xb + boff[kw] + (kh * W3_HW + 4 * j) * 64;
            w2_tr(f, p, p + W3_HW * 64);
        };
        Frag<bf16> a[2], b[RB];
        load_a(a[0], 0);
#pragma unroll
        for (int s = 0; s < RB - 1; ++s) load_b(b[s], s);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int j = s / NT, tap = s % NT;
            if (s + RB - 1 < NS) load_b(b[(s + RB - 1) % RB], s + RB - 1);
            if (tap == 1 && j + 1 < 7) load_a(a[(j + 1) & 1], j + 1);
            acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[j & 1].v, b[s % RB].v, acc[tap], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);         // keep the reads where they are: hoisted further ahead they cost accumulator spills
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                  // this tile's images may be overwritten
    }
    // D[row = co][col = ci]: lane holds ci = lane & 31, co rows 8 (r >> 2) + 4 (lane >> 5) + (r & 3)
    float* out = slab + (long long)blockIdx.x * Cop * 9 * Cip;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + 32 * cot + 8 * (r >> 2) + 4 * h + (r & 3);
            out[((long long)co * 9 + T0 + t) * Cip + ci0 + 32 * cit + (lane & 31)] = acc[t][r];
        }
}

__global__ __launch_bounds__(768) void wgrad_v5_kernel(const bf16* __restrict__ x, float* __restrict__ slab, int N, int H, int W, int Cip, int Cop,
                                                       int tilesX, int tilesY, int numTiles, WgradFuse fz) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16* const xbuf = reinterpret_cast<bf16*>(smem_raw);                 // [W5_XS][W5_XBUF]
    bf16* const dbuf = xbuf + W5_XS * W5_XBUF;                            // [W5_DS][W5_DBUF]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nCiBlk = Cip >> 6;
    const int co0 = (blockIdx.y / nCiBlk) * 64, ci0 = (blockIdx.y % nCiBlk) * 64;
    const int tchunk = (numTiles + (int)gridDim.x - 1) / (int)gridDim.x;   // contiguous run of tiles (the host leaves no run empty)
    const int tbegin = blockIdx.x * tchunk;
    const int tcount = (tbegin + tchunk < numTiles ? tbegin + tchunk : numTiles) - tbegin;

    if (wave < 4) { w5_consume<0, 5>(xbuf, dbuf, slab, wave & 3, lane, tcount, Cip, Cop, co0, ci0); return; }
    if (wave < 8) { w5_consume<5, 4>(xbuf, dbuf, slab, wave & 3, lane, tcount, Cip, Cop, co0, ci0); return; }

    // ================================================= producers =================================================
    const int pw = wave - 8, ptid = tid - 512;
    const int oct = ptid & 7, wslot = ptid >> 3;          // 28 windows per tile; slots 28..31 only issue their share of the DMAs
    const bool idle = wslot >= 28;
    const int w0 = idle ? 27 : wslot;
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
    // x-halo DMA pieces of this wave: k * 4 + pw, k = 0..5 (pieces past 22 repeat piece 22)
    unsigned xoff[6];
    int xyx[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        int wi = k * 4 + pw;
        if (wi > W5_XW - 1) wi = W5_XW - 1;
        const int u = wi * 64 + lane, hp = u >> 3, cp = u & 7;
        const int hy = hp / W3_HW, hx = hp - hy * W3_HW;
        xoff[k] = (unsigned)(((hy * W + hx) * Cip + ((cp ^ (((hx >> 1) & 1) << 2)) << 3)) * 2);
        xyx[k] = hp < W5_HP ? ((hy << 16) | hx) : (0x7fff << 16);
    }
    // Every iteration issues the SAME vector-memory operations in the same order -- 6 DMAs, 5 prefetch loads, 4 stores -- so that the
    // counted waits below and the compiler's own bookkeeping are exact; what the end of the run does not need goes through a
    // descriptor of zero records (loads return zeros, stores are dropped, no memory traffic).
    auto x_dma = [&](const W5Tile& t, bool live, bf16* xb) {
        const long long base = ((long long)(t.n * H + t.ty0 - 1) * W + (t.tx0 - 1)) * Cip + ci0;
        const __amdgpu_buffer_rsrc_t rs = hyb_rsrc(x + base, (live && !(HYB_ABL & 16)) ? W2_RECORDS : 0u);
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const int gy = t.ty0 - 1 + (xyx[k] >> 16), gx = t.tx0 - 1 + (xyx[k] & 0xffff);
            const bool valid = ((unsigned)gy < (unsigned)H) && ((unsigned)gx < (unsigned)W);
            int wi = k * 4 + pw;
            if (wi > W5_XW - 1) wi = W5_XW - 1;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)(xb + wi * 512), 16, valid ? xoff[k] : W2_OOB, 0, 0, 0);
        }
    };
    unsigned yoff[4], ooff[4];
    const bool planar = fz.dyraw_blk != 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int pix = (2 * wy + (j >> 1)) * W + 2 * wx + (j & 1);
        yoff[j] = idle ? W2_OOB : (unsigned)((pix * Cop + 8 * oct) * 2);
        ooff[j] = idle ? W2_OOB : (planar ? (unsigned)((pix * 32 + 8 * (oct & 3)) * 2 + (oct >> 2) * fz.dyraw_blk * 2) : yoff[j]);
    }
    const unsigned goff = idle ? W2_OOB : (unsigned)(((wy * (W >> 1) + wx) * Cop + 8 * oct) * 2);
    const int lds0 = ((2 * wy) * W3_TW + 2 * wx) * 64 + ((oct ^ ((wx & 1) << 2)) << 3);      // element offset of the window's pixel j = 0
    const bool writer = fz.dyraw_out && ci0 == 0;

    auto fuse_load = [&](const W5Tile& t, bool live, W3Unit& un) {
        const int Ho = H >> 1, Wo = W >> 1;
        const bf16* yp = (const bf16*)fz.y + ((long long)(t.n * H + t.ty0) * W + t.tx0) * Cop + co0;
        const bf16* gp = (const bf16*)fz.dp + ((long long)(t.n * Ho + (t.ty0 >> 1)) * Wo + (t.tx0 >> 1)) * Cop + co0;
        const unsigned rec = (live && !(HYB_ABL & 32)) ? W2_RECORDS : 0u;
        const __amdgpu_buffer_rsrc_t y_rs = hyb_rsrc(yp, rec), g_rs = hyb_rsrc(gp, rec);
#pragma unroll
        for (int j = 0; j < 4; ++j) un.y[j].u = __builtin_amdgcn_raw_buffer_load_b128(y_rs, yoff[j], 0, 0);
        un.g.u = __builtin_amdgcn_raw_buffer_load_b128(g_rs, goff, 0, 0);
    };
    auto fuse_compute = [&](const W5Tile& t, W3Unit& un, bf16* db) {
        const long long obase = planar ? ((long long)(t.n * H + t.ty0) * W + t.tx0) * 32 + (long long)(co0 / 32) * fz.dyraw_blk
                                       : ((long long)(t.n * H + t.ty0) * W + t.tx0) * Cop + co0;
        const __amdgpu_buffer_rsrc_t o_rs = hyb_rsrc((bf16*)fz.dyraw_out + obase, (writer && !(HYB_ABL & 8)) ? W2_RECORDS : 0u);
        union { u32x4 u; bf16x8 v; } o[4];
        if (HYB_ABL & 1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j].u = un.y[j].u ^ un.g.u;
        } else
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float yf[4], v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) { yf[j] = (float)un.y[j].v[e]; v[j] = fmaf(yf[j], sc[e], sh[e]); }
            // the window maximum and its FIRST position in torch's scan order (0,0),(0,1),(1,0),(1,1); flags live in scalar masks
            const float vmax = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
            const bool f0 = v[0] == vmax, f1 = !f0 && v[1] == vmax, f2 = !f0 && !f1 && v[2] == vmax;
            const bool fl[4] = {f0, f1, f2, !(f0 || f1 || f2)};
            const float kdy = vmax > 0.f ? kk[e] * (float)un.g.v[e] : 0.f;
            const float a0k = a0[e] + kdy;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j].v[e] = (bf16)fmaf(yf[j], a1[e], fl[j] ? a0k : a0[e]);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (!idle) *reinterpret_cast<bf16x8*>(db + lds0 + ((j >> 1) * W3_TW + (j & 1)) * 64) = o[j].v;
            __builtin_amdgcn_raw_buffer_store_b128(o[j].u, o_rs, ooff[j], 0, 0);
        }
    };
    // End of iteration k: the gradient image of tile k + 1 is complete once this thread's LDS writes are done; its x-halo image was
    // requested three iterations earlier, as the first operations of that iteration: 9 + 15 + 15 = 39 younger vector-memory
    // operations may still be in flight.
    auto publish = [&]() {
        __builtin_amdgcn_s_waitcnt(0x0070 | (39 & 15) | ((39 >> 4) << 14));      // vmcnt(39) lgkmcnt(0)
        __builtin_amdgcn_s_barrier();
    };
    auto tl = [&](int i) { return w5_tile(tbegin + (i < tcount ? i : tcount - 1), tilesX, tilesY); };
    auto xs = [&](int i) { return xbuf + (i & (W5_XS - 1)) * W5_XBUF; };
    auto ds = [&](int i) { return dbuf + (i & (W5_DS - 1)) * W5_DBUF; };
    W3Unit u0, u1, u2;                                    // tile i's y / dpooled registers live in set i % 3
    {
        x_dma(tl(0), true, xs(0));
        x_dma(tl(1), tcount > 1, xs(1));
        x_dma(tl(2), tcount > 2, xs(2));
        fuse_load(tl(0), true, u0);
        fuse_load(tl(1), tcount > 1, u1);
        fuse_load(tl(2), tcount > 2, u2);
        W2_KEEP_EARLY;
        fuse_compute(tl(0), u0, ds(0));
        __builtin_amdgcn_s_waitcnt(0x0070 | (31 & 15) | ((31 >> 4) << 14));      // vmcnt(31): tile 0's DMAs (12 DMAs, 15 loads, 4 stores younger)
        __builtin_amdgcn_s_barrier();                     // barrier 1: tile 0 staged
    }
    // iteration k: the consumers contract tile k; request the x halo of tile k + 3 and the registers of tile k + 3, stage tile k + 1
    auto iter = [&](int k, W3Unit& cur, W3Unit& nxt) {
        x_dma(tl(k + 3), k + 3 < tcount, xs(k + 3));
        fuse_load(tl(k + 3), k + 3 < tcount, nxt);
        W2_KEEP_EARLY;
        fuse_compute(tl(k + 1), cur, ds(k + 1));
        publish();
    };
    for (int k = 0; k + 1 < tcount; k += 3) {
        iter(k, u1, u0);
        if (k + 2 >= tcount) break;
        iter(k + 1, u2, u1);
        if (k + 3 >= tcount) break;
        iter(k + 2, u0, u2);
    }
    __builtin_amdgcn_s_barrier();                         // the last tile's barrier
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

inline int w5_launch(dim3 grid, HybProfileHook* hook, hipStream_t st, const bf16* x, float* slab, int N, int H, int W, int Cip, int Cop, int nT,
                     const WgradFuse& fz) {
    static HybAttrOnce once;
    if (int e = hyb_set_lds_attr(once, (const void*)wgrad_v5_kernel, (int)W5_LDS)) return e;
    if (hook) hipEventRecord(hook->ev0, st);
    hipLaunchKernelGGL(wgrad_v5_kernel, grid, dim3(768), W5_LDS, st, x, slab, N, H, W, Cip, Cop, W / W3_TW, H / 4, nT, fz);
    if (hook) hipEventRecord(hook->ev1, st);
    return 0;
}
