// Fused weight-gradient kernel, 16-wave form of conv_wgrad_v3.h (same shapes, same tile images and swizzle, same slabs).
//
// Why: rocprofv3's SQ counters of the 4 + 4-wave kernel (profiles/r03_wgrad_pmc.txt) showed that in CYCLES its two halves do overlap --
// producers alone 113 k cycles per workgroup, consumers alone 84 k, together 127 k -- and that the producers are the critical path: a
// single producer wave per SIMD issues its ~600 vector + ~300 scalar instructions per tile strictly in order, 50 % of its time issuing,
// 25 % stalled on its own dependencies (compare -> scalar mask -> select chains), 25 % waiting for memory.  (The wall-clock sum came from
// the clock: 2.0 GHz with the matrix cores idle, 1.5 GHz with them busy.)  Two producer waves per SIMD hide each other's stalls.
// Sixteen waves leave 128 registers per lane, so the consumers are split as well: 8 waves = 4 (32 co x 32 ci) tiles x 2 tap groups
// (taps 0..4 on waves 0-3, taps 5..8 on waves 4-7; waves w and w + 4 share a SIMD, which therefore still issues 9 MFMAs per k-step).
// The producers read their per-channel constants from LDS again (no registers to keep 40 of them).
constexpr size_t W4_LDS = W3_LDS + 32 * 12 * sizeof(float);     // + 32 channel-pair records of the BatchNorm constants (48 bytes each)

struct W4Unit { union { u32x4 u; bf16x8 v; } y[4], g; };

template <int T0, int NT>
__device__ __forceinline__ void w4_consume(const bf16* xbuf, const bf16* dbuf, float* __restrict__ slab, int pair, int lane, int tbegin, int tcount,
                                           int tilesX, int tilesY, int H, int Cip, int Cop, int co0, int ci0) {
    const int cot = pair >> 1, cit = pair & 1;
    const int g = lane >> 4, h = lane >> 5, qq = (lane & 15) >> 2, pp = lane & 3;
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
        const bf16* xb = xbuf + (it & 1) * W3_XBUF;
        const bf16* db = dbuf + (it & 1) * W3_DBUF;
        const int rows = w3_tile(tbegin + it, tilesX, tilesY, H).rows;
        // step s = (k-step, tap); x fragments read RB - 1 steps ahead, the gradient fragment of the next k-step during tap 1
        constexpr int RB = W3_RING;
        auto load_a = [&](Frag<bf16>& f, int ks) {
            const int half = ks / 7, j = ks % 7;
            const bf16* p = db + aoff + ((4 * half) * W3_TW + 4 * j) * 64;
            w2_tr(f, p, p + W3_TW * 64);
        };
        auto load_b = [&](Frag<bf16>& f, int s) {
            const int ks = s / NT, tap = T0 + s % NT, half = ks / 7, j = ks % 7, kh = tap / 3, kw = tap % 3;
            const bf16* p = xb + boff[kw] + ((4 * half + kh) * W3_HW + 4 * j) * 64;
            w2_tr(f, p, p + W3_HW * 64);
        };
        if (HYB_ABL & 2) { __builtin_amdgcn_s_barrier(); continue; }
        Frag<bf16> a[2], b[RB];
        load_a(a[0], 0);
#pragma unroll
        for (int s = 0; s < RB - 1; ++s) load_b(b[s], s);
#pragma unroll
        for (int s = 0; s < 7 * NT; ++s) {
            const int ks = s / NT, tap = s % NT;
            load_b(b[(s + RB - 1) % RB], s + RB - 1);
            if (tap == 1) load_a(a[(ks + 1) & 1], ks + 1);
            acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks & 1].v, b[s % RB].v, acc[tap], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (rows > 4) {
#pragma unroll
            for (int s = 7 * NT; s < 14 * NT; ++s) {
                const int ks = s / NT, tap = s % NT;
                if (s + RB - 1 < 14 * NT) load_b(b[(s + RB - 1) % RB], s + RB - 1);
                if (tap == 1 && ks + 1 < 14) load_a(a[(ks + 1) & 1], ks + 1);
                acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks & 1].v, b[s % RB].v, acc[tap], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                  // the other pair of images is complete, this pair may be overwritten
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

__global__ __launch_bounds__(1024) void wgrad_v4_kernel(const bf16* __restrict__ x, float* __restrict__ slab, int N, int H, int W, int Cip, int Cop,
                                                        int tilesX, int tilesY, int numTiles, WgradFuse fz) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16* const xbuf = reinterpret_cast<bf16*>(smem_raw);                 // [2][W3_XBUF]
    bf16* const dbuf = xbuf + 2 * W3_XBUF;                                // [2][W3_DBUF]
    float* const cst = reinterpret_cast<float*>(dbuf + 2 * W3_DBUF);      // [32][12]: per channel pair sc, sh, k, A1, A0

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nCiBlk = Cip >> 6;
    const int co0 = (blockIdx.y / nCiBlk) * 64, ci0 = (blockIdx.y % nCiBlk) * 64;
    const int tchunk = (numTiles + (int)gridDim.x - 1) / (int)gridDim.x;
    const int tbegin = blockIdx.x * tchunk;
    const int tcount = (tbegin + tchunk < numTiles ? tbegin + tchunk : numTiles) - tbegin;

    if (tid < 64) {
        const int ch = co0 + tid;
        const float sc = fz.ss[ch], sh = fz.ss[Cop + ch];
        const float mean = fz.mi[ch], inv = fz.mi[Cop + ch];
        const float k = (ch < fz.Co ? fz.gamma[ch] : 0.f) * inv;
        const float m1 = fz.training ? fz.sums[ch] * fz.inv_count : 0.f, m2 = fz.training ? fz.sums[Cop + ch] * fz.inv_count : 0.f;
        // two-channel records: pair p = ch / 2 holds {sc0, sc1, sh0, sh1, k0, k1, a10, a11, a00, a01, -, -} (48 bytes: 16-byte aligned reads)
        float* r = cst + (tid >> 1) * 12 + (tid & 1);
        r[0] = sc; r[2] = sh; r[4] = k; r[6] = -k * m2 * inv; r[8] = -k * m1 + k * m2 * inv * mean;
    }
    __syncthreads();                                   // barrier 0: constants visible

    if (wave >= 8) {
        // ================================================= producers (8 waves, one unit per thread and tile) =================================================
        const int pw = wave - 8, ptid = tid - 512;
        const int oct = ptid & 7, wslot = ptid >> 3;      // 56 windows per tile; slots 56..63 (the last producer wave) only issue their DMA pieces
        const bool dup = wslot >= 56;
        const int w0 = dup ? 55 : wslot;
        const int wy = w0 / 14, wx = w0 - wy * 14;        // window row 0..3: rows 2 wy, 2 wy + 1 of the tile
        const bool lower = wy >= 2;                       // second half of the tile: nothing to do for a half tile
        // x-halo DMA pieces of this wave: k * 8 + pw, k = 0..4 (pieces past 37 repeat piece 37)
        unsigned xoff[5];
        int xyx[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            int wi = k * 8 + pw;
            if (wi > W3_XW - 1) wi = W3_XW - 1;
            const int u = wi * 64 + lane, hp = u >> 3, cp = u & 7;
            const int hy = hp / W3_HW, hx = hp - hy * W3_HW;
            xoff[k] = (unsigned)(((hy * W + hx) * Cip + ((cp ^ (((hx >> 1) & 1) << 2)) << 3)) * 2);
            xyx[k] = hp < W3_HP ? ((hy << 16) | hx) : (0x7fff << 16);
        }
        auto x_dma = [&](const W3Tile& t, bf16* xb) {
            const long long base = ((long long)(t.n * H + t.ty0 - 1) * W + (t.tx0 - 1)) * Cip + ci0;
            const __amdgpu_buffer_rsrc_t rs = hyb_rsrc(x + base, (HYB_ABL & 16) ? 0u : W2_RECORDS);
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                const int gy = t.ty0 - 1 + (xyx[k] >> 16), gx = t.tx0 - 1 + (xyx[k] & 0xffff);
                const bool valid = ((unsigned)gy < (unsigned)H) && ((unsigned)gx < (unsigned)W);
                int wi = k * 8 + pw;
                if (wi > W3_XW - 1) wi = W3_XW - 1;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)(xb + wi * 512), 16, valid ? xoff[k] : W2_OOB, 0, 0, 0);
            }
        };
        unsigned yoff[4], ooff[4];
        const bool planar = fz.dyraw_blk != 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int pix = (2 * wy + (j >> 1)) * W + 2 * wx + (j & 1);
            yoff[j] = (unsigned)((pix * Cop + 8 * oct) * 2);
            ooff[j] = dup ? W2_OOB : (planar ? (unsigned)((pix * 32 + 8 * (oct & 3)) * 2 + (oct >> 2) * fz.dyraw_blk * 2) : yoff[j]);
        }
        const unsigned goff = (unsigned)(((wy * (W >> 1) + wx) * Cop + 8 * oct) * 2);
        const int lds0 = ((2 * wy) * W3_TW + 2 * wx) * 64 + ((oct ^ ((wx & 1) << 2)) << 3);
        const bool writer = fz.dyraw_out && ci0 == 0;
        const float* const crec = cst + oct * 48;         // this octet's four channel-pair records

        // the same number of vector-memory operations in every iteration (5 loads, 5 DMAs, 4 stores); what is not needed goes through a
        // descriptor of zero records (see conv_wgrad_v3.h)
        auto fuse_load = [&](const W3Tile& t, bool live, W4Unit& un) {
            const int Ho = H >> 1, Wo = W >> 1;
            const bf16* yp = (const bf16*)fz.y + ((long long)(t.n * H + t.ty0) * W + t.tx0) * Cop + co0;
            const bf16* gp = (const bf16*)fz.dp + ((long long)(t.n * Ho + (t.ty0 >> 1)) * Wo + (t.tx0 >> 1)) * Cop + co0;
            const unsigned rec = (live && !(HYB_ABL & 32)) ? W2_RECORDS : 0u;
            const __amdgpu_buffer_rsrc_t y_rs = hyb_rsrc(yp, rec), g_rs = hyb_rsrc(gp, rec);
            const bool mine = !dup && !(lower && t.rows <= 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) un.y[j].u = __builtin_amdgcn_raw_buffer_load_b128(y_rs, mine ? yoff[j] : W2_OOB, 0, 0);
            un.g.u = __builtin_amdgcn_raw_buffer_load_b128(g_rs, mine ? goff : W2_OOB, 0, 0);
        };
        auto fuse_compute = [&](const W3Tile& t, W4Unit& un, bf16* db) {
            const long long obase = planar ? ((long long)(t.n * H + t.ty0) * W + t.tx0) * 32 + (long long)(co0 / 32) * fz.dyraw_blk
                                           : ((long long)(t.n * H + t.ty0) * W + t.tx0) * Cop + co0;
            const __amdgpu_buffer_rsrc_t o_rs = hyb_rsrc((bf16*)fz.dyraw_out + obase, (writer && !(HYB_ABL & 8)) ? W2_RECORDS : 0u);
            const bool mine = !(HYB_ABL & 1) && !dup && !(lower && t.rows <= 4);
            union { u32x4 u; bf16x8 v; } o[4];
            if (__builtin_amdgcn_readfirstlane(__builtin_amdgcn_ballot_w64(mine) != 0)) {      // (waves entirely in the missing half of a half tile skip)
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const f32x4 c0 = *reinterpret_cast<const f32x4*>(crec + p * 12);           // sc0 sc1 sh0 sh1
                    const f32x4 c1 = *reinterpret_cast<const f32x4*>(crec + p * 12 + 4);       // k0 k1 a10 a11
                    const f32x2 c2 = *reinterpret_cast<const f32x2*>(crec + p * 12 + 8);       // a00 a01
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const int e = 2 * p + q;
                        const float sc = c0[q], sh = c0[2 + q], kk = c1[q], a1 = c1[2 + q], a0 = c2[q];
                        float yf[4], v[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) { yf[j] = (float)un.y[j].v[e]; v[j] = fmaf(yf[j], sc, sh); }
                        const float vmax = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
                        const bool f0 = v[0] == vmax, f1 = !f0 && v[1] == vmax, f2 = !f0 && !f1 && v[2] == vmax;
                        const bool fl[4] = {f0, f1, f2, !(f0 || f1 || f2)};
                        const float kdy = vmax > 0.f ? kk * (float)un.g.v[e] : 0.f;
                        const float a0k = a0 + kdy;
#pragma unroll
                        for (int j = 0; j < 4; ++j) o[j].v[e] = (bf16)fmaf(yf[j], a1, fl[j] ? a0k : a0);
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) *reinterpret_cast<bf16x8*>(db + lds0 + ((j >> 1) * W3_TW + (j & 1)) * 64) = o[j].v;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j].u = un.y[j].u;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) __builtin_amdgcn_raw_buffer_store_b128(o[j].u, o_rs, mine ? ooff[j] : W2_OOB, 0, 0);
        };
        auto publish = [&]() {
            __builtin_amdgcn_s_waitcnt(0x0074);                                                // vmcnt(4) lgkmcnt(0): the DMAs have landed, stores in flight
            __builtin_amdgcn_s_barrier();
        };
        auto tl = [&](int i) { return w3_tile(tbegin + (i < tcount ? i : tcount - 1), tilesX, tilesY, H); };
        W4Unit ua, ub;
        {
            const W3Tile t0 = tl(0);
            fuse_load(t0, true, ua);
            x_dma(t0, xbuf);
            fuse_load(tl(1), tcount > 1, ub);
            W2_KEEP_EARLY;
            fuse_compute(t0, ua, dbuf);
            publish();                                    // barrier 1: tile 0 staged
        }
        for (int i = 0; i + 1 < tcount; i += 2) {
            {
                const W3Tile t1 = tl(i + 1);
                fuse_load(tl(i + 2), i + 2 < tcount, ua);
                x_dma(t1, xbuf + W3_XBUF);
                W2_KEEP_EARLY;
                fuse_compute(t1, ub, dbuf + W3_DBUF);
                publish();
            }
            if (i + 2 >= tcount) break;
            {
                const W3Tile t2 = tl(i + 2);
                fuse_load(tl(i + 3), i + 3 < tcount, ub);
                x_dma(t2, xbuf);
                W2_KEEP_EARLY;
                fuse_compute(t2, ua, dbuf);
                publish();
            }
        }
        __builtin_amdgcn_s_barrier();                     // the last tile's barrier
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }

    // =================================================== consumers (8 waves) ===================================================
    // waves 0-3 contract taps 0..4, waves 4-7 taps 5..8 of their 32 co x 32 ci tile: two instantiations, each with its own tile loop
    // and accumulators (one loop with a branch inside makes the compiler copy and spill the accumulators around the branch)
    if (wave < 4) w4_consume<0, 5>(xbuf, dbuf, slab, wave & 3, lane, tbegin, tcount, tilesX, tilesY, H, Cip, Cop, co0, ci0);
    else w4_consume<5, 4>(xbuf, dbuf, slab, wave & 3, lane, tbegin, tcount, tilesX, tilesY, H, Cip, Cop, co0, ci0);
}

inline int w4_launch(dim3 grid, HybProfileHook* hook, hipStream_t st, const bf16* x, float* slab, int N, int H, int W, int Cip, int Cop, int tX, int tY,
                     int nT, const WgradFuse& fz) {
    static HybAttrOnce once;
    if (int e = hyb_set_lds_attr(once, (const void*)wgrad_v4_kernel, (int)W4_LDS)) return e;
    if (hook) hipEventRecord(hook->ev0, st);
    hipLaunchKernelGGL(wgrad_v4_kernel, grid, dim3(1024), W4_LDS, st, x, slab, N, H, W, Cip, Cop, tX, tY, nT, fz);
    if (hook) hipEventRecord(hook->ev1, st);
    return 0;
}
