// Weight gradient of the pixel-side products through LDS (included by fct_bwd.hip inside its anonymous namespace, after ConvGeo and
// direct_wgrad_kernel):   part[slice][n][k] = sum_{pix in slice} dy[pix][n] * x[pix][k],   cpart[slice][n] = sum_pix dy[pix][n].
//
// direct_wgrad_kernel reads both operands of every 16x16x4 MFMA straight from global memory, one dword per lane: ten 256-byte loads per
// sixteen MFMAs and 0.078 B/FLOP of L2 traffic -- 62 TF/s on the Encoder_32K shapes.  Here (the structure of gemm_nt_lds_kernel, with the
// contraction running over pixel ROWS):
//   * a workgroup = 4 waves (WN x WK) owns a (WN*64) x (WK*NKT*16) block of dW for one pixel slice; a wave owns 64 x NKT*16 of it
//     (NKT = 3 makes the k tile 192 = a third of 9 x 64: the 3x3 layers tile without remainder);
//   * 32 pixel rows of dy[.][n tile] and x[.][k tile] per stage go global -> LDS by 16-byte LDS-DMA, two stages; rows beyond the slice,
//     columns beyond Nn / K and -- IMPL -- taps outside the image are out-of-range lanes (zeros);
//   * both tiles keep the global [pixel][feature] layout, so a lane's operand for the 4-deep MFMA of four pixels is ONE ds_read of
//     4 (NKT) consecutive features: lane (m, kq) holds features 4m .. 4m+3 of pixel kq = one element of four different 16x16 tiles
//     (tile t <- features {4m + t}); the permutation is undone when the slab is written;
//   * the bias gradient is one more MFMA per n tile against a constant ones operand (workgroups of k tile 0).
// IMPL: x is the NHWC image of the convolution `g` (Ci % 4 == 0; a 16-byte chunk = 4 channels of one tap), as in direct_wgrad_kernel.
#pragma once

typedef __attribute__((address_space(3))) void wl_lds_void_t;
constexpr unsigned WL_OOB = 0xfffffff0u, WL_RECORDS = 0x80000000u;
constexpr int WL_PB = 32;                                     // pixel rows per stage
__device__ __forceinline__ void wl_dma16(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, float* lds_dst_wave_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (wl_lds_void_t*)lds_dst_wave_base, 16, voff, 0, 0, 0);
}
__device__ __forceinline__ void wl_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

template <int WN, int WK, int NKT> struct WlGeom {
    static constexpr int BN = WN * 64, BK = WK * NKT * 16;
    static constexpr int A_CH = WL_PB * BN / 4, B_CH = WL_PB * BK / 4;          // 16-byte chunks per stage
    static constexpr int A_I = A_CH / 256, B_I = B_CH / 256;                    // DMA instructions per wave and stage
    static constexpr int STAGE = WL_PB * (BN + BK);                             // floats
    static constexpr size_t LDS_BYTES = (size_t)2 * STAGE * 4;
    static_assert(WN * WK == 4 && A_CH % 256 == 0 && B_CH % 256 == 0, "four waves, whole instructions");
};

template <int NKT> struct WlVec;
template <> struct WlVec<2> { typedef float __attribute__((ext_vector_type(2))) type; };
template <> struct WlVec<3> { typedef float __attribute__((ext_vector_type(3))) type; };
template <> struct WlVec<4> { typedef float __attribute__((ext_vector_type(4))) type; };

template <int WN, int WK, int NKT, bool IMPL>
__global__ __launch_bounds__(256, 2) void wgrad_lds_kernel(const float* __restrict__ dy, int lddy, const float* __restrict__ x, int ldx,
                                                           float* __restrict__ part, float* __restrict__ cpart, long long P, int Nn, int K,
                                                           int kgroups, int slice_rows, ConvGeo g, int tiles, int S_) {
    using G = WlGeom<WN, WK, NKT>;
    constexpr int BN = G::BN, BK = G::BK, A_I = G::A_I, B_I = G::B_I, STAGE = G::STAGE;
    extern __shared__ __attribute__((aligned(16))) float wl_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 15, kq = lane >> 4;
    // flat grid, XCD-aware: workgroups v, v + 8, v + 16, .. (one XCD, one L2) take ALL (n, k) tiles of one pixel slice before the next -- the
    // k tiles of a slice are the k*k shifted views of the same pixels (and every n tile re-reads x): dealt round-robin they were fetched once
    // per XCD (3.5 x the algorithmic bytes on the 512 -> 128 layer, profiles/r04_pmc_fetch_write_enc32k.csv)
    const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
    const int slice = (seq / tiles) * 8 + xcd, tl = seq % tiles;
    if (slice >= S_) return;
    const int kg = tl % kgroups, ng = tl / kgroups;
    const int n0 = ng * BN, k0 = kg * BK;
    const long long r_begin = (long long)slice * slice_rows;
    const int rows = (int)((r_begin + slice_rows < P ? r_begin + slice_rows : P) - r_begin);
    const bool want_cs = cpart != nullptr && kg == 0;

    // ---- DMA roles.  dy tile: chunk u = (i*4 + wave)*64 + lane -> pixel row u / (BN/4), chunk column u % (BN/4)
    const __amdgpu_buffer_rsrc_t ars = hyb_rsrc(dy + r_begin * lddy + n0, WL_RECORDS);
    unsigned aoff[A_I];
    int apix[A_I];
#pragma unroll
    for (int i = 0; i < A_I; ++i) {
        const int u = (i * 4 + wave) * 64 + lane, pr = u / (BN / 4), c = u % (BN / 4);
        apix[i] = (n0 + 4 * c + 4 <= Nn) ? pr : 0x40000000;                     // columns beyond Nn: never inside the slice
        aoff[i] = (unsigned)(pr * lddy + 4 * c) << 2;
    }
    // x tile: the same with BK; IMPL: the chunk's tap and channel are fixed per lane and instruction, the pixel coordinates advance by 32
    long long xbase_el;
    int nimg0 = 0;
    if (IMPL) { nimg0 = (int)(r_begin / ((long long)g.Ho * g.Wo)); xbase_el = (long long)nimg0 * g.H * g.W * g.Ci; }
    else xbase_el = r_begin * ldx + k0;
    const __amdgpu_buffer_rsrc_t brs = hyb_rsrc(x + xbase_el, WL_RECORDS);
    unsigned boff[B_I];
    int bpix[B_I];
    int tdy[B_I], tdx[B_I], wo[B_I], ho[B_I], ni[B_I];
#pragma unroll
    for (int i = 0; i < B_I; ++i) {
        const int u = (i * 4 + wave) * 64 + lane, pr = u / (BK / 4), c = u % (BK / 4);
        const int kc = k0 + 4 * c;
        bool ok = kc + 4 <= K;
        if (IMPL) {
            const int tap = kc / g.Ci, ci = kc - tap * g.Ci;
            const int ty = tap / g.k, tx = tap - ty * g.k;
            ok = ok && tap < g.k * g.k;
            tdy[i] = ty * g.dil - g.pad; tdx[i] = tx * g.dil - g.pad;
            boff[i] = (unsigned)((tdy[i] * g.W + tdx[i]) * g.Ci + ci) << 2;     // from the receptive field's centre pixel (may wrap: added mod 2^32)
            const long long pix = r_begin + pr;
            wo[i] = (int)(pix % g.Wo);
            const long long t = pix / g.Wo;
            ho[i] = (int)(t % g.Ho); ni[i] = (int)(t / g.Ho) - nimg0;
        } else {
            tdy[i] = tdx[i] = wo[i] = ho[i] = ni[i] = 0;
            boff[i] = (unsigned)(pr * ldx + 4 * c) << 2;
        }
        bpix[i] = ok ? pr : 0x40000000;
    }
    auto issue = [&](int buf, int p0) {                                        // stage of pixel rows p0 .. p0+31 of the slice
        float* const st = wl_smem + buf * STAGE;
#pragma unroll
        for (int i = 0; i < A_I; ++i)
            wl_dma16(ars, (p0 + apix[i] < rows) ? aoff[i] + (unsigned)(p0 * lddy) * 4u : WL_OOB, st + (i * 4 + wave) * 256);
#pragma unroll
        for (int i = 0; i < B_I; ++i) {
            if (IMPL) {
                const int iy = ho[i] * g.stride, ix = wo[i] * g.stride;
                const bool ok = (p0 + bpix[i] < rows) & ((unsigned)(iy + tdy[i]) < (unsigned)g.H) & ((unsigned)(ix + tdx[i]) < (unsigned)g.W);
                const unsigned centre = (unsigned)(((ni[i] * g.H + iy) * g.W + ix) * g.Ci) << 2;
                wl_dma16(brs, ok ? centre + boff[i] : WL_OOB, st + WL_PB * BN + (i * 4 + wave) * 256);
                wo[i] += WL_PB;
                while (wo[i] >= g.Wo) { wo[i] -= g.Wo; if (++ho[i] == g.Ho) { ho[i] = 0; ++ni[i]; } }
            } else {
                wl_dma16(brs, (p0 + bpix[i] < rows) ? boff[i] + (unsigned)(p0 * ldx) * 4u : WL_OOB, st + WL_PB * BN + (i * 4 + wave) * 256);
            }
        }
    };

    const int wn = wave / WK, wk = wave % WK;
    const int fa = kq * BN + wn * 64 + 4 * m;                                  // + group * 4 * BN
    const int fb = WL_PB * BN + kq * BK + wk * (NKT * 16) + NKT * m;           // + group * 4 * BK
    f32x4 acc[4][NKT], accs[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        accs[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < NKT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int nst = (rows + WL_PB - 1) / WL_PB;
    issue(0, 0);
    for (int it = 0; it < nst; ++it) {
        wl_wait_all();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        issue((it + 1) & 1, (it + 1) * WL_PB);                                 // (beyond the slice: every lane out of range)
        const float* st = wl_smem + (it & 1) * STAGE;
#pragma unroll
        for (int grp = 0; grp < WL_PB / 4; ++grp) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(st + fa + grp * 4 * BN);
            typename WlVec<NKT>::type b;
            if (NKT == 3) { const float* bp = st + fb + grp * 4 * BK; b[0] = bp[0]; b[1] = bp[1]; b[2] = bp[2]; }
            else b = *reinterpret_cast<const typename WlVec<NKT>::type*>(st + fb + grp * 4 * BK);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int j = 0; j < NKT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
                if (want_cs) accs[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], 1.0f, accs[i], 0, 0, 0);
            }
        }
    }
    wl_wait_all();

    // accumulator (tile i, tile j)[r] of lane (m, kq) = dW[n0 + wn*64 + 4*(4*kq + r) + i][k0 + wk*NKT*16 + NKT*m + j]
    float* out = part + (long long)slice * Nn * K;
    const int kb = k0 + wk * (NKT * 16) + NKT * m;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = n0 + wn * 64 + 4 * (4 * kq + r) + i;
            if (n >= Nn) continue;
#pragma unroll
            for (int j = 0; j < NKT; ++j)
                if (kb + j < K) out[(long long)n * K + kb + j] = acc[i][j][r];
            if (want_cs && m == 0 && wk == 0) cpart[(long long)slice * Nn + n] = accs[i][r];
        }
}

// Decision + geometry.  Taken for >= 64 output rows (Nn) and columns (K) with 16-byte-aligned rows; everything else stays on
// direct_wgrad_kernel.  cfg: 0 = 64 x 128 (1 x 4 waves, NKT 2), 1 = 64 x 192 (1 x 4, NKT 3), 2 = 128 x 128 (2 x 2, NKT 4),
// 3 = 128 x 64 (2 x 2, NKT 2), 4 = 128 x 96 (2 x 2, NKT 3).
struct WlPlan { int cfg, BN, BK, ntiles, ktiles, rows, S; };
inline bool wl_plan(WlPlan& pl, long long P, int Nn, int K, int lddy, int ldx, const float* dy, const float* x, const ConvGeo* geo) {
    static const int env = getenv("HYB_WGRAD_LDS") ? atoi(getenv("HYB_WGRAD_LDS")) : 1;
    if (!env || Nn < 64 || K < 64 || P < 4096 || lddy % 4 != 0 || Nn % 4 != 0 || K % 4 != 0) return false;
    if (geo ? geo->Ci % 4 != 0 : ldx % 4 != 0) return false;
    if (dy && x && ((((uintptr_t)dy) | ((uintptr_t)x)) & 15)) return false;
    const bool three = K % 192 == 0 || (K % 96 == 0 && Nn > 64);             // 9 * Ci columns: tile by thirds
    if (Nn <= 64) { pl.cfg = (K % 192 == 0) ? 1 : 0; pl.BN = 64; pl.BK = pl.cfg == 1 ? 192 : 128; }
    else if (K <= 64) { pl.cfg = 3; pl.BN = 128; pl.BK = 64; }
    else if (three && K % 128 != 0) { pl.cfg = 4; pl.BN = 128; pl.BK = 96; }
    else { pl.cfg = 2; pl.BN = 128; pl.BK = 128; }
    pl.ntiles = hyb_cdiv(Nn, pl.BN); pl.ktiles = hyb_cdiv(K, pl.BK);
    const long long tiles = (long long)pl.ntiles * pl.ktiles;
    long long S = (512 + tiles - 1) / tiles;                                   // two workgroups per CU: one round of the grid
    long long r = (P + S - 1) / S;
    r = (r + WL_PB - 1) / WL_PB * WL_PB;
    if (r < 256) r = 256;
    // the kernel addresses a slice with 32-bit byte offsets from its first row (dy, plain x) / its first image (IMPL): shorten the slices until
    // every operand's slice fits 2^31 bytes (more slabs), give up below 256 rows
    auto fits = [&](long long rows) {
        if (rows * lddy * 4 >= 0x7fffffffll) return false;
        if (geo) return (rows / ((long long)geo->Ho * geo->Wo) + 2) * geo->H * geo->W * geo->Ci * 4 < 0x7fffffffll;
        return rows * ldx * 4 < 0x7fffffffll;
    };
    while (!fits(r) && r > 256) r = ((r / 2) + WL_PB - 1) / WL_PB * WL_PB;
    if (!fits(r) || (P + r - 1) / r > 2048) return false;       // (the bias-gradient partials are sized for <= 2064 slices)
    pl.rows = (int)r;
    pl.S = (int)((P + r - 1) / r);
    return true;
}

template <int WN, int WK, int NKT>
void wl_go(const WlPlan& pl, const float* dy, int lddy, const float* x, int ldx, float* part, float* cpart, long long P, int Nn, int K, hipStream_t st,
           const ConvGeo* geo) {
    constexpr int lds = (int)WlGeom<WN, WK, NKT>::LDS_BYTES;
    static bool once = false;
    if (!once) {
        (void)hipFuncSetAttribute((const void*)wgrad_lds_kernel<WN, WK, NKT, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        (void)hipFuncSetAttribute((const void*)wgrad_lds_kernel<WN, WK, NKT, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        once = true;
    }
    const int tiles = pl.ktiles * pl.ntiles;
    const dim3 grid((unsigned)(hyb_cdiv(pl.S, 8) * 8 * tiles));
    if (geo) hipLaunchKernelGGL((wgrad_lds_kernel<WN, WK, NKT, true>), grid, dim3(256), lds, st, dy, lddy, x, ldx, part, cpart, P, Nn, K, pl.ktiles, pl.rows, *geo, tiles, pl.S);
    else hipLaunchKernelGGL((wgrad_lds_kernel<WN, WK, NKT, false>), grid, dim3(256), lds, st, dy, lddy, x, ldx, part, cpart, P, Nn, K, pl.ktiles, pl.rows, ConvGeo{}, tiles, pl.S);
}
inline void wl_launch(const WlPlan& pl, const float* dy, int lddy, const float* x, int ldx, float* part, float* cpart, long long P, int Nn, int K,
                      hipStream_t st, const ConvGeo* geo) {
    switch (pl.cfg) {
        case 0: wl_go<1, 4, 2>(pl, dy, lddy, x, ldx, part, cpart, P, Nn, K, st, geo); break;
        case 1: wl_go<1, 4, 3>(pl, dy, lddy, x, ldx, part, cpart, P, Nn, K, st, geo); break;
        case 2: wl_go<2, 2, 4>(pl, dy, lddy, x, ldx, part, cpart, P, Nn, K, st, geo); break;
        case 3: wl_go<2, 2, 2>(pl, dy, lddy, x, ldx, part, cpart, P, Nn, K, st, geo); break;
        default: wl_go<2, 2, 3>(pl, dy, lddy, x, ldx, part, cpart, P, Nn, K, st, geo); break;
    }
}
