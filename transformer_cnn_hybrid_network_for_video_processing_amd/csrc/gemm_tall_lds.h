// Tall fp32 GEMM through LDS (included by linear.hip after GemmArgs / ConvGather): the pixel-side products of FCT and Encoder_32K,
//   C[mo][no] = sum_r A[mo][r] * B[no][r]  (+ bias, ReLU, accumulate),   M = N*H*W rows (10^5 .. 10^6), No >= 64 columns.
//
// gemm_nt_tall_kernel keeps its fragments in registers and fetches every one of them from L2 / HBM itself: 12 KB per wave and 32-deep
// step, one step ahead.  At the exact-fp32 MFMA rate (64 FLOP/clk/SIMD) that is 0.094 B/FLOP = 7-8 TB/s of L2 traffic at the 82 TF/s it
// reaches, with one step (1 us) of latency cover: it runs at 0.4-0.5 of the 157 TF/s matrix peak.  Here
//   * a workgroup (4 waves, 2 x 2) owns 128 rows x BN columns (BN = 128 / 64): 0.031 / 0.047 B/FLOP;
//   * both operand tiles of a 32-deep step go global -> LDS by buffer_load_dwordx4 ... lds (no registers in the path), S stages deep; rows
//     beyond M / No, columns beyond R and -- IMPLICIT -- taps outside the image are lanes sent out of the descriptor's range: the hardware
//     writes zeros for them, so the padding costs no branch;
//   * one counted s_waitcnt vmcnt + one s_barrier per step; two workgroups per CU interleave their barrier / epilogue phases;
//   * LDS rows are 128 bytes (32 floats); the 16-byte chunk c of row r lives at position c ^ ((r >> 1) & 7): the 16 rows of a fragment
//     read (ds_read_b128) spread over all 64 banks;
//   * the product is formed transposed (weights as the first MFMA operand) like gemm_nt_tall_kernel: a lane ends with 4 consecutive
//     columns of a row, 16-byte stores.
// IMPLICIT: A is the NHWC image of a convolution (ConvGather), Ci a power of two >= 4: a 16-byte chunk is 4 channels of one tap.
// The split-bf16 build (-DHYB_F32_X3) compiles the same source; mma32 then forms its three bf16 products per fragment pair.
#pragma once

typedef __attribute__((address_space(3))) void gt_lds_void_t;
constexpr unsigned GT_OOB = 0xfffffff0u, GT_RECORDS = 0x80000000u;
// (a plain device function: the 16-byte form of the builtin is checked against the target, which must not happen in the host pass of a template)
__device__ __forceinline__ void gt_dma16(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, float* lds_dst_wave_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (gt_lds_void_t*)lds_dst_wave_base, 16, voff, 0, 0, 0);
}
template <int N> __device__ __forceinline__ void gt_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int BN, int S> struct GtGeom {
    static constexpr int BM = 128, NTW = BN / 32;
    static constexpr int A_I = BM / 32, B_I = BN / 32, LW = A_I + B_I;      // DMA instructions per wave and stage
    static constexpr int STAGE = (BM + BN) * 32;                             // floats
    static constexpr size_t LDS_BYTES = (size_t)S * STAGE * 4;
};

// korder (IMPLICIT, Ci % 32 == 0): walk K channel-block-major -- for each 32-channel block its k*k taps in a row -- instead of tap-major.
// Tap-major, the k*k shifted reads of a pixel's channels are Ci/32 steps (tens of us) apart and every one of them misses L2: the 3x3, 512-channel
// layer of Encoder_32K fetched 1.25 GB for 134 MB of input (profiles/r04_pmc_fetch_write_enc32k.csv).  Channel-block-major they are consecutive
// steps on a 25 KB working set per workgroup.  Row blocks are dealt to the XCDs in contiguous ranges for the same reason (neighbouring row
// blocks share their halo rows in one L2).
template <int BN, int S, bool IMPLICIT>
__global__ __launch_bounds__(256, 2) void gemm_nt_lds_kernel(GemmArgs args, int tiles_n, int row_blocks, ConvGather cg, int korder) {
    using G = GtGeom<BN, S>;
    constexpr int BM = G::BM, NTW = G::NTW, A_I = G::A_I, B_I = G::B_I, STAGE = G::STAGE;
    extern __shared__ __attribute__((aligned(16))) float gt_smem[];
    const GemmGroup grp = args.g[0];
    const float* A = (const float*)grp.A;
    const float* B = (const float*)grp.B;
    float* C = (float*)grp.C;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, seq = bid >> 3;
    // the column tiles of one row block share an XCD (its L2 holds A); korder: XCD x owns row blocks [x * per, (x + 1) * per)
    const int per_xcd = (row_blocks + 7) >> 3;
    const int tn = seq % tiles_n, rb = korder ? xcd * per_xcd + seq / tiles_n : (seq / tiles_n) * 8 + xcd;
    if (rb >= row_blocks || (korder && seq / tiles_n >= per_xcd)) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int p = lane & 15, q = lane >> 4;
    const int m0 = rb * BM, n0 = tn * BN;
    const int R = args.R;

    // ---- this lane's part of a stage: rows i*32 + rl of either tile, chunk c of the 32-deep step
    const int rl = wave * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((rl >> 1) & 7);
    unsigned aoff[A_I];                        // byte offset of the row from the descriptor base (IMPLICIT: of its receptive field's origin)
    int ah[A_I], aw[A_I];
    const int hw = IMPLICIT ? cg.Ho * cg.Wo : 1;
    const int n_first = m0 / hw;
    const __amdgpu_buffer_rsrc_t ars = hyb_rsrc(IMPLICIT ? A + ((long long)n_first * cg.H * cg.W << cg.log2ci) : A + (long long)m0 * args.lda, GT_RECORDS);
    const __amdgpu_buffer_rsrc_t brs = hyb_rsrc(B + (long long)n0 * args.ldb, GT_RECORDS);
    if (IMPLICIT) {
#pragma unroll
        for (int i = 0; i < A_I; ++i) {
            const int r = m0 + i * 32 + rl;
            const int n = r / hw, t = r - n * hw;
            const int ho = t / cg.Wo, wo = t - ho * cg.Wo;
            const int h0 = ho * cg.stride - cg.pad;
            aw[i] = wo * cg.stride - cg.pad;
            aoff[i] = (unsigned)((((n - n_first) * cg.H + h0) * cg.W + aw[i]) << cg.log2ci) << 2;
            ah[i] = r < args.Mo ? h0 : -0x100000;                           // rows beyond M: never inside the image
        }
    } else {
#pragma unroll
        for (int i = 0; i < A_I; ++i) {
            const int r = i * 32 + rl;
            ah[i] = (m0 + r < args.Mo) ? 0 : -1; aw[i] = 0;
            aoff[i] = (unsigned)(r * args.lda + 4 * c) << 2;
        }
    }
    unsigned boff[B_I];
    bool bok[B_I];
#pragma unroll
    for (int i = 0; i < B_I; ++i) {
        const int r = i * 32 + rl;
        bok[i] = n0 + r < args.No;
        boff[i] = (unsigned)(r * args.ldb + 4 * c) << 2;
    }
    auto issue = [&](int buf, int k0) {
        float* const st = gt_smem + buf * STAGE;
        const int kc = k0 + 4 * c;
        const bool kok = kc + 4 <= R;
        if (IMPLICIT) {
            const int tap = kc >> cg.log2ci, ci = kc & ((1 << cg.log2ci) - 1);
            const int ky = (tap * cg.kdiv) >> 16, kx = tap - ky * cg.k;
            const int dh = ky * cg.dil, dw = kx * cg.dil;
            const unsigned tapoff = (unsigned)(((dh * cg.W + dw) << cg.log2ci) + ci) << 2;
            const bool tok = kc < cg.kk_ci;
#pragma unroll
            for (int i = 0; i < A_I; ++i) {
                const bool ok = tok & ((unsigned)(ah[i] + dh) < (unsigned)cg.H) & ((unsigned)(aw[i] + dw) < (unsigned)cg.W);   // (no short circuit: no branches)
                gt_dma16(ars, ok ? aoff[i] + tapoff : GT_OOB, st + (i * 32 + wave * 8) * 32);
            }
        } else {
#pragma unroll
            for (int i = 0; i < A_I; ++i)
                gt_dma16(ars, (kok & (ah[i] == 0)) ? aoff[i] + ((unsigned)k0 << 2) : GT_OOB, st + (i * 32 + wave * 8) * 32);
        }
#pragma unroll
        for (int i = 0; i < B_I; ++i)
            gt_dma16(brs, (kok & bok[i]) ? boff[i] + ((unsigned)k0 << 2) : GT_OOB, st + (BM + i * 32 + wave * 8) * 32);
    };

    // ---- fragment addresses: row (16 rows of a tile = lanes p), chunks 2q and 2q+1 of the step
    const int wm = wave >> 1, wn = wave & 1;
    const int fa = (wm * 64 + p) * 32 + (((2 * q) ^ (p >> 1)) << 2);
    const int fb = (BM + wn * (BN / 2) + p) * 32 + (((2 * q) ^ (p >> 1)) << 2);
    f32x4 acc[4][NTW];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NTW; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto lds_frag = [&](Frag<float>& f, const float* st, int off) {
        const f32x4 u = *reinterpret_cast<const f32x4*>(st + off), v = *reinterpret_cast<const f32x4*>(st + (off ^ 4));
        f.v[0] = u[0]; f.v[1] = u[1]; f.v[2] = u[2]; f.v[3] = u[3];
        f.v[4] = v[0]; f.v[5] = v[1]; f.v[6] = v[2]; f.v[7] = v[3];
    };

    const int nk = (R + 31) >> 5;
    const int kk = cg.k * cg.k, cstride = IMPLICIT ? (1 << cg.log2ci) : 0;
    auto k_of = [&](int it) { return (IMPLICIT && korder) ? (it % kk) * cstride + (it / kk) * 32 : it * 32; };      // (beyond nk: >= R, all zeros)
#pragma unroll
    for (int s = 0; s < S - 1; ++s) issue(s, s < nk ? k_of(s) : R);         // (steps beyond R: every lane out of range, zeros)
    int buf = 0, nbuf = S - 1;
    for (int it = 0; it < nk; ++it) {
        gt_wait_vmcnt<(S - 2) * G::LW>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        issue(nbuf, it + S - 1 < nk ? k_of(it + S - 1) : R);                 // into the buffer every wave finished reading before this barrier
        const float* st = gt_smem + buf * STAGE;
        Frag<float> a[4], b[NTW];
#pragma unroll
        for (int j = 0; j < NTW; ++j) lds_frag(b[j], st, fb + j * 16 * 32);
#pragma unroll
        for (int i = 0; i < 4; ++i) lds_frag(a[i], st, fa + i * 16 * 32);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NTW; ++j) acc[i][j] = mma32(b[j], a[i], acc[i][j]);
        buf = buf + 1 == S ? 0 : buf + 1;
        nbuf = nbuf + 1 == S ? 0 : nbuf + 1;
    }
    gt_wait_vmcnt<0>();                                                      // the trailing (all-zero) stages must land before the workgroup's LDS is released

    const bool vec = (args.ldc & 3) == 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int mo = m0 + wm * 64 + i * 16 + p;
        if (mo >= args.Mo) continue;
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            const int no = n0 + wn * (BN / 2) + j * 16 + 4 * q;
            if (no >= args.No) continue;
            float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
            float* dst = C + (long long)mo * args.ldc + no;
            if (vec && no + 4 <= args.No) {
                if (grp.bias) { const f32x4 bb = *reinterpret_cast<const f32x4*>(grp.bias + no); v[0] += bb[0]; v[1] += bb[1]; v[2] += bb[2]; v[3] += bb[3]; }
                if (args.relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
                if (args.accumulate) { const f32x4 o = *reinterpret_cast<const f32x4*>(dst); v[0] += o[0]; v[1] += o[1]; v[2] += o[2]; v[3] += o[3]; }
                *reinterpret_cast<f32x4*>(dst) = f32x4{v[0], v[1], v[2], v[3]};
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (no + r >= args.No) continue;
                    float t = v[r];
                    if (grp.bias) t += grp.bias[no + r];
                    if (args.relu) t = fmaxf(t, 0.f);
                    if (args.accumulate) t += dst[r];
                    dst[r] = t;
                }
            }
        }
    }
}

// Shapes the LDS kernel takes: 16-byte-aligned operands and row strides, at least 64 columns.  Returns false -> gemm_nt_tall_kernel.
bool gt_lds_ok(const GemmArgs& a, bool implicit) {
    static const int env = getenv("HYB_GEMM_LDS") ? atoi(getenv("HYB_GEMM_LDS")) : 1;
    if (!env || a.No < 64 || a.R % 4 != 0 || a.ldb % 4 != 0) return false;
    if (((uintptr_t)a.g[0].A | (uintptr_t)a.g[0].B) & 15) return false;
    if (!implicit && (a.lda % 4 != 0 || (long long)a.lda * 128 * 4 >= 0x7fffffffll)) return false;      // (32-bit byte offsets inside a 128-row block)
    if ((long long)a.ldb * 128 * 4 >= 0x7fffffffll) return false;
    return true;
}

template <int BN, int S, bool IMPLICIT>
void gt_go(const GemmArgs& a, const ConvGather& cg, long long blocks, int tiles_n, int row_blocks, hipStream_t st) {
    constexpr int lds = (int)GtGeom<BN, S>::LDS_BYTES;
    static bool once = false;
    if (!once) {
        (void)hipFuncSetAttribute((const void*)gemm_nt_lds_kernel<BN, S, IMPLICIT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        once = true;
    }
    static const int order_env = getenv("HYB_GEMM_LDS_ORDER") ? atoi(getenv("HYB_GEMM_LDS_ORDER")) : 1;      // (=0: A/B, tap-major K walk)
    const int ci = IMPLICIT ? (1 << cg.log2ci) : 0;
    const int korder = (IMPLICIT && order_env && ci % 32 == 0 && a.R == cg.k * cg.k * ci && cg.k > 1) ? 1 : 0;
    hipLaunchKernelGGL((gemm_nt_lds_kernel<BN, S, IMPLICIT>), dim3((unsigned)blocks), dim3(256), lds, st, a, tiles_n, row_blocks, cg, korder);
}

template <bool IMPLICIT>
int gt_launch(const GemmArgs& a, const ConvGather& cg, hipStream_t st) {
    static const int s_env = getenv("HYB_GEMM_LDS_S") ? atoi(getenv("HYB_GEMM_LDS_S")) : 0;
    const int row_blocks = hyb_cdiv(a.Mo, 128);
    const bool wide = a.No > 64;
    const int tiles_n = hyb_cdiv(a.No, wide ? 128 : 64);
    const long long blocks = (long long)hyb_cdiv(row_blocks, 8) * tiles_n * 8;
    if (blocks > 0x7fffffff) return HYB_E_ARG;
    if (wide) { if (s_env == 3) gt_go<128, 3, IMPLICIT>(a, cg, blocks, tiles_n, row_blocks, st); else gt_go<128, 2, IMPLICIT>(a, cg, blocks, tiles_n, row_blocks, st); }
    else { if (s_env == 2) gt_go<64, 2, IMPLICIT>(a, cg, blocks, tiles_n, row_blocks, st); else gt_go<64, 3, IMPLICIT>(a, cg, blocks, tiles_n, row_blocks, st); }
    return 0;
}
