// nn.Linear forward/backward (TransformerEncoder.pyc src L12-15/L69/L87 projections, L107 FFN) as MFMA GEMMs.
//
// One kernel template computes  C[mo][no] = sum_r A(mo, r) * B(no, r)  for up to 3 independent groups
// (blockIdx.z; used to run the Q, K and V projections in one launch).  Operand tiles are staged into LDS as
// [row][r] (r contiguous, 32 deep) whatever the source orientation:
//   normal source     : element (row, r) at p[row*ld + r]   (16-byte vector loads along r)
//   transposed source : element (row, r) at p[r*ld + row]   (vector loads along row, scattered LDS writes)
// fp32 sources (the master weights) are converted to T while staging.  Token counts (M = B*T = 128..512) are
// small, so these GEMMs are latency-bound; tiles are 64x64 or 32x32 to spread them over more CUs.
#include <stdlib.h>
#include "hyb_common.h"
#include "ln_rows.h"

int hyb_gemm_nt(int dtype, int groups, const void* const* A, const void* const* B, void* const* C, const float* const* bias, int out_f32,
                int Mo, int No, int R, int lda, int ldb, int ldc, int relu, int accumulate, hipStream_t st, const void* const* Amask = nullptr,
                const void* const* Cmask = nullptr);
int hyb_linear_dw_multi(int dtype, int groups, const void* const* dy, const void* const* mask, const void* const* x, float* const* dW,
                        float* const* db, const int* N, const int* K, const int* lddy, const int* ldx, int M, hipStream_t st,
                        int nriders, const HybDwRider* riders);
int hyb_gemm_skinny_wf32(int dtype, const void* A, const float* Bf, void* C, const float* bias, int Mo, int No, int R, int lda, int ldb, int ldc, int relu,
                         int accumulate, int transposed_b, hipStream_t st);

namespace {

struct GemmGroup {
    const void* A;
    const void* B;
    void* C;
    const float* bias;
    const void* Amask;      // optional, same layout/type as A: A is used as A * (Amask > 0)  (ReLU backward fused into the loader)
    float* colsum;          // optional (dW kernel): colsum[mo] = sum_r A(mo, r)   (bias gradient fused into the weight-gradient GEMM)
    const void* Cmask;      // optional (skinny kernel), same layout/type as C: the result is stored as C * (Cmask > 0) -- the ReLU backward of the
                            // layer BELOW fused into this product's epilogue, so that its consumers need no mask operand
};
struct GemmArgs {
    GemmGroup g[3];
    int Mo, No, R;          // output rows, output cols, reduction length
    int lda, ldb, ldc;
    int relu, accumulate;
};

constexpr int BK = 32, LDS_PAD = 8, LDS_ROW = BK + LDS_PAD;

template <typename TS, typename T, bool TRANS, int ROWS, int LDT = LDS_ROW>
__device__ __forceinline__ void stage_tile(const TS* __restrict__ src, const TS* __restrict__ mask, int ld, int row0, int nrows, int r0, int R,
                                           T* __restrict__ tile, int tid) {
    if (!TRANS) {
        if (tid < ROWS * 4) {
            const int row = tid >> 2, seg = tid & 3;
            const int gr = row0 + row, r = r0 + seg * 8;
            Vec8<T> v;
            if (gr < nrows && r + 8 <= R) {
                Vec8<TS> s;
                s.load(src + (long long)gr * ld + r);
                if (mask) {
                    Vec8<TS> mk;
                    mk.load(mask + (long long)gr * ld + r);
#pragma unroll
                    for (int j = 0; j < 8; ++j) v.set(j, mk.get(j) > 0.f ? s.get(j) : 0.f);
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) v.set(j, s.get(j));
                }
            } else {
                v.zero();
            }
            v.store(tile + row * LDT + seg * 8);
        }
    } else {
        if (tid < ROWS * 4) {
            const int r = tid / (ROWS / 8), seg = tid % (ROWS / 8);
            const int gr = row0 + seg * 8, rr = r0 + r;
            Vec8<TS> s;
            if (rr < R && gr + 8 <= nrows) {
                s.load(src + (long long)rr * ld + gr);
                if (mask) {
                    Vec8<TS> mk;
                    mk.load(mask + (long long)rr * ld + gr);
#pragma unroll
                    for (int j = 0; j < 8; ++j) if (!(mk.get(j) > 0.f)) s.set(j, 0.f);
                }
            } else {
                s.zero();
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) tile[(seg * 8 + j) * LDT + r] = from_f32<T>(s.get(j));
        }
    }
}

// KS: 32-deep k-steps staged per round.  The 32 x 32-tile instantiations serve few-tile, latency-bound products (the token projection:
// 32 workgroups); with KS = 4 a round has four times the loads in flight and a quarter of the barriers (15.4 -> 8 us for its dX product).
// Same k order and MFMA grouping: bit-identical results.
template <typename T, typename TA, typename TB, typename TC, bool TRANS_A, bool TRANS_B, int BM, int KS = 1>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs args) {
    constexpr int BN = BM;
    constexpr int WM = BM / 2, WN = BN / 2;            // per-wave sub-tile (2x2 waves)
    constexpr int MT = WM / 16, NTT = WN / 16;
    constexpr int LDR = KS * BK + LDS_PAD;             // LDS row stride (elements)
    __shared__ __attribute__((aligned(16))) T As[BM * LDR];
    __shared__ __attribute__((aligned(16))) T Bs[BN * LDR];
    const GemmGroup grp = args.g[blockIdx.z];
    const TA* A = (const TA*)grp.A;
    const TB* B = (const TB*)grp.B;
    TC* C = (TC*)grp.C;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int p = lane & 15, q = lane >> 4;

    f32x4 acc[MT][NTT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NTT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float csum = 0.f;

    for (int r0 = 0; r0 < args.R; r0 += KS * BK) {
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {                               // (sub-steps past R stage zeros: exact no-ops in the sums)
            stage_tile<TA, T, TRANS_A, BM, LDR>(A, (const TA*)grp.Amask, args.lda, m0, args.Mo, r0 + ks * BK, args.R, As + ks * BK, tid);
            stage_tile<TB, T, TRANS_B, BN, LDR>(B, (const TB*)nullptr, args.ldb, n0, args.No, r0 + ks * BK, args.R, Bs + ks * BK, tid);
        }
        __syncthreads();
        if (grp.colsum && blockIdx.x == 0 && tid < BM) {            // bias gradient: row sums of the (masked) A tile
#pragma unroll 8
            for (int r = 0; r < KS * BK; ++r) csum += to_f32<T>(As[tid * LDR + r]);
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            Frag<T> a[MT], b[NTT];
#pragma unroll
            for (int i = 0; i < MT; ++i) frag_load(a[i], As + (wm * WM + i * 16 + p) * LDR + ks * BK + 8 * q);
#pragma unroll
            for (int j = 0; j < NTT; ++j) frag_load(b[j], Bs + (wn * WN + j * 16 + p) * LDR + ks * BK + 8 * q);
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NTT; ++j) acc[i][j] = mma32(a[i], b[j], acc[i][j]);
        }
    }
    if (grp.colsum && blockIdx.x == 0 && tid < BM && m0 + tid < args.Mo) grp.colsum[m0 + tid] = csum;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NTT; ++j) {
            const int no = n0 + wn * WN + j * 16 + p;
            if (no >= args.No) continue;
            const float bias = grp.bias ? grp.bias[no] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int mo = m0 + wm * WM + i * 16 + 4 * q + r;
                if (mo >= args.Mo) continue;
                float v = acc[i][j][r] + bias;
                if (args.relu) v = fmaxf(v, 0.f);
                TC* dst = C + (long long)mo * args.ldc + no;
                if (args.accumulate) v += to_f32<TC>(*dst);
                *dst = from_f32<TC>(v);
            }
        }
}

// Weight (+ bias) gradients of up to 13 Linear layers with DIFFERENT shapes in one launch (the six matrices of each of two encoder layers
// and the frame-token projection):
//   dW_g[n][k] = sum_m dym_g[m][n] * x_g[m][k],   db_g[n] = sum_m dym_g[m][n],   dym_g = dy_g * (mask_g > 0) (mask optional).
// Flat grid: workgroup -> (group, 64x64 tile) through the groups' tile offsets.
struct DwGroup {
    const void* dy; const void* mask; const void* x;
    float* dW; float* db;
    int N, K, lddy, ldx;       // dW is [N][K]; dy rows have lddy elements, x rows ldx
    int tiles_x, tile_begin;
};
constexpr int DW_MAX_GROUPS = 13;          // two encoder layers' six matrices + the frame-token projection
constexpr int DW_MAX_RIDERS = 3;           // two layers' LayerNorm partial rows + the head's
struct DwArgs {
    DwGroup g[DW_MAX_GROUPS]; int ngroups, M;
    // optional riders (HybDwRider, hyb_common.h: fixed-order sums of partial rows -- the encoder backward's LayerNorm-affine gradients, one
    // per layer, and the head's weight / bias gradient terms): blocks past `tiles`, 256 columns each; rider r starts at block rider_begin[r]
    int tiles, nriders; HybDwRider rd[DW_MAX_RIDERS]; int rider_begin[DW_MAX_RIDERS + 1];
};
// column c of a rider's partial rows, rows in ascending order, eight loads in flight
__device__ __forceinline__ void dw_rider_block(const DwArgs& args) {
    int rid = 0;
#pragma unroll
    for (int i = 1; i < DW_MAX_RIDERS; ++i)
        if (i < args.nriders && (int)blockIdx.x >= args.rider_begin[i]) rid = i;
    const HybDwRider rd = args.rd[rid];
    const long long c = (long long)((int)blockIdx.x - args.rider_begin[rid]) * 256 + (int)threadIdx.x;
    if (c >= rd.n) return;
    const float* col = rd.part + c;
    float s = 0.f;
    int r = 0;
    for (; r + 8 <= rd.rows; r += 8) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = col[(long long)(r + j) * rd.n];
#pragma unroll
        for (int j = 0; j < 8; ++j) s += v[j];
    }
    for (; r < rd.rows; ++r) s += col[(long long)r * rd.n];
    if (c < rd.split) rd.out0[c] = s; else rd.out1[c - rd.split] = s;
}

// bf16 form of the kernel below without the transposing LDS stores (they conflicted on 83 % of the cycles): the two operand tiles are
// staged as they lie in memory -- [row m][64 columns], 16-byte loads and stores, ALL of up to 128 rows in one round (one memory latency
// instead of four) -- and the fragments, whose K dimension is the row m, come from the transposing LDS read ds_read_b64_tr_b16.
constexpr int DWT_ROWS = 128, DWT_STRIDE = 64 + 16;        // row stride 80 elements: conflict-free transposing reads (as conv_wgrad.hip)
typedef __attribute__((address_space(3))) bf16x4 dwt_lds_q;
__device__ __forceinline__ void dwt_frag(Frag<bf16>& f, const bf16* tile, int m0, int c0, int lane) {
    // 8 K values (rows m0 + 4q + 0..3 and m0 + 16 + 4q + 0..3, q = lane >> 4) of column c0 + (lane & 15)
    const int qq = (lane & 15) >> 2, pp = lane & 3, q = lane >> 4;
    const bf16* a0 = tile + (m0 + 4 * q + qq) * DWT_STRIDE + c0 + 4 * pp;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((dwt_lds_q*)a0);
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((dwt_lds_q*)(a0 + 16 * DWT_STRIDE));
    f.v[0] = lo[0]; f.v[1] = lo[1]; f.v[2] = lo[2]; f.v[3] = lo[3];
    f.v[4] = hi[0]; f.v[5] = hi[1]; f.v[6] = hi[2]; f.v[7] = hi[3];
}
__global__ __launch_bounds__(256) void gemm_dw_multi_tr_kernel(DwArgs args) {
    __shared__ __attribute__((aligned(16))) bf16 As[DWT_ROWS * DWT_STRIDE];
    __shared__ __attribute__((aligned(16))) bf16 Bs[DWT_ROWS * DWT_STRIDE];
    if ((int)blockIdx.x >= args.tiles) { dw_rider_block(args); return; }
    int gi = 0;
#pragma unroll
    for (int i = 1; i < DW_MAX_GROUPS; ++i)
        if (i < args.ngroups && (int)blockIdx.x >= args.g[i].tile_begin) gi = i;
    const DwGroup grp = args.g[gi];
    const int local = blockIdx.x - grp.tile_begin;
    const int bx = local % grp.tiles_x, by = local / grp.tiles_x;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = by * 64, n0 = bx * 64;                  // m0: rows of dW (columns of dy), n0: columns of dW (columns of x)
    const int p = lane & 15, q = lane >> 4;
    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float csum = 0.f;
    const bf16* dy = (const bf16*)grp.dy;
    const bf16* mk = (const bf16*)grp.mask;
    const bf16* x = (const bf16*)grp.x;
    const int seg = tid & 7, row_t = tid >> 3;             // thread: 8 columns seg*8.., rows row_t + 32 u
    for (int r0 = 0; r0 < args.M; r0 += DWT_ROWS) {
        if (r0 > 0) __syncthreads();
        Vec8<bf16> va[4], vm[4], vb[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = r0 + row_t + 32 * u;
            const bool rok = r < args.M;
            if (rok && m0 + seg * 8 < grp.N) {
                va[u].load(dy + (long long)r * grp.lddy + m0 + seg * 8);
                if (mk) vm[u].load(mk + (long long)r * grp.lddy + m0 + seg * 8);
            } else { va[u].zero(); if (mk) vm[u].zero(); }
            if (rok && n0 + seg * 8 < grp.K) vb[u].load(x + (long long)r * grp.ldx + n0 + seg * 8); else vb[u].zero();
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (mk) {
#pragma unroll
                for (int j = 0; j < 8; ++j) if (!((float)vm[u].v[j] > 0.f)) va[u].v[j] = (bf16)0.f;
            }
            va[u].store(As + (row_t + 32 * u) * DWT_STRIDE + seg * 8);
            vb[u].store(Bs + (row_t + 32 * u) * DWT_STRIDE + seg * 8);
        }
        __syncthreads();
        const int rows = args.M - r0 < DWT_ROWS ? args.M - r0 : DWT_ROWS;
        if (grp.db && bx == 0 && tid < 64) {                // bias gradient: column sums of the (masked) dy tile, rows in ascending order
            for (int r = 0; r < rows; ++r) csum += (float)As[r * DWT_STRIDE + tid];
        }
        for (int k0 = 0; k0 < rows; k0 += 32) {             // rows beyond M are zero: a ragged last step contributes nothing
            Frag<bf16> a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) dwt_frag(a[i], As, k0, wm * 32 + i * 16, lane);
#pragma unroll
            for (int j = 0; j < 2; ++j) dwt_frag(b[j], Bs, k0, wn * 32 + j * 16, lane);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = mma32(a[i], b[j], acc[i][j]);
        }
    }
    if (grp.db && bx == 0 && tid < 64 && m0 + tid < grp.N) grp.db[m0 + tid] = csum;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int no = n0 + wn * 32 + j * 16 + p;
            if (no >= grp.K) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int mo = m0 + wm * 32 + i * 16 + 4 * q + r;
                if (mo < grp.N) grp.dW[(long long)mo * grp.K + no] = acc[i][j][r];
            }
        }
}

template <typename T>
__global__ __launch_bounds__(256) void gemm_dw_multi_kernel(DwArgs args) {
    constexpr int BM = 64, WM = 32, MT = 2;
    __shared__ __attribute__((aligned(16))) T As[BM * LDS_ROW];
    __shared__ __attribute__((aligned(16))) T Bs[BM * LDS_ROW];
    if ((int)blockIdx.x >= args.tiles) { dw_rider_block(args); return; }
    int gi = 0;
#pragma unroll
    for (int i = 1; i < DW_MAX_GROUPS; ++i)
        if (i < args.ngroups && (int)blockIdx.x >= args.g[i].tile_begin) gi = i;
    const DwGroup grp = args.g[gi];
    const int local = blockIdx.x - grp.tile_begin;
    const int bx = local % grp.tiles_x, by = local / grp.tiles_x;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = by * BM, n0 = bx * BM;                 // m: rows of dW (N), n: columns of dW (K)
    const int p = lane & 15, q = lane >> 4;
    f32x4 acc[MT][MT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float csum = 0.f;
    for (int r0 = 0; r0 < args.M; r0 += BK) {
        __syncthreads();
        stage_tile<T, T, true, BM>((const T*)grp.dy, (const T*)grp.mask, grp.lddy, m0, grp.N, r0, args.M, As, tid);
        stage_tile<T, T, true, BM>((const T*)grp.x, (const T*)nullptr, grp.ldx, n0, grp.K, r0, args.M, Bs, tid);
        __syncthreads();
        if (grp.db && bx == 0 && tid < BM) {
#pragma unroll 8
            for (int r = 0; r < BK; ++r) csum += to_f32<T>(As[tid * LDS_ROW + r]);
        }
        Frag<T> a[MT], b[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i) frag_load(a[i], As + (wm * WM + i * 16 + p) * LDS_ROW + 8 * q);
#pragma unroll
        for (int j = 0; j < MT; ++j) frag_load(b[j], Bs + (wn * WM + j * 16 + p) * LDS_ROW + 8 * q);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < MT; ++j) acc[i][j] = mma32(a[i], b[j], acc[i][j]);
    }
    if (grp.db && bx == 0 && tid < BM && m0 + tid < grp.N) grp.db[m0 + tid] = csum;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) {
            const int no = n0 + wn * WM + j * 16 + p;
            if (no >= grp.K) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int mo = m0 + wm * WM + i * 16 + 4 * q + r;
                if (mo < grp.N) grp.dW[(long long)mo * grp.K + no] = acc[i][j][r];
            }
        }
}

template <typename T, typename TA, typename TB, typename TC, bool TRANS_A, bool TRANS_B>
int launch_gemm(const GemmArgs& a, int groups, hipStream_t st) {
    const long long tiles64 = (long long)hyb_cdiv(a.Mo, 64) * hyb_cdiv(a.No, 64) * groups;
    if (tiles64 >= 128) {
        dim3 grid(hyb_cdiv(a.No, 64), hyb_cdiv(a.Mo, 64), groups);
        hipLaunchKernelGGL((gemm_kernel<T, TA, TB, TC, TRANS_A, TRANS_B, 64>), grid, dim3(256), 0, st, a);
    } else {
        dim3 grid(hyb_cdiv(a.No, 32), hyb_cdiv(a.Mo, 32), groups);
        static const int ks4 = getenv("HYB_GEMM_KS4") ? atoi(getenv("HYB_GEMM_KS4")) : 1;       // (=0: A/B, one k-step per round)
        if (ks4 && a.R >= 128) hipLaunchKernelGGL((gemm_kernel<T, TA, TB, TC, TRANS_A, TRANS_B, 32, 4>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((gemm_kernel<T, TA, TB, TC, TRANS_A, TRANS_B, 32>), grid, dim3(256), 0, st, a);
    }
    HYB_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------------------
// Skinny-M GEMM for the token side (M = B*T = 128..512 rows):  C[mo][no] = sum_r A[mo][r] * B[no][r], A and B both
// of type T, r-contiguous (weights pre-converted/pre-transposed once per step by convert_weights_kernel).
// No LDS staging and no barrier in the K loop: every lane loads its MFMA fragments (16 B) straight from global
// (operands are L2-resident: <= 4 MB), the 4 waves of a workgroup split K (wave w takes k-steps w, w+4, ...), and the
// partial 32x32 tiles are combined through LDS once at the end.  Grid = (No/32, Mo/32, groups).
// ---------------------------------------------------------------------------------------------------------
// BMODE: where the B operand (the weights) comes from.  0: T [No][R], r-contiguous (the pre-converted copies of the encoder);
//   1: the fp32 MASTER weights [No][R] (r-contiguous: two 16-byte loads + conversion per fragment);  2: fp32 master weights [R][No]
//   (the transposed product dx = dy W: a lane's 8 r values are 8 rows, 16 lanes cover 64 contiguous bytes of each).  1 and 2 serve the
//   Linear layers outside the encoder (frame-token projection), whose few-tile products took 9 / 14 us on the LDS-staged kernel.
template <typename T, typename TC, int NWV, int BMODE = 0>
__global__ __launch_bounds__(NWV * 64) void gemm_nt_splitk_kernel(GemmArgs args) {
    __shared__ float red[NWV][32][33];
    const GemmGroup grp = args.g[blockIdx.z];
    const T* A = (const T*)grp.A;
    const T* B = (const T*)grp.B;
    TC* C = (TC*)grp.C;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int p = lane & 15, q = lane >> 4;
    const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
    const T* Mk = (const T*)grp.Amask;
    const T* arow[2];
    const T* mrow[2];
    const T* brow[2];
    int bcol[2];
    const float* Bf = (const float*)grp.B;                 // BMODE 1, 2
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int r = m0 + i * 16 + p; if (r > args.Mo - 1) r = args.Mo - 1;
        int c = n0 + i * 16 + p; if (c > args.No - 1) c = args.No - 1;
        arow[i] = A + (long long)r * args.lda + 8 * q;
        mrow[i] = Mk ? Mk + (long long)r * args.lda + 8 * q : nullptr;
        brow[i] = B + (long long)c * args.ldb + 8 * q;
        bcol[i] = c;
    }
    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int R = args.R;
#pragma unroll 4
    for (int k0 = wave * 32; k0 < R; k0 += NWV * 32) {
        Frag<T> a[2], b[2];
        const bool ok = (k0 + 8 * q + 8) <= R;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (ok) {
                frag_load(a[i], arow[i] + k0);
                if (BMODE == 0) frag_load(b[i], brow[i] + k0);
                else if (BMODE == 1) {
                    const float* src = Bf + (long long)bcol[i] * args.ldb + k0 + 8 * q;
                    const f32x4 lo = *reinterpret_cast<const f32x4*>(src), hi = *reinterpret_cast<const f32x4*>(src + 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) { b[i].v[j] = from_f32<T>(lo[j]); b[i].v[4 + j] = from_f32<T>(hi[j]); }
                } else {
                    const float* src = Bf + (long long)(k0 + 8 * q) * args.ldb + bcol[i];
#pragma unroll
                    for (int j = 0; j < 8; ++j) b[i].v[j] = from_f32<T>(src[(long long)j * args.ldb]);
                }
                if (Mk) {
                    Frag<T> mk;
                    frag_load(mk, mrow[i] + k0);
#pragma unroll
                    for (int j = 0; j < 8; ++j) if (!((float)mk.v[j] > 0.f)) a[i].v[j] = (T)0.0f;
                }
            } else { frag_zero(a[i]); frag_zero(b[i]); }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = mma32(a[i], b[j], acc[i][j]);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wave][i * 16 + 4 * q + r][j * 16 + p] = acc[i][j][r];
    __syncthreads();
    // 1024 outputs: thread -> row, 1024 / (NWV * 64) consecutive columns; the wave partials are added in a fixed order
    constexpr int CPT = 1024 / (NWV * 64), TPR = 32 / CPT;
    const int row = tid / TPR, c0 = (tid % TPR) * CPT;
    const int mo = m0 + row;
    if (mo >= args.Mo) return;
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
        const int no = n0 + c0 + c;
        if (no >= args.No) continue;
        float v = (red[0][row][c0 + c] + red[1][row][c0 + c]) + (red[2][row][c0 + c] + red[3][row][c0 + c]);
        if (NWV == 8) v += (red[4][row][c0 + c] + red[5][row][c0 + c]) + (red[6][row][c0 + c] + red[7][row][c0 + c]);
        if (grp.bias) v += grp.bias[no];
        if (args.relu) v = fmaxf(v, 0.f);
        TC* dst = C + (long long)mo * args.ldc + no;
        if (args.accumulate) v += to_f32<TC>(*dst);
        if (grp.Cmask && !(to_f32<TC>(((const TC*)grp.Cmask)[(long long)mo * args.ldc + no]) > 0.f)) v = 0.f;
        *dst = from_f32<TC>(v);
    }
}

// The skinny product with LayerNorm + residual as its PROLOGUE:  C = act(LN(x + skip) . B^T + bias), the A operand formed by the workgroup
// itself.  TransformerEncoder.forward normalises (src L116-117, L120-123) and immediately projects (the feed-forward's first Linear, the next
// layer's Q / K / V projections); as launches of their own the two LayerNorms of a layer were 2 x 4.8 us for 128 rows (a dependent launch
// costs >= 4.6 us whatever it computes).  Here the eight waves of a workgroup normalise its 32 rows first -- four rows each, every load in
// flight at once, the arithmetic of ln_residual_fwd_kernel -- into an LDS image that replaces the global A operand; the column-0 workgroups
// of group 0 also write the rows and their statistics to global memory (the residual of the next LayerNorm and the backward pass read them).
// The weight fragments of a wave's first two k-steps are requested before the LayerNorm: their latency and the rows' overlap.
struct LnPro {
    const void* x; const void* skip; const float* gamma; const float* beta; void* y; float* stats;
    float eps, out_scale, p_drop; unsigned long long seed; const unsigned long long* seed_inc;
};
constexpr int LNG_PAD = 16;                          // image row stride R + 16 elements: conflict-free 16-byte fragment reads (ds_read_b128 lane groups)
template <int MAXC>
__global__ __launch_bounds__(512) void gemm_nt_ln_kernel(GemmArgs args, LnPro ln) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lng_smem[];
    bf16* const aimg = reinterpret_cast<bf16*>(lng_smem);                          // [32][R + LNG_PAD]
    float (*red)[32][33] = reinterpret_cast<float (*)[32][33]>(lng_smem);          // [8][32][33], after the k loop
    const GemmGroup grp = args.g[blockIdx.z];
    const bf16* B = (const bf16*)grp.B;
    bf16* C = (bf16*)grp.C;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int p = lane & 15, q = lane >> 4;
    const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
    const int R = args.R, ld = R + LNG_PAD;
    const bf16* brow[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int c = n0 + i * 16 + p; if (c > args.No - 1) c = args.No - 1;
        brow[i] = B + (long long)c * args.ldb + 8 * q;
    }
    // weight fragments of this wave's first two k-steps (k0 = 32 wave, 32 wave + 256): independent of the rows
    Frag<bf16> bpre[2][2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int k0 = wave * 32 + s * 256;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (k0 + 8 * q + 8 <= R) frag_load(bpre[s][i], brow[i] + k0); else frag_zero(bpre[s][i]);
        }
    }
    const bool writer = blockIdx.x == 0 && blockIdx.z == 0;
    ln_fwd_rows_lds<bf16, MAXC, 4>((const bf16*)ln.x, (const bf16*)ln.skip, ln.gamma, ln.beta, writer ? (bf16*)ln.y : nullptr, writer ? ln.stats : nullptr,
                                   aimg, ld, args.Mo, m0 + wave * 4, wave * 4, R, ln.eps, ln.out_scale, ln.p_drop, ln.seed, ln.seed_inc, lane);
    __syncthreads();
    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bf16* arow[2] = {aimg + (long long)p * ld + 8 * q, aimg + (long long)(16 + p) * ld + 8 * q};
    int step = 0;
    for (int k0 = wave * 32; k0 < R; k0 += 256, ++step) {
        Frag<bf16> a[2], b[2];
        const bool ok = (k0 + 8 * q + 8) <= R;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (ok) frag_load(a[i], arow[i] + k0); else frag_zero(a[i]);
            if (step < 2) b[i] = step == 0 ? bpre[0][i] : bpre[1][i];
            else if (ok) frag_load(b[i], brow[i] + k0);
            else frag_zero(b[i]);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = mma32(a[i], b[j], acc[i][j]);
    }
    __syncthreads();                                   // every wave has read its fragments: the image may become the partial-tile buffer
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wave][i * 16 + 4 * q + r][j * 16 + p] = acc[i][j][r];
    __syncthreads();
    // 1024 outputs: thread -> row, 2 consecutive columns; the wave partials are added in gemm_nt_splitk_kernel's order
    const int row = tid >> 4, c0 = (tid & 15) * 2;
    const int mo = m0 + row;
    if (mo >= args.Mo) return;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int no = n0 + c0 + c;
        if (no >= args.No) continue;
        float v = (red[0][row][c0 + c] + red[1][row][c0 + c]) + (red[2][row][c0 + c] + red[3][row][c0 + c]);
        v += (red[4][row][c0 + c] + red[5][row][c0 + c]) + (red[6][row][c0 + c] + red[7][row][c0 + c]);
        if (grp.bias) v += grp.bias[no];
        if (args.relu) v = fmaxf(v, 0.f);
        C[(long long)mo * args.ldc + no] = from_f32<bf16>(v);
    }
}

// ---------------------------------------------------------------------------------------------------------
// Tall GEMM for the pixel side (M = N*H*W rows, up to millions: the im2col convolutions of FCT and Encoder_32K):
//   C[mo][no] = sum_r A[mo][r] * B[no][r]  (+ bias, ReLU, accumulate), all fp32, exact-fp32 MFMA.
// A is streamed from HBM exactly once per column tile; B (the packed weights, <= a few MB) stays in L2.  A workgroup is four
// independent waves stacked in M (no LDS, no barrier); a wave owns 32 rows x NT*16 columns and walks K in steps of 32 with its
// fragments double-buffered in registers (the loads of step s+1 are issued before the MFMAs of step s).  The product is formed
// TRANSPOSED (weights as the first MFMA operand), so a lane ends up with 4 consecutive columns of one row: 16-byte stores.
// Workgroup -> tile map is XCD-aware: workgroups b, b+8, b+16, .. (same XCD, same L2) take the column tiles of ONE 128-row
// block, so the re-reads of A by the other column tiles hit that L2 instead of HBM.
// ---------------------------------------------------------------------------------------------------------
// IMPLICIT: A is not a matrix but an NHWC image x [n][H][W][Ci] (Ci a power of two >= 8) and row m is the output pixel (n, ho, wo) of a
// convolution: column r = tap * Ci + ci is gathered from pixel (ho*stride - pad + ky*dil, wo*stride - pad + kx*dil) -- zero outside
// the image and for r >= k*k*Ci -- so a lane's 8 consecutive r are 8 consecutive channels of one tap: still one 32-byte load.  The
// patch matrix is never written (im2col + its re-read were 1.2 GB of HBM traffic for a 64-channel 3x3 layer at 16 x 128^2 pixels).
struct ConvGather { int H, W, Ho, Wo, k, stride, pad, dil, log2ci, kk_ci, kdiv; };

template <int NT, bool IMPLICIT>
__global__ __launch_bounds__(256) void gemm_nt_tall_kernel(GemmArgs args, int tiles_n, int row_blocks, ConvGather cg) {
    const GemmGroup grp = args.g[0];
    const float* A = (const float*)grp.A;
    const float* B = (const float*)grp.B;
    float* C = (float*)grp.C;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, seq = bid >> 3;
    const int tn = seq % tiles_n, rb = (seq / tiles_n) * 8 + xcd;
    if (rb >= row_blocks) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int p = lane & 15, q = lane >> 4;
    const int m0 = rb * 128 + wave * 32, n0 = tn * (NT * 16);
    if (m0 >= args.Mo) return;
    const float* arow[2];
    const float* brow[NT];
    int ah[2], aw[2];                     // IMPLICIT: top-left input coordinate of the row's receptive field
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int r = m0 + i * 16 + p; if (r > args.Mo - 1) r = args.Mo - 1;
        if (IMPLICIT) {
            const int wo = r % cg.Wo, t = r / cg.Wo;
            const int ho = t % cg.Ho, n = t / cg.Ho;
            ah[i] = ho * cg.stride - cg.pad; aw[i] = wo * cg.stride - cg.pad;
            arow[i] = A + (((long long)n * cg.H + ah[i]) * cg.W + aw[i]) * (long long)(1 << cg.log2ci);
        } else {
            ah[i] = aw[i] = 0;
            arow[i] = A + (long long)r * args.lda + 8 * q;
        }
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        int c = n0 + j * 16 + p; if (c > args.No - 1) c = args.No - 1;
        brow[j] = B + (long long)c * args.ldb + 8 * q;
    }
    f32x4 acc[2][NT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int R = args.R;
    Frag<float> a[2][2], b[2][NT];
    auto load_step = [&](int buf, int k0) {
        const bool ok = (k0 + 8 * q + 8) <= R;
        if (IMPLICIT) {
            const int kc = k0 + 8 * q;
            const int tap = kc >> cg.log2ci, ci = kc & ((1 << cg.log2ci) - 1);
            const int ky = (tap * cg.kdiv) >> 16, kx = tap - ky * cg.k;
            const int dh = ky * cg.dil, dw = kx * cg.dil;
            const long long off = ((long long)dh * cg.W + dw) * (long long)(1 << cg.log2ci) + ci;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int hh = ah[i] + dh, ww = aw[i] + dw;
                if (kc < cg.kk_ci && (unsigned)hh < (unsigned)cg.H && (unsigned)ww < (unsigned)cg.W) frag_load(a[buf][i], arow[i] + off);
                else frag_zero(a[buf][i]);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i) { if (ok) frag_load(a[buf][i], arow[i] + k0); else frag_zero(a[buf][i]); }
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) { if (ok) frag_load(b[buf][j], brow[j] + k0); else frag_zero(b[buf][j]); }
    };
    auto mma_step = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = mma32(b[buf][j], a[buf][i], acc[i][j]);
    };
    load_step(0, 0);
    for (int k0 = 0; k0 < R; k0 += 64) {
        if (k0 + 32 < R) load_step(1, k0 + 32);
        mma_step(0);
        if (k0 + 32 < R) {
            if (k0 + 64 < R) load_step(0, k0 + 64);
            mma_step(1);
        }
    }
    const bool vec = (args.ldc & 3) == 0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int mo = m0 + i * 16 + p;
        if (mo >= args.Mo) continue;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int no = n0 + j * 16 + 4 * q;
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

#include "gemm_tall_lds.h"

// fp32 master weight W[N][K] -> T copy Wc[N][K] and T transpose Wt[K][N] (one 32x32 tile per block, grouped over blockIdx.z)
struct ConvertArgs {
    const float* W[18];
    void* Wc[18];
    void* Wt[18];
    int N[18], K[18];
    int ldt[18];             // row stride of the transposed copy (>= N; lets several transposes share one [K][sum N] buffer)
    int tile_begin[19];      // vector kernel: flat grid, matrix g owns tiles [tile_begin[g], tile_begin[g+1]) of 64 x 64 (no empty workgroups)
    int count;
};
template <typename T>
__global__ __launch_bounds__(256) void convert_weights_kernel(ConvertArgs a) {
    __shared__ float tile[32][33];
    const int g = blockIdx.z;
    const int N = a.N[g], K = a.K[g];
    const int n0 = blockIdx.y * 32, k0 = blockIdx.x * 32;
    if (n0 >= N || k0 >= K) return;
    const float* W = a.W[g];
    T* Wc = (T*)a.Wc[g];
    T* Wt = (T*)a.Wt[g];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const int n = n0 + r, k = k0 + tx;
        float v = 0.f;
        if (n < N && k < K) { v = W[(long long)n * K + k]; if (Wc) Wc[(long long)n * K + k] = from_f32<T>(v); }
        tile[r][tx] = v;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int k = k0 + r, n = n0 + tx;
        if (n < N && k < K) Wt[(long long)k * a.ldt[g] + n] = from_f32<T>(tile[tx][r]);
    }
}

// Same conversion on 64x64 tiles with 16-byte loads and 4-element stores in both orientations (N, K, ldt multiples of 4)
template <typename T>
__global__ __launch_bounds__(256) void convert_weights_vec_kernel(ConvertArgs a) {
    __shared__ float tile[64][65];
    int g = 0;
#pragma unroll
    for (int i = 1; i < 18; ++i)
        if (i < a.count && (int)blockIdx.x >= a.tile_begin[i]) g = i;
    const int N = a.N[g], K = a.K[g];
    const int local = (int)blockIdx.x - a.tile_begin[g], tk = (K + 63) / 64;
    const int n0 = (local / tk) * 64, k0 = (local % tk) * 64;
    const float* W = a.W[g];
    T* Wc = (T*)a.Wc[g];
    T* Wt = (T*)a.Wt[g];
    const int c4 = (threadIdx.x & 15) * 4, r0 = threadIdx.x >> 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + 16 * i, n = n0 + r, k = k0 + c4;
        f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
        if (n < N && k < K) {                                   // K % 4 == 0: the whole quad is inside
            v = *reinterpret_cast<const f32x4*>(W + (long long)n * K + k);
            T o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = from_f32<T>(v[j]);
            if (!Wc) {}                                         // (fp32 storage: the forward reads the master weights in place)
            else if (sizeof(T) == 2) *reinterpret_cast<unsigned long long*>(Wc + (long long)n * K + k) = *reinterpret_cast<const unsigned long long*>(o);
            else *reinterpret_cast<f32x4*>(Wc + (long long)n * K + k) = *reinterpret_cast<const f32x4*>(o);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) tile[r][c4 + j] = v[j];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + 16 * i, k = k0 + r, n = n0 + c4;      // row r of the transposed tile = column r of the source tile
        if (k < K && n < N) {
            T o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = from_f32<T>(tile[c4 + j][r]);
            T* dst = Wt + (long long)k * a.ldt[g] + n;
            if (sizeof(T) == 2) *reinterpret_cast<unsigned long long*>(dst) = *reinterpret_cast<const unsigned long long*>(o);
            else *reinterpret_cast<f32x4*>(dst) = *reinterpret_cast<const f32x4*>(o);
        }
    }
}

// dym = dy * (y > 0)
template <typename T>
__global__ void relu_mask_kernel(const T* __restrict__ dy, const T* __restrict__ y, T* __restrict__ out, long long n8) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long long)gridDim.x * blockDim.x) {
        Vec8<T> a, b, o;
        a.load(dy + i * 8);
        b.load(y + i * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) o.set(j, b.get(j) > 0.f ? a.get(j) : 0.f);
        o.store(out + i * 8);
    }
}

// db[n] = sum_m dy[m][n]: 32 columns x 8 row groups per block
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ dy, float* __restrict__ db, int M, int N) {
    __shared__ float red[8][33];
    const int col = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const int n = blockIdx.x * 32 + col;
    float s = 0.f;
    if (n < N)
        for (int m = grp; m < M; m += 8) s += to_f32<T>(dy[(long long)m * N + n]);
    red[grp][col] = s;
    __syncthreads();
    if (grp == 0 && n < N) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += red[k][col];
        db[n] = t;
    }
}

template <typename T>
int linear_fwd_t(const void* x, int ldx, const float* W, const float* b, void* y, int M, int N, int K, int relu, hipStream_t st) {
    {                                                       // few-tile products: eight waves split K, fragments straight from the master weights
        const int rc = hyb_gemm_skinny_wf32(sizeof(T) == 2 ? HYB_BF16 : HYB_F32, x, W, y, b, M, N, K, ldx, K, N, relu, 0, 0, st);
        if (rc != -100) return rc;
    }
    GemmArgs a{};
    a.g[0] = GemmGroup{x, W, y, b, nullptr, nullptr};
    a.Mo = M; a.No = N; a.R = K; a.lda = ldx; a.ldb = K; a.ldc = N; a.relu = relu; a.accumulate = 0;
    return launch_gemm<T, T, float, T, false, false>(a, 1, st);
}

template <typename T>
int linear_bwd_t(const void* x, int ldx, const float* W, const void* Wt, const void* y, const void* dy, void* dx, int accumulate_dx, float* dW,
                 float* db, int M, int N, int K, int relu, void* ws, size_t ws_bytes, hipStream_t st) {
    const T* dym = (const T*)dy;
    if (relu) {
        if (!y || !ws || ws_bytes < (size_t)M * N * sizeof(T)) return HYB_E_WORKSPACE;
        const long long n8 = (long long)M * N / 8;
        int blocks = hyb_cdiv(n8, 256);
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(relu_mask_kernel<T>, dim3(blocks), dim3(256), 0, st, (const T*)dy, (const T*)y, (T*)ws, n8);
        HYB_LAUNCH_CHECK();
        dym = (const T*)ws;
    }
    if (dx && Wt) {  // dx[m][k] = sum_n dym[m][n] * Wt[k][n]  (pre-transposed T weights: skinny NT GEMM, no LDS staging)
        const void* A_[1] = {dym};
        const void* B_[1] = {Wt};
        void* C_[1] = {dx};
        int rc = hyb_gemm_nt(sizeof(T) == 4 ? HYB_F32 : HYB_BF16, 1, A_, B_, C_, nullptr, 0, M, K, N, N, N, ldx, 0, accumulate_dx, st);
        if (rc) return rc;
    } else if (dx) { // dx[m][k] = sum_n dym[m][n] * W[n][k]
        int rc = hyb_gemm_skinny_wf32(sizeof(T) == 2 ? HYB_BF16 : HYB_F32, dym, W, dx, nullptr, M, K, N, N, K, ldx, 0, accumulate_dx, 1, st);
        if (rc != -100 && rc != 0) return rc;
        if (rc == -100) {            // not a few-tile shape: the LDS-staged kernel
        GemmArgs a{};
        a.g[0] = GemmGroup{dym, W, dx, nullptr, nullptr, nullptr};
        a.Mo = M; a.No = K; a.R = N; a.lda = N; a.ldb = K; a.ldc = ldx; a.relu = 0; a.accumulate = accumulate_dx;
        rc = launch_gemm<T, T, float, T, false, true>(a, 1, st);
        if (rc) return rc;
        }
    }
    if (dW && N % 8 == 0 && K % 8 == 0) {
        // the multi-matrix weight-gradient kernel with one matrix -- the model-level path computes the same gradient as one group of
        // its encoder launch, and the two must agree bit for bit
        const void* dy_[1] = {dym}; const void* x_[1] = {x}; float* dW_[1] = {dW}; float* db_[1] = {db};
        const int N_[1] = {N}, K_[1] = {K}, lddy_[1] = {N}, ldx_[1] = {ldx};
        int rc = hyb_linear_dw_multi(sizeof(T) == 2 ? HYB_BF16 : HYB_F32, 1, dy_, nullptr, x_, dW_, db_, N_, K_, lddy_, ldx_, M, st, 0, nullptr);
        if (rc) return rc;
    } else if (dW) {        // dW[n][k] = sum_m dym[m][n] * x[m][k]; the bias gradient (column sums of dym) rides in the same launch
        GemmArgs a{};
        a.g[0] = GemmGroup{dym, x, dW, nullptr, nullptr, db};
        a.Mo = N; a.No = K; a.R = M; a.lda = N; a.ldb = ldx; a.ldc = K; a.relu = 0; a.accumulate = 0;
        int rc = launch_gemm<T, T, T, float, true, true>(a, 1, st);
        if (rc) return rc;
    } else if (db) {
        hipLaunchKernelGGL(colsum_kernel<T>, dim3(hyb_cdiv(N, 32)), dim3(256), 0, st, dym, db, M, N);
        HYB_LAUNCH_CHECK();
    }
    return 0;
}

}  // namespace

// Internal: linear backward with pre-transposed T weights for the dx product
int hyb_linear_bwd_wt(int dtype, const void* x, int ldx, const float* W, const void* Wt, const void* y, const void* dy, void* dx, int accumulate_dx,
                      float* dW, float* db, int M, int N, int K, int relu, void* ws, size_t ws_bytes, hipStream_t st) {
    if (dtype == HYB_F32) return linear_bwd_t<float>(x, ldx, W, Wt, y, dy, dx, accumulate_dx, dW, db, M, N, K, relu, ws, ws_bytes, st);
    if (dtype == HYB_BF16) return linear_bwd_t<bf16>(x, ldx, W, Wt, y, dy, dx, accumulate_dx, dW, db, M, N, K, relu, ws, ws_bytes, st);
    return HYB_E_ARG;
}

// Internal (same shared object): skinny NT GEMM on pre-converted T operands, up to 3 groups.
int hyb_gemm_nt(int dtype, int groups, const void* const* A, const void* const* B, void* const* C, const float* const* bias, int out_f32,
                int Mo, int No, int R, int lda, int ldb, int ldc, int relu, int accumulate, hipStream_t st, const void* const* Amask,
                const void* const* Cmask) {
    if (groups < 1 || groups > 3 || R % 8 != 0 || lda % 8 != 0 || ldb % 8 != 0) return HYB_E_ARG;
    if (Cmask && out_f32) return HYB_E_ARG;              // (the output mask has the storage type; it lives in the skinny kernel only)
    GemmArgs a{};
    for (int i = 0; i < groups; ++i)
        a.g[i] = GemmGroup{A[i], B[i], C[i], bias ? bias[i] : nullptr, Amask ? Amask[i] : nullptr, nullptr, Cmask ? Cmask[i] : nullptr};
    a.Mo = Mo; a.No = No; a.R = R; a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.relu = relu; a.accumulate = accumulate;
    dim3 grid(hyb_cdiv(No, 32), hyb_cdiv(Mo, 32), groups);
    // few tiles (M = 128, N = 512: 64 workgroups on 256 CUs): eight waves split K to shorten the per-wave load/MFMA chain
    static const int w8env = getenv("HYB_GEMM_W8") ? atoi(getenv("HYB_GEMM_W8")) : 1;
    const bool w8 = w8env && (long long)grid.x * grid.y * grid.z <= 256 && R >= 256;
    static const int tall_env = getenv("HYB_GEMM_TALL") ? atoi(getenv("HYB_GEMM_TALL")) : 1;
    if (dtype == HYB_F32 && tall_env && groups == 1 && !Amask && !Cmask && Mo >= 2048) {
        // pixel-side GEMMs (M = N*H*W): one pass over A per column tile, four independent waves per workgroup
        const int row_blocks = hyb_cdiv(Mo, 128);
        const int nt = No > 32 ? 4 : No > 16 ? 2 : 1;
        const int tiles_n = hyb_cdiv(No, nt * 16);
        const long long blocks = (long long)hyb_cdiv(row_blocks, 8) * tiles_n * 8;
        if (blocks > 0x7fffffff) return HYB_E_ARG;
        const ConvGather none{};
        HybProfileHook* hook = hyb_find_hook(4, No, R);
        if (hook) hipEventRecord(hook->ev0, st);
        if (gt_lds_ok(a, false)) { const int rc = gt_launch<false>(a, none, st); if (rc) return rc; }
        else if (nt == 4) hipLaunchKernelGGL((gemm_nt_tall_kernel<4, false>), dim3((unsigned)blocks), dim3(256), 0, st, a, tiles_n, row_blocks, none);
        else if (nt == 2) hipLaunchKernelGGL((gemm_nt_tall_kernel<2, false>), dim3((unsigned)blocks), dim3(256), 0, st, a, tiles_n, row_blocks, none);
        else hipLaunchKernelGGL((gemm_nt_tall_kernel<1, false>), dim3((unsigned)blocks), dim3(256), 0, st, a, tiles_n, row_blocks, none);
        if (hook) hipEventRecord(hook->ev1, st);
    } else if (dtype == HYB_F32 && w8) hipLaunchKernelGGL((gemm_nt_splitk_kernel<float, float, 8>), grid, dim3(512), 0, st, a);      // (the temporal part of 'mixed' / 'bf16x3' / 'fp32')
    else if (dtype == HYB_F32) hipLaunchKernelGGL((gemm_nt_splitk_kernel<float, float, 4>), grid, dim3(256), 0, st, a);
    else if (dtype == HYB_BF16 && out_f32) hipLaunchKernelGGL((gemm_nt_splitk_kernel<bf16, float, 4>), grid, dim3(256), 0, st, a);
    else if (dtype == HYB_BF16 && w8) hipLaunchKernelGGL((gemm_nt_splitk_kernel<bf16, bf16, 8>), grid, dim3(512), 0, st, a);
    else if (dtype == HYB_BF16) hipLaunchKernelGGL((gemm_nt_splitk_kernel<bf16, bf16, 4>), grid, dim3(256), 0, st, a);
    else return HYB_E_ARG;
    HYB_LAUNCH_CHECK();
    return 0;
}

// Internal: hyb_gemm_nt with LayerNorm + residual as the prologue (gemm_nt_ln_kernel): A = dropout((LN(x) * gamma + beta + skip) * out_scale),
// also written to y (+ stats [2][Mo]) by the column-0 workgroups.  Returns -100 when the shape is not taken (the caller then runs the two
// launches): bf16 only, few-tile grids (every column tile re-forms its rows), R = D a multiple of 8 up to 1024 with the image in 64 KB of LDS.
int hyb_gemm_nt_ln(int dtype, int groups, const void* x, const void* skip, const float* gamma, const float* beta, void* y, float* stats, float eps,
                   float out_scale, float p_drop, unsigned long long seed, const unsigned long long* seed_inc, const void* const* B, void* const* C,
                   const float* const* bias, int Mo, int No, int R, int ldb, int ldc, int relu, hipStream_t st) {
    static const int env = getenv("HYB_GEMM_LN") ? atoi(getenv("HYB_GEMM_LN")) : 1;          // (=0: A/B, LayerNorm as its own launch)
    const dim3 grid(hyb_cdiv(No, 32), hyb_cdiv(Mo, 32), groups);
    const size_t img = (size_t)32 * (R + LNG_PAD) * sizeof(bf16), red = (size_t)8 * 32 * 33 * sizeof(float);
    const size_t lds = img > red ? img : red;
    if (!env || dtype != HYB_BF16 || groups < 1 || groups > 3 || (long long)grid.x * grid.y * grid.z > 256 || R < 256 || R > 1024 || R % 8 != 0 ||
        ldb % 8 != 0 || lds > 64 * 1024)
        return -100;
    GemmArgs a{};
    for (int i = 0; i < groups; ++i) a.g[i] = GemmGroup{nullptr, B[i], C[i], bias ? bias[i] : nullptr, nullptr, nullptr, nullptr};
    a.Mo = Mo; a.No = No; a.R = R; a.lda = R; a.ldb = ldb; a.ldc = ldc; a.relu = relu; a.accumulate = 0;
    const LnPro ln{x, skip, gamma, beta, y, stats, eps, out_scale, p_drop, seed, seed_inc};
    if (R <= 512) hipLaunchKernelGGL(gemm_nt_ln_kernel<1>, grid, dim3(512), lds, st, a, ln);
    else hipLaunchKernelGGL(gemm_nt_ln_kernel<2>, grid, dim3(512), lds, st, a, ln);
    HYB_LAUNCH_CHECK();
    return 0;
}

// Internal: skinny product on the fp32 master weights (BMODE 1 / 2 of gemm_nt_splitk_kernel), bf16 activations; returns -100 when the
// shape is not a few-tile one (the caller then takes the LDS-staged kernel)
int hyb_gemm_skinny_wf32(int dtype, const void* A, const float* Bf, void* C, const float* bias, int Mo, int No, int R, int lda, int ldb, int ldc, int relu,
                         int accumulate, int transposed_b, hipStream_t st) {
    static const int env = getenv("HYB_GEMM_WF32") ? atoi(getenv("HYB_GEMM_WF32")) : 1;
    const dim3 grid(hyb_cdiv(No, 32), hyb_cdiv(Mo, 32), 1);
    if (!env || (long long)grid.x * grid.y > 256 || R < 256 || R % 32 != 0 || lda % 8 != 0 || ((uintptr_t)A % 16) != 0) return -100;
    if (!transposed_b && (ldb % 4 != 0 || ((uintptr_t)Bf % 16) != 0)) return -100;
    GemmArgs a{};
    a.g[0] = GemmGroup{A, Bf, C, bias, nullptr, nullptr};
    a.Mo = Mo; a.No = No; a.R = R; a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.relu = relu; a.accumulate = accumulate;
    if (dtype == HYB_F32) {        // fp32 storage ('fp32' / 'bf16x3' / the temporal part of 'mixed'): the weights are read as they are
        if (transposed_b) hipLaunchKernelGGL((gemm_nt_splitk_kernel<float, float, 8, 2>), grid, dim3(512), 0, st, a);
        else hipLaunchKernelGGL((gemm_nt_splitk_kernel<float, float, 8, 0>), grid, dim3(512), 0, st, a);
    } else if (transposed_b) hipLaunchKernelGGL((gemm_nt_splitk_kernel<bf16, bf16, 8, 2>), grid, dim3(512), 0, st, a);
    else hipLaunchKernelGGL((gemm_nt_splitk_kernel<bf16, bf16, 8, 1>), grid, dim3(512), 0, st, a);
    HYB_LAUNCH_CHECK();
    return 0;
}

// Internal: convolution as an implicit GEMM (no patch matrix): y[m][co] = sum_r patch(x)[m][r] * wp[co][r] (+ bias, ReLU), m = (n, ho, wo).
// x [n][H][W][Ci] fp32 with Ci a power of two >= 8; wp [Co][Kp] packed with r = (ky*k + kx)*Ci + ci (Kp % 8 == 0); y row stride ldy.
// hyb_conv_implicit_ok is the callers' decision rule (power-of-two channels, enough rows to fill the chip); the launcher itself only needs the
// former.
bool hyb_conv_implicit_ok(int Ci, long long rows) { return Ci >= 8 && (Ci & (Ci - 1)) == 0 && rows >= 2048; }
int hyb_conv_implicit_gemm(const float* x, const float* wp, const float* bias, float* y, int n_img, int H, int W, int Ci, int Ho, int Wo, int Co,
                           int Kp, int k, int stride, int pad, int dil, int ldy, int relu, hipStream_t st) {
    const long long rows = (long long)n_img * Ho * Wo;
    if (Ci < 8 || (Ci & (Ci - 1)) != 0 || rows < 1 || rows > 0x7fffffff / 32 * 32 || (long long)n_img * H * W > 0x7fffffff / 32 * 32 || Kp % 8 != 0 || k < 1 || k > 7)
        return HYB_E_ARG;
    GemmArgs a{};
    a.g[0] = GemmGroup{x, wp, y, bias, nullptr, nullptr};
    a.Mo = (int)rows; a.No = Co; a.R = Kp; a.lda = 0; a.ldb = Kp; a.ldc = ldy; a.relu = relu; a.accumulate = 0;
    int l2 = 0; while ((1 << l2) < Ci) ++l2;
    const ConvGather cg{H, W, Ho, Wo, k, stride, pad, dil, l2, k * k * Ci, (65536 + k - 1) / k};
    const int row_blocks = hyb_cdiv(rows, 128);
    const int nt = Co > 32 ? 4 : Co > 16 ? 2 : 1;
    const int tiles_n = hyb_cdiv(Co, nt * 16);
    const long long blocks = (long long)hyb_cdiv(row_blocks, 8) * tiles_n * 8;
    if (blocks > 0x7fffffff) return HYB_E_ARG;
    HybProfileHook* hook = hyb_find_hook(4, Co, Kp);          // measurement hook (hyb_profile_set): kernel 4 = the tall GEMM, keyed by (columns, K)
    if (hook) hipEventRecord(hook->ev0, st);
    // (the LDS kernel addresses its 128 rows relative to their first image with 32-bit byte offsets: two images must fit 2^31 bytes)
    if (Ci >= 4 && (long long)H * W * Ci * 8 < 0x7fffffffll && gt_lds_ok(a, true)) { const int rc = gt_launch<true>(a, cg, st); if (rc) return rc; }
    else if (nt == 4) hipLaunchKernelGGL((gemm_nt_tall_kernel<4, true>), dim3((unsigned)blocks), dim3(256), 0, st, a, tiles_n, row_blocks, cg);
    else if (nt == 2) hipLaunchKernelGGL((gemm_nt_tall_kernel<2, true>), dim3((unsigned)blocks), dim3(256), 0, st, a, tiles_n, row_blocks, cg);
    else hipLaunchKernelGGL((gemm_nt_tall_kernel<1, true>), dim3((unsigned)blocks), dim3(256), 0, st, a, tiles_n, row_blocks, cg);
    if (hook) hipEventRecord(hook->ev1, st);
    HYB_LAUNCH_CHECK();
    return 0;
}

// Internal: weight + bias gradients of up to 3 Linear layers that share the input x, one launch:
//   dW_g[n][k] = sum_m dym_g[m][n] * x[m][k],  db_g[n] = sum_m dym_g[m][n],  dym_g = dy_g * (mask_g > 0) (mask optional)
int hyb_linear_dw_grouped(int dtype, int groups, const void* const* dy, const void* const* mask, const void* x, float* const* dW,
                          float* const* db, int M, int N, int K, int lddy, int ldx, hipStream_t st) {
    if (groups < 1 || groups > 3) return HYB_E_ARG;
    GemmArgs a{};
    for (int i = 0; i < groups; ++i) a.g[i] = GemmGroup{dy[i], x, dW[i], nullptr, mask ? mask[i] : nullptr, db ? db[i] : nullptr};
    a.Mo = N; a.No = K; a.R = M; a.lda = lddy; a.ldb = ldx; a.ldc = K; a.relu = 0; a.accumulate = 0;
    if (dtype == HYB_F32) return launch_gemm<float, float, float, float, true, true>(a, groups, st);
    if (dtype == HYB_BF16) return launch_gemm<bf16, bf16, bf16, float, true, true>(a, groups, st);
    return HYB_E_ARG;
}

// Internal: weight + bias gradients of up to 6 Linear layers of different shapes in one launch (see gemm_dw_multi_kernel).
int hyb_linear_dw_multi(int dtype, int groups, const void* const* dy, const void* const* mask, const void* const* x, float* const* dW,
                        float* const* db, const int* N, const int* K, const int* lddy, const int* ldx, int M, hipStream_t st,
                        int nriders, const HybDwRider* riders) {
    if (groups < 1 || groups > DW_MAX_GROUPS || M < 1 || nriders < 0 || nriders > DW_MAX_RIDERS || (nriders > 0 && !riders)) return HYB_E_ARG;
    DwArgs a{};
    int tiles = 0;
    for (int i = 0; i < groups; ++i) {
        if (N[i] % 8 != 0 || K[i] % 8 != 0) return HYB_E_ARG;
        const int tx = hyb_cdiv(K[i], 64), ty = hyb_cdiv(N[i], 64);
        a.g[i] = DwGroup{dy[i], mask ? mask[i] : nullptr, x[i], dW[i], db ? db[i] : nullptr, N[i], K[i], lddy[i], ldx[i], tx, tiles};
        tiles += tx * ty;
    }
    a.ngroups = groups; a.M = M; a.tiles = tiles;
    int blocks = tiles;
    for (int r = 0; r < nriders; ++r) {
        if (!riders[r].part || riders[r].rows < 1 || riders[r].n < 1 || !riders[r].out0 || (riders[r].split < riders[r].n && !riders[r].out1)) return HYB_E_ARG;
        a.rd[r] = riders[r];
        a.rider_begin[r] = blocks;
        blocks += hyb_cdiv(riders[r].n, 256);
    }
    a.nriders = nriders;
    a.rider_begin[nriders] = blocks;
    static const int tr_env = getenv("HYB_DW_TR") ? atoi(getenv("HYB_DW_TR")) : 1;      // (=0: A/B, the transposing-store form)
    bool aligned = true;                                   // 16-byte rows and bases for the straight vector staging
    for (int i = 0; i < groups; ++i)
        aligned = aligned && lddy[i] % 8 == 0 && ldx[i] % 8 == 0 && ((uintptr_t)dy[i] % 16 == 0) && ((uintptr_t)x[i] % 16 == 0) &&
                  (!mask || !mask[i] || (uintptr_t)mask[i] % 16 == 0);
    // (an fp32-storage twin of the untransposed-staging kernel -- eight ds_read_b32 per fragment instead of the transposing 16-bit read --
    // was built and measured: 28.3 us against the 25.5 us of the generic kernel below, profiles/r04_ab_mixed_f32_twins.txt; not kept)
    if (dtype == HYB_F32) hipLaunchKernelGGL(gemm_dw_multi_kernel<float>, dim3(blocks), dim3(256), 0, st, a);
    else if (dtype == HYB_BF16 && tr_env && aligned) hipLaunchKernelGGL(gemm_dw_multi_tr_kernel, dim3(blocks), dim3(256), 0, st, a);
    else if (dtype == HYB_BF16) hipLaunchKernelGGL(gemm_dw_multi_kernel<bf16>, dim3(blocks), dim3(256), 0, st, a);
    else return HYB_E_ARG;
    HYB_LAUNCH_CHECK();
    return 0;
}

// Internal: convert up to 8 fp32 weight matrices to T (plain + transposed copies) in one launch.
int hyb_convert_weights(int dtype, int count, const float* const* W, void* const* Wc, void* const* Wt, const int* N, const int* K,
                        const int* ldt, hipStream_t st) {
    if (count < 1 || count > 18) return HYB_E_ARG;      // up to three encoder layers in one launch
    ConvertArgs a{};
    int maxN = 0, maxK = 0;
    for (int i = 0; i < count; ++i) {
        a.W[i] = W[i]; a.Wc[i] = Wc[i]; a.Wt[i] = Wt[i]; a.N[i] = N[i]; a.K[i] = K[i]; a.ldt[i] = ldt ? ldt[i] : N[i];
        if (N[i] > maxN) maxN = N[i];
        if (K[i] > maxK) maxK = K[i];
    }
    dim3 grid(hyb_cdiv(maxK, 32), hyb_cdiv(maxN, 32), count);
    bool vec = true;
    for (int i = 0; i < count; ++i)
        vec = vec && N[i] % 4 == 0 && K[i] % 4 == 0 && a.ldt[i] % 4 == 0 && ((uintptr_t)W[i] % 16 == 0) && ((uintptr_t)Wc[i] % 16 == 0) &&
              ((uintptr_t)Wt[i] % 16 == 0);
    if (vec) {
        int tiles = 0;
        for (int i = 0; i < count; ++i) { a.tile_begin[i] = tiles; tiles += hyb_cdiv(N[i], 64) * hyb_cdiv(K[i], 64); }
        a.tile_begin[count] = tiles; a.count = count;
        dim3 gridv(tiles);
        if (dtype == HYB_F32) hipLaunchKernelGGL(convert_weights_vec_kernel<float>, gridv, dim3(256), 0, st, a);
        else if (dtype == HYB_BF16) hipLaunchKernelGGL(convert_weights_vec_kernel<bf16>, gridv, dim3(256), 0, st, a);
        else return HYB_E_ARG;
        HYB_LAUNCH_CHECK();
        return 0;
    }
    if (dtype == HYB_F32) hipLaunchKernelGGL(convert_weights_kernel<float>, grid, dim3(256), 0, st, a);
    else if (dtype == HYB_BF16) hipLaunchKernelGGL(convert_weights_kernel<bf16>, grid, dim3(256), 0, st, a);
    else return HYB_E_ARG;
    HYB_LAUNCH_CHECK();
    return 0;
}

// Internal: dym = dy * (y > 0) and db[n] = sum_m dym[m][n] helpers for the encoder backward
int hyb_relu_mask(int dtype, const void* dy, const void* y, void* out, long long n, hipStream_t st) {
    const long long n8 = n / 8;
    int blocks = hyb_cdiv(n8, 256);
    if (blocks > 2048) blocks = 2048;
    if (dtype == HYB_F32) hipLaunchKernelGGL(relu_mask_kernel<float>, dim3(blocks), dim3(256), 0, st, (const float*)dy, (const float*)y, (float*)out, n8);
    else if (dtype == HYB_BF16) hipLaunchKernelGGL(relu_mask_kernel<bf16>, dim3(blocks), dim3(256), 0, st, (const bf16*)dy, (const bf16*)y, (bf16*)out, n8);
    else return HYB_E_ARG;
    HYB_LAUNCH_CHECK();
    return 0;
}

// Internal (same shared object): grouped forward used by the encoder to run Q, K, V in one launch.
int hyb_linear_fwd_grouped3(int dtype, const void* const* x, const float* const* W, const float* const* b, void* const* y, int groups,
                            int M, int N, int K, int relu, hipStream_t st) {
    GemmArgs a{};
    for (int i = 0; i < groups; ++i) a.g[i] = GemmGroup{x[i], W[i], y[i], b[i], nullptr, nullptr};
    a.Mo = M; a.No = N; a.R = K; a.lda = K; a.ldb = K; a.ldc = N; a.relu = relu; a.accumulate = 0;
    if (dtype == HYB_F32) return launch_gemm<float, float, float, float, false, false>(a, groups, st);
    if (dtype == HYB_BF16) return launch_gemm<bf16, bf16, float, bf16, false, false>(a, groups, st);
    return HYB_E_ARG;
}

extern "C" int hyb_linear_fwd(int dtype, const void* x, int ldx, const float* W, const float* b, void* y, int M, int N, int K, int relu,
                              void* stream) {
    HYB_CHECK_ARG(x && W && y && M > 0 && N > 0 && K > 0 && K % 8 == 0 && N % 8 == 0 && ldx % 8 == 0 && ldx >= K);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == HYB_F32) return linear_fwd_t<float>(x, ldx, W, b, y, M, N, K, relu, st);
    if (dtype == HYB_BF16) return linear_fwd_t<bf16>(x, ldx, W, b, y, M, N, K, relu, st);
    return HYB_E_ARG;
}

extern "C" int hyb_linear_bwd(int dtype, const void* x, int ldx, const float* W, const void* y, const void* dy, void* dx, int accumulate_dx,
                              float* dW, float* db, int M, int N, int K, int relu, void* workspace, size_t workspace_bytes, void* stream) {
    HYB_CHECK_ARG(W && dy && M > 0 && N > 0 && K > 0 && K % 8 == 0 && N % 8 == 0 && ldx % 8 == 0 && ldx >= K);
    HYB_CHECK_ARG(!dW || x);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == HYB_F32) return linear_bwd_t<float>(x, ldx, W, nullptr, y, dy, dx, accumulate_dx, dW, db, M, N, K, relu, workspace, workspace_bytes, st);
    if (dtype == HYB_BF16) return linear_bwd_t<bf16>(x, ldx, W, nullptr, y, dy, dx, accumulate_dx, dW, db, M, N, K, relu, workspace, workspace_bytes, st);
    return HYB_E_ARG;
}
