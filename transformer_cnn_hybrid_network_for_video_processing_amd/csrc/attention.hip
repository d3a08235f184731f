// Temporal attention core of MultiheadAttention.attention (TransformerEncoder.pyc src L49-62) with the head
// split/merge of __reshape_to_batches__/__reshape_from_batches__ (src L22-45) done as index math.
//
// Second generation (round 2).  One wave owns one 16-query tile of one (clip, head) problem and keeps the whole softmax in
// registers; a workgroup is ceil(S/16) such waves (ONE wave at T = 16).  Nothing is padded to 32 tokens, no score or
// probability ever touches LDS or HBM, and there is no transposing scatter:
//
//   forward   S^T = K Q^T  (16x16x32 MFMA, both fragments loaded straight from HBM/L2, 16 B per lane).  The accumulator of the
//             swapped product holds  S^T[key = 4g + r][query = p]  (p = lane & 15, g = lane >> 4): a query's keys sit in 4
//             registers x 4 lanes, so the row max / sum are 3 register ops + 2 cross-lane steps, and the SAME registers,
//             converted to T, are the B operand (k = key) of the 16x16x16 MFMA computing  O^T = V^T P^T  -- the A operand
//             V^T[d][key] is a transposed LDS read (ds_read_b64_tr_b16) of the row-major V image.  O^T's accumulator holds 4
//             consecutive features of one query: an 8-byte store.
//             Saved for backward: the row max and row sum (2 floats per query) instead of the fp32 probabilities.
//   backward  phase A (wave = query tile): P^T and dP^T = V dO^T recomputed in the swapped orientation -> delta, dS^T ->
//             dQ^T = K^T dS^T (K^T by transposed LDS reads).  phase B (wave = KEY tile, all query tiles): P and dP = dO V^T in
//             the normal orientation (accumulator = [query = 4g + r][key = p]) are the B operands (k = query) of
//             dK^T = Q^T dS and dV^T = dO^T (P * dropout).  Each wave ends with complete dK / dV tiles: no cross-wave reduction.
//
// LDS images are row-major with a row stride that is an odd multiple of 32 bytes, which makes every transposed read
// conflict-free (8 rows x 32 B of a 32-lane half tile the 64 banks exactly).
// Reference quirks kept: scale 1/sqrt(d_model) (Q1), mask row b*H+h reads mask[(b*H+h) % B] (Q4), masked_fill(-1e9),
// dropout on the weights (src L58).  fp32 mode runs the same structure on v_mfma_f32_16x16x4_f32.
#include <stdlib.h>
#include "hyb_common.h"

namespace {

struct AttnDims {
    int B, S, D, H, dh;
    int nt;      // 16-token tiles per sequence = waves per workgroup
    int DT;      // 16-feature tiles per head
    int ldi;     // LDS image row stride in elements
    int ldp;     // backward, several tiles: row stride of the P / dS images [query][key]
    int ld_qkv;  // token row stride (elements) of q, k, v and dq, dk, dv (D, or 3D when they are packed as [M][3D])
    int ld_o;    // token row stride of out / dout
    int ppw;     // problems per workgroup: > 1 only for single-tile sequences (S <= 16), where every wave takes its own (clip, head)
    int relu_out; // backward: dq, dk, dv are the gradients of ReLU outputs (the reference's projections end in a ReLU, src L69-70): zero them
                  // where q, k, v are not positive, so that the consumers of the packed gradient need no mask operand
};

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4_a;

// ---- a 16-deep K step (the k = token contractions) -----------------------------------------------------------------
// One lane's share: 4 consecutive K elements of one row (A) / column (B); lane l holds row/col (l & 15), K = 4*(l >> 4) + j.
//   bf16: the operand of v_mfma_f32_16x16x16_bf16.   fp32: four v_mfma_f32_16x16x4_f32, MFMA j contracting K = {4g + j}.
// An accumulator tile (rows 4g + r, column p) is therefore directly the B operand whose K index is the accumulator's row.
template <typename T> struct Frag16;
template <> struct Frag16<bf16> { bf16x4 v; };
template <> struct Frag16<float> { float v[4]; };
__device__ __forceinline__ f32x4 mma16(const Frag16<bf16>& a, const Frag16<bf16>& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, a.v), __builtin_bit_cast(s16x4, b.v), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mma16(const Frag16<float>& a, const Frag16<float>& b, f32x4 c) {
#pragma unroll
    for (int j = 0; j < 4; ++j) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v[j], b.v[j], c, 0, 0, 0);
    return c;
}
__device__ __forceinline__ void acc_to_frag(Frag16<bf16>& f, const float (&x)[4]) {
    f.v[0] = (bf16)x[0]; f.v[1] = (bf16)x[1]; f.v[2] = (bf16)x[2]; f.v[3] = (bf16)x[3];
}
__device__ __forceinline__ void acc_to_frag(Frag16<float>& f, const float (&x)[4]) {
    f.v[0] = x[0]; f.v[1] = x[1]; f.v[2] = x[2]; f.v[3] = x[3];
}
// A operand  M^T[col c0 + p][rows r0 + 4g + j]  of a row-major LDS image img[row][ld]: 4 rows x 16 columns per 16-lane group,
// delivered column-major by the hardware transpose read (every lane active, every address inside the image).
__device__ __forceinline__ void tr_read(Frag16<bf16>& f, const bf16* img, int ld, int r0, int c0, int lane) {
    const int g = lane >> 4, qq = (lane & 15) >> 2, pp = lane & 3;
    f.v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_a*)(img + (r0 + 4 * g + qq) * ld + c0 + 4 * pp));
}
__device__ __forceinline__ void tr_read(Frag16<float>& f, const float* img, int ld, int r0, int c0, int lane) {
    const int g = lane >> 4, p = lane & 15;
#pragma unroll
    for (int j = 0; j < 4; ++j) f.v[j] = img[(r0 + 4 * g + j) * ld + c0 + p];
}

// rows [16*nt][ld] of an LDS image <- src[(b*S + s)*gld + h*dh + c]; rows >= S and columns >= dh are zero
template <typename T>
__device__ __forceinline__ void stage_image(T* img, const T* __restrict__ src, const AttnDims& d, int b, int h, int gld, int tid, int nthreads) {
    const int segs = d.DT * 2;                       // 8-element chunks per row
    for (int u = tid; u < d.nt * 16 * segs; u += nthreads) {
        const int s = u / segs, c = (u - s * segs) * 8;
        Vec8<T> v;
        if (s < d.S && c < d.dh) v.load(src + ((long long)(b * d.S + s)) * gld + h * d.dh + c);
        else v.zero();
        v.store(img + s * d.ldi + c);
    }
}

// k32 fragment of token row `tok` (clamped; rows >= S are masked later), features f0 .. f0+7; zero past the head width
template <typename T>
__device__ __forceinline__ void row_frag(Frag<T>& f, const T* __restrict__ base, long long gld, int tok, int S, int f0, int dh) {
    if (f0 < dh) frag_load(f, base + (long long)(tok < S ? tok : S - 1) * gld + f0);
    else frag_zero(f);
}

// the same fragment from a staged LDS image (rows >= S and columns >= dh of the image are zero; f0 may lie beyond the staged columns)
template <typename T>
__device__ __forceinline__ void img_frag(Frag<T>& f, const T* img, int ldi, int tok, int f0, int dh) {
    if (f0 < dh) frag_load(f, img + tok * ldi + f0);
    else frag_zero(f);
}

__device__ __forceinline__ float quad_lane_max(float v) {        // over the 4 lanes p, p+16, p+32, p+48
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float quad_lane_sum(float v) {
    v += __shfl_xor(v, 16, 64);
    return v + __shfl_xor(v, 32, 64);
}

template <typename T>
__device__ __forceinline__ void store4(T* dst, const f32x4& o) {
    if (sizeof(T) == 2) {
        bf16x4 w; w[0] = (bf16)o[0]; w[1] = (bf16)o[1]; w[2] = (bf16)o[2]; w[3] = (bf16)o[3];
        *reinterpret_cast<bf16x4*>(dst) = w;
    } else {
        *reinterpret_cast<f32x4*>(dst) = o;
    }
}

// store4 behind a ReLU: zero where the forward value (same position in `ref`) is not positive
template <typename T>
__device__ __forceinline__ void store4_relu(T* dst, const T* ref, f32x4 o) {
    if (sizeof(T) == 2) {
        const bf16x4 r = *reinterpret_cast<const bf16x4*>(ref);
#pragma unroll
        for (int j = 0; j < 4; ++j) if (!((float)r[j] > 0.f)) o[j] = 0.f;
    } else {
        const f32x4 r = *reinterpret_cast<const f32x4*>(ref);
#pragma unroll
        for (int j = 0; j < 4; ++j) if (!(r[j] > 0.f)) o[j] = 0.f;
    }
    store4(dst, o);
}

constexpr int MAXT = 4;       // S <= 64
constexpr int MAXDT = 8;      // head width <= 128

// SINGLE: S <= 16 (one token tile): the tile loops collapse at compile time and the kernel fits 128 VGPRs (4 waves per SIMD)
template <typename T, bool SINGLE>
__global__ __launch_bounds__(256, SINGLE ? 4 : 2) void attention_fwd_kernel(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v, const float* __restrict__ mask,
                                     T* __restrict__ out, float* __restrict__ stats, AttnDims d, float scale, float p_drop, unsigned long long seed, const unsigned long long* __restrict__ seed_inc) {
    // device-side step counter (captured launches draw a fresh mask per replay): requested here, consumed after the V image barrier, so the
    // scalar round trip overlaps the staging loads instead of preceding everything
    unsigned long long seed_step = 0;
    if (p_drop > 0.f && seed_inc) seed_step = *seed_inc;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int NTC = SINGLE ? 1 : MAXT;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool multi = d.ppw > 1;                              // S <= 16: one wave = one whole problem, several problems per workgroup
    const int slot = multi ? wave_all : 0, wave = multi ? 0 : wave_all;
    const int nw = multi ? 1 : (int)(blockDim.x >> 6);        // waves per problem: each takes query tiles wave, wave + nw, ...
    int pidx = blockIdx.x * d.ppw + slot;
    const bool valid = pidx < d.B * d.H;
    if (!valid) pidx = d.B * d.H - 1;                          // keep every wave alive for the barrier and the cross-lane reads; no stores
    const int b = pidx / d.H, h = pidx - b * d.H;
    T* Vimg = reinterpret_cast<T*>(smem_raw) + (size_t)slot * d.nt * 16 * d.ldi;      // [16*nt][ldi]
    const int p = lane & 15, g = lane >> 4;
    const T* qb = q + (long long)b * d.S * d.ld_qkv + h * d.dh;
    const T* kb = k + (long long)b * d.S * d.ld_qkv + h * d.dh;

    stage_image(Vimg, v, d, b, h, d.ld_qkv, multi ? lane : tid, multi ? 64 : (int)blockDim.x);
    __syncthreads();                                           // V image complete
    seed += seed_step;
    const float inv_keep = p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f;
  for (int qt = wave; qt < d.nt; qt += nw) {
    // S^T tiles: rows = keys of tile kt, columns = this tile's 16 queries
    const int query = qt * 16 + p;
    f32x4 sT[NTC];
    const int ks = (d.dh + 31) >> 5;
    // (the loops over the <= 4 k-steps and the <= 4 key tiles have compile-time bounds with run-time predicates: the compiler then issues
    // every fragment load of a query tile before the first MFMA -- with run-time bounds each k-step waited for its own loads, one memory
    // round trip after the other: 8.7 us forward / 23.7 us backward for the 64 problems of BASELINE config 4)
    Frag<T> bq[4];
#pragma unroll
    for (int s = 0; s < 4; ++s)
        if (s < ks) row_frag(bq[s], qb, d.ld_qkv, query, d.S, s * 32 + 8 * g, d.dh);
#pragma unroll
    for (int kt = 0; kt < NTC; ++kt) {
        sT[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (kt < d.nt) {
            Frag<T> a[4];
#pragma unroll
            for (int s = 0; s < 4; ++s)
                if (s < ks) row_frag(a[s], kb, d.ld_qkv, kt * 16 + p, d.S, s * 32 + 8 * g, d.dh);
#pragma unroll
            for (int s = 0; s < 4; ++s)
                if (s < ks) sT[kt] = mma32(a[s], bq[s], sT[kt]);
        }
    }
    // scale, mask, softmax over the keys of each query (register + 4-lane reduction)
    const float* mrow = mask ? mask + ((long long)(pidx % d.B) * d.S + (query < d.S ? query : 0)) * d.S : nullptr;
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NTC; ++kt)
        if (kt < d.nt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = kt * 16 + 4 * g + r;
                float sv = sT[kt][r] * scale;
                if (key >= d.S) sv = -INFINITY;
                else if (mrow && mrow[key] == 0.f) sv = -1e9f;
                sT[kt][r] = sv;
                mx = fmaxf(mx, sv);
            }
        }
    mx = quad_lane_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < NTC; ++kt)
        if (kt < d.nt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float e = __expf(sT[kt][r] - mx); sT[kt][r] = e; sum += e; }
        }
    sum = quad_lane_sum(sum);
    const float inv = 1.f / sum;
    if (g == 0 && query < d.S && valid) {
        float* st = stats + ((long long)pidx * d.S + query) * 2;
        st[0] = mx; st[1] = sum;
    }
    Frag16<T> P[NTC];
#pragma unroll
    for (int kt = 0; kt < NTC; ++kt)
        if (kt < d.nt) {
            float pv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = kt * 16 + 4 * g + r;
                float x = sT[kt][r] * inv;
                if (p_drop > 0.f && key < d.S && query < d.S)
                    x *= dropout_mult(seed, ((unsigned long long)pidx * d.S + query) * d.S + key, p_drop, inv_keep);
                pv[r] = x;
            }
            acc_to_frag(P[kt], pv);
        }
    // O^T = V^T P^T, one 16-feature tile at a time; lane (p, g) ends with O[query p][16 dt + 4g .. + 3]
#pragma unroll
    for (int dt = 0; dt < MAXDT; ++dt) {
        if (dt < d.DT) {
            f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kt = 0; kt < NTC; ++kt)
                if (kt < d.nt) {
                    Frag16<T> a;
                    tr_read(a, Vimg, d.ldi, kt * 16, dt * 16, lane);
                    o = mma16(a, P[kt], o);
                }
            const int f0 = dt * 16 + 4 * g;
            if (valid && query < d.S && f0 < d.dh) store4(out + ((long long)(b * d.S + query)) * d.ld_o + h * d.dh + f0, o);
        }
    }
  }
}

template <typename T, bool SINGLE>
__global__ __launch_bounds__(256, 2) void attention_bwd_kernel(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v, const float* __restrict__ mask,
                                     const float* __restrict__ stats, const T* __restrict__ dout, T* __restrict__ dq, T* __restrict__ dk,
                                     T* __restrict__ dv, AttnDims d, float scale, float p_drop, unsigned long long seed, const unsigned long long* __restrict__ seed_inc) {
    unsigned long long seed_step = 0;                           // device-side step counter: requested here, consumed after the image barrier
    if (p_drop > 0.f && seed_inc) seed_step = *seed_inc;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int NTC = SINGLE ? 1 : MAXT;
    const int rows = d.nt * 16;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool multi = d.ppw > 1;
    const int slot = multi ? wave_all : 0, wave = multi ? 0 : wave_all;
    const int nw = multi ? 1 : (int)(blockDim.x >> 6);
    int pidx = blockIdx.x * d.ppw + slot;
    const bool valid = pidx < d.B * d.H;
    if (!valid) pidx = d.B * d.H - 1;
    const int b = pidx / d.H, h = pidx - b * d.H;
    T* Kimg = reinterpret_cast<T*>(smem_raw + (size_t)slot * ((size_t)3 * rows * d.ldi * sizeof(T) + (size_t)rows * sizeof(float)));
    T* Qimg = Kimg + rows * d.ldi;
    T* Gimg = Qimg + rows * d.ldi;                             // dO
    float* delta = reinterpret_cast<float*>(Gimg + rows * d.ldi);   // [rows]
    // several tiles: phase A leaves the (dropped) probabilities and the score gradients of its query tile here, [query][key]; phase B reads
    // them back transposed instead of recomputing both products, the exponentials and the row statistics per (key tile, query tile) pair
    T* Pimg = reinterpret_cast<T*>(delta + rows);
    T* Simg = Pimg + rows * d.ldp;
    // 16-bit storage, several tiles: V is staged too and phase A takes every row fragment from the images (one global pass per operand)
    constexpr bool VIMG = !SINGLE && sizeof(T) == 2;
    T* Vimg = Simg + rows * d.ldp;
    const int p = lane & 15, g = lane >> 4;
    const long long boff = (long long)b * d.S * d.ld_qkv + h * d.dh;
    const T* qb = q + boff;
    const T* kb = k + boff;
    const T* vb = v + boff;
    const T* gb = dout + (long long)b * d.S * d.ld_o + h * d.dh;
    const float* mbase = mask ? mask + (long long)(pidx % d.B) * d.S * d.S : nullptr;
    const float inv_keep = p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f;
    const int ks = (d.dh + 31) >> 5;

    {
        const int pt = multi ? lane : tid, pn = multi ? 64 : (int)blockDim.x;
        stage_image(Kimg, k, d, b, h, d.ld_qkv, pt, pn);
        stage_image(Qimg, q, d, b, h, d.ld_qkv, pt, pn);
        stage_image(Gimg, dout, d, b, h, d.ld_o, pt, pn);
        if (VIMG) stage_image(Vimg, v, d, b, h, d.ld_qkv, pt, pn);
    }
    // the softmax statistics of this wave's first query tile: requested before the barrier, so the round trip rides under the staging
    float st0_mx = 0.f, st0_l = 1.f;
    if (wave * 16 + p < d.S) { const float* st = stats + ((long long)pidx * d.S + wave * 16 + p) * 2; st0_mx = st[0]; st0_l = st[1]; }
    __syncthreads();                                           // images complete
    seed += seed_step;
    // single-tile sequences (S <= 16): the row fragments of K, Q, V, dO are the same registers in both phases (A and B operands
    // of the 16x16x32 MFMA have the same lane layout), so they are loaded once
    // (several tiles: fq / fg hold the current QUERY tile's rows in phase A and fk / fv the current KEY tile's rows in phase B -- loaded once
    // per tile, not once per (tile, tile) pair; every loop over k-steps and tiles has a compile-time bound, see attention_fwd_kernel)
    // (16-bit storage only: fp32 fragments are twice the registers, its multi-tile path keeps one k-step's fragments at a time)
    constexpr bool HOIST = SINGLE || sizeof(T) == 2;
    Frag<T> fk[HOIST ? 4 : 1], fq[HOIST ? 4 : 1], fv[HOIST ? 4 : 1], fg[HOIST ? 4 : 1];

    // ---------------- phase A: one query tile at a time, all keys (swapped orientation: rows = keys, column = query p)
    for (int qt = wave; qt < d.nt; qt += nw) {
        const int query = qt * 16 + p;
        const bool qok = query < d.S;
        float mx = 0.f, inv = 0.f;
        if (qt == wave) { mx = st0_mx; inv = qok ? 1.f / st0_l : 0.f; }
        else if (qok) { const float* st = stats + ((long long)pidx * d.S + query) * 2; mx = st[0]; inv = 1.f / st[1]; }
        const float* mrow = mbase ? mbase + (long long)(qok ? query : 0) * d.S : nullptr;
        f32x4 pT[NTC], dpT[NTC];
        float dl = 0.f;
        if constexpr (!SINGLE && HOIST) {
#pragma unroll
            for (int s = 0; s < 4; ++s)
                if (s < ks) {
                    const int f0 = s * 32 + 8 * g;
                    img_frag(fq[s], Qimg, d.ldi, qt * 16 + p, f0, d.dh);
                    img_frag(fg[s], Gimg, d.ldi, qt * 16 + p, f0, d.dh);
                }
        }
#pragma unroll
        for (int kt = 0; kt < NTC; ++kt) {
            pT[kt] = f32x4{0.f, 0.f, 0.f, 0.f}; dpT[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (kt < d.nt) {
                if (SINGLE) {
#pragma unroll
                    for (int s = 0; s < 4; ++s)
                        if (s < ks) {
                            const int f0 = s * 32 + 8 * g;
                            row_frag(fk[s], kb, d.ld_qkv, p, d.S, f0, d.dh);
                            row_frag(fq[s], qb, d.ld_qkv, p, d.S, f0, d.dh);
                            row_frag(fv[s], vb, d.ld_qkv, p, d.S, f0, d.dh);
                            row_frag(fg[s], gb, d.ld_o, p, d.S, f0, d.dh);
                        }
#pragma unroll
                    for (int s = 0; s < 4; ++s)
                        if (s < ks) {
                            pT[kt] = mma32(fk[s], fq[s], pT[kt]);
                            dpT[kt] = mma32(fv[s], fg[s], dpT[kt]);
                        }
                } else if constexpr (HOIST) {
                    Frag<T> ak[4], av[4];
#pragma unroll
                    for (int s = 0; s < 4; ++s)
                        if (s < ks) {
                            const int f0 = s * 32 + 8 * g;
                            img_frag(ak[s], Kimg, d.ldi, kt * 16 + p, f0, d.dh);
                            img_frag(av[s], Vimg, d.ldi, kt * 16 + p, f0, d.dh);
                        }
#pragma unroll
                    for (int s = 0; s < 4; ++s)
                        if (s < ks) {
                            pT[kt] = mma32(ak[s], fq[s], pT[kt]);
                            dpT[kt] = mma32(av[s], fg[s], dpT[kt]);
                        }
                } else {
                    for (int s = 0; s < ks; ++s) {
                        Frag<T> ak, bq, av, bg;
                        const int f0 = s * 32 + 8 * g;
                        row_frag(ak, kb, d.ld_qkv, kt * 16 + p, d.S, f0, d.dh);
                        row_frag(bq, qb, d.ld_qkv, query, d.S, f0, d.dh);
                        row_frag(av, vb, d.ld_qkv, kt * 16 + p, d.S, f0, d.dh);
                        row_frag(bg, gb, d.ld_o, query, d.S, f0, d.dh);
                        pT[kt] = mma32(ak, bq, pT[kt]);
                        dpT[kt] = mma32(av, bg, dpT[kt]);
                    }
                }
                float pd[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = kt * 16 + 4 * g + r;
                    float sv = pT[kt][r] * scale;
                    if (mrow && key < d.S && mrow[key] == 0.f) sv = -1e9f;
                    float pv = (qok && key < d.S) ? __expf(sv - mx) * inv : 0.f;
                    float dp = dpT[kt][r], mult = 1.f;
                    if (p_drop > 0.f && qok && key < d.S) {
                        mult = dropout_mult(seed, ((unsigned long long)pidx * d.S + query) * d.S + key, p_drop, inv_keep);
                        dp *= mult;
                    }
                    pT[kt][r] = pv; dpT[kt][r] = dp;
                    dl += pv * dp;
                    if (!SINGLE) pd[r] = pv * mult;
                }
                if (!SINGLE) {                                 // P (with the dropout mask applied) of this (query row, 4 keys): for dV in phase B
                    Frag16<T> pf;
                    acc_to_frag(pf, pd);
                    *reinterpret_cast<Frag16<T>*>(Pimg + (qt * 16 + p) * d.ldp + kt * 16 + 4 * g) = pf;
                }
            }
        }
        dl = quad_lane_sum(dl);
        if (SINGLE && g == 0) delta[qt * 16 + p] = dl;
        Frag16<T> dsT[NTC];
#pragma unroll
        for (int kt = 0; kt < NTC; ++kt)
            if (kt < d.nt) {
                float x[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) x[r] = pT[kt][r] * (dpT[kt][r] - dl) * scale;
                acc_to_frag(dsT[kt], x);
                if (!SINGLE) *reinterpret_cast<Frag16<T>*>(Simg + (qt * 16 + p) * d.ldp + kt * 16 + 4 * g) = dsT[kt];      // dS: for dK in phase B
            }
#pragma unroll
        for (int dt = 0; dt < MAXDT; ++dt)
            if (dt < d.DT) {
                f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kt = 0; kt < NTC; ++kt)
                    if (kt < d.nt) {
                        Frag16<T> a;
                        tr_read(a, Kimg, d.ldi, kt * 16, dt * 16, lane);      // K^T[feature][key]
                        o = mma16(a, dsT[kt], o);
                    }
                const int f0 = dt * 16 + 4 * g;
                if (valid && qok && f0 < d.dh) {
                    const long long oq = ((long long)(b * d.S + query)) * d.ld_qkv + h * d.dh + f0;
                    if (d.relu_out) store4_relu(dq + oq, q + oq, o); else store4(dq + oq, o);
                }
            }
    }
    __syncthreads();                                           // delta (one tile) / the P and dS images (several) are complete
    // ---------------- phase B: one KEY tile at a time, all query tiles (normal orientation: rows = queries, column = key p)
    if constexpr (SINGLE) {
        for (int kt = wave; kt < d.nt; kt += nw) {
            const int key = kt * 16 + p;
            const bool kok = key < d.S;
            f32x4 sN = f32x4{0.f, 0.f, 0.f, 0.f}, dpN = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 4; ++s)
                if (s < ks) {
                    sN = mma32(fq[s], fk[s], sN);
                    dpN = mma32(fg[s], fv[s], dpN);
                }
            float ds[4], pd[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int query = 4 * g + r;
                const bool ok = kok && query < d.S;
                float pv = 0.f, mult = 1.f;
                if (ok) {
                    const float* st = stats + ((long long)pidx * d.S + query) * 2;
                    float sv = sN[r] * scale;
                    if (mbase && mbase[(long long)query * d.S + key] == 0.f) sv = -1e9f;
                    pv = __expf(sv - st[0]) / st[1];
                    if (p_drop > 0.f) mult = dropout_mult(seed, ((unsigned long long)pidx * d.S + query) * d.S + key, p_drop, inv_keep);
                }
                ds[r] = pv * (dpN[r] * mult - delta[4 * g + r]) * scale;
                pd[r] = pv * mult;
            }
            Frag16<T> dsF, pdF;
            acc_to_frag(dsF, ds);
            acc_to_frag(pdF, pd);
#pragma unroll
            for (int dt = 0; dt < MAXDT; ++dt)
                if (dt < d.DT) {
                    Frag16<T> aq, ag;
                    tr_read(aq, Qimg, d.ldi, 0, dt * 16, lane);                // Q^T[feature][query]
                    tr_read(ag, Gimg, d.ldi, 0, dt * 16, lane);                // dO^T[feature][query]
                    const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
                    const f32x4 tk = mma16(aq, dsF, z), tv = mma16(ag, pdF, z);          // matrix ops stay outside the divergent store branch
                    const int f0 = dt * 16 + 4 * g;
                    if (valid && kok && f0 < d.dh) {
                        const long long o = ((long long)(b * d.S + key)) * d.ld_qkv + h * d.dh + f0;
                        if (d.relu_out) { store4_relu(dk + o, k + o, tk); store4_relu(dv + o, v + o, tv); }
                        else { store4(dk + o, tk); store4(dv + o, tv); }
                    }
                }
        }
    } else {
        // several tiles: dK^T = Q^T dS, dV^T = dO^T P with all four operands read transposed from the LDS images -- no global loads, no
        // exponentials and no row statistics in this phase
        for (int kt = wave; kt < d.nt; kt += nw) {
            const int key = kt * 16 + p;
            const bool kok = key < d.S;
            f32x4 dkT[MAXDT], dvT[MAXDT];
#pragma unroll
            for (int dt = 0; dt < MAXDT; ++dt) { dkT[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dvT[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
            for (int qt = 0; qt < MAXT; ++qt)
                if (qt < d.nt) {
                    Frag16<T> dsF, pdF;
                    tr_read(dsF, Simg, d.ldp, qt * 16, kt * 16, lane);         // dS[query 4g + j][key p]
                    tr_read(pdF, Pimg, d.ldp, qt * 16, kt * 16, lane);
#pragma unroll
                    for (int dt = 0; dt < MAXDT; ++dt)
                        if (dt < d.DT) {
                            Frag16<T> aq, ag;
                            tr_read(aq, Qimg, d.ldi, qt * 16, dt * 16, lane);  // Q^T[feature][query]
                            tr_read(ag, Gimg, d.ldi, qt * 16, dt * 16, lane);  // dO^T[feature][query]
                            dkT[dt] = mma16(aq, dsF, dkT[dt]);
                            dvT[dt] = mma16(ag, pdF, dvT[dt]);
                        }
                }
#pragma unroll
            for (int dt = 0; dt < MAXDT; ++dt)
                if (dt < d.DT) {
                    const int f0 = dt * 16 + 4 * g;
                    if (valid && kok && f0 < d.dh) {
                        const long long o = ((long long)(b * d.S + key)) * d.ld_qkv + h * d.dh + f0;
                        if (d.relu_out) { store4_relu(dk + o, k + o, dkT[dt]); store4_relu(dv + o, v + o, dvT[dt]); }
                        else { store4(dk + o, dkT[dt]); store4(dv + o, dvT[dt]); }
                    }
                }
        }
    }
}

inline bool attn_dims(AttnDims& d, int B, int S, int D, int H, size_t es) {
    if (B <= 0 || S <= 0 || S > 16 * MAXT || D <= 0 || H <= 0 || D % H != 0) return false;
    d.B = B; d.S = S; d.D = D; d.H = H; d.dh = D / H;
    if (d.dh % 8 != 0 || d.dh > 16 * MAXDT) return false;
    d.nt = (S + 15) / 16;
    d.DT = (d.dh + 15) / 16;
    // row stride = an odd multiple of 32 bytes: the 8 rows x 32 B of a half-wave's transposed read then tile all 64 banks
    int bytes = d.DT * 16 * (int)es;
    if ((bytes / 32) % 2 == 0) bytes += 32;
    d.ldi = bytes / (int)es;
    int pbytes = d.nt * 16 * (int)es;
    if ((pbytes / 32) % 2 == 0) pbytes += 32;
    d.ldp = pbytes / (int)es;
    d.ld_qkv = D;
    d.ld_o = D;
    d.relu_out = 0;
    // a grid of one-wave workgroups is dispatch-bound beyond a few thousand problems: pack four single-tile problems per workgroup then
    static const int ppw_env = getenv("HYB_ATTN_PPW") ? atoi(getenv("HYB_ATTN_PPW")) : 4;      // waves (= problems) per workgroup at S <= 16
    d.ppw = (d.nt == 1 && (long long)B * H >= 2048) ? (ppw_env >= 1 && ppw_env <= 16 ? ppw_env : 4) : 1;
    return true;
}

// waves per workgroup: default one per 16-token tile; HYB_ATTN_QROWS = 16 / 32 / 64 gives a wave that many query rows (the
// tile-size sweep of BASELINE config 4, profiles/r02_attention_sweep.json)
inline int attn_waves(const AttnDims& d) {
    static const int qrows = getenv("HYB_ATTN_QROWS") ? atoi(getenv("HYB_ATTN_QROWS")) : 16;
    const int per = qrows >= 64 ? 4 : qrows >= 32 ? 2 : 1;
    return (d.nt + per - 1) / per;
}

template <typename T>
int attn_fwd_t(const void* q, const void* k, const void* v, const float* mask, void* out, float* stats, const AttnDims& d, float p_drop,
               unsigned long long seed, const unsigned long long* seed_inc, hipStream_t st) {
    const size_t lds = (size_t)d.ppw * d.nt * 16 * d.ldi * sizeof(T);
    const float scale = 1.0f / sqrtf((float)d.D);
    const dim3 grid(hyb_cdiv((long long)d.B * d.H, d.ppw)), block((d.ppw > 1 ? d.ppw : attn_waves(d)) * 64);
    if (d.nt == 1)
        hipLaunchKernelGGL((attention_fwd_kernel<T, true>), grid, block, lds, st, (const T*)q, (const T*)k, (const T*)v, mask, (T*)out, stats, d, scale,
                           p_drop, seed, seed_inc);
    else
        hipLaunchKernelGGL((attention_fwd_kernel<T, false>), grid, block, lds, st, (const T*)q, (const T*)k, (const T*)v, mask, (T*)out, stats, d, scale,
                           p_drop, seed, seed_inc);
    HYB_LAUNCH_CHECK();
    return 0;
}
template <typename T>
int attn_bwd_t(const void* q, const void* k, const void* v, const float* mask, const float* stats, const void* dout, void* dq, void* dk, void* dv,
               const AttnDims& d, float p_drop, unsigned long long seed, const unsigned long long* seed_inc, hipStream_t st) {
    const size_t lds = (size_t)d.ppw * ((size_t)3 * d.nt * 16 * d.ldi * sizeof(T) + (size_t)d.nt * 16 * sizeof(float)) +
                       (d.nt > 1 ? (size_t)2 * d.nt * 16 * d.ldp * sizeof(T) + (sizeof(T) == 2 ? (size_t)d.nt * 16 * d.ldi * sizeof(T) : 0) : 0);
    // (fp32, 128-wide heads, S = 64: 104 + 36 KiB; bf16, 96-wide, S = 64: 43 + 20 + 14 KiB -- two workgroups per CU)
    const float scale = 1.0f / sqrtf((float)d.D);
    const dim3 grid(hyb_cdiv((long long)d.B * d.H, d.ppw)), block((d.ppw > 1 ? d.ppw : attn_waves(d)) * 64);
    if (d.nt == 1) {
        if (lds > 64 * 1024) { static HybAttrOnce once; if (int e = hyb_set_lds_attr(once, (const void*)attention_bwd_kernel<T, true>, 160 * 1024)) return e; }
        hipLaunchKernelGGL((attention_bwd_kernel<T, true>), grid, block, lds, st, (const T*)q, (const T*)k, (const T*)v, mask, stats,
                           (const T*)dout, (T*)dq, (T*)dk, (T*)dv, d, scale, p_drop, seed, seed_inc);
    } else {
        if (lds > 64 * 1024) { static HybAttrOnce once; if (int e = hyb_set_lds_attr(once, (const void*)attention_bwd_kernel<T, false>, 160 * 1024)) return e; }
        hipLaunchKernelGGL((attention_bwd_kernel<T, false>), grid, block, lds, st, (const T*)q, (const T*)k, (const T*)v, mask, stats,
                           (const T*)dout, (T*)dq, (T*)dk, (T*)dv, d, scale, p_drop, seed, seed_inc);
    }
    HYB_LAUNCH_CHECK();
    return 0;
}

}  // namespace

extern "C" int hyb_attention_fwd(int dtype, const void* q, const void* k, const void* v, const float* mask, void* out, float* stats, int B,
                                 int S, int D, int H, float p_drop, unsigned long long seed, void* stream) {
    AttnDims d;
    HYB_CHECK_ARG(dtype == HYB_F32 || dtype == HYB_BF16);
    HYB_CHECK_ARG(q && k && v && out && stats && attn_dims(d, B, S, D, H, dtype == HYB_F32 ? 4 : 2) && p_drop >= 0.f && p_drop < 1.f);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == HYB_F32) return attn_fwd_t<float>(q, k, v, mask, out, stats, d, p_drop, seed, nullptr, st);
    return attn_fwd_t<bf16>(q, k, v, mask, out, stats, d, p_drop, seed, nullptr, st);
}

extern "C" int hyb_attention_bwd(int dtype, const void* q, const void* k, const void* v, const float* mask, const float* stats, const void* dout,
                                 void* dq, void* dk, void* dv, int B, int S, int D, int H, float p_drop, unsigned long long seed, void* stream) {
    AttnDims d;
    HYB_CHECK_ARG(dtype == HYB_F32 || dtype == HYB_BF16);
    HYB_CHECK_ARG(q && k && v && stats && dout && dq && dk && dv && attn_dims(d, B, S, D, H, dtype == HYB_F32 ? 4 : 2) && p_drop >= 0.f && p_drop < 1.f);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == HYB_F32) return attn_bwd_t<float>(q, k, v, mask, stats, dout, dq, dk, dv, d, p_drop, seed, nullptr, st);
    return attn_bwd_t<bf16>(q, k, v, mask, stats, dout, dq, dk, dv, d, p_drop, seed, nullptr, st);
}

// Internal (same shared object): q/k/v (and dq/dk/dv) packed as [B*S][3D] -- used by hyb_encoder_{fwd,bwd}
int hyb_attention_fwd_packed(int dtype, const void* qkv, const float* mask, void* out, float* stats, int B, int S, int D, int H, float p_drop,
                             unsigned long long seed, const unsigned long long* seed_inc, hipStream_t st) {
    AttnDims d;
    const size_t es = dtype == HYB_F32 ? 4 : 2;
    if (!qkv || !out || !stats || !attn_dims(d, B, S, D, H, es)) return HYB_E_ARG;
    d.ld_qkv = 3 * D;
    const char* base = (const char*)qkv;
    if (dtype == HYB_F32) return attn_fwd_t<float>(base, base + D * es, base + 2 * D * es, mask, out, stats, d, p_drop, seed, seed_inc, st);
    if (dtype == HYB_BF16) return attn_fwd_t<bf16>(base, base + D * es, base + 2 * D * es, mask, out, stats, d, p_drop, seed, seed_inc, st);
    return HYB_E_ARG;
}
int hyb_attention_bwd_packed(int dtype, const void* qkv, const float* mask, const float* stats, const void* dout, void* dqkv, int B, int S, int D,
                             int H, float p_drop, unsigned long long seed, const unsigned long long* seed_inc, hipStream_t st, int relu_out) {
    AttnDims d;
    const size_t es = dtype == HYB_F32 ? 4 : 2;
    if (!qkv || !stats || !dout || !dqkv || !attn_dims(d, B, S, D, H, es)) return HYB_E_ARG;
    d.ld_qkv = 3 * D;
    d.relu_out = relu_out;
    const char* base = (const char*)qkv;
    char* g = (char*)dqkv;
    if (dtype == HYB_F32) return attn_bwd_t<float>(base, base + D * es, base + 2 * D * es, mask, stats, dout, g, g + D * es, g + 2 * D * es, d, p_drop, seed, seed_inc, st);
    if (dtype == HYB_BF16) return attn_bwd_t<bf16>(base, base + D * es, base + 2 * D * es, mask, stats, dout, g, g + D * es, g + 2 * D * es, d, p_drop, seed, seed_inc, st);
    return HYB_E_ARG;
}

// ---------------------------------------------------------------------------------------------------------------------------
// Long-sequence forward (FCT's spatial attention: nn.MultiheadAttention over H*W = 1 024 .. 12 544 pixel tokens, FCT.py:37,75):
// the same swapped-product structure with an ONLINE softmax over 64-key blocks.  A workgroup = 4 waves = 64 queries of one
// (image, head); K and V blocks are staged in LDS once per block for all four waves.  Per lane the running max / sum belong to
// ONE query (column p of the swapped product), so the rescale of the O^T accumulators is a per-lane scalar multiply.
// q, k, v, out: [N*L][ld] with head h at features [h*dhp, (h+1)*dhp), dhp a multiple of 8 (narrower heads are zero-padded by the
// caller: zero features change neither scores nor outputs).  No mask, no dropout (the reference passes neither).
// ---------------------------------------------------------------------------------------------------------------------------
namespace {

typedef __attribute__((address_space(3))) const bf16x8 lds_bf16x8_c;

// What the temporal encoder's attention() (TransformerEncoder.pyc src L49-62) has and FCT's nn.MultiheadAttention call does not: the mask
// (masked_fill(mask == 0, -1e9) with the reference's head-replication order, quirk Q4: problem b*H + h reads mask[(b*H + h) mod B]) and the
// dropout on the softmax weights (quirk Q5).  All-zero = neither.  The dropout multiplier of weight (problem, query, key) is a pure function
// of (seed, index), so the backward kernels regenerate it; the normaliser and the saved log-sum-exp are those of the UNdropped softmax, and
// delta = rowsum(dO * O) stays valid because O already contains the dropped weights.
struct FlashExtra {
    const float* mask;                   // [Bmask][L][L] or nullptr
    int Bmask;
    float p_drop, inv_keep;
    unsigned long long seed;
    const unsigned long long* seed_inc;  // device-side step counter added to the seed (graph replay), or nullptr
};
__device__ __forceinline__ unsigned long long flash_seed(const FlashExtra& ex) {
    return ex.seed + ((ex.p_drop > 0.f && ex.seed_inc) ? *ex.seed_inc : 0ull);
}

// Feature-contraction step of the long-sequence kernels (S = K Q^T, dP = V dO^T).  fp32: 16 features per step = four 16x16x4 MFMAs with
// 4 consecutive features per lane -- FCT's heads are 4..64 wide (padded to 8..64), and a 32-wide step would spend eight MFMAs on a
// head of 8; bf16: the 32-wide single MFMA.
template <typename T> struct Feat;
template <> struct Feat<float> { static constexpr int W = 16; typedef Frag16<float> F; };
template <> struct Feat<bf16> { static constexpr int W = 32; typedef Frag<bf16> F; };
__device__ __forceinline__ void ff_load(Frag16<float>& f, const float* p) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p);
    f.v[0] = a[0]; f.v[1] = a[1]; f.v[2] = a[2]; f.v[3] = a[3];
}
__device__ __forceinline__ void ff_load(Frag<bf16>& f, const bf16* p) { frag_load(f, p); }
__device__ __forceinline__ void ff_zero(Frag16<float>& f) { f.v[0] = f.v[1] = f.v[2] = f.v[3] = 0.f; }
__device__ __forceinline__ void ff_zero(Frag<bf16>& f) { frag_zero(f); }
__device__ __forceinline__ f32x4 ff_mma(const Frag16<float>& a, const Frag16<float>& b, f32x4 c) { return mma16(a, b, c); }
__device__ __forceinline__ f32x4 ff_mma(const Frag<bf16>& a, const Frag<bf16>& b, f32x4 c) { return mma32(a, b, c); }
// narrowest fp32 step: 8 features = two 16x16x4 MFMAs, 2 consecutive features per lane -- heads of 4 (padded to 8) and 8 features, which
// are FCT's 12 544- and 3 136-token attentions (where the time goes)
struct Frag8f { float v[2]; };
__device__ __forceinline__ void ff_load(Frag8f& f, const float* p) {
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
    const f32x2_ a = *reinterpret_cast<const f32x2_*>(p);
    f.v[0] = a[0]; f.v[1] = a[1];
}
__device__ __forceinline__ void ff_zero(Frag8f& f) { f.v[0] = f.v[1] = 0.f; }
__device__ __forceinline__ f32x4 ff_mma(const Frag8f& a, const Frag8f& b, f32x4 c) {
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v[0], b.v[0], c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a.v[1], b.v[1], c, 0, 0, 0);
}
template <typename T, int FWP> struct FeatSel { typedef typename Feat<T>::F F; static constexpr int W = Feat<T>::W; };
template <> struct FeatSel<float, 8> { typedef Frag8f F; static constexpr int W = 8; };

template <typename T, typename F>
__device__ __forceinline__ void ff_row(F& f, const T* row, int f0, int dhp) {      // features f0 .. of one row (global or LDS)
    if (f0 < dhp) ff_load(f, row + f0);
    else ff_zero(f);
}

template <typename T, int FWP = 0 /* 8: the 8-feature fp32 step (dhp == 8 only) */>
__global__ __launch_bounds__(256, 2) void flash_fwd_kernel(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v, T* __restrict__ out,
                                                           float* __restrict__ lse, int L, int H, int dhp, int ld, int ldi, float scale, FlashExtra ex) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const unsigned long long dseed = flash_seed(ex);
    T* Kimg = reinterpret_cast<T*>(smem_raw);                  // [64][ldi]
    T* Vimg = Kimg + 64 * ldi;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int p = lane & 15, g = lane >> 4;
    const int n = blockIdx.y / H, h = blockIdx.y - n * H;
    const long long base = (long long)n * L * ld + h * dhp;
    using FF = typename FeatSel<T, FWP>::F;
    constexpr int FW = FeatSel<T, FWP>::W, FL = FW / 4, MAXFS = FWP == 8 ? 1 : 16 * MAXDT / FW;      // step width, features per lane and step, most steps
    const int DT = (dhp + 15) >> 4, ks = (dhp + FW - 1) / FW, segs = dhp >> 3;
    const int query = blockIdx.x * 64 + wave * 16 + p;
    const int qrow = query < L ? query : L - 1;
    FF fq[MAXFS];
#pragma unroll
    for (int s = 0; s < MAXFS; ++s)
        if (s < ks) ff_row<T>(fq[s], q + base + (long long)qrow * ld, s * FW + FL * g, dhp);
    float m = -INFINITY, l = 0.f;
    f32x4 o[MAXDT];
#pragma unroll
    for (int dt = 0; dt < MAXDT; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int kb = 0; kb < L; kb += 64) {
        __syncthreads();                                       // every wave is done with the previous block's images
        for (int u = tid; u < 64 * segs; u += 256) {
            const int r = u / segs, c = (u - r * segs) * 8;
            Vec8<T> a, b;
            if (kb + r < L) { a.load(k + base + (long long)(kb + r) * ld + c); b.load(v + base + (long long)(kb + r) * ld + c); }
            else { a.zero(); b.zero(); }
            a.store(Kimg + r * ldi + c);
            b.store(Vimg + r * ldi + c);
        }
        if ((dhp & 15) != 0)                                   // 8-wide tail of the last 16-feature tile: zero it (transposed reads touch it)
            for (int u = tid; u < 64; u += 256) { Vec8<T> z; z.zero(); z.store(Vimg + u * ldi + dhp); }
        __syncthreads();
        f32x4 sT[4];
        float mx = m;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            sT[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < MAXFS; ++s)
                if (s < ks) {
                    FF a;
                    ff_row<T>(a, Kimg + (kt * 16 + p) * ldi, s * FW + FL * g, dhp);
                    sT[kt] = ff_mma(a, fq[s], sT[kt]);
                }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = kb + kt * 16 + 4 * g + r;
                float sv = (key < L) ? sT[kt][r] * scale : -INFINITY;
                if (ex.mask && key < L && ex.mask[((long long)(blockIdx.y % ex.Bmask) * L + qrow) * L + key] == 0.f) sv = -1e9f;
                sT[kt][r] = sv;
                mx = fmaxf(mx, sv);
            }
        }
        mx = quad_lane_max(mx);                                // >= the previous m, finite (every block holds at least one real key)
        const float alpha = __expf(m - mx);
        float sum = 0.f;
        Frag16<T> P[4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            float pv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                pv[r] = __expf(sT[kt][r] - mx);
                sum += pv[r];
                if (ex.p_drop > 0.f)         // only the weights that multiply V are dropped
                    pv[r] *= dropout_mult(dseed, ((unsigned long long)blockIdx.y * L + qrow) * L + (kb + kt * 16 + 4 * g + r), ex.p_drop, ex.inv_keep);
            }
            acc_to_frag(P[kt], pv);
        }
        l = l * alpha + quad_lane_sum(sum);
        m = mx;
#pragma unroll
        for (int dt = 0; dt < MAXDT; ++dt)
            if (dt < DT) {
                f32x4 acc = o[dt] * alpha;
#pragma unroll
                for (int kt = 0; kt < 4; ++kt) {
                    Frag16<T> a;
                    tr_read(a, Vimg, ldi, kt * 16, dt * 16, lane);
                    acc = mma16(a, P[kt], acc);
                }
                o[dt] = acc;
            }
    }
    const float inv = 1.f / l;
    if (lse && g == 0 && query < L) lse[(long long)blockIdx.y * L + query] = m + __logf(l);      // log-sum-exp of the scaled scores (backward)
#pragma unroll
    for (int dt = 0; dt < MAXDT; ++dt)
        if (dt < DT) {
            const int f0 = dt * 16 + 4 * g;
            if (query < L && f0 < dhp) store4(out + base + (long long)query * ld + f0, o[dt] * inv);
        }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Heads of <= 4 features (FCT's two 12 544-token attentions: embed 8, 2 heads -- 157 M score pairs per image and head, where the
// forward's time goes).  The 16x16x4 MFMA wastes most of itself on such a head (S = K Q^T pads 4 features to 8, O^T = V^T P^T fills 4 of
// 16 output rows: 24 instructions x 32 cycles per 1024 pairs).  v_mfma_f32_4x4x1_16B (16 independent 4x4 outer products, 8 cycles; lane
// l: block l/4, A row / B column l%4, result register r = A[block][r] * B[block][l%4] -- scripts/micro/mfma4x4_layout.hip) fits exactly:
//   * a LANE owns one query (64 per wave, 256 per workgroup): B operand = its own 4 (pre-scaled) q features, held in registers;
//   * S step: A operand = feature f of key (k0 + l%4) -> 4 instructions (f = 0..3) leave the lane's scores for keys k0..k0+3 in the 4
//     result registers: the whole online softmax (max, exp2, sum, rescale) is per-lane arithmetic, no cross-lane step at all;
//   * PV step: A operand = feature l%4 of key k0+i, B operand = the lane's own probability for that key -> 4 instructions (i = 0..3)
//     accumulate the lane's 4 output features.  8 instructions x 8 cycles per 256 pairs: 4x fewer matrix cycles; the kernel is
//     bound by the exponentials (one per pair, quarter rate) instead.
// K rows and V columns of a 256-key block are staged in LDS (8 KB; the next block's rows are already in flight in registers); every
// operand read is a 16-byte broadcast read (all 16 blocks read the same four addresses; V rows padded by 4 floats: conflict-free).
// ---------------------------------------------------------------------------------------------------------------------------
constexpr int F4_KB = 256, F4_VLD = F4_KB + 4;
constexpr float F4_LAZY = 12.f;           // the exponent's reference point moves when a score exceeds it by more than 2^12 (fp32 sums: far from overflow)
// NF = feature quads per head: 1 for heads of <= 4 features (12 544 tokens), 2 for heads of 5..8 (3 136 tokens); the padded head width is 8.
template <int NF>
__global__ __launch_bounds__(256, 2) void flash_fwd4_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                                            float* __restrict__ out, float* __restrict__ lse, int L, int H, int ld,
                                                            float scale_log2e) {
    constexpr int FW = 4 * NF;
    __shared__ __attribute__((aligned(16))) float Kimg[F4_KB * FW];       // [key][feature]
    __shared__ __attribute__((aligned(16))) float Vt[FW * F4_VLD];        // [feature][key]
    const int tid = threadIdx.x, j4 = tid & 3;
    const int n = blockIdx.y / H, h = blockIdx.y - n * H;
    const long long base = (long long)n * L * ld + h * 8;                 // padded head width 8
    const int query = blockIdx.x * 256 + tid;
    const int qrow = query < L ? query : L - 1;
    f32x4 qv[NF];
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) { qv[nf] = *reinterpret_cast<const f32x4*>(q + base + (long long)qrow * ld + 4 * nf); qv[nf] *= scale_log2e; }
    // LAZY RESCALING.  m is a per-lane reference point of the exponent, not the running maximum: the score accumulators START at -m, so
    // the matrix instruction delivers s - m and p = exp2(s - m) needs no subtraction; m moves (with the usual rescale of l and o) only
    // when some score exceeds it by more than 2^F4_LAZY -- after the first keys that is rare, and the test is one wave-uniform branch.
    // softmax is shift-invariant, so the result is the same function; lse = m ln2 + ln(l) stays exact.  m starts at the score of key 0.
    float l = 0.f;
    const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 o[NF][4];
#pragma unroll
    for (int nf = 0; nf < NF; ++nf)
#pragma unroll
        for (int g = 0; g < 4; ++g) o[nf][g] = zero4;
    float m = 0.f;
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) {
        const f32x4 k0v = *reinterpret_cast<const f32x4*>(k + base + 4 * nf);
        m += (qv[nf][0] * k0v[0] + qv[nf][1] * k0v[1]) + (qv[nf][2] * k0v[2] + qv[nf][3] * k0v[3]);
    }
    f32x4 kr[NF], vr[NF];
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) {
        kr[nf] = zero4; vr[nf] = zero4;
        if (tid < L) { kr[nf] = *reinterpret_cast<const f32x4*>(k + base + (long long)tid * ld + 4 * nf); vr[nf] = *reinterpret_cast<const f32x4*>(v + base + (long long)tid * ld + 4 * nf); }
    }
    for (int kb = 0; kb < L; kb += F4_KB) {
        __syncthreads();                                                   // every wave is done with the previous block's images
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) {
            *reinterpret_cast<f32x4*>(Kimg + tid * FW + 4 * nf) = kr[nf];
#pragma unroll
            for (int f = 0; f < 4; ++f) Vt[(4 * nf + f) * F4_VLD + tid] = vr[nf][f];
        }
        __syncthreads();
        {
            const int nx = kb + F4_KB + tid;                               // the next block's row of this thread: in flight during the block
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) {
                kr[nf] = zero4; vr[nf] = zero4;
                if (nx < L) { kr[nf] = *reinterpret_cast<const f32x4*>(k + base + (long long)nx * ld + 4 * nf); vr[nf] = *reinterpret_cast<const f32x4*>(v + base + (long long)nx * ld + 4 * nf); }
            }
        }
        const int nk = L - kb < F4_KB ? L - kb : F4_KB;
        for (int k0 = 0; k0 < nk; k0 += 16) {
            f32x4 s[4];
            const f32x4 nm4 = f32x4{-m, -m, -m, -m};
#pragma unroll
            for (int g = 0; g < 4; ++g) s[g] = nm4;
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) {
                f32x4 kk[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) kk[g] = *reinterpret_cast<const f32x4*>(Kimg + (k0 + 4 * g + j4) * FW + 4 * nf);
#pragma unroll
                for (int f = 0; f < 4; ++f)
#pragma unroll
                    for (int g = 0; g < 4; ++g) s[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(kk[g][f], qv[nf][f], s[g], 0, 0, 0);
            }
            if (k0 + 16 > nk) {                                            // ragged tail of the last block (uniform branch)
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (k0 + 4 * g + r >= nk) s[g][r] = -INFINITY;
            }
            float mx = fmaxf(fmaxf(fmaxf(s[0][0], s[0][1]), s[0][2]), s[0][3]);                 // chains of three: v_max3_f32 (8 instead of 15)
#pragma unroll
            for (int g = 1; g < 4; ++g) mx = fmaxf(fmaxf(mx, fmaxf(fmaxf(s[g][0], s[g][1]), s[g][2])), s[g][3]);
            if (__builtin_amdgcn_ballot_w64(mx > F4_LAZY) != 0ull) {       // rare after the first keys: move the reference of the lanes that need it
                const float d = mx > F4_LAZY ? mx : 0.f;
                const float alpha = __builtin_amdgcn_exp2f(-d);
                m += d;
                l *= alpha;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
#pragma unroll
                    for (int nf = 0; nf < NF; ++nf) o[nf][g] *= alpha;
#pragma unroll
                    for (int r = 0; r < 4; ++r) s[g][r] -= d;
                }
            }
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int r = 0; r < 4; ++r) s[g][r] = __builtin_amdgcn_exp2f(s[g][r]);
            {
                const f32x4 t4 = (s[0] + s[1]) + (s[2] + s[3]);            // packed adds
                l += (t4[0] + t4[1]) + (t4[2] + t4[3]);
            }
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) {
                f32x4 vv[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) vv[g] = *reinterpret_cast<const f32x4*>(Vt + (4 * nf + j4) * F4_VLD + k0 + 4 * g);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int g = 0; g < 4; ++g) o[nf][g] = __builtin_amdgcn_mfma_f32_4x4x1f32(vv[g][i], s[g][i], o[nf][g], 0, 0, 0);
            }
        }
    }
    if (query >= L) return;
    const float inv = 1.f / l;
    float* dst = out + base + (long long)query * ld;
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) *reinterpret_cast<f32x4*>(dst + 4 * nf) = ((o[nf][0] + o[nf][1]) + (o[nf][2] + o[nf][3])) * inv;
    if (NF == 1) *reinterpret_cast<f32x4*>(dst + 4) = zero4;               // the padded features of the head
    if (lse) lse[(long long)blockIdx.y * L + query] = m * 0.6931471805599453f + __logf(l);     // natural-log sum-exp of the scaled scores
}

// Backward of the same heads, the same way.  dQ kernel: a lane owns a QUERY and walks the keys (256-key blocks in LDS: K rows, K columns,
// V rows): per 4 keys 4 NF instructions give its scores, 4 NF its dP = dO . V, the softmax backward dS = P (dP - delta) scale is per-lane
// arithmetic (the row's log-sum-exp and delta are the lane's own scalars), 4 NF instructions accumulate dQ += dS K.
template <int NF>
__global__ __launch_bounds__(256, 2) void flash_bwd4_dq_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                                               const float* __restrict__ dout, const float* __restrict__ lse,
                                                               const float* __restrict__ delta, float* __restrict__ dq, int L, int H, int ld,
                                                               float scale) {
    constexpr int FW = 4 * NF;
    __shared__ __attribute__((aligned(16))) float Kr[F4_KB * FW];         // [key][feature]
    __shared__ __attribute__((aligned(16))) float Vr[F4_KB * FW];
    __shared__ __attribute__((aligned(16))) float Kt[FW * F4_VLD];        // [feature][key]
    const int tid = threadIdx.x, j4 = tid & 3;
    const int n = blockIdx.y / H, h = blockIdx.y - n * H;
    const long long base = (long long)n * L * ld + h * 8;
    const int query = blockIdx.x * 256 + tid;
    const int qrow = query < L ? query : L - 1;
    const float LOG2E = 1.4426950408889634f;
    f32x4 qv[NF], gv[NF];
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) {
        qv[nf] = *reinterpret_cast<const f32x4*>(q + base + (long long)qrow * ld + 4 * nf);
        qv[nf] *= scale * LOG2E;
        gv[nf] = *reinterpret_cast<const f32x4*>(dout + base + (long long)qrow * ld + 4 * nf);
    }
    const float lse2 = lse[(long long)blockIdx.y * L + qrow] * LOG2E, dl = delta[(long long)blockIdx.y * L + qrow];
    const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
    const f32x4 nlse4 = f32x4{-lse2, -lse2, -lse2, -lse2}, ndl4 = f32x4{-dl, -dl, -dl, -dl};
    f32x4 acc[NF][4];
#pragma unroll
    for (int nf = 0; nf < NF; ++nf)
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[nf][g] = zero4;
    f32x4 kr[NF], vr[NF];
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) {
        kr[nf] = zero4; vr[nf] = zero4;
        if (tid < L) { kr[nf] = *reinterpret_cast<const f32x4*>(k + base + (long long)tid * ld + 4 * nf); vr[nf] = *reinterpret_cast<const f32x4*>(v + base + (long long)tid * ld + 4 * nf); }
    }
    for (int kb = 0; kb < L; kb += F4_KB) {
        __syncthreads();
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) {
            *reinterpret_cast<f32x4*>(Kr + tid * FW + 4 * nf) = kr[nf];
            *reinterpret_cast<f32x4*>(Vr + tid * FW + 4 * nf) = vr[nf];
#pragma unroll
            for (int f = 0; f < 4; ++f) Kt[(4 * nf + f) * F4_VLD + tid] = kr[nf][f];
        }
        __syncthreads();
        {
            const int nx = kb + F4_KB + tid;
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) {
                kr[nf] = zero4; vr[nf] = zero4;
                if (nx < L) { kr[nf] = *reinterpret_cast<const f32x4*>(k + base + (long long)nx * ld + 4 * nf); vr[nf] = *reinterpret_cast<const f32x4*>(v + base + (long long)nx * ld + 4 * nf); }
            }
        }
        const int nk = L - kb < F4_KB ? L - kb : F4_KB;
        for (int k0 = 0; k0 < nk; k0 += 16) {
            f32x4 s[4], dp[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) { s[g] = nlse4; dp[g] = ndl4; }  // the accumulators start at -lse and -delta: the subtractions are free
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) {
                f32x4 kk[4], vv[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    kk[g] = *reinterpret_cast<const f32x4*>(Kr + (k0 + 4 * g + j4) * FW + 4 * nf);
                    vv[g] = *reinterpret_cast<const f32x4*>(Vr + (k0 + 4 * g + j4) * FW + 4 * nf);
                }
#pragma unroll
                for (int f = 0; f < 4; ++f)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        s[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(kk[g][f], qv[nf][f], s[g], 0, 0, 0);
                        dp[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(vv[g][f], gv[nf][f], dp[g], 0, 0, 0);
                    }
            }
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int r = 0; r < 4; ++r) s[g][r] = __builtin_amdgcn_exp2f(s[g][r]) * dp[g][r];       // (the factor `scale` is applied once, to the sum)
            if (k0 + 16 > nk) {                                            // ragged tail: keys past the sequence contribute nothing
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (k0 + 4 * g + r >= nk) s[g][r] = 0.f;
            }
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) {
                f32x4 kt[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) kt[g] = *reinterpret_cast<const f32x4*>(Kt + (4 * nf + j4) * F4_VLD + k0 + 4 * g);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int g = 0; g < 4; ++g) acc[nf][g] = __builtin_amdgcn_mfma_f32_4x4x1f32(kt[g][i], s[g][i], acc[nf][g], 0, 0, 0);
            }
        }
    }
    if (query >= L) return;
    float* dst = dq + base + (long long)query * ld;
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) *reinterpret_cast<f32x4*>(dst + 4 * nf) = ((acc[nf][0] + acc[nf][1]) + (acc[nf][2] + acc[nf][3])) * scale;
    if (NF == 1) *reinterpret_cast<f32x4*>(dst + 4) = zero4;
}

// dK / dV kernel: a lane owns a KEY and walks the queries (256-query blocks in LDS: q and dO as rows and as columns, the rows' log-sum-exp
// and delta): per 4 queries 4 NF instructions give S^T, 4 NF give dP^T, then dV += P^T dO and dK += dS^T Q with 4 NF instructions each.
// Queries past the sequence are staged with lse = +inf: their probabilities are exactly 0, no tail branch.
template <int NF>
__global__ __launch_bounds__(256, 2) void flash_bwd4_dkv_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                                                const float* __restrict__ dout, const float* __restrict__ lse,
                                                                const float* __restrict__ delta, float* __restrict__ dk, float* __restrict__ dv, int L,
                                                                int H, int ld, float scale) {
    constexpr int FW = 4 * NF;
    __shared__ __attribute__((aligned(16))) float Qr[F4_KB * FW];         // [query][feature]
    __shared__ __attribute__((aligned(16))) float Gr[F4_KB * FW];
    __shared__ __attribute__((aligned(16))) float Qt[FW * F4_VLD];        // [feature][query]
    __shared__ __attribute__((aligned(16))) float Gt[FW * F4_VLD];
    __shared__ __attribute__((aligned(16))) float Ls[F4_KB];              // MINUS the log2-domain log-sum-exp per query
    __shared__ __attribute__((aligned(16))) float Ds[F4_KB];              // MINUS delta per query
    const int tid = threadIdx.x, j4 = tid & 3;
    const int n = blockIdx.y / H, h = blockIdx.y - n * H;
    const long long base = (long long)n * L * ld + h * 8;
    const long long sbase = (long long)blockIdx.y * L;
    const int key = blockIdx.x * 256 + tid;
    const int krow = key < L ? key : L - 1;
    const float LOG2E = 1.4426950408889634f;
    f32x4 ks[NF], vo[NF];
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) {
        ks[nf] = *reinterpret_cast<const f32x4*>(k + base + (long long)krow * ld + 4 * nf);
        ks[nf] *= scale * LOG2E;
        vo[nf] = *reinterpret_cast<const f32x4*>(v + base + (long long)krow * ld + 4 * nf);
    }
    const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 adk[NF][4], adv[NF][4];
#pragma unroll
    for (int nf = 0; nf < NF; ++nf)
#pragma unroll
        for (int g = 0; g < 4; ++g) { adk[nf][g] = zero4; adv[nf][g] = zero4; }
    f32x4 qr[NF], gr[NF];
    float lr = INFINITY, dr = 0.f;
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) {
        qr[nf] = zero4; gr[nf] = zero4;
        if (tid < L) { qr[nf] = *reinterpret_cast<const f32x4*>(q + base + (long long)tid * ld + 4 * nf); gr[nf] = *reinterpret_cast<const f32x4*>(dout + base + (long long)tid * ld + 4 * nf); }
    }
    if (tid < L) { lr = lse[sbase + tid] * LOG2E; dr = delta[sbase + tid]; }
    for (int qb = 0; qb < L; qb += F4_KB) {
        __syncthreads();
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) {
            *reinterpret_cast<f32x4*>(Qr + tid * FW + 4 * nf) = qr[nf];
            *reinterpret_cast<f32x4*>(Gr + tid * FW + 4 * nf) = gr[nf];
#pragma unroll
            for (int f = 0; f < 4; ++f) { Qt[(4 * nf + f) * F4_VLD + tid] = qr[nf][f]; Gt[(4 * nf + f) * F4_VLD + tid] = gr[nf][f]; }
        }
        Ls[tid] = -lr; Ds[tid] = -dr;
        __syncthreads();
        {
            const int nx = qb + F4_KB + tid;
            lr = INFINITY; dr = 0.f;
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) {
                qr[nf] = zero4; gr[nf] = zero4;
                if (nx < L) { qr[nf] = *reinterpret_cast<const f32x4*>(q + base + (long long)nx * ld + 4 * nf); gr[nf] = *reinterpret_cast<const f32x4*>(dout + base + (long long)nx * ld + 4 * nf); }
            }
            if (nx < L) { lr = lse[sbase + nx] * LOG2E; dr = delta[sbase + nx]; }
        }
        const int nq = L - qb < F4_KB ? L - qb : F4_KB;
        for (int q0 = 0; q0 < nq; q0 += 16) {
            f32x4 s[4], dp[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                s[g] = *reinterpret_cast<const f32x4*>(Ls + q0 + 4 * g);        // accumulators start at -lse / -delta of the four queries
                dp[g] = *reinterpret_cast<const f32x4*>(Ds + q0 + 4 * g);
            }
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) {
                f32x4 qq[4], gg[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    qq[g] = *reinterpret_cast<const f32x4*>(Qr + (q0 + 4 * g + j4) * FW + 4 * nf);
                    gg[g] = *reinterpret_cast<const f32x4*>(Gr + (q0 + 4 * g + j4) * FW + 4 * nf);
                }
#pragma unroll
                for (int f = 0; f < 4; ++f)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        s[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(qq[g][f], ks[nf][f], s[g], 0, 0, 0);
                        dp[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(gg[g][f], vo[nf][f], dp[g], 0, 0, 0);
                    }
            }
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pv = __builtin_amdgcn_exp2f(s[g][r]);                   // 0 for the padding queries (lse = +inf)
                    s[g][r] = pv;
                    dp[g][r] *= pv;                                                    // (the factor `scale` is applied once, to dK)
                }
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) {
                f32x4 qt[4], gt[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    qt[g] = *reinterpret_cast<const f32x4*>(Qt + (4 * nf + j4) * F4_VLD + q0 + 4 * g);
                    gt[g] = *reinterpret_cast<const f32x4*>(Gt + (4 * nf + j4) * F4_VLD + q0 + 4 * g);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        adv[nf][g] = __builtin_amdgcn_mfma_f32_4x4x1f32(gt[g][i], s[g][i], adv[nf][g], 0, 0, 0);
                        adk[nf][g] = __builtin_amdgcn_mfma_f32_4x4x1f32(qt[g][i], dp[g][i], adk[nf][g], 0, 0, 0);
                    }
            }
        }
    }
    if (key >= L) return;
    float* dkd = dk + base + (long long)key * ld;
    float* dvd = dv + base + (long long)key * ld;
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) {
        *reinterpret_cast<f32x4*>(dkd + 4 * nf) = ((adk[nf][0] + adk[nf][1]) + (adk[nf][2] + adk[nf][3])) * scale;
        *reinterpret_cast<f32x4*>(dvd + 4 * nf) = (adv[nf][0] + adv[nf][1]) + (adv[nf][2] + adv[nf][3]);
    }
    if (NF == 1) { *reinterpret_cast<f32x4*>(dkd + 4) = zero4; *reinterpret_cast<f32x4*>(dvd + 4) = zero4; }
}

// delta[nh][q] = sum_d dO[q][d] * O[q][d]  (the softmax-backward row term; one thread per (image, head, query))
template <typename T>
__global__ void flash_delta_kernel(const T* __restrict__ o, const T* __restrict__ dout, float* __restrict__ delta, int N, int L, int H, int dhp, int ld) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)N * H * L) return;
    const int qi = (int)(i % L);
    const int nh = (int)(i / L), n = nh / H, h = nh - n * H;
    const long long off = ((long long)n * L + qi) * ld + h * dhp;
    float s = 0.f;
    for (int d = 0; d < dhp; ++d) s += to_f32<T>(o[off + d]) * to_f32<T>(dout[off + d]);
    delta[i] = s;
}

// dQ: workgroup = 64 queries (4 waves x 16) of one (image, head); loops over 64-key blocks staged in LDS (K and V images).
template <typename T, int DTC, int FWP = 0>
__global__ __launch_bounds__(256, 2) void flash_bwd_dq_kernel(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v,
                                                              const T* __restrict__ dout, const float* __restrict__ lse,
                                                              const float* __restrict__ delta, T* __restrict__ dq, int L, int H, int dhp, int ld,
                                                              int ldi, float scale, FlashExtra ex) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const unsigned long long dseed = flash_seed(ex);
    T* Kimg = reinterpret_cast<T*>(smem_raw);
    T* Vimg = Kimg + 64 * ldi;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int p = lane & 15, g = lane >> 4;
    const int n = blockIdx.y / H, h = blockIdx.y - n * H;
    const long long base = (long long)n * L * ld + h * dhp;
    using FF = typename FeatSel<T, FWP>::F;
    constexpr int FW = FeatSel<T, FWP>::W, FL = FW / 4;
    const int DT = (dhp + 15) >> 4, ks = (dhp + FW - 1) / FW, segs = dhp >> 3;
    const int query = blockIdx.x * 64 + wave * 16 + p;
    const int qrow = query < L ? query : L - 1;
    constexpr int KSC = FWP == 8 ? 1 : (DTC * 16 + FW - 1) / FW;
    FF fq[KSC], fg[KSC];
#pragma unroll
    for (int s = 0; s < KSC; ++s)
        if (s < ks) {
            ff_row<T>(fq[s], q + base + (long long)qrow * ld, s * FW + FL * g, dhp);
            ff_row<T>(fg[s], dout + base + (long long)qrow * ld, s * FW + FL * g, dhp);
        }
    const float lse_q = lse[(long long)blockIdx.y * L + qrow], dl = delta[(long long)blockIdx.y * L + qrow];
    f32x4 acc[DTC];
#pragma unroll
    for (int dt = 0; dt < DTC; ++dt) acc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int kb = 0; kb < L; kb += 64) {
        __syncthreads();
        for (int u = tid; u < 64 * segs; u += 256) {
            const int r = u / segs, c = (u - r * segs) * 8;
            Vec8<T> a, b;
            if (kb + r < L) { a.load(k + base + (long long)(kb + r) * ld + c); b.load(v + base + (long long)(kb + r) * ld + c); }
            else { a.zero(); b.zero(); }
            a.store(Kimg + r * ldi + c);
            b.store(Vimg + r * ldi + c);
        }
        if ((dhp & 15) != 0)
            for (int u = tid; u < 64; u += 256) { Vec8<T> z; z.zero(); z.store(Kimg + u * ldi + dhp); }
        __syncthreads();
        Frag16<T> ds[4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            f32x4 sT = f32x4{0.f, 0.f, 0.f, 0.f}, dpT = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < KSC; ++s)
                if (s < ks) {
                    FF a, b;
                    ff_row<T>(a, Kimg + (kt * 16 + p) * ldi, s * FW + FL * g, dhp);
                    ff_row<T>(b, Vimg + (kt * 16 + p) * ldi, s * FW + FL * g, dhp);
                    sT = ff_mma(a, fq[s], sT);
                    dpT = ff_mma(b, fg[s], dpT);
                }
            float x[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = kb + kt * 16 + 4 * g + r;
                const bool ok = key < L;
                float sv = sT[r] * scale, dpv = dpT[r];
                if (ex.mask && ok && ex.mask[((long long)(blockIdx.y % ex.Bmask) * L + qrow) * L + key] == 0.f) sv = -1e9f;
                if (ex.p_drop > 0.f) dpv *= dropout_mult(dseed, ((unsigned long long)blockIdx.y * L + qrow) * L + key, ex.p_drop, ex.inv_keep);
                const float pv = ok ? __expf(sv - lse_q) : 0.f;
                x[r] = pv * (dpv - dl) * scale;
            }
            acc_to_frag(ds[kt], x);
        }
#pragma unroll
        for (int dt = 0; dt < DTC; ++dt)
            if (dt < DT) {
#pragma unroll
                for (int kt = 0; kt < 4; ++kt) {
                    Frag16<T> a;
                    tr_read(a, Kimg, ldi, kt * 16, dt * 16, lane);
                    acc[dt] = mma16(a, ds[kt], acc[dt]);
                }
            }
    }
#pragma unroll
    for (int dt = 0; dt < DTC; ++dt)
        if (dt < DT) {
            const int f0 = dt * 16 + 4 * g;
            if (query < L && f0 < dhp) store4(dq + base + (long long)query * ld + f0, acc[dt]);
        }
}

// dK, dV: workgroup = 64 keys (4 waves x 16) of one (image, head); loops over 64-query blocks staged in LDS (Q and dO images + lse, delta).
template <typename T, int DTC, int FWP = 0>
__global__ __launch_bounds__(256, 2) void flash_bwd_dkv_kernel(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v,
                                                               const T* __restrict__ dout, const float* __restrict__ lse,
                                                               const float* __restrict__ delta, T* __restrict__ dk, T* __restrict__ dv, int L,
                                                               int H, int dhp, int ld, int ldi, float scale, FlashExtra ex) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const unsigned long long dseed = flash_seed(ex);
    T* Qimg = reinterpret_cast<T*>(smem_raw);
    T* Gimg = Qimg + 64 * ldi;
    float* lseL = reinterpret_cast<float*>(Gimg + 64 * ldi);   // [64]
    float* delL = lseL + 64;                                   // [64]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int p = lane & 15, g = lane >> 4;
    const int n = blockIdx.y / H, h = blockIdx.y - n * H;
    const long long base = (long long)n * L * ld + h * dhp;
    using FF = typename FeatSel<T, FWP>::F;
    constexpr int FW = FeatSel<T, FWP>::W, FL = FW / 4;
    const int DT = (dhp + 15) >> 4, ks = (dhp + FW - 1) / FW, segs = dhp >> 3;
    const int key = blockIdx.x * 64 + wave * 16 + p;
    const int krow = key < L ? key : L - 1;
    constexpr int KSC = FWP == 8 ? 1 : (DTC * 16 + FW - 1) / FW;
    FF fk[KSC], fv[KSC];
#pragma unroll
    for (int s = 0; s < KSC; ++s)
        if (s < ks) {
            ff_row<T>(fk[s], k + base + (long long)krow * ld, s * FW + FL * g, dhp);
            ff_row<T>(fv[s], v + base + (long long)krow * ld, s * FW + FL * g, dhp);
        }
    f32x4 dkT[DTC], dvT[DTC];
#pragma unroll
    for (int dt = 0; dt < DTC; ++dt) { dkT[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dvT[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    for (int qb = 0; qb < L; qb += 64) {
        __syncthreads();
        for (int u = tid; u < 64 * segs; u += 256) {
            const int r = u / segs, c = (u - r * segs) * 8;
            Vec8<T> a, b;
            if (qb + r < L) { a.load(q + base + (long long)(qb + r) * ld + c); b.load(dout + base + (long long)(qb + r) * ld + c); }
            else { a.zero(); b.zero(); }
            a.store(Qimg + r * ldi + c);
            b.store(Gimg + r * ldi + c);
        }
        if ((dhp & 15) != 0)
            for (int u = tid; u < 64; u += 256) { Vec8<T> z; z.zero(); z.store(Qimg + u * ldi + dhp); z.store(Gimg + u * ldi + dhp); }
        if (tid < 64) {
            const bool ok = qb + tid < L;
            lseL[tid] = ok ? lse[(long long)blockIdx.y * L + qb + tid] : 0.f;
            delL[tid] = ok ? delta[(long long)blockIdx.y * L + qb + tid] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int qt = 0; qt < 4; ++qt) {
            f32x4 sN = f32x4{0.f, 0.f, 0.f, 0.f}, dpN = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < KSC; ++s)
                if (s < ks) {
                    FF a, b;
                    ff_row<T>(a, Qimg + (qt * 16 + p) * ldi, s * FW + FL * g, dhp);
                    ff_row<T>(b, Gimg + (qt * 16 + p) * ldi, s * FW + FL * g, dhp);
                    sN = ff_mma(a, fk[s], sN);
                    dpN = ff_mma(b, fv[s], dpN);
                }
            float ds[4], pv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ql = qt * 16 + 4 * g + r;
                const bool ok = key < L && qb + ql < L;
                float sv = sN[r] * scale, mult = 1.f;
                if (ex.mask && ok && ex.mask[((long long)(blockIdx.y % ex.Bmask) * L + qb + ql) * L + key] == 0.f) sv = -1e9f;
                if (ex.p_drop > 0.f && ok) mult = dropout_mult(dseed, ((unsigned long long)blockIdx.y * L + qb + ql) * L + key, ex.p_drop, ex.inv_keep);
                const float pu = ok ? __expf(sv - lseL[ql]) : 0.f;                 // undropped softmax weight
                ds[r] = pu * (dpN[r] * mult - delL[ql]) * scale;
                pv[r] = pu * mult;                                                 // the weight that multiplied V
            }
            Frag16<T> dsF, pF;
            acc_to_frag(dsF, ds);
            acc_to_frag(pF, pv);
#pragma unroll
            for (int dt = 0; dt < DTC; ++dt)
                if (dt < DT) {
                    Frag16<T> aq, ag;
                    tr_read(aq, Qimg, ldi, qt * 16, dt * 16, lane);
                    tr_read(ag, Gimg, ldi, qt * 16, dt * 16, lane);
                    dkT[dt] = mma16(aq, dsF, dkT[dt]);
                    dvT[dt] = mma16(ag, pF, dvT[dt]);
                }
        }
    }
#pragma unroll
    for (int dt = 0; dt < DTC; ++dt)
        if (dt < DT) {
            const int f0 = dt * 16 + 4 * g;
            if (key < L && f0 < dhp) {
                store4(dk + base + (long long)key * ld + f0, dkT[dt]);
                store4(dv + base + (long long)key * ld + f0, dvT[dt]);
            }
        }
}

}  // namespace

// Internal (fct.hip): out = softmax(q k^T * scale) v per (image, head), L tokens, heads of padded width dhp at stride ld
static int flash_fwd_impl(int dtype, const void* q, const void* k, const void* v, void* out, float* lse, int N, int L, int H, int dhp, int ld,
                          float scale, hipStream_t st, int dh_true, const FlashExtra& ex) {
    if (!q || !k || !v || !out || N < 1 || L < 1 || H < 1 || dhp < 8 || dhp % 8 != 0 || dhp > 16 * MAXDT || ld % 8 != 0 || (long long)N * H > 65535) return HYB_E_ARG;
    static const int f4_env = getenv("HYB_FLASH_FWD4") ? atoi(getenv("HYB_FLASH_FWD4")) : 1;
    if (dtype == HYB_F32 && f4_env && dhp == 8 && dh_true >= 1 && dh_true <= 8 && L >= 1024) {
        // heads of <= 8 features: the 4x4x1 matrix instruction, a query per lane (for <= 4 features the quad 4..7 of q, k, v is zero padding)
        if (dh_true <= 4) hipLaunchKernelGGL(flash_fwd4_kernel<1>, dim3(hyb_cdiv(L, 256), N * H), dim3(256), 0, st, (const float*)q, (const float*)k, (const float*)v,
                                             (float*)out, lse, L, H, ld, scale * 1.4426950408889634f);
        else hipLaunchKernelGGL(flash_fwd4_kernel<2>, dim3(hyb_cdiv(L, 256), N * H), dim3(256), 0, st, (const float*)q, (const float*)k, (const float*)v,
                           (float*)out, lse, L, H, ld, scale * 1.4426950408889634f);
        HYB_LAUNCH_CHECK();
        return 0;
    }
    const int es = dtype == HYB_F32 ? 4 : 2;
    int bytes = ((dhp + 15) / 16) * 16 * es;
    if ((bytes / 32) % 2 == 0) bytes += 32;
    const int ldi = bytes / es;
    const size_t lds = (size_t)2 * 64 * ldi * es;
    const dim3 grid(hyb_cdiv(L, 64), N * H);
    if (dtype == HYB_F32) {
        if (lds > 64 * 1024) { static HybAttrOnce once; if (int e = hyb_set_lds_attr(once, (const void*)flash_fwd_kernel<float>, 160 * 1024)) return e; }
        if (dhp == 8) hipLaunchKernelGGL((flash_fwd_kernel<float, 8>), grid, dim3(256), lds, st, (const float*)q, (const float*)k, (const float*)v, (float*)out, lse, L, H, dhp, ld, ldi, scale, ex);
        else hipLaunchKernelGGL(flash_fwd_kernel<float>, grid, dim3(256), lds, st, (const float*)q, (const float*)k, (const float*)v, (float*)out, lse, L, H, dhp, ld, ldi, scale, ex);
    } else if (dtype == HYB_BF16) {
        hipLaunchKernelGGL(flash_fwd_kernel<bf16>, grid, dim3(256), lds, st, (const bf16*)q, (const bf16*)k, (const bf16*)v, (bf16*)out, lse, L, H, dhp, ld, ldi, scale, ex);
    } else return HYB_E_ARG;
    HYB_LAUNCH_CHECK();
    return 0;
}

// Internal (fct_bwd.hip): gradients of the long-sequence attention core.  lse from the forward pass; delta_ws: N*H*L floats of scratch.
static int flash_bwd_impl(int dtype, const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse, float* delta_ws,
                          void* dq, void* dk, void* dv, int N, int L, int H, int dhp, int ld, float scale, hipStream_t st, int dh_true, const FlashExtra& ex) {
    if (!q || !k || !v || !o || !dout || !lse || !delta_ws || !dq || !dk || !dv || N < 1 || L < 1 || H < 1 || dhp < 8 || dhp % 8 != 0 ||
        dhp > 16 * MAXDT || ld % 8 != 0 || (long long)N * H > 65535 || dtype != HYB_F32) return HYB_E_ARG;      // fp32 only so far (FCT runs in fp32)
    const int es = 4;
    int bytes = ((dhp + 15) / 16) * 16 * es;
    if ((bytes / 32) % 2 == 0) bytes += 32;
    const int ldi = bytes / es;
    const size_t lds = (size_t)2 * 64 * ldi * es + 2 * 64 * sizeof(float);
    if (lds > 64 * 1024) {                                       // 128-wide fp32 heads (the temporal encoder's long sequences): 2 x 64 x 136 x 4 B = 68 KiB
        static HybAttrOnce once_dq, once_dkv;
        if (int e = hyb_set_lds_attr(once_dq, (const void*)flash_bwd_dq_kernel<float, 8>, 160 * 1024)) return e;
        if (int e = hyb_set_lds_attr(once_dkv, (const void*)flash_bwd_dkv_kernel<float, 8>, 160 * 1024)) return e;
    }
    const long long nq = (long long)N * H * L;
    hipLaunchKernelGGL(flash_delta_kernel<float>, dim3(hyb_cdiv(nq, 256)), dim3(256), 0, st, (const float*)o, (const float*)dout, delta_ws, N, L, H, dhp, ld);
    static const int f4_env = getenv("HYB_FLASH_BWD4") ? atoi(getenv("HYB_FLASH_BWD4")) : 1;
    if (f4_env && dhp == 8 && dh_true >= 1 && dh_true <= 8 && L >= 1024) {      // heads of <= 8 features: the 4x4x1 matrix instruction
        const dim3 grid4(hyb_cdiv(L, 256), N * H);
        HybProfileHook* hook = hyb_find_hook(5, L, H);           // measurement hook (hyb_profile_set): kernel 5 = the dK / dV kernel of this path, keyed by (tokens, heads)
        if (dh_true <= 4) {
            hipLaunchKernelGGL(flash_bwd4_dq_kernel<1>, grid4, dim3(256), 0, st, (const float*)q, (const float*)k, (const float*)v, (const float*)dout, lse,
                               (const float*)delta_ws, (float*)dq, L, H, ld, scale);
            if (hook) hipEventRecord(hook->ev0, st);
            hipLaunchKernelGGL(flash_bwd4_dkv_kernel<1>, grid4, dim3(256), 0, st, (const float*)q, (const float*)k, (const float*)v, (const float*)dout, lse,
                               (const float*)delta_ws, (float*)dk, (float*)dv, L, H, ld, scale);
        } else {
            hipLaunchKernelGGL(flash_bwd4_dq_kernel<2>, grid4, dim3(256), 0, st, (const float*)q, (const float*)k, (const float*)v, (const float*)dout, lse,
                               (const float*)delta_ws, (float*)dq, L, H, ld, scale);
            if (hook) hipEventRecord(hook->ev0, st);
            hipLaunchKernelGGL(flash_bwd4_dkv_kernel<2>, grid4, dim3(256), 0, st, (const float*)q, (const float*)k, (const float*)v, (const float*)dout, lse,
                               (const float*)delta_ws, (float*)dk, (float*)dv, L, H, ld, scale);
        }
        if (hook) hipEventRecord(hook->ev1, st);
        HYB_LAUNCH_CHECK();
        return 0;
    }
    const dim3 grid(hyb_cdiv(L, 64), N * H);
#define FLASH_BWD(DTC_) do { \
        hipLaunchKernelGGL((flash_bwd_dq_kernel<float, DTC_>), grid, dim3(256), lds, st, (const float*)q, (const float*)k, (const float*)v, (const float*)dout, lse, \
                           (const float*)delta_ws, (float*)dq, L, H, dhp, ld, ldi, scale, ex); \
        hipLaunchKernelGGL((flash_bwd_dkv_kernel<float, DTC_>), grid, dim3(256), lds, st, (const float*)q, (const float*)k, (const float*)v, (const float*)dout, lse, \
                           (const float*)delta_ws, (float*)dk, (float*)dv, L, H, dhp, ld, ldi, scale, ex); } while (0)
    const int DT = (dhp + 15) / 16;
    if (dhp == 8) {
        hipLaunchKernelGGL((flash_bwd_dq_kernel<float, 1, 8>), grid, dim3(256), lds, st, (const float*)q, (const float*)k, (const float*)v, (const float*)dout, lse,
                           (const float*)delta_ws, (float*)dq, L, H, dhp, ld, ldi, scale, ex);
        hipLaunchKernelGGL((flash_bwd_dkv_kernel<float, 1, 8>), grid, dim3(256), lds, st, (const float*)q, (const float*)k, (const float*)v, (const float*)dout, lse,
                           (const float*)delta_ws, (float*)dk, (float*)dv, L, H, dhp, ld, ldi, scale, ex);
    } else if (DT <= 1) FLASH_BWD(1); else if (DT <= 2) FLASH_BWD(2); else if (DT <= 4) FLASH_BWD(4); else FLASH_BWD(8);
#undef FLASH_BWD
    HYB_LAUNCH_CHECK();
    return 0;
}

// Internal (fct.hip / fct_bwd.hip): FCT's nn.MultiheadAttention core -- no mask, no dropout
int hyb_flash_attention_fwd(int dtype, const void* q, const void* k, const void* v, void* out, float* lse, int N, int L, int H, int dhp, int ld,
                            float scale, hipStream_t st, int dh_true) {
    return flash_fwd_impl(dtype, q, k, v, out, lse, N, L, H, dhp, ld, scale, st, dh_true, FlashExtra{});
}
int hyb_flash_attention_bwd(int dtype, const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse, float* delta_ws,
                            void* dq, void* dk, void* dv, int N, int L, int H, int dhp, int ld, float scale, hipStream_t st, int dh_true) {
    return flash_bwd_impl(dtype, q, k, v, o, dout, lse, delta_ws, dq, dk, dv, N, L, H, dhp, ld, scale, st, dh_true, FlashExtra{});
}

// ---------------------------------------------------------------------------------------------------------------------------
// MultiheadAttention.attention (TransformerEncoder.pyc src L49-62) for sequences of ANY length: the reference has no limit on S, the
// register-resident kernels above hold at most 64 keys per query.  Longer sequences take the online-softmax kernels with the reference's
// scale (1/sqrt(input_dim), quirk Q1), mask (Q4) and attention-weight dropout (Q5).  The core runs on fp32 operands: a bf16 caller's
// q, k, v (and, backward, the saved output and its gradient) are widened into the workspace and the results rounded once on the way out.
// q, k, v: [B*S] rows at stride ld_qkv elements (D for separate tensors, 3 D for the encoder's packed q|k|v), head h = features [h*dh, (h+1)*dh).
// ---------------------------------------------------------------------------------------------------------------------------
namespace {
template <typename TS, typename TD>
__global__ void cast_rows_kernel(const TS* __restrict__ src, long long lds, TD* __restrict__ dst, long long ldd, int M, int D) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)M * D) return;
    const long long r = i / D, c = i - r * D;
    dst[r * ldd + c] = from_f32<TD>(to_f32<TS>(src[r * lds + c]));
}
inline size_t al256(size_t v) { return (v + 255) & ~(size_t)255; }
inline bool long_dims_ok(int B, int S, int D, int H) {
    return B > 0 && S > 0 && D > 0 && H > 0 && D % H == 0 && (D / H) % 8 == 0 && D / H <= 16 * MAXDT && (long long)B * H <= 65535;
}
}  // namespace

extern "C" size_t hyb_attention_long_workspace(int dtype, int B, int S, int D, int H) {
    (void)dtype;
    if (!long_dims_ok(B, S, D, H)) return 0;
    const size_t M = (size_t)B * S;
    return al256((size_t)B * H * S * sizeof(float)) + 8 * al256(M * D * sizeof(float));     // delta + q, k, v, out, dout, dq, dk, dv as dense fp32 [M][D]
}

namespace {
template <typename T>
int long_fwd_t(const T* q, const T* k, const T* v, int ld_qkv, T* out, float* lse, int B, int S, int D, int H, const FlashExtra& ex, char* ws, hipStream_t st) {
    const int M = B * S, dh = D / H;
    const float scale = 1.0f / sqrtf((float)D);                                // quirk Q1: sqrt(input_dim), not sqrt(head width)
    if (sizeof(T) == 4 && ld_qkv == D) return flash_fwd_impl(HYB_F32, q, k, v, out, lse, B, S, H, dh, D, scale, st, 0, ex);
    // the kernels address q, k, v and out with one row stride: packed q|k|v (stride 3 D) and bf16 operands go through dense fp32 copies
    char* w = ws + al256((size_t)B * H * S * sizeof(float));
    const size_t one = al256((size_t)M * D * sizeof(float));
    float* f[4];
    for (int i = 0; i < 4; ++i) f[i] = (float*)(w + i * one);
    const dim3 g(hyb_cdiv((long long)M * D, 256));
    hipLaunchKernelGGL((cast_rows_kernel<T, float>), g, dim3(256), 0, st, q, (long long)ld_qkv, f[0], (long long)D, M, D);
    hipLaunchKernelGGL((cast_rows_kernel<T, float>), g, dim3(256), 0, st, k, (long long)ld_qkv, f[1], (long long)D, M, D);
    hipLaunchKernelGGL((cast_rows_kernel<T, float>), g, dim3(256), 0, st, v, (long long)ld_qkv, f[2], (long long)D, M, D);
    if (int rc = flash_fwd_impl(HYB_F32, f[0], f[1], f[2], f[3], lse, B, S, H, dh, D, scale, st, 0, ex)) return rc;
    hipLaunchKernelGGL((cast_rows_kernel<float, T>), g, dim3(256), 0, st, (const float*)f[3], (long long)D, out, (long long)D, M, D);
    HYB_LAUNCH_CHECK();
    return 0;
}
template <typename T>
int long_bwd_t(const T* q, const T* k, const T* v, int ld_qkv, const T* out, const float* lse, const T* dout, T* dq, T* dk, T* dv, int ld_d, int B, int S,
               int D, int H, const FlashExtra& ex, char* ws, hipStream_t st) {
    const int M = B * S, dh = D / H;
    const float scale = 1.0f / sqrtf((float)D);
    float* delta = (float*)ws;
    if (sizeof(T) == 4 && ld_qkv == D && ld_d == D)
        return flash_bwd_impl(HYB_F32, q, k, v, out, dout, lse, delta, dq, dk, dv, B, S, H, dh, D, scale, st, 0, ex);
    char* w = ws + al256((size_t)B * H * S * sizeof(float));
    const size_t one = al256((size_t)M * D * sizeof(float));
    float* f[8];
    for (int i = 0; i < 8; ++i) f[i] = (float*)(w + i * one);
    const dim3 g(hyb_cdiv((long long)M * D, 256));
    hipLaunchKernelGGL((cast_rows_kernel<T, float>), g, dim3(256), 0, st, q, (long long)ld_qkv, f[0], (long long)D, M, D);
    hipLaunchKernelGGL((cast_rows_kernel<T, float>), g, dim3(256), 0, st, k, (long long)ld_qkv, f[1], (long long)D, M, D);
    hipLaunchKernelGGL((cast_rows_kernel<T, float>), g, dim3(256), 0, st, v, (long long)ld_qkv, f[2], (long long)D, M, D);
    hipLaunchKernelGGL((cast_rows_kernel<T, float>), g, dim3(256), 0, st, out, (long long)D, f[3], (long long)D, M, D);
    hipLaunchKernelGGL((cast_rows_kernel<T, float>), g, dim3(256), 0, st, dout, (long long)D, f[4], (long long)D, M, D);
    if (int rc = flash_bwd_impl(HYB_F32, f[0], f[1], f[2], f[3], f[4], lse, delta, f[5], f[6], f[7], B, S, H, dh, D, scale, st, 0, ex)) return rc;
    hipLaunchKernelGGL((cast_rows_kernel<float, T>), g, dim3(256), 0, st, (const float*)f[5], (long long)D, dq, (long long)ld_d, M, D);
    hipLaunchKernelGGL((cast_rows_kernel<float, T>), g, dim3(256), 0, st, (const float*)f[6], (long long)D, dk, (long long)ld_d, M, D);
    hipLaunchKernelGGL((cast_rows_kernel<float, T>), g, dim3(256), 0, st, (const float*)f[7], (long long)D, dv, (long long)ld_d, M, D);
    HYB_LAUNCH_CHECK();
    return 0;
}
}  // namespace

extern "C" int hyb_attention_long_fwd(int dtype, const void* q, const void* k, const void* v, int ld_qkv, const float* mask, void* out, float* lse, int B,
                                      int S, int D, int H, float p_drop, unsigned long long seed, const unsigned long long* seed_inc, void* workspace,
                                      size_t workspace_bytes, void* stream) {
    HYB_CHECK_ARG(q && k && v && out && lse && workspace && long_dims_ok(B, S, D, H) && ld_qkv >= D && ld_qkv % 8 == 0 && p_drop >= 0.f && p_drop < 1.f);
    HYB_CHECK_ARG(dtype == HYB_F32 || dtype == HYB_BF16);
    if (workspace_bytes < hyb_attention_long_workspace(dtype, B, S, D, H)) return HYB_E_WORKSPACE;
    const FlashExtra ex{mask, B, p_drop, p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f, seed, seed_inc};
    if (dtype == HYB_F32) return long_fwd_t<float>((const float*)q, (const float*)k, (const float*)v, ld_qkv, (float*)out, lse, B, S, D, H, ex, (char*)workspace, (hipStream_t)stream);
    return long_fwd_t<bf16>((const bf16*)q, (const bf16*)k, (const bf16*)v, ld_qkv, (bf16*)out, lse, B, S, D, H, ex, (char*)workspace, (hipStream_t)stream);
}

extern "C" int hyb_attention_long_bwd(int dtype, const void* q, const void* k, const void* v, int ld_qkv, const float* mask, const void* out, const float* lse,
                                      const void* dout, void* dq, void* dk, void* dv, int ld_d, int B, int S, int D, int H, float p_drop,
                                      unsigned long long seed, const unsigned long long* seed_inc, void* workspace, size_t workspace_bytes, void* stream) {
    HYB_CHECK_ARG(q && k && v && out && lse && dout && dq && dk && dv && workspace && long_dims_ok(B, S, D, H) && ld_qkv >= D && ld_qkv % 8 == 0 &&
                  ld_d >= D && ld_d % 8 == 0 && p_drop >= 0.f && p_drop < 1.f);
    HYB_CHECK_ARG(dtype == HYB_F32 || dtype == HYB_BF16);
    if (workspace_bytes < hyb_attention_long_workspace(dtype, B, S, D, H)) return HYB_E_WORKSPACE;
    const FlashExtra ex{mask, B, p_drop, p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f, seed, seed_inc};
    if (dtype == HYB_F32)
        return long_bwd_t<float>((const float*)q, (const float*)k, (const float*)v, ld_qkv, (const float*)out, lse, (const float*)dout, (float*)dq, (float*)dk,
                                 (float*)dv, ld_d, B, S, D, H, ex, (char*)workspace, (hipStream_t)stream);
    return long_bwd_t<bf16>((const bf16*)q, (const bf16*)k, (const bf16*)v, ld_qkv, (const bf16*)out, lse, (const bf16*)dout, (bf16*)dq, (bf16*)dk, (bf16*)dv, ld_d,
                            B, S, D, H, ex, (char*)workspace, (hipStream_t)stream);
}
