// Temporal attention core of MultiheadAttention.attention (TransformerEncoder.pyc src L49-62) with the head
// split/merge of __reshape_to_batches__/__reshape_from_batches__ (src L22-45) done as index math.
//
// One workgroup (4 waves) per (batch, head) problem; S <= 64 tokens, so Q, K, V, the score matrix and P all
// live in LDS.  QK^T, P.V and the five backward products run on the matrix cores (16x16 tiles, k32 steps);
// the softmax row reduce is a 16-lane wave shuffle.  Reference quirks kept: scale 1/sqrt(d_model) (Q1),
// mask row b*H+h reads mask[(b*H+h) % B] (Q4), masked_fill(-1e9), dropout on the weights (src L58).
#include "hyb_common.h"

namespace {

struct AttnDims {
    int B, S, D, H, dh;
    int SK;      // S rounded up to 32 (token dimension padding)
    int dhp;     // dh rounded up to 32 (feature padding for the k32 step)
    int ld_qkv;  // token row stride (elements) of q, k, v and dq, dk, dv (D, or 3D when they are packed as [M][3D])
    int ld_o;    // token row stride of out / dout
};

// dst[s][c] = src[(b*S + s)*D + h*dh + c], zero padded to [SK][dhp]; row stride ld
template <typename T>
__device__ __forceinline__ void stage_rows(T* dst, int ld, const T* src, const AttnDims& d, int b, int h, int tid, int gld) {
    const int segs = d.dhp >> 3;
    for (int u = tid; u < d.SK * segs; u += 256) {
        const int s = u / segs, c = (u - s * segs) * 8;
        Vec8<T> v;
        if (s < d.S && c < d.dh) v.load(src + ((long long)(b * d.S + s)) * gld + h * d.dh + c);
        else v.zero();
        v.store(dst + s * ld + c);
    }
}
// dst[c][s] = src[(b*S + s)*D + h*dh + c], zero padded to [dhp][SK]; row stride ld
template <typename T>
__device__ __forceinline__ void stage_cols(T* dst, int ld, const T* src, const AttnDims& d, int b, int h, int tid, int gld) {
    const int segs = d.dhp >> 3;
    for (int u = tid; u < d.SK * segs; u += 256) {
        const int s = u / segs, c = (u - s * segs) * 8;
        Vec8<T> v;
        if (s < d.S && c < d.dh) v.load(src + ((long long)(b * d.S + s)) * gld + h * d.dh + c);
        else v.zero();
#pragma unroll
        for (int j = 0; j < 8; ++j) dst[(c + j) * ld + s] = from_f32<T>(v.get(j));
    }
}

// one 16x16 output tile: rows from Arows (row-major, k contiguous), cols from Brows
template <typename T>
__device__ __forceinline__ f32x4 mm_tile(const T* Arows, int lda, const T* Brows, int ldb, int kdim, int lane) {
    const int p = lane & 15, q = lane >> 4;
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < kdim; k0 += 32) {
        Frag<T> a, b;
        frag_load(a, Arows + p * lda + k0 + 8 * q);
        frag_load(b, Brows + p * ldb + k0 + 8 * q);
        acc = mma32(a, b, acc);
    }
    return acc;
}

template <typename T>
__global__ __launch_bounds__(256) void attention_fwd_kernel(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v,
                                                            const float* __restrict__ mask, T* __restrict__ out, float* __restrict__ probs,
                                                            AttnDims d, float scale, float p_drop, unsigned long long seed) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int ldq = d.dhp + 8, lds_ = d.SK + 8, ldsc = d.SK + 4;
    T* Ql = reinterpret_cast<T*>(smem_raw);
    T* Kl = Ql + d.SK * ldq;
    T* Vt = Kl + d.SK * ldq;            // [dhp][SK+8]
    T* Pl = Vt + d.dhp * lds_;          // [SK][SK+8]
    float* Sc = reinterpret_cast<float*>(Pl + d.SK * lds_);   // [SK][SK+4]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pidx = blockIdx.x, b = pidx / d.H, h = pidx % d.H;
    const int p = lane & 15, qq = lane >> 4;

    stage_rows(Ql, ldq, q, d, b, h, tid, d.ld_qkv);
    stage_rows(Kl, ldq, k, d, b, h, tid, d.ld_qkv);
    stage_cols(Vt, lds_, v, d, b, h, tid, d.ld_qkv);
    __syncthreads();

    const int nt = d.SK >> 4;
    const float* mrow = mask ? mask + (long long)(pidx % d.B) * d.S * d.S : nullptr;
    for (int t = wave; t < nt * nt; t += 4) {
        const int ti = t / nt, tj = t % nt;
        const f32x4 acc = mm_tile(Ql + ti * 16 * ldq, ldq, Kl + tj * 16 * ldq, ldq, d.dhp, lane);
        const int col = tj * 16 + p;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = ti * 16 + 4 * qq + r;
            float sv = acc[r] * scale;
            if (mrow && row < d.S && col < d.S && mrow[row * d.S + col] == 0.f) sv = -1e9f;
            if (col >= d.S) sv = -INFINITY;
            Sc[row * ldsc + col] = sv;
        }
    }
    __syncthreads();

    // softmax: one row per 16-lane group
    const float inv_keep = p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f;
    for (int row = wave * 4 + qq; row < d.SK; row += 16) {
        float vals[4];
        float mx = -INFINITY;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int col = p + 16 * c;
            vals[c] = col < d.SK ? Sc[row * ldsc + col] : -INFINITY;
            mx = fmaxf(mx, vals[c]);
        }
        mx = group16_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) { vals[c] = __expf(vals[c] - mx); sum += vals[c]; }
        sum = group16_sum(sum);
        const float inv = 1.f / sum;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int col = p + 16 * c;
            if (col >= d.SK) continue;
            float pv = vals[c] * inv;
            if (row < d.S && col < d.S) {
                probs[((long long)pidx * d.S + row) * d.S + col] = pv;
                if (p_drop > 0.f) pv *= dropout_mult(seed, ((unsigned long long)pidx * d.S + row) * d.S + col, p_drop, inv_keep);
            } else {
                pv = 0.f;
            }
            Pl[row * lds_ + col] = from_f32<T>(pv);
        }
    }
    __syncthreads();

    const int ntd = d.dhp >> 4;
    for (int t = wave; t < nt * ntd; t += 4) {
        const int ti = t / ntd, tj = t % ntd;
        const f32x4 acc = mm_tile(Pl + ti * 16 * lds_, lds_, Vt + tj * 16 * lds_, lds_, d.SK, lane);
        const int col = tj * 16 + p;
        if (col < d.dh) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = ti * 16 + 4 * qq + r;
                if (row < d.S) out[((long long)(b * d.S + row)) * d.ld_o + h * d.dh + col] = from_f32<T>(acc[r]);
            }
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void attention_bwd_kernel(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v,
                                                            const float* __restrict__ probs, const T* __restrict__ dout,
                                                            T* __restrict__ dq, T* __restrict__ dk, T* __restrict__ dv,
                                                            AttnDims d, float scale, float p_drop, unsigned long long seed) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int ldq = d.dhp + 8, lds_ = d.SK + 8, ldsc = d.SK + 4;
    const int regA = 2 * d.SK * ldq, regB = d.dhp * lds_;
    T* R1 = reinterpret_cast<T*>(smem_raw);                 // phase A: dO | V ; phase C: transposed operand
    T* dSl = R1 + (regA > regB ? regA : regB);              // [SK][SK+8]  dS (scaled)
    T* dSt = dSl + d.SK * lds_;                             // [SK][SK+8]  dS^T
    T* PdT = dSt + d.SK * lds_;                             // [SK][SK+8]  (P*dropout)^T
    float* Sc = reinterpret_cast<float*>(PdT + d.SK * lds_);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pidx = blockIdx.x, b = pidx / d.H, h = pidx % d.H;
    const int p = lane & 15, qq = lane >> 4;
    const int nt = d.SK >> 4, ntd = d.dhp >> 4;

    // ---- phase A: dPd = dO V^T
    T* dOl = R1;
    T* Vl = R1 + d.SK * ldq;
    stage_rows(dOl, ldq, dout, d, b, h, tid, d.ld_o);
    stage_rows(Vl, ldq, v, d, b, h, tid, d.ld_qkv);
    __syncthreads();
    for (int t = wave; t < nt * nt; t += 4) {
        const int ti = t / nt, tj = t % nt;
        const f32x4 acc = mm_tile(dOl + ti * 16 * ldq, ldq, Vl + tj * 16 * ldq, ldq, d.dhp, lane);
#pragma unroll
        for (int r = 0; r < 4; ++r) Sc[(ti * 16 + 4 * qq + r) * ldsc + tj * 16 + p] = acc[r];
    }
    __syncthreads();

    // ---- phase B: softmax backward, one row per 16-lane group
    const float inv_keep = p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f;
    for (int row = wave * 4 + qq; row < d.SK; row += 16) {
        float pv[4], dp[4], mult[4];
        float delta = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int col = p + 16 * c;
            pv[c] = 0.f; dp[c] = 0.f; mult[c] = 1.f;
            if (row < d.S && col < d.S) {
                const long long idx = ((long long)pidx * d.S + row) * d.S + col;
                pv[c] = probs[idx];
                if (p_drop > 0.f) mult[c] = dropout_mult(seed, (unsigned long long)idx, p_drop, inv_keep);
                dp[c] = Sc[row * ldsc + col] * mult[c];
                delta += dp[c] * pv[c];
            }
        }
        delta = group16_sum(delta);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int col = p + 16 * c;
            if (col >= d.SK) continue;
            const float ds = pv[c] * (dp[c] - delta) * scale;
            const float pd = pv[c] * mult[c];
            dSl[row * lds_ + col] = from_f32<T>(ds);
            dSt[col * lds_ + row] = from_f32<T>(ds);
            PdT[col * lds_ + row] = from_f32<T>(pd);
        }
    }
    __syncthreads();

    // ---- phase C1: dV[key][d] = sum_query PdT[key][query] * dO^T[d][query]
    stage_cols(R1, lds_, dout, d, b, h, tid, d.ld_o);
    __syncthreads();
    for (int t = wave; t < nt * ntd; t += 4) {
        const int ti = t / ntd, tj = t % ntd;
        const f32x4 acc = mm_tile(PdT + ti * 16 * lds_, lds_, R1 + tj * 16 * lds_, lds_, d.SK, lane);
        const int col = tj * 16 + p;
        if (col < d.dh) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = ti * 16 + 4 * qq + r;
                if (row < d.S) dv[((long long)(b * d.S + row)) * d.ld_qkv + h * d.dh + col] = from_f32<T>(acc[r]);
            }
        }
    }
    __syncthreads();
    // ---- phase C2: dQ[query][d] = sum_key dS[query][key] * K^T[d][key]
    stage_cols(R1, lds_, k, d, b, h, tid, d.ld_qkv);
    __syncthreads();
    for (int t = wave; t < nt * ntd; t += 4) {
        const int ti = t / ntd, tj = t % ntd;
        const f32x4 acc = mm_tile(dSl + ti * 16 * lds_, lds_, R1 + tj * 16 * lds_, lds_, d.SK, lane);
        const int col = tj * 16 + p;
        if (col < d.dh) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = ti * 16 + 4 * qq + r;
                if (row < d.S) dq[((long long)(b * d.S + row)) * d.ld_qkv + h * d.dh + col] = from_f32<T>(acc[r]);
            }
        }
    }
    __syncthreads();
    // ---- phase C3: dK[key][d] = sum_query dS^T[key][query] * Q^T[d][query]
    stage_cols(R1, lds_, q, d, b, h, tid, d.ld_qkv);
    __syncthreads();
    for (int t = wave; t < nt * ntd; t += 4) {
        const int ti = t / ntd, tj = t % ntd;
        const f32x4 acc = mm_tile(dSt + ti * 16 * lds_, lds_, R1 + tj * 16 * lds_, lds_, d.SK, lane);
        const int col = tj * 16 + p;
        if (col < d.dh) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = ti * 16 + 4 * qq + r;
                if (row < d.S) dk[((long long)(b * d.S + row)) * d.ld_qkv + h * d.dh + col] = from_f32<T>(acc[r]);
            }
        }
    }
}

inline bool attn_dims(AttnDims& d, int B, int S, int D, int H) {
    if (B <= 0 || S <= 0 || S > 64 || D <= 0 || H <= 0 || D % H != 0) return false;
    d.B = B; d.S = S; d.D = D; d.H = H; d.dh = D / H;
    if (d.dh % 8 != 0 || d.dh > 128) return false;
    d.SK = (S + 31) / 32 * 32;
    d.dhp = (d.dh + 31) / 32 * 32;
    d.ld_qkv = D;
    d.ld_o = D;
    return true;
}

template <typename T>
size_t attn_fwd_lds(const AttnDims& d) {
    const size_t ldq = d.dhp + 8, lds_ = d.SK + 8, ldsc = d.SK + 4;
    return (2 * d.SK * ldq + d.dhp * lds_ + d.SK * lds_) * sizeof(T) + d.SK * ldsc * sizeof(float);
}
template <typename T>
size_t attn_bwd_lds(const AttnDims& d) {
    const size_t ldq = d.dhp + 8, lds_ = d.SK + 8, ldsc = d.SK + 4;
    const size_t regA = 2 * d.SK * ldq, regB = d.dhp * lds_;
    return ((regA > regB ? regA : regB) + 3 * d.SK * lds_) * sizeof(T) + d.SK * ldsc * sizeof(float);
}

template <typename T>
int attn_fwd_t(const void* q, const void* k, const void* v, const float* mask, void* out, float* probs, const AttnDims& d, float p_drop,
               unsigned long long seed, hipStream_t st) {
    const size_t lds = attn_fwd_lds<T>(d);
    if (lds > 160 * 1024) return HYB_E_ARG;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)attention_fwd_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);   // size varies per call: not cached
        if (e != hipSuccess) return (int)e;
    }
    const float scale = 1.0f / sqrtf((float)d.D);
    hipLaunchKernelGGL(attention_fwd_kernel<T>, dim3(d.B * d.H), dim3(256), lds, st, (const T*)q, (const T*)k, (const T*)v, mask, (T*)out,
                       probs, d, scale, p_drop, seed);
    HYB_LAUNCH_CHECK();
    return 0;
}
template <typename T>
int attn_bwd_t(const void* q, const void* k, const void* v, const float* probs, const void* dout, void* dq, void* dk, void* dv,
               const AttnDims& d, float p_drop, unsigned long long seed, hipStream_t st) {
    const size_t lds = attn_bwd_lds<T>(d);
    if (lds > 160 * 1024) return HYB_E_ARG;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)attention_bwd_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    const float scale = 1.0f / sqrtf((float)d.D);
    hipLaunchKernelGGL(attention_bwd_kernel<T>, dim3(d.B * d.H), dim3(256), lds, st, (const T*)q, (const T*)k, (const T*)v, probs,
                       (const T*)dout, (T*)dq, (T*)dk, (T*)dv, d, scale, p_drop, seed);
    HYB_LAUNCH_CHECK();
    return 0;
}

}  // namespace

extern "C" int hyb_attention_fwd(int dtype, const void* q, const void* k, const void* v, const float* mask, void* out, float* probs, int B,
                                 int S, int D, int H, float p_drop, unsigned long long seed, void* stream) {
    AttnDims d;
    HYB_CHECK_ARG(q && k && v && out && probs && attn_dims(d, B, S, D, H) && p_drop >= 0.f && p_drop < 1.f);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == HYB_F32) return attn_fwd_t<float>(q, k, v, mask, out, probs, d, p_drop, seed, st);
    if (dtype == HYB_BF16) return attn_fwd_t<bf16>(q, k, v, mask, out, probs, d, p_drop, seed, st);
    return HYB_E_ARG;
}

extern "C" int hyb_attention_bwd(int dtype, const void* q, const void* k, const void* v, const float* probs, const void* dout, void* dq,
                                 void* dk, void* dv, int B, int S, int D, int H, float p_drop, unsigned long long seed, void* stream) {
    AttnDims d;
    HYB_CHECK_ARG(q && k && v && probs && dout && dq && dk && dv && attn_dims(d, B, S, D, H) && p_drop >= 0.f && p_drop < 1.f);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == HYB_F32) return attn_bwd_t<float>(q, k, v, probs, dout, dq, dk, dv, d, p_drop, seed, st);
    if (dtype == HYB_BF16) return attn_bwd_t<bf16>(q, k, v, probs, dout, dq, dk, dv, d, p_drop, seed, st);
    return HYB_E_ARG;
}

// Internal (same shared object): q/k/v (and dq/dk/dv) packed as [B*S][3D] -- used by hyb_encoder_{fwd,bwd}
int hyb_attention_fwd_packed(int dtype, const void* qkv, const float* mask, void* out, float* probs, int B, int S, int D, int H, float p_drop,
                             unsigned long long seed, hipStream_t st) {
    AttnDims d;
    if (!qkv || !out || !probs || !attn_dims(d, B, S, D, H)) return HYB_E_ARG;
    d.ld_qkv = 3 * D;
    const size_t es = dtype == HYB_F32 ? 4 : 2;
    const char* base = (const char*)qkv;
    if (dtype == HYB_F32) return attn_fwd_t<float>(base, base + D * es, base + 2 * D * es, mask, out, probs, d, p_drop, seed, st);
    if (dtype == HYB_BF16) return attn_fwd_t<bf16>(base, base + D * es, base + 2 * D * es, mask, out, probs, d, p_drop, seed, st);
    return HYB_E_ARG;
}
int hyb_attention_bwd_packed(int dtype, const void* qkv, const float* probs, const void* dout, void* dqkv, int B, int S, int D, int H,
                             float p_drop, unsigned long long seed, hipStream_t st) {
    AttnDims d;
    if (!qkv || !probs || !dout || !dqkv || !attn_dims(d, B, S, D, H)) return HYB_E_ARG;
    d.ld_qkv = 3 * D;
    const size_t es = dtype == HYB_F32 ? 4 : 2;
    const char* base = (const char*)qkv;
    char* g = (char*)dqkv;
    if (dtype == HYB_F32) return attn_bwd_t<float>(base, base + D * es, base + 2 * D * es, probs, dout, g, g + D * es, g + 2 * D * es, d, p_drop, seed, st);
    if (dtype == HYB_BF16) return attn_bwd_t<bf16>(base, base + D * es, base + 2 * D * es, probs, dout, g, g + D * es, g + 2 * D * es, d, p_drop, seed, st);
    return HYB_E_ARG;
}
