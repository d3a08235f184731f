// LayerNorm + residual rows for a GEMM's prologue (linear.hip: gemm_nt_ln_kernel).  The arithmetic of one row is ln_fwd_row's
// (layernorm.hip), expression for expression -- y = dropout_p((LayerNorm(x) * gamma + beta + skip) * out_scale), statistics by wave
// reductions -- so the rows a GEMM normalises for itself carry the bits the stand-alone kernel writes.
#pragma once
#include "hyb_common.h"

// NR consecutive token rows by one wave, every global load of all of them issued before the first wait (one memory round trip); the results go
// to an LDS image (row lrow0 + r at img + (lrow0 + r) * img_ld) and, when y / stats are given, to global memory as ln_residual_fwd_kernel
// writes them.  Rows >= M repeat row M - 1 into the image (defined operands for the MFMA tiles) and write nothing to global memory.
template <typename T, int MAXC, int NR>
__device__ __forceinline__ void ln_fwd_rows_lds(const T* __restrict__ x, const T* __restrict__ skip, const float* __restrict__ gamma,
                                                const float* __restrict__ beta, T* __restrict__ y, float* __restrict__ stats, T* img, int img_ld,
                                                int M, int row0, int lrow0, int D, float eps, float out_scale, float p_drop, unsigned long long seed,
                                                const unsigned long long* __restrict__ seed_inc, int lane) {
    const int nchunk = D >> 3;
    Vec8<T> xv[NR][MAXC], sk[NR][MAXC];
    Vec8<float> gm[MAXC], bt[MAXC];
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int ch = lane + 64 * c;
        if (ch < nchunk) {
            gm[c].load(gamma + ch * 8);
            bt[c].load(beta + ch * 8);
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const int row = row0 + r < M ? row0 + r : M - 1;
                xv[r][c].load(x + (long long)row * D + ch * 8);
                sk[r][c].load(skip + (long long)row * D + ch * 8);
            }
        }
    }
    unsigned long long inc = 0;
    if (p_drop > 0.f && seed_inc) inc = *seed_inc;
    seed += inc;
    const float inv_keep = p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int row = row0 + r < M ? row0 + r : M - 1;
        const bool live = row0 + r < M;
        float sum = 0.f;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            const int ch = lane + 64 * c;
            if (ch < nchunk) {
#pragma unroll
                for (int j = 0; j < 8; ++j) sum += xv[r][c].get(j);
            }
        }
        const float mean = wave_sum(sum) / (float)D;
        float var = 0.f;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            const int ch = lane + 64 * c;
            if (ch < nchunk) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { const float dlt = xv[r][c].get(j) - mean; var += dlt * dlt; }
            }
        }
        var = wave_sum(var) / (float)D;
        const float rstd = rsqrtf(var + eps);
        if (stats && live && lane == 0) { stats[row] = mean; stats[M + row] = rstd; }
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            const int ch = lane + 64 * c;
            if (ch < nchunk) {
                Vec8<T> o;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int col = ch * 8 + j;
                    float v = ((xv[r][c].get(j) - mean) * rstd * gm[c].get(j) + bt[c].get(j) + sk[r][c].get(j)) * out_scale;
                    if (p_drop > 0.f) v *= dropout_mult(seed, (unsigned long long)row * D + col, p_drop, inv_keep);
                    o.set(j, v);
                }
                if (y && live) o.store(y + (long long)row * D + ch * 8);
                o.store(img + (long long)(lrow0 + r) * img_ld + ch * 8);
            }
        }
    }
}
