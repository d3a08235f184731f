// LayerNorm + residual epilogue of TransformerEncoder.forward (TransformerEncoder.pyc src L116-117, L120-123):
//   y = dropout_p( (LayerNorm(x) * gamma + beta + skip) * out_scale )
// One wave per token row (D <= 2048), 8 features per lane per chunk, statistics by wave shuffles.
// Also the classifier head (mean over T + Linear) and the cross-entropy loss: tiny, one kernel each.
#include "hyb_common.h"

namespace {

constexpr int LN_MAXC = 4;        // chunks of 8 features per lane: D <= 64*8*4 = 2048

// Latency is all this kernel has (128 rows of 512 features): every global load of a row -- x, skip, gamma, beta and the dropout counter --
// is issued before the first wait, so the kernel pays ONE memory round trip (the first version loaded skip / gamma / beta after the two
// reductions and the counter before anything else: three dependent round trips, 4.2 us hot; reductions by DPP, not ds_bpermute).
// One token row by one wave (the body of ln_residual_fwd_kernel; also called per row by temporal_tail_fwd_kernel, which is why the two
// produce the same bits).  ycopy: optional second destination (an LDS image of the row), same values.
template <typename T>
__device__ __forceinline__ void ln_fwd_row(const T* __restrict__ x, const T* __restrict__ skip, const float* __restrict__ gamma,
                                           const float* __restrict__ beta, T* __restrict__ y, T* ycopy, float* __restrict__ stats, int M, int row,
                                           int D, float eps, float out_scale, float p_drop, unsigned long long seed,
                                           const unsigned long long* __restrict__ seed_inc, int lane) {
    const int nchunk = D >> 3;
    Vec8<T> xv[LN_MAXC], sk[LN_MAXC];
    Vec8<float> gm[LN_MAXC], bt[LN_MAXC];
#pragma unroll
    for (int c = 0; c < LN_MAXC; ++c) {
        const int ch = lane + 64 * c;
        if (ch < nchunk) {
            xv[c].load(x + (long long)row * D + ch * 8);
            sk[c].load(skip + (long long)row * D + ch * 8);
            gm[c].load(gamma + ch * 8);
            bt[c].load(beta + ch * 8);
        }
    }
    unsigned long long inc = 0;
    if (p_drop > 0.f && seed_inc) inc = *seed_inc;              // device-side step counter: the same captured launch draws a fresh mask per replay
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < LN_MAXC; ++c) {
        const int ch = lane + 64 * c;
        if (ch < nchunk) {
#pragma unroll
            for (int j = 0; j < 8; ++j) sum += xv[c].get(j);
        }
    }
    const float mean = wave_sum(sum) / (float)D;
    float var = 0.f;
#pragma unroll
    for (int c = 0; c < LN_MAXC; ++c) {
        const int ch = lane + 64 * c;
        if (ch < nchunk) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float dlt = xv[c].get(j) - mean; var += dlt * dlt; }
        }
    }
    var = wave_sum(var) / (float)D;
    const float rstd = rsqrtf(var + eps);
    if (lane == 0) { stats[row] = mean; stats[M + row] = rstd; }
    const float inv_keep = p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f;
    seed += inc;
#pragma unroll
    for (int c = 0; c < LN_MAXC; ++c) {
        const int ch = lane + 64 * c;
        if (ch < nchunk) {
            Vec8<T> o;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int col = ch * 8 + j;
                float v = ((xv[c].get(j) - mean) * rstd * gm[c].get(j) + bt[c].get(j) + sk[c].get(j)) * out_scale;
                if (p_drop > 0.f) v *= dropout_mult(seed, (unsigned long long)row * D + col, p_drop, inv_keep);
                o.set(j, v);
            }
            o.store(y + (long long)row * D + ch * 8);
            if (ycopy) o.store(ycopy + ch * 8);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void ln_residual_fwd_kernel(const T* __restrict__ x, const T* __restrict__ skip,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              T* __restrict__ y, float* __restrict__ stats, int M, int D, float eps,
                                                              float out_scale, float p_drop, unsigned long long seed, const unsigned long long* __restrict__ seed_inc) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    ln_fwd_row<T>(x, skip, gamma, beta, y, nullptr, stats, M, row, D, eps, out_scale, p_drop, seed, seed_inc, lane);
}

// One token row of the LayerNorm + residual backward by one wave (the loop body of ln_residual_bwd_kernel; temporal_tail_bwd_kernel calls it
// with the row's upstream gradient in LDS: DY_LDS, dyl[D] floats holding T-rounded values -- the same for every token of a clip).
template <typename T, bool DY_LDS, bool XPRE = false>
__device__ __forceinline__ void ln_bwd_row(const T* __restrict__ dy, const float* dyl, const T* __restrict__ x, const Vec8<float> (&gmv)[LN_MAXC],
                                           const float* __restrict__ stats, T* __restrict__ dx, T* __restrict__ dskip, int accumulate_dskip,
                                           float (&dg)[LN_MAXC][8], float (&db)[LN_MAXC][8], int M, int row, int D, float out_scale, float p_drop,
                                           unsigned long long seed, float inv_keep, int lane, const Vec8<T>* xpre = nullptr) {
    // XPRE: the caller loaded the row of x already (xpre[LN_MAXC]), before it formed dyl -- one memory round trip instead of two
    const int nchunk = D >> 3;
    Vec8<T> dvv[LN_MAXC], xvv[LN_MAXC], dsv[LN_MAXC];
#pragma unroll
    for (int c = 0; c < LN_MAXC; ++c) {                    // every load of the row before the first wait
        const int ch = lane + 64 * c;
        if (ch < nchunk) {
            if (DY_LDS) {
#pragma unroll
                for (int j = 0; j < 8; ++j) dvv[c].set(j, dyl[ch * 8 + j]);
            } else dvv[c].load(dy + (long long)row * D + ch * 8);
            if (XPRE) xvv[c] = xpre[c]; else xvv[c].load(x + (long long)row * D + ch * 8);
            if (accumulate_dskip) dsv[c].load(dskip + (long long)row * D + ch * 8);
        }
    }
    const float mean = stats[row], rstd = stats[M + row];
    float gl[LN_MAXC][8], xh[LN_MAXC][8];      // g = d(ln_out) * gamma ; xhat
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < LN_MAXC; ++c) {
        const int ch = lane + 64 * c;
        if (ch < nchunk) {
            const Vec8<T>& dv = dvv[c];
            const Vec8<T>& xv = xvv[c];
            Vec8<T> ds = dsv[c];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int col = ch * 8 + j;
                float d = dv.get(j) * out_scale;
                if (p_drop > 0.f) d *= dropout_mult(seed, (unsigned long long)row * D + col, p_drop, inv_keep);
                const float xhat = (xv.get(j) - mean) * rstd;
                xh[c][j] = xhat;
                dg[c][j] += d * xhat;
                db[c][j] += d;
                const float g = d * gmv[c].get(j);
                gl[c][j] = g;
                s1 += g;
                s2 += g * xhat;
                ds.set(j, accumulate_dskip ? ds.get(j) + d : d);
            }
            ds.store(dskip + (long long)row * D + ch * 8);
        }
    }
    s1 = wave_sum(s1) / (float)D;
    s2 = wave_sum(s2) / (float)D;
#pragma unroll
    for (int c = 0; c < LN_MAXC; ++c) {
        const int ch = lane + 64 * c;
        if (ch < nchunk) {
            Vec8<T> o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o.set(j, rstd * (gl[c][j] - s1 - xh[c][j] * s2));
            o.store(dx + (long long)row * D + ch * 8);
        }
    }
}

// ROWS = false: dgamma/dbeta += block sums (float atomics; the public per-op entry point).  ROWS = true (encoder composite): the
// block writes its sums as one partial row dgamma[blockIdx][2][D] (fixed order inside the block), summed later in a fixed order.
template <typename T, bool ROWS>
__global__ __launch_bounds__(256) void ln_residual_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x, const float* __restrict__ gamma,
                                                              const float* __restrict__ stats, T* __restrict__ dx, T* __restrict__ dskip,
                                                              int accumulate_dskip, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                              int M, int D, float out_scale, float p_drop, unsigned long long seed, const unsigned long long* __restrict__ seed_inc) {
    const int lane = threadIdx.x & 63;
    const int nchunk = D >> 3;
    const int total_waves = gridDim.x * 4;
    // gamma is row-independent: loaded once, together with the first row's operands (one memory round trip; see the forward kernel)
    Vec8<float> gmv[LN_MAXC];
#pragma unroll
    for (int c = 0; c < LN_MAXC; ++c) { const int ch = lane + 64 * c; if (ch < nchunk) gmv[c].load(gamma + ch * 8); }
    bool seeded = false;
    float dg[LN_MAXC][8], db[LN_MAXC][8];
#pragma unroll
    for (int c = 0; c < LN_MAXC; ++c)
#pragma unroll
        for (int j = 0; j < 8; ++j) { dg[c][j] = 0.f; db[c][j] = 0.f; }
    const float inv_keep = p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f;
    for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < M; row += total_waves) {
        if (!seeded) { if (p_drop > 0.f && seed_inc) seed += *seed_inc; seeded = true; }      // device-side step counter (fresh mask per replay)
        ln_bwd_row<T, false>(dy, nullptr, x, gmv, stats, dx, dskip, accumulate_dskip, dg, db, M, row, D, out_scale, p_drop, seed, inv_keep, lane);
    }
    extern __shared__ float lnred[];
    if (ROWS) {
        // per-wave slots [4][2][D] (single owner per entry), then the four waves in a fixed order
        float* mine = lnred + (threadIdx.x >> 6) * 2 * D;
#pragma unroll
        for (int c = 0; c < LN_MAXC; ++c) {
            const int ch = lane + 64 * c;
            if (ch < nchunk) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { mine[ch * 8 + j] = dg[c][j]; mine[D + ch * 8 + j] = db[c][j]; }
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < 2 * D; i += blockDim.x)
            dgamma[(long long)blockIdx.x * 2 * D + i] = (lnred[i] + lnred[2 * D + i]) + (lnred[4 * D + i] + lnred[6 * D + i]);
        return;
    }
    // combine the block's 4 waves in LDS, then one atomic per feature per block
    for (int i = threadIdx.x; i < 2 * D; i += blockDim.x) lnred[i] = 0.f;
    __syncthreads();
#pragma unroll
    for (int c = 0; c < LN_MAXC; ++c) {
        const int ch = lane + 64 * c;
        if (ch < nchunk) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                atomicAdd(&lnred[ch * 8 + j], dg[c][j]);
                atomicAdd(&lnred[D + ch * 8 + j], db[c][j]);
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < D; i += blockDim.x) {
        atomicAdd(dgamma + i, lnred[i]);
        atomicAdd(dbeta + i, lnred[D + i]);
    }
}

// [G rows][2][D] partial rows -> dgamma[D], dbeta[D] (overwritten), fixed order
__global__ __launch_bounds__(1024) void ln_rows_reduce_kernel(const float* __restrict__ part, int G, int D, float* __restrict__ dgamma,
                                                              float* __restrict__ dbeta) {
    long long i; float v;
    if (!rows_reduce_1024(part, G, 2LL * D, i, v)) return;
    if (i < D) dgamma[i] = v; else dbeta[i - D] = v;
}

// mean over the S tokens of feature d of clip b: the loads are issued eight at a time (independent), the additions keep the
// order of a plain loop
template <typename T>
__device__ __forceinline__ float token_mean(const T* __restrict__ x, int b, int S, int D, int d) {
    const T* col = x + (long long)b * S * D + d;
    float s = 0.f;
    int t = 0;
    for (; t + 8 <= S; t += 8) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = to_f32<T>(col[(long long)(t + j) * D]);
#pragma unroll
        for (int j = 0; j < 8; ++j) s += v[j];
    }
    for (; t < S; ++t) s += to_f32<T>(col[(long long)t * D]);
    return s / (float)S;
}

// ---- head: logits[b][c] = bias[c] + sum_d mean_s(x[b][s][d]) * W[c][d] ------------------------------
template <typename T>
__global__ __launch_bounds__(256) void head_fwd_kernel(const T* __restrict__ x, const float* __restrict__ W, const float* __restrict__ bias,
                                                       float* __restrict__ logits, int S, int D, int C) {
    extern __shared__ float pooled[];          // [D]
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // this wave's first class: its weight row does not depend on the tokens -- requested before the token means (a latency-bound kernel:
    // one memory round trip fewer on the critical path), up to 8 x 64 features in registers
    float w0[8];
    const bool pre = wave < C && D <= 512;
#pragma unroll
    for (int j = 0; j < 8; ++j) w0[j] = (pre && lane + 64 * j < D) ? W[(long long)wave * D + lane + 64 * j] : 0.f;
    for (int d = threadIdx.x; d < D; d += blockDim.x) pooled[d] = token_mean(x, b, S, D, d);
    __syncthreads();
    for (int c = wave; c < C; c += 4) {
        float s = 0.f;
        if (pre && c == wave) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { const int d = lane + 64 * j; if (d < D) s = fmaf(pooled[d], w0[j], s); }      // same order and rounding as head_logit
        } else {
            for (int d = lane; d < D; d += 64) s = fmaf(pooled[d], W[(long long)c * D + d], s);
        }
        s = wave_sum(s);
        if (lane == 0) logits[b * C + c] = s + (bias ? bias[c] : 0.f);
    }
}
// dx[b][s][d] = (1/S) sum_c dlogits[b][c] W[c][d]
template <typename T>
__global__ void head_bwd_dx_kernel(const float* __restrict__ W, const float* __restrict__ dlogits, T* __restrict__ dx, int S, int D, int C) {
    const int b = blockIdx.x;
    for (int d = threadIdx.x; d < D; d += blockDim.x) {
        float s = 0.f;
        for (int c = 0; c < C; ++c) s = fmaf(dlogits[b * C + c], W[(long long)c * D + d], s);      // (explicit: the same rounding in every kernel that forms this row)
        const T v = from_f32<T>(s / (float)S);
        for (int t = 0; t < S; ++t) dx[((long long)b * S + t) * D + d] = v;
    }
}
// dW[c][d] = sum_b dlogits[b][c] * mean_s x[b][s][d];  db[c] = sum_b dlogits[b][c]
// grid (D/64, C-chunks of 16), 256 threads = 64 features x 4 clip groups (clips b = group, group + 4, ...); the groups are
// combined through LDS in a fixed order
// (with dx_rider: the blocks y >= ceil(C/16) of the same launch form dx, the head's input gradient -- one launch for the whole head backward)
template <typename T>
__global__ __launch_bounds__(256) void head_bwd_dw_kernel(const T* __restrict__ x, const float* __restrict__ dlogits, float* __restrict__ dW,
                                                          float* __restrict__ db, int B, int S, int D, int C, const float* __restrict__ W,
                                                          T* __restrict__ dx_rider, int yblocks) {
    __shared__ float red[4][16][64];
    if ((int)blockIdx.y >= yblocks) {                      // dx[b][s][d] = (1/S) sum_c dlogits[b][c] W[c][d] for this block's 64 features
        const int d = blockIdx.x * 64 + (threadIdx.x & 63);
        if (d >= D) return;
        for (int b = ((int)blockIdx.y - yblocks) * 4 + (threadIdx.x >> 6); b < B; b += 4 * ((int)gridDim.y - yblocks)) {
            float s = 0.f;
            for (int c = 0; c < C; ++c) s = fmaf(dlogits[b * C + c], W[(long long)c * D + d], s);
            const T v = from_f32<T>(s / (float)S);
            for (int t = 0; t < S; ++t) dx_rider[((long long)b * S + t) * D + d] = v;
        }
        return;
    }
    const int dl = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int d = blockIdx.x * 64 + dl;
    const int c0 = blockIdx.y * 16;
    float acc[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) acc[c] = 0.f;
    if (d < D) {
        for (int b = grp; b < B; b += 4) {
            const float s = token_mean(x, b, S, D, d);
#pragma unroll
            for (int c = 0; c < 16; ++c)
                if (c0 + c < C) acc[c] += dlogits[b * C + c0 + c] * s;
        }
    }
#pragma unroll
    for (int c = 0; c < 16; ++c) red[grp][c][dl] = acc[c];
    __syncthreads();
    for (int i = threadIdx.x; i < 16 * 64; i += 256) {
        const int c = i >> 6, dd = i & 63;
        if (c0 + c < C && blockIdx.x * 64 + dd < D)
            dW[(long long)(c0 + c) * D + blockIdx.x * 64 + dd] = (red[0][c][dd] + red[1][c][dd]) + (red[2][c][dd] + red[3][c][dd]);
    }
    if (db && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < C) {
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += dlogits[b * C + threadIdx.x];
        db[threadIdx.x] = s;
    }
}

// ---- cross entropy (mean over the batch) --------------------------------------------------------------
// one clip's loss term from its C logits (any address space); shared by ce_fwd_kernel and the fused temporal tail
__device__ __forceinline__ float ce_clip_loss(const float* lg, long long t, int C) {
    float mx = -INFINITY;
    for (int c = 0; c < C; ++c) mx = fmaxf(mx, lg[c]);
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += expf(lg[c] - mx);
    // an out-of-range class index (torch raises a device assert) poisons the loss instead of reading out of bounds
    return (t >= 0 && t < C) ? (logf(s) + mx) - lg[(int)t] : NAN;
}
// one clip's d(loss)/d(logits[c]) for every class c, g = dloss / B
__device__ __forceinline__ float ce_clip_dlogit(const float* lg, long long t, int C, int c, float g) {
    float mx = -INFINITY;
    for (int k = 0; k < C; ++k) mx = fmaxf(mx, lg[k]);
    float s = 0.f;
    for (int k = 0; k < C; ++k) s += expf(lg[k] - mx);
    const bool ok = t >= 0 && t < C;                          // out of range: NaN gradient row (see ce_clip_loss)
    return ok ? g * (expf(lg[c] - mx) / s - (c == (int)t ? 1.f : 0.f)) : NAN;
}
// mean of the per-clip terms: threads 0..255 of the calling workgroup (every thread of it must call), thread t owns clips t, t+256, ..;
// a fixed tree -- the same one whether the terms were just computed (ce_fwd_kernel) or come from other workgroups (temporal tail)
__device__ __forceinline__ void ce_tree_mean(float acc, float* part /* LDS [256] */, float* __restrict__ loss, int B) {
    if (threadIdx.x < 256) part[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) part[threadIdx.x] += part[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss[0] = part[0] / (float)B;
}
__global__ void ce_fwd_kernel(const float* __restrict__ logits, const long long* __restrict__ target, float* __restrict__ loss, int B, int C) {
    __shared__ float part[256];
    float acc = 0.f;
    for (int b = threadIdx.x; b < B; b += blockDim.x) acc += ce_clip_loss(logits + (long long)b * C, target[b], C);
    ce_tree_mean(acc, part, loss, B);
}
__global__ void ce_bwd_kernel(const float* __restrict__ logits, const long long* __restrict__ target, const float* __restrict__ dloss,
                              float* __restrict__ dlogits, int B, int C) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float g = dloss[0] / (float)B;
    const long long t = target[b];
    for (int c = 0; c < C; ++c) dlogits[b * C + c] = ce_clip_dlogit(logits + (long long)b * C, t, C, c, g);
}

// ---- the tail of the temporal part as ONE launch each way --------------------------------------------------------------------
// forward: the last encoder layer's second LayerNorm (+ residual, scale, dropout: src L120-123) -> mean over the clip's tokens -> Linear
// head -> (when a target is given) the clip's cross-entropy term, and the batch mean by the last workgroup to finish.  One workgroup per
// clip, a wave per token row.  Every per-row / per-clip expression is the device function the stand-alone kernels call (ln_fwd_row,
// token_mean's order, head_logit, ce_clip_loss, ce_tree_mean): the fused launch produces the bits of the four launches it replaces.
// A dependent launch costs >= 4.6 us on this chip whatever it computes: these were 4 x ~5 us for 8 x 512 x 8 numbers.
__device__ __forceinline__ float head_logit(const float* pooled, const float* __restrict__ Wrow, int D, int lane) {
    float s = 0.f;
    for (int d = lane; d < D; d += 64) s = fmaf(pooled[d], Wrow[d], s);
    return wave_sum(s);
}

template <typename T>
__global__ __launch_bounds__(512) void temporal_tail_fwd_kernel(const T* __restrict__ f, const T* __restrict__ x1, const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, T* __restrict__ enc_out, float* __restrict__ stats,
                                                                 int B, int S, int D, float eps, float out_scale, float p_drop, unsigned long long seed,
                                                                 const unsigned long long* __restrict__ seed_inc, const float* __restrict__ W,
                                                                 const float* __restrict__ bias, float* __restrict__ logits, int C,
                                                                 const long long* __restrict__ target, float* __restrict__ loss,
                                                                 float* __restrict__ ce_scratch, int rows_in_lds) {
    extern __shared__ __attribute__((aligned(16))) unsigned char tail_smem[];
    float* pooled = reinterpret_cast<float*>(tail_smem);                    // [D]
    float* lg = pooled + D;                                                  // [64] this clip's logits
    float* part = lg + 64;                                                   // [256] loss tree
    T* rows = reinterpret_cast<T*>(part + 256);                              // [S][D] when rows_in_lds
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
    const int M = B * S;
    // this wave's first class: its weight row does not depend on the tokens -- requested before the LayerNorm rows (a latency-bound kernel: one
    // memory round trip fewer on the critical path), up to 8 x 64 features in registers
    float w0[8];
    const bool pre = wave < C && D <= 512;
#pragma unroll
    for (int j = 0; j < 8; ++j) w0[j] = (pre && lane + 64 * j < D) ? W[(long long)wave * D + lane + 64 * j] : 0.f;
    const float bias0 = (pre && bias) ? bias[wave] : 0.f;
    for (int s = wave; s < S; s += nwaves)
        ln_fwd_row<T>(f, x1, gamma, beta, enc_out, rows_in_lds ? rows + (long long)s * D : nullptr, stats, M, b * S + s, D, eps, out_scale, p_drop,
                      seed, seed_inc, lane);
    __syncthreads();                                                         // (also makes the rows just stored to enc_out readable by this workgroup)
    for (int d = tid; d < D; d += blockDim.x) {
        if (rows_in_lds) {
            float s = 0.f;
            for (int t = 0; t < S; ++t) s += to_f32<T>(rows[(long long)t * D + d]);      // token_mean's order
            pooled[d] = s / (float)S;
        } else pooled[d] = token_mean(enc_out, b, S, D, d);
    }
    __syncthreads();
    for (int c = wave; c < C; c += nwaves) {
        float s;
        if (pre && c == wave) {
            s = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) { const int d = lane + 64 * j; if (d < D) s = fmaf(pooled[d], w0[j], s); }      // head_logit's order and rounding
            s = wave_sum(s);
        } else s = head_logit(pooled, W + (long long)c * D, D, lane);
        if (lane == 0) { const float v = s + ((pre && c == wave) ? bias0 : (bias ? bias[c] : 0.f)); logits[b * C + c] = v; lg[c] = v; }
    }
    if (!target) return;
    __syncthreads();
    // ce_scratch: [B] per-clip terms, then one ticket word (zero at rest).  The term is published by an agent-scope atomic store (performed at
    // the device's coherence point, not in this XCD's L2), which has completed when vmcnt reaches 0 -- only then is the ticket taken; the last
    // workgroup reads the terms with agent-scope atomic loads.  (The release / acquire pair of the memory model would be an L2 write-back +
    // invalidate per workgroup: ~3 us each on this chip, see optim.hip -- as much as the launch this fusion removes.)
    __shared__ int s_last;
    if (tid == 0) {
        __hip_atomic_store(ce_scratch + b, ce_clip_loss(lg, target[b], C), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned int* ticket = reinterpret_cast<unsigned int*>(ce_scratch + B);
        const unsigned int t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = t == gridDim.x - 1;
        if (s_last) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (!s_last) return;
    float acc = 0.f;
    if (tid < 256)
        for (int i = tid; i < B; i += 256) acc += __hip_atomic_load(ce_scratch + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ce_tree_mean(acc, part, loss, B);
}

// backward: cross-entropy backward (from the saved logits, when a target is given; else the caller's dlogits) -> head backward (the
// clip's token-gradient row v = dlogits W / S; its weight / bias gradient terms as one partial row per clip) -> LayerNorm backward of the
// clip's token rows, whose upstream gradient is v for every token.  Grid (row blocks per clip, clips), four waves, a wave per token row:
// every workgroup of a clip forms the clip's dlogits and v for itself (C x D multiply-adds), the clip's head terms are split among them by
// column.  Everything a workgroup reads -- its rows of the LayerNorm input, the head weights, the token means -- is requested up front: one
// memory round trip.  Writes dx (d LN input), dskip and ln_rows affine-gradient partial rows as ln_residual_bwd_kernel<T, true> does for
// the launch it replaces (workgroup (sb, b) writes row b * nsb + sb; rows no workgroup owns are zero-filled).
template <typename T>
__global__ __launch_bounds__(256) void temporal_tail_bwd_kernel(const float* __restrict__ dlogits_in, const float* __restrict__ logits,
                                                                const long long* __restrict__ target, const float* __restrict__ dloss,
                                                                const float* __restrict__ W, const T* __restrict__ enc_out, const T* __restrict__ f,
                                                                const float* __restrict__ gamma, const float* __restrict__ stats, T* __restrict__ dx,
                                                                T* __restrict__ dskip, float* __restrict__ ln_part, int ln_rows,
                                                                float* __restrict__ head_part, int B, int S, int D, int C, int rpw, float out_scale,
                                                                float p_drop, unsigned long long seed, const unsigned long long* __restrict__ seed_inc) {
    extern __shared__ __attribute__((aligned(16))) unsigned char tail_smem[];
    float* dl = reinterpret_cast<float*>(tail_smem);                         // [64]
    float* v = dl + 64;                                                      // [D] the clip's token-gradient row (T-rounded values)
    float* lnred = v + D;                                                    // [4][2][D]
    const int sb = blockIdx.x, nsb = gridDim.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int M = B * S, nchunk = D >> 3;
    const int s_begin = sb * rpw, s_end = s_begin + rpw < S ? s_begin + rpw : S;
    // ---- requests first
    Vec8<float> gmv[LN_MAXC];
    Vec8<T> xpre[LN_MAXC];
    const int s0 = s_begin + wave;                                           // this wave's first row
#pragma unroll
    for (int c = 0; c < LN_MAXC; ++c) {
        const int ch = lane + 64 * c;
        if (ch < nchunk) {
            gmv[c].load(gamma + ch * 8);
            if (s0 < s_end) xpre[c].load(f + (long long)(b * S + s0) * D + ch * 8);
        }
    }
    const int d_first = sb + tid * nsb;                                      // this thread's first head column: its token mean does not wait for dl
    const float pooled_first = d_first < D ? token_mean(enc_out, b, S, D, d_first) : 0.f;
    if (tid < C) dl[tid] = target ? ce_clip_dlogit(logits + (long long)b * C, target[b], C, tid, dloss[0] / (float)B) : dlogits_in[b * C + tid];
    if (p_drop > 0.f && seed_inc) seed += *seed_inc;
    __syncthreads();
    for (int d = tid; d < D; d += 256) {
        float s = 0.f;
        for (int c = 0; c < C; ++c) s = fmaf(dl[c], W[(long long)c * D + d], s);           // head_bwd_dx_kernel's expression
        v[d] = to_f32<T>(from_f32<T>(s / (float)S));
    }
    // the clip's head terms, columns [sb, sb + nsb, ...) of this workgroup
    float* hp = head_part + (long long)b * ((long long)C * D + C);
    for (int d = d_first; d < D; d += 256 * nsb) {
        const float pooled = d == d_first ? pooled_first : token_mean(enc_out, b, S, D, d);
        for (int c = 0; c < C; ++c) hp[(long long)c * D + d] = dl[c] * pooled;
    }
    if (sb == 0 && tid < C) hp[(long long)C * D + tid] = dl[tid];
    __syncthreads();
    float dg[LN_MAXC][8], db[LN_MAXC][8];
#pragma unroll
    for (int c = 0; c < LN_MAXC; ++c)
#pragma unroll
        for (int j = 0; j < 8; ++j) { dg[c][j] = 0.f; db[c][j] = 0.f; }
    const float inv_keep = p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f;
    if (s0 < s_end) ln_bwd_row<T, true, true>(nullptr, v, f, gmv, stats, dx, dskip, 0, dg, db, M, b * S + s0, D, out_scale, p_drop, seed, inv_keep, lane, xpre);
    for (int s = s0 + 4; s < s_end; s += 4)
        ln_bwd_row<T, true, false>(nullptr, v, f, gmv, stats, dx, dskip, 0, dg, db, M, b * S + s, D, out_scale, p_drop, seed, inv_keep, lane);
    // the workgroup's partial row: per-wave slots, then the four waves in a fixed order (as ln_residual_bwd_kernel<T, true>)
    float* mine = lnred + wave * 2 * D;
#pragma unroll
    for (int c = 0; c < LN_MAXC; ++c) {
        const int ch = lane + 64 * c;
        if (ch < nchunk) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { mine[ch * 8 + j] = dg[c][j]; mine[D + ch * 8 + j] = db[c][j]; }
        }
    }
    __syncthreads();
    const int prow = b * nsb + sb, nrows = B * nsb;
    for (int i = tid; i < 2 * D; i += 256) {
        ln_part[(long long)prow * 2 * D + i] = (lnred[i] + lnred[2 * D + i]) + (lnred[4 * D + i] + lnred[6 * D + i]);
        for (int r = nrows + prow; r < ln_rows; r += nrows) ln_part[(long long)r * 2 * D + i] = 0.f;
    }
}

}  // namespace

int hyb_ln_residual_fwd_inc(int dtype, const void* x, const void* skip, const float* gamma, const float* beta, void* y, float* stats,
                            int M, int D, float eps, float out_scale, float p_drop, unsigned long long seed, const unsigned long long* seed_inc,
                            void* stream);
extern "C" int hyb_ln_residual_fwd(int dtype, const void* x, const void* skip, const float* gamma, const float* beta, void* y, float* stats,
                                   int M, int D, float eps, float out_scale, float p_drop, unsigned long long seed, void* stream) {
    return hyb_ln_residual_fwd_inc(dtype, x, skip, gamma, beta, y, stats, M, D, eps, out_scale, p_drop, seed, nullptr, stream);
}
int hyb_ln_residual_fwd_inc(int dtype, const void* x, const void* skip, const float* gamma, const float* beta, void* y, float* stats,
                            int M, int D, float eps, float out_scale, float p_drop, unsigned long long seed, const unsigned long long* seed_inc,
                            void* stream) {
    HYB_CHECK_ARG(x && skip && gamma && beta && y && stats && M > 0 && D > 0 && D % 8 == 0 && D <= 64 * 8 * LN_MAXC && p_drop >= 0.f && p_drop < 1.f);
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(hyb_cdiv(M, 4));
    if (dtype == HYB_F32)
        hipLaunchKernelGGL(ln_residual_fwd_kernel<float>, grid, dim3(256), 0, st, (const float*)x, (const float*)skip, gamma, beta, (float*)y, stats, M, D, eps, out_scale, p_drop, seed, seed_inc);
    else if (dtype == HYB_BF16)
        hipLaunchKernelGGL(ln_residual_fwd_kernel<bf16>, grid, dim3(256), 0, st, (const bf16*)x, (const bf16*)skip, gamma, beta, (bf16*)y, stats, M, D, eps, out_scale, p_drop, seed, seed_inc);
    else return HYB_E_ARG;
    HYB_LAUNCH_CHECK();
    return 0;
}

extern "C" int hyb_ln_residual_bwd(int dtype, const void* dy, const void* x, const float* gamma, const float* stats, void* dx, void* dskip,
                                   int accumulate_dskip, float* dgamma, float* dbeta, int M, int D, float out_scale, float p_drop,
                                   unsigned long long seed, void* stream) {
    HYB_CHECK_ARG(dy && x && gamma && stats && dx && dskip && dgamma && dbeta && M > 0 && D > 0 && D % 8 == 0 && D <= 64 * 8 * LN_MAXC);
    hipStream_t st = (hipStream_t)stream;
    int blocks = hyb_cdiv(M, 4);
    if (blocks > 32) blocks = 32;
    const size_t lnlds = 2 * (size_t)D * sizeof(float);
    if (dtype == HYB_F32)
        hipLaunchKernelGGL((ln_residual_bwd_kernel<float, false>), dim3(blocks), dim3(256), lnlds, st, (const float*)dy, (const float*)x, gamma, stats, (float*)dx, (float*)dskip, accumulate_dskip, dgamma, dbeta, M, D, out_scale, p_drop, seed, (const unsigned long long*)nullptr);
    else if (dtype == HYB_BF16)
        hipLaunchKernelGGL((ln_residual_bwd_kernel<bf16, false>), dim3(blocks), dim3(256), lnlds, st, (const bf16*)dy, (const bf16*)x, gamma, stats, (bf16*)dx, (bf16*)dskip, accumulate_dskip, dgamma, dbeta, M, D, out_scale, p_drop, seed, (const unsigned long long*)nullptr);
    else return HYB_E_ARG;
    HYB_LAUNCH_CHECK();
    return 0;
}

// Internal (encoder composite): deterministic variant.  Writes hyb_ln_bwd_rows(M) partial rows [rows][2][D] to `part`;
// hyb_ln_rows_reduce sums any number of such rows into dgamma/dbeta (overwriting them).
int hyb_ln_bwd_rows(int M) { int b = hyb_cdiv(M, 4); return b > 32 ? 32 : b; }
int hyb_ln_residual_bwd_rows(int dtype, const void* dy, const void* x, const float* gamma, const float* stats, void* dx, void* dskip,
                             int accumulate_dskip, float* part, int M, int D, float out_scale, float p_drop, unsigned long long seed, const unsigned long long* seed_inc,
                             hipStream_t st) {
    HYB_CHECK_ARG(dy && x && gamma && stats && dx && dskip && part && M > 0 && D > 0 && D % 8 == 0 && D <= 64 * 8 * LN_MAXC);
    const int blocks = hyb_ln_bwd_rows(M);
    const size_t lnlds = 8 * (size_t)D * sizeof(float);
    if (dtype == HYB_F32)
        hipLaunchKernelGGL((ln_residual_bwd_kernel<float, true>), dim3(blocks), dim3(256), lnlds, st, (const float*)dy, (const float*)x, gamma, stats, (float*)dx, (float*)dskip, accumulate_dskip, part, (float*)nullptr, M, D, out_scale, p_drop, seed, seed_inc);
    else if (dtype == HYB_BF16)
        hipLaunchKernelGGL((ln_residual_bwd_kernel<bf16, true>), dim3(blocks), dim3(256), lnlds, st, (const bf16*)dy, (const bf16*)x, gamma, stats, (bf16*)dx, (bf16*)dskip, accumulate_dskip, part, (float*)nullptr, M, D, out_scale, p_drop, seed, seed_inc);
    else return HYB_E_ARG;
    HYB_LAUNCH_CHECK();
    return 0;
}
int hyb_ln_rows_reduce(const float* part, int rows, int D, float* dgamma, float* dbeta, hipStream_t st) {
    hipLaunchKernelGGL(ln_rows_reduce_kernel, dim3(hyb_cdiv(2 * D, 32)), dim3(1024), 0, st, part, rows, D, dgamma, dbeta);
    HYB_LAUNCH_CHECK();
    return 0;
}

// Internal (fused.hip): the temporal tail.  hyb_temporal_tail_ok: the shapes the one-workgroup-per-clip kernels take.
int hyb_temporal_tail_ok(int B, int S, int D, int C, int ln_rows) {
    return B >= 1 && B <= ln_rows && S >= 1 && D % 8 == 0 && D <= 1536 && C >= 1 && C <= 64;       // (D: 64 KB of LDS in the backward)
}
int hyb_temporal_tail_fwd(int dtype, const void* f, const void* x1, const float* gamma, const float* beta, void* enc_out, float* stats, int B, int S,
                          int D, float eps, float out_scale, float p_drop, unsigned long long seed, const unsigned long long* seed_inc, const float* W,
                          const float* bias, float* logits, int C, const long long* target, float* loss, float* ce_scratch, hipStream_t st) {
    HYB_CHECK_ARG(f && x1 && gamma && beta && enc_out && stats && W && logits && (!target || (loss && ce_scratch)));
    const size_t es = dtype == HYB_F32 ? 4 : 2;
    const size_t base = ((size_t)D + 64 + 256) * sizeof(float);
    const int rows_in_lds = base + (size_t)S * D * es <= 60 * 1024;
    const size_t lds = base + (rows_in_lds ? (size_t)S * D * es : 0);
    const int threads = S >= 8 ? 512 : 256;             // (1024 threads leave 128 registers per lane: the LayerNorm row spills -- measured 13 -> 24 us)
    if (dtype == HYB_F32)
        hipLaunchKernelGGL(temporal_tail_fwd_kernel<float>, dim3(B), dim3(threads), lds, st, (const float*)f, (const float*)x1, gamma, beta, (float*)enc_out, stats, B, S, D,
                           eps, out_scale, p_drop, seed, seed_inc, W, bias, logits, C, target, loss, ce_scratch, rows_in_lds);
    else if (dtype == HYB_BF16)
        hipLaunchKernelGGL(temporal_tail_fwd_kernel<bf16>, dim3(B), dim3(threads), lds, st, (const bf16*)f, (const bf16*)x1, gamma, beta, (bf16*)enc_out, stats, B, S, D,
                           eps, out_scale, p_drop, seed, seed_inc, W, bias, logits, C, target, loss, ce_scratch, rows_in_lds);
    else return HYB_E_ARG;
    HYB_LAUNCH_CHECK();
    return 0;
}
int hyb_temporal_tail_bwd(int dtype, const float* dlogits, const float* logits, const long long* target, const float* dloss, const float* W,
                          const void* enc_out, const void* f, const float* gamma, const float* stats, void* dx, void* dskip, float* ln_part,
                          int ln_rows, float* head_part, int B, int S, int D, int C, float out_scale, float p_drop, unsigned long long seed,
                          const unsigned long long* seed_inc, hipStream_t st) {
    HYB_CHECK_ARG((dlogits || (logits && target && dloss)) && W && enc_out && f && gamma && stats && dx && dskip && ln_part && head_part);
    const size_t lds = (64 + 9 * (size_t)D) * sizeof(float);
    // row blocks per clip: as many as the ln_rows partial rows allow (>= 1: hyb_temporal_tail_ok), four-row granules
    int nsb = hyb_cdiv(S, 4);
    if (nsb > ln_rows / B) nsb = ln_rows / B;
    const int rpw = hyb_cdiv(hyb_cdiv(S, nsb), 4) * 4;
    nsb = hyb_cdiv(S, rpw);
    const dim3 grid(nsb, B);
    if (lds > 64 * 1024) return HYB_E_ARG;
    if (dtype == HYB_F32)
        hipLaunchKernelGGL(temporal_tail_bwd_kernel<float>, grid, dim3(256), lds, st, dlogits, logits, target, dloss, W, (const float*)enc_out, (const float*)f, gamma,
                           stats, (float*)dx, (float*)dskip, ln_part, ln_rows, head_part, B, S, D, C, rpw, out_scale, p_drop, seed, seed_inc);
    else if (dtype == HYB_BF16)
        hipLaunchKernelGGL(temporal_tail_bwd_kernel<bf16>, grid, dim3(256), lds, st, dlogits, logits, target, dloss, W, (const bf16*)enc_out, (const bf16*)f, gamma,
                           stats, (bf16*)dx, (bf16*)dskip, ln_part, ln_rows, head_part, B, S, D, C, rpw, out_scale, p_drop, seed, seed_inc);
    else return HYB_E_ARG;
    HYB_LAUNCH_CHECK();
    return 0;
}

extern "C" int hyb_head_fwd(int dtype, const void* x, const float* W, const float* b, float* logits, int B, int S, int D, int C, void* stream) {
    HYB_CHECK_ARG(x && W && logits && B > 0 && S > 0 && D > 0 && C > 0);
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = (size_t)D * sizeof(float);
    if (dtype == HYB_F32) hipLaunchKernelGGL(head_fwd_kernel<float>, dim3(B), dim3(256), lds, st, (const float*)x, W, b, logits, S, D, C);
    else if (dtype == HYB_BF16) hipLaunchKernelGGL(head_fwd_kernel<bf16>, dim3(B), dim3(256), lds, st, (const bf16*)x, W, b, logits, S, D, C);
    else return HYB_E_ARG;
    HYB_LAUNCH_CHECK();
    return 0;
}

extern "C" int hyb_head_bwd(int dtype, const void* x, const float* W, const float* dlogits, void* dx, float* dW, float* db, int B, int S, int D,
                            int C, void* stream) {
    HYB_CHECK_ARG(x && W && dlogits && B > 0 && S > 0 && D > 0 && C > 0 && C <= 64);
    hipStream_t st = (hipStream_t)stream;
    if (dtype != HYB_F32 && dtype != HYB_BF16) return HYB_E_ARG;
    if (dx && !dW) {
        if (dtype == HYB_F32) hipLaunchKernelGGL(head_bwd_dx_kernel<float>, dim3(B), dim3(256), 0, st, W, dlogits, (float*)dx, S, D, C);
        else hipLaunchKernelGGL(head_bwd_dx_kernel<bf16>, dim3(B), dim3(256), 0, st, W, dlogits, (bf16*)dx, S, D, C);
        HYB_LAUNCH_CHECK();
    }
    if (dW) {        // one launch: weight/bias gradient blocks, and (when asked) the input-gradient blocks riding behind them
        const int yb = hyb_cdiv(C, 16), xb = dx ? hyb_cdiv(B, 4) : 0;
        if (dtype == HYB_F32) hipLaunchKernelGGL(head_bwd_dw_kernel<float>, dim3(hyb_cdiv(D, 64), yb + xb), dim3(256), 0, st, (const float*)x, dlogits, dW, db, B, S, D, C, W, (float*)dx, yb);
        else hipLaunchKernelGGL(head_bwd_dw_kernel<bf16>, dim3(hyb_cdiv(D, 64), yb + xb), dim3(256), 0, st, (const bf16*)x, dlogits, dW, db, B, S, D, C, W, (bf16*)dx, yb);
        HYB_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int hyb_cross_entropy_fwd(const float* logits, const long long* target, float* loss, int B, int C, void* stream) {
    HYB_CHECK_ARG(logits && target && loss && B > 0 && C > 0);
    hipLaunchKernelGGL(ce_fwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, logits, target, loss, B, C);
    HYB_LAUNCH_CHECK();
    return 0;
}
extern "C" int hyb_cross_entropy_bwd(const float* logits, const long long* target, const float* dloss, float* dlogits, int B, int C, void* stream) {
    HYB_CHECK_ARG(logits && target && dloss && dlogits && B > 0 && C > 0);
    hipLaunchKernelGGL(ce_bwd_kernel, dim3(hyb_cdiv(B, 64)), dim3(64), 0, (hipStream_t)stream, logits, target, dloss, dlogits, B, C);
    HYB_LAUNCH_CHECK();
    return 0;
}
