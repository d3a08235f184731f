// AdamW step over all parameter tensors of the model in ONE launch (the step right after the hot path;
// the reference's optimizer is torch.optim.AdamW, Model.py:153 / FCT.py:305).  Same update as torch.optim.AdamW
// (decoupled weight decay, bias correction, amsgrad = False, maximize = False):
//     p <- p * (1 - lr*wd);  m <- m + (1-b1)(g - m);  v <- b2*v + (1-b2) g^2;
//     p <- p - (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// The tensor table travels in the kernel arguments (up to 80 tensors per launch), a workgroup owns 4096 consecutive elements
// of one tensor and finds it by scanning the table's chunk offsets; HBM-bound: 7 x 4 bytes per parameter.
#include "hyb_common.h"

namespace {

constexpr int ADAM_MAX = 80, ADAM_CHUNK = 4096;

struct AdamTensor { float* p; const float* g; float* m; float* v; long long n; };
struct AdamArgs {
    AdamTensor t[ADAM_MAX];
    int chunk_begin[ADAM_MAX + 1];
    int count;
    float decay, omb1, beta2, omb2, eps, step_size, inv_sqrt_bc2;      // omb = 1 - beta, formed in double on the host like torch does
    // device-side step counter (hipGraph replays): when step_inc != NULL the bias corrections are formed on the device, in double,
    // from step + *step_inc
    const long long* step_inc;
    long long* advance;          // NULL, or = step_inc: the last workgroup to finish adds 1 to it (every workgroup has read it by then)
    unsigned int* ticket;        // the caller's ticket word of THIS counter (zero at rest): workgroups finished in the running launch
    long long step;
    double lr, beta1d, beta2d;
};

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, const AdamArgs& a, float step_size, float inv_sqrt_bc2) {
    p *= a.decay;
    m = m + a.omb1 * (g - m);
    v = a.beta2 * v + a.omb2 * g * g;
    const float denom = sqrtf(v) * inv_sqrt_bc2 + a.eps;
    p -= step_size * (m / denom);
}

__global__ __launch_bounds__(256) void adamw_kernel(AdamArgs a) {
    __shared__ float s_corr[2];
    int ti = 0;
    for (int i = 1; i < a.count; ++i)
        if ((int)blockIdx.x >= a.chunk_begin[i]) ti = i;
    const AdamTensor t = a.t[ti];
    const long long base = (long long)(blockIdx.x - a.chunk_begin[ti]) * ADAM_CHUNK;
    const bool vec = ((((uintptr_t)t.p | (uintptr_t)t.g | (uintptr_t)t.m | (uintptr_t)t.v) & 15) == 0);
    constexpr int NK = ADAM_CHUNK / (256 * 4);
    // a full, aligned chunk (all but each tensor's last): its 16 loads are issued BEFORE the bias corrections are formed (two double-
    // precision pow() on one thread, ~2 us) -- the kernel used to start every workgroup with that, then run four load -> store rounds
    const bool full = vec && base + ADAM_CHUNK <= t.n;
    f32x4 p[NK], m[NK], v[NK], g[NK];
    if (full) {
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const long long i = base + ((long long)k * 256 + threadIdx.x) * 4;
            p[k] = *reinterpret_cast<const f32x4*>(t.p + i); m[k] = *reinterpret_cast<const f32x4*>(t.m + i);
            v[k] = *reinterpret_cast<const f32x4*>(t.v + i); g[k] = *reinterpret_cast<const f32x4*>(t.g + i);
        }
    }
    float step_size = a.step_size, inv_sqrt_bc2 = a.inv_sqrt_bc2;
    if (a.step_inc) {                                          // uniform branch: every thread reaches the barrier
        if (threadIdx.x == 0) {
            const double tt = (double)(a.step + *a.step_inc);
            s_corr[0] = (float)(a.lr / (1.0 - pow(a.beta1d, tt)));
            s_corr[1] = (float)(1.0 / sqrt(1.0 - pow(a.beta2d, tt)));
        }
        __syncthreads();
        step_size = s_corr[0]; inv_sqrt_bc2 = s_corr[1];
    }
    if (full) {
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const long long i = base + ((long long)k * 256 + threadIdx.x) * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) { float pj = p[k][j], mj = m[k][j], vj = v[k][j]; adam_one(pj, g[k][j], mj, vj, a, step_size, inv_sqrt_bc2); p[k][j] = pj; m[k][j] = mj; v[k][j] = vj; }
            *reinterpret_cast<f32x4*>(t.p + i) = p[k]; *reinterpret_cast<f32x4*>(t.m + i) = m[k]; *reinterpret_cast<f32x4*>(t.v + i) = v[k];
        }
    } else {
#pragma unroll 1
        for (int k = 0; k < NK; ++k) {
            const long long i = base + ((long long)k * 256 + threadIdx.x) * 4;
            if (i >= t.n) break;
            if (vec && i + 4 <= t.n) {
                f32x4 pp = *reinterpret_cast<f32x4*>(t.p + i), mm = *reinterpret_cast<f32x4*>(t.m + i), vv = *reinterpret_cast<f32x4*>(t.v + i);
                const f32x4 gg = *reinterpret_cast<const f32x4*>(t.g + i);
#pragma unroll
                for (int j = 0; j < 4; ++j) { float pj = pp[j], mj = mm[j], vj = vv[j]; adam_one(pj, gg[j], mj, vj, a, step_size, inv_sqrt_bc2); pp[j] = pj; mm[j] = mj; vv[j] = vj; }
                *reinterpret_cast<f32x4*>(t.p + i) = pp; *reinterpret_cast<f32x4*>(t.m + i) = mm; *reinterpret_cast<f32x4*>(t.v + i) = vv;
            } else {
                for (long long e = i; e < i + 4 && e < t.n; ++e) adam_one(t.p[e], t.g[e], t.m[e], t.v[e], a, step_size, inv_sqrt_bc2);
            }
        }
    }
    // Thread 0 consumed the counter's value before the barrier at the top, so its read is complete here; the ticket only orders "every
    // workgroup has read" before the one write, which needs no fence (a release fence per workgroup costs an L2 write-back each: measured
    // 38 -> 126 us for this kernel).  The relaxed atomic is performed at the L2, in order per address.
    if (a.advance && threadIdx.x == 0) {
        if (__hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1) {
            __hip_atomic_store(a.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *a.advance += 1;
        }
    }
}

}  // namespace

extern "C" int hyb_adamw_step(int count, float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                              const long long* numel, double lr, double beta1, double beta2, double eps, double weight_decay, long long step,
                              long long* step_inc, unsigned int* advance_ticket, void* stream) {
    HYB_CHECK_ARG(count > 0 && params && grads && exp_avg && exp_avg_sq && numel && step >= 1 && lr >= 0.0 && (!advance_ticket || step_inc));
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    for (int first = 0; first < count; first += ADAM_MAX) {
        AdamArgs a{};
        const int n = count - first < ADAM_MAX ? count - first : ADAM_MAX;
        int chunks = 0;
        for (int i = 0; i < n; ++i) {
            HYB_CHECK_ARG(params[first + i] && grads[first + i] && exp_avg[first + i] && exp_avg_sq[first + i] && numel[first + i] > 0);
            a.t[i] = AdamTensor{params[first + i], grads[first + i], exp_avg[first + i], exp_avg_sq[first + i], numel[first + i]};
            a.chunk_begin[i] = chunks;
            chunks += hyb_cdiv(numel[first + i], ADAM_CHUNK);
        }
        a.chunk_begin[n] = chunks;
        a.count = n;
        // every scalar is formed in double from the caller's doubles and rounded once, as torch does with its Python floats
        a.decay = (float)(1.0 - lr * weight_decay);
        a.omb1 = (float)(1.0 - beta1); a.beta2 = (float)beta2; a.omb2 = (float)(1.0 - beta2); a.eps = (float)eps;
        a.step_size = (float)(lr / bc1);
        a.inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
        a.step_inc = step_inc; a.step = step; a.lr = lr; a.beta1d = beta1; a.beta2d = beta2;
        a.advance = (advance_ticket && first + ADAM_MAX >= count) ? step_inc : nullptr;       // the last launch of the call advances the counter
        a.ticket = advance_ticket;
        hipLaunchKernelGGL(adamw_kernel, dim3(chunks), dim3(256), 0, (hipStream_t)stream, a);
        HYB_LAUNCH_CHECK();
    }
    return 0;
}
