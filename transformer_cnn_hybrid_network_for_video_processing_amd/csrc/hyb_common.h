// Shared device helpers for the gfx950 kernels: storage types, the MFMA "k32" step used by
// every contraction, wave reductions and the counter-based dropout RNG.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include "hybrid_hip.h"

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

#define HYB_WAVE 64

#define HYB_CHECK_ARG(cond) do { if (!(cond)) return HYB_E_ARG; } while (0)
#define HYB_LAUNCH_CHECK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return (int)e_; } while (0)

static inline int hyb_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) sticks to (function, device).  One HybAttrOnce per kernel instantiation
// remembers the devices it has been set on in a lock-free bitmask: the call is idempotent, so two threads racing on the
// first launch merely both make it; a second device in the same process gets its own call.
struct HybAttrOnce { std::atomic<unsigned long long> done{0}; };
static inline int hyb_set_lds_attr(HybAttrOnce& once, const void* fn, int bytes) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    const unsigned long long bit = 1ull << (dev & 63);
    if (once.done.load(std::memory_order_acquire) & bit) return 0;
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return (int)e;
    once.done.fetch_or(bit, std::memory_order_release);
    return 0;
}

// measurement hooks (hyb_profile_set): defined in bn_pool.hip
struct HybProfileHook { int kernel_id, a, b; hipEvent_t ev0, ev1; };
extern HybProfileHook g_hyb_hooks[16];
extern int g_hyb_hooks_active;
static inline HybProfileHook* hyb_find_hook(int kernel_id, int a, int b) {
    if (!g_hyb_hooks_active) return nullptr;
    for (int i = 0; i < 16; ++i)
        if (g_hyb_hooks[i].kernel_id == kernel_id && g_hyb_hooks[i].a == a && g_hyb_hooks[i].b == b) return &g_hyb_hooks[i];
    return nullptr;
}

template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<bf16>(bf16 v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16 from_f32<bf16>(float v) { return (bf16)v; }   // v_cvt_pk_bf16_f32: RNE, NaN-safe

// ---------------------------------------------------------------------------------------
// One lane's share of an MFMA operand for a 32-deep K step: 8 consecutive K elements of
// one row (A) / column (B).  Lane l holds row/col (l & 15), K elements 8*(l >> 4) + j.
//   bf16: exactly the operand of v_mfma_f32_16x16x32_bf16.
//   fp32: eight v_mfma_f32_16x16x4_f32, MFMA j contracting K = {8q + j : q = 0..3}; the
//         reduction order inside the 32-step is permuted but identical for A and B.
// C/D: lane l holds column (l & 15), rows 4*(l >> 4) + r, r = 0..3.
// ---------------------------------------------------------------------------------------
// one more Linear weight (+ bias) gradient for the encoder backward's final multi-matrix launch (hyb_encoder_bwd_impl): dW[N][K] = dy^T x
struct HybDwExtra { const void* dy; const void* x; float* dW; float* db; int N, K, lddy, ldx; };

// a rider of that launch: out[c] = sum over `rows` partial rows of n floats each (part[r*n + c], rows in ascending order); columns < split go
// to out0[c], the others to out1[c - split] (LayerNorm: n = 2D, split = D -> dgamma, dbeta; the head: n = C*D + C, split = C*D -> dW, db)
struct HybDwRider { const float* part; float* out0; float* out1; int rows; long long n, split; };

// operands of the last encoder layer's second LayerNorm, handed to the fused temporal tail launches (model.hip -> fused.hip -> layernorm.hip)
struct HybEncTail { const void* f; const void* x1; float* st2; const float* gamma; const float* beta; float eps, out_scale, p_drop; unsigned long long seed; };
struct HybEncBwdTail { const void* f; const float* stats; const float* gamma; void* dx; void* dskip; float* ln_part; int ln_rows; float out_scale, p_drop; unsigned long long seed; };

// partial weight-gradient slabs a contraction kernel left behind for a LATER fixed-order sum (hyb_wgrad_reduce_multi, conv_wgrad.hip):
// dW[co][ci][tap] = sum_s slab[s][(co * 9 + tap) * Cip + ci]; S = 0: nothing deferred (the gradient is already in dW)
struct HybSlabInfo { const float* slab; float* dw; int S, Co, Ci, Cip; long long per_slab; };

template <typename T> struct Frag;
template <> struct Frag<bf16> { bf16x8 v; };
template <> struct Frag<float> { float v[8]; };

__device__ __forceinline__ f32x4 mma32(const Frag<bf16>& a, const Frag<bf16>& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, c, 0, 0, 0);
}
#ifdef HYB_F32_X3
// libhybrid_hip_x3.so ("bf16x3" mode of the host side): fp32 STORAGE everywhere, but every product of a contraction is formed on the
// bf16 matrix cores from the two-term splits x = hi + lo (hi = bf16(x), lo = bf16(x - hi): 16 significant bits):
//   a*b ~ hi_a*hi_b + hi_a*lo_b + lo_a*hi_b      (the dropped lo*lo term is 2^-16 relative), fp32 accumulation,
// three v_mfma_f32_16x16x32_bf16 (48 cycles) instead of eight v_mfma_f32_16x16x4_f32 (256 cycles) per 32-deep step.  The splits of a
// fragment are pure functions of it: the compiler forms them once per fragment, not once per product.
__device__ __forceinline__ void hyb_split_bf16(const Frag<float>& f, bf16x8& hi, bf16x8& lo) {
#pragma unroll
    for (int j = 0; j < 8; ++j) { hi[j] = (bf16)f.v[j]; lo[j] = (bf16)(f.v[j] - (float)hi[j]); }
}
__device__ __forceinline__ f32x4 mma32(const Frag<float>& a, const Frag<float>& b, f32x4 c) {
    bf16x8 ah, al, bh, bl;
    hyb_split_bf16(a, ah, al);
    hyb_split_bf16(b, bh, bl);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, c, 0, 0, 0);      // small terms first
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, c, 0, 0, 0);
}
// PRE-SPLIT operand storage: where a kernel reads the same fragment many times (a staged image read by nine taps, packed weights),
// the 32 bytes of eight fp32 values hold [hi bf16 x 8][lo bf16 x 8] instead -- written once by whoever stages / packs the operand
// (hyb_presplit8), consumed without any vector arithmetic (mma32_pre).  Same hi / lo values and the same three MFMAs in the same order as
// mma32: results are bit-identical.
constexpr bool HYB_X3 = true;
__device__ __forceinline__ void hyb_presplit8(f32x4& a, f32x4& b) {
    bf16x8 hi, lo;
#pragma unroll
    for (int j = 0; j < 4; ++j) { hi[j] = (bf16)a[j]; lo[j] = (bf16)(a[j] - (float)hi[j]); hi[4 + j] = (bf16)b[j]; lo[4 + j] = (bf16)(b[j] - (float)hi[4 + j]); }
    a = __builtin_bit_cast(f32x4, hi);
    b = __builtin_bit_cast(f32x4, lo);
}
__device__ __forceinline__ f32x4 mma32_pre(const Frag<float>& a, const Frag<float>& b, f32x4 c) {
    const bf16x8 ah = __builtin_bit_cast(bf16x8, f32x4{a.v[0], a.v[1], a.v[2], a.v[3]}), al = __builtin_bit_cast(bf16x8, f32x4{a.v[4], a.v[5], a.v[6], a.v[7]});
    const bf16x8 bh = __builtin_bit_cast(bf16x8, f32x4{b.v[0], b.v[1], b.v[2], b.v[3]}), bl = __builtin_bit_cast(bf16x8, f32x4{b.v[4], b.v[5], b.v[6], b.v[7]});
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, c, 0, 0, 0);
}
// one element of a pre-split fp32 buffer: element i of the logical array lives in the 8-group i >> 3 as halves (i & 7) and 8 + (i & 7)
__device__ __forceinline__ void hyb_presplit_store(float* base, long long i, float v) {
    const bf16 hi = (bf16)v, lo = (bf16)(v - (float)hi);
    unsigned short* h = reinterpret_cast<unsigned short*>(base) + (i >> 3) * 16 + (i & 7);
    h[0] = __builtin_bit_cast(unsigned short, hi);
    h[8] = __builtin_bit_cast(unsigned short, lo);
}
// four fp32 values (one pixel's channel quad, 16 bytes) -> the same 16 bytes holding [hi bf16 x 4][lo bf16 x 4]; two such quads q0, q1 of
// horizontally adjacent pixels make the pre-split fragment {q0.hi, q1.hi | q0.lo, q1.lo} by register naming alone (conv_first_wave.hip)
__device__ __forceinline__ void hyb_presplit4(float (&v)[4]) {
    bf16x4 hi, lo;
#pragma unroll
    for (int j = 0; j < 4; ++j) { hi[j] = (bf16)v[j]; lo[j] = (bf16)(v[j] - (float)hi[j]); }
    const f32x2 h = __builtin_bit_cast(f32x2, hi), l = __builtin_bit_cast(f32x2, lo);
    v[0] = h[0]; v[1] = h[1]; v[2] = l[0]; v[3] = l[1];
}
// ELEMENT-PACKED operand storage, for kernels that gather a fragment's eight values from eight places (transposing reads of a staged
// image, conv_wgrad.hip / conv_first.hip): every fp32 LDS word holds (bits of hi) << 16 | bits of lo.  hyb_epack once per staged element;
// mma32_e rebuilds the two bf16 fragments with byte permutes instead of conversions and subtractions.  Same values, same MFMAs.
__device__ __forceinline__ float hyb_epack(float x) {
    const bf16 hi = (bf16)x, lo = (bf16)(x - (float)hi);
    return __builtin_bit_cast(float, ((unsigned)__builtin_bit_cast(unsigned short, hi) << 16) | (unsigned)__builtin_bit_cast(unsigned short, lo));
}
__device__ __forceinline__ void hyb_eunpack(const Frag<float>& f, bf16x8& hi, bf16x8& lo) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const unsigned u = __builtin_bit_cast(unsigned, f.v[j]);
        hi[j] = __builtin_bit_cast(bf16, (unsigned short)(u >> 16));
        lo[j] = __builtin_bit_cast(bf16, (unsigned short)(u & 0xffffu));
    }
}
__device__ __forceinline__ f32x4 mma32_e(const Frag<float>& a, const Frag<float>& b, f32x4 c) {
    bf16x8 ah, al, bh, bl;
    hyb_eunpack(a, ah, al);
    hyb_eunpack(b, bh, bl);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, c, 0, 0, 0);
}
#else
constexpr bool HYB_X3 = false;
__device__ __forceinline__ f32x4 mma32(const Frag<float>& a, const Frag<float>& b, f32x4 c) {
#pragma unroll
    for (int j = 0; j < 8; ++j) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v[j], b.v[j], c, 0, 0, 0);
    return c;
}
// (only the split-bf16 build stores pre-split operands; these keep the shared kernel sources compiling)
__device__ __forceinline__ void hyb_presplit8(f32x4&, f32x4&) {}
__device__ __forceinline__ f32x4 mma32_pre(const Frag<float>& a, const Frag<float>& b, f32x4 c) { return mma32(a, b, c); }
__device__ __forceinline__ void hyb_presplit_store(float* base, long long i, float v) { base[i] = v; }
__device__ __forceinline__ void hyb_presplit4(float (&)[4]) {}
__device__ __forceinline__ float hyb_epack(float x) { return x; }
__device__ __forceinline__ f32x4 mma32_e(const Frag<float>& a, const Frag<float>& b, f32x4 c) { return mma32(a, b, c); }
#endif
__device__ __forceinline__ void hyb_presplit4(bf16 (&)[4]) {}
__device__ __forceinline__ bf16 hyb_epack(bf16 x) { return x; }
__device__ __forceinline__ f32x4 mma32_e(const Frag<bf16>& a, const Frag<bf16>& b, f32x4 c) { return mma32(a, b, c); }
__device__ __forceinline__ f32x4 mma32_pre(const Frag<bf16>& a, const Frag<bf16>& b, f32x4 c) { return mma32(a, b, c); }

// load a fragment from 8 consecutive T (16-byte aligned for bf16, 16-byte aligned for fp32)
__device__ __forceinline__ void frag_load(Frag<bf16>& f, const bf16* p) { f.v = *reinterpret_cast<const bf16x8*>(p); }
__device__ __forceinline__ void frag_load(Frag<float>& f, const float* p) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
    f.v[0] = a[0]; f.v[1] = a[1]; f.v[2] = a[2]; f.v[3] = a[3];
    f.v[4] = b[0]; f.v[5] = b[1]; f.v[6] = b[2]; f.v[7] = b[3];
}
__device__ __forceinline__ void frag_zero(Frag<bf16>& f) {
#pragma unroll
    for (int j = 0; j < 8; ++j) f.v[j] = (bf16)0.0f;
}
__device__ __forceinline__ void frag_zero(Frag<float>& f) {
#pragma unroll
    for (int j = 0; j < 8; ++j) f.v[j] = 0.0f;
}
template <typename T> __device__ __forceinline__ void frag_set(Frag<T>& f, int j, float v);
template <> __device__ __forceinline__ void frag_set<bf16>(Frag<bf16>& f, int j, float v) { f.v[j] = (bf16)v; }
template <> __device__ __forceinline__ void frag_set<float>(Frag<float>& f, int j, float v) { f.v[j] = v; }

// 8 consecutive elements as a vector (global/LDS, 16-byte aligned for bf16; fp32 = 2 x 16 B)
template <typename T> struct Vec8;
template <> struct Vec8<bf16> {
    bf16x8 v;
    __device__ __forceinline__ void load(const bf16* p) { v = *reinterpret_cast<const bf16x8*>(p); }
    __device__ __forceinline__ void store(bf16* p) const { *reinterpret_cast<bf16x8*>(p) = v; }
    // streaming (non-temporal) forms: touched once, should not displace L2-resident weights
    __device__ __forceinline__ void load_nt(const bf16* p) { v = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(p)); }
    __device__ __forceinline__ void store_nt(bf16* p) const { __builtin_nontemporal_store(v, reinterpret_cast<bf16x8*>(p)); }
    __device__ __forceinline__ float get(int j) const { return (float)v[j]; }
    __device__ __forceinline__ void set(int j, float x) { v[j] = (bf16)x; }
    __device__ __forceinline__ void presplit() {}
    __device__ __forceinline__ void epack() {}
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (bf16)0.0f;
    }
};
template <> struct Vec8<float> {
    f32x4 a, b;
    __device__ __forceinline__ void load(const float* p) { a = *reinterpret_cast<const f32x4*>(p); b = *reinterpret_cast<const f32x4*>(p + 4); }
    __device__ __forceinline__ void store(float* p) const { *reinterpret_cast<f32x4*>(p) = a; *reinterpret_cast<f32x4*>(p + 4) = b; }
    __device__ __forceinline__ void load_nt(const float* p) {
        a = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p)); b = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p + 4));
    }
    __device__ __forceinline__ void store_nt(float* p) const {
        __builtin_nontemporal_store(a, reinterpret_cast<f32x4*>(p)); __builtin_nontemporal_store(b, reinterpret_cast<f32x4*>(p + 4));
    }
    __device__ __forceinline__ float get(int j) const { return j < 4 ? a[j] : b[j - 4]; }
    __device__ __forceinline__ void set(int j, float x) { if (j < 4) a[j] = x; else b[j - 4] = x; }
    __device__ __forceinline__ void zero() { a = f32x4{0, 0, 0, 0}; b = f32x4{0, 0, 0, 0}; }
    __device__ __forceinline__ void epack() {
#pragma unroll
        for (int j = 0; j < 4; ++j) { a[j] = hyb_epack(a[j]); b[j] = hyb_epack(b[j]); }
    }
    __device__ __forceinline__ void presplit() { hyb_presplit8(a, b); }          // split-bf16 build only (hyb_presplit8); zeros stay zeros
};

// ---------------------------------------------------------------------------------------
// Buffer descriptor over [p, p + records) bytes, built from wave-uniform inputs.  The readfirstlane makes the uniformity
// provable to the compiler (otherwise every buffer op is wrapped in a waterfall loop).
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ __amdgpu_buffer_rsrc_t hyb_rsrc(const void* p, unsigned records) {
    const unsigned long long a = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, __builtin_amdgcn_readfirstlane(records), 0x00020000);
}

// ---------------------------------------------------------------------------------------
// wave-level reductions (64 lanes)
// ---------------------------------------------------------------------------------------
// DPP row rotations inside the four 16-lane rows (v_add_f32_dpp: no LDS crossbar round trips -- six dependent ds_bpermute of the
// shuffle form cost ~0.7 us, a visible share of a 4 us latency-bound kernel), then the four row totals through v_readlane.  Every lane
// gets the same value; the order of the additions is fixed.
template <int CTRL> __device__ __forceinline__ float hyb_dpp_mov(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float wave_sum(float v) {
    v += hyb_dpp_mov<0x128>(v);      // row_ror:8
    v += hyb_dpp_mov<0x124>(v);      // row_ror:4
    v += hyb_dpp_mov<0x122>(v);      // row_ror:2
    v += hyb_dpp_mov<0x121>(v);      // row_ror:1
    const int iv = __float_as_int(v);
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(iv, 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(iv, 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(iv, 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(iv, 48));
    return (r0 + r1) + (r2 + r3);
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, hyb_dpp_mov<0x128>(v));
    v = fmaxf(v, hyb_dpp_mov<0x124>(v));
    v = fmaxf(v, hyb_dpp_mov<0x122>(v));
    v = fmaxf(v, hyb_dpp_mov<0x121>(v));
    const int iv = __float_as_int(v);
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(iv, 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(iv, 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(iv, 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(iv, 48));
    return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}
// reduce across the 16 lanes that share (lane >> 4)
__device__ __forceinline__ float group16_sum(float v) {
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// same sum over the 16 lanes of a DPP row, but with row rotations (v_add_f32_dpp, no LDS crossbar): every lane of the row
// ends up with the total.  The order of the additions differs from group16_sum (both are fixed orders).
template <int CTRL> __device__ __forceinline__ float dpp_rot_add(float v) {
    const int t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false);
    return v + __int_as_float(t);
}
__device__ __forceinline__ float row16_sum(float v) {
    v = dpp_rot_add<0x128>(v);      // row_ror:8
    v = dpp_rot_add<0x124>(v);      // row_ror:4
    v = dpp_rot_add<0x122>(v);      // row_ror:2
    v = dpp_rot_add<0x121>(v);      // row_ror:1
    return v;
}
__device__ __forceinline__ float group16_max(float v) {
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ---------------------------------------------------------------------------------------
// dropout: stateless counter-based RNG.  keep(seed, idx) is a pure function so the
// backward pass regenerates the forward mask from the same (seed, element index).
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t hyb_hash(unsigned long long seed, unsigned long long idx) {
    unsigned long long z = idx * 0x9E3779B97F4A7C15ull + seed;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (uint32_t)(z >> 32);
}
// returns the multiplier: 0 (dropped) or 1/(1-p)
__device__ __forceinline__ float dropout_mult(unsigned long long seed, unsigned long long idx, float p, float inv_keep) {
    const float u = (float)(hyb_hash(seed, idx) >> 8) * (1.0f / 16777216.0f);
    return u >= p ? inv_keep : 0.0f;
}

// ---------------------------------------------------------------------------------------
// Fixed-order sum of G partial rows of n floats: column i = sum_g part[g*n + i].  Launch with 1024 threads per block
// (32 columns x 32 row groups), grid = ceil(n / 32).  Deterministic: no float atomics.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ bool rows_reduce_1024(const float* __restrict__ part, int G, long long n, long long& col_out, float& sum_out,
                                                 long long col_offset = 0, long long ncols = -1) {
    __shared__ float rr_red[32][33];
    const int col = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const long long ii = (long long)blockIdx.x * 32 + col;
    const long long i = (ncols < 0 || ii < ncols) ? ii + col_offset : n;      // columns [col_offset, col_offset + ncols) of rows of length n
    __syncthreads();                                                          // allow several calls per kernel (shared scratch reuse)
    float a0 = 0.f, a1 = 0.f;
    if (i < n) {
        int g = grp;
        // eight independent loads in flight per thread (these kernels are one or two workgroups: pure latency), fixed add order
        for (; g + 7 * 32 < G; g += 8 * 32) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = part[(long long)(g + j * 32) * n + i];
            a0 += v[0]; a1 += v[1]; a0 += v[2]; a1 += v[3]; a0 += v[4]; a1 += v[5]; a0 += v[6]; a1 += v[7];
        }
        for (; g + 32 < G; g += 64) { a0 += part[(long long)g * n + i]; a1 += part[(long long)(g + 32) * n + i]; }
        if (g < G) a0 += part[(long long)g * n + i];
    }
    rr_red[grp][col] = a0 + a1;
    __syncthreads();
    if (grp != 0 || i >= n) return false;
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 32; ++k) s += rr_red[k][col];
    col_out = i;
    sum_out = s;
    return true;
}
