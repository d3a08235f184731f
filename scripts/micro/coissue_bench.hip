// Do a matrix-only wave and a vector-only wave that share a SIMD run concurrently on gfx950?
// 512-thread workgroups, one per CU: waves 0-3 issue NM bf16 MFMAs per round (independent accumulators), waves 4-7 NV dependent-free
// v_fma_f32 per round; a workgroup barrier ends every round (the structure of the fused weight-gradient kernel).  Times: matrix
// waves alone, vector waves alone, both.  Variants: accumulators in VGPRs / AGPRs (inline asm), 32x32x16 / 16x16x32, s_setprio.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

// MODE: bit0 matrix waves work, bit1 vector waves work;  ACC: 0 VGPR (builtin), 1 AGPR (asm), 2 16x16x32 builtin;  PRIO: 0 none, 1 vector waves prio 3, 2 matrix waves prio 3
template <int MODE, int ACC, int PRIO, int NM, int NV>
__global__ __launch_bounds__(512) void k(float* out, const float* in, int rounds) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wave < 4) {
        if (PRIO == 2) __builtin_amdgcn_s_setprio(3);
        bf16x8 a, b;
        for (int j = 0; j < 8; ++j) { a[j] = (__bf16)in[threadIdx.x + j]; b[j] = (__bf16)in[threadIdx.x + 8 + j]; }
        if (ACC == 2) {
            f32x4 acc[36];
            for (int t = 0; t < 36; ++t) acc[t] = f32x4{0, 0, 0, 0};
            for (int r = 0; r < rounds; ++r) {
                if (MODE & 1)
#pragma unroll
                    for (int m = 0; m < 2 * NM; ++m) acc[m % 36] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[m % 36], 0, 0, 0);
                __builtin_amdgcn_s_barrier();
            }
            float s = 0;
            for (int t = 0; t < 36; ++t) s += acc[t][0] + acc[t][3];
            out[blockIdx.x * 512 + threadIdx.x] = s;
        } else if (ACC == 0) {
            f32x16 acc[9];
            for (int t = 0; t < 9; ++t) for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
            for (int r = 0; r < rounds; ++r) {
                if (MODE & 1)
#pragma unroll
                    for (int m = 0; m < NM; ++m) acc[m % 9] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m % 9], 0, 0, 0);
                __builtin_amdgcn_s_barrier();
            }
            float s = 0;
            for (int t = 0; t < 9; ++t) s += acc[t][0] + acc[t][15];
            out[blockIdx.x * 512 + threadIdx.x] = s;
        } else {
            f32x16 acc[9];
            for (int t = 0; t < 9; ++t) for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
#pragma unroll
            for (int t = 0; t < 9; ++t) asm volatile("" : "+a"(acc[t]));
            for (int r = 0; r < rounds; ++r) {
                if (MODE & 1)
#pragma unroll
                    for (int m = 0; m < NM; ++m) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[m % 9]) : "v"(a), "v"(b));
                __builtin_amdgcn_s_barrier();
            }
            asm volatile("s_nop 15\n s_nop 15" ::: "memory");
            float s = 0;
            for (int t = 0; t < 9; ++t) s += acc[t][0] + acc[t][15];
            out[blockIdx.x * 512 + threadIdx.x] = s;
        }
    } else {
        if (PRIO == 1) __builtin_amdgcn_s_setprio(3);
        float v[8];
        for (int j = 0; j < 8; ++j) v[j] = in[threadIdx.x + j];
        const float c0 = in[0], c1 = in[1];
        for (int r = 0; r < rounds; ++r) {
            if (MODE & 2)
#pragma unroll
                for (int m = 0; m < NV; ++m) { v[m % 8] = fmaf(v[m % 8], c0, c1); asm volatile("" : "+v"(v[m % 8])); }
            __builtin_amdgcn_s_barrier();
        }
        float s = 0;
        for (int j = 0; j < 8; ++j) s += v[j];
        out[blockIdx.x * 512 + threadIdx.x] = s;
    }
}

template <int MODE, int ACC, int PRIO, int NM, int NV> float run(float* out, const float* in, int rounds) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((k<MODE, ACC, PRIO, NM, NV>), dim3(256), dim3(512), 0, 0, out, in, rounds);
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k<MODE, ACC, PRIO, NM, NV>), dim3(256), dim3(512), 0, 0, out, in, rounds);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1000.f / 5;
}

template <int ACC, int PRIO, int NM, int NV> void trio(const char* name, float* out, const float* in, int rounds) {
    const float m = run<1, ACC, PRIO, NM, NV>(out, in, rounds), v = run<2, ACC, PRIO, NM, NV>(out, in, rounds), b = run<3, ACC, PRIO, NM, NV>(out, in, rounds);
    printf("%-44s NM=%3d NV=%4d : matrix %.1f us, vector %.1f us, both %.1f us  (sum %.1f, max %.1f)\n", name, NM, NV, m, v, b, m + v, m > v ? m : v);
}


// The same question for a memory-streaming partner: waves 4-7 copy NL x 1 KiB per wave and round (global -> registers -> global, 16 bytes per lane),
// waves 0-3 issue NM MFMAs per round; a barrier ends every round.
template <int MODE, int NM, int NL>
__global__ __launch_bounds__(512) void kmem(float* out, const float* in, const f32x4* src, f32x4* dst, int rounds) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wave < 4) {
        bf16x8 a, b;
        for (int j = 0; j < 8; ++j) { a[j] = (__bf16)in[threadIdx.x + j]; b[j] = (__bf16)in[threadIdx.x + 8 + j]; }
        f32x16 acc[9];
        for (int t = 0; t < 9; ++t) for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
        for (int r = 0; r < rounds; ++r) {
            if (MODE & 1)
#pragma unroll
                for (int m = 0; m < NM; ++m) acc[m % 9] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m % 9], 0, 0, 0);
            __builtin_amdgcn_s_barrier();
        }
        float s = 0;
        for (int t = 0; t < 9; ++t) s += acc[t][0] + acc[t][15];
        out[blockIdx.x * 512 + threadIdx.x] = s;
    } else {
        const size_t per_round = (size_t)256 * NL * 64;                 // f32x4 elements per round per workgroup
        const size_t base = (size_t)blockIdx.x * per_round * rounds + (threadIdx.x - 256);
        for (int r = 0; r < rounds; ++r) {
            if (MODE & 2) {
                f32x4 v[NL];
#pragma unroll
                for (int m = 0; m < NL; ++m) v[m] = src[base + (size_t)r * per_round + m * 256];
#pragma unroll
                for (int m = 0; m < NL; ++m) dst[base + (size_t)r * per_round + m * 256] = v[m];
            }
            __builtin_amdgcn_s_barrier();
        }
    }
}
template <int MODE, int NM, int NL> float runmem(float* out, const float* in, const f32x4* src, f32x4* dst, int rounds) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((kmem<MODE, NM, NL>), dim3(256), dim3(512), 0, 0, out, in, src, dst, rounds);
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((kmem<MODE, NM, NL>), dim3(256), dim3(512), 0, 0, out, in, src, dst, rounds);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1000.f / 5;
}
template <int NM, int NL> void triomem(float* out, const float* in, const f32x4* src, f32x4* dst, int rounds) {
    const float m = runmem<1, NM, NL>(out, in, src, dst, rounds), v = runmem<2, NM, NL>(out, in, src, dst, rounds), b = runmem<3, NM, NL>(out, in, src, dst, rounds);
    const double mb = 256.0 * 256 * NL * 64 * 16 * rounds / 1e6;
    printf("MFMA %3d + copy %2d KiB per wave and round (%4.0f MB each way): matrix %.1f us, copy %.1f us (%.2f TB/s r+w), both %.1f us  (sum %.1f, max %.1f)\n", NM, NL, mb, m, v,
           2 * mb / v / 1e6 * 1e6 / 1e6, b, m + v, m > v ? m : v);
}

int main() {
    float *out, *in;
    CK(hipMalloc(&out, 256 * 512 * 4)); CK(hipMalloc(&in, 4096 * 4));
    float h[4096]; for (int i = 0; i < 4096; ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f;
    CK(hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice));
    const int rounds = 200;
    trio<0, 0, 126, 250>("32x32x16 VGPR acc", out, in, rounds);
    trio<0, 0, 126, 500>("32x32x16 VGPR acc", out, in, rounds);
    trio<0, 0, 126, 750>("32x32x16 VGPR acc", out, in, rounds);
    trio<1, 0, 126, 500>("32x32x16 AGPR acc (asm)", out, in, rounds);
    trio<1, 0, 126, 750>("32x32x16 AGPR acc (asm)", out, in, rounds);
    trio<2, 0, 126, 500>("16x16x32 VGPR acc", out, in, rounds);
    trio<0, 1, 126, 500>("32x32x16 VGPR acc, vector waves prio 3", out, in, rounds);
    trio<0, 2, 126, 500>("32x32x16 VGPR acc, matrix waves prio 3", out, in, rounds);
    trio<1, 1, 126, 500>("32x32x16 AGPR acc, vector waves prio 3", out, in, rounds);
    {
        const int rounds = 50;
        const size_t elems = (size_t)256 * 256 * 16 * 64 * rounds;      // f32x4 elements for NL = 16
        f32x4 *src, *dst; CK(hipMalloc(&src, elems * 16)); CK(hipMalloc(&dst, elems * 16));
        CK(hipMemset(src, 1, elems * 16));
        triomem<126, 4>(out, in, src, dst, rounds);
        triomem<126, 8>(out, in, src, dst, rounds);
        triomem<126, 16>(out, in, src, dst, rounds);
        triomem<63, 16>(out, in, src, dst, rounds);
    }
    return 0;
}
