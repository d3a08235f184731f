// Microbenchmark: does a chain of dependent tiny launches cost more when consecutive launches are DIFFERENT kernels (different code
// objects / register and LDS footprints) than when one kernel is launched repeatedly?  Replayed hipGraph, 256 nodes per graph.
// Variants: same kernel; 8 distinct kernels round-robin (template instances with different LDS sizes and register pressure);
// one "uber" kernel that contains the 8 bodies behind a runtime switch (one symbol, one footprint).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int V>
__device__ __forceinline__ void body(float* buf, int i, float* lds) {
    // a little dependent work whose shape depends on V (so the instances really are different code)
    const int t = threadIdx.x;
    float acc[4 + 4 * V];
#pragma unroll
    for (int j = 0; j < 4 + 4 * V; ++j) acc[j] = buf[(blockIdx.x * 256 + t + 64 * j) & 65535];
    lds[t] = acc[0];
    __syncthreads();
    float s = lds[(t + 1 + V) & 255];
#pragma unroll
    for (int j = 0; j < 4 + 4 * V; ++j) s += acc[j] * (float)(j + 1 + V);
    buf[(blockIdx.x * 256 + t + i) & 65535] = s * 1e-3f;
}
template <int V>
__global__ __launch_bounds__(256) void distinct_kernel(float* buf, int i) {
    __shared__ float lds[256 + 1024 * V];
    body<V>(buf, i, lds);
}
__global__ __launch_bounds__(256) void uber_kernel(float* buf, int i, int mode) {
    __shared__ float lds[256 + 1024 * 7];
    switch (mode) {
        case 0: body<0>(buf, i, lds); break; case 1: body<1>(buf, i, lds); break; case 2: body<2>(buf, i, lds); break; case 3: body<3>(buf, i, lds); break;
        case 4: body<4>(buf, i, lds); break; case 5: body<5>(buf, i, lds); break; case 6: body<6>(buf, i, lds); break; default: body<7>(buf, i, lds); break;
    }
}
static void launch_distinct(int v, int nwg, float* buf, int i, hipStream_t st) {
    switch (v) {
        case 0: distinct_kernel<0><<<nwg, 256, 0, st>>>(buf, i); break; case 1: distinct_kernel<1><<<nwg, 256, 0, st>>>(buf, i); break;
        case 2: distinct_kernel<2><<<nwg, 256, 0, st>>>(buf, i); break; case 3: distinct_kernel<3><<<nwg, 256, 0, st>>>(buf, i); break;
        case 4: distinct_kernel<4><<<nwg, 256, 0, st>>>(buf, i); break; case 5: distinct_kernel<5><<<nwg, 256, 0, st>>>(buf, i); break;
        case 6: distinct_kernel<6><<<nwg, 256, 0, st>>>(buf, i); break; default: distinct_kernel<7><<<nwg, 256, 0, st>>>(buf, i); break;
    }
}
int main() {
    float* buf; CK(hipMalloc(&buf, 65536 * 4)); CK(hipMemset(buf, 0, 65536 * 4));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int NODES = 256;
    printf("{\"graph_chain_us_per_launch\": [\n");
    bool first = true;
    for (int nwg : {32, 128}) for (int variant = 0; variant < 3; ++variant) {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < NODES; ++i) {
            if (variant == 0) distinct_kernel<3><<<nwg, 256, 0, st>>>(buf, i);
            else if (variant == 1) launch_distinct(i & 7, nwg, buf, i, st);
            else uber_kernel<<<nwg, 256, 0, st>>>(buf, i, i & 7);
        }
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
        float best = 1e30f;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipEventRecord(e0, st)); CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        printf("%s {\"workgroups\": %d, \"chain\": \"%s\", \"us_per_launch\": %.3f}", first ? "" : ",\n", nwg,
               variant == 0 ? "one kernel repeated" : variant == 1 ? "8 distinct kernels round-robin" : "one uber kernel, 8 modes round-robin", best * 1000.f / NODES);
        first = false;
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    printf("\n],\n\"cold_pages_us_per_launch\": [\n");
    // the same one-kernel chain, but launch i works in a different 2 MB region of a 2 GB buffer (address translations not resident)
    {
        const size_t REGION = 2u << 20, NREG = 1024;
        float* big; CK(hipMalloc(&big, REGION * NREG)); CK(hipMemset(big, 0, REGION * NREG));
        first = true;
        for (int spread = 0; spread < 2; ++spread) {
            hipGraph_t g; hipGraphExec_t ge;
            CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            for (int i = 0; i < NODES; ++i) {
                float* base = big + (spread ? ((size_t)(i * 7919) % NREG) * (REGION / 4) : 0);
                distinct_kernel<3><<<32, 256, 0, st>>>(base, i);
            }
            CK(hipStreamEndCapture(st, &g));
            CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
            float best = 1e30f;
            for (int rep = 0; rep < 5; ++rep) {
                CK(hipMemsetAsync(big, 0, REGION * NREG, st));                 // sweep 2 GB between replays: evicts caches and translations
                CK(hipEventRecord(e0, st)); CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            printf("%s {\"working_set\": \"%s\", \"us_per_launch\": %.3f}", first ? "" : ",\n",
                   spread ? "a different 2 MB region of 2 GB per launch" : "one 256 KB region", best * 1000.f / NODES);
            first = false;
            CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
        }
    }
    printf("\n]}\n");
    return 0;
}
