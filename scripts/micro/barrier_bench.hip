// Microbenchmark: cost of a device-wide barrier inside one persistent launch vs the boundary between two dependent
// launches (stream order and replayed hipGraph), on gfx950.  Decides whether a persistent encoder kernel can pay.
// Build: hipcc --offload-arch=gfx950 -O3 -o barrier_bench barrier_bench.hip ; run: ./barrier_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

struct Bar { unsigned* counter; unsigned* abort_flag; };

__device__ __forceinline__ bool grid_barrier(Bar b, unsigned& target, unsigned nwg) {
    __syncthreads();
    __shared__ int ok;
    if (threadIdx.x == 0) {
        target += nwg;
        __threadfence();
        __hip_atomic_fetch_add(b.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int good = 1;
        unsigned spins = 0;
        while (__hip_atomic_load(b.counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (++spins > (1u << 22) || __hip_atomic_load(b.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                __hip_atomic_store(b.abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                good = 0;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        __threadfence();
        ok = good;
    }
    __syncthreads();
    return ok != 0;
}

// nb barriers; between barriers every workgroup writes a token and reads its neighbour's token of the previous phase.
__global__ void persistent_kernel(Bar b, unsigned* tokens, unsigned* errors, int nb, int with_data) {
    unsigned target = 0;
    const unsigned nwg = gridDim.x;
    for (int i = 0; i < nb; ++i) {
        if (with_data && threadIdx.x == 0) tokens[blockIdx.x * 32] = (unsigned)(i + 1);
        if (!grid_barrier(b, target, nwg)) return;
        if (with_data && threadIdx.x == 0) {
            unsigned nbr = (blockIdx.x + nwg / 2 + 1) % nwg;
            unsigned v = __hip_atomic_load(&tokens[nbr * 32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (v != (unsigned)(i + 1) && v != (unsigned)(i + 2)) atomicAdd(errors, 1u);
        }
    }
}

// Clip-local variant (round 3): 256 workgroups in 8 groups of 32 = blockIdx % 8 (workgroups are dealt round-robin over the 8 XCDs, so a
// group shares an XCD and its L2); each group synchronises on ITS OWN counter (one 256-byte line per group) -- the structure of a persistent
// encoder layer in which a clip's 32 workgroups only ever wait for each other (attention never crosses clips).  All 8 groups run at once.
__global__ void persistent_local_kernel(unsigned* counters, unsigned* abort_flag, unsigned* tokens, unsigned* errors, int nb, int with_data) {
    unsigned target = 0;
    const unsigned grp = blockIdx.x & 7, idx = blockIdx.x >> 3, gsz = gridDim.x >> 3;
    Bar b{counters + grp * 64, abort_flag};
    for (int i = 0; i < nb; ++i) {
        if (with_data && threadIdx.x == 0) tokens[blockIdx.x * 32] = (unsigned)(i + 1);
        if (!grid_barrier(b, target, gsz)) return;
        if (with_data && threadIdx.x == 0) {
            unsigned nbr = ((idx + gsz / 2 + 1) % gsz) * 8 + grp;                   // a workgroup of the same group
            unsigned v = __hip_atomic_load(&tokens[nbr * 32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (v != (unsigned)(i + 1) && v != (unsigned)(i + 2)) atomicAdd(errors, 1u);
        }
    }
}

__global__ void tiny_kernel(unsigned* tokens, int i) {
    if (threadIdx.x == 0) {
        unsigned nbr = (blockIdx.x + gridDim.x / 2 + 1) % gridDim.x;
        unsigned v = tokens[nbr * 32 + 1];
        tokens[blockIdx.x * 32 + 1] = v + (unsigned)i;
    }
}

int main() {
    unsigned *counter, *abortf, *tokens, *errors;
    CK(hipMalloc(&counter, 4)); CK(hipMalloc(&abortf, 4)); CK(hipMalloc(&errors, 4));
    CK(hipMalloc(&tokens, 4 * 32 * 1024));
    CK(hipMemset(tokens, 0, 4 * 32 * 1024));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int NB = 2000;
    printf("{\"barriers\": [\n");
    bool first = true;
    for (int threads : {256, 512}) for (int nwg : {32, 64, 128, 256}) for (int with_data : {0, 1}) {
        float best = 1e30f; unsigned herr = 0, habort = 0;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemsetAsync(counter, 0, 4, st)); CK(hipMemsetAsync(abortf, 0, 4, st)); CK(hipMemsetAsync(errors, 0, 4, st));
            Bar b{counter, abortf};
            CK(hipEventRecord(e0, st));
            persistent_kernel<<<nwg, threads, 0, st>>>(b, tokens, errors, NB, with_data);
            CK(hipEventRecord(e1, st));
            CK(hipStreamSynchronize(st));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
            CK(hipMemcpy(&herr, errors, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&habort, abortf, 4, hipMemcpyDeviceToHost));
            if (habort) break;
        }
        printf("%s {\"threads\": %d, \"workgroups\": %d, \"with_data\": %d, \"us_per_barrier\": %.3f, \"stale_reads\": %u, \"aborted\": %u}",
               first ? "" : ",\n", threads, nwg, with_data, best * 1000.f / NB, herr, habort);
        first = false;
        if (habort) { printf("]}\n"); return 2; }
    }
    printf("\n],\n\"clip_local_barriers\": [\n");
    {
        unsigned* counters; CK(hipMalloc(&counters, 8 * 64 * 4));
        first = true;
        for (int with_data : {0, 1}) {
            float best = 1e30f; unsigned herr = 0, habort = 0;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipMemsetAsync(counters, 0, 8 * 64 * 4, st)); CK(hipMemsetAsync(abortf, 0, 4, st)); CK(hipMemsetAsync(errors, 0, 4, st));
                CK(hipEventRecord(e0, st));
                persistent_local_kernel<<<256, 256, 0, st>>>(counters, abortf, tokens, errors, NB, with_data);
                CK(hipEventRecord(e1, st));
                CK(hipStreamSynchronize(st));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
                CK(hipMemcpy(&herr, errors, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&habort, abortf, 4, hipMemcpyDeviceToHost));
                if (habort) break;
            }
            printf("%s {\"workgroups\": 256, \"groups\": 8, \"group_size\": 32, \"with_data\": %d, \"us_per_barrier\": %.3f, \"stale_reads\": %u, \"aborted\": %u}",
                   first ? "" : ",\n", with_data, best * 1000.f / NB, herr, habort);
            first = false;
            if (habort) { printf("]}\n"); return 2; }
        }
    }
    printf("\n],\n\"launch_boundaries\": [\n");
    first = true;
    for (int nwg : {32, 256}) {
        // stream order
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < NB; ++i) tiny_kernel<<<nwg, 256, 0, st>>>(tokens, i);
            CK(hipEventRecord(e1, st));
            CK(hipStreamSynchronize(st));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        printf("%s {\"mode\": \"stream\", \"workgroups\": %d, \"us_per_launch\": %.3f}", first ? "" : ",\n", nwg, best * 1000.f / NB);
        first = false;
        // graph replay
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < 200; ++i) tiny_kernel<<<nwg, 256, 0, st>>>(tokens, i);
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
        best = 1e30f;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipEventRecord(e0, st));
            CK(hipGraphLaunch(ge, st));
            CK(hipEventRecord(e1, st));
            CK(hipStreamSynchronize(st));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        printf(",\n {\"mode\": \"graph\", \"workgroups\": %d, \"us_per_launch\": %.3f}", nwg, best * 1000.f / 200);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    printf("\n]}\n");
    return 0;
}
