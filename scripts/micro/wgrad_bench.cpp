// Stand-alone timing / A-B harness of the fused weight-gradient kernel (hyb_conv3x3_wgrad_fused, csrc/conv_wgrad.hip).
//   wgrad_bench <lib.so> [<ref.so>] : for the three config-2 shapes (stages 2-4, N = 128) run the fused wgrad of <lib.so>, print the
//   average time of the contraction launch sequence and, with <ref.so>, compare dW / dyraw against that build (same inputs).
// Each library is conv_wgrad.hip (+ wgrad_stub.hip) compiled on its own, e.g. with -DHYB_ABL=<bits> for the timing-only ablations.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef int (*fused_fn)(int dtype, const void* x, const void* y, const void* dp, const float* ss, const float* mi, const float* gamma,
                        const float* sums, int training, long long count, void* dyraw_out, long long dyraw_blk, float* dw, int N, int H, int W,
                        int Ci, int Cip, int Co, int Cop, void* workspace, size_t workspace_bytes, hipStream_t st);
typedef size_t (*ws_fn)(int first, int N, int H, int W, int Cip, int Cop);

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

static uint64_t rng_state = 0x1234567ull;
static inline uint32_t rnd() { rng_state = rng_state * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(rng_state >> 33); }
static inline float urand() { return (rnd() & 0xffffff) * (1.0f / 16777216.0f); }
static inline float nrand() { float s = 0; for (int i = 0; i < 4; ++i) s += urand(); return (s - 2.0f) * 1.7320508f; }
static inline uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (uint16_t)(u >> 16); }
static inline float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }

struct Lib { void* h; fused_fn fused; ws_fn ws; };
static Lib load(const char* path) {
    Lib l;
    l.h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    if (!l.h) { fprintf(stderr, "dlopen %s: %s\n", path, dlerror()); exit(1); }
    l.fused = (fused_fn)dlsym(l.h, "wgb_fused");
    l.ws = (ws_fn)dlsym(l.h, "hyb_conv3x3_wgrad_workspace");
    if (!l.fused || !l.ws) { fprintf(stderr, "symbols missing in %s\n", path); exit(1); }
    return l;
}

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: wgrad_bench lib.so [ref.so] [reps]\n"); return 2; }
    Lib lib = load(argv[1]);
    const bool have_ref = argc > 2 && strcmp(argv[2], "-") != 0;
    Lib ref{};
    if (have_ref) ref = load(argv[2]);
    const int reps = argc > 3 ? atoi(argv[3]) : 50;
    const int N = 128;
    const int shapes[3][3] = {{112, 32, 64}, {56, 64, 128}, {28, 128, 256}};
    hipStream_t st;
    CK(hipStreamCreate(&st));
    for (int s = (getenv("WGB_FIRST") ? atoi(getenv("WGB_FIRST")) : 0); s < 3; ++s) {
        const int H = shapes[s][0], W = H, Ci = shapes[s][1], Co = shapes[s][2];
        const size_t nx = (size_t)N * H * W * Ci, ny = (size_t)N * H * W * Co, np = ny / 4;
        std::vector<uint16_t> hx(nx), hy(ny), hp(np);
        for (auto& v : hx) v = f2bf(urand());
        for (auto& v : hy) v = f2bf(nrand());
        for (auto& v : hp) v = f2bf(nrand() * 0.1f);
        std::vector<float> ss(2 * Co), mi(2 * Co), gamma(Co), sums(2 * Co);
        for (int c = 0; c < Co; ++c) {
            ss[c] = (0.5f + urand()) * ((c % 7 == 3) ? -1.f : 1.f); ss[Co + c] = nrand() * 0.3f;
            mi[c] = nrand() * 0.1f; mi[Co + c] = 0.5f + 1.5f * urand();
            gamma[c] = ss[c] / mi[Co + c];
            sums[c] = nrand() * 100.f; sums[Co + c] = nrand() * 100.f;
        }
        void *dx, *dy, *dp, *dyraw[2]; float *dss, *dmi, *dg, *dsu, *dw[2]; void* ws;
        CK(hipMalloc(&dx, nx * 2)); CK(hipMalloc(&dy, ny * 2)); CK(hipMalloc(&dp, np * 2));
        CK(hipMalloc(&dyraw[0], ny * 2)); CK(hipMalloc(&dyraw[1], ny * 2));
        CK(hipMalloc(&dss, 8 * Co)); CK(hipMalloc(&dmi, 8 * Co)); CK(hipMalloc(&dg, 4 * Co)); CK(hipMalloc(&dsu, 8 * Co));
        CK(hipMalloc(&dw[0], (size_t)Co * Ci * 9 * 4)); CK(hipMalloc(&dw[1], (size_t)Co * Ci * 9 * 4));
        size_t wsb = lib.ws(0, N, H, W, Ci, Co);
        if (have_ref) { size_t r = ref.ws(0, N, H, W, Ci, Co); if (r > wsb) wsb = r; }
        CK(hipMalloc(&ws, wsb));
        CK(hipMemcpy(dx, hx.data(), nx * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dy, hy.data(), ny * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(dp, hp.data(), np * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(dss, ss.data(), 8 * Co, hipMemcpyHostToDevice)); CK(hipMemcpy(dmi, mi.data(), 8 * Co, hipMemcpyHostToDevice));
        CK(hipMemcpy(dg, gamma.data(), 4 * Co, hipMemcpyHostToDevice)); CK(hipMemcpy(dsu, sums.data(), 8 * Co, hipMemcpyHostToDevice));
        const long long blk = (long long)N * H * W * 32;
        auto run = [&](Lib& l, int which) {
            CK(hipMemsetAsync(dyraw[which], 0xff, ny * 2, st));
            int rc = l.fused(1, dx, dy, dp, dss, dmi, dg, dsu, 1, (long long)N * H * W, dyraw[which], blk, dw[which], N, H, W, Ci, Ci, Co, Co, ws, wsb, st);
            if (rc) { fprintf(stderr, "fused rc %d\n", rc); exit(1); }
        };
        run(lib, 0); CK(hipStreamSynchronize(st));
        // timing: the whole launch sequence (contraction + slab reduce) and, via dw = NULL, the contraction alone
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        float t_all = 0, t_k = 0;
        for (int pass = 0; pass < 2; ++pass) {
            float* dwp = pass == 0 ? dw[0] : nullptr;
            for (int i = 0; i < 3; ++i) lib.fused(1, dx, dy, dp, dss, dmi, dg, dsu, 1, (long long)N * H * W, dyraw[0], blk, dwp, N, H, W, Ci, Ci, Co, Co, ws, wsb, st);
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < reps; ++i) lib.fused(1, dx, dy, dp, dss, dmi, dg, dsu, 1, (long long)N * H * W, dyraw[0], blk, dwp, N, H, W, Ci, Ci, Co, Co, ws, wsb, st);
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            (pass == 0 ? t_all : t_k) = ms * 1000.f / reps;
        }
        const double flops = 2.0 * 9 * Ci * Co * (double)N * H * W;
        const double bytes = (double)(nx + 2 * ny + np) * 2;
        printf("stage%d H=%d Ci=%d Co=%d : kernel %.1f us (%.0f TF/s, %.2f TB/s algorithmic)  with reduce %.1f us\n", s + 2, H, Ci, Co, t_k,
               flops / t_k / 1e6, bytes / t_k / 1e6, t_all);
        if (have_ref) {
            run(lib, 0); run(ref, 1); CK(hipStreamSynchronize(st));
            std::vector<float> a((size_t)Co * Ci * 9), b(a.size());
            CK(hipMemcpy(a.data(), dw[0], a.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), dw[1], b.size() * 4, hipMemcpyDeviceToHost));
            double maxd = 0, maxr = 0;
            for (size_t i = 0; i < a.size(); ++i) { maxd = fmax(maxd, fabs((double)a[i] - b[i])); maxr = fmax(maxr, fabs((double)b[i])); }
            std::vector<uint16_t> ra(ny), rb(ny);
            CK(hipMemcpy(ra.data(), dyraw[0], ny * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(rb.data(), dyraw[1], ny * 2, hipMemcpyDeviceToHost));
            size_t ndiff = 0;
            for (size_t i = 0; i < ny; ++i) ndiff += ra[i] != rb[i];
            printf("   vs ref: dW max|d| %.3e / max|ref| %.3e = %.2e ; dyraw words differing %zu of %zu\n", maxd, maxr, maxd / (maxr + 1e-30), ndiff, ny);
        }
        hipFree(dx); hipFree(dy); hipFree(dp); hipFree(dyraw[0]); hipFree(dyraw[1]); hipFree(dss); hipFree(dmi); hipFree(dg); hipFree(dsu);
        hipFree(dw[0]); hipFree(dw[1]); hipFree(ws);
    }
    return 0;
}
