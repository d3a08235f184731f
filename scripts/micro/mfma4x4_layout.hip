// Layout probe of v_mfma_f32_4x4x1_16B_f32 on gfx950: 16 independent 4x4 outer products per instruction.
// Hypothesis checked here: lane l -> block b = l / 4; A operand = A[b][i = l % 4]; B operand = B[b][j = l % 4];
// result register r of lane l = D[b][i = r][j = l % 4] = A[b][r] * B[b][l % 4].
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(const float* a, const float* b, float* d) {
    const int l = threadIdx.x;
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) d[l * 4 + r] = c[r];
}
int main() {
    float ha[64], hb[64], hd[256];
    for (int l = 0; l < 64; ++l) { ha[l] = 1.0f + l; hb[l] = 100.0f + 3 * l; }
    float *a, *b, *d;
    hipMalloc(&a, 256); hipMalloc(&b, 256); hipMalloc(&d, 1024);
    hipMemcpy(a, ha, 256, hipMemcpyHostToDevice); hipMemcpy(b, hb, 256, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(a, b, d);
    hipMemcpy(hd, d, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 4; ++r) {
            const int blk = l / 4;
            const float want = ha[blk * 4 + r] * hb[l];
            if (hd[l * 4 + r] != want) { if (bad < 8) printf("lane %d reg %d: got %g want %g\n", l, r, hd[l * 4 + r], want); ++bad; }
        }
    printf("{\"mfma_4x4x1_layout_hypothesis\": \"%s\", \"mismatches\": %d}\n", bad ? "WRONG" : "confirmed", bad);
    return 0;
}
