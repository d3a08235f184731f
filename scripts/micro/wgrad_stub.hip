// Link stub for scripts/micro/wgrad_bench: the two globals conv_wgrad.hip expects from bn_pool.hip, and a C-linkage door to the
// library-internal fused entry point.
#include "hyb_common.h"
HybProfileHook g_hyb_hooks[16];
int g_hyb_hooks_active = 0;
int hyb_conv3x3_wgrad_fused(int dtype, const void* x, const void* y, const void* dp, const float* ss, const float* mi, const float* gamma,
                            const float* sums, int training, long long count, void* dyraw_out, long long dyraw_blk, float* dw, int N, int H, int W,
                            int Ci, int Cip, int Co, int Cop, void* workspace, size_t workspace_bytes, hipStream_t st);
extern "C" int wgb_fused(int dtype, const void* x, const void* y, const void* dp, const float* ss, const float* mi, const float* gamma,
                         const float* sums, int training, long long count, void* dyraw_out, long long dyraw_blk, float* dw, int N, int H, int W,
                         int Ci, int Cip, int Co, int Cop, void* workspace, size_t workspace_bytes, hipStream_t st) {
    return hyb_conv3x3_wgrad_fused(dtype, x, y, dp, ss, mi, gamma, sums, training, count, dyraw_out, dyraw_blk, dw, N, H, W, Ci, Cip, Co, Cop, workspace,
                                   workspace_bytes, st);
}
