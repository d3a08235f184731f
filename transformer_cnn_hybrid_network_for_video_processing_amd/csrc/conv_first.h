// Shared declarations of the stage-1 kernels (conv_first.hip: block-level kernels and the backward pass; conv_first_wave.hip: the
// wave-private forward passes, compiled with MFMA results in VGPRs).
#pragma once
#include "hyb_common.h"

constexpr int S1_KP = 64;                      // padded K of the packed first-layer weights

// 4 channels of one pixel as one 8-byte (bf16) / 16-byte (fp32) LDS access
template <typename T> struct alignas(sizeof(T) * 4) Quad { T v[4]; };

struct S1Args {
    const float* x;          // [N,Ci,H,W] fp32
    const void* wp;          // packed weights T [Cop][64], k = tap*4 + c
    const void* wp2;         // the same weights in the wave-private kernels' K order (s1w_pack_kernel)
    const float* ss;         // scale/shift [2][Cop]
    const float* mi;         // mean/invstd [2][Cop]
    const float* gamma;      // [Co]
    const float* sums;       // [2][Cop] (sum dy, sum dy*xhat)
    const void* dp;          // dpooled NHWC T [N,H/2,W/2,Cop]
    void* pooled;            // pooled NHWC T
    float* part;             // per-workgroup partial rows (stats / sums / wgrad slabs)
    int N, H, W, Ci, Co, Cop;
    int training;
    float inv_count;
    int tilesX, tilesY, numTiles;
    float inv_tpi, inv_tx;   // 1 / (tilesX*tilesY), 1 / tilesX: tile -> (n, ty, tx) without integer division
    int vec_ok;              // x 16-byte aligned and W % 4 == 0: halo rows are loaded as aligned float4
    unsigned* route;         // NULL, or the routing codes of the apply + pool pass, one 32-bit word per (pooled pixel, 8 channels): [N,H/2,W/2,Cop/8]
                             // (4 bits per channel: bits 0-1 = which pixel of the 2x2 window is the first maximum, bit 2 = the ReLU passed).
                             // Written by the wave-private forward pass, read by the wave-private backward pass INSTEAD of recomputing the conv.
};

// One element of the packed first-layer weights [2][Cop][64] (shared by s1w_pack_kernel and the backbone's one pack launch, conv_fwd.hip):
// half 0 in the block-level kernels' K order (k = tap*4 + c), half 1 in the wave-private kernels' order (k = slot*4 + c, slot -> tap below;
// slot pairs (2q, 2q+1) of a k-step are horizontally adjacent pixels; -1 = zero weights).  i indexes one half: co = i / 64, k = i % 64.
__device__ __forceinline__ float s1w_pack_value(const float* __restrict__ w, long long i, bool second, int Co, int Ci) {
    const int k = (int)(i % 64), co = (int)(i / 64);
    const int slot = k >> 2, c = k & 3;
    //                  k-step 0: (0,0) (0,1) (1,0) (1,1) (2,0) (2,1) (0,2)  -    k-step 1: (1,2) -  (2,2) -   -   -   -   -
    const int tap_of_slot[16] = {0, 1, 3, 4, 6, 7, 2, -1, 5, -1, 8, -1, -1, -1, -1, -1};
    const int tap = second ? tap_of_slot[slot] : (slot < 9 ? slot : -1);
    return (co < Co && tap >= 0 && c < Ci) ? w[((long long)co * Ci + c) * 9 + tap] : 0.f;
}

// conv_first_wave.hip
int hyb_stage1w_pack(int dtype, const float* weight, void* wp2, int Co, int Ci, int Cop, int both, hipStream_t st);
int hyb_stage1w_bwd(int dtype, S1Args a, int with_g /* 0: rows are Cop x 48 (G saved by the forward pass) */, int& grid_x /* in: wanted workgroups; out: launched = partial rows */, hipStream_t st);
int hyb_stage1w_gram(int dtype, S1Args a, int& grid_x, hipStream_t st);
constexpr int S1_GRAM_DOUBLES = 2304 + 2;          // saved Gram matrix (48 x 48, upper-triangle tiles) + pad, after the two packed weight layouts
int hyb_stage1w_pass(int dtype, int mode /* 0 statistics, 1 apply + pool */, const S1Args& a, int& grid_x /* in: wanted workgroups; out: launched */,
                     hipStream_t st);
