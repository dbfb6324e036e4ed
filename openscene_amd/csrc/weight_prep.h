// Weight images of the split-bf16 convolution kernels (device functions shared by the per-weight entry points in
// spconv.hip / spconv_tl.hip and the batched entry point in weight_prep.hip).
#pragma once
#include "common.h"

namespace osn {

// Wp[plane][k][n][c] (bf16, c padded to cp with zeros) = piece `plane` of the weight that multiplies input
// channel c into output channel n at offset k:  forward  W[k][c][n];  input gradient  W[flip ? K-1-k : k][n][c].
__device__ __forceinline__ void weight_prep_x6_one(const float* __restrict__ W, int K, int cin, int cout, int flip,
                                                    int for_dgrad, int64_t e, __bf16* __restrict__ Wp) {
    const int nn = for_dgrad ? cin : cout, nc = for_dgrad ? cout : cin;
    const int cp = (nc + 31) / 32 * 32;
    const int64_t per_plane = int64_t(K) * nn * cp;
    const int c = int(e % cp);
    const int n = int((e / cp) % nn);
    const int k = int(e / (int64_t(cp) * nn));
    float v = 0.f;
    if (c < nc) {
        const int ks = flip ? K - 1 - k : k;
        v = for_dgrad ? W[(int64_t(ks) * cin + n) * cout + c] : W[(int64_t(ks) * cin + c) * cout + n];
    }
    const __bf16 h1 = (__bf16)v;
    const float r1 = v - (float)h1;
    const __bf16 h2 = (__bf16)r1;
    const float r2 = r1 - (float)h2;
    Wp[e] = h1;
    Wp[per_plane + e] = h2;
    Wp[2 * per_plane + e] = (__bf16)r2;
}


// MFMA-ready image of a weight: Wp[plane][k][s][cb][lane][8] bf16, one 1 KB block per (k, 32-deep k-step s,
// 16-column block cb, plane); lane l of a wave holds the B fragment of v_mfma_f32_16x16x32_bf16:
//   element e of lane l = piece `plane` of  B[c = 32 s + 8 (l >> 4) + e][n = 16 cb + (l & 15)]
// with B[c][n] = W[k][c][n] (forward) or W[flip ? K-1-k : k][n][c] (input gradient: contraction over the
// conv's OUTPUT channels).  Channels / columns beyond the real ones are zero.
__device__ __forceinline__ void weight_prep_tl_one(const float* __restrict__ W, int K, int cin, int cout, int flip,
                                                    int for_dgrad, int64_t e, __bf16* __restrict__ Wp) {
    const int nc = for_dgrad ? cout : cin;     // contraction length
    const int nn = for_dgrad ? cin : cout;     // columns of B
    const int ns = (nc + 31) >> 5, ncb = (nn + 15) >> 4;
    const int64_t per_plane = int64_t(K) * ns * ncb * 512;
    const int el = int(e & 7);
    const int lane = int((e >> 3) & 63);
    int64_t blk = e >> 9;
    const int cb = int(blk % ncb); blk /= ncb;
    const int s = int(blk % ns);
    const int k = int(blk / ns);
    const int c = 32 * s + 8 * (lane >> 4) + el;
    const int n = 16 * cb + (lane & 15);
    float v = 0.f;
    if (c < nc && n < nn) {
        const int ks = flip ? K - 1 - k : k;
        v = for_dgrad ? W[(int64_t(ks) * cin + n) * cout + c] : W[(int64_t(ks) * cin + c) * cout + n];
    }
    const __bf16 h1 = (__bf16)v;
    const float r1 = v - (float)h1;
    const __bf16 h2 = (__bf16)r1;
    const float r2 = r1 - (float)h2;
    Wp[e] = h1;
    Wp[per_plane + e] = h2;
    Wp[2 * per_plane + e] = (__bf16)r2;
}

// The same image, eight elements at a time: group g = the 8 consecutive contraction indices one lane of a B fragment holds
// (element e = 8 g + el): one index computation and three 16-byte stores instead of eight of each and 24 two-byte stores.
__device__ __forceinline__ void weight_prep_tl_group(const float* __restrict__ W, int K, int cin, int cout, int flip,
                                                      int for_dgrad, int64_t g, __bf16* __restrict__ Wp) {
    typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
    const int nc = for_dgrad ? cout : cin;
    const int nn = for_dgrad ? cin : cout;
    const int ns = (nc + 31) >> 5, ncb = (nn + 15) >> 4;
    const int64_t per_plane = int64_t(K) * ns * ncb * 512;
    const int lane = int(g & 63);
    int64_t blk = g >> 6;
    const int cb = int(blk % ncb); blk /= ncb;
    const int s = int(blk % ns);
    const int k = int(blk / ns);
    const int c0 = 32 * s + 8 * (lane >> 4);
    const int n = 16 * cb + (lane & 15);
    const int ks = flip ? K - 1 - k : k;
    bf16x8_t p1, p2, p3;
#pragma unroll
    for (int el = 0; el < 8; ++el) {
        const int c = c0 + el;
        float v = 0.f;
        if (c < nc && n < nn) v = for_dgrad ? W[(int64_t(ks) * cin + n) * cout + c] : W[(int64_t(ks) * cin + c) * cout + n];
        const __bf16 h1 = (__bf16)v;
        const float r1 = v - (float)h1;
        const __bf16 h2 = (__bf16)r1;
        const float r2 = r1 - (float)h2;
        p1[el] = h1; p2[el] = h2; p3[el] = (__bf16)r2;
    }
    *reinterpret_cast<bf16x8_t*>(Wp + 8 * g) = p1;
    *reinterpret_cast<bf16x8_t*>(Wp + per_plane + 8 * g) = p2;
    *reinterpret_cast<bf16x8_t*>(Wp + 2 * per_plane + 8 * g) = p3;
}

}  // namespace osn
