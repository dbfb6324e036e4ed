// The 3-channel stem convolution of the U-Net (conv0p1s1: 5^3 kernel, 3 -> 32 channels,
// models/mink_unet.py:47-50) on gfx950: plain fp32 VALU kernels.
//
// With 3 input channels there is no contraction worth a matrix unit (96 FMAs per pair); what the op does is
// stream the 125 x N neighbour table once (50 MB on S100k) and gather 12 bytes per pair.  The generic MFMA
// kernel spent 191 us on it; this one 127 us (measured; a pair-driven variant would skip the 89 % empty
// table entries).  One thread per output row, 32 accumulators, offsets in ascending order: a plain fmaf
// chain, bitwise reproducible, exact fp32 products.
// Weight gradient (round 3): the generic fp32-MFMA table kernel needs 260 us for it, and the stem is the LAST weight
// gradient of a backward pass -- nothing is left to hide it behind.  stem_wgrad_kernel treats the table densely (an absent
// neighbour contributes a zero row): per chunk of 64 output rows a workgroup stages the gradient rows and, for 32 offsets,
// the gathered input rows in LDS, and thread (4 offsets, channel n) runs a plain fmaf chain down the rows -- LDS-broadcast
// operands, 12 accumulators, no cross-lane reduction; per-workgroup partial sums, summed in workgroup order.
#include "common.h"
#include "epilogue.h"
#include "split.h"

namespace osn {

constexpr int STEM_CMAX = 4;       // input channels at most
constexpr int64_t STEM_MFMA_MIN_ROWS = 32768;   // output rows from which the forward convolution runs on the matrix cores (stem_mfma_fwd_kernel)
constexpr int STEM_COUT = 32;      // output channels handled per thread

// out[o][0..31] = sum_k in[nbr[k][o]][0..cin) @ W[k]      (cout == 32)
// One thread per output row; the offset loop is unrolled by 5 with the five table entries loaded up front (a
// 100 k-row table gives only 1.5 waves per SIMD, so the latency of the dependent table -> row -> FMA chain has to
// be covered by instruction-level parallelism); W[k] is indexed by wave-uniform values only, so it is fetched
// through the scalar cache (staging it in LDS per workgroup cost more than the whole convolution).
template <bool EPI>
__global__ __launch_bounds__(256) void stem_fwd_kernel(const float* __restrict__ in, const float* __restrict__ W,
                                                       const int32_t* __restrict__ nbr, float* __restrict__ out,
                                                       int64_t n_out, int K, int cin, const Epi epi) {
    const int64_t o = int64_t(blockIdx.x) * 256 + threadIdx.x;
    const bool row_ok = o < n_out;
    const int64_t oc = row_ok ? o : 0;
    float acc[STEM_COUT];
#pragma unroll
    for (int c = 0; c < STEM_COUT; ++c) acc[c] = 0.f;
    for (int k0 = 0; k0 < K; k0 += 5) {
        int idx[5];
#pragma unroll
        for (int u = 0; u < 5; ++u) idx[u] = (row_ok && k0 + u < K) ? nbr[int64_t(k0 + u) * n_out + oc] : -1;
        float x[5][STEM_CMAX];
#pragma unroll
        for (int u = 0; u < 5; ++u)
#pragma unroll
            for (int ci = 0; ci < STEM_CMAX; ++ci)
                x[u][ci] = (idx[u] >= 0 && ci < cin) ? in[int64_t(idx[u]) * cin + ci] : 0.f;
#pragma unroll
        for (int u = 0; u < 5; ++u) {
            if (k0 + u < K) {
                const float* w = W + int64_t(k0 + u) * cin * STEM_COUT;        // wave-uniform address
#pragma unroll
                for (int ci = 0; ci < STEM_CMAX; ++ci)
                    if (ci < cin) {
#pragma unroll
                        for (int c = 0; c < STEM_COUT; ++c) acc[c] = fmaf(x[u][ci], w[ci * STEM_COUT + c], acc[c]);
                    }
            }
        }
    }
    if (row_ok) {
        float4* dst = reinterpret_cast<float4*>(out + o * STEM_COUT);
#pragma unroll
        for (int c = 0; c < STEM_COUT / 4; ++c) {
            float4 v = make_float4(acc[4 * c], acc[4 * c + 1], acc[4 * c + 2], acc[4 * c + 3]);
            if constexpr (EPI) v = epi_quad(epi, v, o, 4 * c, STEM_COUT);        // evaluation-mode batch norm (epilogue.h)
            dst[c] = v;
        }
    }
}

// The same convolution with the offsets of a row dealt to the FOUR WAVES of a workgroup (round 3): wave w takes the offsets
// k = w, w + 4, ... of the workgroup's 64 rows (one thread per row leaves a 100 k-row map with 1.5 waves per SIMD and a chain
// of 125 dependent table -> row -> FMA steps; this gives 6 waves per SIMD and a chain of 32, and k stays wave-uniform, so
// the weights still come through the scalar cache), the four partial rows meet in LDS and are added in a fixed tree
// ((p0 + p1) + (p2 + p3)): deterministic, fp32 products exact.
template <bool EPI>
__global__ __launch_bounds__(256) void stem_fwd4_kernel(const float* __restrict__ in, const float* __restrict__ W,
                                                        const int32_t* __restrict__ nbr, float* __restrict__ out,
                                                        int64_t n_out, int K, int cin, const Epi epi) {
    __shared__ float part[4][STEM_COUT][64 + 1];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t o = int64_t(blockIdx.x) * 64 + lane;
    const bool row_ok = o < n_out;
    const int64_t oc = row_ok ? o : 0;
    float acc[STEM_COUT];
#pragma unroll
    for (int c = 0; c < STEM_COUT; ++c) acc[c] = 0.f;
    for (int k0 = wave; k0 < K; k0 += 16) {                   // 4 offsets of this wave per trip: k0, k0 + 4, k0 + 8, k0 + 12
        int idx[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) idx[u] = (row_ok && k0 + 4 * u < K) ? nbr[int64_t(k0 + 4 * u) * n_out + oc] : -1;
        float x[4][STEM_CMAX];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int ci = 0; ci < STEM_CMAX; ++ci)
                x[u][ci] = (idx[u] >= 0 && ci < cin) ? in[int64_t(idx[u]) * cin + ci] : 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (k0 + 4 * u < K) {
                const float* w = W + int64_t(k0 + 4 * u) * cin * STEM_COUT;        // wave-uniform address
#pragma unroll
                for (int ci = 0; ci < STEM_CMAX; ++ci)
                    if (ci < cin) {
#pragma unroll
                        for (int c = 0; c < STEM_COUT; ++c) acc[c] = fmaf(x[u][ci], w[ci * STEM_COUT + c], acc[c]);
                    }
            }
        }
    }
#pragma unroll
    for (int c = 0; c < STEM_COUT; ++c) part[wave][c][lane] = acc[c];
    __syncthreads();
    // thread (row r, quarter q): channels 8 q .. 8 q + 7 of row r
    const int r = threadIdx.x >> 2, q = threadIdx.x & 3;
    const int64_t orow = int64_t(blockIdx.x) * 64 + r;
    if (orow < n_out) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = 8 * q + j;
            v[j] = (part[0][c][r] + part[1][c][r]) + (part[2][c][r] + part[3][c][r]);
        }
        float4* dst = reinterpret_cast<float4*>(out + orow * STEM_COUT + 8 * q);
        float4 v0 = make_float4(v[0], v[1], v[2], v[3]), v1 = make_float4(v[4], v[5], v[6], v[7]);
        if constexpr (EPI) {                                                     // evaluation-mode batch norm (epilogue.h)
            v0 = epi_quad(epi, v0, orow, 8 * q, STEM_COUT);
            v1 = epi_quad(epi, v1, orow, 8 * q + 4, STEM_COUT);
        }
        dst[0] = v0;
        dst[1] = v1;
    }
}

// Round 6: the stem convolution of the LARGE maps on the matrix cores.  The plain-fp32 kernels above walk the table densely: 125 x 96
// FMAs per row of which 89 % multiply the zero row of an absent neighbour, and a wave64 VALU instruction holds its 16-lane SIMD for four
// cycles -- 81 us for 0.27 GFLOP of real work, alone on the device at the head of every forward pass.  As a matrix product the same walk
// is cheap: the convolution is  out[o][0..32) = A[o][0..8 P) @ B[0..8 P)][0..32)  with P = ceil(K / 2) "offset pairs",
//   A[o][8 v + 4 h + c] = in[nbr[2 v + h][o]][c]   (c < cin, zero otherwise / for an absent neighbour),   B[8 v + 4 h + c][n] = W[2 v + h][c][n],
// i.e. EIGHT contraction elements per lane are two table entries and two gathered input rows: the register-gather kernel's scheme
// (spconv_rg.hip) with the A operand assembled from two neighbours.  Four waves split the 32-deep k-steps (four offset pairs each) of a
// 64-row workgroup, every wave splits its B fragments from the fp32 weight itself (no weight image: the stem's kernel changes every
// step and is 48 KB), partial tiles are summed in wave order through LDS.  Arithmetic: "bf16x6" like every other convolution here
// (three bf16 pieces per operand, six MFMAs per block, fp32 accumulate: fp32-class, not the exact fmaf chain of the kernels above --
// the small maps keep those).
typedef float stem_f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 stem_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 stem_bf16x4 __attribute__((ext_vector_type(4)));

template <bool EPI>
__global__ __launch_bounds__(256, 2) void stem_mfma_fwd_kernel(const float* __restrict__ in, const float* __restrict__ W,
                                                               const int32_t* __restrict__ nbr, float* __restrict__ out,
                                                               int64_t n_out, int K, int cin, const Epi epi) {
    constexpr int LDR = 32 + 4;
    __shared__ __attribute__((aligned(16))) float red[4][64][LDR];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lg = lane >> 4;
    const int64_t row0 = int64_t(blockIdx.x) * 64;
    const int n_steps = ((K + 1) / 2 + 3) / 4;                        // 32-deep k-steps (four offset pairs each)
    stem_f32x4 acc[4][2];
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) acc[rb][cb] = stem_f32x4{0.f, 0.f, 0.f, 0.f};

    for (int s = wave; s < n_steps; s += 4) {
        const int k0 = 2 * (4 * s + lg), k1 = k0 + 1;               // this lane's two offsets
        // ---- table entries and input rows of the four row blocks (absent neighbour / offset past the kernel: zeros)
        int i0[4], i1[4];
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) {
            const int64_t r = row0 + 16 * rb + l15;
            const bool on = r < n_out;
            i0[rb] = (on && k0 < K) ? nbr[int64_t(k0) * n_out + r] : -1;
            i1[rb] = (on && k1 < K) ? nbr[int64_t(k1) * n_out + r] : -1;
        }
        // ---- B fragments of this k-step from the fp32 weight: element e of the lane = W[k0 + (e >> 2)][e & 3][16 cb + l15]
        stem_bf16x8 B[2][3];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            float4 w0 = make_float4(0.f, 0.f, 0.f, 0.f), w1 = make_float4(0.f, 0.f, 0.f, 0.f);
            const int n = 16 * cb + l15;
            if (k0 < K) {
                const float* p = W + int64_t(k0) * cin * STEM_COUT + n;
                w0.x = p[0];
                if (cin > 1) w0.y = p[STEM_COUT];
                if (cin > 2) w0.z = p[2 * STEM_COUT];
                if (cin > 3) w0.w = p[3 * STEM_COUT];
            }
            if (k1 < K) {
                const float* p = W + int64_t(k1) * cin * STEM_COUT + n;
                w1.x = p[0];
                if (cin > 1) w1.y = p[STEM_COUT];
                if (cin > 2) w1.z = p[2 * STEM_COUT];
                if (cin > 3) w1.w = p[3 * STEM_COUT];
            }
            stem_bf16x4 a1, a2, a3, b1, b2, b3;
            tl_split4(w0, a1, a2, a3);
            tl_split4(w1, b1, b2, b3);
            B[cb][0] = __builtin_shufflevector(a1, b1, 0, 1, 2, 3, 4, 5, 6, 7);
            B[cb][1] = __builtin_shufflevector(a2, b2, 0, 1, 2, 3, 4, 5, 6, 7);
            B[cb][2] = __builtin_shufflevector(a3, b3, 0, 1, 2, 3, 4, 5, 6, 7);
        }
        float4 x0[4], x1[4];
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) {
            x0[rb] = make_float4(0.f, 0.f, 0.f, 0.f);
            x1[rb] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i0[rb] >= 0) {
                const float* p = in + int64_t(i0[rb]) * cin;
                x0[rb].x = p[0];
                if (cin > 1) x0[rb].y = p[1];
                if (cin > 2) x0[rb].z = p[2];
                if (cin > 3) x0[rb].w = p[3];
            }
            if (i1[rb] >= 0) {
                const float* p = in + int64_t(i1[rb]) * cin;
                x1[rb].x = p[0];
                if (cin > 1) x1[rb].y = p[1];
                if (cin > 2) x1[rb].z = p[2];
                if (cin > 3) x1[rb].w = p[3];
            }
        }
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) {
            // (a row block none of whose 16 rows x 8 offsets has a neighbour: nothing to add -- wave-uniform)
            if (__ballot(i0[rb] >= 0 || i1[rb] >= 0) == 0ull) continue;
            stem_bf16x4 a1, a2, a3, b1, b2, b3;
            tl_split4(x0[rb], a1, a2, a3);
            tl_split4(x1[rb], b1, b2, b3);
            const stem_bf16x8 A1 = __builtin_shufflevector(a1, b1, 0, 1, 2, 3, 4, 5, 6, 7);
            const stem_bf16x8 A2 = __builtin_shufflevector(a2, b2, 0, 1, 2, 3, 4, 5, 6, 7);
            const stem_bf16x8 A3 = __builtin_shufflevector(a3, b3, 0, 1, 2, 3, 4, 5, 6, 7);
            // smallest terms first per accumulator; consecutive MFMAs on different accumulators
#define STEM_MFMA(AP, BP)                                                                                      \
    _Pragma("unroll") for (int cb = 0; cb < 2; ++cb)                                                           \
        acc[rb][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(B[cb][BP], AP, acc[rb][cb], 0, 0, 0);
            STEM_MFMA(A3, 0) STEM_MFMA(A2, 1) STEM_MFMA(A1, 2) STEM_MFMA(A2, 0) STEM_MFMA(A1, 1) STEM_MFMA(A1, 0)
#undef STEM_MFMA
        }
    }
    // ---- the four waves' partial tiles -> out, summed in wave order (64 rows x 32 columns through LDS)
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
            *reinterpret_cast<stem_f32x4*>(&red[wave][16 * rb + l15][16 * cb + 4 * lg]) = acc[rb][cb];
    __syncthreads();
    EpiCols ec;
    if constexpr (EPI) ec = epi_cols(epi, 4 * (tid & 7));
    for (int e = tid; e < 64 * 8; e += 256) {
        const int j = e >> 3, c4 = e & 7;
        const int64_t r = row0 + j;
        if (r < n_out) {
            float4 sum = *reinterpret_cast<const float4*>(&red[0][j][4 * c4]);
#pragma unroll
            for (int w = 1; w < 4; ++w) {
                const float4 v = *reinterpret_cast<const float4*>(&red[w][j][4 * c4]);
                sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
            }
            if constexpr (EPI) sum = epi_apply(epi, ec, sum, r, 4 * c4, STEM_COUT);      // evaluation-mode batch norm (epilogue.h)
            *reinterpret_cast<float4*>(out + r * STEM_COUT + 4 * c4) = sum;
        }
    }
}

constexpr int SW_ROWS = 64;        // output rows per chunk
constexpr int SW_KB = 32;          // offsets per workgroup (8 thread groups x 4)
constexpr int SW_PARTS = 128;      // row parts (workgroups along the rows) = partial sums per weight element

// partial[part][k][ci][n] = sum over the part's rows o of in[nbr[k][o]][ci] * gout[o][n]      (cout == 32)
__global__ __launch_bounds__(256) void stem_wgrad_kernel(const float* __restrict__ in, const float* __restrict__ gout,
                                                         const int32_t* __restrict__ nbr, float* __restrict__ partial,
                                                         int64_t n_out, int K, int cin) {
    __shared__ __attribute__((aligned(16))) float A[SW_KB * SW_ROWS * STEM_CMAX];      // [offset][row][channel]
    __shared__ __attribute__((aligned(16))) float G[SW_ROWS * STEM_COUT];
    const int tid = threadIdx.x, n = tid & 31, kq = tid >> 5;
    const int k0 = blockIdx.y * SW_KB;
    float acc[4][STEM_CMAX];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int c = 0; c < STEM_CMAX; ++c) acc[j][c] = 0.f;
    const int64_t n_chunks = (n_out + SW_ROWS - 1) / SW_ROWS;
    for (int64_t ch = blockIdx.x; ch < n_chunks; ch += gridDim.x) {            // ascending rows: fixed summation order
        const int64_t r0 = ch * SW_ROWS;
        // ---- gathered input rows of 32 offsets x 64 rows (zeros for an absent neighbour), gradient rows
        // (256 row parts and the table entries fetched one chunk ahead were measured slower: 150 us against 128 us)
        int idx[SW_KB * SW_ROWS / 256];
#pragma unroll
        for (int u = 0; u < SW_KB * SW_ROWS / 256; ++u) {
            const int e = tid + 256 * u;
            const int kk = e / SW_ROWS, o = e % SW_ROWS;
            idx[u] = (k0 + kk < K && r0 + o < n_out) ? nbr[int64_t(k0 + kk) * n_out + r0 + o] : -1;
        }
#pragma unroll
        for (int u = 0; u < SW_KB * SW_ROWS / 256; ++u) {
            const int e = tid + 256 * u;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx[u] >= 0) {
                const float* x = in + int64_t(idx[u]) * cin;
                v.x = x[0];
                if (cin > 1) v.y = x[1];
                if (cin > 2) v.z = x[2];
                if (cin > 3) v.w = x[3];
            }
            *reinterpret_cast<float4*>(&A[e * STEM_CMAX]) = v;
        }
#pragma unroll
        for (int u = 0; u < SW_ROWS * STEM_COUT / 4 / 256; ++u) {
            const int e = tid + 256 * u;
            const int o = e / (STEM_COUT / 4);
            *reinterpret_cast<float4*>(&G[e * 4]) =
                r0 + o < n_out ? *reinterpret_cast<const float4*>(gout + r0 * STEM_COUT + int64_t(e) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        __syncthreads();
#pragma unroll 4
        for (int o = 0; o < SW_ROWS; ++o) {
            const float g = G[o * STEM_COUT + n];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 a = *reinterpret_cast<const float4*>(&A[((4 * kq + j) * SW_ROWS + o) * STEM_CMAX]);     // broadcast
                acc[j][0] = fmaf(a.x, g, acc[j][0]);
                acc[j][1] = fmaf(a.y, g, acc[j][1]);
                acc[j][2] = fmaf(a.z, g, acc[j][2]);
                acc[j][3] = fmaf(a.w, g, acc[j][3]);
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int k = k0 + 4 * kq + j;
        if (k < K)
            for (int c = 0; c < cin; ++c)
                partial[((int64_t(blockIdx.x) * K + k) * cin + c) * STEM_COUT + n] = acc[j][c];
    }
}

// Round 6: the weight gradient of the large maps on the matrix cores (the last kernel of every backward pass, alone on the device:
// 100 us dense over the table).  gW[k][c][n] = sum over rows o of in[nbr[k][o]][c] * gout[o][n] is the product
//   D_c[k][n] = sum_o A_c[k][o] G[o][n],   A_c[k][o] = in[nbr[k][o]][c]  (zero for an absent neighbour),   one product per input channel c,
// contracted over the ROWS: an MFMA step takes 32 rows, lane (l & 15, l >> 4) of the first operand holds A_c[16 kb + (l & 15)][8 rows]
// = eight consecutive entries of ONE table row (loaded once for all channels) and the gathered values, of the second
// G[8 rows][16 nb + (l & 15)].  A workgroup = 4 waves x 2 blocks of 16 offsets (all 125 offsets x cin x 32 channels), walks the 32-row
// steps blockIdx.x, + gridDim.x, ... in
// ascending order and leaves ONE partial gradient; the partials are summed in workgroup order by stem_wgrad_reduce_kernel
// (deterministic).  Split-bf16 arithmetic ("bf16x6"), fp32 accumulate.
constexpr int SWM_PARTS = 512;     // workgroups (two per CU) = partial gradients of the matrix-core weight gradient

__global__ __launch_bounds__(256, 2) void stem_mfma_wgrad_kernel(const float* __restrict__ in, const float* __restrict__ gout,
                                                                 const int32_t* __restrict__ nbr, float* __restrict__ partial,
                                                                 int64_t n_out, int K, int cin) {
    // wave w owns the offsets 32 w .. 32 w + 31 as two blocks of 16 (lane & 15 = offset within the block), one accumulator set per
    // input channel: a table entry is loaded ONCE and serves the (up to four) channels of its row
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lg = lane >> 4;
    stem_f32x4 acc[2][STEM_CMAX][2];
#pragma unroll
    for (int ob = 0; ob < 2; ++ob)
#pragma unroll
        for (int c = 0; c < STEM_CMAX; ++c)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) acc[ob][c][nb] = stem_f32x4{0.f, 0.f, 0.f, 0.f};
    const int64_t n_steps = (n_out + 31) / 32;
    const int kk[2] = {32 * wave + l15, 32 * wave + 16 + l15};               // this lane's offset in either block
    for (int64_t rs = blockIdx.x; rs < n_steps; rs += gridDim.x) {            // ascending rows: fixed summation order
        const int64_t o0 = rs * 32 + 8 * lg;                                  // this lane's eight rows
        // ---- every load of the step first: table entries of both blocks, gradient rows; then the gathers
        int idx[2][8];
#pragma unroll
        for (int ob = 0; ob < 2; ++ob)
#pragma unroll
            for (int e = 0; e < 8; ++e) idx[ob][e] = (kk[ob] < K && o0 + e < n_out) ? nbr[int64_t(kk[ob]) * n_out + o0 + e] : -1;
        float g[2][8];
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int e = 0; e < 8; ++e) g[nb][e] = o0 + e < n_out ? gout[(o0 + e) * STEM_COUT + 16 * nb + l15] : 0.f;
        float x[2][STEM_CMAX][8];
#pragma unroll
        for (int ob = 0; ob < 2; ++ob)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float* p = in + int64_t(idx[ob][e] >= 0 ? idx[ob][e] : 0) * cin;
#pragma unroll
                for (int c = 0; c < STEM_CMAX; ++c) x[ob][c][e] = (idx[ob][e] >= 0 && c < cin) ? p[c] : 0.f;
            }
        // ---- G fragments: element e of the lane = gout[o0 + e][16 nb + l15]
        stem_bf16x8 G[2][3];
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            stem_bf16x4 a1, a2, a3, b1, b2, b3;
            tl_split4(make_float4(g[nb][0], g[nb][1], g[nb][2], g[nb][3]), a1, a2, a3);
            tl_split4(make_float4(g[nb][4], g[nb][5], g[nb][6], g[nb][7]), b1, b2, b3);
            G[nb][0] = __builtin_shufflevector(a1, b1, 0, 1, 2, 3, 4, 5, 6, 7);
            G[nb][1] = __builtin_shufflevector(a2, b2, 0, 1, 2, 3, 4, 5, 6, 7);
            G[nb][2] = __builtin_shufflevector(a3, b3, 0, 1, 2, 3, 4, 5, 6, 7);
        }
#pragma unroll
        for (int ob = 0; ob < 2; ++ob) {
            bool any = false;
#pragma unroll
            for (int e = 0; e < 8; ++e) any = any || idx[ob][e] >= 0;
            if (__ballot(any) == 0ull) continue;                              // no pair in these 16 offsets x 32 rows (wave-uniform)
#pragma unroll
            for (int c = 0; c < STEM_CMAX; ++c) {
                if (c < cin) {
                    stem_bf16x4 a1, a2, a3, b1, b2, b3;
                    tl_split4(make_float4(x[ob][c][0], x[ob][c][1], x[ob][c][2], x[ob][c][3]), a1, a2, a3);
                    tl_split4(make_float4(x[ob][c][4], x[ob][c][5], x[ob][c][6], x[ob][c][7]), b1, b2, b3);
                    const stem_bf16x8 A1 = __builtin_shufflevector(a1, b1, 0, 1, 2, 3, 4, 5, 6, 7);
                    const stem_bf16x8 A2 = __builtin_shufflevector(a2, b2, 0, 1, 2, 3, 4, 5, 6, 7);
                    const stem_bf16x8 A3 = __builtin_shufflevector(a3, b3, 0, 1, 2, 3, 4, 5, 6, 7);
#define STEM_WMFMA(AP, GP)                                                                                     \
    _Pragma("unroll") for (int nb = 0; nb < 2; ++nb)                                                           \
        acc[ob][c][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(AP, G[nb][GP], acc[ob][c][nb], 0, 0, 0);
                    STEM_WMFMA(A3, 0) STEM_WMFMA(A2, 1) STEM_WMFMA(A1, 2) STEM_WMFMA(A2, 0) STEM_WMFMA(A1, 1) STEM_WMFMA(A1, 0)
#undef STEM_WMFMA
                }
            }
        }
    }
    // ---- partial[blockIdx.x][k][c][n]: C row = 4 (lane >> 4) + r (offset within the block), column = lane & 15 (output channel)
    float* d = partial + int64_t(blockIdx.x) * K * cin * STEM_COUT;
#pragma unroll
    for (int ob = 0; ob < 2; ++ob)
#pragma unroll
        for (int c = 0; c < STEM_CMAX; ++c)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int k = 32 * wave + 16 * ob + 4 * lg + r;
                    if (k < K && c < cin) d[(k * cin + c) * STEM_COUT + 16 * nb + l15] = acc[ob][c][nb][r];
                }
}

// gW[e] = sum of the partials in a FIXED two-level order: 16 thread rows each add a contiguous sixteenth of the partials (ascending),
// thread row 0 adds the 16 sums (ascending).  (The one-level kernel below walks all 512 partials per thread with 47 workgroups: 20 us.)
__global__ __launch_bounds__(1024) void stem_wgrad_reduce2_kernel(const float* __restrict__ partial, int parts, int total,
                                                                  float* __restrict__ gW) {
    __shared__ float sums[16][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + tx;
    const int per = (parts + 15) / 16;
    const int p0 = ty * per, p1 = min(parts, p0 + per);
    float s = 0.f;
    if (e < total) {
        for (int p = p0; p < p1; p += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = partial[int64_t(p + u < p1 ? p + u : p0) * total + e];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += p + u < p1 ? v[u] : 0.f;
        }
    }
    sums[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && e < total) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) t += sums[q][tx];
        gW[e] = t;
    }
}

// gW[e] = partial[0][e] + partial[1][e] + ...   (fixed order)
__global__ void stem_wgrad_reduce_kernel(const float* __restrict__ partial, int parts, int total, float* __restrict__ gW) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    // 16 loads in flight, then their adds in order (a loop of dependent load + add pairs is 128 memory round trips)
    float s = 0.f;
    for (int p0 = 0; p0 < parts; p0 += 16) {
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = partial[int64_t(p0 + u < parts ? p0 + u : p0) * total + e];
#pragma unroll
        for (int u = 0; u < 16; ++u) s += p0 + u < parts ? v[u] : 0.f;
    }
    gW[e] = s;
}

}  // namespace osn

using namespace osn;

extern "C" size_t osn_stem_conv_wgrad_ws_bytes(int K, int cin) {
    return size_t(SWM_PARTS > SW_PARTS ? SWM_PARTS : SW_PARTS) * size_t(K > 0 ? K : 1) * size_t(cin > 0 ? cin : 1) * STEM_COUT * 4;
}

extern "C" int osn_stem_conv_wgrad(const float* in, const float* gout, const int32_t* nbr, float* gW, int64_t n_out, int K,
                                   int cin, int cout, void* ws, size_t ws_bytes, osn_stream_t stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    OSN_REQUIRE(n_out >= 0 && n_out < (int64_t(1) << 31), OSN_E_ARG, "osn_stem_conv_wgrad: n_out out of range");
    OSN_REQUIRE(K >= 1 && K <= 125 && cin >= 1 && cin <= STEM_CMAX && cout == STEM_COUT, OSN_E_ARG,
                "osn_stem_conv_wgrad: needs K <= 125, cin <= %d, cout == %d (K=%d cin=%d cout=%d)", STEM_CMAX, STEM_COUT, K, cin, cout);
    OSN_REQUIRE(gW, OSN_E_ARG, "osn_stem_conv_wgrad: null gradient pointer");
    const int total = K * cin * STEM_COUT;
    if (n_out == 0) {
        OSN_HIP(hipMemsetAsync(gW, 0, size_t(total) * 4, st));
        return OSN_OK;
    }
    OSN_REQUIRE(in && gout && nbr && aligned16(gout), OSN_E_ARG, "osn_stem_conv_wgrad: null or unaligned pointer");
    OSN_REQUIRE(ws && ws_bytes >= osn_stem_conv_wgrad_ws_bytes(K, cin), OSN_E_WS, "osn_stem_conv_wgrad: workspace %zu < %zu",
                ws_bytes, osn_stem_conv_wgrad_ws_bytes(K, cin));
    float* partial = static_cast<float*>(ws);
    int parts;
    if (n_out >= STEM_MFMA_MIN_ROWS) {                 // the large maps: matrix cores (stem_mfma_wgrad_kernel), split-bf16
        const int64_t n_steps = cdiv(n_out, 32);
        parts = int(n_steps < SWM_PARTS ? n_steps : SWM_PARTS);
        hipLaunchKernelGGL(stem_mfma_wgrad_kernel, dim3(unsigned(parts)), dim3(256), 0, st, in, gout, nbr, partial, n_out, K, cin);
        hipLaunchKernelGGL(stem_wgrad_reduce2_kernel, dim3(unsigned(cdiv(total, 64))), dim3(1024), 0, st, partial, parts, total, gW);
        OSN_LAUNCH_CHECK();
        return OSN_OK;
    } else {
        const int64_t n_chunks = cdiv(n_out, SW_ROWS);
        parts = int(n_chunks < SW_PARTS ? n_chunks : SW_PARTS);
        hipLaunchKernelGGL(stem_wgrad_kernel, dim3(unsigned(parts), unsigned(cdiv(K, SW_KB))), dim3(256), 0, st, in, gout, nbr, partial,
                           n_out, K, cin);
    }
    hipLaunchKernelGGL(stem_wgrad_reduce_kernel, dim3(unsigned(cdiv(total, 256))), dim3(256), 0, st, partial, parts, total, gW);
    OSN_LAUNCH_CHECK();
    return OSN_OK;
}

int osn::stem_conv_fwd_epi(const float* in, const float* W, const int32_t* nbr, float* out, int64_t n_out, int K, int cin, int cout,
                           const Epi& epi, osn_stream_t stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    OSN_REQUIRE(n_out >= 0 && n_out < (int64_t(1) << 31), OSN_E_ARG, "osn_stem_conv_fwd: n_out out of range");
    OSN_REQUIRE(K >= 1 && K <= 125 && cin >= 1 && cin <= STEM_CMAX && cout == STEM_COUT, OSN_E_ARG,
                "osn_stem_conv_fwd: needs K <= 125, cin <= %d, cout == %d (K=%d cin=%d cout=%d)", STEM_CMAX, STEM_COUT, K, cin, cout);
    if (n_out == 0) return OSN_OK;
    OSN_REQUIRE(in && W && nbr && out && aligned16(out), OSN_E_ARG, "osn_stem_conv_fwd: null or unaligned pointer");
    // four lanes per row from 4096 rows on (below that the single-lane kernel's launch is all there is); from STEM_MFMA_MIN_ROWS on
    // the matrix-core kernel (split-bf16, fp32-class instead of exact fp32 products)
    if (n_out >= STEM_MFMA_MIN_ROWS)
        hipLaunchKernelGGL(epi.mean ? stem_mfma_fwd_kernel<true> : stem_mfma_fwd_kernel<false>, dim3(unsigned(cdiv(n_out, 64))), dim3(256), 0, st, in, W, nbr,
                           out, n_out, K, cin, epi);
    else if (n_out >= 4096)
        hipLaunchKernelGGL(epi.mean ? stem_fwd4_kernel<true> : stem_fwd4_kernel<false>, dim3(unsigned(cdiv(n_out, 64))), dim3(256), 0, st, in, W, nbr, out,
                           n_out, K, cin, epi);
    else
        hipLaunchKernelGGL(epi.mean ? stem_fwd_kernel<true> : stem_fwd_kernel<false>, dim3(unsigned(cdiv(n_out, 256))), dim3(256), 0, st, in, W, nbr, out,
                           n_out, K, cin, epi);
    OSN_LAUNCH_CHECK();
    return OSN_OK;
}

extern "C" int osn_stem_conv_fwd(const float* in, const float* W, const int32_t* nbr, float* out, int64_t n_out, int K,
                                 int cin, int cout, osn_stream_t stream) {
    return stem_conv_fwd_epi(in, W, nbr, out, n_out, K, cin, cout, epi_none(), stream);
}
