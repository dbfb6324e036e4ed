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

namespace osn {

constexpr int STEM_CMAX = 4;       // input channels at most
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
    return size_t(SW_PARTS) * size_t(K > 0 ? K : 1) * size_t(cin > 0 ? cin : 1) * STEM_COUT * 4;
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
    const int64_t n_chunks = cdiv(n_out, SW_ROWS);
    const int parts = int(n_chunks < SW_PARTS ? n_chunks : SW_PARTS);
    float* partial = static_cast<float*>(ws);
    hipLaunchKernelGGL(stem_wgrad_kernel, dim3(unsigned(parts), unsigned(cdiv(K, SW_KB))), dim3(256), 0, st, in, gout, nbr, partial,
                       n_out, K, cin);
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
    // four lanes per row from 4096 rows on (below that the single-lane kernel's launch is all there is)
    if (n_out >= 4096)
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
