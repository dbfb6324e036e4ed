// The 3-channel stem convolution of the U-Net (conv0p1s1: 5^3 kernel, 3 -> 32 channels,
// models/mink_unet.py:47-50) on gfx950: plain fp32 VALU kernels.
//
// With 3 input channels there is no contraction worth a matrix unit (96 FMAs per pair); what the op does is
// stream the 125 x N neighbour table once (50 MB on S100k) and gather 12 bytes per pair.  The generic MFMA
// kernels spent 194 us (forward) and 253 us (weight gradient) on it, almost all of it tile bookkeeping.
//   forward:  one thread per output row, 32 accumulators, W[k] broadcast from LDS, offsets in ascending order
//             (a plain fmaf chain: bitwise reproducible, exact fp32 products);
//   weight gradient: one workgroup per (offset, row chunk), per-thread partial sums over its rows, a fixed
//             LDS tree, then a second launch sums the chunks in order.
#include "common.h"

namespace osn {

constexpr int STEM_CMAX = 4;       // input channels at most
constexpr int STEM_COUT = 32;      // output channels handled per thread
constexpr int STEM_CHUNKS = 16;    // row chunks per offset in the weight gradient

// out[o][0..31] = sum_k in[nbr[k][o]][0..cin) @ W[k]      (cout == 32)
__global__ __launch_bounds__(256) void stem_fwd_kernel(const float* __restrict__ in, const float* __restrict__ W,
                                                       const int32_t* __restrict__ nbr, float* __restrict__ out,
                                                       int64_t n_out, int K, int cin) {
    extern __shared__ float Ws[];                  // [K][cin][32]
    for (int i = threadIdx.x; i < K * cin * STEM_COUT; i += 256) Ws[i] = W[i];
    __syncthreads();
    const int64_t o = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (o >= n_out) return;
    float acc[STEM_COUT];
#pragma unroll
    for (int c = 0; c < STEM_COUT; ++c) acc[c] = 0.f;
    for (int k = 0; k < K; ++k) {
        const int i = nbr[int64_t(k) * n_out + o];
        if (i >= 0) {
            const float* w = Ws + k * cin * STEM_COUT;
            for (int ci = 0; ci < cin; ++ci) {
                const float x = in[int64_t(i) * cin + ci];
#pragma unroll
                for (int c = 0; c < STEM_COUT; ++c) acc[c] = fmaf(x, w[ci * STEM_COUT + c], acc[c]);
            }
        }
    }
    float4* dst = reinterpret_cast<float4*>(out + o * STEM_COUT);
#pragma unroll
    for (int c = 0; c < STEM_COUT / 4; ++c) dst[c] = make_float4(acc[4 * c], acc[4 * c + 1], acc[4 * c + 2], acc[4 * c + 3]);
}

// partial[k][chunk][ci][co] = sum over the rows o of the chunk with nbr[k][o] >= 0 of in[nbr[k][o]][ci] * gout[o][co]
// thread = (row lane r = tid / 32, output channel co = tid % 32): 8 rows in flight per workgroup, coalesced gout rows
__global__ __launch_bounds__(256) void stem_wgrad_kernel(const float* __restrict__ in, const float* __restrict__ gout,
                                                         const int32_t* __restrict__ nbr, float* __restrict__ partial,
                                                         int64_t n_out, int cin) {
    __shared__ float red[8][STEM_CMAX][STEM_COUT];
    const int k = blockIdx.x, chunk = blockIdx.y;
    const int co = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int64_t r0 = n_out * chunk / STEM_CHUNKS, r1 = n_out * (chunk + 1) / STEM_CHUNKS;
    float acc[STEM_CMAX] = {0.f, 0.f, 0.f, 0.f};
    for (int64_t o = r0 + rl; o < r1; o += 8) {
        const int i = nbr[int64_t(k) * n_out + o];
        if (i >= 0) {
            const float g = gout[o * STEM_COUT + co];
#pragma unroll
            for (int ci = 0; ci < STEM_CMAX; ++ci)
                if (ci < cin) acc[ci] = fmaf(in[int64_t(i) * cin + ci], g, acc[ci]);
        }
    }
#pragma unroll
    for (int ci = 0; ci < STEM_CMAX; ++ci) red[rl][ci][co] = acc[ci];
    __syncthreads();
    if (rl == 0) {
        for (int ci = 0; ci < cin; ++ci) {
            float s = 0.f;
#pragma unroll
            for (int r = 0; r < 8; ++r) s += red[r][ci][co];
            partial[((int64_t(k) * STEM_CHUNKS + chunk) * cin + ci) * STEM_COUT + co] = s;
        }
    }
}

__global__ void stem_wgrad_reduce_kernel(const float* __restrict__ partial, int K, int cin, float* __restrict__ gW) {
    const int per = cin * STEM_COUT;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= K * per) return;
    const int k = e / per, r = e - k * per;
    float s = 0.f;
    for (int c = 0; c < STEM_CHUNKS; ++c) s += partial[(int64_t(k) * STEM_CHUNKS + c) * per + r];
    gW[e] = s;
}

}  // namespace osn

using namespace osn;

extern "C" int osn_stem_conv_fwd(const float* in, const float* W, const int32_t* nbr, float* out, int64_t n_out, int K,
                                 int cin, int cout, osn_stream_t stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    OSN_REQUIRE(n_out >= 0 && n_out < (int64_t(1) << 31), OSN_E_ARG, "osn_stem_conv_fwd: n_out out of range");
    OSN_REQUIRE(K >= 1 && K <= 125 && cin >= 1 && cin <= STEM_CMAX && cout == STEM_COUT, OSN_E_ARG,
                "osn_stem_conv_fwd: needs K <= 125, cin <= %d, cout == %d (K=%d cin=%d cout=%d)", STEM_CMAX, STEM_COUT, K, cin, cout);
    if (n_out == 0) return OSN_OK;
    OSN_REQUIRE(in && W && nbr && out && aligned16(out), OSN_E_ARG, "osn_stem_conv_fwd: null or unaligned pointer");
    const size_t lds = size_t(K) * cin * STEM_COUT * 4;
    hipLaunchKernelGGL(stem_fwd_kernel, dim3(unsigned(cdiv(n_out, 256))), dim3(256), lds, st, in, W, nbr, out, n_out, K, cin);
    OSN_LAUNCH_CHECK();
    return OSN_OK;
}

extern "C" size_t osn_stem_conv_wgrad_ws_bytes(int K, int cin) { return size_t(K) * STEM_CHUNKS * size_t(cin) * STEM_COUT * 4; }

extern "C" int osn_stem_conv_wgrad(const float* in, const float* gout, const int32_t* nbr, float* gW, int64_t n_out, int K,
                                   int cin, int cout, void* ws, size_t ws_bytes, osn_stream_t stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    OSN_REQUIRE(n_out >= 0 && n_out < (int64_t(1) << 31), OSN_E_ARG, "osn_stem_conv_wgrad: n_out out of range");
    OSN_REQUIRE(K >= 1 && K <= 125 && cin >= 1 && cin <= STEM_CMAX && cout == STEM_COUT && gW, OSN_E_ARG,
                "osn_stem_conv_wgrad: needs K <= 125, cin <= %d, cout == %d", STEM_CMAX, STEM_COUT);
    if (n_out == 0) {
        OSN_HIP(hipMemsetAsync(gW, 0, size_t(K) * cin * cout * 4, st));
        return OSN_OK;
    }
    OSN_REQUIRE(in && gout && nbr, OSN_E_ARG, "osn_stem_conv_wgrad: null pointer");
    const size_t need = osn_stem_conv_wgrad_ws_bytes(K, cin);
    OSN_REQUIRE(ws && ws_bytes >= need, OSN_E_WS, "osn_stem_conv_wgrad: workspace %zu < %zu", ws_bytes, need);
    float* partial = static_cast<float*>(ws);
    hipLaunchKernelGGL(stem_wgrad_kernel, dim3(K, STEM_CHUNKS), dim3(256), 0, st, in, gout, nbr, partial, n_out, cin);
    hipLaunchKernelGGL(stem_wgrad_reduce_kernel, dim3(unsigned(cdiv(int64_t(K) * cin * STEM_COUT, 256))), dim3(256), 0, st,
                       partial, K, cin, gW);
    OSN_LAUNCH_CHECK();
    return OSN_OK;
}
