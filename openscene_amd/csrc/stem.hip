// The 3-channel stem convolution of the U-Net (conv0p1s1: 5^3 kernel, 3 -> 32 channels,
// models/mink_unet.py:47-50) on gfx950: plain fp32 VALU kernels.
//
// With 3 input channels there is no contraction worth a matrix unit (96 FMAs per pair); what the op does is
// stream the 125 x N neighbour table once (50 MB on S100k) and gather 12 bytes per pair.  The generic MFMA
// kernel spent 191 us on it; this one 127 us (measured; a pair-driven variant would skip the 89 % empty
// table entries).  One thread per output row, 32 accumulators, offsets in ascending order: a plain fmaf
// chain, bitwise reproducible, exact fp32 products.  (A dedicated weight-gradient kernel was measured slower
// than the generic fp32-MFMA one -- 314 vs 257 us -- and is not kept.)
#include "common.h"

namespace osn {

constexpr int STEM_CMAX = 4;       // input channels at most
constexpr int STEM_COUT = 32;      // output channels handled per thread

// out[o][0..31] = sum_k in[nbr[k][o]][0..cin) @ W[k]      (cout == 32)
// One thread per output row; the offset loop is unrolled by 5 with the five table entries loaded up front (a
// 100 k-row table gives only 1.5 waves per SIMD, so the latency of the dependent table -> row -> FMA chain has to
// be covered by instruction-level parallelism); W[k] is indexed by wave-uniform values only, so it is fetched
// through the scalar cache (staging it in LDS per workgroup cost more than the whole convolution).
__global__ __launch_bounds__(256) void stem_fwd_kernel(const float* __restrict__ in, const float* __restrict__ W,
                                                       const int32_t* __restrict__ nbr, float* __restrict__ out,
                                                       int64_t n_out, int K, int cin) {
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
        for (int c = 0; c < STEM_COUT / 4; ++c) dst[c] = make_float4(acc[4 * c], acc[4 * c + 1], acc[4 * c + 2], acc[4 * c + 3]);
    }
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
    hipLaunchKernelGGL(stem_fwd_kernel, dim3(unsigned(cdiv(n_out, 256))), dim3(256), 0, st, in, W, nbr, out, n_out, K, cin);
    OSN_LAUNCH_CHECK();
    return OSN_OK;
}
