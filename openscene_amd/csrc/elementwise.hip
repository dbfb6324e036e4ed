// Row-aligned elementwise pieces of the U-Net on gfx950: stand-alone ReLU
// (ME.MinkowskiReLU, models/mink_unet.py:114), the un-fused residual `out += residual` of a BasicBlock, and
// ME.cat of two tensors on one coordinate map (models/mink_unet.py:147,155,163,171) with its backward split.
// Pure streaming work: 16-byte lanes, grid-stride loops, one launch each (cat / split: ONE launch for both halves).
#include "common.h"

namespace osn {

__device__ inline float4 ew_ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ inline void ew_st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ inline float relu_nan(float x) { return x > 0.f ? x : (x != x ? x : 0.f); }

// MODE 0: y = x > 0 ? x : (x != x ? x : 0)  (NaN propagates, as torch.relu does)      MODE 1: gx = y > 0 ? gy : 0  (a = y, b = gy)      MODE 2: out = a + b
template <int MODE>
__global__ __launch_bounds__(256) void ew_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                 float* __restrict__ out, int64_t total) {
    const int64_t total4 = total >> 2;
    const int64_t stride = int64_t(gridDim.x) * blockDim.x;
    for (int64_t e = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; e < total4; e += stride) {
        const float4 x = ew_ld4(a + 4 * e);
        float4 o;
        if (MODE == 0) {
            o = make_float4(relu_nan(x.x), relu_nan(x.y), relu_nan(x.z), relu_nan(x.w));
        } else {
            const float4 y = ew_ld4(b + 4 * e);
            if (MODE == 1) o = make_float4(x.x > 0.f ? y.x : 0.f, x.y > 0.f ? y.y : 0.f, x.z > 0.f ? y.z : 0.f, x.w > 0.f ? y.w : 0.f);
            else o = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
        }
        ew_st4(out + 4 * e, o);
    }
    // tail (total not a multiple of 4): the first workgroup's first lanes
    if (blockIdx.x == 0 && threadIdx.x < (total & 3)) {
        const int64_t e = (total4 << 2) + threadIdx.x;
        const float x = a[e];
        out[e] = MODE == 0 ? relu_nan(x) : (MODE == 1 ? (x > 0.f ? b[e] : 0.f) : x + b[e]);
    }
}

// out[r, 0:ca] = a[r, :], out[r, ca:ca+cb] = b[r, :]  (SPLIT: the other way round, gout -> ga, gb)
// one thread per 4 output channels; ca % 4 == 0 and cb % 4 == 0
template <bool SPLIT>
__global__ __launch_bounds__(256) void cat2_kernel(float* __restrict__ a, int ca4, float* __restrict__ b, int cb4,
                                                   float* __restrict__ out, int64_t n) {
    const int c4 = ca4 + cb4;
    const int64_t total = n * c4;
    for (int64_t e = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; e < total; e += int64_t(gridDim.x) * blockDim.x) {
        const int64_t r = e / c4;
        const int c = int(e - r * c4);
        float* side = c < ca4 ? a + (r * ca4 + c) * 4 : b + (r * cb4 + (c - ca4)) * 4;
        if (SPLIT) ew_st4(side, ew_ld4(out + e * 4));
        else ew_st4(out + e * 4, ew_ld4(side));
    }
}

// out[j, :] = in[idx[j], :]
__global__ __launch_bounds__(256) void rows_gather_kernel(const float* __restrict__ in, const int64_t* __restrict__ idx, int64_t n_idx,
                                                          int c4, float* __restrict__ out) {
    const int64_t total = n_idx * c4;
    for (int64_t e = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; e < total; e += int64_t(gridDim.x) * blockDim.x) {
        const int64_t j = e / c4;
        const int c = int(e - j * c4);
        ew_st4(out + e * 4, ew_ld4(in + (idx[j] * c4 + c) * 4));
    }
}

// out[r, :] = pos[r] >= 0 ? in[pos[r], :] : 0
__global__ __launch_bounds__(256) void rows_scatter_zero_kernel(const float* __restrict__ in, const int32_t* __restrict__ pos, int64_t n,
                                                                int c4, float* __restrict__ out) {
    const int64_t total = n * c4;
    for (int64_t e = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; e < total; e += int64_t(gridDim.x) * blockDim.x) {
        const int64_t r = e / c4;
        const int c = int(e - r * c4);
        const int j = pos[r];
        ew_st4(out + e * 4, j >= 0 ? ew_ld4(in + (int64_t(j) * c4 + c) * 4) : make_float4(0.f, 0.f, 0.f, 0.f));
    }
}

static int ew_blocks(int64_t work) {
    int64_t g = cdiv(work, 256);
    if (g > 4096) g = 4096;
    if (g < 1) g = 1;
    return int(g);
}

}  // namespace osn

using namespace osn;

extern "C" int osn_relu_fwd(const float* x, float* y, int64_t total, osn_stream_t stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    OSN_REQUIRE(total >= 0, OSN_E_ARG, "osn_relu_fwd: negative size");
    if (total == 0) return OSN_OK;
    OSN_REQUIRE(x && y && aligned16(x) && aligned16(y), OSN_E_ARG, "osn_relu_fwd: null or unaligned pointer");
    hipLaunchKernelGGL((ew_kernel<0>), dim3(ew_blocks(total / 4 + 1)), dim3(256), 0, st, x, (const float*)nullptr, y, total);
    OSN_LAUNCH_CHECK();
    return OSN_OK;
}

extern "C" int osn_relu_bwd(const float* y, const float* gy, float* gx, int64_t total, osn_stream_t stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    OSN_REQUIRE(total >= 0, OSN_E_ARG, "osn_relu_bwd: negative size");
    if (total == 0) return OSN_OK;
    OSN_REQUIRE(y && gy && gx && aligned16(y) && aligned16(gy) && aligned16(gx), OSN_E_ARG, "osn_relu_bwd: null or unaligned pointer");
    hipLaunchKernelGGL((ew_kernel<1>), dim3(ew_blocks(total / 4 + 1)), dim3(256), 0, st, y, gy, gx, total);
    OSN_LAUNCH_CHECK();
    return OSN_OK;
}

extern "C" int osn_add(const float* a, const float* b, float* out, int64_t total, osn_stream_t stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    OSN_REQUIRE(total >= 0, OSN_E_ARG, "osn_add: negative size");
    if (total == 0) return OSN_OK;
    OSN_REQUIRE(a && b && out && aligned16(a) && aligned16(b) && aligned16(out), OSN_E_ARG, "osn_add: null or unaligned pointer");
    hipLaunchKernelGGL((ew_kernel<2>), dim3(ew_blocks(total / 4 + 1)), dim3(256), 0, st, a, b, out, total);
    OSN_LAUNCH_CHECK();
    return OSN_OK;
}

extern "C" int osn_cat2(const float* a, int ca, const float* b, int cb, float* out, int64_t n, osn_stream_t stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    OSN_REQUIRE(n >= 0 && ca >= 4 && cb >= 4 && (ca & 3) == 0 && (cb & 3) == 0, OSN_E_ARG,
                "osn_cat2: channel counts must be positive multiples of 4 (ca=%d cb=%d)", ca, cb);
    if (n == 0) return OSN_OK;
    OSN_REQUIRE(a && b && out && aligned16(a) && aligned16(b) && aligned16(out), OSN_E_ARG, "osn_cat2: null or unaligned pointer");
    hipLaunchKernelGGL((cat2_kernel<false>), dim3(ew_blocks(n * ((ca + cb) / 4))), dim3(256), 0, st, const_cast<float*>(a), ca / 4,
                       const_cast<float*>(b), cb / 4, out, n);
    OSN_LAUNCH_CHECK();
    return OSN_OK;
}

extern "C" int osn_cat2_bwd(const float* gout, float* ga, int ca, float* gb, int cb, int64_t n, osn_stream_t stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    OSN_REQUIRE(n >= 0 && ca >= 4 && cb >= 4 && (ca & 3) == 0 && (cb & 3) == 0, OSN_E_ARG,
                "osn_cat2_bwd: channel counts must be positive multiples of 4 (ca=%d cb=%d)", ca, cb);
    if (n == 0) return OSN_OK;
    OSN_REQUIRE(gout && ga && gb && aligned16(gout) && aligned16(ga) && aligned16(gb), OSN_E_ARG, "osn_cat2_bwd: null or unaligned pointer");
    hipLaunchKernelGGL((cat2_kernel<true>), dim3(ew_blocks(n * ((ca + cb) / 4))), dim3(256), 0, st, ga, ca / 4, gb, cb / 4,
                       const_cast<float*>(gout), n);
    OSN_LAUNCH_CHECK();
    return OSN_OK;
}

extern "C" int osn_rows_gather(const float* in, const int64_t* idx, int64_t n_idx, int c, float* out, osn_stream_t stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    OSN_REQUIRE(n_idx >= 0 && c >= 4 && (c & 3) == 0, OSN_E_ARG, "osn_rows_gather: needs c %% 4 == 0 (n_idx=%lld c=%d)", (long long)n_idx, c);
    if (n_idx == 0) return OSN_OK;
    OSN_REQUIRE(in && idx && out && aligned16(in) && aligned16(out), OSN_E_ARG, "osn_rows_gather: null or unaligned pointer");
    hipLaunchKernelGGL(rows_gather_kernel, dim3(ew_blocks(n_idx * (c / 4))), dim3(256), 0, st, in, idx, n_idx, c / 4, out);
    OSN_LAUNCH_CHECK();
    return OSN_OK;
}

extern "C" int osn_rows_scatter_zero(const float* in, const int32_t* pos, int64_t n, int c, float* out, osn_stream_t stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    OSN_REQUIRE(n >= 0 && c >= 4 && (c & 3) == 0, OSN_E_ARG, "osn_rows_scatter_zero: needs c %% 4 == 0 (n=%lld c=%d)", (long long)n, c);
    if (n == 0) return OSN_OK;
    OSN_REQUIRE(in && pos && out && aligned16(in) && aligned16(out), OSN_E_ARG, "osn_rows_scatter_zero: null or unaligned pointer");
    hipLaunchKernelGGL(rows_scatter_zero_kernel, dim3(ew_blocks(n * (c / 4))), dim3(256), 0, st, in, pos, n, c / 4, out);
    OSN_LAUNCH_CHECK();
    return OSN_OK;
}
