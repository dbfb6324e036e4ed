// Evaluation-mode batch norm (+ residual) (+ ReLU) (+ second store) in the epilogue of the kernel that produces a convolution's final
// rows (round 6; the network executor's inference pass, net.hip).
//
// models/mink_unet.py:126-172 / models/resnet_base.py:92-118 follow every convolution but the last with a batch norm; in evaluation mode
// that is a per-column affine map of the finished row -- a second launch that reads the convolution's output and writes it again
// (48 launches, 0.40 of 4.0 ms of kernel time of a MinkUNet18A inference pass on 101 k voxels: profiles/r06_s7_*).  Every kernel
// below holds the finished fp32 value in a register right before its store, so it applies
//     y = relu(bn_val(x, mean, 1 / sqrt(var + eps), gamma, beta) + residual)
// there, with the SAME expression in the same order as bn_apply_kernel (bn.hip): the result is bitwise the two-launch path's (and
// therefore the module-by-module path's, functional.py).  Training keeps the separate launches: batch statistics need every row first.
#pragma once
#include "common.h"

namespace osn {

// The normalised value, ONE definition for the forward kernels, the epilogues and the backward kernels that recompute the ReLU mask
// from x instead of reading y (round 4): explicit fma so that every side rounds identically whatever the surrounding code.
__device__ inline float bn_val(float x, float mu, float is, float ga, float be) { return __fmaf_rn((x - mu) * is, ga, be); }
__device__ inline float bn_is(float var, float eps) { return 1.f / sqrtf(var + eps); }

struct Epi {
    const float* mean;      // nullptr: no epilogue, the plain convolution result is stored
    const float* var;
    const float* gamma;
    const float* beta;
    const float* res;       // residual rows [n_out, cout] in the OUTPUT's row order, or nullptr
    float* y2;              // second store (ME.cat written in place): rows of pitch ld2, first column already applied; or nullptr
    int64_t ld2;
    float eps;
    int relu;
};

inline Epi epi_none() { return Epi{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0.f, 0}; }

// The per-column constants of four consecutive columns (a thread that stores several rows of the same columns loads them and takes the
// four reciprocal square roots ONCE: bn_is is a pure function of (var, eps), so where it is evaluated does not change a bit)
struct EpiCols {
    float4 mu, is, ga, be;
};
__device__ inline EpiCols epi_cols(const Epi& e, int col) {
    EpiCols c;
    c.mu = *reinterpret_cast<const float4*>(e.mean + col);
    const float4 vv = *reinterpret_cast<const float4*>(e.var + col);
    c.is = make_float4(bn_is(vv.x, e.eps), bn_is(vv.y, e.eps), bn_is(vv.z, e.eps), bn_is(vv.w, e.eps));
    c.ga = *reinterpret_cast<const float4*>(e.gamma + col);
    c.be = *reinterpret_cast<const float4*>(e.beta + col);
    return c;
}

// columns col .. col + 3 of output row `row` (row pitch cout for the residual): normalise, add, clamp, second store; returns the quad to store
__device__ inline float4 epi_apply(const Epi& e, const EpiCols& c, float4 x, int64_t row, int col, int cout) {
    float4 o;
    o.x = bn_val(x.x, c.mu.x, c.is.x, c.ga.x, c.be.x);
    o.y = bn_val(x.y, c.mu.y, c.is.y, c.ga.y, c.be.y);
    o.z = bn_val(x.z, c.mu.z, c.is.z, c.ga.z, c.be.z);
    o.w = bn_val(x.w, c.mu.w, c.is.w, c.ga.w, c.be.w);
    if (e.res) {
        const float4 rv = *reinterpret_cast<const float4*>(e.res + row * cout + col);
        o.x += rv.x; o.y += rv.y; o.z += rv.z; o.w += rv.w;
    }
    if (e.relu) {
        o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
    }
    if (e.y2) *reinterpret_cast<float4*>(e.y2 + row * e.ld2 + col) = o;
    return o;
}
__device__ inline float4 epi_quad(const Epi& e, float4 x, int64_t row, int col, int cout) {
    return epi_apply(e, epi_cols(e, col), x, row, col, cout);
}

// ---- the library's convolution launches with an epilogue (net.hip); the extern "C" entry points pass epi_none()
int spconv_fwd_tl_epi(const float* in, int64_t n_in, const void* Wp, const void* tl, const int32_t* out_rows, float* out, int64_t n_out,
                      int K, int cin, int cout, int bm, void* ws, size_t ws_bytes, int32_t* counters, const Epi& epi, osn_stream_t stream);
int spconv_fwd_ws_epi(const float* in, int64_t n_in, const void* Wp, const void* pl, int64_t pl_rows, int swap, int direct,
                      const int32_t* nbr_dst, float* out, int64_t n_dst, int K, int cin, int cout, void* ws, size_t ws_bytes, const Epi& epi,
                      osn_stream_t stream);
int spconv_fwd_rg_epi(const float* in, int64_t n_in, const void* Wp, const int32_t* nbr, const int32_t* out_rows, float* out,
                      int64_t n_out, int K, int cin, int cout, const Epi& epi, osn_stream_t stream);
int dense_fwd_epi(const float* in, const void* Wp, float* out, int64_t n, int cin, int cout, const Epi& epi, osn_stream_t stream);
int stem_conv_fwd_epi(const float* in, const float* W, const int32_t* nbr, float* out, int64_t n_out, int K, int cin, int cout,
                      const Epi& epi, osn_stream_t stream);

}  // namespace osn
