// Adam over ONE flat parameter buffer (torch.optim.Adam, run/distill.py:170-178 builds the reference's optimizer).
//
// The network executor hands every gradient out as a slice of one flat fp32 buffer; with the parameters and both moment
// estimates laid out the same way an optimizer step is a single streaming pass -- 4 arrays read, 3 written, 28 bytes per
// parameter (62 MB of weights: 0.43 GB, ~60 us at the HBM roof) -- instead of torch's fused multi-tensor Adam, which
// walks the 145 tensors in chunks (5 launches, 236 us per step measured).  Same update rule and operation order as
// torch's fused kernel (lerp form of the first moment, bias corrections folded into step size and denominator).
#include "common.h"

namespace osn {

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, int64_t n4, float lr, float beta1, float beta2,
                                                   float eps, float weight_decay, float bc1, float bc2_sqrt) {
    const float step_size = lr / bc1;
    const int64_t stride = int64_t(gridDim.x) * blockDim.x;
    for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 pv = reinterpret_cast<float4*>(p)[i];
        float4 gv = reinterpret_cast<const float4*>(g)[i];
        float4 mv = reinterpret_cast<float4*>(m)[i];
        float4 vv = reinterpret_cast<float4*>(v)[i];
        float* pp = &pv.x; float* gp = &gv.x; float* mp = &mv.x; float* vp = &vv.x;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float grad = gp[e];
            if (weight_decay != 0.f) grad += weight_decay * pp[e];
            mp[e] = mp[e] + (1.f - beta1) * (grad - mp[e]);                       // lerp(exp_avg, grad, 1 - beta1)
            vp[e] = beta2 * vp[e] + (1.f - beta2) * grad * grad;
            const float denom = sqrtf(vp[e]) / bc2_sqrt + eps;
            pp[e] -= step_size * mp[e] / denom;
        }
        reinterpret_cast<float4*>(p)[i] = pv;
        reinterpret_cast<float4*>(m)[i] = mv;
        reinterpret_cast<float4*>(v)[i] = vv;
    }
}

}  // namespace osn

using namespace osn;

extern "C" int osn_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, int64_t step,
                             float lr, float beta1, float beta2, float eps, float weight_decay, osn_stream_t stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    OSN_REQUIRE(n >= 0 && (n & 3) == 0 && step >= 1, OSN_E_ARG, "osn_adam_step: n=%lld must be a multiple of 4, step=%lld >= 1",
                (long long)n, (long long)step);
    if (n == 0) return OSN_OK;
    OSN_REQUIRE(params && grads && exp_avg && exp_avg_sq && aligned16(params) && aligned16(grads) && aligned16(exp_avg) && aligned16(exp_avg_sq),
                OSN_E_ARG, "osn_adam_step: null or unaligned pointer");
    // bias corrections in double, as torch computes them on the host side of its fused kernel
    const double bc1 = 1.0 - pow(double(beta1), double(step));
    const double bc2 = 1.0 - pow(double(beta2), double(step));
    int64_t g = cdiv(n / 4, 256);
    if (g > 8192) g = 8192;
    hipLaunchKernelGGL(adam_kernel, dim3(unsigned(g)), dim3(256), 0, st, params, grads, exp_avg, exp_avg_sq, n / 4, lr, beta1, beta2,
                       eps, weight_decay, float(bc1), float(sqrt(bc2)));
    OSN_LAUNCH_CHECK();
    return OSN_OK;
}
