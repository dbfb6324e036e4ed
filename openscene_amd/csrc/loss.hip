// Distillation loss on the supervised rows, forward and backward in two streaming passes.
//
// Replaces run/distill.py:322-328:
//     output_3d = output_3d[mask]
//     loss = (1 - torch.nn.CosineSimilarity()(output_3d, feat_3d)).mean()          # loss_type == 'cosine'
//     loss = torch.nn.L1Loss()(output_3d, feat_3d)                                 # loss_type == 'l1'
// and the autograd chain behind it (index -> normalise -> multiply -> sum -> mean, and back through an index_add into a
// zero-filled [N, D] tensor): ~25 launches that stream the [N, 768] output gradient several times (0.35 ms of an 11 ms
// step on S100k).  Here the forward pass reads the n_sel selected rows once (one wave per row: dot product and the two
// squared norms), and the backward pass writes the FULL output gradient [N, D] once -- the closed-form row gradient on the
// selected rows, zeros elsewhere -- from an inverse index built in the forward pass.  HBM-bound streaming work.
#include "common.h"

namespace osn {

constexpr float LOSS_EPS = 1e-8f;     // torch.nn.CosineSimilarity's eps: each norm is clamped from below

__device__ inline float wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

// pos[sel[j]] = j  (pos pre-filled with -1)
__global__ void loss_inverse_kernel(const int64_t* __restrict__ sel, int64_t n_sel, int64_t n, int32_t* __restrict__ pos,
                                    int32_t* __restrict__ err) {
    const int64_t j = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (j >= n_sel) return;
    const int64_t r = sel[j];
    if (r < 0 || r >= n) { atomicOr(err, 1); return; }
    if (atomicCAS(&pos[r], -1, int32_t(j)) != -1) atomicOr(err, 2);        // a row selected twice
}

// one wave per selected row: stats[j] = (dot, |a|^2, |b|^2) and val[j] = 1 - cos  (KIND 0)   /   sum |a - b|  (KIND 1)
template <int KIND>
__global__ __launch_bounds__(256) void loss_rows_kernel(const float* __restrict__ out, const int64_t* __restrict__ sel,
                                                        const float* __restrict__ target, int64_t n_sel, int64_t n, int d,
                                                        float* __restrict__ stats, float* __restrict__ val) {
    const int lane = threadIdx.x & 63;
    const int64_t j = int64_t(blockIdx.x) * 4 + (threadIdx.x >> 6);
    if (j >= n_sel) return;
    int64_t r = sel[j];
    if (r < 0 || r >= n) r = 0;                                             // reported by loss_inverse_kernel
    const float* a = out + r * d;
    const float* b = target + j * d;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    for (int c = 4 * lane; c < d; c += 256) {
        const float4 x = *reinterpret_cast<const float4*>(a + c);
        const float4 y = *reinterpret_cast<const float4*>(b + c);
        if (KIND == 0) {
            s0 += x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w;
            s1 += x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w;
            s2 += y.x * y.x + y.y * y.y + y.z * y.z + y.w * y.w;
        } else {
            s0 += fabsf(x.x - y.x) + fabsf(x.y - y.y) + fabsf(x.z - y.z) + fabsf(x.w - y.w);
        }
    }
    s0 = wave_sum(s0);
    if (KIND == 0) {
        s1 = wave_sum(s1);
        s2 = wave_sum(s2);
        if (lane == 0) {
            const float na = fmaxf(sqrtf(s1), LOSS_EPS), nb = fmaxf(sqrtf(s2), LOSS_EPS);
            stats[3 * j + 0] = s0; stats[3 * j + 1] = s1; stats[3 * j + 2] = s2;
            val[j] = 1.f - s0 / (na * nb);
        }
    } else if (lane == 0) {
        val[j] = s0;
    }
}

// loss = sum(val) * scale: ONE workgroup, fixed order, fp64 accumulation (n_sel floats: a few tens of KB)
__global__ __launch_bounds__(1024) void loss_mean_kernel(const float* __restrict__ val, int64_t n_sel, double scale,
                                                         float* __restrict__ loss) {
    __shared__ double red[1024];
    double s = 0;
    for (int64_t j = threadIdx.x; j < n_sel; j += 1024) s += double(val[j]);
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 512; w >= 1; w >>= 1) {
        if (int(threadIdx.x) < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) *loss = float(red[0] * scale);
}

// one wave per OUTPUT row: gout[r] = g * d loss / d out[r]  (zeros for the rows the loss does not see)
//   cosine: -(1/n_sel) (b / (|a||b|) - cos a / |a|^2)          l1: sign(a - b) / (n_sel d)
template <int KIND>
__global__ __launch_bounds__(256) void loss_grad_kernel(const float* __restrict__ out, const float* __restrict__ target,
                                                        const int32_t* __restrict__ pos, const float* __restrict__ stats,
                                                        const float* __restrict__ gloss, int64_t n, int64_t n_sel, int d,
                                                        float* __restrict__ gout, float* __restrict__ grows) {
    const int lane = threadIdx.x & 63;
    const int64_t r = int64_t(blockIdx.x) * 4 + (threadIdx.x >> 6);
    if (r >= n) return;
    float* g = gout + r * d;
    const int j = pos[r];
    if (j < 0) {
        for (int c = 4 * lane; c < d; c += 256) *reinterpret_cast<float4*>(g + c) = make_float4(0.f, 0.f, 0.f, 0.f);
        return;
    }
    const float* a = out + r * d;
    const float* b = target + int64_t(j) * d;
    float* g2 = grows ? grows + int64_t(j) * d : nullptr;      // the same row once more, compacted: [n_sel, d]
    const float up = gloss ? *gloss : 1.f;
    if (KIND == 0) {
        const float dot = stats[3 * j + 0], a2 = stats[3 * j + 1], b2 = stats[3 * j + 2];
        const float na = fmaxf(sqrtf(a2), LOSS_EPS), nb = fmaxf(sqrtf(b2), LOSS_EPS);
        // cos = dot / (na nb) with clamped norms (torch normalises each operand by its clamped norm):
        // d cos / d a = b / (na nb) - dot a / (na^3 nb)   (the clamp is inactive for any row of a trained network)
        const float s = -up / float(n_sel);
        const float kb = s / (na * nb);
        const float ka = sqrtf(a2) > LOSS_EPS ? -s * dot / (na * na * na * nb) : 0.f;
        for (int c = 4 * lane; c < d; c += 256) {
            const float4 x = *reinterpret_cast<const float4*>(a + c);
            const float4 y = *reinterpret_cast<const float4*>(b + c);
            const float4 v = make_float4(kb * y.x + ka * x.x, kb * y.y + ka * x.y, kb * y.z + ka * x.z, kb * y.w + ka * x.w);
            *reinterpret_cast<float4*>(g + c) = v;
            if (g2) *reinterpret_cast<float4*>(g2 + c) = v;
        }
    } else {
        const float s = up / (float(n_sel) * float(d));
        auto sg = [s](float v) { return v > 0.f ? s : (v < 0.f ? -s : 0.f); };
        for (int c = 4 * lane; c < d; c += 256) {
            const float4 x = *reinterpret_cast<const float4*>(a + c);
            const float4 y = *reinterpret_cast<const float4*>(b + c);
            const float4 v = make_float4(sg(x.x - y.x), sg(x.y - y.y), sg(x.z - y.z), sg(x.w - y.w));
            *reinterpret_cast<float4*>(g + c) = v;
            if (g2) *reinterpret_cast<float4*>(g2 + c) = v;
        }
    }
}

}  // namespace osn

using namespace osn;

// state kept between the passes: pos int32 [n] | stats float [3 n_sel] | val float [n_sel] | err int32
extern "C" size_t osn_distill_loss_state_bytes(int64_t n, int64_t n_sel) {
    return align_up(size_t(n > 0 ? n : 1) * 4, 256) + align_up(size_t(n_sel > 0 ? n_sel : 1) * 12, 256) +
           align_up(size_t(n_sel > 0 ? n_sel : 1) * 4, 256) + 256;
}

namespace {
struct LossState {
    int32_t* pos;
    float *stats, *val;
    int32_t* err;
};
LossState loss_state(void* p, int64_t n, int64_t n_sel) {
    char* c = static_cast<char*>(p);
    LossState s;
    s.pos = reinterpret_cast<int32_t*>(c);
    c += align_up(size_t(n > 0 ? n : 1) * 4, 256);
    s.stats = reinterpret_cast<float*>(c);
    c += align_up(size_t(n_sel > 0 ? n_sel : 1) * 12, 256);
    s.val = reinterpret_cast<float*>(c);
    c += align_up(size_t(n_sel > 0 ? n_sel : 1) * 4, 256);
    s.err = reinterpret_cast<int32_t*>(c);
    return s;
}
}  // namespace

extern "C" int osn_distill_loss_fwd(const float* out, const int64_t* sel, const float* target, int64_t n, int64_t n_sel, int d,
                                    int kind, float* loss, void* state, size_t state_bytes, osn_stream_t stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    OSN_REQUIRE(n >= 0 && n_sel >= 1 && n_sel <= n && n_sel < (int64_t(1) << 31) && d >= 4 && (d & 3) == 0 && (kind == 0 || kind == 1),
                OSN_E_ARG, "osn_distill_loss_fwd: n=%lld n_sel=%lld d=%d kind=%d (needs 1 <= n_sel <= n, d %% 4 == 0, kind 0 | 1)",
                (long long)n, (long long)n_sel, d, kind);
    OSN_REQUIRE(out && sel && target && loss && state && state_bytes >= osn_distill_loss_state_bytes(n, n_sel), OSN_E_ARG,
                "osn_distill_loss_fwd: null pointer or state buffer too small");
    OSN_REQUIRE(aligned16(out) && aligned16(target), OSN_E_ARG, "osn_distill_loss_fwd: out and target must be 16-byte aligned");
    LossState s = loss_state(state, n, n_sel);
    OSN_HIP(hipMemsetAsync(s.pos, 0xFF, size_t(n) * 4, st));
    OSN_HIP(hipMemsetAsync(s.err, 0, 4, st));
    hipLaunchKernelGGL(loss_inverse_kernel, dim3(unsigned(cdiv(n_sel, 256))), dim3(256), 0, st, sel, n_sel, n, s.pos, s.err);
    const dim3 grid(unsigned(cdiv(n_sel, 4)));
    if (kind == 0) hipLaunchKernelGGL(loss_rows_kernel<0>, grid, dim3(256), 0, st, out, sel, target, n_sel, n, d, s.stats, s.val);
    else hipLaunchKernelGGL(loss_rows_kernel<1>, grid, dim3(256), 0, st, out, sel, target, n_sel, n, d, s.stats, s.val);
    const double scale = kind == 0 ? 1.0 / double(n_sel) : 1.0 / (double(n_sel) * double(d));
    hipLaunchKernelGGL(loss_mean_kernel, dim3(1), dim3(1024), 0, st, s.val, n_sel, scale, loss);
    OSN_LAUNCH_CHECK();
    return OSN_OK;
}

extern "C" int osn_distill_loss_bwd_rows(const float* out, const float* target, const float* gloss, int64_t n, int64_t n_sel, int d,
                                         int kind, float* gout, float* grows, const void* state, size_t state_bytes,
                                         osn_stream_t stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    OSN_REQUIRE(n >= 1 && n_sel >= 1 && n_sel <= n && d >= 4 && (d & 3) == 0 && (kind == 0 || kind == 1), OSN_E_ARG,
                "osn_distill_loss_bwd: n=%lld n_sel=%lld d=%d kind=%d", (long long)n, (long long)n_sel, d, kind);
    OSN_REQUIRE(out && target && gout && state && state_bytes >= osn_distill_loss_state_bytes(n, n_sel), OSN_E_ARG,
                "osn_distill_loss_bwd: null pointer or state buffer too small");
    OSN_REQUIRE(aligned16(out) && aligned16(target) && aligned16(gout) && (!grows || aligned16(grows)), OSN_E_ARG,
                "osn_distill_loss_bwd: pointers must be 16-byte aligned");
    LossState s = loss_state(const_cast<void*>(state), n, n_sel);
    const dim3 grid(unsigned(cdiv(n, 4)));
    if (kind == 0) hipLaunchKernelGGL(loss_grad_kernel<0>, grid, dim3(256), 0, st, out, target, s.pos, s.stats, gloss, n, n_sel, d, gout, grows);
    else hipLaunchKernelGGL(loss_grad_kernel<1>, grid, dim3(256), 0, st, out, target, s.pos, s.stats, gloss, n, n_sel, d, gout, grows);
    OSN_LAUNCH_CHECK();
    return OSN_OK;
}

extern "C" int osn_distill_loss_bwd(const float* out, const float* target, const float* gloss, int64_t n, int64_t n_sel, int d,
                                    int kind, float* gout, const void* state, size_t state_bytes, osn_stream_t stream) {
    return osn_distill_loss_bwd_rows(out, target, gloss, n, n_sel, d, kind, gout, nullptr, state, state_bytes, stream);
}

// 0 = fine; bit 0: an index outside [0, n); bit 1: a row selected twice  (blocks until the forward pass has run)
extern "C" int osn_distill_loss_check(const void* state, int64_t n, int64_t n_sel, osn_stream_t stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    OSN_REQUIRE(state, OSN_E_ARG, "osn_distill_loss_check: null state");
    LossState s = loss_state(const_cast<void*>(state), n, n_sel);
    int32_t host = 0;
    OSN_HIP(hipMemcpyAsync(&host, s.err, 4, hipMemcpyDeviceToHost, st));
    OSN_HIP(hipStreamSynchronize(st));
    OSN_REQUIRE(host == 0, OSN_E_ARG, "osn_distill_loss: %s", (host & 1) ? "a selected row index is outside the output" : "a row is selected twice");
    return OSN_OK;
}
