// Multi-view feature fusion on gfx950: the step BEFORE the hot path that produces the fused
// per-point features the distillation trains against.
//
//   osn_fusion_project      scripts/feature_fusion/fusion_util.py:93-139  (PointCloudToImageMapper.compute_mapping)
//   osn_fusion_accumulate   scripts/feature_fusion/scannet_openseg.py:93-106 (per view: counter += 1, sum += feat[:, y, x])
//   osn_fusion_finish       scripts/feature_fusion/scannet_openseg.py:108-109 (counter == 0 -> 1e-5, sum / counter)
//
// Pure streaming / gather work (HBM- and L2-bound), fp64 for the projection exactly as numpy computes it:
//   p = world_to_camera @ [x y z 1]^T     an FMA chain in dgemm's order (m0*x, then +m1*y, +m2*z, +m3*1 fused)
//   u = (p0 * fx) / p2 + cx               separate IEEE multiply, divide, add (numpy does not contract)
//   round half to even (np.round), pixel-boundary test on the rounded value, then
//   |depth[v, u] - p2| <= vis * depth[v, u]   (or p2 > 0 when there is no depth image).
#include "common.h"

namespace osn {

struct ProjectArgs {
    double m[12];          // rows 0..2 of world_to_camera (row-major 3 x 4)
    double fx, fy, cx, cy;
    double vis;
    int H, W, cut;
};

__global__ __launch_bounds__(256) void fusion_project_kernel(const double* __restrict__ coords, int64_t n, ProjectArgs a,
                                                             const double* __restrict__ depth,
                                                             int64_t* __restrict__ mapping) {
    const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double x = coords[3 * i + 0], y = coords[3 * i + 1], z = coords[3 * i + 2];
    double p[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        double s = __dmul_rn(a.m[4 * r + 0], x);
        s = fma(a.m[4 * r + 1], y, s);
        s = fma(a.m[4 * r + 2], z, s);
        p[r] = fma(a.m[4 * r + 3], 1.0, s);
    }
    const double u = __dadd_rn(__ddiv_rn(__dmul_rn(p[0], a.fx), p[2]), a.cx);
    const double v = __dadd_rn(__ddiv_rn(__dmul_rn(p[1], a.fy), p[2]), a.cy);
    const double ur = rint(u), vr = rint(v);                // round half to even; NaN / inf fail every comparison below
    bool inside = ur >= double(a.cut) && vr >= double(a.cut) && ur < double(a.W - a.cut) && vr < double(a.H - a.cut);
    int64_t pu = 0, pv = 0;
    if (inside) {
        pu = int64_t(ur);
        pv = int64_t(vr);
        if (depth) {
            const double d = depth[pv * a.W + pu];
            inside = fabs(__dsub_rn(d, p[2])) <= __dmul_rn(a.vis, d);
        } else {
            inside = p[2] > 0.0;
        }
    }
    mapping[3 * i + 0] = inside ? pv : 0;
    mapping[3 * i + 1] = inside ? pu : 0;
    mapping[3 * i + 2] = inside ? 1 : 0;
}

// sum[p][c] += feat[c][y_p][x_p] for the visible points of one view; counter[p] += 1.
// A workgroup = 64 points x 64 channels: the image planes are read with the lanes along the POINTS (a plane is
// H*W*4 = 307 KB and stays in L2; the alternative -- lanes along channels -- touches a different plane per lane) into
// an LDS tile that is then added to the point-major sums with the lanes along the CHANNELS (256-byte row segments).
constexpr int FA_P = 64, FA_C = 64;

__global__ __launch_bounds__(256) void fusion_accumulate_kernel(const float* __restrict__ feat, int D, int H, int W,
                                                                const int64_t* __restrict__ mapping, int64_t n,
                                                                float* __restrict__ sum, float* __restrict__ counter) {
    __shared__ float tile[FA_C][FA_P + 1];
    __shared__ int pix[FA_P];                        // y * W + x, or -1
    const int tid = threadIdx.x;
    const int64_t p0 = int64_t(blockIdx.x) * FA_P;
    const int c0 = blockIdx.y * FA_C;
    if (tid < FA_P) {
        const int64_t p = p0 + tid;
        int px = -1;
        if (p < n && mapping[3 * p + 2] != 0) {
            // a mapping computed for another image size must never read outside feat[D][H][W]: such a point is skipped
            // (the host wrapper refuses mismatched sizes up front; the reference raises IndexError)
            const int64_t y = mapping[3 * p + 0], x = mapping[3 * p + 1];
            if (y >= 0 && y < H && x >= 0 && x < W) px = int(y) * W + int(x);
        }
        pix[tid] = px;
        if (px >= 0 && blockIdx.y == 0) counter[p] += 1.f;
    }
    __syncthreads();
    const int64_t plane = int64_t(H) * W;
    {
        const int pl = tid & (FA_P - 1), cg = tid >> 6;      // point lane, channel group (4 x 16 channels)
        const int px = pix[pl];
#pragma unroll 4
        for (int j = 0; j < FA_C / 4; ++j) {
            const int c = cg * (FA_C / 4) + j;
            float v = 0.f;
            if (px >= 0 && c0 + c < D) v = feat[int64_t(c0 + c) * plane + px];
            tile[c][pl] = v;
        }
    }
    __syncthreads();
    {
        const int cl = tid & (FA_C - 1), pg = tid >> 6;      // channel lane, point group (4 x 16 points)
        if (c0 + cl < D) {
#pragma unroll 4
            for (int j = 0; j < FA_P / 4; ++j) {
                const int pl = pg * (FA_P / 4) + j;
                if (pix[pl] >= 0) sum[(p0 + pl) * D + c0 + cl] += tile[cl][pl];
            }
        }
    }
}

__global__ __launch_bounds__(256) void fusion_finish_kernel(const float* __restrict__ sum, const float* __restrict__ counter,
                                                            int64_t n, int D, float* __restrict__ bank) {
    const int64_t total = n * D;
    for (int64_t e = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; e < total; e += int64_t(gridDim.x) * blockDim.x) {
        float c = counter[e / D];
        if (c == 0.f) c = 1e-5f;
        bank[e] = sum[e] / c;
    }
}

}  // namespace osn

using namespace osn;

extern "C" int osn_fusion_project(const double* coords3, int64_t n, const double* world_to_camera16,
                                  const double* intrinsic4, const double* depth, int H, int W, int cut_bound,
                                  double vis_thres, int64_t* mapping, osn_stream_t stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    OSN_REQUIRE(n >= 0 && H >= 1 && W >= 1 && cut_bound >= 0, OSN_E_ARG, "osn_fusion_project: bad sizes");
    OSN_REQUIRE(world_to_camera16 && intrinsic4, OSN_E_ARG, "osn_fusion_project: null host matrices");
    if (n == 0) return OSN_OK;
    OSN_REQUIRE(coords3 && mapping, OSN_E_ARG, "osn_fusion_project: null pointer");
    ProjectArgs a;
    for (int i = 0; i < 12; ++i) a.m[i] = world_to_camera16[i];
    a.fx = intrinsic4[0]; a.fy = intrinsic4[1]; a.cx = intrinsic4[2]; a.cy = intrinsic4[3];
    a.vis = vis_thres;
    a.H = H; a.W = W; a.cut = cut_bound;
    hipLaunchKernelGGL(fusion_project_kernel, dim3(unsigned(cdiv(n, 256))), dim3(256), 0, st, coords3, n, a, depth, mapping);
    OSN_LAUNCH_CHECK();
    return OSN_OK;
}

extern "C" int osn_fusion_accumulate(const float* feat2d, int D, int H, int W, const int64_t* mapping, int64_t n,
                                     float* sum_features, float* counter, osn_stream_t stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    OSN_REQUIRE(n >= 0 && D >= 1 && H >= 1 && W >= 1 && int64_t(H) * W < (int64_t(1) << 31), OSN_E_ARG,
                "osn_fusion_accumulate: bad sizes");
    if (n == 0) return OSN_OK;
    OSN_REQUIRE(feat2d && mapping && sum_features && counter, OSN_E_ARG, "osn_fusion_accumulate: null pointer");
    hipLaunchKernelGGL(fusion_accumulate_kernel, dim3(unsigned(cdiv(n, FA_P)), unsigned(cdiv(D, FA_C))), dim3(256), 0, st,
                       feat2d, D, H, W, mapping, n, sum_features, counter);
    OSN_LAUNCH_CHECK();
    return OSN_OK;
}

extern "C" int osn_fusion_finish(const float* sum_features, const float* counter, int64_t n, int D, float* feat_bank,
                                 osn_stream_t stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    OSN_REQUIRE(n >= 0 && D >= 1, OSN_E_ARG, "osn_fusion_finish: bad sizes");
    if (n == 0) return OSN_OK;
    OSN_REQUIRE(sum_features && counter && feat_bank, OSN_E_ARG, "osn_fusion_finish: null pointer");
    int64_t g = cdiv(n * D, 256);
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(fusion_finish_kernel, dim3(unsigned(g)), dim3(256), 0, st, sum_features, counter, n, D, feat_bank);
    OSN_LAUNCH_CHECK();
    return OSN_OK;
}
