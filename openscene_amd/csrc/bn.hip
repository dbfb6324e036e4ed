// Batch-norm statistics, fused normalise(+residual)(+ReLU), and backward on
// gfx950.  Pure streaming work (HBM-bound): 16-byte loads, lanes along the
// channel axis, fp64 accumulators, deterministic two-stage column reductions
// (per-block partials -> ordered final sum; no floating-point atomics).
//
// Function parity with torch.nn.BatchNorm1d as wrapped by MinkowskiBatchNorm,
// MinkowskiReLU and the BasicBlock residual add.
#include "common.h"
#include "epilogue.h"

namespace osn {

constexpr int CR_COLS = 64;    // columns per block (16 lanes x float4)
constexpr int CR_RL = 16;      // row lanes per block
constexpr int CR_MAX_BLOCKS = 512;

// The gradient arriving at a batch norm's output may be the SUM of several row-aligned matrices (a block input feeds
// conv1 and the residual; an encoder output feeds the next stride-2 conv and, through ME.cat, two convs of the decoder --
// models/mink_unet.py:116-174), each possibly a column window of a wider matrix (ld = row stride in floats).  The
// backward kernels add them while they read: no separate accumulation pass, no torch add.
constexpr int BN_MAX_SRC = 3;
struct GySrc {
    const float* p[BN_MAX_SRC];
    int64_t ld[BN_MAX_SRC];
    int n;
};

__device__ inline float4 gy_load(const GySrc& s, int64_t r, int col) {
    float4 g = *reinterpret_cast<const float4*>(s.p[0] + r * s.ld[0] + col);
#pragma unroll
    for (int i = 1; i < BN_MAX_SRC; ++i)
        if (i < s.n) {
            const float4 v = *reinterpret_cast<const float4*>(s.p[i] + r * s.ld[i] + col);
            g.x += v.x; g.y += v.y; g.z += v.z; g.w += v.w;
        }
    return g;
}


struct ColReducePlan {
    int n_rb;            // row blocks
    int n_cg;            // column groups
    int rows_per_block;
};

static ColReducePlan plan_colreduce(int64_t n, int c) {
    ColReducePlan p;
    p.n_cg = int(cdiv(c, CR_COLS));
    int64_t rb = cdiv(n, 4 * CR_RL);
    if (rb > CR_MAX_BLOCKS) rb = CR_MAX_BLOCKS;
    if (rb < 1) rb = 1;
    p.rows_per_block = int(cdiv(cdiv(n, rb), CR_RL) * CR_RL);
    p.n_rb = int(cdiv(n, p.rows_per_block));
    if (p.n_rb < 1) p.n_rb = 1;
    return p;
}

// MODE 0: (sum x, sum x^2)        MODE 1: (sum g, sum g*xhat), g = relu ? gy*(y>0) : gy
template <int MODE>
__global__ __launch_bounds__(256) void col_reduce_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                         const GySrc gy, const float* __restrict__ mean,
                                                         const float* __restrict__ var, float eps, int relu, int64_t n,
                                                         int c, int rows_per_block, double* __restrict__ partial,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta) {
    __shared__ double red[2][CR_RL][CR_COLS];
    const int tid = threadIdx.x;
    const int cl = tid & 15, rl = tid >> 4;
    const int col = blockIdx.y * CR_COLS + cl * 4;
    const bool on = col < c;
    const int64_t r0 = int64_t(blockIdx.x) * rows_per_block;
    const int64_t r1 = min(n, r0 + rows_per_block);
    double s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
    float4 mu = make_float4(0, 0, 0, 0), is = make_float4(0, 0, 0, 0);
    float4 ga = make_float4(0, 0, 0, 0), be = make_float4(0, 0, 0, 0);
    const bool from_x = MODE == 1 && relu && !y;             // the ReLU mask recomputed from x (no residual): y is not read
    if (MODE == 1 && on) {
        mu = *reinterpret_cast<const float4*>(mean + col);
        const float4 v = *reinterpret_cast<const float4*>(var + col);
        is = make_float4(bn_is(v.x, eps), bn_is(v.y, eps), bn_is(v.z, eps), bn_is(v.w, eps));
        if (from_x) {
            ga = *reinterpret_cast<const float4*>(gamma + col);
            be = *reinterpret_cast<const float4*>(beta + col);
        }
    }
    if (on) {
        for (int64_t r = r0 + rl; r < r1; r += CR_RL) {
            const float4 xv = *reinterpret_cast<const float4*>(x + r * c + col);
            if (MODE == 0) {
                s1[0] += xv.x; s1[1] += xv.y; s1[2] += xv.z; s1[3] += xv.w;
                s2[0] += double(xv.x) * xv.x; s2[1] += double(xv.y) * xv.y;
                s2[2] += double(xv.z) * xv.z; s2[3] += double(xv.w) * xv.w;
            } else {
                float4 g = gy_load(gy, r, col);
                if (from_x) {
                    g.x = bn_val(xv.x, mu.x, is.x, ga.x, be.x) > 0.f ? g.x : 0.f;
                    g.y = bn_val(xv.y, mu.y, is.y, ga.y, be.y) > 0.f ? g.y : 0.f;
                    g.z = bn_val(xv.z, mu.z, is.z, ga.z, be.z) > 0.f ? g.z : 0.f;
                    g.w = bn_val(xv.w, mu.w, is.w, ga.w, be.w) > 0.f ? g.w : 0.f;
                } else if (relu) {
                    const float4 yv = *reinterpret_cast<const float4*>(y + r * c + col);
                    g.x = yv.x > 0.f ? g.x : 0.f; g.y = yv.y > 0.f ? g.y : 0.f;
                    g.z = yv.z > 0.f ? g.z : 0.f; g.w = yv.w > 0.f ? g.w : 0.f;
                }
                s1[0] += g.x; s1[1] += g.y; s1[2] += g.z; s1[3] += g.w;
                s2[0] += double(g.x) * ((xv.x - mu.x) * is.x); s2[1] += double(g.y) * ((xv.y - mu.y) * is.y);
                s2[2] += double(g.z) * ((xv.z - mu.z) * is.z); s2[3] += double(g.w) * ((xv.w - mu.w) * is.w);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        red[0][rl][cl * 4 + j] = s1[j];
        red[1][rl][cl * 4 + j] = s2[j];
    }
    __syncthreads();
    if (tid < 2 * CR_COLS) {
        const int which = tid / CR_COLS, cc = tid % CR_COLS;
        double s = 0;
#pragma unroll
        for (int r = 0; r < CR_RL; ++r) s += red[which][r][cc];
        const int gc = blockIdx.y * CR_COLS + cc;
        if (gc < c) partial[(int64_t(blockIdx.x) * 2 + which) * c + gc] = s;
    }
}

// Final column sums: 64 columns x 16 slices of the partial blocks per workgroup
// (the partial blocks are summed in a fixed order => deterministic).
constexpr int FIN_COLS = 16, FIN_PARTS = 64;      // 64 slices: 8 dependent partial reads per thread at 512 row blocks (16 slices: 32 reads, 7.5 us per launch)

constexpr int FIN_ITERS = CR_MAX_BLOCKS / FIN_PARTS;     // partial blocks per thread (8)

__device__ inline void finalize_sums(const double* __restrict__ partial, int n_rb, int c, double& s1, double& s2,
                                     int& col) {
    __shared__ double red[2][FIN_PARTS][FIN_COLS];
    const int cl = threadIdx.x & (FIN_COLS - 1), part = threadIdx.x / FIN_COLS;
    col = blockIdx.x * FIN_COLS + cl;
    // all 2 x FIN_ITERS loads are issued before the first add (clamped addresses, masked afterwards): the kernel is
    // one memory round trip long instead of FIN_ITERS dependent ones (5 us -> 3 us per launch, 96 launches a step)
    double va[FIN_ITERS], vb[FIN_ITERS];
    const int cc = col < c ? col : 0;
#pragma unroll
    for (int i = 0; i < FIN_ITERS; ++i) {
        const int blk = part + FIN_PARTS * i;
        const int bc = blk < n_rb ? blk : 0;
        va[i] = partial[(int64_t(bc) * 2 + 0) * c + cc];
        vb[i] = partial[(int64_t(bc) * 2 + 1) * c + cc];
    }
    double a = 0, b = 0;
#pragma unroll
    for (int i = 0; i < FIN_ITERS; ++i) {
        const bool on = col < c && part + FIN_PARTS * i < n_rb;
        a += on ? va[i] : 0.0;
        b += on ? vb[i] : 0.0;
    }
    red[0][part][cl] = a;
    red[1][part][cl] = b;
    __syncthreads();
    // two levels, fixed order: slices 8 g .. 8 g + 7 by thread (g, cl), then the eight group sums by thread (0, cl)
    // (a single thread adding 64 slices is a chain of 128 dependent LDS reads + fp64 adds: 3 of the kernel's 5 us)
    double t1 = 0, t2 = 0;
    if (part < 8) {
#pragma unroll
        for (int q = 0; q < 8; ++q) { t1 += red[0][part * 8 + q][cl]; t2 += red[1][part * 8 + q][cl]; }
    }
    __syncthreads();
    if (part < 8) {
        red[0][part][cl] = t1;
        red[1][part][cl] = t2;
    }
    __syncthreads();
    s1 = 0; s2 = 0;
    if (part == 0) {
#pragma unroll
        for (int q = 0; q < 8; ++q) { s1 += red[0][q][cl]; s2 += red[1][q][cl]; }
    }
}

__global__ __launch_bounds__(FIN_COLS * FIN_PARTS) void bn_stats_finalize_kernel(
    const double* __restrict__ partial, int n_rb, int64_t n, int c, float* __restrict__ mean, float* __restrict__ var,
    float* __restrict__ running_mean, float* __restrict__ running_var, float momentum) {
    double s1, s2;
    int j;
    finalize_sums(partial, n_rb, c, s1, s2, j);
    if (threadIdx.x >= FIN_COLS || j >= c) return;
    const double m = s1 / double(n);
    double v = s2 / double(n) - m * m;
    if (v < 0) v = 0;
    mean[j] = float(m);
    var[j] = float(v);
    if (running_mean) running_mean[j] = (1.f - momentum) * running_mean[j] + momentum * float(m);
    if (running_var) {
        const double unb = n > 1 ? v * double(n) / double(n - 1) : v;
        running_var[j] = (1.f - momentum) * running_var[j] + momentum * float(unb);
    }
}

__global__ __launch_bounds__(FIN_COLS * FIN_PARTS) void bn_bwd_finalize_kernel(const double* __restrict__ partial,
                                                                              int n_rb, int c,
                                                                              float* __restrict__ sum_g,
                                                                              float* __restrict__ sum_gx) {
    double s1, s2;
    int j;
    finalize_sums(partial, n_rb, c, s1, s2, j);
    if (threadIdx.x >= FIN_COLS || j >= c) return;
    sum_g[j] = float(s1);
    sum_gx[j] = float(s2);
}

__device__ inline float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// (bn_val / bn_is: epilogue.h -- one definition shared with the convolution kernels' evaluation-mode epilogues)

__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                                       const float* __restrict__ var, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float eps,
                                                       const float* __restrict__ residual, int relu,
                                                       float* __restrict__ y, float* __restrict__ y2, int64_t ld2,
                                                       int64_t total4, int c4) {
    for (int64_t e = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; e < total4; e += int64_t(gridDim.x) * blockDim.x) {
        const int col = int(e % c4) * 4;
        const float4 xv = ld4(x + e * 4), mu = ld4(mean + col), vv = ld4(var + col), ga = ld4(gamma + col),
                     be = ld4(beta + col);
        float4 o;
        o.x = bn_val(xv.x, mu.x, bn_is(vv.x, eps), ga.x, be.x);
        o.y = bn_val(xv.y, mu.y, bn_is(vv.y, eps), ga.y, be.y);
        o.z = bn_val(xv.z, mu.z, bn_is(vv.z, eps), ga.z, be.z);
        o.w = bn_val(xv.w, mu.w, bn_is(vv.w, eps), ga.w, be.w);
        if (residual) {
            const float4 rv = ld4(residual + e * 4);
            o.x += rv.x; o.y += rv.y; o.z += rv.z; o.w += rv.w;
        }
        if (relu) {
            o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
        }
        *reinterpret_cast<float4*>(y + e * 4) = o;
        // second store: the same rows as a column window of a wider matrix (ME.cat written in place by its producers)
        if (y2) *reinterpret_cast<float4*>(y2 + (e / c4) * ld2 + col) = o;
    }
}

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                           const GySrc gy, const float* __restrict__ mean,
                                                           const float* __restrict__ var, const float* __restrict__ gamma,
                                                           float eps, int relu, int training,
                                                           const float* __restrict__ sum_g,
                                                           const float* __restrict__ sum_gx, float inv_n,
                                                           float* __restrict__ gx, float* __restrict__ gres,
                                                           int64_t total4, int c4, const float* __restrict__ beta) {
    const bool from_x = relu && !y;                          // ReLU mask recomputed from x: one tensor less to read
    for (int64_t e = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; e < total4; e += int64_t(gridDim.x) * blockDim.x) {
        const int col = int(e % c4) * 4;
        float4 g = gy_load(gy, e / c4, col);
        const float4 vv = ld4(var + col), ga = ld4(gamma + col);
        const float4 is = make_float4(bn_is(vv.x, eps), bn_is(vv.y, eps), bn_is(vv.z, eps), bn_is(vv.w, eps));
        float4 xv = make_float4(0, 0, 0, 0), mu = make_float4(0, 0, 0, 0);
        if (training || from_x) { xv = ld4(x + e * 4); mu = ld4(mean + col); }
        if (from_x) {
            const float4 be = ld4(beta + col);
            g.x = bn_val(xv.x, mu.x, is.x, ga.x, be.x) > 0.f ? g.x : 0.f;
            g.y = bn_val(xv.y, mu.y, is.y, ga.y, be.y) > 0.f ? g.y : 0.f;
            g.z = bn_val(xv.z, mu.z, is.z, ga.z, be.z) > 0.f ? g.z : 0.f;
            g.w = bn_val(xv.w, mu.w, is.w, ga.w, be.w) > 0.f ? g.w : 0.f;
        } else if (relu) {
            const float4 yv = ld4(y + e * 4);
            g.x = yv.x > 0.f ? g.x : 0.f; g.y = yv.y > 0.f ? g.y : 0.f;
            g.z = yv.z > 0.f ? g.z : 0.f; g.w = yv.w > 0.f ? g.w : 0.f;
        }
        if (gres) *reinterpret_cast<float4*>(gres + e * 4) = g;
        float4 o;
        if (training) {
            const float4 sg = ld4(sum_g + col), sx = ld4(sum_gx + col);
            o.x = ga.x * is.x * (g.x - sg.x * inv_n - (xv.x - mu.x) * is.x * sx.x * inv_n);
            o.y = ga.y * is.y * (g.y - sg.y * inv_n - (xv.y - mu.y) * is.y * sx.y * inv_n);
            o.z = ga.z * is.z * (g.z - sg.z * inv_n - (xv.z - mu.z) * is.z * sx.z * inv_n);
            o.w = ga.w * is.w * (g.w - sg.w * inv_n - (xv.w - mu.w) * is.w * sx.w * inv_n);
        } else {
            o.x = ga.x * is.x * g.x; o.y = ga.y * is.y * g.y; o.z = ga.z * is.z * g.z; o.w = ga.w * is.w * g.w;
        }
        *reinterpret_cast<float4*>(gx + e * 4) = o;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Small maps (the deep U-Net levels: 700 - 3 000 rows): the three launches of a training-mode batch norm (column sums,
// finalize, apply) are each a few microseconds of launch latency around almost no work.  Up to SB_MAX_ROWS rows one
// workgroup of 1024 threads owns 8 columns, keeps its rows IN REGISTERS between the statistics and the apply pass (x is
// read once), and reduces with wave shuffles + one LDS round in a fixed order (deterministic).  Same formulas as the
// three-kernel path (fp64 sums, fp32 mean / var / apply).  Measured (profiles/r03_s3): 14.5 us per launch against
// ~15 us for the three launches it replaces -- a tie in GPU time, two launches (and their gaps) fewer.  The same design
// for the BACKWARD pass (x, y and up to three gradient sources per row in registers) was slower (24 us against 14 us)
// and is not kept.
constexpr int SB_THREADS = 1024, SB_RL = 512, SB_PER = 8, SB_COLS = 8;
constexpr int SB_MAX_ROWS = SB_RL * SB_PER;       // 4096

__device__ inline double sb_wave_sum(double v) {
    // lanes of one wave that share bit 0 (the column lane): xor-shuffle over the 32 row lanes
#pragma unroll
    for (int off = 2; off < 64; off <<= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// sums[0..3] / sums[4..7] of this thread's 4 columns -> totals over the workgroup's rows, in every thread
__device__ inline void sb_block_sums(double (&s)[8], int tid, int cl) {
    __shared__ double red[SB_THREADS / 64][2][8];
    __shared__ double tot[2][8];
#pragma unroll
    for (int q = 0; q < 8; ++q) s[q] = sb_wave_sum(s[q]);
    const int lane = tid & 63, wave = tid >> 6;
    if ((lane >> 1) == 0) {
#pragma unroll
        for (int q = 0; q < 8; ++q) red[wave][cl][q] = s[q];
    }
    __syncthreads();
    if (tid < 16) {
        const int c2 = tid >> 3, q = tid & 7;
        double t = 0;
#pragma unroll
        for (int w = 0; w < SB_THREADS / 64; ++w) t += red[w][c2][q];
        tot[c2][q] = t;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 8; ++q) s[q] = tot[cl][q];
}

__global__ __launch_bounds__(SB_THREADS) void bn_small_fwd_kernel(const float* __restrict__ x, int n, int c,
                                                                  const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                  float eps, const float* __restrict__ residual, int relu,
                                                                  float momentum, float* __restrict__ mean_out,
                                                                  float* __restrict__ var_out, float* __restrict__ running_mean,
                                                                  float* __restrict__ running_var, float* __restrict__ y,
                                                                  float* __restrict__ y2, int64_t ld2) {
    const int tid = threadIdx.x, cl = tid & 1, rl = tid >> 1;
    const int col = blockIdx.x * SB_COLS + cl * 4;
    const bool con = col < c;
    const int cc = con ? col : 0;
    float4 v[SB_PER];
    double s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < SB_PER; ++j) {
        const int r = rl + SB_RL * j;
        const bool ok = con && r < n;
        v[j] = ld4(x + int64_t(ok ? r : 0) * c + cc);
        if (ok) {
            s[0] += v[j].x; s[1] += v[j].y; s[2] += v[j].z; s[3] += v[j].w;
            s[4] += double(v[j].x) * v[j].x; s[5] += double(v[j].y) * v[j].y;
            s[6] += double(v[j].z) * v[j].z; s[7] += double(v[j].w) * v[j].w;
        }
    }
    sb_block_sums(s, tid, cl);
    float mu[4], vr[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const double m = s[q] / double(n);
        double var = s[4 + q] / double(n) - m * m;
        if (var < 0) var = 0;
        mu[q] = float(m);
        vr[q] = float(var);
        if (rl == 0 && con) {
            mean_out[col + q] = mu[q];
            var_out[col + q] = vr[q];
            if (running_mean) running_mean[col + q] = (1.f - momentum) * running_mean[col + q] + momentum * mu[q];
            if (running_var) {
                const double unb = n > 1 ? var * double(n) / double(n - 1) : var;
                running_var[col + q] = (1.f - momentum) * running_var[col + q] + momentum * float(unb);
            }
        }
    }
    if (!con) return;
    const float4 ga = ld4(gamma + col), be = ld4(beta + col);
    const float is0 = bn_is(vr[0], eps), is1 = bn_is(vr[1], eps), is2 = bn_is(vr[2], eps), is3 = bn_is(vr[3], eps);
#pragma unroll
    for (int j = 0; j < SB_PER; ++j) {
        const int r = rl + SB_RL * j;
        if (r >= n) continue;
        float4 o;
        o.x = bn_val(v[j].x, mu[0], is0, ga.x, be.x);
        o.y = bn_val(v[j].y, mu[1], is1, ga.y, be.y);
        o.z = bn_val(v[j].z, mu[2], is2, ga.z, be.z);
        o.w = bn_val(v[j].w, mu[3], is3, ga.w, be.w);
        if (residual) {
            const float4 rv = ld4(residual + int64_t(r) * c + col);
            o.x += rv.x; o.y += rv.y; o.z += rv.z; o.w += rv.w;
        }
        if (relu) {
            o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
        }
        *reinterpret_cast<float4*>(y + int64_t(r) * c + col) = o;
        if (y2) *reinterpret_cast<float4*>(y2 + int64_t(r) * ld2 + col) = o;
    }
}

static int ew_grid(int64_t total4) {
    int64_t g = cdiv(total4, 256);
    if (g > 2048) g = 2048;
    if (g < 1) g = 1;
    return int(g);
}

}  // namespace osn

using namespace osn;

extern "C" size_t osn_bn_ws_bytes(int64_t n, int c) {
    (void)n;
    return size_t(CR_MAX_BLOCKS) * 2 * size_t(c) * 8 + 256;
}

extern "C" int osn_bn_stats(const float* x, int64_t n, int c, float* mean, float* var, float* running_mean,
                            float* running_var, float momentum, void* ws, size_t ws_bytes, osn_stream_t stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    OSN_REQUIRE(n >= 1 && c >= 4 && (c & 3) == 0, OSN_E_ARG, "osn_bn_stats: need n >= 1 and c %% 4 == 0 (n=%lld c=%d)", (long long)n, c);
    OSN_REQUIRE(x && mean && var && aligned16(x), OSN_E_ARG, "osn_bn_stats: null or unaligned pointer");
    ColReducePlan p = plan_colreduce(n, c);
    const size_t need = size_t(p.n_rb) * 2 * size_t(c) * 8;
    OSN_REQUIRE(ws && ws_bytes >= need, OSN_E_WS, "osn_bn_stats: workspace %zu < %zu", ws_bytes, need);
    double* partial = static_cast<double*>(ws);
    hipLaunchKernelGGL((col_reduce_kernel<0>), dim3(p.n_rb, p.n_cg), dim3(256), 0, st, x, (const float*)nullptr,
                       GySrc{}, (const float*)nullptr, (const float*)nullptr, 0.f, 0, n, c,
                       p.rows_per_block, partial, (const float*)nullptr, (const float*)nullptr);
    hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3(cdiv(c, FIN_COLS)), dim3(FIN_COLS * FIN_PARTS), 0, st, partial, p.n_rb, n, c, mean, var,
                       running_mean, running_var, momentum);
    OSN_LAUNCH_CHECK();
    return OSN_OK;
}

extern "C" int osn_bn_apply2(const float* x, const float* mean, const float* var, const float* gamma, const float* beta,
                             float eps, const float* residual, int relu, float* y, float* y2, int64_t ld2, int64_t n, int c,
                             osn_stream_t stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    OSN_REQUIRE(n >= 0 && c >= 4 && (c & 3) == 0, OSN_E_ARG, "osn_bn_apply: need c %% 4 == 0 (c=%d)", c);
    if (n == 0) return OSN_OK;
    OSN_REQUIRE(x && mean && var && gamma && beta && y, OSN_E_ARG, "osn_bn_apply: null pointer");
    OSN_REQUIRE(aligned16(x) && aligned16(y) && aligned16(mean) && aligned16(var) && aligned16(gamma) && aligned16(beta) &&
                    (!residual || aligned16(residual)),
                OSN_E_ARG, "osn_bn_apply: pointers must be 16-byte aligned");
    OSN_REQUIRE(!y2 || (aligned16(y2) && ld2 >= c && (ld2 & 3) == 0), OSN_E_ARG,
                "osn_bn_apply2: the second destination needs a 16-byte aligned pointer and a row stride >= c, %% 4 == 0");
    const int64_t total4 = n * (c / 4);
    hipLaunchKernelGGL(bn_apply_kernel, dim3(ew_grid(total4)), dim3(256), 0, st, x, mean, var, gamma, beta, eps,
                       residual, relu, y, y2, ld2, total4, c / 4);
    OSN_LAUNCH_CHECK();
    return OSN_OK;
}

extern "C" int osn_bn_apply(const float* x, const float* mean, const float* var, const float* gamma, const float* beta,
                            float eps, const float* residual, int relu, float* y, int64_t n, int c,
                            osn_stream_t stream) {
    return osn_bn_apply2(x, mean, var, gamma, beta, eps, residual, relu, y, nullptr, 0, n, c, stream);
}

// Training-mode forward in ONE call: statistics (+ running buffers) and the fused normalise (+ residual) (+ ReLU) pass
// (+ the second destination of osn_bn_apply2).  Up to SB_MAX_ROWS rows: ONE launch (bn_small_fwd_kernel); above: the
// kernels of osn_bn_stats + osn_bn_apply2.
extern "C" int osn_bn_forward_train2(const float* x, int64_t n, int c, const float* gamma, const float* beta, float eps,
                                     const float* residual, int relu, float momentum, float* mean, float* var,
                                     float* running_mean, float* running_var, float* y, float* y2, int64_t ld2, void* ws,
                                     size_t ws_bytes, osn_stream_t stream) {
    if (n >= 1 && n <= SB_MAX_ROWS && c >= 4 && (c & 3) == 0) {
        hipStream_t st = static_cast<hipStream_t>(stream);
        OSN_REQUIRE(x && mean && var && gamma && beta && y, OSN_E_ARG, "osn_bn_forward_train: null pointer");
        OSN_REQUIRE(aligned16(x) && aligned16(y) && aligned16(gamma) && aligned16(beta) && (!residual || aligned16(residual)),
                    OSN_E_ARG, "osn_bn_forward_train: pointers must be 16-byte aligned");
        OSN_REQUIRE(!y2 || (aligned16(y2) && ld2 >= c && (ld2 & 3) == 0), OSN_E_ARG,
                    "osn_bn_forward_train2: the second destination needs a 16-byte aligned pointer and a row stride >= c, %% 4 == 0");
        hipLaunchKernelGGL(bn_small_fwd_kernel, dim3(unsigned(cdiv(c, SB_COLS))), dim3(SB_THREADS), 0, st, x, int(n), c, gamma,
                           beta, eps, residual, relu, momentum, mean, var, running_mean, running_var, y, y2, ld2);
        OSN_LAUNCH_CHECK();
        return OSN_OK;
    }
    int rc = osn_bn_stats(x, n, c, mean, var, running_mean, running_var, momentum, ws, ws_bytes, stream);
    if (rc) return rc;
    return osn_bn_apply2(x, mean, var, gamma, beta, eps, residual, relu, y, y2, ld2, n, c, stream);
}

extern "C" int osn_bn_forward_train(const float* x, int64_t n, int c, const float* gamma, const float* beta, float eps,
                                    const float* residual, int relu, float momentum, float* mean, float* var,
                                    float* running_mean, float* running_var, float* y, void* ws, size_t ws_bytes,
                                    osn_stream_t stream) {
    return osn_bn_forward_train2(x, n, c, gamma, beta, eps, residual, relu, momentum, mean, var, running_mean, running_var, y,
                                 nullptr, 0, ws, ws_bytes, stream);
}

extern "C" int osn_bn_backward_multi2(const float* x, const float* y, const float* const* gy, const int64_t* gy_ld, int n_gy,
                                      const float* mean, const float* var, const float* gamma, const float* beta, float eps, int relu,
                                     int training, float* gx, float* gres, float* ggamma, float* gbeta, int64_t n, int c,
                                     void* ws, size_t ws_bytes, osn_stream_t stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    OSN_REQUIRE(n >= 1 && c >= 4 && (c & 3) == 0, OSN_E_ARG, "osn_bn_backward: need n >= 1 and c %% 4 == 0");
    OSN_REQUIRE(gy && gy_ld && n_gy >= 1 && n_gy <= BN_MAX_SRC, OSN_E_ARG,
                "osn_bn_backward: %d gradient sources (1 .. %d are supported)", n_gy, BN_MAX_SRC);
    // y == null with relu: the mask is recomputed from x (the forward expression, bit for bit) -- needs beta, and a batch
    // norm WITHOUT a residual (with one, y = relu(bn(x) + residual) cannot be rebuilt from x alone)
    OSN_REQUIRE(x && mean && var && gamma && gx && ggamma && gbeta && (!relu || y || beta), OSN_E_ARG,
                "osn_bn_backward: null pointer");
    OSN_REQUIRE(!(relu && !y) || !gres, OSN_E_ARG, "osn_bn_backward: the ReLU mask can be recomputed from x only without a residual");
    OSN_REQUIRE(!beta || aligned16(beta), OSN_E_ARG, "osn_bn_backward: beta must be 16-byte aligned");
    OSN_REQUIRE(aligned16(x) && aligned16(gx) && aligned16(mean) && aligned16(var) && aligned16(gamma) &&
                    aligned16(ggamma) && aligned16(gbeta) && (!y || aligned16(y)) && (!gres || aligned16(gres)),
                OSN_E_ARG, "osn_bn_backward: pointers must be 16-byte aligned");
    GySrc src{};
    src.n = n_gy;
    for (int i = 0; i < BN_MAX_SRC; ++i) {
        const int j = i < n_gy ? i : 0;                  // unused slots repeat source 0 (never read)
        OSN_REQUIRE(gy[j] && aligned16(gy[j]) && gy_ld[j] >= c && (gy_ld[j] & 3) == 0, OSN_E_ARG,
                    "osn_bn_backward: gradient source %d needs a 16-byte aligned pointer and a row stride >= c, %% 4 == 0", j);
        src.p[i] = gy[j];
        src.ld[i] = gy_ld[j];
    }
    ColReducePlan p = plan_colreduce(n, c);
    const size_t need = size_t(p.n_rb) * 2 * size_t(c) * 8;
    OSN_REQUIRE(ws && ws_bytes >= need, OSN_E_WS, "osn_bn_backward: workspace %zu < %zu", ws_bytes, need);
    double* partial = static_cast<double*>(ws);
    hipLaunchKernelGGL((col_reduce_kernel<1>), dim3(p.n_rb, p.n_cg), dim3(256), 0, st, x, y, src, mean, var, eps, relu, n,
                       c, p.rows_per_block, partial, gamma, beta);
    // ggamma = sum g*xhat, gbeta = sum g  (also the two column sums the apply pass needs)
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(cdiv(c, FIN_COLS)), dim3(FIN_COLS * FIN_PARTS), 0, st, partial, p.n_rb, c, gbeta, ggamma);
    const int64_t total4 = n * (c / 4);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(ew_grid(total4)), dim3(256), 0, st, x, y, src, mean, var, gamma, eps,
                       relu, training, gbeta, ggamma, 1.f / float(n), gx, gres, total4, c / 4, beta);
    OSN_LAUNCH_CHECK();
    return OSN_OK;
}

extern "C" int osn_bn_backward_multi(const float* x, const float* y, const float* const* gy, const int64_t* gy_ld, int n_gy,
                                     const float* mean, const float* var, const float* gamma, float eps, int relu,
                                     int training, float* gx, float* gres, float* ggamma, float* gbeta, int64_t n, int c,
                                     void* ws, size_t ws_bytes, osn_stream_t stream) {
    OSN_REQUIRE(!relu || y, OSN_E_ARG, "osn_bn_backward_multi: relu needs y (or osn_bn_backward_multi2 with beta)");
    return osn_bn_backward_multi2(x, y, gy, gy_ld, n_gy, mean, var, gamma, nullptr, eps, relu, training, gx, gres, ggamma, gbeta, n, c,
                                  ws, ws_bytes, stream);
}

extern "C" int osn_bn_backward(const float* x, const float* y, const float* gy, const float* mean, const float* var,
                               const float* gamma, float eps, int relu, int training, float* gx, float* gres,
                               float* ggamma, float* gbeta, int64_t n, int c, void* ws, size_t ws_bytes,
                               osn_stream_t stream) {
    const int64_t ld = c;
    return osn_bn_backward_multi(x, y, &gy, &ld, 1, mean, var, gamma, eps, relu, training, gx, gres, ggamma, gbeta, n, c, ws,
                                 ws_bytes, stream);
}
