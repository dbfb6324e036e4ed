// Open-vocabulary query on gfx950: fused gather + fp32->fp16 cast + fp16 MFMA
// GEMM against the CLIP text matrix + fp16 rounding + row argmax.
//
// Replaces run/evaluate.py:290-292 / 302-324 and run/distill.py:423-425:
//   pred = feats[inds_reverse].half() @ text_features.t();  torch.max(pred, 1)[1]
//
// HBM-bound on reading X once through the gather; the text
// matrix (<= 246 KB) stays L2-resident.  MFMA is used because this IS a dense
// contraction: v_mfma_f32_32x32x16_f16, A = 32 point rows x 16 k, B = text rows
// (text is [C, D] row-major = the K-contiguous B operand, no transpose needed).
#include "common.h"

namespace osn {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

constexpr int Q_BM = 128;   // points per workgroup (4 waves x 32)
constexpr int Q_DK = 64;    // feature chunk
constexpr int Q_LD = Q_DK + 8;  // padded LDS row (144 B: 16-B aligned, conflict-free ds_read_b128)

// sources: point p reads row g0[p] of X0 (or p if g0 null); if sel && sel[p], row g1[p] of X1.
// NBUF chunks of point rows (and of the text) in flight per thread, WGS workgroups per CU (register budget 512 / WGS).
template <int CT, int NBUF, int WGS>
__global__ __launch_bounds__(256, WGS) void query_kernel(const float* __restrict__ X0, const int64_t* __restrict__ g0,
                                                    const float* __restrict__ X1, const int64_t* __restrict__ g1,
                                                    const uint8_t* __restrict__ sel, const float* __restrict__ rowdiv,
                                                    const _Float16* __restrict__ T, _Float16* __restrict__ scores,
                                                    int64_t* __restrict__ argmax, float* __restrict__ rowmax,
                                                    int64_t n, int d, int c) {
    __shared__ __attribute__((aligned(16))) _Float16 Xs[Q_BM][Q_LD];
    __shared__ __attribute__((aligned(16))) _Float16 Ts[CT * 32][Q_LD];
    // byte offset of every point's source row RELATIVE TO X0 (also for rows of X1): the gathers then stay X0-based
    // global loads.  (A pointer fetched from LDS has lost its address space: the loads become FLAT, which also count on
    // the LDS counter, and every LDS wait of the MFMA loop would drain the prefetched rows.)
    __shared__ int64_t rowoff[Q_BM];
    __shared__ float rowden[Q_BM];
    __shared__ float bestv_s[Q_BM];      // running row maximum / its column over the column groups (first maximum wins)
    __shared__ int besti_s[Q_BM];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t row0 = int64_t(blockIdx.x) * Q_BM;

    if (tid < Q_BM) {
        const int64_t p = row0 + tid;
        int64_t off = -1;                                   // -1: no such point
        float den = 1.f;
        if (p < n) {
            if (sel && sel[p])
                off = int64_t(reinterpret_cast<uintptr_t>(X1) - reinterpret_cast<uintptr_t>(X0)) + (g1 ? g1[p] : p) * int64_t(d) * 4;
            else
                off = (g0 ? g0[p] : p) * int64_t(d) * 4;
            if (rowdiv) den = rowdiv[p];
        }
        rowoff[tid] = off;
        rowden[tid] = den;
        bestv_s[tid] = -INFINITY;
        besti_s[tid] = 0x7fffffff;
    }
    __syncthreads();

    for (int cg0 = 0; cg0 < c; cg0 += 32 * CT) {
        f32x16 acc[CT];
#pragma unroll
        for (int t = 0; t < CT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

        // software pipeline over the feature chunks with NBUF chunks in flight: while the MFMAs of chunk i run from LDS,
        // chunks i+1 .. i+NBUF-1 are on their way and chunk i+NBUF is issued into the register buffer chunk i just left.
        // The kernel is bound by bytes in flight x latency (random 256-byte segments) and the register file is the
        // largest buffer a CU has.  The text chunk (L2-resident) travels with its rows at the same distance and is
        // issued first: the memory counter retires in order, so waiting for chunk i+1 leaves the NBUF-1 younger
        // fetches in flight.  All loads are unconditional from clamped addresses.
        float4 px[NBUF][Q_BM / 16];
        uint4 pt[NBUF][CT];
        const int xq = tid & 15, xr = tid >> 4;
        auto fetch = [&](int d0, float4 (&pxb)[Q_BM / 16], uint4 (&ptb)[CT]) {
#pragma unroll
            for (int j = 0; j < CT; ++j) {
                const int f = tid + 256 * j;
                const int trow = f >> 3, ch = f & 7;
                const int col = cg0 + trow;
                const bool ok = col < c && d0 + ch * 8 < d;
                ptb[j] = *reinterpret_cast<const uint4*>(ok ? T + int64_t(col) * d + d0 + ch * 8 : T);
            }
#pragma unroll
            for (int ps = 0; ps < Q_BM / 16; ++ps) {
                const int64_t off = rowoff[ps * 16 + xr];
                const bool ok = row0 + ps * 16 + xr < n && d0 + xq * 4 < d;
                pxb[ps] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(X0) + (ok ? off + (d0 + xq * 4) * 4 : 0));
            }
        };
        auto stash = [&](int d0, float4 (&pxb)[Q_BM / 16], uint4 (&ptb)[CT]) {
#pragma unroll
            for (int ps = 0; ps < Q_BM / 16; ++ps) {
                const int row = ps * 16 + xr;
                const bool ok = row0 + row < n && d0 + xq * 4 < d;
                float4 v = pxb[ps];
                if (!ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (rowdiv) {
                    const float den = rowden[row];
                    v.x /= den; v.y /= den; v.z /= den; v.w /= den;
                }
                half4 h;
                h[0] = (_Float16)v.x; h[1] = (_Float16)v.y; h[2] = (_Float16)v.z; h[3] = (_Float16)v.w;
                *reinterpret_cast<half4*>(&Xs[row][xq * 4]) = h;
            }
#pragma unroll
            for (int j = 0; j < CT; ++j) {
                const int f = tid + 256 * j;
                const int trow = f >> 3, ch = f & 7;
                const bool ok = cg0 + trow < c && d0 + ch * 8 < d;
                uint4 v = ptb[j];
                if (!ok) v = make_uint4(0, 0, 0, 0);
                *reinterpret_cast<uint4*>(&Ts[trow][ch * 8]) = v;
            }
        };
        // MFMA phase: the fragments of k-step ks+1 are read from LDS while the MFMAs of k-step ks run (two register
        // sets; the scheduling fences keep the compiler from sinking the reads back to their first use, which leaves
        // every MFMA waiting for its own LDS round trip)
        auto mfmas = [&]() {
            const int arow = wave * 32 + (lane & 31);
            const int kh = 8 * (lane >> 5);
            half8 fa[2], fb[2][CT];
            auto frags = [&](int ks, int w) {
                fa[w] = *reinterpret_cast<const half8*>(&Xs[arow][ks * 16 + kh]);
#pragma unroll
                for (int t = 0; t < CT; ++t)
                    fb[w][t] = *reinterpret_cast<const half8*>(&Ts[t * 32 + (lane & 31)][ks * 16 + kh]);
            };
            frags(0, 0);
#pragma unroll
            for (int ks = 0; ks < Q_DK / 16; ++ks) {
                __builtin_amdgcn_sched_barrier(0);
                if (ks + 1 < Q_DK / 16) frags(ks + 1, (ks + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < CT; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[ks & 1], fb[ks & 1][t], acc[t], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        // (scheduling fences: the fetches must be ISSUED in program order -- the compiler is free to reorder independent
        // loads, and the in-order memory counter then makes the oldest buffer wait for the youngest load)
#pragma unroll
        for (int b = 0; b < NBUF; ++b) {
            fetch(b * Q_DK, px[b], pt[b]);
            __builtin_amdgcn_sched_barrier(0);
        }
        stash(0, px[0], pt[0]);
        __syncthreads();
        for (int base = 0; base < d; base += NBUF * Q_DK) {
#pragma unroll
            for (int b = 0; b < NBUF; ++b) {
                const int d0 = base + b * Q_DK;                    // chunk d0 is in LDS, register buffer b is free
                if (d0 < d) {
                    __builtin_amdgcn_sched_barrier(0);
                    // unconditional (a chunk past the end loads the first 16 bytes of X0 / T for every lane): with a
                    // conditional fetch the compiler must assume the older buffers may be the YOUNGEST loads on some
                    // path and drains the counter to zero before every stash
                    fetch(d0 + NBUF * Q_DK, px[b], pt[b]);
                    __builtin_amdgcn_sched_barrier(0);
                    mfmas();
                    __syncthreads();
                    if (d0 + Q_DK < d) stash(d0 + Q_DK, px[(b + 1) % NBUF], pt[(b + 1) % NBUF]);
                    __syncthreads();
                }
            }
        }
        // ---- group epilogue: round to fp16, optional store, row argmax of the group merged into the running one
        // (kept in LDS between groups: 32 registers less in the main loop; a thread always owns the same rows)
        float bestv[16];
        int besti[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) { bestv[r] = -INFINITY; besti[r] = 0x7fffffff; }
#pragma unroll
        for (int t = 0; t < CT; ++t) {
            const int col = cg0 + t * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t row = row0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const _Float16 hv = (_Float16)acc[t][r];
                const float v = (float)hv;
                if (col < c) {
                    if (scores && row < n) scores[row * c + col] = hv;
                    if (v > bestv[r] || (v == bestv[r] && col < besti[r])) { bestv[r] = v; besti[r] = col; }
                }
            }
        }
        // reduce over the 32 lanes that hold one row's columns
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float v = bestv[r];
            int i = besti[r];
#pragma unroll
            for (int m = 1; m < 32; m <<= 1) {
                const float ov = __shfl_xor(v, m, 64);
                const int oi = __shfl_xor(i, m, 64);
                if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
            }
            if ((lane & 31) == 0) {
                const int lrow = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const float pv = bestv_s[lrow];
                const int pi = besti_s[lrow];
                if (v > pv || (v == pv && i < pi)) { bestv_s[lrow] = v; besti_s[lrow] = i; }
            }
        }
    }
    __syncthreads();
    if (tid < Q_BM && row0 + tid < n) {
        const int i = besti_s[tid];
        if (argmax) argmax[row0 + tid] = (i == 0x7fffffff) ? 0 : i;
        if (rowmax) rowmax[row0 + tid] = bestv_s[tid];
    }
}

// den[p] = ||X[g[p]]||_2 + eps   (one wave per point)
__global__ __launch_bounds__(256) void row_norm_kernel(const float* __restrict__ X, const int64_t* __restrict__ g,
                                                       int64_t n, int d, float eps, float* __restrict__ den) {
    const int lane = threadIdx.x & 63;
    const int64_t p = int64_t(blockIdx.x) * 4 + (threadIdx.x >> 6);
    if (p >= n) return;
    const float* src = X + (g ? g[p] : p) * int64_t(d);
    float s = 0.f;
    for (int j = lane * 4; j < d; j += 256) {
        const float4 v = *reinterpret_cast<const float4*>(src + j);
        s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
    if (lane == 0) den[p] = sqrtf(s) + eps;
}

__global__ void ensemble_select_kernel(const float* __restrict__ max_d, const float* __restrict__ max_f, int64_t n,
                                       uint8_t* __restrict__ sel) {
    const int64_t p = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (p < n) sel[p] = max_d[p] < max_f[p] ? 1 : 0;
}

// labels[p] = argmax_c scores[g ? g[p] : p][0 .. c)   (first maximum wins, as torch.max / torch.argmax): one wave per point,
// lanes over the row's columns (coalesced), butterfly reduction on (value, column)
__global__ __launch_bounds__(256) void rows_argmax_kernel(const float* __restrict__ scores, int64_t ld, int c,
                                                          const int64_t* __restrict__ g, int64_t n_pts, int64_t n_rows,
                                                          int64_t* __restrict__ labels) {
    const int lane = threadIdx.x & 63;
    const int64_t p = int64_t(blockIdx.x) * 4 + (threadIdx.x >> 6);
    if (p >= n_pts) return;
    int64_t r = g ? g[p] : p;
    if (r < 0 || r >= n_rows) r = 0;                          // (the wrapper validates the range of a gather index it is given)
    const float* row = scores + r * ld;
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int col = lane; col < c; col += 64) {
        const float v = row[col];
        if (v > bv || (v == bv && col < bi) || bi == 0x7fffffff) { bv = v; bi = col; }
    }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
        const float ov = __shfl_xor(bv, m, 64);
        const int oi = __shfl_xor(bi, m, 64);
        if (oi != 0x7fffffff && (bi == 0x7fffffff || ov > bv || (ov == bv && oi < bi))) { bv = ov; bi = oi; }
    }
    if (lane == 0) labels[p] = bi == 0x7fffffff ? 0 : bi;
}

static int launch_query(hipStream_t st, const float* X0, const int64_t* g0, const float* X1, const int64_t* g1,
                        const uint8_t* sel, const float* rowdiv, const _Float16* T, _Float16* scores, int64_t* argmax,
                        float* rowmax, int64_t n, int d, int c) {
    const dim3 grid(cdiv(n, Q_BM)), block(256);
    const int ct = int(cdiv(c, 32));
#define OSN_Q(CT, NB, WG) hipLaunchKernelGGL((query_kernel<CT, NB, WG>), grid, block, 0, st, X0, g0, X1, g1, sel, rowdiv, T, scores, argmax, rowmax, n, d, c)
    // one chunk in flight per thread; three workgroups per CU where the registers allow it (<= 64 labels).  Measured
    // alternatives (profiles/r02_s6_query_variants.txt): deeper register pipelines (2-4 chunks, also with one 512-register
    // workgroup per CU) and a column-split kernel with the text read straight from L2 are all slower.
    if (ct <= 1) OSN_Q(1, 1, 3);
    else if (ct <= 2) OSN_Q(2, 1, 3);
    else if (ct <= 3) OSN_Q(3, 1, 2);
    else OSN_Q(5, 1, 2);
#undef OSN_Q
    OSN_LAUNCH_CHECK();
    return OSN_OK;
}

}  // namespace osn

using namespace osn;

extern "C" int osn_cosine_query(const float* X, const int64_t* gather, const void* text_f16, void* scores_f16,
                                int64_t* argmax, int64_t n, int d, int c, osn_stream_t stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    OSN_REQUIRE(n >= 0 && d >= 8 && (d & 7) == 0 && c >= 1, OSN_E_ARG,
                "osn_cosine_query: need d %% 8 == 0 and c >= 1 (d=%d c=%d)", d, c);
    if (n == 0) return OSN_OK;
    OSN_REQUIRE(X && text_f16 && (argmax || scores_f16), OSN_E_ARG, "osn_cosine_query: null pointer");
    OSN_REQUIRE(aligned16(X) && aligned16(text_f16), OSN_E_ARG, "osn_cosine_query: X and text must be 16-byte aligned");
    return launch_query(st, X, gather, nullptr, nullptr, nullptr, nullptr, static_cast<const _Float16*>(text_f16),
                        static_cast<_Float16*>(scores_f16), argmax, nullptr, n, d, c);
}

extern "C" int osn_rows_argmax(const float* scores, int64_t ld, int c, const int64_t* gather, int64_t n_pts, int64_t n_rows,
                               int64_t* labels, osn_stream_t stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    OSN_REQUIRE(n_pts >= 0 && n_rows >= 1 && c >= 1 && ld >= c, OSN_E_ARG, "osn_rows_argmax: n_pts=%lld n_rows=%lld c=%d ld=%lld",
                (long long)n_pts, (long long)n_rows, c, (long long)ld);
    if (n_pts == 0) return OSN_OK;
    OSN_REQUIRE(scores && labels, OSN_E_ARG, "osn_rows_argmax: null pointer");
    hipLaunchKernelGGL(rows_argmax_kernel, dim3(unsigned(cdiv(n_pts, 4))), dim3(256), 0, st, scores, ld, c, gather, n_pts, n_rows, labels);
    OSN_LAUNCH_CHECK();
    return OSN_OK;
}

extern "C" size_t osn_query_ensemble_ws_bytes(int64_t n) {
    const size_t m = size_t(n > 0 ? n : 1);
    return 4 * align_up(m * 4, 256) + align_up(m, 256);
}

extern "C" int osn_query_ensemble(const float* X_distill, const int64_t* gather_distill, const float* X_fusion,
                                  const int64_t* gather_fusion, const void* text_f16, void* scores_f16,
                                  int64_t* argmax, uint8_t* sel_out, int64_t n, int d, int c, void* ws, size_t ws_bytes,
                                  osn_stream_t stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    OSN_REQUIRE(n >= 0 && d >= 8 && (d & 7) == 0 && c >= 1, OSN_E_ARG, "osn_query_ensemble: need d %% 8 == 0 and c >= 1");
    if (n == 0) return OSN_OK;
    OSN_REQUIRE(X_distill && X_fusion && text_f16 && (argmax || scores_f16), OSN_E_ARG, "osn_query_ensemble: null pointer");
    OSN_REQUIRE(aligned16(X_distill) && aligned16(X_fusion) && aligned16(text_f16), OSN_E_ARG,
                "osn_query_ensemble: feature and text pointers must be 16-byte aligned");
    OSN_REQUIRE(ws && ws_bytes >= osn_query_ensemble_ws_bytes(n), OSN_E_WS, "osn_query_ensemble: workspace too small");
    char* p = static_cast<char*>(ws);
    const size_t fsz = align_up(size_t(n) * 4, 256);
    float* den_d = reinterpret_cast<float*>(p);
    float* den_f = reinterpret_cast<float*>(p + fsz);
    float* max_d = reinterpret_cast<float*>(p + 2 * fsz);
    float* max_f = reinterpret_cast<float*>(p + 3 * fsz);
    uint8_t* sel = sel_out ? sel_out : reinterpret_cast<uint8_t*>(p + 4 * fsz);
    const _Float16* T = static_cast<const _Float16*>(text_f16);
    const dim3 ngrid(cdiv(n, 4)), block(256);
    hipLaunchKernelGGL(row_norm_kernel, ngrid, block, 0, st, X_distill, gather_distill, n, d, 1e-5f, den_d);
    hipLaunchKernelGGL(row_norm_kernel, ngrid, block, 0, st, X_fusion, gather_fusion, n, d, 1e-5f, den_f);
    OSN_LAUNCH_CHECK();
    int rc = launch_query(st, X_distill, gather_distill, nullptr, nullptr, nullptr, den_d, T, nullptr, nullptr, max_d, n, d, c);
    if (rc) return rc;
    rc = launch_query(st, X_fusion, gather_fusion, nullptr, nullptr, nullptr, den_f, T, nullptr, nullptr, max_f, n, d, c);
    if (rc) return rc;
    hipLaunchKernelGGL(ensemble_select_kernel, dim3(cdiv(n, 256)), block, 0, st, max_d, max_f, n, sel);
    OSN_LAUNCH_CHECK();
    return launch_query(st, X_distill, gather_distill, X_fusion, gather_fusion, sel, nullptr, T,
                        static_cast<_Float16*>(scores_f16), argmax, nullptr, n, d, c);
}
