// Weight gradient of the sparse convolution on gfx950, second generation: per-offset compacted pair arrays,
// split-bf16 arithmetic on v_mfma_f32_16x16x32_bf16 with LDS transpose reads.
//
// Function parity (not a port) with MinkowskiEngine's convolution backward w.r.t. the kernel (SURVEY.md
// appendix C item 5):  gW[k] = sum over the pairs (i, o) of offset k of  in[i, :]^T (x) gout[o, :].
//
// Round 1 measured what held the first kernel (fp32 MFMA, dense neighbour table) at 38 % of its roof: not the
// matrix pipe and not gather locality, but the per-chunk ballot compaction of the table and two barriers per
// 32-pair stage.  Here the pairs of every offset are compacted ONCE per map into arrays (input row, output row)
// in tile order -- MinkowskiEngine's own kernel-map format -- shared by every convolution and every step of the
// map's life, so the hot loop is gather -> split -> MFMA:
//   * a work item = (offset, range of <= quota pairs), planned on the device from the per-offset totals (a pure
//     function of the map => deterministic); partial results are summed in item order by a second launch;
//   * per 32-pair step the workgroup gathers the pairs' input rows and output-gradient rows (16-byte loads),
//     splits them into three bf16 pieces and stages them ROW-MAJOR [pair][channel] in LDS;
//   * the contraction runs over the pairs, i.e. down the LDS rows: operands are fetched with
//     ds_read_b64_tr_b16 (lane i of a 16-lane group receives channel i of four consecutive pairs; semantics
//     measured in round 1, tools/probes/probe_gfx950.hip), 2 reads per 16 x 32 operand fragment;
//   * 4 waves as 2 x 2 over the (input channel, output channel) block, MB x NB accumulator blocks of 16 x 16
//     per wave, six MFMAs per block and step (a3g1 + a2g2 + a1g3 + a2g1 + a1g2 + a1g1, fp32 accumulate).
#include "common.h"
#include "split.h"
#include <cstdlib>
#include "pairlist.h"

namespace osn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

__device__ inline int wave_incl_scan_i32(int v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    return v;
}

// pref[k][tile] = pairs of offset k in the tiles before `tile`; total[k].  One workgroup per offset.
__global__ __launch_bounds__(256) void pair_prefix_kernel(const int32_t* __restrict__ cnt, int n_tiles, int K,
                                                          int32_t* __restrict__ pref, int32_t* __restrict__ total) {
    __shared__ int wtot[4];
    __shared__ int carry_s;
    const int k = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int t0 = 0; t0 < n_tiles; t0 += 256) {
        const int t = t0 + tid;
        const int v = t < n_tiles ? cnt[int64_t(t) * K + k] : 0;
        const int incl = wave_incl_scan_i32(v, lane);
        if (lane == 63) wtot[wave] = incl;
        __syncthreads();
        int woff = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) woff += w < wave ? wtot[w] : 0;
        const int carry = carry_s;
        if (t < n_tiles) pref[int64_t(k) * n_tiles + t] = carry + woff + incl - v;
        __syncthreads();
        if (tid == 255) carry_s = carry + woff + incl;
        __syncthreads();
    }
    if (tid == 0) total[k] = carry_s;
}

// pin / pout of every offset, in (tile, local row) order; workgroup 0 also publishes poff.
__global__ __launch_bounds__(256) void pair_fill_kernel(const int32_t* __restrict__ cnt, const int2* __restrict__ lst,
                                                        const int32_t* __restrict__ out_rows,
                                                        const int32_t* __restrict__ pref, const int32_t* __restrict__ total,
                                                        int n_tiles, int K, int bm, int32_t* __restrict__ pin,
                                                        int32_t* __restrict__ pout, int32_t* __restrict__ poff) {
    __shared__ int poff_s[PL_KMAX + 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t tile = blockIdx.x;
    if (wave == 0) {
        int carry = 0;
        for (int k0 = 0; k0 < K; k0 += 64) {
            const int k = k0 + lane;
            const int v = k < K ? total[k] : 0;
            const int incl = wave_incl_scan_i32(v, lane);
            if (k < K) poff_s[k] = carry + incl - v;
            carry += __shfl(incl, 63, 64);
        }
        if (lane == 0) poff_s[K] = carry;
    }
    __syncthreads();
    if (tile == 0)
        for (int k = tid; k <= K; k += 256) poff[k] = poff_s[k];
    for (int k = wave; k < K; k += 4) {
        const int c = cnt[tile * K + k];
        const int base = poff_s[k] + pref[int64_t(k) * n_tiles + tile];
        const int2* src = lst + (tile * K + k) * bm;
        for (int p = lane; p < c; p += 64) {
            const int2 e = src[p];
            pin[base + p] = e.x;
            const int64_t r = tile * bm + e.y;
            pout[base + p] = out_rows ? out_rows[r] : int(r);
        }
    }
}

// Work items: every offset's pairs cut into ranges of `quota` pairs (quota: whole 32-pair steps, at least
// PL_MIN_QUOTA, such that at most PL_ITEMS items exist).  One workgroup.
__global__ __launch_bounds__(256) void pair_plan_kernel(const int32_t* __restrict__ total, int K, int4* __restrict__ items,
                                                        int2* __restrict__ range, int items_max) {
    __shared__ int nk[PL_KMAX];
    __shared__ int start[PL_KMAX + 1];
    __shared__ int quota_s;
    const int tid = threadIdx.x;
    if (tid == 0) {
        long long P = 0;
        for (int k = 0; k < K; ++k) P += total[k];
        long long q = (P + (items_max - K) - 1) / (items_max - K);
        q = (q + 31) / 32 * 32;
        if (q < PL_MIN_QUOTA) q = PL_MIN_QUOTA;
        quota_s = int(q);
    }
    __syncthreads();
    const int quota = quota_s;
    if (tid < K) nk[tid] = (total[tid] + quota - 1) / quota;
    __syncthreads();
    if (tid == 0) {
        int acc = 0;
        for (int k = 0; k < K; ++k) { start[k] = acc; acc += nk[k]; }
        start[K] = acc;
    }
    __syncthreads();
    if (tid < K) range[tid] = make_int2(start[tid], start[tid + 1]);
    for (int t = tid; t < PL_ITEMS; t += 256) {
        int4 it = make_int4(-1, 0, 0, 0);
        if (t < start[K]) {
            int k = 0;
            while (k + 1 < K && start[k + 1] <= t) ++k;
            const int j = t - start[k];
            it.x = k;
            it.y = j * quota;
            it.z = min(it.y + quota, total[k]);
        }
        items[t] = it;
    }
}

// ------------------------------------------------------------------------------------- the kernel
// MB / NB: 16 x 16 accumulator blocks per wave along the input / output channels; a workgroup covers
// (32 MB) x (32 NB) of gW[k].  identity (pin == nullptr): K == 1, pair p = (row p, row p), items cut by rows.
// BUFG (operands below 2^24 rows and 2 GB -- every launch of the benchmark configurations; round 5's A/B with the tile-list kernel's,
// profiles/r05_s1_knobs_ab.txt): both operands' rows come through buffer resources -- 32-bit offset
// row * (4 c) + 4 (block's first channel + quad) from one v_mad_u32_u24 -- instead of a 64-bit multiply-add and two 64-bit adds per
// quad; needs < 2^24 rows and < 2 GB per operand (the launch checks).  A padded pair reads row 0 (the split zeroes its quad as before).
template <int MB, int NB, bool BUFG = false>
__global__ __launch_bounds__(256, 2) void wgrad_tl_kernel(const float* __restrict__ rows_a, const float* __restrict__ rows_g,
                                                          const int32_t* __restrict__ idx_a, const int32_t* __restrict__ idx_g,
                                                          const int32_t* __restrict__ poff, const int4* __restrict__ items,
                                                          float* __restrict__ partial, int ca, int cg, int n_cgb,
                                                          int ident_rows, int ident_quota, unsigned a_bytes, unsigned g_bytes) {
    constexpr int CA_T = 32 * MB, CG_T = 32 * NB;
    constexpr int LDA = CA_T + 8, LDG = CG_T + 8;        // bf16 row pitch (16-byte aligned rows)
    constexpr int QA = CA_T / 4, QG = CG_T / 4;          // quads per staged row
    __shared__ __attribute__((aligned(16))) __bf16 Ap[3][32][LDA];
    __shared__ __attribute__((aligned(16))) __bf16 Gp[3][32][LDG];
    __shared__ int idxbuf[2][32];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wi = wave >> 1, wj = wave & 1;
    const int a0 = (blockIdx.x / n_cgb) * CA_T;          // first input channel of the workgroup's block
    const int g0 = (blockIdx.x % n_cgb) * CG_T;
    int k, p0, p1, base;
    int pstep = 32;                                      // pairs between this item's consecutive 32-pair steps
    if (idx_a) {
        const int4 it = items[blockIdx.y];
        k = it.x; p0 = it.y; p1 = it.z;
        if (k < 0) return;
        base = poff[k];
        // item.w = j | n << 16 (0 = the whole range): the item takes the steps j, j + n, ... of [p0, p1) -- strided items of one row
        // region walk its rows together (round 6: Z-ordered pair arrays, tools/micro_crowd.py)
        const int n_it = it.w >> 16;
        if (n_it > 1) { p0 += 32 * (it.w & 0xFFFF); pstep = 32 * n_it; }
        if (p0 >= p1) p1 = p0;                           // (an item past the range's last step: nothing to add, zeros are stored)
    } else {
        k = 0; base = 0;
        p0 = blockIdx.y * ident_quota;
        p1 = min(p0 + ident_quota, ident_rows);
        if (p0 >= p1) return;
    }
    (void)k;

    f32x4 acc[MB][NB];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // staging coordinates of this thread's quads (MB of the A rows, NB of the G rows)
    int ar[MB], ac[MB], gr[NB], gc[NB];
#pragma unroll
    for (int j = 0; j < MB; ++j) { const int idx = tid + 256 * j; ar[j] = idx / QA; ac[j] = (idx - ar[j] * QA) * 4; }
#pragma unroll
    for (int j = 0; j < NB; ++j) { const int idx = tid + 256 * j; gr[j] = idx / QG; gc[j] = (idx - gr[j] * QG) * 4; }

    const __amdgpu_buffer_rsrc_t arsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(rows_a), 0, int(BUFG ? a_bytes : 0u), 0x00020000);
    const __amdgpu_buffer_rsrc_t grsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(rows_g), 0, int(BUFG ? g_bytes : 0u), 0x00020000);
    const unsigned ca4 = unsigned(ca) * 4u, cg4 = unsigned(cg) * 4u;
    float4 pa[MB], pg[NB];
    // the (input row, output row) of pair p0 + s * 32 + (tid & 31): threads 0..31 hold the A index, 32..63 the G index
    auto load_idx = [&](int p) -> int {
        const int q = p + (tid & 31);
        if (tid >= 64) return -1;
        if (q >= p1) return -1;
        if (!idx_a) return q;
        return tid < 32 ? idx_a[base + q] : idx_g[base + q];
    };
    auto fetch = [&](int par) {
#pragma unroll
        for (int j = 0; j < MB; ++j) {
            const int r = idxbuf[0][ar[j]];
            const int ch = a0 + ac[j];
            if constexpr (BUFG) {       // (channels past the operand: inside the matrix or past its end, where the resource returns zeros)
                pa[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(
                                                       arsrc, __umul24(unsigned(max(r, 0)), ca4) + 4u * unsigned(ch), 0, 0));
                continue;
            }
            const bool ok = r >= 0 && ch < ca;
            const unsigned ru = ok ? unsigned(r) : 0u, cu = ok ? unsigned(ch) : 0u;
            pa[j] = *reinterpret_cast<const float4*>(rows_a + (uint64_t(ru) * unsigned(ca) + cu));
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int r = idxbuf[1][gr[j]];
            const int ch = g0 + gc[j];
            if constexpr (BUFG) {
                pg[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(
                                                       grsrc, __umul24(unsigned(max(r, 0)), cg4) + 4u * unsigned(ch), 0, 0));
                continue;
            }
            const bool ok = r >= 0 && ch < cg;
            const unsigned ru = ok ? unsigned(r) : 0u, cu = ok ? unsigned(ch) : 0u;
            pg[j] = *reinterpret_cast<const float4*>(rows_g + (uint64_t(ru) * unsigned(cg) + cu));
        }
        (void)par;
    };
    // three bf16 pieces per element, two elements per conversion (split.h); rows of padded pairs and channels past the
    // operand are zeroed first
    auto split4 = [&](const float4& v, bool ok, bf16x4& h1, bf16x4& h2, bf16x4& h3) {
        tl_split4(ok ? v : make_float4(0.f, 0.f, 0.f, 0.f), h1, h2, h3);
    };

    // prologue: indices of step 0 -> LDS -> rows of step 0 in flight, indices of step 1 in registers
    int ireg = load_idx(p0);
    if (tid < 64) idxbuf[tid >> 5][tid & 31] = ireg;
    __syncthreads();
    bool ok_a[MB], ok_g[NB];
    auto valid_now = [&]() {
#pragma unroll
        for (int j = 0; j < MB; ++j) ok_a[j] = idxbuf[0][ar[j]] >= 0 && a0 + ac[j] < ca;
#pragma unroll
        for (int j = 0; j < NB; ++j) ok_g[j] = idxbuf[1][gr[j]] >= 0 && g0 + gc[j] < cg;
    };
    valid_now();
    fetch(0);
    ireg = load_idx(p0 + pstep);

    for (int p = p0; p < p1; p += pstep) {
        // ---- split the rows of this step (registers)
        bf16x4 a1[MB], a2[MB], a3[MB], g1[NB], g2[NB], g3[NB];
#pragma unroll
        for (int j = 0; j < MB; ++j) split4(pa[j], ok_a[j], a1[j], a2[j], a3[j]);
#pragma unroll
        for (int j = 0; j < NB; ++j) split4(pg[j], ok_g[j], g1[j], g2[j], g3[j]);
        __syncthreads();                                   // the previous step's fragments have been read
#pragma unroll
        for (int j = 0; j < MB; ++j) {
            *reinterpret_cast<bf16x4*>(&Ap[0][ar[j]][ac[j]]) = a1[j];
            *reinterpret_cast<bf16x4*>(&Ap[1][ar[j]][ac[j]]) = a2[j];
            *reinterpret_cast<bf16x4*>(&Ap[2][ar[j]][ac[j]]) = a3[j];
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            *reinterpret_cast<bf16x4*>(&Gp[0][gr[j]][gc[j]]) = g1[j];
            *reinterpret_cast<bf16x4*>(&Gp[1][gr[j]][gc[j]]) = g2[j];
            *reinterpret_cast<bf16x4*>(&Gp[2][gr[j]][gc[j]]) = g3[j];
        }
        if (tid < 64) idxbuf[tid >> 5][tid & 31] = ireg;  // indices of the next step
        __syncthreads();
        if (p + pstep < p1) {
            valid_now();
            fetch(0);                                      // rows of the next step: in flight during the MFMAs
            ireg = load_idx(p + 2 * pstep);
        }
        // ---- fragments by transpose reads: lane i of a 16-lane group points at (pair 8 g + (i >> 2), channels
        // 4 (i & 3) .. + 3) of a 16-channel block and receives channel i of pairs 8 g .. 8 g + 3 (then + 4)
        const int li = lane & 15, lg = lane >> 4;
        auto frag = [&](const __bf16* plane, int pitch, int c16) -> bf16x8 {
            const __bf16* q = plane + (8 * lg + (li >> 2)) * pitch + c16 + 4 * (li & 3);
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (__attribute__((address_space(3))) s16x4*)(q));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (__attribute__((address_space(3))) s16x4*)(q + 4 * pitch));
            union { s16x4 h[2]; bf16x8 v; } u;
            u.h[0] = lo; u.h[1] = hi;
            return u.v;
        };
        bf16x8 fa[MB][3], fg[NB][3];
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) fa[i][pl] = frag(&Ap[pl][0][0], LDA, (wi * MB + i) * 16);
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) fg[j][pl] = frag(&Gp[pl][0][0], LDG, (wj * NB + j) * 16);
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                f32x4 t = acc[i][j];
                t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i][2], fg[j][0], t, 0, 0, 0);
                t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i][1], fg[j][1], t, 0, 0, 0);
                t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i][0], fg[j][2], t, 0, 0, 0);
                t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i][1], fg[j][0], t, 0, 0, 0);
                t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i][0], fg[j][1], t, 0, 0, 0);
                t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i][0], fg[j][0], t, 0, 0, 0);
                acc[i][j] = t;
            }
    }

    // ---- partial[item][ca][cg]: C row = 4 (lane >> 4) + r (input channel), col = lane & 15 (output channel)
    float* d = partial + int64_t(blockIdx.y) * ca * cg;
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int co = g0 + (wj * NB + j) * 16 + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ci = a0 + (wi * MB + i) * 16 + 4 * (lane >> 4) + r;
                if (ci < ca && co < cg) d[int64_t(ci) * cg + co] = acc[i][j][r];
            }
        }
}

// gW[k] = sum of the partials of offset k's items, in item order (offsets without pairs: zero).
// transpose: the kernel ran with the operand roles swapped (partials are [cout][cin]).
__global__ void wgrad_tl_reduce_kernel(const float* __restrict__ partial, const int2* __restrict__ range, int K, int cin,
                                       int cout, int transpose, int ident_items, float* __restrict__ out) {
    const int64_t per_k = int64_t(cin) * cout;
    const int64_t total = int64_t(K) * per_k;
    for (int64_t e = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; e < total; e += int64_t(gridDim.x) * blockDim.x) {
        const int k = int(e / per_k);
        const int64_t r = e - int64_t(k) * per_k;
        int64_t src = r;
        if (transpose) {
            const int ci = int(r / cout), co = int(r - int64_t(ci) * cout);
            src = int64_t(co) * cin + ci;
        }
        const int t0 = range ? range[k].x : 0, t1 = range ? range[k].y : ident_items;
        float s = 0.f;
        for (int t = t0; t < t1; ++t) s += partial[int64_t(t) * per_k + src];
        out[e] = s;
    }
}

// The same reduction for MANY weight gradients in one launch (the network executor defers every convolution's reduce to
// the end of the backward pass: 49 launches of a few microseconds -> 1): blockIdx.y = job, identical summation order.
constexpr int WG_BATCH = 32;
struct WgJobs {
    osn_wgrad_job j[WG_BATCH];
};
__global__ __launch_bounds__(256) void wgrad_tl_reduce_batch_kernel(const WgJobs jobs) {
    const osn_wgrad_job& jb = jobs.j[blockIdx.y];
    const float4* __restrict__ partial = reinterpret_cast<const float4*>(jb.partial);
    const int2* __restrict__ range = static_cast<const int2*>(jb.range);
    float4* __restrict__ out = reinterpret_cast<float4*>(jb.gW);
    const int64_t per_k4 = int64_t(jb.cin) * jb.cout / 4;               // cin % 4 == 0 and cout % 4 == 0
    const int64_t total4 = int64_t(jb.K) * per_k4;
    for (int64_t e = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; e < total4; e += int64_t(gridDim.x) * blockDim.x) {
        const int k = int(e / per_k4);
        const int64_t r = e - int64_t(k) * per_k4;
        const int t0 = range ? range[k].x : 0, t1 = range ? range[k].y : jb.ident_items;
        // item order, as osn_spconv_wgrad_tl sums: s = ((0 + p[t0]) + p[t0 + 1]) + ...; eight loads in flight per round (the loop is a
        // chain of dependent adds behind ~2 us loads: with two in flight the launch ran at 0.9 TB/s)
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        int t = t0;
        for (; t + 7 < t1; t += 8) {
            float4 v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = partial[int64_t(t + q) * per_k4 + r];
#pragma unroll
            for (int q = 0; q < 8; ++q) { s.x += v[q].x; s.y += v[q].y; s.z += v[q].z; s.w += v[q].w; }
        }
        for (; t + 1 < t1; t += 2) {
            const float4 a = partial[int64_t(t) * per_k4 + r], b = partial[int64_t(t + 1) * per_k4 + r];
            s.x = (s.x + a.x) + b.x; s.y = (s.y + a.y) + b.y; s.z = (s.z + a.z) + b.z; s.w = (s.w + a.w) + b.w;
        }
        if (t < t1) {
            const float4 a = partial[int64_t(t) * per_k4 + r];
            s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
        }
        out[e] = s;
    }
}

}  // namespace osn

using namespace osn;

extern "C" size_t osn_pair_lists_bytes(int64_t n_out, int K, int bm) {
    if (K < 1 || K > PL_KMAX) return 0;
    return pl_view(nullptr, n_out, K, bm).bytes;
}

extern "C" int osn_pair_lists_build(const void* tl, const int32_t* out_rows, int64_t n_out, int K, int bm, void* pl,
                                    osn_stream_t stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    OSN_REQUIRE(n_out >= 0 && n_out < (int64_t(1) << 31), OSN_E_ARG, "osn_pair_lists_build: n_out out of range");
    OSN_REQUIRE(K >= 1 && K <= PL_KMAX && bm >= 1, OSN_E_ARG, "osn_pair_lists_build: K=%d bm=%d", K, bm);
    OSN_REQUIRE(int64_t(K) * n_out < (int64_t(1) << 31), OSN_E_ARG, "osn_pair_lists_build: K * n_out exceeds 32-bit pair indices");
    OSN_REQUIRE(pl, OSN_E_ARG, "osn_pair_lists_build: null pointer");
    PlView v = pl_view(pl, n_out, K, bm);
    if (n_out == 0) {
        OSN_HIP(hipMemsetAsync(pl, 0, PL_OFF_ITEMS, st));
        OSN_HIP(hipMemsetAsync(v.items, 0xFF, size_t(PL_ITEMS) * 16, st));
        OSN_HIP(hipMemsetAsync(v.range, 0, size_t(PL_KMAX) * 8, st));
        return OSN_OK;
    }
    OSN_REQUIRE(tl, OSN_E_ARG, "osn_pair_lists_build: null tile lists");
    const int64_t nt = cdiv(n_out, bm);
    const int32_t* cnt = static_cast<const int32_t*>(tl);
    const int2* lst = reinterpret_cast<const int2*>(static_cast<const char*>(tl) + align_up(size_t(nt) * K * 4, 256));
    hipLaunchKernelGGL(pair_prefix_kernel, dim3(K), dim3(256), 0, st, cnt, int(nt), K, v.pref, v.total);
    // items per map: PL_ITEMS (one round of two workgroups per CU).  Fewer, longer items (less partial-sum traffic for the reduction,
    // less parallelism for the kernel) were measured per step: 256 items +0.02 .. +0.13 ms, 384 equal (profiles/r05_s1_knobs_ab.txt)
    const int items_max = PL_ITEMS;
    hipLaunchKernelGGL(pair_plan_kernel, dim3(1), dim3(256), 0, st, v.total, K, v.items, v.range, items_max);
    hipLaunchKernelGGL(pair_fill_kernel, dim3(unsigned(nt)), dim3(256), 0, st, cnt, lst, out_rows, v.pref, v.total, int(nt), K,
                       bm, v.pin, v.pout, v.poff);
    OSN_LAUNCH_CHECK();
    return OSN_OK;
}

namespace {
struct WgTlPlan {
    int mb, nb, n_ab, n_gb;     // blocks per wave, workgroup blocks along the A / G channels
};
WgTlPlan plan_wg_tl(int ca, int cg) {
    auto blocks = [](int c) { int t = (c + 31) / 32; return t > 4 ? 4 : t; };    // 32-channel units per workgroup (<= 128)
    WgTlPlan p;
    p.mb = blocks(ca);
    p.nb = blocks(cg);
    p.n_ab = int(cdiv(ca, 32 * p.mb));
    p.n_gb = int(cdiv(cg, 32 * p.nb));
    return p;
}
int ident_quota(int64_t n) {
    int64_t q = cdiv(n, PL_ITEMS);
    q = (q + 31) / 32 * 32;
    if (q < 64) q = 64;        // identity maps (1x1 convs) of the deep levels have a few hundred rows: keep >= ~10 items in flight
    return int(q);
}
}  // namespace

extern "C" size_t osn_spconv_wgrad_tl_ws_bytes(int K, int cin, int cout) {
    (void)K;
    return size_t(PL_ITEMS) * size_t(cin) * size_t(cout) * 4;
}

template <int MB>
static void launch_wg_tl_nb(int nb, dim3 grid, hipStream_t st, const float* ra, const float* rg, const int32_t* ia,
                            const int32_t* ig, const int32_t* poff, const int4* items, float* partial, int ca, int cg,
                            int n_gb, int irows, int iquota, unsigned a_bytes, unsigned g_bytes) {
    const bool bufg = a_bytes != 0u && g_bytes != 0u;
#define OSN_WGTL(NB_)                                                                                              \
    do {                                                                                                           \
        if (bufg)                                                                                                  \
            hipLaunchKernelGGL((wgrad_tl_kernel<MB, NB_, true>), grid, dim3(256), 0, st, ra, rg, ia, ig, poff, items, partial, ca, \
                               cg, n_gb, irows, iquota, a_bytes, g_bytes);                                         \
        else                                                                                                       \
            hipLaunchKernelGGL((wgrad_tl_kernel<MB, NB_, false>), grid, dim3(256), 0, st, ra, rg, ia, ig, poff, items, partial, ca, \
                               cg, n_gb, irows, iquota, 0u, 0u);                                                   \
    } while (0)
    switch (nb) {
        case 1: OSN_WGTL(1); break;
        case 2: OSN_WGTL(2); break;
        case 3: OSN_WGTL(3); break;
        default: OSN_WGTL(4); break;
    }
#undef OSN_WGTL
}

// kernel launch only: partial[item][cin][cout] + the job describing its reduction (job->gW = gW); gW itself is written only
// when the map is empty (zeros).  Returns through *launched whether a reduction is pending.
static int wgrad_tl_partial(const float* in, const float* gout, const void* pl, int swap, float* gW, int64_t n_in,
                            int64_t n_out, int K, int cin, int cout, void* ws, size_t ws_bytes, osn_wgrad_job* job,
                            int* launched, osn_stream_t stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    OSN_REQUIRE(K >= 1 && K <= PL_KMAX && cin >= 4 && (cin & 3) == 0 && cout >= 4 && (cout & 3) == 0 && gW, OSN_E_ARG,
                "osn_spconv_wgrad_tl: needs cin %% 4 == 0 and cout %% 4 == 0 (K=%d cin=%d cout=%d)", K, cin, cout);
    OSN_REQUIRE(n_in >= 0 && n_out >= 0 && n_in < (int64_t(1) << 31) && n_out < (int64_t(1) << 31), OSN_E_ARG,
                "osn_spconv_wgrad_tl: row counts out of range");
    const int64_t wtotal = int64_t(K) * cin * cout;
    *launched = 0;
    if (n_out == 0 || n_in == 0) {
        OSN_HIP(hipMemsetAsync(gW, 0, size_t(wtotal) * 4, st));
        return OSN_OK;
    }
    OSN_REQUIRE(in && gout, OSN_E_ARG, "osn_spconv_wgrad_tl: null pointer");
    OSN_REQUIRE(pl || (K == 1 && n_in == n_out), OSN_E_ARG, "osn_spconv_wgrad_tl: pair lists may be null only for K == 1");
    OSN_REQUIRE(aligned16(in) && aligned16(gout) && aligned16(gW), OSN_E_ARG, "osn_spconv_wgrad_tl: pointers must be 16-byte aligned");
    const size_t need = osn_spconv_wgrad_tl_ws_bytes(K, cin, cout);
    OSN_REQUIRE(ws && ws_bytes >= need, OSN_E_WS, "osn_spconv_wgrad_tl: workspace %zu < %zu", ws_bytes, need);
    float* partial = static_cast<float*>(ws);
    // operand A = rows contracted against the weight's FIRST channel index.  The pair arrays hold (table input
    // row, table output row); a transposed conv runs on the strided conv's arrays with the roles swapped.
    const float* ra = in;
    const float* rg = gout;
    const int ca = cin, cg = cout;
    const int32_t *ia = nullptr, *ig = nullptr, *poff = nullptr;
    const int4* items = nullptr;
    const int2* range = nullptr;
    int n_items = PL_ITEMS, irows = 0, iquota = 0;
    if (pl) {
        // the view only needs K and the capacity = K * (rows of the TABLE the lists were built from)
        const int64_t table_rows = swap ? n_in : n_out;
        PlView v = pl_view(const_cast<void*>(pl), table_rows, K, 1);
        ia = swap ? v.pout : v.pin;
        ig = swap ? v.pin : v.pout;
        poff = v.poff;
        items = v.items;
        range = v.range;
    } else {
        irows = int(n_out);
        iquota = ident_quota(n_out);
        n_items = int(cdiv(n_out, iquota));
    }
    const WgTlPlan p = plan_wg_tl(ca, cg);
    const dim3 grid(unsigned(p.n_ab * p.n_gb), unsigned(n_items));
    // operand rows through buffer resources -- rows as 24-bit factors, 32-bit byte offsets -- whenever both operands allow it
    const uint64_t ab64 = uint64_t(n_in) * uint64_t(cin) * 4u, gb64 = uint64_t(n_out) * uint64_t(cout) * 4u;
    const bool bufg = n_in < (int64_t(1) << 24) && n_out < (int64_t(1) << 24) && ab64 < (uint64_t(1) << 31) &&
                      gb64 < (uint64_t(1) << 31);
    const unsigned a_bytes = bufg ? unsigned(ab64) : 0u, g_bytes = bufg ? unsigned(gb64) : 0u;
    switch (p.mb) {
        case 1: launch_wg_tl_nb<1>(p.nb, grid, st, ra, rg, ia, ig, poff, items, partial, ca, cg, p.n_gb, irows, iquota, a_bytes, g_bytes); break;
        case 2: launch_wg_tl_nb<2>(p.nb, grid, st, ra, rg, ia, ig, poff, items, partial, ca, cg, p.n_gb, irows, iquota, a_bytes, g_bytes); break;
        case 3: launch_wg_tl_nb<3>(p.nb, grid, st, ra, rg, ia, ig, poff, items, partial, ca, cg, p.n_gb, irows, iquota, a_bytes, g_bytes); break;
        default: launch_wg_tl_nb<4>(p.nb, grid, st, ra, rg, ia, ig, poff, items, partial, ca, cg, p.n_gb, irows, iquota, a_bytes, g_bytes); break;
    }
    OSN_LAUNCH_CHECK();
    job->partial = partial;
    job->range = range;
    job->gW = gW;
    job->K = K; job->cin = cin; job->cout = cout; job->ident_items = n_items;
    *launched = 1;
    return OSN_OK;
}

extern "C" int osn_spconv_wgrad_tl(const float* in, const float* gout, const void* pl, int swap, float* gW, int64_t n_in,
                                   int64_t n_out, int K, int cin, int cout, void* ws, size_t ws_bytes,
                                   osn_stream_t stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    osn_wgrad_job job;
    int launched = 0;
    int rc = wgrad_tl_partial(in, gout, pl, swap, gW, n_in, n_out, K, cin, cout, ws, ws_bytes, &job, &launched, stream);
    if (rc || !launched) return rc;
    const int64_t wtotal = int64_t(K) * cin * cout;
    int g = int(cdiv(wtotal, 256));
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(wgrad_tl_reduce_kernel, dim3(g), dim3(256), 0, st, job.partial, static_cast<const int2*>(job.range), K, cin,
                       cout, 0, job.ident_items, gW);
    OSN_LAUNCH_CHECK();
    return OSN_OK;
}

extern "C" int osn_spconv_wgrad_tl_partial(const float* in, const float* gout, const void* pl, int swap, float* gW, int64_t n_in,
                                           int64_t n_out, int K, int cin, int cout, void* partial, size_t partial_bytes,
                                           osn_wgrad_job* job, osn_stream_t stream) {
    OSN_REQUIRE(job, OSN_E_ARG, "osn_spconv_wgrad_tl_partial: null job");
    int launched = 0;
    job->partial = nullptr; job->range = nullptr; job->gW = nullptr;
    job->K = 0; job->cin = 0; job->cout = 0; job->ident_items = 0;
    return wgrad_tl_partial(in, gout, pl, swap, gW, n_in, n_out, K, cin, cout, partial, partial_bytes, job, &launched, stream);
}

extern "C" int osn_wgrad_tl_reduce_batch(const osn_wgrad_job* jobs, int n_jobs, osn_stream_t stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    OSN_REQUIRE(n_jobs >= 0 && (n_jobs == 0 || jobs), OSN_E_ARG, "osn_wgrad_tl_reduce_batch: bad arguments");
    WgJobs b;
    int nb = 0;
    auto flush = [&]() -> int {
        if (nb == 0) return OSN_OK;
        hipLaunchKernelGGL(wgrad_tl_reduce_batch_kernel, dim3(96, unsigned(nb)), dim3(256), 0, st, b);
        OSN_LAUNCH_CHECK();
        nb = 0;
        return OSN_OK;
    };
    for (int i = 0; i < n_jobs; ++i) {
        if (!jobs[i].gW) continue;                        // nothing was launched for this job (empty map: gW already zero)
        OSN_REQUIRE(jobs[i].partial && jobs[i].K >= 1, OSN_E_ARG, "osn_wgrad_tl_reduce_batch: job %d is incomplete", i);
        b.j[nb++] = jobs[i];
        if (nb == WG_BATCH) {
            int rc = flush();
            if (rc) return rc;
        }
    }
    return flush();
}
