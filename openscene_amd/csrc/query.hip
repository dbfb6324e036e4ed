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
#include <atomic>
#include <type_traits>

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

// ---------------------------------------------------------------------------------------------------------------
// Many labels (65 .. 160: Matterport-40/80/160, run/evaluate.py:64-75): the kernel above re-stages the text matrix for
// every 128 points -- 246 KB from L2 per 393 KB of point rows -- and at 160 labels stops being bound by HBM (0.53 of the
// roof, profiles/r04_*).  Here the text matrix is loaded ONCE per workgroup and stays in registers:
//   * one persistent workgroup per CU, 8 waves.  Waves 0 .. NCW-1 ("consumers") each own 32 labels: the B fragments of
//     their 32 text rows over the WHOLE feature dimension (D / 16 k-steps x 4 VGPRs = 192 at D = 768) live in registers
//     for the kernel's life;
//   * waves NCW .. 7 ("producers") stream the point rows: a tile = 32 points, lane q of the producers holds float4 q of
//     each of the tile's rows (32 loads in flight per lane, re-issued for the next tile as soon as a register is
//     converted: a tile's worth of bytes -- 98 KB per CU -- is always in flight), fp32 -> fp16 (the reference's
//     `.half()`) into one of two LDS buffers;
//   * one workgroup barrier per tile; a consumer's tile is D / 16 MFMAs (32x32x16 f16, A fragments from LDS two k-steps
//     ahead: every consumer reads the same tile, LDS is the broadcast medium), the result rounded to fp16 ONCE and
//     written to a [32 points][labels] LDS tile;
//   * the argmax over a consumer's 32 labels runs on DPP row rotations over one unsigned key per lane (monotone fp16 bits <<
//     16 | complemented label: the lower label wins a tie, torch.max) -- no LDS crossbar (the first version reduced the MFMA
//     accumulator layout with 160 ds_bpermute per tile and wave: half the kernel's time, profiles/r05_s9_query_wide_ablation.txt);
//     one tile later a producer wave takes the maximum of the NCW keys of every point and writes label / best score, and --
//     when the score matrix is wanted -- all producer lanes copy the tile's scores out in 16-byte pieces (320-byte rows).
// Traffic per point: its row once from HBM, nothing else.  Same scores as query_kernel (fp32 accumulation over the
// feature dimension in ascending order, one fp16 rounding).
constexpr int QW_TP = 32;        // points per tile

template <int D, int NCW>
__global__ __launch_bounds__(512, 1) void query_wide_kernel(const float* __restrict__ X0, const int64_t* __restrict__ g0,
                                                            const float* __restrict__ X1, const int64_t* __restrict__ g1,
                                                            const uint8_t* __restrict__ sel, const float* __restrict__ rowdiv,
                                                            const _Float16* __restrict__ T, _Float16* __restrict__ scores,
                                                            int64_t* __restrict__ argmax, float* __restrict__ rowmax,
                                                            int64_t n, int c) {
    constexpr int NP = 8 - NCW;                  // producer waves
    constexpr int NPL = 64 * NP;                 // producer lanes
    constexpr int D4 = D / 4;                    // float4 per point row
    constexpr int KSTEPS = D / 16;
    constexpr int LD = D + 8;                    // padded LDS row (halfs): 16-byte aligned rows, conflict-free 16-byte reads
    constexpr int CP = 32 * NCW;                 // labels, padded to the consumers' 32
    constexpr int LS = CP + 8;                   // padded row of the score tile (halfs)
    static_assert(D % 32 == 0 && NCW >= 1 && NCW <= 6 && D4 <= NPL, "shape: one producer lane per float4 of a point row");
    extern __shared__ __attribute__((aligned(16))) _Float16 xs[];              // [2][QW_TP][LD] point rows, then [2][QW_TP][LS] scores
    _Float16* const sc = xs + size_t(2) * QW_TP * LD;
    __shared__ uint32_t mkey[2][6][QW_TP];           // per consumer wave and point: (orderable fp16 score << 16) | (0xFFFF - label)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t n_tiles = (n + QW_TP - 1) / QW_TP;
    const int64_t t_first = blockIdx.x, t_step = gridDim.x;
    const int64_t my_tiles = t_first < n_tiles ? (n_tiles - t_first + t_step - 1) / t_step : 0;
    auto tile_of = [&](int64_t i) { return t_first + i * t_step; };

    if (wave < NCW) {
        // ------------------------------------------------------------------------------------------ consumers
        typedef _Float16 h8 __attribute__((ext_vector_type(8)));
        h8 tf[KSTEPS];
        {
            const int row = 32 * wave + (lane & 31);
            const bool ok = row < c;
            const _Float16* src = T + int64_t(ok ? row : 0) * D + 8 * (lane >> 5);
#pragma unroll
            for (int s = 0; s < KSTEPS; ++s) {
                uint4 v = *reinterpret_cast<const uint4*>(src + 16 * s);
                if (!ok) v = make_uint4(0, 0, 0, 0);
                tf[s] = __builtin_bit_cast(h8, v);
            }
        }
        for (int64_t i = 0; i < my_tiles; ++i) {
            __syncthreads();                                             // tile i is staged in buffer i & 1
            const _Float16* a = xs + size_t(i & 1) * QW_TP * LD + (lane & 31) * LD + 8 * (lane >> 5);
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            // the A fragments of the NEXT pair of k-steps are read from LDS while the MFMAs of the current pair run (two register
            // sets; the scheduling fences keep the compiler from sinking the reads back to their first use)
            constexpr int G = 2, NG = KSTEPS / G;
            h8 aa[2][G];
#pragma unroll
            for (int j = 0; j < G; ++j) aa[0][j] = *reinterpret_cast<const h8*>(a + 16 * j);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                __builtin_amdgcn_sched_barrier(0);
                if (g + 1 < NG) {
#pragma unroll
                    for (int j = 0; j < G; ++j) aa[(g + 1) & 1][j] = *reinterpret_cast<const h8*>(a + 16 * ((g + 1) * G + j));
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < G; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(aa[g & 1][j], tf[g * G + j], acc, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            // fp16 rounding (the reference's half GEMM output).  Row argmax over this wave's 32 labels WITHOUT the LDS crossbar: a
            // score and its label become one unsigned key -- the fp16 bits made monotone, the label complemented so that the
            // LOWER label wins a tie (torch.max) -- and the maximum over the 32 lanes that hold a point's labels is four
            // row-rotation DPP steps per 16-lane row plus one row broadcast.  (The first version shuffled (value, index) pairs
            // through ds_bpermute: 160 per tile and wave, half the kernel's time.)
            const int col = 32 * wave + (lane & 31);
            _Float16* dst = sc + size_t(i & 1) * QW_TP * LS + col;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int lrow = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const _Float16 hv = (_Float16)acc[r];
                if (scores) dst[lrow * LS] = hv;
                // torch.max compares VALUES: -0.0 == +0.0 (the lower label wins the tie), so -0 is made +0 before the bits are ordered.
                // NaN: the bit order puts a positive NaN above every number and a negative one below -- torch.max would return the
                // first NaN; scores here are cosine products of finite features, a NaN score means NaN input rows (documented in
                // labels of such points are unspecified in both kernels).
                uint32_t hb = uint32_t(__builtin_bit_cast(uint16_t, hv));
                hb = hb == 0x8000u ? 0u : hb;
                const uint32_t ord = (hb & 0x8000u) ? (~hb & 0xFFFFu) : (hb | 0x8000u);
                uint32_t key = col < c ? ((ord << 16) | (0xFFFFu - uint32_t(col))) : 0u;
#define QW_ROR(K_) key = max(key, uint32_t(__builtin_amdgcn_update_dpp(0, int(key), 0x120 + (K_), 0xF, 0xF, false)));
                QW_ROR(8) QW_ROR(4) QW_ROR(2) QW_ROR(1)
#undef QW_ROR
                // lane 15 of rows 0 / 2 into rows 1 / 3 (row_bcast15, rows 1 and 3 written): lanes 16 .. 31 and 48 .. 63 then hold the
                // maximum of their 32-lane half
                const uint32_t up = uint32_t(__builtin_amdgcn_update_dpp(int(key), int(key), 0x142, 0xA, 0xF, false));
                key = max(key, up);
                if ((lane & 31) == 31) mkey[i & 1][wave][lrow] = key;
            }
        }
        __syncthreads();                                                 // (the producers' last merge)
    } else {
        // ------------------------------------------------------------------------------------------ producers
        const int pl = tid - 64 * NCW;                                   // producer lane 0 .. NPL - 1
        // RING rows in flight per lane.  Row r of the workgroup's row stream lives in v[r % RING]; the loop below is unrolled over
        // three tiles so that a ring of 1.5 tiles (48) has static register indices -- measured: 48 is no faster than 32 (0.329 vs
        // 0.320 ms for 500 k points x 160 labels; with the MFMAs and the epilogue removed both run 0.265 ms = 5.8 TB/s, the pace of
        // the 3 KB row gather itself, profiles/r05_s9_query_wide_ablation.txt), so one tile it is.
        constexpr int RING = 32;
        float4 v[RING];
        // lane j < 32 of every producer wave holds point j's source-row byte offset (relative to X0) and divisor, for the tile being
        // staged (den only) and the two tiles after it (requested ahead: a row load never waits for an index load)
        int64_t off_a = 0, off_b = 0;            // offsets of tiles t + 1, t + 2 while tile t is staged
        float den_t = 1.f, den_a = 1.f, den_b = 1.f;
        auto offsets = [&](int64_t i, int64_t& off, float& den) {
            int64_t p = tile_of(i < my_tiles ? i : my_tiles - 1) * QW_TP + (lane & 31);
            if (p >= n) p = n - 1;
            if (sel && sel[p])
                off = int64_t(reinterpret_cast<uintptr_t>(X1) - reinterpret_cast<uintptr_t>(X0)) + (g1 ? g1[p] : p) * int64_t(D) * 4;
            else
                off = (g0 ? g0[p] : p) * int64_t(D) * 4;
            den = rowdiv ? rowdiv[p] : 1.f;
        };
        const int qb = (pl < D4 ? pl : 0) * 16;
        // point k's row: a wave-uniform base (the offset travels from lane k through an SGPR) + this lane's 16 bytes; lanes past
        // the row width re-read its first float4 (never staged)
        auto load_row = [&](int k, int64_t off_l) -> float4 {
            const unsigned lo = unsigned(uint64_t(off_l)), hi = unsigned(uint64_t(off_l) >> 32);
            const uint64_t off = (uint64_t(unsigned(__builtin_amdgcn_readlane(int(hi), k))) << 32) | unsigned(__builtin_amdgcn_readlane(int(lo), k));
            return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(X0) + off + qb);
        };
        // tile t (phase = t % 3, static): its 32 rows -> fp16 -> buffer t & 1; every freed register immediately requests the row
        // RING places further down the stream (of tile t + 1, or t + 2 with a longer ring): the queue of loads never drains
        auto stage_and_issue = [&](int64_t t, auto phase_c) {
            constexpr int PH = decltype(phase_c)::value;
            _Float16* dst = xs + size_t(t & 1) * QW_TP * LD + 4 * pl;
#pragma unroll
            for (int k = 0; k < QW_TP; ++k) {
                const int reg = (QW_TP * PH + k) % RING;
                float4 x = v[reg];
                if (rowdiv) {
                    const float den = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, den_t), k));
                    x.x /= den; x.y /= den; x.z /= den; x.w /= den;
                }
                half4 h;
                h[0] = (_Float16)x.x; h[1] = (_Float16)x.y; h[2] = (_Float16)x.z; h[3] = (_Float16)x.w;
                if (pl < D4) *reinterpret_cast<half4*>(dst + k * LD) = h;
                v[reg] = (k + RING) / QW_TP == 1 ? load_row((k + RING) % QW_TP, off_a) : load_row((k + RING) % QW_TP, off_b);   // (past the end: the last tile's rows again)
            }
            den_t = den_a;
            den_a = den_b;
            off_a = off_b;
            offsets(t + 3, off_b, den_b);
        };
        // the consumers' score tile of tile i -> the outputs
        auto merge = [&](int64_t i) {
            const _Float16* src = sc + size_t(i & 1) * QW_TP * LS;
            const int64_t row0 = tile_of(i) * QW_TP;
            if (scores) {                                                // rows of c fp16: 16-byte chunks where the row width allows it
                if ((c & 7) == 0) {
                    const int cpr = c >> 3;
                    for (int e = pl; e < QW_TP * cpr; e += NPL) {
                        const int pt = e / cpr, ch = e - pt * cpr;
                        if (row0 + pt < n)
                            *reinterpret_cast<uint4*>(scores + (row0 + pt) * c + 8 * ch) = *reinterpret_cast<const uint4*>(src + pt * LS + 8 * ch);
                    }
                } else {
                    for (int e = pl; e < QW_TP * c; e += NPL) {
                        const int pt = e / c, col = e - pt * c;
                        if (row0 + pt < n) scores[(row0 + pt) * c + col] = src[pt * LS + col];
                    }
                }
            }
            if (pl < QW_TP) {                                            // the consumers' keys of point pl: the largest wins
                uint32_t key = 0u;
#pragma unroll
                for (int w = 0; w < NCW; ++w) key = max(key, mkey[i & 1][w][pl]);
                const int64_t p = row0 + pl;
                if (p < n) {
                    if (argmax) argmax[p] = key ? int64_t(0xFFFFu - (key & 0xFFFFu)) : 0;
                    if (rowmax) {
                        const uint32_t ord = key >> 16;
                        const uint16_t hb = uint16_t((ord & 0x8000u) ? (ord & 0x7FFFu) : (~ord & 0xFFFFu));
                        rowmax[p] = key ? float(__builtin_bit_cast(_Float16, hb)) : -INFINITY;
                    }
                }
            }
        };
        typedef std::integral_constant<int, 0> P0;
        typedef std::integral_constant<int, 1> P1;
        typedef std::integral_constant<int, 2> P2;
        if (my_tiles > 0) {
            int64_t off0;
            offsets(0, off0, den_t);
            offsets(1, off_a, den_a);
            offsets(2, off_b, den_b);
#pragma unroll
            for (int k = 0; k < QW_TP; ++k) v[k] = load_row(k, off0);
            if constexpr (RING > QW_TP) {
#pragma unroll
                for (int k = 0; k < RING - QW_TP; ++k) v[(QW_TP + k) % RING] = load_row(k, off_a);
            }
            stage_and_issue(0, P0());
        }
        for (int64_t i = 0; i < my_tiles; i += 3) {                      // tile i + u is multiplied while tile i + u + 1 is staged
            __syncthreads();
            if (i + 1 < my_tiles) stage_and_issue(i + 1, P1());
            if (i > 0) merge(i - 1);
            if (i + 1 >= my_tiles) break;
            __syncthreads();
            if (i + 2 < my_tiles) stage_and_issue(i + 2, P2());
            merge(i);
            if (i + 2 >= my_tiles) break;
            __syncthreads();
            if (i + 3 < my_tiles) stage_and_issue(i + 3, P0());
            merge(i + 1);
        }
        __syncthreads();
        if (my_tiles > 0) merge(my_tiles - 1);
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
    // 65 .. 160 labels at the CLIP widths: the text matrix stays in registers, one persistent workgroup per CU
    if (c > 64 && c <= 160 && (d == 768 || d == 512) && n >= 4096) {
        // compute units of the current device (one persistent workgroup each), queried once per device
        static std::atomic<int> cu_cache[64];
        int dev_id = 0;
        OSN_HIP(hipGetDevice(&dev_id));
        int cus = (dev_id >= 0 && dev_id < 64) ? cu_cache[dev_id].load(std::memory_order_relaxed) : 0;
        if (cus <= 0) {
            OSN_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev_id));
            if (cus <= 0) cus = 256;
            if (dev_id >= 0 && dev_id < 64) cu_cache[dev_id].store(cus, std::memory_order_relaxed);
        }
        const int64_t tiles = cdiv(n, QW_TP);
        const unsigned gx = unsigned(tiles < cus ? tiles : cus);
        const int ncw = int(cdiv(c, 32));
        const size_t lds = size_t(2) * QW_TP * size_t(d + 8) * 2 + size_t(2) * QW_TP * size_t(32 * ncw + 8) * 2;
        // the ~120 KB dynamic-LDS opt-in: once per (instance, device), like the tile-list kernel's; if the runtime refuses it (a part
        // with less LDS per workgroup) the launch falls through to query_kernel below, which takes every shape
        bool wide_ok = true;
        const bool dev_slot_ok = dev_id >= 0 && dev_id < 64;
#define OSN_QW(D_, W_)                                                                                                      \
    do {                                                                                                                   \
        auto kern = query_wide_kernel<D_, W_>;                                                                             \
        static std::atomic<signed char> attr_state[64];          /* 0 unknown, 1 set, -1 refused */                        \
        signed char stt = dev_slot_ok ? attr_state[dev_id].load(std::memory_order_relaxed) : 0;                            \
        if (stt == 0) {                                                                                                    \
            stt = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,     \
                                      int(size_t(2) * QW_TP * size_t(D_ + 8) * 2 + size_t(2) * QW_TP * size_t(32 * W_ + 8) * 2)) == hipSuccess ? 1 : -1; \
            if (stt < 0) (void)hipGetLastError();                /* the refusal is handled here: clear the sticky error */ \
            if (dev_slot_ok) attr_state[dev_id].store(stt, std::memory_order_relaxed);                                     \
        }                                                                                                                  \
        if (stt < 0) { wide_ok = false; break; }                                                                           \
        hipLaunchKernelGGL(kern, dim3(gx), dim3(512), lds, st, X0, g0, X1, g1, sel, rowdiv, T, scores, argmax, rowmax, n, c); \
    } while (0)
        if (d == 768) {
            if (ncw == 3) OSN_QW(768, 3);
            else if (ncw == 4) OSN_QW(768, 4);
            else OSN_QW(768, 5);
        } else {
            if (ncw == 3) OSN_QW(512, 3);
            else if (ncw == 4) OSN_QW(512, 4);
            else OSN_QW(512, 5);
        }
#undef OSN_QW
        if (wide_ok) {
            OSN_LAUNCH_CHECK();
            return OSN_OK;
        }
    }
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
