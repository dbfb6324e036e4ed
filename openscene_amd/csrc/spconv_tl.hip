// Sparse 3-D convolution on gfx950, second generation: per-tile COMPACTED pair lists, weights held in
// registers, output tile accumulated in LDS.
//
// Function parity (not a port) with MinkowskiEngine's convolution kernels (SURVEY.md section 2.1,
// appendix C item 5):  out[o] = sum_k in[nbr[k,o]] @ W[k].
//
// Why this shape (measured in round 1): on a 2 cm indoor scene a voxel has 5.3 of its 27
// neighbours, so an output-stationary register tile pads every 32-row group to ~2x the real work, and it
// re-stages the weight chunk of every (tile, offset, channel chunk) through LDS -- the kernel sat at 27 % MFMA
// pipe utilisation with more bytes of weights than of activations entering each CU.  Here:
//   * the kernel map of a tile of `bm` output rows is stored per offset as a dense list of (input row, local
//     output row) pairs (osn_tile_lists_build), so the MFMAs only ever see real pairs (padded to 32 per
//     (tile, offset): 79-91 % efficiency at bm = 128 on S100k instead of 50 %);
//   * a workgroup = 4 waves = 2 pair-halves x 2 column groups; each wave keeps the split-bf16 B fragments of
//     W[k] for ITS output columns in registers for the whole (tile, offset), so weights enter the CU once per
//     (tile, offset) and never touch LDS;
//   * gathered rows are split into three bf16 pieces ONCE per workgroup while they are staged to LDS (all
//     waves read ready-made MFMA A fragments with ds_read_b128, no per-wave conversion);
//   * the 16 x 16 result blocks are added into an fp32 output tile in LDS at the pairs' local output rows.
//     Within one offset an output row occurs at most once, offsets are walked in ascending order with
//     workgroup barriers in between, and a workgroup owns its rows exclusively => no atomics are needed and
//     the summation order is fixed: bitwise reproducible;
//   * the epilogue streams the tile to HBM through the `out_rows` permutation (features stay in the caller's
//     row order) and can emit the per-tile column sums / sums of squares the following batch norm needs.
// Arithmetic: "bf16x6" -- x = x1 + x2 + x3 (bf16 pieces), a*b ~= a3b1 + a2b2 + a1b3 + a2b1 + a1b2 + a1b1
// accumulated in fp32 on v_mfma_f32_16x16x32_bf16 (dropped terms <= 2^-24 relative): fp32-class accuracy.
#include "common.h"

namespace osn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int TL_BMAX = 104;      // rows per tile (LDS budget of the widest instance at 2 workgroups per CU)
constexpr int TL_LCAP = 1536;     // packed list entries resident in LDS per batch of offsets
constexpr int TL_KMAX = 128;      // kernel offsets a list-mode launch can take (5^3 = 125)

// ------------------------------------------------------------------------------------------- lists
// cnt[tile][K] pairs per (tile, offset); lst[tile][K][bm] = (input row, local output row), valid pairs first
// in ascending local row.  One wave per offset (k = wave, wave + 4, ...): ballot compaction, no barriers.
__global__ __launch_bounds__(256) void tile_lists_kernel(const int32_t* __restrict__ nbr, int64_t n_out, int K, int bm,
                                                         int32_t* __restrict__ cnt, int2* __restrict__ lst) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t tile = blockIdx.x;
    const int64_t row0 = tile * bm;
    const int rows = int(min(int64_t(bm), n_out - row0));
    for (int k = wave; k < K; k += 4) {
        int base = 0;
        int2* dst = lst + (tile * K + k) * bm;
        for (int j0 = 0; j0 < rows; j0 += 64) {
            const int j = j0 + lane;
            const int v = j < rows ? nbr[int64_t(k) * n_out + row0 + j] : -1;
            const bool ok = v >= 0;
            const unsigned long long m = __ballot(ok);
            if (ok) dst[base + __popcll(m & ((1ull << lane) - 1ull))] = make_int2(v, j);
            base += __popcll(m);
        }
        if (lane == 0) cnt[tile * K + k] = base;
    }
}

// ------------------------------------------------------------------------------------- weight prep
// MFMA-ready image of a weight: Wp[plane][k][s][cb][lane][8] bf16, one 1 KB block per (k, 32-deep k-step s,
// 16-column block cb, plane); lane l of a wave holds the B fragment of v_mfma_f32_16x16x32_bf16:
//   element e of lane l = piece `plane` of  B[c = 32 s + 8 (l >> 4) + e][n = 16 cb + (l & 15)]
// with B[c][n] = W[k][c][n] (forward) or W[flip ? K-1-k : k][n][c] (input gradient: contraction over the
// conv's OUTPUT channels).  Channels / columns beyond the real ones are zero.
__device__ __forceinline__ void weight_prep_tl_one(const float* __restrict__ W, int K, int cin, int cout, int flip,
                                                    int for_dgrad, int64_t e, __bf16* __restrict__ Wp) {
    const int nc = for_dgrad ? cout : cin;     // contraction length
    const int nn = for_dgrad ? cin : cout;     // columns of B
    const int ns = (nc + 31) >> 5, ncb = (nn + 15) >> 4;
    const int64_t per_plane = int64_t(K) * ns * ncb * 512;
    const int el = int(e & 7);
    const int lane = int((e >> 3) & 63);
    int64_t blk = e >> 9;
    const int cb = int(blk % ncb); blk /= ncb;
    const int s = int(blk % ns);
    const int k = int(blk / ns);
    const int c = 32 * s + 8 * (lane >> 4) + el;
    const int n = 16 * cb + (lane & 15);
    float v = 0.f;
    if (c < nc && n < nn) {
        const int ks = flip ? K - 1 - k : k;
        v = for_dgrad ? W[(int64_t(ks) * cin + n) * cout + c] : W[(int64_t(ks) * cin + c) * cout + n];
    }
    const __bf16 h1 = (__bf16)v;
    const float r1 = v - (float)h1;
    const __bf16 h2 = (__bf16)r1;
    const float r2 = r1 - (float)h2;
    Wp[e] = h1;
    Wp[per_plane + e] = h2;
    Wp[2 * per_plane + e] = (__bf16)r2;
}

__global__ void weight_prep_tl_kernel(const float* __restrict__ W, int K, int cin, int cout, int flip_b, int64_t per_plane_f,
                                      int64_t per_plane_b, __bf16* __restrict__ Wf, __bf16* __restrict__ Wb) {
    const int64_t total = per_plane_f + per_plane_b;
    for (int64_t e = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; e < total; e += int64_t(gridDim.x) * blockDim.x) {
        if (e < per_plane_f) weight_prep_tl_one(W, K, cin, cout, 0, 0, e, Wf);
        else weight_prep_tl_one(W, K, cin, cout, flip_b, 1, e - per_plane_f, Wb);
    }
}

// ------------------------------------------------------------------------------------------ conv
// NB16: 16-column blocks per wave (workgroup = 2 * NB16 * 16 output columns); KS: 32-deep k-steps per channel
// chunk (B fragments of a chunk live in NB16 * KS * 12 VGPRs).
//
// Latency structure (measured: the first version paid three dependent global-memory latencies per
// (tile, offset) -- list, gather, weights -- and ran at the speed of the round-1 kernel): the pair lists of a
// whole BATCH of the tile's offsets are brought into LDS with one round of loads, and the (offset, chunk,
// 32-pair iteration) steps of the batch then run as ONE flat software pipeline whose gathers are issued two
// steps ahead, across offset boundaries; only the B-fragment load of a new (offset, chunk) is exposed, and it
// hits L2.
struct TlIter {
    int a;        // offset index in klist
    int s0;       // first 32-deep k-step of the channel chunk
    int g;        // 32-pair iteration within the offset
    int niter;    // iterations of offset a
};

template <int NB16, int KS>
__global__ __launch_bounds__(256, 2) void spconv_tl_kernel(const float* __restrict__ in, const bf16x8* __restrict__ Wp,
                                                           const int32_t* __restrict__ cnt, const int2* __restrict__ lst,
                                                           const int32_t* __restrict__ out_rows, float* __restrict__ out,
                                                           double* __restrict__ bn_partial, int n_out, int K, int cin,
                                                           int cout, int bm, int ns, int ncb) {
    constexpr int CW = 2 * NB16 * 16;         // output columns of the workgroup
    constexpr int S = CW + 4;                 // fp32 row stride of the output tile
    constexpr int CK = 32 * KS;               // input channels per chunk
    constexpr int LDA = CK + 8;               // bf16 row stride of a staged plane (16-byte aligned rows)
    constexpr int QR = CK / 4;                // 4-channel quads per staged row
    constexpr int NQ = KS;                    // quads per thread per iteration (32 rows * QR / 256)
    constexpr int NL = TL_LCAP / 256;         // list entries per thread per batch
    __shared__ __attribute__((aligned(16))) float otile[(TL_BMAX + 1) * S];     // + one dump row for padded pairs
    __shared__ __attribute__((aligned(16))) __bf16 stage[3][32][LDA];
    __shared__ uint32_t plist[TL_LCAP];       // (local output row << 24) | input row, 32-padded per offset
    __shared__ int klist[TL_KMAX];
    __shared__ int kcnt[TL_KMAX];
    __shared__ int lstart[TL_KMAX + 1];       // first plist slot of each offset of the current batch
    __shared__ unsigned char gowner[TL_LCAP / 32];
    __shared__ int nact_s, bend_s;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);         // wave-uniform => scalar registers
    const int ph = wave & 1, cg = wave >> 1;
    const int tile = blockIdx.x;
    const int row0 = tile * bm;
    const int rows = min(bm, n_out - row0);
    const int col0 = blockIdx.y * CW;
    const int cb0 = blockIdx.y * (2 * NB16) + cg * NB16;      // this wave's first 16-column block

    for (int i = tid; i < (TL_BMAX + 1) * S / 4; i += 256)
        reinterpret_cast<float4*>(otile)[i] = make_float4(0.f, 0.f, 0.f, 0.f);

    // ---- active offsets of the tile, ascending (=> fixed summation order)
    if (cnt) {
        if (wave == 0) {
            int n = 0;
            for (int k0 = 0; k0 < K; k0 += 64) {
                const int k = k0 + lane;
                const int c = k < K ? cnt[int64_t(tile) * K + k] : 0;
                const unsigned long long m = __ballot(c > 0);
                if (c > 0) {
                    const int pos = n + __popcll(m & ((1ull << lane) - 1ull));
                    klist[pos] = k;
                    kcnt[pos] = c;
                }
                n += __popcll(m);
            }
            if (lane == 0) nact_s = n;
        }
    } else if (tid == 0) {                  // identity map (K == 1): the tile's own rows, in order
        klist[0] = 0;
        kcnt[0] = rows;
        nact_s = 1;
    }
    __syncthreads();
    const int nact = __builtin_amdgcn_readfirstlane(nact_s);

    // loop-invariant staging coordinates of this thread's quads
    int q_row[NQ], q_col[NQ];
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
        const int idx = tid + 256 * j;
        q_row[j] = idx / QR;
        q_col[j] = (idx - q_row[j] * QR) * 4;
    }
    const int nchunk = (ns + KS - 1) / KS;

    bf16x8 B[KS][NB16][3];
    float4 P0[NQ], P1[NQ];

    int a0 = 0;
    while (a0 < nact) {
        // ---- batch [a0, a1): as many consecutive offsets as fit the LDS list buffer
        if (tid == 0) {
            int tot = 0, a = a0;
            while (a < nact) {
                const int np = (kcnt[a] + 31) & ~31;
                if (tot + np > TL_LCAP) break;             // a single offset (<= 128 entries) always fits
                lstart[a] = tot;
                tot += np;
                ++a;
            }
            lstart[a] = tot;
            bend_s = a;
        }
        __syncthreads();                                   // (also: the previous batch is done with plist / lstart)
        const int a1 = __builtin_amdgcn_readfirstlane(bend_s);
        const int E = __builtin_amdgcn_readfirstlane(lstart[a1]);
        if (tid < a1 - a0) {
            const int a = a0 + tid;
            for (int g = lstart[a] >> 5; g < (lstart[a + 1] >> 5); ++g) gowner[g] = (unsigned char)a;
        }
        __syncthreads();
        if (cnt) {
            // all loads first (unconditional, clamped addresses), then the selects and the LDS writes: one
            // exposed memory latency per batch instead of one per entry
            int2 x[NL];
            bool okv[NL];
#pragma unroll
            for (int j = 0; j < NL; ++j) {
                const int e = tid + 256 * j;
                const int ec = e < E ? e : 0;
                const int a = gowner[ec >> 5];
                const int p = ec - lstart[a];
                okv[j] = e < E && p < kcnt[a];
                x[j] = lst[(int64_t(tile) * K + klist[a]) * bm + (okv[j] ? p : 0)];
            }
#pragma unroll
            for (int j = 0; j < NL; ++j) {
                // padded pair: input row 0 (valid address), dump row
                const uint32_t v = okv[j] ? ((uint32_t(x[j].y) << 24) | uint32_t(x[j].x)) : (uint32_t(TL_BMAX) << 24);
                if (tid + 256 * j < E) plist[tid + 256 * j] = v;
            }
        } else {
            for (int e = tid; e < E; e += 256)
                plist[e] = e < rows ? ((uint32_t(e) << 24) | uint32_t(row0 + e)) : (uint32_t(TL_BMAX) << 24);
        }
        __syncthreads();

        // ---- flat pipeline over the (offset, chunk, iteration) steps of the batch
        auto first = [&](int a) {
            TlIter it;
            it.a = a; it.s0 = 0; it.g = 0;
            it.niter = a < a1 ? (__builtin_amdgcn_readfirstlane(kcnt[a]) + 31) >> 5 : 0;
            return it;
        };
        auto advance = [&](TlIter& it) {
            if (++it.g == it.niter) {
                it.g = 0;
                it.s0 += KS;
                if (it.s0 >= ns) {
                    it.s0 = 0;
                    ++it.a;
                    it.niter = it.a < a1 ? (__builtin_amdgcn_readfirstlane(kcnt[it.a]) + 31) >> 5 : 0;
                }
            }
        };
        // gather of one step: 32 list entries x CK channels, one 16-byte load per quad (unconditional, clamped
        // address; masked at conversion time)
        auto fetch = [&](const TlIter& it, float4 (&P)[NQ]) {
            const int base = __builtin_amdgcn_readfirstlane(lstart[it.a]) + 32 * it.g;
#pragma unroll
            for (int j = 0; j < NQ; ++j) {
                const unsigned row = plist[base + q_row[j]] & 0xFFFFFFu;
                const int ch = 32 * it.s0 + q_col[j];
                const unsigned cu = ch < cin ? unsigned(ch) : 0u;
                P[j] = *reinterpret_cast<const float4*>(in + (uint64_t(row) * unsigned(cin) + cu));
            }
        };
        auto step = [&](const TlIter& it, float4 (&P)[NQ], const TlIter& nf) {
            // ---- split the fetched quads into three bf16 pieces (registers)
            bf16x4 p1[NQ], p2[NQ], p3[NQ];
#pragma unroll
            for (int j = 0; j < NQ; ++j) {
                const bool ok = 32 * it.s0 + q_col[j] < cin;
                const float x[4] = {P[j].x, P[j].y, P[j].z, P[j].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = ok ? x[e] : 0.f;
                    const __bf16 h1 = (__bf16)v;
                    const float r1 = v - (float)h1;
                    const __bf16 h2 = (__bf16)r1;
                    const float r2 = r1 - (float)h2;
                    p1[j][e] = h1; p2[j][e] = h2; p3[j][e] = (__bf16)r2;
                }
            }
            // ---- new (offset, chunk): B fragments of this wave's columns, one coalesced 1 KB load each
            if (it.g == 0) {
                const int k = __builtin_amdgcn_readfirstlane(klist[it.a]);        // wave-uniform: scalar address math
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                    for (int nb = 0; nb < NB16; ++nb) {
                        // k-steps past the last chunk are never multiplied and column blocks past the weight feed
                        // output columns that are never stored: load block 0 instead (valid memory), and do NOT touch
                        // the loaded value here -- any use would make the compiler wait for the loads before the barrier
                        const bool on = it.s0 + ks < ns && cb0 + nb < ncb;
                        const unsigned sb = on ? unsigned(it.s0 + ks) : 0u, cb = on ? unsigned(cb0 + nb) : 0u;
#pragma unroll
                        for (int pl = 0; pl < 3; ++pl) {
                            const unsigned blk = ((unsigned(pl * K + k) * unsigned(ns) + sb) * unsigned(ncb) + cb);   // 1 KB blocks
                            B[ks][nb][pl] = (Wp + (size_t(blk) << 6))[lane];
                        }
                    }
            }
            __syncthreads();                               // every wave is done reading the previous stage
#pragma unroll
            for (int j = 0; j < NQ; ++j) {
                *reinterpret_cast<bf16x4*>(&stage[0][q_row[j]][q_col[j]]) = p1[j];
                *reinterpret_cast<bf16x4*>(&stage[1][q_row[j]][q_col[j]]) = p2[j];
                *reinterpret_cast<bf16x4*>(&stage[2][q_row[j]][q_col[j]]) = p3[j];
            }
            if (nf.a < a1) fetch(nf, P);                   // two steps ahead; in flight during the MFMAs below
            __syncthreads();                               // stage ready
            // ---- 16 pairs x NB16 * 16 columns per wave
            f32x4 acc[NB16];
#pragma unroll
            for (int nb = 0; nb < NB16; ++nb) acc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
            const int arow = ph * 16 + (lane & 15);
            const int akq = 8 * (lane >> 4);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if (it.s0 + ks < ns) {
                    const bf16x8 a1f = *reinterpret_cast<const bf16x8*>(&stage[0][arow][ks * 32 + akq]);
                    const bf16x8 a2f = *reinterpret_cast<const bf16x8*>(&stage[1][arow][ks * 32 + akq]);
                    const bf16x8 a3f = *reinterpret_cast<const bf16x8*>(&stage[2][arow][ks * 32 + akq]);
#pragma unroll
                    for (int nb = 0; nb < NB16; ++nb) {
                        f32x4 t = acc[nb];
                        t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3f, B[ks][nb][0], t, 0, 0, 0);
                        t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2f, B[ks][nb][1], t, 0, 0, 0);
                        t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1f, B[ks][nb][2], t, 0, 0, 0);
                        t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2f, B[ks][nb][0], t, 0, 0, 0);
                        t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1f, B[ks][nb][1], t, 0, 0, 0);
                        t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1f, B[ks][nb][0], t, 0, 0, 0);
                        acc[nb] = t;
                    }
                }
            }
            // ---- add the result block into the output tile: C row = 4 (lane >> 4) + r, col = lane & 15
            // (all reads first, then all writes: the 4 * NB16 cells of a lane are distinct -- one output row per
            // pair within an offset -- but the compiler cannot know, and a read-add-write chain per cell would
            // serialise 4 * NB16 LDS round trips)
            const int pbase = __builtin_amdgcn_readfirstlane(lstart[it.a]) + 32 * it.g + ph * 16 + 4 * (lane >> 4);
            int orow[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) orow[r] = int(plist[pbase + r] >> 24);
            float cur[NB16][4];
#pragma unroll
            for (int nb = 0; nb < NB16; ++nb)
#pragma unroll
                for (int r = 0; r < 4; ++r) cur[nb][r] = otile[orow[r] * S + (cg * NB16 + nb) * 16 + (lane & 15)];
#pragma unroll
            for (int nb = 0; nb < NB16; ++nb)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    otile[orow[r] * S + (cg * NB16 + nb) * 16 + (lane & 15)] = cur[nb][r] + acc[nb][r];
        };

        TlIter cur = first(a0);
        TlIter n1 = cur;
        advance(n1);
        TlIter n2 = n1;
        if (n1.a < a1) advance(n2);
        fetch(cur, P0);
        if (n1.a < a1) fetch(n1, P1);
        while (cur.a < a1) {
            step(cur, P0, n2);
            cur = n1; n1 = n2;
            if (n2.a < a1) advance(n2);
            if (cur.a >= a1) break;
            step(cur, P1, n2);
            cur = n1; n1 = n2;
            if (n2.a < a1) advance(n2);
        }
        a0 = a1;
    }
    __syncthreads();

    // ---- epilogue: tile rows -> out[out_rows[row]] (16-byte stores), optional batch-norm partial sums
    constexpr int V = CW / 4;
    for (int idx = tid; idx < rows * V; idx += 256) {
        const int j = idx / V, c4 = idx - j * V;
        const int col = col0 + 4 * c4;
        if (col < cout) {
            const int64_t orow = out_rows ? out_rows[row0 + j] : row0 + j;
            *reinterpret_cast<float4*>(out + orow * cout + col) = *reinterpret_cast<const float4*>(&otile[j * S + 4 * c4]);
        }
    }
    if (bn_partial && tid < CW && col0 + tid < cout) {
        double s1 = 0, s2 = 0;
        for (int j = 0; j < rows; ++j) {
            const double v = otile[j * S + tid];
            s1 += v;
            s2 += v * v;
        }
        bn_partial[(int64_t(tile) * 2 + 0) * cout + col0 + tid] = s1;
        bn_partial[(int64_t(tile) * 2 + 1) * cout + col0 + tid] = s2;
    }
}

struct TlPlan {
    int nb16, ks, cw, gy;
};

static TlPlan plan_tl(int cout) {
    // column group width: the candidate with the least padding, the wider one on ties
    TlPlan best = {0, 0, 0, 0};
    int best_pad = 1 << 30;
    static const int widths[4] = {128, 96, 64, 32};
    static const int kss[4] = {2, 3, 4, 4};
    for (int i = 0; i < 4; ++i) {
        const int cw = widths[i];
        const int gy = int(cdiv(cout, cw));
        const int pad = gy * cw - cout;
        if (pad < best_pad) {
            best_pad = pad;
            best = {cw / 32, kss[i], cw, gy};
        }
    }
    return best;
}

}  // namespace osn

using namespace osn;

extern "C" int osn_tile_rows(int64_t n_out) {
    // rows per tile: whole rounds of 2 workgroups on each of the 256 CUs, never more than TL_BMAX
    if (n_out <= 0) return 32;
    const int64_t slots = 512;
    const int64_t rounds = cdiv(n_out, slots * TL_BMAX);
    int64_t bm = cdiv(n_out, slots * rounds);
    bm = (bm + 3) / 4 * 4;
    if (bm < 32) bm = 32;
    if (bm > TL_BMAX) bm = TL_BMAX;
    return int(bm);
}

extern "C" size_t osn_tile_lists_bytes(int64_t n_out, int K, int bm) {
    if (n_out <= 0 || K < 1 || bm < 1) return 256;
    const size_t nt = size_t(cdiv(n_out, bm));
    return align_up(nt * size_t(K) * 4, 256) + nt * size_t(K) * size_t(bm) * 8;
}

extern "C" int osn_tile_lists_build(const int32_t* nbr, int64_t n_out, int K, int bm, void* tl, osn_stream_t stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    OSN_REQUIRE(n_out >= 0 && n_out < (int64_t(1) << 31), OSN_E_ARG, "osn_tile_lists_build: n_out out of range");
    OSN_REQUIRE(K >= 1 && K <= TL_KMAX, OSN_E_ARG, "osn_tile_lists_build: K=%d (at most %d offsets)", K, TL_KMAX);
    OSN_REQUIRE(bm >= 1 && bm <= TL_BMAX, OSN_E_ARG, "osn_tile_lists_build: bm=%d (at most %d rows per tile)", bm, TL_BMAX);
    if (n_out == 0) return OSN_OK;
    OSN_REQUIRE(nbr && tl, OSN_E_ARG, "osn_tile_lists_build: null pointer");
    const int64_t nt = cdiv(n_out, bm);
    int32_t* cnt = static_cast<int32_t*>(tl);
    int2* lst = reinterpret_cast<int2*>(static_cast<char*>(tl) + align_up(size_t(nt) * K * 4, 256));
    hipLaunchKernelGGL(tile_lists_kernel, dim3(unsigned(nt)), dim3(256), 0, st, nbr, n_out, K, bm, cnt, lst);
    OSN_LAUNCH_CHECK();
    return OSN_OK;
}

extern "C" size_t osn_weight_prep_tl_bytes(int K, int cin, int cout, int for_dgrad) {
    const int nc = for_dgrad ? cout : cin, nn = for_dgrad ? cin : cout;
    return size_t(3) * size_t(K) * size_t((nc + 31) / 32) * size_t((nn + 15) / 16) * 1024;
}

extern "C" int osn_weight_prep_tl(const float* W, int K, int cin, int cout, int flip, void* Wp_fwd, void* Wp_dgrad,
                                  osn_stream_t stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    OSN_REQUIRE(W && (Wp_fwd || Wp_dgrad) && K >= 1 && cin >= 1 && cout >= 1, OSN_E_ARG, "osn_weight_prep_tl: bad arguments");
    const int64_t ppf = Wp_fwd ? int64_t(K) * ((cin + 31) / 32) * ((cout + 15) / 16) * 512 : 0;
    const int64_t ppb = Wp_dgrad ? int64_t(K) * ((cout + 31) / 32) * ((cin + 15) / 16) * 512 : 0;
    int g = int(cdiv(ppf + ppb, 256));
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(weight_prep_tl_kernel, dim3(g), dim3(256), 0, st, W, K, cin, cout, flip, ppf, ppb,
                       static_cast<__bf16*>(Wp_fwd), static_cast<__bf16*>(Wp_dgrad));
    OSN_LAUNCH_CHECK();
    return OSN_OK;
}

extern "C" int osn_spconv_fwd_tl(const float* in, int64_t n_in, const void* Wp, const void* tl, const int32_t* out_rows,
                                 float* out, double* bn_partial, int64_t n_out, int K, int cin, int cout, int bm,
                                 osn_stream_t stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    OSN_REQUIRE(n_out >= 0 && n_out < (int64_t(1) << 31), OSN_E_ARG, "osn_spconv_fwd_tl: n_out out of range");
    OSN_REQUIRE(K >= 1 && K <= TL_KMAX && cin >= 4 && (cin & 3) == 0 && cout >= 4 && (cout & 3) == 0, OSN_E_ARG,
                "osn_spconv_fwd_tl: needs K <= %d, cin %% 4 == 0, cout %% 4 == 0 (K=%d cin=%d cout=%d)", TL_KMAX, K, cin, cout);
    OSN_REQUIRE(bm >= 1 && bm <= TL_BMAX, OSN_E_ARG, "osn_spconv_fwd_tl: bm=%d (at most %d rows per tile)", bm, TL_BMAX);
    OSN_REQUIRE(n_in >= 0 && n_in <= (int64_t(1) << 24), OSN_E_RANGE,
                "osn_spconv_fwd_tl: %lld input rows (the kernel packs an input row into 24 bits; use osn_spconv_fwd_x6)", (long long)n_in);
    if (n_out == 0) return OSN_OK;
    OSN_REQUIRE(in && Wp && out, OSN_E_ARG, "osn_spconv_fwd_tl: null pointer");
    OSN_REQUIRE(tl || (K == 1 && !out_rows), OSN_E_ARG, "osn_spconv_fwd_tl: tile lists may be null only for K == 1 (identity map)");
    OSN_REQUIRE(aligned16(in) && aligned16(Wp) && aligned16(out), OSN_E_ARG, "osn_spconv_fwd_tl: pointers must be 16-byte aligned");
    const int64_t nt = cdiv(n_out, bm);
    const int32_t* cnt = static_cast<const int32_t*>(tl);
    const int2* lst = tl ? reinterpret_cast<const int2*>(static_cast<const char*>(tl) + align_up(size_t(nt) * K * 4, 256)) : nullptr;
    const TlPlan p = plan_tl(cout);
    const int ns = (cin + 31) / 32, ncb = (cout + 15) / 16;
    const dim3 grid(unsigned(nt), unsigned(p.gy)), block(256);
    const bf16x8* wp = static_cast<const bf16x8*>(Wp);
#define OSN_TL(NB_, KS_)                                                                                              \
    hipLaunchKernelGGL((spconv_tl_kernel<NB_, KS_>), grid, block, 0, st, in, wp, cnt, lst, out_rows, out, bn_partial, \
                       int(n_out), K, cin, cout, bm, ns, ncb)
    switch (p.nb16) {
        case 4: OSN_TL(4, 2); break;
        case 3: OSN_TL(3, 3); break;
        case 2: OSN_TL(2, 4); break;
        default: OSN_TL(1, 4); break;
    }
#undef OSN_TL
    OSN_LAUNCH_CHECK();
    return OSN_OK;
}
