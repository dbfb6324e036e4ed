// Sparse 3-D convolution on gfx950 for the SMALL maps (<= 16 k rows): weight-stationary workgroups over the per-offset
// pair arrays, partial rows per offset, ordered sum.
//
// Function parity (not a port) with MinkowskiEngine's convolution kernels:
//   out[o] = sum_k in[nbr[k,o]] @ W[k]      (forward, and -- on the transposed map with the mirrored / transposed weight
//                                            image -- the input gradient).
//
// Why another kernel (measured): on the 13 k / 3.3 k / 730-row levels of a 100 k-voxel
// scene an output-stationary tile (spconv_fwd_x6, spconv_tl) meets a (tile, offset) with 10-30 real pairs and has to
// bring the offset's WHOLE weight slice (128 x 128 x 3 bf16 planes = 98 KB) into the CU for it: 52-210 tiles x 27 offsets
// x 98 KB = 140-560 MB through the 64 B/clk load path of the CUs for a convolution of 0.04-0.2 M pairs -- 45-100 us each
// at 25-50 TF, with a per-workgroup chain of 27 fragment loads.  Here the roles are swapped:
//   * a workgroup owns ONE offset k and a chunk of <= 128 of its pairs (pin / pout / poff of osn_pair_lists_build, the
//     arrays the weight gradient already uses): W[k]'s B fragments are loaded once into registers and multiply every
//     pair of the chunk -- weight traffic drops to (chunks x 98 KB), the chain to one fragment load + <= 4 steps;
//   * the result row of pair (i -> o) at offset k goes to partial[k][o] (a row is written by exactly one workgroup:
//     an output row has at most one input row per offset);
//   * a second kernel sums out[o] = sum over k ascending of partial[k][o] for the offsets the map has at o (the
//     neighbour table of the destination side tells which) => fixed summation order, bitwise reproducible.
// The partial rows cost pairs x Cout x 4 B written and read once (20-150 MB on these levels, L2 / Infinity-Cache
// resident), which is why the 48 k / 100 k-row levels stay on the tile-list kernel.
// Arithmetic: "bf16x6" as in spconv_tl.hip (three bf16 pieces per operand, six MFMAs per product block, fp32 accumulate).
#include "common.h"
#include "split.h"
#include "pairlist.h"
#include "epilogue.h"

namespace osn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int WS_NG = 4;            // 32-pair steps per workgroup at most
constexpr int WS_CH = 32 * WS_NG;   // pairs per workgroup at most

// 4 waves; wave w < NW owns output columns [32 w, 32 w + 32) of the workgroup's column group (waves >= NW only gather
// and stage); KS k-steps of 32 input channels per chunk (B fragments of a chunk: KS x 2 column blocks x 3 planes).
// EPI (inference, direct launches): the stage's evaluation-mode batch norm in the epilogue (epilogue.h)
template <int NW, int KS, bool EPI = false>
__global__ __launch_bounds__(256) void spconv_ws_kernel(const float* __restrict__ in, const bf16x8* __restrict__ Wp,
                                                        const int32_t* __restrict__ gsrc, const int32_t* __restrict__ gdst,
                                                        const int32_t* __restrict__ poff, float* __restrict__ partial,
                                                        float* __restrict__ zeros, int n_dst, int K, int cin, int cout, int ns, int ncb, int chunk,
                                                        int direct, const Epi epi) {
    constexpr int NT = 256;
    constexpr int CK = 32 * KS;               // input channels per chunk
    constexpr int LDA = CK + 8;               // bf16 row stride of a staged plane (16-byte aligned rows)
    constexpr int QPR = CK / 4;               // 4-channel quads per staged row
    constexpr int NQ = (32 * QPR + NT - 1) / NT;
    __shared__ __attribute__((aligned(16))) __bf16 stage[2][3][32][LDA];
    __shared__ int sidx[WS_CH], didx[WS_CH];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int k = blockIdx.y;
    // the line of zeros the reduction reads for the offsets a row does not have
    if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && tid < 64) zeros[tid] = 0.f;
    const int pk0 = poff[k], pk1 = poff[k + 1];
    const int p0 = pk0 + blockIdx.x * chunk;
    if (p0 >= pk1) return;                                             // this offset has fewer chunks
    const int np = min(chunk, pk1 - p0);
    const int nsteps = (np + 31) >> 5;
    const int col0 = blockIdx.z * (32 * NW);
    const int cb0 = blockIdx.z * (2 * NW) + 2 * wave;                  // this wave's first 16-column block

    for (int e = tid; e < WS_CH; e += NT) {
        sidx[e] = e < np ? gsrc[p0 + e] : 0;                           // padded pair: input row 0 (valid address)
        didx[e] = e < np ? gdst[p0 + e] : -1;
    }
    int q_row[NQ], q_col[NQ];
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
        const int idx = tid + NT * j;
        q_row[j] = (idx / QPR) & 31;
        q_col[j] = (idx % QPR) * 4;
    }
    __syncthreads();

    f32x4 acc[WS_NG][2][2];
#pragma unroll
    for (int g = 0; g < WS_NG; ++g)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) acc[g][h][nb] = f32x4{0.f, 0.f, 0.f, 0.f};

    float4 P[NQ];
    auto fetch = [&](int g, int s0) {
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            const unsigned row = unsigned(sidx[32 * g + q_row[j]]);
            const int ch = 32 * s0 + q_col[j];
            const unsigned cu = ch < cin ? unsigned(ch) : 0u;
            P[j] = *reinterpret_cast<const float4*>(in + (uint64_t(row) * unsigned(cin) + cu));
        }
    };
    int buf = 0;
    for (int s0 = 0; s0 < ns; s0 += KS) {
        // ---- the chunk's weight fragments: one coalesced 1 KB load each, consumed after the first barrier
        bf16x8 B[KS][2][3];
        if (wave < NW) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) {
                    const bool on = s0 + ks < ns && cb0 + nb < ncb;
                    const unsigned sb = on ? unsigned(s0 + ks) : 0u, cb = on ? unsigned(cb0 + nb) : 0u;
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) {
                        const unsigned blk = (unsigned(pl * K + k) * unsigned(ns) + sb) * unsigned(ncb) + cb;
                        B[ks][nb][pl] = (Wp + (size_t(blk) << 6))[lane];
                    }
                }
        }
        fetch(0, s0);
#pragma unroll
        for (int g = 0; g < WS_NG; ++g) {
            if (g < nsteps) {
                // ---- split the fetched quads into three bf16 pieces, stage them
#pragma unroll
                for (int j = 0; j < NQ; ++j) {
                    const bool ok = 32 * s0 + q_col[j] < cin;
                    bf16x4 p1, p2, p3;
                    tl_split4(ok ? P[j] : make_float4(0.f, 0.f, 0.f, 0.f), p1, p2, p3);      // split.h: two elements per conversion
                    if (NT * NQ == 32 * QPR || tid + NT * j < 32 * QPR) {
                        *reinterpret_cast<bf16x4*>(&stage[buf][0][q_row[j]][q_col[j]]) = p1;
                        *reinterpret_cast<bf16x4*>(&stage[buf][1][q_row[j]][q_col[j]]) = p2;
                        *reinterpret_cast<bf16x4*>(&stage[buf][2][q_row[j]][q_col[j]]) = p3;
                    }
                }
                if (g + 1 < nsteps) fetch(g + 1, s0);      // the next step's rows: in flight during the MFMAs below
                // one barrier per step: the other stage buffer was last read in the step before this barrier's
                __syncthreads();
                if (wave < NW) {
                    const bool half1 = np - 32 * g > 16;               // the second 16-pair half holds real pairs
                    const int akq = 8 * (lane >> 4);
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
                        if (s0 + ks < ns) {
                            bf16x8 af[2][3];
#pragma unroll
                            for (int h = 0; h < 2; ++h)
#pragma unroll
                                for (int pl = 0; pl < 3; ++pl)
                                    af[h][pl] = *reinterpret_cast<const bf16x8*>(
                                        &stage[buf][pl][(half1 ? h : 0) * 16 + (lane & 15)][ks * 32 + akq]);
                            // smallest terms first per accumulator, consecutive MFMAs on different accumulators
#define WS_MFMA(H1, AP, BP)                                                                                   \
    _Pragma("unroll") for (int h = 0; h < H1; ++h) _Pragma("unroll") for (int nb = 0; nb < 2; ++nb)           \
        acc[g][h][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(B[ks][nb][BP], af[h][AP], acc[g][h][nb], 0, 0, 0);
                            if (half1) {
                                WS_MFMA(2, 2, 0) WS_MFMA(2, 1, 1) WS_MFMA(2, 0, 2) WS_MFMA(2, 1, 0) WS_MFMA(2, 0, 1) WS_MFMA(2, 0, 0)
                            } else {
                                WS_MFMA(1, 2, 0) WS_MFMA(1, 1, 1) WS_MFMA(1, 0, 2) WS_MFMA(1, 1, 0) WS_MFMA(1, 0, 1) WS_MFMA(1, 0, 0)
                            }
#undef WS_MFMA
                        }
                    }
                }
                buf ^= 1;
            }
        }
    }
    if (wave >= NW) return;
    EpiCols ec[2];
    if constexpr (EPI) {                         // (direct launches: the lane's two column quads, constants once)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            const int col = col0 + 32 * wave + 16 * nb + 4 * (lane >> 4);
            ec[nb] = epi_cols(epi, col < cout ? col : 0);
        }
    }
    // ---- result rows -> partial[k][dst].  The weight fragment is the MFMA's FIRST operand (the staged rows the second), so a
    // block comes out transposed: lane l holds columns 4 (l >> 4) .. + 3 of pair l & 15 -- one 16-byte store per block
#pragma unroll
    for (int g = 0; g < WS_NG; ++g) {
        if (g < nsteps) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int d = didx[32 * g + 16 * h + (lane & 15)];
                if (d >= 0) {
                    // direct: every destination row has exactly one pair in the whole map -- `partial` IS the output
                    float* row = partial + ((direct ? int64_t(0) : int64_t(k) * n_dst) + d) * cout;
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb) {
                        const int col = col0 + 32 * wave + 16 * nb + 4 * (lane >> 4);
                        if (col < cout) {
                            float4 v = make_float4(acc[g][h][nb][0], acc[g][h][nb][1], acc[g][h][nb][2], acc[g][h][nb][3]);
                            if constexpr (EPI) v = epi_apply(epi, ec[nb], v, d, col, cout);
                            *reinterpret_cast<float4*>(row + col) = v;
                        }
                    }
                }
            }
        }
    }
}

// out[r] = sum over k ascending, nbr[k][r] >= 0, of partial[k][r]  (rows without any neighbour: zeros)
// KT: offsets of the map when it is one of the two sizes the U-Net has (fully unrolled: all loads in flight), else 0.
template <int KT, bool EPI = false>
__global__ __launch_bounds__(256) void spconv_ws_reduce_kernel(const float* __restrict__ partial, const int32_t* __restrict__ nbr,
                                                               const float* __restrict__ zeros, float* __restrict__ out,
                                                               int64_t n_dst, int K, int c4, const Epi epi) {
    const int64_t total = n_dst * c4;
    const int64_t e = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int64_t r = e / c4;
    const int c = int(e - r * c4);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (KT > 0) {
        // unconditional loads (an absent offset reads a line of zeros: x + 0 == x): K independent loads per thread
        float4 v[KT > 0 ? KT : 1];
#pragma unroll
        for (int k = 0; k < KT; ++k) {
            const bool on = nbr[int64_t(k) * n_dst + r] >= 0;
            const float* src = on ? partial + ((int64_t(k) * n_dst + r) * c4 + c) * 4 : zeros;
            v[k] = *reinterpret_cast<const float4*>(src);
        }
#pragma unroll
        for (int k = 0; k < KT; ++k) { s.x += v[k].x; s.y += v[k].y; s.z += v[k].z; s.w += v[k].w; }
    } else {
        for (int k = 0; k < K; ++k) {
            if (nbr[int64_t(k) * n_dst + r] >= 0) {
                const float4 v = *reinterpret_cast<const float4*>(partial + ((int64_t(k) * n_dst + r) * c4 + c) * 4);
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            }
        }
    }
    if constexpr (EPI) s = epi_quad(epi, s, r, 4 * c, 4 * c4);                 // evaluation-mode batch norm (epilogue.h)
    reinterpret_cast<float4*>(out)[e] = s;
}

static int ws_waves(int cout) {
    // waves (= 32-column groups) per workgroup: the width with the least padding, the wider one on ties
    int best = 1, best_pad = 1 << 30;
    for (int nw = 4; nw >= 1; --nw) {
        const int cw = 32 * nw;
        const int pad = int(cdiv(cout, cw)) * cw - cout;
        if (pad < best_pad) { best_pad = pad; best = nw; }
    }
    return best;
}

}  // namespace osn

using namespace osn;

// 256 bytes of zeros + the partial rows [K][n_dst][cout]
extern "C" size_t osn_spconv_fwd_ws_ws_bytes(int64_t n_dst, int K, int cout, int direct) {
    return 256 + (direct ? 0 : size_t(K) * size_t(n_dst > 0 ? n_dst : 0) * size_t(cout) * 4);
}

int osn::spconv_fwd_ws_epi(const float* in, int64_t n_in, const void* Wp, const void* pl, int64_t pl_rows, int swap, int direct,
                           const int32_t* nbr_dst, float* out, int64_t n_dst, int K, int cin, int cout, void* ws, size_t ws_bytes,
                           const Epi& epi, osn_stream_t stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    OSN_REQUIRE(n_dst >= 0 && n_dst < (int64_t(1) << 31) && n_in >= 0 && n_in < (int64_t(1) << 31), OSN_E_ARG,
                "osn_spconv_fwd_ws: row counts out of range");
    OSN_REQUIRE(K >= 2 && K <= PL_KMAX && cin >= 4 && (cin & 3) == 0 && cout >= 4 && (cout & 3) == 0, OSN_E_ARG,
                "osn_spconv_fwd_ws: needs 2 <= K <= %d, cin %% 4 == 0, cout %% 4 == 0 (K=%d cin=%d cout=%d)", PL_KMAX, K, cin, cout);
    OSN_REQUIRE(pl_rows == (swap ? n_in : n_dst), OSN_E_ARG,
                "osn_spconv_fwd_ws: the pair arrays are of a map with %lld output rows, this launch writes %lld rows (swap=%d)",
                (long long)pl_rows, (long long)(swap ? n_in : n_dst), swap);
    if (n_dst == 0) return OSN_OK;
    OSN_REQUIRE(in && Wp && pl && (nbr_dst || direct) && out, OSN_E_ARG, "osn_spconv_fwd_ws: null pointer");
    OSN_REQUIRE(aligned16(in) && aligned16(Wp) && aligned16(out) && aligned16(ws), OSN_E_ARG, "osn_spconv_fwd_ws: pointers must be 16-byte aligned");
    const size_t need = osn_spconv_fwd_ws_ws_bytes(n_dst, K, cout, direct);
    OSN_REQUIRE(ws && ws_bytes >= need, OSN_E_WS, "osn_spconv_fwd_ws: workspace %zu < %zu", ws_bytes, need);
    PlView v = pl_view(const_cast<void*>(pl), pl_rows, K, 1);      // (the tile height only places the scratch behind the arrays)
    const int32_t* gsrc = swap ? v.pout : v.pin;
    const int32_t* gdst = swap ? v.pin : v.pout;
    float* zeros = static_cast<float*>(ws);
    float* partial = reinterpret_cast<float*>(static_cast<char*>(ws) + 256);
    const int nw = ws_waves(cout);
    const int gz = int(cdiv(cout, 32 * nw));
    const int ns = (cin + 31) / 32, ncb = (cout + 15) / 16;
    const int ks = ns <= 4 ? ns : (ns % 4 == 0 ? 4 : (ns % 3 == 0 ? 3 : 4));
    // pairs per workgroup: 64 on the deepest maps (more workgroups than CUs matters more than weight traffic), else 128;
    // an offset has at most min(n_in, n_dst) pairs
    const int64_t pmax = n_in < n_dst ? n_in : n_dst;
    const int chunk = pmax <= 4096 ? 64 : WS_CH;       // (32 / 64 / 128 measured within 5 % of each other)
    const dim3 grid(unsigned(cdiv(pmax, chunk)), unsigned(K), unsigned(gz));
    const bf16x8* wp = static_cast<const bf16x8*>(Wp);
#define OSN_WS2(NW_, KS_)                                                                                              \
    do {                                                                                                               \
        if (direct && epi.mean)                                                                                        \
            hipLaunchKernelGGL((spconv_ws_kernel<NW_, KS_, true>), grid, dim3(256), 0, st, in, wp, gsrc, gdst, v.poff, out, zeros, \
                               int(n_dst), K, cin, cout, ns, ncb, chunk, direct, epi);                                 \
        else                                                                                                           \
            hipLaunchKernelGGL((spconv_ws_kernel<NW_, KS_, false>), grid, dim3(256), 0, st, in, wp, gsrc, gdst, v.poff,  \
                               direct ? out : partial, zeros, int(n_dst), K, cin, cout, ns, ncb, chunk, direct, epi_none()); \
    } while (0)
#define OSN_WS(NW_)                                                                                                    \
    do {                                                                                                               \
        switch (ks) {                                                                                                  \
            case 1: OSN_WS2(NW_, 1); break;                                                                            \
            case 2: OSN_WS2(NW_, 2); break;                                                                            \
            case 3: OSN_WS2(NW_, 3); break;                                                                            \
            default: OSN_WS2(NW_, 4); break;                                                                           \
        }                                                                                                              \
    } while (0)
    switch (nw) {
        case 4: OSN_WS(4); break;
        case 3: OSN_WS(3); break;
        case 2: OSN_WS(2); break;
        default: OSN_WS(1); break;
    }
#undef OSN_WS
#undef OSN_WS2
    if (direct) {
        OSN_LAUNCH_CHECK();
        return OSN_OK;
    }
    const int c4 = cout / 4;
    const dim3 rgrid(unsigned(cdiv(n_dst * c4, 256)));
#define OSN_WSR(KT_)                                                                                                                    \
    do {                                                                                                                                \
        if (epi.mean) hipLaunchKernelGGL((spconv_ws_reduce_kernel<KT_, true>), rgrid, dim3(256), 0, st, partial, nbr_dst, zeros, out, n_dst, K, c4, epi); \
        else hipLaunchKernelGGL((spconv_ws_reduce_kernel<KT_, false>), rgrid, dim3(256), 0, st, partial, nbr_dst, zeros, out, n_dst, K, c4, epi);         \
    } while (0)
    if (K == 27) OSN_WSR(27);
    else if (K == 8) OSN_WSR(8);
    else OSN_WSR(0);
#undef OSN_WSR
    OSN_LAUNCH_CHECK();
    return OSN_OK;
}

extern "C" int osn_spconv_fwd_ws(const float* in, int64_t n_in, const void* Wp, const void* pl, int64_t pl_rows,
                                 int swap, int direct, const int32_t* nbr_dst, float* out, int64_t n_dst, int K, int cin, int cout,
                                 void* ws, size_t ws_bytes, osn_stream_t stream) {
    return spconv_fwd_ws_epi(in, n_in, Wp, pl, pl_rows, swap, direct, nbr_dst, out, n_dst, K, cin, cout, ws, ws_bytes, epi_none(), stream);
}
