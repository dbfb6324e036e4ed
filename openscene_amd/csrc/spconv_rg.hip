// Sparse 3-D convolution on gfx950 for NARROW layers (32 / 64 channels): register gather, no LDS staging, no barrier in the loop.
//
// Function parity (not a port) with MinkowskiEngine's convolution kernels:
//   out[o] = sum_k in[nbr[k, o]] @ W[k]        (forward; with the input-gradient weight image and the mirrored table: the input gradient)
// for the convolutions the other kernels serve badly (round 5's per-stage table, profiles/r05_s20_bench_detail.json): the 32 -> 32
// 3^3 layers of the 52 k-row level and the 32 / 64-channel layers of the 13 k-row level ran 30 - 50 us each on the first-generation
// kernel (spconv.hip: 128-row tiles, gathered rows AND the weight chunk staged through LDS, two barriers per (offset, chunk)) for
// 1 - 3 us of matrix work -- a chain of ~30 barrier-separated stages per tile.  With <= 64 input channels none of that machinery is
// needed:
//   * the A operand of v_mfma_f32_16x16x32_bf16 is "lane l: row l & 15, 8 consecutive channels 8 (l >> 4) ..": exactly 32 bytes of
//     one gathered fp32 row.  Every lane loads ITS 32 bytes straight from the feature matrix (two 16-byte buffer loads through the
//     neighbour table; an absent neighbour is an out-of-range offset: the load returns zeros and moves no data), splits them into the
//     three bf16 pieces in registers and multiplies -- no staging buffer, no LDS round trip, no barrier;
//   * the weight fragments of one offset (<= 24 KB) are loaded once per offset and wave from the fragment-order image the tile-list
//     kernel uses (osn_weight_prep_tl), and reused over all row blocks of the workgroup;
//   * a workgroup = 64 (x HALVES) table rows; its four waves take the offsets k = w, w + 4, ... (a quarter of the chain each) and
//     their partial tiles are summed through LDS in wave order at the end => fixed summation order, bitwise reproducible;
//   * (block, offset) groups without a single pair skip their MFMAs (wave-uniform); on a tile-ordered table (osn_kmap_sort) most are.
// Arithmetic: "bf16x6" as everywhere (three bf16 pieces per operand, six MFMAs per product block, fp32 accumulate, smallest terms first).
#include "common.h"
#include "split.h"
#include "epilogue.h"

namespace osn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int RG_RB = 4;                 // 16-row blocks per wave and half: 64 rows

// KS = input channels / 32 (1, 2), NCB = output channels / 16 (2, 4), HALVES = 64-row halves per workgroup (weight fragments reused)
// OCC: workgroups per CU the register allocation is held to (32 -> 32: 168 registers without a spill = three workgroups per CU, all
// 745 workgroups of the 48 k-row level resident at once; the 64-channel instances need more)
// BD: the weight fragments of the NEXT offset loaded one iteration ahead into a second register set.  Measured SLOWER (48 k rows
// 32 -> 32: 24.8 against 22.0 us; 13 k rows 32 -> 64: 18.9 / 15.6): the extra registers cost a resident workgroup per CU, and resident
// waves are what hides this kernel's per-offset latency chain.  Not instantiated.
// EPI (inference): the stage's evaluation-mode batch norm in the epilogue (epilogue.h)
template <int KS, int NCB, int HALVES, int OCC, bool BD, bool EPI = false>
__global__ __launch_bounds__(256, OCC) void spconv_rg_kernel(const float* __restrict__ in, const bf16x8* __restrict__ Wp,
                                                        const int32_t* __restrict__ nbr, const int32_t* __restrict__ out_rows,
                                                        float* __restrict__ out, int n_out, int K, int cin, int cout,
                                                        unsigned in_bytes, const Epi epi) {
    constexpr int ROWS = 64 * HALVES;
    constexpr int LDR = 32 + 4;                                   // fp32 row pitch of a partial tile in LDS (32 columns per pass)
    __shared__ __attribute__((aligned(16))) float red[4][64][LDR];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lg = lane >> 4;
    const int row0 = blockIdx.x * ROWS;
    const __amdgpu_buffer_rsrc_t insrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), 0, int(in_bytes), 0x00020000);
    const unsigned cin4 = unsigned(cin) * 4u;
    const unsigned ns = unsigned(KS), ncb = unsigned(NCB);
    const uint32_t plane_blocks = unsigned(K) * ns * ncb;        // 1 KB blocks per bf16 piece of the whole weight

    f32x4 acc[HALVES][RG_RB][NCB];
#pragma unroll
    for (int h = 0; h < HALVES; ++h)
#pragma unroll
        for (int rb = 0; rb < RG_RB; ++rb)
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) acc[h][rb][cb] = f32x4{0.f, 0.f, 0.f, 0.f};

    // table entries of offset k for this lane's A rows: row (h, rb, l15); rows past the table: -1
    auto load_idx = [&](int k, int (&idx)[HALVES][RG_RB]) {
#pragma unroll
        for (int h = 0; h < HALVES; ++h)
#pragma unroll
            for (int rb = 0; rb < RG_RB; ++rb) {
                const int r = row0 + 64 * h + 16 * rb + l15;
                const bool on = k < K && r < n_out;
                const int v = nbr[on ? int64_t(k) * n_out + r : 0];
                idx[h][rb] = on ? v : -1;
            }
    };
    // the lane's 32 bytes of every gathered row (absent neighbour: offset past the matrix -> zeros, no memory access)
    auto gather = [&](const int (&idx)[HALVES][RG_RB], float4 (&P)[HALVES][RG_RB][KS][2]) {
#pragma unroll
        for (int h = 0; h < HALVES; ++h)
#pragma unroll
            for (int rb = 0; rb < RG_RB; ++rb) {
                const unsigned base = idx[h][rb] >= 0 ? __umul24(unsigned(idx[h][rb]), cin4) + 32u * unsigned(lg) : 0xFFFFFF00u;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const unsigned o = idx[h][rb] >= 0 ? base + 128u * unsigned(ks) : base;
                    P[h][rb][ks][0] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(insrc, o, 0, 0));
                    P[h][rb][ks][1] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(insrc, o + 16u, 0, 0));
                }
            }
    };

    int idxA[HALVES][RG_RB], idxB[HALVES][RG_RB];
    float4 PA[HALVES][RG_RB][KS][2], PB[HALVES][RG_RB][KS][2];
    // software pipeline over this wave's offsets k = wave, wave + 4, ...: table entries two offsets ahead, gathered rows one ahead
    load_idx(wave, idxA);
    load_idx(wave + 4, idxB);
    gather(idxA, PA);

    auto load_b = [&](int k, bf16x8 (&B)[KS][NCB][3]) {
        const unsigned kk = k < K ? unsigned(k) : 0u;            // (past the last offset: any valid block, never multiplied)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    const uint32_t blk = uint32_t(pl) * plane_blocks + (kk * ns + unsigned(ks)) * ncb + unsigned(cb);
                    B[ks][cb][pl] = (Wp + (size_t(blk) << 6))[lane];
                }
    };
    bf16x8 BA[KS][NCB][3], BB[BD ? KS : 1][BD ? NCB : 1][3];
    if (BD) load_b(wave, BA);

    auto body = [&](int k, int (&idx_cur)[HALVES][RG_RB], float4 (&P_cur)[HALVES][RG_RB][KS][2], int (&idx_nxt)[HALVES][RG_RB],
                    float4 (&P_nxt)[HALVES][RG_RB][KS][2], bf16x8 (&B)[KS][NCB][3], bf16x8 (&B_nxt)[BD ? KS : 1][BD ? NCB : 1][3]) {
        // which (half, block) groups have any pair at this offset (wave-uniform masks), before idx_cur is recycled
        bool any[HALVES][RG_RB];
#pragma unroll
        for (int h = 0; h < HALVES; ++h)
#pragma unroll
            for (int rb = 0; rb < RG_RB; ++rb) any[h][rb] = __ballot(idx_cur[h][rb] >= 0) != 0ull;
        // weight fragments: one coalesced 1 KB load each -- this offset's now, or (BD) the next offset's into the other set
        if constexpr (BD) {
            load_b(k + 4, reinterpret_cast<bf16x8(&)[KS][NCB][3]>(B_nxt));
        } else {
            load_b(k, B);
        }
        // next offset's rows in flight during this offset's MFMAs; the offset after that: its table entries
        gather(idx_nxt, P_nxt);
        load_idx(k + 8, idx_cur);
#pragma unroll
        for (int h = 0; h < HALVES; ++h)
#pragma unroll
            for (int rb = 0; rb < RG_RB; ++rb) {
                if (any[h][rb]) {
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
                        bf16x4 a1, a2, a3, b1, b2, b3;
                        tl_split4(P_cur[h][rb][ks][0], a1, a2, a3);
                        tl_split4(P_cur[h][rb][ks][1], b1, b2, b3);
                        const bf16x8 A1 = __builtin_shufflevector(a1, b1, 0, 1, 2, 3, 4, 5, 6, 7);
                        const bf16x8 A2 = __builtin_shufflevector(a2, b2, 0, 1, 2, 3, 4, 5, 6, 7);
                        const bf16x8 A3 = __builtin_shufflevector(a3, b3, 0, 1, 2, 3, 4, 5, 6, 7);
                        // smallest terms first per accumulator; consecutive MFMAs on different accumulators
#define RG_MFMA(AP, BP)                                                                                            \
    _Pragma("unroll") for (int cb = 0; cb < NCB; ++cb)                                                             \
        acc[h][rb][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(B[ks][cb][BP], AP, acc[h][rb][cb], 0, 0, 0);
                        RG_MFMA(A3, 0) RG_MFMA(A2, 1) RG_MFMA(A1, 2) RG_MFMA(A2, 0) RG_MFMA(A1, 1) RG_MFMA(A1, 0)
#undef RG_MFMA
                    }
                }
            }
    };
    for (int k = wave; k < K; k += 8) {
        if constexpr (BD) {
            body(k, idxA, PA, idxB, PB, BA, BB);
            if (k + 4 < K) body(k + 4, idxB, PB, idxA, PA, reinterpret_cast<bf16x8(&)[KS][NCB][3]>(BB), reinterpret_cast<bf16x8(&)[BD ? KS : 1][BD ? NCB : 1][3]>(BA));
        } else {
            body(k, idxA, PA, idxB, PB, BA, BB);
            if (k + 4 < K) body(k + 4, idxB, PB, idxA, PA, BA, BB);
        }
    }

    // ---- the four waves' partial tiles -> out, summed in wave order (64 rows x 32 columns at a time through LDS)
#pragma unroll
    for (int h = 0; h < HALVES; ++h) {
#pragma unroll
        for (int cp = 0; cp < NCB / 2; ++cp) {
            if (h > 0 || cp > 0) __syncthreads();                // the previous pass has been read
#pragma unroll
            for (int rb = 0; rb < RG_RB; ++rb)
#pragma unroll
                for (int c2 = 0; c2 < 2; ++c2)
                    *reinterpret_cast<f32x4*>(&red[wave][16 * rb + l15][16 * c2 + 4 * lg]) = acc[h][rb][2 * cp + c2];
            __syncthreads();
            EpiCols ec;
            if constexpr (EPI) ec = epi_cols(epi, 32 * cp + 4 * (tid & 7));       // (256 % 8 == 0: a thread keeps its column quad)
            for (int e = tid; e < 64 * 8; e += 256) {
                const int j = e >> 3, c4 = e & 7;
                const int r = row0 + 64 * h + j;
                if (r < n_out) {
                    float4 s = *reinterpret_cast<const float4*>(&red[0][j][4 * c4]);
#pragma unroll
                    for (int w = 1; w < 4; ++w) {
                        const float4 v = *reinterpret_cast<const float4*>(&red[w][j][4 * c4]);
                        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
                    }
                    const int64_t orow = out_rows ? out_rows[r] : r;
                    if constexpr (EPI) s = epi_apply(epi, ec, s, orow, 32 * cp + 4 * c4, cout);
                    *reinterpret_cast<float4*>(out + orow * cout + 32 * cp + 4 * c4) = s;
                }
            }
        }
    }
}

}  // namespace osn

using namespace osn;

// shapes the kernel takes: 32 or 64 channels on both sides, a table (K > 1), a feature matrix below 2 GB and 2^24 rows
extern "C" int osn_spconv_fwd_rg_ok(int64_t n_in, int K, int cin, int cout) {
    return K > 1 && K <= 128 && (cin == 32 || cin == 64) && (cout == 32 || cout == 64) && n_in >= 1 && n_in < (int64_t(1) << 24) &&
           uint64_t(n_in) * uint64_t(cin) * 4u < (uint64_t(1) << 31);
}

int osn::spconv_fwd_rg_epi(const float* in, int64_t n_in, const void* Wp, const int32_t* nbr, const int32_t* out_rows, float* out,
                           int64_t n_out, int K, int cin, int cout, const Epi& epi, osn_stream_t stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    OSN_REQUIRE(n_out >= 0 && n_out < (int64_t(1) << 31), OSN_E_ARG, "osn_spconv_fwd_rg: n_out out of range");
    OSN_REQUIRE(osn_spconv_fwd_rg_ok(n_in > 0 ? n_in : 1, K, cin, cout), OSN_E_ARG,
                "osn_spconv_fwd_rg: needs K > 1, 32 or 64 channels on both sides, a feature matrix below 2 GB (K=%d cin=%d cout=%d n_in=%lld)",
                K, cin, cout, (long long)n_in);
    if (n_out == 0) return OSN_OK;
    OSN_REQUIRE(in && Wp && nbr && out, OSN_E_ARG, "osn_spconv_fwd_rg: null pointer");
    OSN_REQUIRE(aligned16(in) && aligned16(Wp) && aligned16(out), OSN_E_ARG, "osn_spconv_fwd_rg: pointers must be 16-byte aligned");
    const unsigned in_bytes = unsigned(uint64_t(n_in) * uint64_t(cin) * 4u);
    const bf16x8* wp = static_cast<const bf16x8*>(Wp);
    const int ks = cin / 32, ncb = cout / 16;
    // (two 64-row halves per workgroup -- each offset's fragments loaded once for both -- need 256 + registers: not instantiated)
    const dim3 grid(unsigned(cdiv(n_out, 64)));
#define OSN_RG(KS_, NCB_, H_, OCC_, BD_)                                                                                                              \
    do {                                                                                                                                              \
        if (epi.mean)                                                                                                                                 \
            hipLaunchKernelGGL((spconv_rg_kernel<KS_, NCB_, H_, OCC_, BD_, true>), grid, dim3(256), 0, st, in, wp, nbr, out_rows, out, int(n_out), K, \
                               cin, cout, in_bytes, epi);                                                                                             \
        else                                                                                                                                          \
            hipLaunchKernelGGL((spconv_rg_kernel<KS_, NCB_, H_, OCC_, BD_, false>), grid, dim3(256), 0, st, in, wp, nbr, out_rows, out, int(n_out), K, \
                               cin, cout, in_bytes, epi);                                                                                             \
    } while (0)
    if (ks == 1 && ncb == 2) OSN_RG(1, 2, 1, 3, false);
    else if (ks == 1 && ncb == 4) OSN_RG(1, 4, 1, 2, false);
    else if (ks == 2 && ncb == 2) OSN_RG(2, 2, 1, 2, false);
    else OSN_RG(2, 4, 1, 1, false);
#undef OSN_RG
    OSN_LAUNCH_CHECK();
    return OSN_OK;
}

extern "C" int osn_spconv_fwd_rg(const float* in, int64_t n_in, const void* Wp, const int32_t* nbr, const int32_t* out_rows,
                                 float* out, int64_t n_out, int K, int cin, int cout, osn_stream_t stream) {
    return spconv_fwd_rg_epi(in, n_in, Wp, nbr, out_rows, out, n_out, K, cin, cout, epi_none(), stream);
}
