// 1x1 convolutions on gfx950: out[N, Cout] = in[N, Cin] @ B  -- the U-Net's head (`final`, 96 -> 768), the BasicBlock
// shortcut convolutions, their input gradients, and the folded head of the fused queries.
//
// Function parity (not a port) with [ME] MinkowskiConvolution(kernel_size=1) (models/mink_unet.py:108-113,
// models/resnet_base.py:101-107): a plain row-wise matrix product, no kernel map.
//
// Why a kernel of its own (measured): the gather kernels treat a 1x1 conv as a one-offset
// map -- 32-row steps with two barriers each, the result block added into an LDS tile -- and run the head at 92 TF
// (161 us), a 128 -> 96 shortcut on 100 k rows at 16 TF (153 us): 10x off either roof.  Without a map there is nothing
// to compact: a workgroup takes 64 consecutive rows, stages their split-bf16 pieces ONCE per channel chunk, and every
// wave multiplies them with the fragments of its 32 output columns held in registers -- 4 row blocks x 2 column blocks
// x 6 products per k-step between two barriers -- and stores its accumulators straight to the output rows.  When the
// whole contraction fits one chunk (Cin <= 128: the head, most shortcuts) the staged rows serve ALL column groups of the
// workgroup, so the input is read and split once.  Same arithmetic as spconv_tl.hip ("bf16x6": three bf16 pieces per
// operand, a3b1 + a2b2 + a1b3 + a2b1 + a1b2 + a1b1 in fp32, smallest terms first) and the same weight image
// (osn_weight_prep_tl with K = 1).
#include "common.h"
#include "split.h"
#include "epilogue.h"

namespace osn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int DN_BM = 64;           // rows per workgroup

// 4 waves; wave w owns output columns [128 cg + 32 w, + 32) of column group cg; KS k-steps of 32 input channels per chunk
// EPI (inference): the stage's evaluation-mode batch norm in the epilogue (epilogue.h); a template flag so that the training instances
// keep their register allocation
template <int KS, bool EPI = false>
__global__ __launch_bounds__(256, 2) void dense_kernel(const float* __restrict__ in, const bf16x8* __restrict__ Wp,
                                                    float* __restrict__ out, int64_t n, int cin, int cout, int ns, int ncb,
                                                    int64_t per_plane /* 1 KB blocks per weight plane */, const Epi epi) {
    constexpr int NT = 256;
    constexpr int CK = 32 * KS;
    constexpr int LDA = CK + 8;                     // bf16 row stride of a staged plane (16-byte aligned rows)
    constexpr int QPR = CK / 4;                     // 4-channel quads per staged row
    constexpr int NQ = DN_BM * QPR / NT;            // quads per thread per chunk (2 KS)
    __shared__ __attribute__((aligned(16))) __bf16 stage[3][DN_BM][LDA];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t r0 = int64_t(blockIdx.x) * DN_BM;
    const int ncg = (cout + 127) / 128;
    const bool single = ns <= KS;                   // the contraction is one chunk: the staged rows serve every column group

    int q_row[NQ], q_col[NQ];
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
        const int idx = tid + NT * j;
        q_row[j] = idx / QPR;
        q_col[j] = (idx % QPR) * 4;
    }
    float4 P[NQ];
    auto fetch = [&](int s0) {                      // unconditional, clamped addresses; masked at conversion time
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            const int64_t row = r0 + q_row[j] < n ? r0 + q_row[j] : n - 1;
            const int ch = 32 * s0 + q_col[j];
            P[j] = *reinterpret_cast<const float4*>(in + row * cin + (ch < cin ? ch : 0));
        }
    };
    auto stage_rows = [&](int s0) {
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            const bool ok = 32 * s0 + q_col[j] < cin && r0 + q_row[j] < n;
            bf16x4 p1, p2, p3;
            tl_split4(ok ? P[j] : make_float4(0.f, 0.f, 0.f, 0.f), p1, p2, p3);      // split.h: two elements per conversion
            *reinterpret_cast<bf16x4*>(&stage[0][q_row[j]][q_col[j]]) = p1;
            *reinterpret_cast<bf16x4*>(&stage[1][q_row[j]][q_col[j]]) = p2;
            *reinterpret_cast<bf16x4*>(&stage[2][q_row[j]][q_col[j]]) = p3;
        }
    };

    // column groups of this workgroup: all of them (single chunk), else the one blockIdx.y names
    const int cg_first = single ? 0 : int(blockIdx.y);
    const int cg_end = single ? ncg : int(blockIdx.y) + 1;
    fetch(0);
    for (int cg = cg_first; cg < cg_end; ++cg) {
        const int cb0 = cg * 8 + 2 * wave;          // this wave's first 16-column block
        f32x4 acc[4][2];
#pragma unroll
        for (int rb = 0; rb < 4; ++rb)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) acc[rb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
        const bool wave_on = cg * 128 + 32 * wave < cout;      // (a 96-column layer leaves the fourth wave staging only)
        for (int s0 = 0; s0 < ns; s0 += KS) {
            // ---- the chunk's weight fragments of this wave's columns: one coalesced 1 KB load each
            bf16x8 B[KS][2][3];
            if (wave_on)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) {
                    const bool on = s0 + ks < ns && cb0 + nb < ncb;
                    const int64_t blk = on ? int64_t(s0 + ks) * ncb + cb0 + nb : 0;
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) B[ks][nb][pl] = (Wp + ((pl * per_plane + blk) << 6))[lane];
                }
            if (!(single && cg > cg_first)) {       // (single chunk: the rows staged for the first column group stay)
                stage_rows(s0);
                if (s0 + KS < ns) fetch(s0 + KS);   // the next chunk's rows: in flight during the MFMAs below
                __syncthreads();
            }
            const int akq = 8 * (lane >> 4);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if (wave_on && s0 + ks < ns) {
#pragma unroll
                    for (int rb = 0; rb < 4; ++rb) {
                        bf16x8 af[3];
#pragma unroll
                        for (int pl = 0; pl < 3; ++pl)
                            af[pl] = *reinterpret_cast<const bf16x8*>(&stage[pl][16 * rb + (lane & 15)][ks * 32 + akq]);
#define DN_MFMA(AP, BP)                                                                                         \
    _Pragma("unroll") for (int nb = 0; nb < 2; ++nb)                                                            \
        acc[rb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(B[ks][nb][BP], af[AP], acc[rb][nb], 0, 0, 0);
                        DN_MFMA(2, 0) DN_MFMA(1, 1) DN_MFMA(0, 2) DN_MFMA(1, 0) DN_MFMA(0, 1) DN_MFMA(0, 0)
#undef DN_MFMA
                    }
                }
            }
            if (!single && s0 + KS < ns) __syncthreads();   // every wave is done reading the stage before the next chunk lands
        }
        // ---- accumulators -> output rows.  The weight fragment is the MFMA's FIRST operand (the staged rows the second), so a
        // block comes out transposed: lane l holds columns 4 (l >> 4) .. + 3 of row l & 15 -- one 16-byte store per block
        if (wave_on) {
            EpiCols ec[2];
            if constexpr (EPI) {                 // the lane's two column quads: constants once for the four row blocks
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) {
                    const int col = cg * 128 + 32 * wave + 16 * nb + 4 * (lane >> 4);
                    ec[nb] = epi_cols(epi, col < cout ? col : 0);
                }
            }
#pragma unroll
            for (int rb = 0; rb < 4; ++rb) {
                const int64_t row = r0 + 16 * rb + (lane & 15);
                if (row < n) {
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb) {
                        const int col = cg * 128 + 32 * wave + 16 * nb + 4 * (lane >> 4);
                        if (col < cout) {        // (cout % 4 == 0: a quad is inside or outside as a whole)
                            float4 v = make_float4(acc[rb][nb][0], acc[rb][nb][1], acc[rb][nb][2], acc[rb][nb][3]);
                            if constexpr (EPI) v = epi_apply(epi, ec[nb], v, row, col, cout);  // evaluation-mode batch norm (epilogue.h)
                            *reinterpret_cast<float4*>(out + row * cout + col) = v;
                        }
                    }
                }
            }
        }
    }
}

}  // namespace osn

using namespace osn;

int osn::dense_fwd_epi(const float* in, const void* Wp, float* out, int64_t n, int cin, int cout, const Epi& epi, osn_stream_t stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    OSN_REQUIRE(n >= 0 && n < (int64_t(1) << 31) && cin >= 4 && (cin & 3) == 0 && cout >= 4 && (cout & 3) == 0, OSN_E_ARG,
                "osn_dense_fwd: needs cin %% 4 == 0 and cout %% 4 == 0 (n=%lld cin=%d cout=%d)", (long long)n, cin, cout);
    if (n == 0) return OSN_OK;
    OSN_REQUIRE(in && Wp && out && aligned16(in) && aligned16(Wp) && aligned16(out), OSN_E_ARG, "osn_dense_fwd: null or unaligned pointer");
    const int ns = (cin + 31) / 32, ncb = (cout + 15) / 16;
    const int ks = ns <= 4 ? ns : (ns % 4 == 0 ? 4 : (ns % 3 == 0 ? 3 : 4));
    const int ncg = (cout + 127) / 128;
    const int64_t per_plane = int64_t(ns) * ncb;
    const dim3 grid(unsigned(cdiv(n, DN_BM)), unsigned(ns <= ks ? 1 : ncg));
    const bf16x8* wp = static_cast<const bf16x8*>(Wp);
    switch (ks) {
#define OSN_DN(KS_)                                                                                                              \
    do {                                                                                                                         \
        if (epi.mean) hipLaunchKernelGGL((dense_kernel<KS_, true>), grid, dim3(256), 0, st, in, wp, out, n, cin, cout, ns, ncb, per_plane, epi); \
        else hipLaunchKernelGGL((dense_kernel<KS_, false>), grid, dim3(256), 0, st, in, wp, out, n, cin, cout, ns, ncb, per_plane, epi);         \
    } while (0)
        case 1: OSN_DN(1); break;
        case 2: OSN_DN(2); break;
        case 3: OSN_DN(3); break;
        default: OSN_DN(4); break;
    }
#undef OSN_DN
    OSN_LAUNCH_CHECK();
    return OSN_OK;
}

extern "C" int osn_dense_fwd(const float* in, const void* Wp, float* out, int64_t n, int cin, int cout, osn_stream_t stream) {
    return dense_fwd_epi(in, Wp, out, n, cin, cout, epi_none(), stream);
}
