// Sparse 3-D convolution on gfx950: forward / input-gradient (one kernel) and
// weight-gradient, fp32 in / fp32 accumulate on v_mfma_f32_32x32x2_f32.
//
// Function parity (not a port) with MinkowskiEngine's convolution kernels
// (SURVEY.md section 2.1, appendix C item 5):  out[o] = sum_k in[nbr[k,o]] @ W[k].
//
// Design (MI355X-first):
//   * output-stationary implicit GEMM over the dense k-major neighbour table:
//     a workgroup owns BM output rows x BN output channels and walks the K
//     kernel offsets; per offset it gathers the BM input rows (16-byte loads,
//     one 128-byte row segment per 8 lanes) into LDS, stages W[k] once for all
//     four waves, and accumulates in registers -> no atomics, no scatter, one
//     coalesced store per output row, bitwise reproducible;
//   * an offset none of the tile's rows uses is skipped (block-uniform vote);
//   * 64-wide waves: each wave owns 32 rows x (32*TN) channels of accumulators
//     in the 32x32x2 f32 MFMA layout; LDS A tile padded to an odd stride so the
//     32-row column reads are conflict-free;
//   * small maps (deep U-Net levels) use narrower row tiles and split the offset
//     loop across workgroups (deterministic partial buffers + ordered reduce) so
//     the launch still covers the 256 CUs.
#include "common.h"
#include "weight_prep.h"
#include <stdlib.h>

namespace osn {

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int WM, int WN, int TN, int BK>
__global__ __launch_bounds__(256) void spconv_fwd_kernel(const float* __restrict__ in, const float* __restrict__ W,
                                                         const int32_t* __restrict__ nbr,
                                                         const int32_t* __restrict__ out_rows,
                                                         float* __restrict__ out, int n_out, int K, int cin, int cout,
                                                         int k_per_split, int to_partial) {
    static_assert(WM * WN == 4, "four waves per workgroup");
    constexpr int BM = 32 * WM;
    constexpr int BN = 32 * TN * WN;
    __shared__ float As[BM][BK + 1];
    __shared__ __attribute__((aligned(16))) float Bs[BK][BN];
    __shared__ int rowidx[BM];
    __shared__ int gflag[2][4];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int row0 = blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;
    const int k_begin = blockIdx.z * k_per_split;
    const int k_end = min(K, k_begin + k_per_split);
    const bool a_vec = (cin & 3) == 0;
    const bool b_vec = (cout & 3) == 0;

    f32x16 acc[TN];
#pragma unroll
    for (int t = 0; t < TN; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    for (int k = k_begin; k < k_end; ++k) {
        int idx = -1;
        if (tid < BM) {
            const int r = row0 + tid;
            if (r < n_out) idx = nbr ? nbr[int64_t(k) * n_out + r] : r;
            rowidx[tid] = idx;
        }
        // vote per 32-row group (double-buffered flags: a wave is never more than one barrier
        // ahead of the slowest one); the barrier also publishes rowidx
        {
            const unsigned long long b = __ballot(idx >= 0);
            if (lane == 0 && 2 * wave < WM) {
                gflag[k & 1][2 * wave] = (b & 0xFFFFFFFFull) != 0;
                if (2 * wave + 1 < WM) gflag[k & 1][2 * wave + 1] = (b >> 32) != 0;
            }
        }
        __syncthreads();
        unsigned gm = 0;
#pragma unroll
        for (int g = 0; g < WM; ++g) gm |= (gflag[k & 1][g] ? 1u : 0u) << g;
        if (gm == 0) continue;                 // no row of this tile uses offset k
        const bool my_rows_on = (gm >> wm) & 1u;

        for (int c0 = 0; c0 < cin; c0 += BK) {
            // ---- gather the A tile: BM rows x BK input channels
            if (BK == 32 && a_vec) {
                const int sub = tid & 7, r = tid >> 3;
#pragma unroll
                for (int p = 0; p < BM / 32; ++p) {
                    const int row = p * 32 + r;
                    if (!((gm >> p) & 1u)) continue;        // nobody will read this 32-row group
                    const int i = rowidx[row];
                    const int c = c0 + sub * 4;
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (i >= 0 && c < cin) v = *reinterpret_cast<const float4*>(in + int64_t(i) * cin + c);
                    As[row][sub * 4 + 0] = v.x;
                    As[row][sub * 4 + 1] = v.y;
                    As[row][sub * 4 + 2] = v.z;
                    As[row][sub * 4 + 3] = v.w;
                }
            } else {
                for (int e = tid; e < BM * BK; e += 256) {
                    const int row = e / BK, cc = e % BK;
                    const int i = rowidx[row];
                    const int c = c0 + cc;
                    As[row][cc] = (i >= 0 && c < cin) ? in[int64_t(i) * cin + c] : 0.f;
                }
            }
            // ---- stage the B tile: W[k][c0:c0+BK][n0:n0+BN]
            if (b_vec) {
                constexpr int V = BN / 4;  // float4 per row
                for (int f = tid; f < BK * V; f += 256) {
                    const int r = f / V, c4 = f % V;
                    const int c = c0 + r, n = n0 + c4 * 4;
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (c < cin && n < cout)
                        v = *reinterpret_cast<const float4*>(W + (int64_t(k) * cin + c) * cout + n);
                    *reinterpret_cast<float4*>(&Bs[r][c4 * 4]) = v;
                }
            } else {
                for (int e = tid; e < BK * BN; e += 256) {
                    const int r = e / BN, cc = e % BN;
                    const int c = c0 + r, n = n0 + cc;
                    Bs[r][cc] = (c < cin && n < cout) ? W[(int64_t(k) * cin + c) * cout + n] : 0.f;
                }
            }
            __syncthreads();
            // ---- MFMA: lane l feeds A[row = l&31][kk + (l>>5)], B[kk + (l>>5)][col = l&31]
            if (my_rows_on) {
                const int arow = wm * 32 + (lane & 31);
                const int kh = lane >> 5;
#pragma unroll
                for (int kk = 0; kk < BK; kk += 2) {
                    const float a = As[arow][kk + kh];
#pragma unroll
                    for (int t = 0; t < TN; ++t) {
                        const float b = Bs[kk + kh][(wn * TN + t) * 32 + (lane & 31)];
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
                    }
                }
            }
            __syncthreads();
        }
    }

    // ---- epilogue: C layout col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    float* dst = out;
    if (to_partial) dst = out + int64_t(blockIdx.z) * n_out * cout;
#pragma unroll
    for (int t = 0; t < TN; ++t) {
        const int col = n0 + (wn * TN + t) * 32 + (lane & 31);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = row0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (row < n_out && col < cout) {
                const int orow = (!to_partial && out_rows) ? out_rows[row] : row;
                dst[int64_t(orow) * cout + col] = acc[t][r];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// Pipelined variant (the one that runs for every 2^3 / 3^3 kernel of the U-Net): the offsets
// handled by the block (<= 32) are scanned once, the active ones are listed in LDS together
// with their row indices, and the (offset, channel-chunk) stages then run as a software
// pipeline: the gather + weight loads of stage s+1 are issued into registers before the MFMAs
// of stage s and written to LDS after them (one barrier pair per stage, global-load latency
// hidden behind 48-64 MFMAs per wave).
template <int WM, int WN, int TN>
__global__ __launch_bounds__(256, 2) void spconv_fwd_pipe_kernel(const float* __restrict__ in, const float* __restrict__ W,
                                                              const int32_t* __restrict__ nbr,
                                                              const int32_t* __restrict__ out_rows,
                                                              const uint32_t* __restrict__ gmask,
                                                              int32_t* __restrict__ tile_parts,
                                                              float* __restrict__ out, float* __restrict__ extra,
                                                              int n_out, int K, int cin, int cout, int k_per_split,
                                                              int to_partial, int unit_k) {
    static_assert(WM * WN == 4, "four waves per workgroup");
    constexpr int BM = 32 * WM;
    constexpr int BN = 32 * TN * WN;
    constexpr int BK = 32;
    constexpr int KC = 32;                       // offsets per block (k_per_split <= KC)
    constexpr int NA = BM / 32;                  // A float4 per thread per stage
    constexpr int BV = BN / 4;                   // float4 per B row
    constexpr int NB = (BK * BV + 255) / 256;    // B float4 per thread per stage
    __shared__ float As[BM][BK + 1];
    __shared__ __attribute__((aligned(16))) float Bs[BK][BN];
    __shared__ int ridx[2][BM];                  // row indices of the next two stages (by stage parity)
    __shared__ unsigned char gbits[KC][2];
    __shared__ int klist[KC];
    __shared__ int kgm[KC];
    __shared__ int nact_s;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int row0 = blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;
    // Two ways to cut the offset loop across blockIdx.z:
    //  * units mode (gmask given): part z takes the z-th group of `unit_k` ACTIVE offsets of this tile
    //    (work per block is bounded and even; parts > 0 go to `extra`, summed in order by a fix-up pass);
    //  * uniform mode: part z takes offsets [z*k_per_split, (z+1)*k_per_split) (small maps).
    const bool units = gmask != nullptr;
    const int part = blockIdx.z;
    const int k_begin = units ? 0 : part * k_per_split;
    const int nk = units ? K : min(K, k_begin + k_per_split) - k_begin;     // <= KC
    const int my_row = row0 + tid;                               // meaningful for tid < BM
    const bool row_ok = tid < BM && my_row < n_out;
    __shared__ uint32_t gm_s[4];

    if (units) {
        // ---- prologue from the precomputed 32-row group masks: no table scan
        if (tid < 4) {
            const int64_t g = int64_t(blockIdx.x) * WM + tid;
            gm_s[tid] = (tid < WM && g * 32 < n_out) ? gmask[g] : 0u;
        }
        __syncthreads();
        if (tid == 0) {
            const uint32_t tm = gm_s[0] | gm_s[1] | gm_s[2] | gm_s[3];
            int i = 0, n = 0;
            for (int kk = 0; kk < nk; ++kk) {
                if (!((tm >> kk) & 1u)) continue;
                if (i >= part * unit_k && i < (part + 1) * unit_k) {
                    klist[n] = kk;
                    kgm[n] = int((gm_s[0] >> kk) & 1u) | (int((gm_s[1] >> kk) & 1u) << 1) |
                             (int((gm_s[2] >> kk) & 1u) << 2) | (int((gm_s[3] >> kk) & 1u) << 3);
                    ++n;
                }
                ++i;
            }
            nact_s = n;
            if (part == 0 && blockIdx.y == 0) tile_parts[blockIdx.x] = (i + unit_k - 1) / unit_k;
        }
        __syncthreads();
        if (part > 0 && nact_s == 0) return;     // this tile has no offsets left for part z
    } else {
    // ---- prologue: which offsets does this tile use, and which of its 32-row groups
    for (int kk0 = 0; kk0 < nk; kk0 += 8) {
        int idx[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            idx[j] = -1;
            if (row_ok && kk0 + j < nk) idx[j] = nbr ? nbr[int64_t(k_begin + kk0 + j) * n_out + my_row] : my_row;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const unsigned long long b = __ballot(idx[j] >= 0);
            if (lane == 0 && wave < 2 && kk0 + j < nk)
                gbits[kk0 + j][wave] = (unsigned char)(((b & 0xFFFFFFFFull) != 0 ? 1 : 0) | ((b >> 32) != 0 ? 2 : 0));
        }
    }
    __syncthreads();
    if (tid == 0) {
        int n = 0;
        for (int kk = 0; kk < nk; ++kk) {
            const int gm = int(gbits[kk][0]) | (int(gbits[kk][1]) << 2);
            if (gm) { klist[n] = kk; kgm[n] = gm; ++n; }
        }
        nact_s = n;
    }
    __syncthreads();
    }   // !units
    const int nact = nact_s;
    const int nchunk = (cin + BK - 1) / BK;
    const int nstage = nact * nchunk;

    f32x16 acc[TN];
#pragma unroll
    for (int t = 0; t < TN; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    float4 pa[NA], pb[NB];
    const int a_sub = tid & 7, a_r = tid >> 3;

    // All loads below are UNCONDITIONAL (invalid lanes read a clamped, valid address and the value
    // is zeroed when it is written to LDS): straight-line code, so the compiler keeps the loads in
    // flight across the MFMA block instead of draining them at a control-flow join.
    unsigned pa_ok = 0, pb_ok = 0;
    auto fetch = [&](int slot, int c0, int par) {
        const int kk = klist[slot], gm = kgm[slot];
        const int c = c0 + a_sub * 4;
        pa_ok = 0;
#pragma unroll
        for (int p = 0; p < NA; ++p) {
            const int i = ridx[par][p * 32 + a_r];
            const bool ok = ((gm >> p) & 1) && i >= 0 && c < cin;
            pa[p] = *reinterpret_cast<const float4*>(in + (ok ? int64_t(i) * cin + c : 0));
            pa_ok |= (ok ? 1u : 0u) << p;
        }
        const float* wk = W + int64_t(k_begin + kk) * cin * cout;
        pb_ok = 0;
#pragma unroll
        for (int h = 0; h < NB; ++h) {
            const int f = tid + h * 256;
            const int r = f / BV, c4 = f - r * BV;
            const int cr = c0 + r, n = n0 + c4 * 4;
            const bool ok = f < BK * BV && cr < cin && n < cout;
            pb[h] = *reinterpret_cast<const float4*>(ok ? wk + int64_t(cr) * cout + n : W);
            pb_ok |= (ok ? 1u : 0u) << h;
        }
    };
    auto stash = [&](int slot) {
        const int gm = kgm[slot];
#pragma unroll
        for (int p = 0; p < NA; ++p) {
            if ((gm >> p) & 1) {
                const int row = p * 32 + a_r;
                const bool ok = (pa_ok >> p) & 1;
                As[row][a_sub * 4 + 0] = ok ? pa[p].x : 0.f;
                As[row][a_sub * 4 + 1] = ok ? pa[p].y : 0.f;
                As[row][a_sub * 4 + 2] = ok ? pa[p].z : 0.f;
                As[row][a_sub * 4 + 3] = ok ? pa[p].w : 0.f;
            }
        }
#pragma unroll
        for (int h = 0; h < NB; ++h) {
            const int f = tid + h * 256;
            const int r = f / BV, c4 = f - r * BV;
            if (f < BK * BV) {
                const bool ok = (pb_ok >> h) & 1;
                float4 v = pb[h];
                if (!ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
                *reinterpret_cast<float4*>(&Bs[r][c4 * 4]) = v;
            }
        }
    };

    if (nstage > 0) {
        // stage s = (slot, chunk ch).  Gathered rows + weights are fetched one stage ahead into
        // registers; the row indices of stage s+2 are fetched during stage s.  That index load is
        // issued through inline asm: hipcc's waitcnt pass otherwise drains every load in flight
        // around a loop-carried load destination (vmcnt is in-order), which serialises the prefetch.
        // The value is consumed in the same iteration, after an explicit vmcnt(0).
        auto advance = [&](int& sl, int& c) { if (++c == nchunk) { c = 0; ++sl; } };
        int slot = 0;
        int slot1 = 0, ch1 = 0; advance(slot1, ch1);
        int slot2 = slot1, ch2 = ch1; advance(slot2, ch2);
        auto idx_ptr = [&](int sl) -> const int32_t* {
            return nbr + (row_ok ? int64_t(k_begin + klist[sl]) * n_out + my_row : 0);
        };
        if (tid < BM) ridx[0][tid] = row_ok ? (nbr ? *idx_ptr(0) : my_row) : -1;
        __syncthreads();
        fetch(0, 0, 0);
        const int i1 = nbr ? *idx_ptr(nstage > 1 ? slot1 : 0) : my_row;
        stash(0);
        if (tid < BM) ridx[1][tid] = row_ok ? i1 : -1;
        __syncthreads();
        for (int s = 0; s < nstage; ++s) {
            const bool more = s + 1 < nstage, more2 = s + 2 < nstage;
            if (more) fetch(slot1, ch1 * BK, (s + 1) & 1);            // loads in flight during the MFMAs
            int i2 = my_row;
            if (nbr) {
                const int32_t* p2 = idx_ptr(more2 ? slot2 : slot);
                asm volatile("global_load_dword %0, %1, off" : "=v"(i2) : "v"(p2) : "memory");
            }
            if ((kgm[slot] >> wm) & 1) {
                // operands of step kk+1 are read from LDS while the MFMAs of step kk issue
                // (rolling two-step register window; the compiler otherwise waits lgkmcnt(0) per MFMA pair)
                const int arow = wm * 32 + (lane & 31);
                const int kh = lane >> 5;
                const int bcol = wn * TN * 32 + (lane & 31);
                float av[2], bv[2][TN];
                av[0] = As[arow][kh];
#pragma unroll
                for (int t = 0; t < TN; ++t) bv[0][t] = Bs[kh][bcol + t * 32];
#pragma unroll
                for (int kk = 0; kk < BK / 2; ++kk) {
                    const int cur = kk & 1, nxt = cur ^ 1;
                    if (kk + 1 < BK / 2) {
                        av[nxt] = As[arow][2 * (kk + 1) + kh];
#pragma unroll
                        for (int t = 0; t < TN; ++t) bv[nxt][t] = Bs[2 * (kk + 1) + kh][bcol + t * 32];
                    }
                    __builtin_amdgcn_sched_barrier(0);       // keep the next step's reads ahead of these MFMAs
#pragma unroll
                    for (int t = 0; t < TN; ++t)
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur], bv[cur][t], acc[t], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            __syncthreads();                  // stage s consumed; ridx[s&1] is free (its fetch ran an iteration ago)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // covers the asm index load above
            if (more) stash(slot1);
            if (more2 && tid < BM) ridx[s & 1][tid] = row_ok ? i2 : -1;       // stage s+2 reuses stage s' slot
            __syncthreads();
            slot = slot1;
            slot1 = slot2; ch1 = ch2;
            advance(slot2, ch2);
        }
    }

    float* dst = out;
    bool direct = !to_partial;                   // rows go to their final place (through out_rows)
    if (units) {
        if (part > 0) { dst = extra + int64_t(part - 1) * n_out * cout; direct = false; }
    } else if (to_partial) {
        dst = out + int64_t(blockIdx.z) * n_out * cout;
    }
#pragma unroll
    for (int t = 0; t < TN; ++t) {
        const int col = n0 + (wn * TN + t) * 32 + (lane & 31);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = row0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (row < n_out && col < cout) {
                const int orow = (direct && out_rows) ? out_rows[row] : row;
                dst[int64_t(orow) * cout + col] = acc[t][r];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// Split-bf16 variant of the pipelined kernel: identical tiling / pipeline / units logic, but the
// contraction runs on v_mfma_f32_32x32x16_bf16 (16x the fp32 MFMA rate).  Each fp32 operand is
// split into three bf16 pieces (x = x1 + x2 + x3) and the six significant cross products are
// accumulated in fp32 ("bf16x6"): fp32-level accuracy at 2.7x the fp32 MFMA throughput.  The
// weights arrive pre-split and k-contiguous (osn_weight_prep_x6); the gathered rows stay fp32 in
// LDS and are split in registers right before the MFMAs (VALU work that overlaps the matrix pipe).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int WM, int WN, int TN>
__global__ __launch_bounds__(256, 3) void spconv_fwd_x6_kernel(const float* __restrict__ in, const __bf16* __restrict__ Wp,
                                                              const int32_t* __restrict__ nbr,
                                                              const int32_t* __restrict__ out_rows,
                                                              const uint32_t* __restrict__ gmask,
                                                              int32_t* __restrict__ tile_parts,
                                                              float* __restrict__ out, float* __restrict__ extra,
                                                              int n_out, int K, int cin, int cinp, int cout,
                                                              int k_per_split, int to_partial, int unit_k) {
    static_assert(WM * WN == 4, "four waves per workgroup");
    constexpr int BM = 32 * WM;
    constexpr int BN = 32 * TN * WN;
    constexpr int BK = 32;
    constexpr int KC = 32;                       // offsets per block (k_per_split <= KC)
    constexpr int NA = BM / 32;                  // A float4 per thread per stage
    constexpr int NBV = 3 * BN * 4;              // B: 3 planes x BN rows x 4 x (8 bf16 = 16 bytes)
    constexpr int NB = (NBV + 255) / 256;        // B 16-byte pieces per thread per stage
    constexpr int LDA = BK + 4;                  // fp32 A rows of 144 B: aligned + conflict-free ds_read_b128
    constexpr int LDB = BK + 8;                  // bf16 B rows of 80 B:  aligned + conflict-free ds_read_b128
    __shared__ __attribute__((aligned(16))) float As[BM][LDA];
    __shared__ __attribute__((aligned(16))) __bf16 Bp[3][BN][LDB];
    __shared__ int ridx[2][BM];                  // row indices of the next two stages (by stage parity)
    __shared__ unsigned char gbits[KC][2];
    __shared__ int klist[KC];
    __shared__ int kgm[KC];
    __shared__ int nact_s;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int row0 = blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;
    // Two ways to cut the offset loop across blockIdx.z:
    //  * units mode (gmask given): part z takes the z-th group of `unit_k` ACTIVE offsets of this tile
    //    (work per block is bounded and even; parts > 0 go to `extra`, summed in order by a fix-up pass);
    //  * uniform mode: part z takes offsets [z*k_per_split, (z+1)*k_per_split) (small maps).
    const bool units = gmask != nullptr;
    const int part = blockIdx.z;
    const int k_begin = units ? 0 : part * k_per_split;
    const int nk = units ? K : min(K, k_begin + k_per_split) - k_begin;     // <= KC
    const int my_row = row0 + tid;                               // meaningful for tid < BM
    const bool row_ok = tid < BM && my_row < n_out;
    __shared__ uint32_t gm_s[4];

    if (units) {
        // ---- prologue from the precomputed 32-row group masks: no table scan
        if (tid < 4) {
            const int64_t g = int64_t(blockIdx.x) * WM + tid;
            gm_s[tid] = (tid < WM && g * 32 < n_out) ? gmask[g] : 0u;
        }
        __syncthreads();
        if (tid == 0) {
            const uint32_t tm = gm_s[0] | gm_s[1] | gm_s[2] | gm_s[3];
            int i = 0, n = 0;
            for (int kk = 0; kk < nk; ++kk) {
                if (!((tm >> kk) & 1u)) continue;
                if (i >= part * unit_k && i < (part + 1) * unit_k) {
                    klist[n] = kk;
                    kgm[n] = int((gm_s[0] >> kk) & 1u) | (int((gm_s[1] >> kk) & 1u) << 1) |
                             (int((gm_s[2] >> kk) & 1u) << 2) | (int((gm_s[3] >> kk) & 1u) << 3);
                    ++n;
                }
                ++i;
            }
            nact_s = n;
            if (part == 0 && blockIdx.y == 0) tile_parts[blockIdx.x] = (i + unit_k - 1) / unit_k;
        }
        __syncthreads();
        if (part > 0 && nact_s == 0) return;     // this tile has no offsets left for part z
    } else {
    // ---- prologue: which offsets does this tile use, and which of its 32-row groups
    for (int kk0 = 0; kk0 < nk; kk0 += 8) {
        int idx[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            idx[j] = -1;
            if (row_ok && kk0 + j < nk) idx[j] = nbr ? nbr[int64_t(k_begin + kk0 + j) * n_out + my_row] : my_row;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const unsigned long long b = __ballot(idx[j] >= 0);
            if (lane == 0 && wave < 2 && kk0 + j < nk)
                gbits[kk0 + j][wave] = (unsigned char)(((b & 0xFFFFFFFFull) != 0 ? 1 : 0) | ((b >> 32) != 0 ? 2 : 0));
        }
    }
    __syncthreads();
    if (tid == 0) {
        int n = 0;
        for (int kk = 0; kk < nk; ++kk) {
            const int gm = int(gbits[kk][0]) | (int(gbits[kk][1]) << 2);
            if (gm) { klist[n] = kk; kgm[n] = gm; ++n; }
        }
        nact_s = n;
    }
    __syncthreads();
    }   // !units
    const int nact = nact_s;
    const int nchunk = (cin + BK - 1) / BK;
    const int nstage = nact * nchunk;

    f32x16 acc[TN];
#pragma unroll
    for (int t = 0; t < TN; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    float4 pa[NA];
    uint4 pb[NB];
    const int a_sub = tid & 7, a_r = tid >> 3;

    // All loads below are UNCONDITIONAL (invalid lanes read a clamped, valid address and the value
    // is zeroed when it is written to LDS): straight-line code, so the compiler keeps the loads in
    // flight across the MFMA block instead of draining them at a control-flow join.
    // Weight staging: everything that depends only on the thread (plane, column, 8-element piece) is
    // computed once; per stage the source is base + one wave-uniform offset (k, chunk) -> one VALU add.
    int b_off[NB];                               // element offsets into Wp
    const int b_lds0 = (tid >> 2) * LDB + 8 * (tid & 3);      // piece f = tid + 256 h lands at b_lds0 + h * 64 * LDB
    unsigned b_in = 0, b_okm = 0;                // piece exists in the tile / its column exists in the weight
#pragma unroll
    for (int h = 0; h < NB; ++h) {
        const int f = tid + h * 256;
        const int pl = f / (BN * 4), rem = f - pl * (BN * 4);
        const int nn = rem >> 2, j = rem & 3;
        const bool in_tile = f < NBV;
        const bool ok = in_tile && n0 + nn < cout;
        b_off[h] = ok ? ((pl * K) * cout + n0 + nn) * cinp + 8 * j : 0;
        b_in |= (in_tile ? 1u : 0u) << h;
        b_okm |= (ok ? 1u : 0u) << h;
    }
    const int w_k_stride = cout * cinp;
    unsigned pa_ok = 0;
    auto fetch = [&](int slot, int c0, int par) {
        const int kk = __builtin_amdgcn_readfirstlane(klist[slot]), gm = kgm[slot];
        const int c = c0 + a_sub * 4;
        pa_ok = 0;
        int ri[NA];
#pragma unroll
        for (int p = 0; p < NA; ++p) ri[p] = ridx[par][p * 32 + a_r];      // all LDS reads first: one wait
#pragma unroll
        for (int p = 0; p < NA; ++p) {
            const bool ok = ((gm >> p) & 1) && ri[p] >= 0 && c < cin;
            // select the operands, not the address: one unsigned 32x32->64 multiply-add, no exec-masked branch
            const unsigned iu = ok ? unsigned(ri[p]) : 0u, cu = ok ? unsigned(c) : 0u;
            pa[p] = *reinterpret_cast<const float4*>(in + (uint64_t(iu) * unsigned(cin) + cu));
            pa_ok |= (ok ? 1u : 0u) << p;
        }
        // weights: pre-split bf16 planes, k-contiguous rows  Wp[plane][K][cout][cinp]  (cinp % 32 == 0, so a
        // 32-channel chunk never runs off a row; lanes without a piece read plane 0 / column 0: valid memory)
        const unsigned s_off = unsigned((k_begin + kk) * w_k_stride + c0);
#pragma unroll
        for (int h = 0; h < NB; ++h)
            pb[h] = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(Wp) +
                                                    size_t(2u * (unsigned(b_off[h]) + s_off)));
    };
    auto stash = [&](int slot) {
        const int gm = kgm[slot];
#pragma unroll
        for (int p = 0; p < NA; ++p) {
            if ((gm >> p) & 1) {
                const bool ok = (pa_ok >> p) & 1;
                float4 v = pa[p];
                if (!ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
                *reinterpret_cast<float4*>(&As[p * 32 + a_r][a_sub * 4]) = v;
            }
        }
#pragma unroll
        for (int h = 0; h < NB; ++h) {
            if ((b_in >> h) & 1) {
                uint4 v = pb[h];
                if (!((b_okm >> h) & 1)) v = make_uint4(0, 0, 0, 0);
                *reinterpret_cast<uint4*>(&Bp[0][0][0] + b_lds0 + h * 64 * LDB) = v;
            }
        }
    };

    if (nstage > 0) {
        // stage s = (slot, chunk ch).  Gathered rows + weights are fetched one stage ahead into
        // registers; the row indices of stage s+2 are fetched during stage s.  That index load is
        // issued through inline asm: hipcc's waitcnt pass otherwise drains every load in flight
        // around a loop-carried load destination (vmcnt is in-order), which serialises the prefetch.
        // The value is consumed in the same iteration, after an explicit vmcnt(0).
        auto advance = [&](int& sl, int& c) { if (++c == nchunk) { c = 0; ++sl; } };
        int slot = 0;
        int slot1 = 0, ch1 = 0; advance(slot1, ch1);
        int slot2 = slot1, ch2 = ch1; advance(slot2, ch2);
        auto idx_ptr = [&](int sl) -> const int32_t* {
            return nbr + (row_ok ? int64_t(k_begin + klist[sl]) * n_out + my_row : 0);
        };
        if (tid < BM) ridx[0][tid] = row_ok ? (nbr ? *idx_ptr(0) : my_row) : -1;
        __syncthreads();
        fetch(0, 0, 0);
        const int i1 = nbr ? *idx_ptr(nstage > 1 ? slot1 : 0) : my_row;
        stash(0);
        if (tid < BM) ridx[1][tid] = row_ok ? i1 : -1;
        __syncthreads();
        for (int s = 0; s < nstage; ++s) {
            const bool more = s + 1 < nstage, more2 = s + 2 < nstage;
            if (more) fetch(slot1, ch1 * BK, (s + 1) & 1);            // loads in flight during the MFMAs
            int i2 = my_row;
            if (nbr) {
                const int32_t* p2 = idx_ptr(more2 ? slot2 : slot);
                asm volatile("global_load_dword %0, %1, off" : "=v"(i2) : "v"(p2) : "memory");
            }
            if ((kgm[slot] >> wm) & 1) {
                // fp32-equivalent product from six bf16 MFMAs: x = x1 + x2 + x3 (8 mantissa bits each),
                // a*b ~= a1b1 + a1b2 + a2b1 + a1b3 + a2b2 + a3b1  (dropped terms <= 2^-24 relative).
                // The A rows of BOTH 16-deep k-steps are read up front (the two LDS round trips overlap) and the
                // VALU split of step 1 is scheduled BETWEEN the MFMAs of step 0 (sched_group_barrier), so the
                // matrix pipe is fed while the VALU converts instead of idling for ~40 VALU per k-step.
                const int arow = wm * 32 + (lane & 31);
                const int kq = 8 * (lane >> 5);
                const int bcol = wn * TN * 32 + (lane & 31);
                static_assert(BK == 32, "two k-steps per stage");
                float fv[2][8];
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const float4 f0 = *reinterpret_cast<const float4*>(&As[arow][ks * 16 + kq]);
                    const float4 f1 = *reinterpret_cast<const float4*>(&As[arow][ks * 16 + kq + 4]);
                    fv[ks][0] = f0.x; fv[ks][1] = f0.y; fv[ks][2] = f0.z; fv[ks][3] = f0.w;
                    fv[ks][4] = f1.x; fv[ks][5] = f1.y; fv[ks][6] = f1.z; fv[ks][7] = f1.w;
                }
                bf16x8 a1[2], a2[2], a3[2];
                auto split = [&](int ks) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const __bf16 h1 = (__bf16)fv[ks][e];
                        const float r1 = fv[ks][e] - (float)h1;
                        const __bf16 h2 = (__bf16)r1;
                        const float r2 = r1 - (float)h2;
                        a1[ks][e] = h1; a2[ks][e] = h2; a3[ks][e] = (__bf16)r2;
                    }
                };
                auto mfmas = [&](int ks) {
#pragma unroll
                    for (int t = 0; t < TN; ++t) {
                        const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(&Bp[0][bcol + t * 32][ks * 16 + kq]);
                        const bf16x8 b2 = *reinterpret_cast<const bf16x8*>(&Bp[1][bcol + t * 32][ks * 16 + kq]);
                        const bf16x8 b3 = *reinterpret_cast<const bf16x8*>(&Bp[2][bcol + t * 32][ks * 16 + kq]);
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[ks], b1, acc[t], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2[ks], b2, acc[t], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[ks], b3, acc[t], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2[ks], b1, acc[t], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[ks], b2, acc[t], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[ks], b1, acc[t], 0, 0, 0);
                    }
                };
                split(0);
                // k-step 0: after the 2nd MFMA of each column block (and once more) convert one PAIR of step-1
                // elements; sched_barrier pins the order so the VALU issues in the shadow of the running MFMAs
                auto split_pair = [&](int pr) {
#pragma unroll
                    for (int e = 2 * pr; e < 2 * pr + 2; ++e) {
                        const __bf16 h1 = (__bf16)fv[1][e];
                        const float r1 = fv[1][e] - (float)h1;
                        const __bf16 h2 = (__bf16)r1;
                        const float r2 = r1 - (float)h2;
                        a1[1][e] = h1; a2[1][e] = h2; a3[1][e] = (__bf16)r2;
                    }
                };
                int pr = 0;
#pragma unroll
                for (int t = 0; t < TN; ++t) {
                    const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(&Bp[0][bcol + t * 32][kq]);
                    const bf16x8 b2 = *reinterpret_cast<const bf16x8*>(&Bp[1][bcol + t * 32][kq]);
                    const bf16x8 b3 = *reinterpret_cast<const bf16x8*>(&Bp[2][bcol + t * 32][kq]);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[0], b1, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2[0], b2, acc[t], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (pr < 4) split_pair(pr++);
                    __builtin_amdgcn_sched_barrier(0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[0], b3, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2[0], b1, acc[t], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (TN < 4 && t < 4 - TN && pr < 4) split_pair(pr++);      // TN < 4: a second pair in the first blocks
                    __builtin_amdgcn_sched_barrier(0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[0], b2, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[0], b1, acc[t], 0, 0, 0);
                }
#pragma unroll
                for (; pr < 4; ++pr) split_pair(pr);                         // TN = 1 / 2: the rest, not hidden
                __builtin_amdgcn_sched_barrier(0);
                mfmas(1);
            }
            __syncthreads();                  // stage s consumed; ridx[s&1] is free (its fetch ran an iteration ago)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // covers the asm index load above
            if (more) stash(slot1);
            if (more2 && tid < BM) ridx[s & 1][tid] = row_ok ? i2 : -1;       // stage s+2 reuses stage s' slot
            __syncthreads();
            slot = slot1;
            slot1 = slot2; ch1 = ch2;
            advance(slot2, ch2);
        }
    }

    float* dst = out;
    bool direct = !to_partial;                   // rows go to their final place (through out_rows)
    if (units) {
        if (part > 0) { dst = extra + int64_t(part - 1) * n_out * cout; direct = false; }
    } else if (to_partial) {
        dst = out + int64_t(blockIdx.z) * n_out * cout;
    }
#pragma unroll
    for (int t = 0; t < TN; ++t) {
        const int col = n0 + (wn * TN + t) * 32 + (lane & 31);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = row0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (row < n_out && col < cout) {
                const int orow = (direct && out_rows) ? out_rows[row] : row;
                dst[int64_t(orow) * cout + col] = acc[t][r];
            }
        }
    }
}

__global__ void reduce_partial_rows_kernel(const float* __restrict__ partial, int S, int64_t n_out, int cout,
                                           const int32_t* __restrict__ out_rows, float* __restrict__ out) {
    const int64_t total = n_out * cout;
    for (int64_t e = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; e < total; e += int64_t(gridDim.x) * blockDim.x) {
        float s = 0.f;
        for (int z = 0; z < S; ++z) s += partial[int64_t(z) * total + e];
        if (out_rows) {
            const int64_t r = e / cout, c = e % cout;
            out[int64_t(out_rows[r]) * cout + c] = s;
        } else {
            out[e] = s;
        }
    }
}

// units mode: out[row] += extra[0][j] + extra[1][j] + ... (parts 1 .. tile_parts-1 of the row's tile, in order)
__global__ void fixup_units_kernel(const float* __restrict__ extra, const int32_t* __restrict__ tile_parts, int bm,
                                   int64_t n_out, int cout, const int32_t* __restrict__ out_rows,
                                   float* __restrict__ out) {
    const int64_t total = n_out * cout;
    for (int64_t e = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; e < total; e += int64_t(gridDim.x) * blockDim.x) {
        const int64_t j = e / cout;
        const int np = tile_parts[j / bm];
        if (np <= 1) continue;
        const int64_t c = e - j * cout;
        const int64_t o = (out_rows ? int64_t(out_rows[j]) : j) * cout + c;
        float s = out[o];
        for (int p = 1; p < np; ++p) s += extra[int64_t(p - 1) * total + e];
        out[o] = s;
    }
}

// One launch fills the forward planes (Wf, nullable) and/or the input-gradient planes (Wb, nullable).
__global__ void weight_prep_x6_kernel(const float* __restrict__ W, int K, int cin, int cout, int flip_b,
                                      int64_t per_plane_f, int64_t per_plane_b, __bf16* __restrict__ Wf,
                                      __bf16* __restrict__ Wb, int flip_f, int dgrad_f) {
    const int64_t total = per_plane_f + per_plane_b;
    for (int64_t e = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; e < total; e += int64_t(gridDim.x) * blockDim.x) {
        if (e < per_plane_f) weight_prep_x6_one(W, K, cin, cout, flip_f, dgrad_f, e, Wf);
        else weight_prep_x6_one(W, K, cin, cout, flip_b, 1, e - per_plane_f, Wb);
    }
}

__global__ void weight_transpose_kernel(const float* __restrict__ W, int K, int cin, int cout, int flip,
                                        float* __restrict__ Wt) {
    const int64_t total = int64_t(K) * cin * cout;
    for (int64_t e = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; e < total; e += int64_t(gridDim.x) * blockDim.x) {
        // e indexes Wt [K][cout][cin]
        const int ci = int(e % cin);
        const int co = int((e / cin) % cout);
        const int k = int(e / (int64_t(cin) * cout));
        const int ks = flip ? K - 1 - k : k;
        Wt[e] = W[(int64_t(ks) * cin + ci) * cout + co];
    }
}

// ------------------------------------------------------------- weight grad ----
// gW[k] (CI_T x CO_T tile) = sum over the valid (in,out) pairs of offset k inside the
// block's row range.  The pairs are compacted in LDS in a deterministic (ballot/prefix)
// order => exact work, no zero rows fed to the MFMA.  32 pairs per stage; the next
// stage's rows are fetched into registers while the MFMAs of the current one run
// (global -> VGPR -> LDS software pipeline, one barrier pair per stage).  The 32x32
// output tiles of the block are dealt round-robin to the four waves.
constexpr int WG_T = 128;      // max tile edge (input / output channels per block)
constexpr int WG_RB = 32;      // pairs per stage
constexpr int WG_SUB = 1024;   // rows compacted at a time

template <int TPW, bool AVEC, bool GVEC>   // 32x32 tiles per wave (1..4); 16-byte loads possible for in / gout rows
__global__ __launch_bounds__(256, 2) void spconv_wgrad_kernel(const float* __restrict__ in, const float* __restrict__ gout,
                                                           const int32_t* __restrict__ nbr, float* __restrict__ dst,
                                                           const int* __restrict__ items, int n_out, int K, int cin,
                                                           int cout, int n_co_blocks, int ci_t, int co_t) {
    __shared__ __attribute__((aligned(16))) float As[WG_RB][WG_T];
    __shared__ __attribute__((aligned(16))) float Gs[WG_RB][WG_T];
    __shared__ int list_o[WG_SUB];
    __shared__ int list_i[WG_SUB];
    __shared__ int wcnt[WG_SUB / 256][4];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ci0 = (blockIdx.x / n_co_blocks) * ci_t;
    const int co0 = (blockIdx.x % n_co_blocks) * co_t;
    // work item = (kernel offset, row range); items are balanced by pair count on the device
    const int k = items[blockIdx.y * 4 + 0];
    if (k < 0) return;
    const int r_begin = items[blockIdx.y * 4 + 1];
    const int r_end = items[blockIdx.y * 4 + 2];
    constexpr bool a_vec = AVEC, g_vec = GVEC;
    const int nti = (min(ci_t, cin - ci0) + 31) >> 5;       // live 32-row tiles along the input channels
    const int ntj = (min(co_t, cout - co0) + 31) >> 5;
    const int ntiles = nti * ntj;

    // staging geometry: A rows hold ci_t floats (ci_t/4 float4), G rows co_t floats
    const int a_v = ci_t >> 2, g_v = co_t >> 2;               // float4 per row
    const int a_total = WG_RB * a_v, g_total = WG_RB * g_v;   // <= 1024 each  (<= 4 per thread)

    f32x16 acc[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    float4 pa[4], pg[4];
    // loop-invariant staging coordinates of this thread's four A / four G float4 slots
    int aj[4], ac[4], gj[4], gc[4];
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        const int f = tid + h * 256;
        aj[h] = f < a_total ? f / a_v : -1;
        ac[h] = f < a_total ? (f - aj[h] * a_v) * 4 : 0;
        gj[h] = f < g_total ? f / g_v : -1;
        gc[h] = f < g_total ? (f - gj[h] * g_v) * 4 : 0;
    }

    // Vector path: unconditional loads from clamped addresses (zeroed at LDS-write time) so that the
    // prefetch stays in flight across the MFMA block; the scalar path (channels % 4 != 0) keeps guards.
    unsigned a_ok = 0, g_ok = 0;
    auto fetch = [&](int p0, int cnt) {
        a_ok = 0; g_ok = 0;
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            {
                const int p = p0 + max(aj[h], 0), c = ci0 + ac[h];
                const bool ok = aj[h] >= 0 && p < cnt && c < cin;
                if (a_vec) {
                    pa[h] = *reinterpret_cast<const float4*>(in + (ok ? int64_t(list_i[p]) * cin + c : 0));
                } else {
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (ok) {
                        const float* src = in + int64_t(list_i[p]) * cin + c;
                        v.x = src[0];
                        if (c + 1 < cin) v.y = src[1];
                        if (c + 2 < cin) v.z = src[2];
                        if (c + 3 < cin) v.w = src[3];
                    }
                    pa[h] = v;
                }
                a_ok |= (ok ? 1u : 0u) << h;
            }
            {
                const int p = p0 + max(gj[h], 0), c = co0 + gc[h];
                const bool ok = gj[h] >= 0 && p < cnt && c < cout;
                if (g_vec) {
                    pg[h] = *reinterpret_cast<const float4*>(gout + (ok ? int64_t(list_o[p]) * cout + c : 0));
                } else {
                    float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (ok) {
                        const float* src = gout + int64_t(list_o[p]) * cout + c;
                        w.x = src[0];
                        if (c + 1 < cout) w.y = src[1];
                        if (c + 2 < cout) w.z = src[2];
                        if (c + 3 < cout) w.w = src[3];
                    }
                    pg[h] = w;
                }
                g_ok |= (ok ? 1u : 0u) << h;
            }
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            // element-wise selects (a struct-level `c ? pa[h] : z` makes the arrays address-taken -> scratch)
            const bool oa = (a_ok >> h) & 1, og = (g_ok >> h) & 1;
            float4 va, vg;
            va.x = oa ? pa[h].x : 0.f; va.y = oa ? pa[h].y : 0.f; va.z = oa ? pa[h].z : 0.f; va.w = oa ? pa[h].w : 0.f;
            vg.x = og ? pg[h].x : 0.f; vg.y = og ? pg[h].y : 0.f; vg.z = og ? pg[h].z : 0.f; vg.w = og ? pg[h].w : 0.f;
            if (aj[h] >= 0) *reinterpret_cast<float4*>(&As[aj[h]][ac[h]]) = va;
            if (gj[h] >= 0) *reinterpret_cast<float4*>(&Gs[gj[h]][gc[h]]) = vg;
        }
    };

    for (int base = r_begin; base < r_end; base += WG_SUB) {
        // ---- compact the valid pairs of rows [base, base + WG_SUB): all table loads first
        // (unconditional, clamped), then one ballot round and two barriers for the whole chunk
        constexpr int NJ = WG_SUB / 256;
        int iv[NJ];
        bool vv[NJ];
        unsigned long long mm[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int o = base + j * 256 + tid;
            const bool in_range = o < r_end;
            iv[j] = nbr ? nbr[int64_t(k) * n_out + (in_range ? o : r_begin)] : o;
            vv[j] = in_range;
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            vv[j] = vv[j] && iv[j] >= 0;
            mm[j] = __ballot(vv[j]);
            if (lane == 0) wcnt[j][wave] = __popcll(mm[j]);
        }
        __syncthreads();
        int cnt = 0;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            int woff = 0, tot = 0;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const int c = wcnt[j][w];
                if (w < wave) woff += c;
                tot += c;
            }
            if (vv[j]) {
                const int pos = cnt + woff + __popcll(mm[j] & ((1ull << lane) - 1ull));
                list_o[pos] = base + j * 256 + tid;
                list_i[pos] = iv[j];
            }
            cnt += tot;
        }
        __syncthreads();
        if (cnt == 0) continue;
        // ---- pipelined reduction over the compacted pairs
        fetch(0, cnt);
        stash();
        __syncthreads();
        for (int p0 = 0; p0 < cnt; p0 += WG_RB) {
            const bool more = p0 + WG_RB < cnt;
            if (more) fetch(p0 + WG_RB, cnt);            // loads in flight during the MFMAs below
            const int kh = lane >> 5;
#pragma unroll
            for (int t = 0; t < TPW; ++t) {
                const int tile = wave + 4 * t;
                if (tile < ntiles) {
                    const int ti = tile / ntj, tj = tile - ti * ntj;
                    const int ca = ti * 32 + (lane & 31), cb = tj * 32 + (lane & 31);
                    // rolling operand window, WG_LOOK steps deep: the LDS reads of step kk + WG_LOOK - 1 are issued
                    // before the MFMA of step kk (one 32x32x2 MFMA = 64 cycles; a one-step lookahead left every MFMA
                    // waiting on lgkmcnt for an LDS round trip of 100+ cycles under load)
                    constexpr int WG_LOOK = 4;
                    float av[WG_LOOK], bv[WG_LOOK];
#pragma unroll
                    for (int d = 0; d < WG_LOOK - 1; ++d) {
                        av[d] = As[2 * d + kh][ca];
                        bv[d] = Gs[2 * d + kh][cb];
                    }
#pragma unroll
                    for (int kk = 0; kk < WG_RB / 2; ++kk) {
                        const int cur = kk % WG_LOOK, nxt = (kk + WG_LOOK - 1) % WG_LOOK;
                        if (kk + WG_LOOK - 1 < WG_RB / 2) {
                            av[nxt] = As[2 * (kk + WG_LOOK - 1) + kh][ca];
                            bv[nxt] = Gs[2 * (kk + WG_LOOK - 1) + kh][cb];
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur], bv[cur], acc[t], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            __syncthreads();                              // everyone is done reading this stage
            if (more) stash();
            __syncthreads();
        }
    }

    // ---- store: dst[item][ci][co]
    float* d = dst + int64_t(blockIdx.y) * cin * cout;
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const int tile = wave + 4 * t;
        if (tile < ntiles) {
            const int ti = tile / ntj, tj = tile - ti * ntj;
            const int co = co0 + tj * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ci = ci0 + ti * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (ci < cin && co < cout) d[int64_t(ci) * cout + co] = acc[t][r];
            }
        }
    }
}

// One block: split every offset's row range into a number of items proportional to its pair
// count (count_k = n_out when `counts` is null), at most T items in total.  items[t] =
// (k, row_begin, row_end, 0), k = -1 for unused slots; range[k] = (first item, last item + 1).
__global__ __launch_bounds__(256) void wgrad_plan_kernel(const long long* __restrict__ counts, int n_out, int K, int T,
                                                         int min_rows, int* __restrict__ items, int* __restrict__ range) {
    __shared__ long long total_s;
    __shared__ int sk[128];
    __shared__ int start[129];
    const int tid = threadIdx.x;
    if (tid == 0) {
        long long tot = 0;
        for (int k = 0; k < K; ++k) tot += counts ? counts[k] : (long long)n_out;
        total_s = tot;
    }
    __syncthreads();
    const long long total = total_s;
    long long quota = (T > K) ? (total + (T - K) - 1) / (T - K) : total;
    if (quota < 1) quota = 1;
    const int max_s = max(1, (n_out + min_rows - 1) / min_rows);
    if (tid < K) {
        const long long c = counts ? counts[tid] : (long long)n_out;
        int sp = c > 0 ? int((c + quota - 1) / quota) : 0;
        if (sp > max_s) sp = max_s;
        sk[tid] = sp;
    }
    __syncthreads();
    if (tid == 0) {
        int acc = 0;
        for (int k = 0; k < K; ++k) { start[k] = acc; acc += sk[k]; }
        start[K] = acc;
    }
    __syncthreads();
    if (tid < K) {
        range[2 * tid] = start[tid];
        range[2 * tid + 1] = start[tid + 1];
    }
    const int used = start[K];
    for (int t = tid; t < T; t += 256) {
        int k = -1, rb = 0, re = 0;
        if (t < used) {
            // offset of item t: the last k with start[k] <= t  (K <= 125: linear scan)
            k = 0;
            while (k + 1 < K && start[k + 1] <= t) ++k;
            const int j = t - start[k], sp = sk[k];
            const long long rows = n_out;
            rb = int(rows * j / sp);
            re = int(rows * (j + 1) / sp);
        }
        items[4 * t + 0] = k; items[4 * t + 1] = rb; items[4 * t + 2] = re; items[4 * t + 3] = 0;
    }
}

__global__ void reduce_items_kernel(const float* __restrict__ partial, const int* __restrict__ range, int K,
                                    int64_t per_k, float* __restrict__ out) {
    const int64_t total = int64_t(K) * per_k;
    for (int64_t e = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; e < total; e += int64_t(gridDim.x) * blockDim.x) {
        const int k = int(e / per_k);
        const int64_t r = e - int64_t(k) * per_k;
        float s = 0.f;
        for (int t = range[2 * k]; t < range[2 * k + 1]; ++t) s += partial[int64_t(t) * per_k + r];
        out[e] = s;
    }
}

__global__ void reduce_partial_flat_kernel(const float* __restrict__ partial, int S, int64_t total,
                                           float* __restrict__ out) {
    for (int64_t e = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; e < total; e += int64_t(gridDim.x) * blockDim.x) {
        float s = 0.f;
        for (int z = 0; z < S; ++z) s += partial[int64_t(z) * total + e];
        out[e] = s;
    }
}

// ------------------------------------------------------------ host planning ----
struct FwdPlan {
    int cfg;      // 0: BM128 (4,1,TN) ; 1: BM64 (2,2,TN) ; 2: BM32 (1,4,1)
    int tn;
    int bm, bn;
    int gx, gy, S, kps;
    int unit_k, unit_parts;    // units mode (needs gmask): active offsets per block, max parts per tile
};

constexpr int UNIT_K = 8;

static FwdPlan plan_fwd(int64_t n_out, int K, int cin, int cout) {
    FwdPlan p;
    const int ct = int(cdiv(cout, 32));
    if (n_out >= 32768 || ct == 1) {
        p.cfg = 0; p.tn = ct >= 4 ? 4 : ct; p.bm = 128; p.bn = 32 * p.tn;
    } else if (n_out >= 4096 || ct == 2) {
        p.cfg = 1; p.tn = ct >= 3 ? 2 : 1; p.bm = 64; p.bn = 64 * p.tn;
    } else {
        p.cfg = 2; p.tn = 1; p.bm = 32; p.bn = 128;
    }
    p.gx = int(cdiv(n_out, p.bm));
    p.gy = int(cdiv(cout, p.bn));
    const int64_t blocks = int64_t(p.gx) * p.gy;
    int S = 1;
    if (K > 1 && blocks < 384) {
        S = int(cdiv(768, blocks));
        if (S > 16) S = 16;
        if (S > K) S = K;
    }
    p.kps = int(cdiv(K, S));
    p.S = int(cdiv(K, p.kps));
    // units mode pays when the plain launch would not be split and every tile walks many offsets
    constexpr int tune_U = UNIT_K;
    p.unit_k = tune_U;
    // ONE eligibility rule for both pipelined kernels and for osn_spconv_fwd_ws_bytes (the fp32 pipe kernel
    // additionally needs cout % 4 == 0 to run at all; when it does not, the simple kernel ignores gmask)
    p.unit_parts = (p.S == 1 && K > tune_U && K <= 32 && cin > 4 && (cin & 3) == 0) ? int(cdiv(K, tune_U)) : 0;
    return p;
}

struct WgradPlan {
    int ci_t, co_t, n_ci, n_co, tpw, T, min_rows;
    size_t items_bytes, partial_bytes;
};

static int pad32(int c) { return (c + 31) / 32 * 32; }

static WgradPlan plan_wgrad(int64_t n_out, int K, int cin, int cout) {
    WgradPlan p;
    p.ci_t = pad32(cin) < WG_T ? pad32(cin) : WG_T;
    p.co_t = pad32(cout) < WG_T ? pad32(cout) : WG_T;
    p.n_ci = int(cdiv(cin, p.ci_t));
    p.n_co = int(cdiv(cout, p.co_t));
    const int tiles = (p.ci_t / 32) * (p.co_t / 32);
    p.tpw = (tiles + 3) / 4;
    p.min_rows = 512;
    // ~1024 workgroups per launch (4 per CU), but never more items than the rows can feed
    constexpr int tune_T = 512;
    int64_t T = cdiv(tune_T, int64_t(p.n_ci) * p.n_co);
    const int64_t tmax = int64_t(K) * cdiv(n_out, p.min_rows);
    if (T > tmax) T = tmax;
    if (T < K) T = K;
    if (T > 4096) T = 4096;
    p.T = int(T);
    p.items_bytes = align_up(size_t(p.T) * 16 + size_t(K) * 8, 256);
    p.partial_bytes = size_t(p.T) * size_t(cin) * size_t(cout) * 4;
    return p;
}

}  // namespace osn

using namespace osn;

static size_t units_ws_bytes(const FwdPlan& p, int64_t n_out, int cout) {
    if (p.unit_parts <= 1) return 0;
    return align_up(size_t(p.gx) * 4, 256) + size_t(p.unit_parts - 1) * size_t(n_out) * size_t(cout) * 4;
}

extern "C" size_t osn_spconv_fwd_ws_bytes(int64_t n_out, int K, int cin, int cout) {
    if (n_out <= 0) return 0;
    FwdPlan p = plan_fwd(n_out, K, cin, cout);
    const size_t uniform = p.S > 1 ? size_t(p.S) * size_t(n_out) * size_t(cout) * 4 : 0;
    const size_t units = units_ws_bytes(p, n_out, cout);       // only used when the caller passes gmask
    return uniform > units ? uniform : units;
}

extern "C" int osn_spconv_fwd_plan(int64_t n_out, int K, int cin, int cout, int32_t* plan6) {
    OSN_REQUIRE(plan6 && n_out >= 0 && K >= 1 && cin >= 1 && cout >= 1, OSN_E_ARG, "osn_spconv_fwd_plan: bad arguments");
    FwdPlan p = plan_fwd(n_out > 0 ? n_out : 1, K, cin, cout);
    static const int wm[3] = {4, 2, 1}, wn[3] = {1, 2, 4};
    plan6[0] = wm[p.cfg];
    plan6[1] = wn[p.cfg];
    plan6[2] = p.tn;
    plan6[3] = cin <= 4 ? 4 : 32;
    plan6[4] = p.S;
    plan6[5] = p.gx * p.gy * p.S;
    return OSN_OK;
}

struct UnitsArgs {
    const uint32_t* gmask;
    int32_t* tile_parts;
    float* extra;
};

template <int WM, int WN, int TN>
static void launch_fwd_pipe(const FwdPlan& p, hipStream_t st, const float* in, const float* W, const int32_t* nbr,
                            const int32_t* out_rows, float* dst, int n_out, int K, int cin, int cout,
                            const UnitsArgs& u) {
    const int gz = u.gmask ? p.unit_parts : p.S;
    hipLaunchKernelGGL((spconv_fwd_pipe_kernel<WM, WN, TN>), dim3(p.gx, p.gy, gz), dim3(256), 0, st, in, W, nbr,
                       out_rows, u.gmask, u.tile_parts, dst, u.extra, n_out, K, cin, cout, p.kps,
                       (!u.gmask && p.S > 1) ? 1 : 0, p.unit_k);
}

template <int WM, int WN, int TN>
static void launch_fwd_x6(const FwdPlan& p, hipStream_t st, const float* in, const __bf16* Wp, const int32_t* nbr,
                          const int32_t* out_rows, float* dst, int n_out, int K, int cin, int cinp, int cout,
                          const UnitsArgs& u) {
    const int gz = u.gmask ? p.unit_parts : p.S;
    hipLaunchKernelGGL((spconv_fwd_x6_kernel<WM, WN, TN>), dim3(p.gx, p.gy, gz), dim3(256), 0, st, in, Wp, nbr,
                       out_rows, u.gmask, u.tile_parts, dst, u.extra, n_out, K, cin, cinp, cout, p.kps,
                       (!u.gmask && p.S > 1) ? 1 : 0, p.unit_k);
}

template <int WM, int WN, int TN, int BK>
static void launch_fwd(const FwdPlan& p, hipStream_t st, const float* in, const float* W, const int32_t* nbr,
                       const int32_t* out_rows, float* dst, int n_out, int K, int cin, int cout) {
    hipLaunchKernelGGL((spconv_fwd_kernel<WM, WN, TN, BK>), dim3(p.gx, p.gy, p.S), dim3(256), 0, st, in, W, nbr,
                       out_rows, dst, n_out, K, cin, cout, p.kps, p.S > 1 ? 1 : 0);
}

extern "C" int osn_spconv_fwd(const float* in, const float* W, const int32_t* nbr, const int32_t* out_rows,
                              const uint32_t* gmask, float* out, int64_t n_out, int K, int cin, int cout, void* ws,
                              size_t ws_bytes, osn_stream_t stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    OSN_REQUIRE(n_out >= 0 && n_out < (int64_t(1) << 31), OSN_E_ARG, "osn_spconv_fwd: n_out out of range");
    OSN_REQUIRE(K >= 1 && cin >= 1 && cout >= 1, OSN_E_ARG, "osn_spconv_fwd: bad K/cin/cout (%d,%d,%d)", K, cin, cout);
    if (n_out == 0) return OSN_OK;
    OSN_REQUIRE(in && W && out, OSN_E_ARG, "osn_spconv_fwd: null pointer");
    OSN_REQUIRE(nbr || K == 1, OSN_E_ARG, "osn_spconv_fwd: nbr may be null only for K == 1 (identity map)");
    OSN_REQUIRE(aligned16(in) && aligned16(W) && aligned16(out), OSN_E_ARG, "osn_spconv_fwd: pointers must be 16-byte aligned");
    FwdPlan p = plan_fwd(n_out, K, cin, cout);
    float* dst = out;
    UnitsArgs ua = {nullptr, nullptr, nullptr};
    const bool pipe_ok = cin > 4 && (cin & 3) == 0 && (cout & 3) == 0 && p.kps <= 32;
    const bool use_units = gmask && nbr && p.unit_parts > 1 && pipe_ok;
    if (use_units) {
        const size_t need = units_ws_bytes(p, n_out, cout);
        OSN_REQUIRE(ws && ws_bytes >= need, OSN_E_WS, "osn_spconv_fwd: workspace %zu < %zu", ws_bytes, need);
        ua.gmask = gmask;
        ua.tile_parts = static_cast<int32_t*>(ws);
        ua.extra = reinterpret_cast<float*>(static_cast<char*>(ws) + align_up(size_t(p.gx) * 4, 256));
    } else if (p.S > 1) {
        const size_t need = size_t(p.S) * size_t(n_out) * size_t(cout) * 4;
        OSN_REQUIRE(ws && ws_bytes >= need, OSN_E_WS, "osn_spconv_fwd: workspace %zu < %zu", ws_bytes, need);
        dst = static_cast<float*>(ws);
    }
    const int n = int(n_out);
    const bool pipe = pipe_ok;
    if (pipe) {
        switch (p.cfg * 10 + p.tn) {
            case 1: launch_fwd_pipe<4, 1, 1>(p, st, in, W, nbr, out_rows, dst, n, K, cin, cout, ua); break;
            case 2: launch_fwd_pipe<4, 1, 2>(p, st, in, W, nbr, out_rows, dst, n, K, cin, cout, ua); break;
            case 3: launch_fwd_pipe<4, 1, 3>(p, st, in, W, nbr, out_rows, dst, n, K, cin, cout, ua); break;
            case 4: launch_fwd_pipe<4, 1, 4>(p, st, in, W, nbr, out_rows, dst, n, K, cin, cout, ua); break;
            case 11: launch_fwd_pipe<2, 2, 1>(p, st, in, W, nbr, out_rows, dst, n, K, cin, cout, ua); break;
            case 12: launch_fwd_pipe<2, 2, 2>(p, st, in, W, nbr, out_rows, dst, n, K, cin, cout, ua); break;
            default: launch_fwd_pipe<1, 4, 1>(p, st, in, W, nbr, out_rows, dst, n, K, cin, cout, ua); break;
        }
    } else if (cin <= 4) {
        // stem convolution (3 -> 32): 4-wide channel chunks
        switch (p.cfg * 10 + p.tn) {
            case 1: launch_fwd<4, 1, 1, 4>(p, st, in, W, nbr, out_rows, dst, n, K, cin, cout); break;
            case 2: launch_fwd<4, 1, 2, 4>(p, st, in, W, nbr, out_rows, dst, n, K, cin, cout); break;
            case 3: launch_fwd<4, 1, 3, 4>(p, st, in, W, nbr, out_rows, dst, n, K, cin, cout); break;
            case 4: launch_fwd<4, 1, 4, 4>(p, st, in, W, nbr, out_rows, dst, n, K, cin, cout); break;
            case 11: launch_fwd<2, 2, 1, 4>(p, st, in, W, nbr, out_rows, dst, n, K, cin, cout); break;
            case 12: launch_fwd<2, 2, 2, 4>(p, st, in, W, nbr, out_rows, dst, n, K, cin, cout); break;
            default: launch_fwd<1, 4, 1, 4>(p, st, in, W, nbr, out_rows, dst, n, K, cin, cout); break;
        }
    } else {
        switch (p.cfg * 10 + p.tn) {
            case 1: launch_fwd<4, 1, 1, 32>(p, st, in, W, nbr, out_rows, dst, n, K, cin, cout); break;
            case 2: launch_fwd<4, 1, 2, 32>(p, st, in, W, nbr, out_rows, dst, n, K, cin, cout); break;
            case 3: launch_fwd<4, 1, 3, 32>(p, st, in, W, nbr, out_rows, dst, n, K, cin, cout); break;
            case 4: launch_fwd<4, 1, 4, 32>(p, st, in, W, nbr, out_rows, dst, n, K, cin, cout); break;
            case 11: launch_fwd<2, 2, 1, 32>(p, st, in, W, nbr, out_rows, dst, n, K, cin, cout); break;
            case 12: launch_fwd<2, 2, 2, 32>(p, st, in, W, nbr, out_rows, dst, n, K, cin, cout); break;
            default: launch_fwd<1, 4, 1, 32>(p, st, in, W, nbr, out_rows, dst, n, K, cin, cout); break;
        }
    }
    OSN_LAUNCH_CHECK();
    if (use_units) {
        const int64_t total = n_out * cout;
        int g = int(cdiv(total, 256));
        if (g > 4096) g = 4096;
        hipLaunchKernelGGL(fixup_units_kernel, dim3(g), dim3(256), 0, st, ua.extra, ua.tile_parts, p.bm, n_out, cout,
                           out_rows, out);
        OSN_LAUNCH_CHECK();
    } else if (p.S > 1) {
        const int64_t total = n_out * cout;
        int g = int(cdiv(total, 256));
        if (g > 4096) g = 4096;
        hipLaunchKernelGGL(reduce_partial_rows_kernel, dim3(g), dim3(256), 0, st, dst, p.S, n_out, cout, out_rows, out);
        OSN_LAUNCH_CHECK();
    }
    return OSN_OK;
}

extern "C" size_t osn_weight_prep_x6_bytes(int K, int cin, int cout, int for_dgrad) {
    const int nn = for_dgrad ? cin : cout, nc = for_dgrad ? cout : cin;
    const int cp = (nc + 31) / 32 * 32;
    return size_t(3) * size_t(K) * size_t(nn) * size_t(cp) * 2;
}

extern "C" int osn_weight_prep_x6(const float* W, int K, int cin, int cout, int flip, int for_dgrad, void* Wp,
                                  osn_stream_t stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    OSN_REQUIRE(W && Wp && K >= 1 && cin >= 1 && cout >= 1, OSN_E_ARG, "osn_weight_prep_x6: bad arguments");
    const int nn = for_dgrad ? cin : cout, nc = for_dgrad ? cout : cin;
    const int cp = (nc + 31) / 32 * 32;
    const int64_t per_plane = int64_t(K) * nn * cp;
    int g = int(cdiv(per_plane, 256));
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(weight_prep_x6_kernel, dim3(g), dim3(256), 0, st, W, K, cin, cout, 0, per_plane, int64_t(0),
                       static_cast<__bf16*>(Wp), (__bf16*)nullptr, flip, for_dgrad);
    OSN_LAUNCH_CHECK();
    return OSN_OK;
}

extern "C" int osn_weight_prep_x6_pair(const float* W, int K, int cin, int cout, int flip, void* Wp_fwd, void* Wp_dgrad,
                                       osn_stream_t stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    OSN_REQUIRE(W && Wp_fwd && Wp_dgrad && K >= 1 && cin >= 1 && cout >= 1, OSN_E_ARG, "osn_weight_prep_x6_pair: bad arguments");
    const int64_t ppf = int64_t(K) * cout * ((cin + 31) / 32 * 32);
    const int64_t ppb = int64_t(K) * cin * ((cout + 31) / 32 * 32);
    int g = int(cdiv(ppf + ppb, 256));
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(weight_prep_x6_kernel, dim3(g), dim3(256), 0, st, W, K, cin, cout, flip, ppf, ppb,
                       static_cast<__bf16*>(Wp_fwd), static_cast<__bf16*>(Wp_dgrad), 0, 0);
    OSN_LAUNCH_CHECK();
    return OSN_OK;
}

extern "C" int osn_spconv_fwd_x6(const float* in, const void* Wp, const int32_t* nbr, const int32_t* out_rows,
                                 const uint32_t* gmask, float* out, int64_t n_out, int K, int cin, int cout, void* ws,
                                 size_t ws_bytes, osn_stream_t stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    OSN_REQUIRE(n_out >= 0 && n_out < (int64_t(1) << 31), OSN_E_ARG, "osn_spconv_fwd_x6: n_out out of range");
    OSN_REQUIRE(K >= 1 && cin >= 4 && (cin & 3) == 0 && cout >= 1, OSN_E_ARG,
                "osn_spconv_fwd_x6: needs cin %% 4 == 0 (K=%d cin=%d cout=%d)", K, cin, cout);
    OSN_REQUIRE(int64_t(3) * K * cout * ((cin + 31) / 32 * 32) < (int64_t(1) << 30), OSN_E_ARG,
                "osn_spconv_fwd_x6: prepared weight of %d x %d x %d exceeds the kernel's 32-bit offsets", K, cin, cout);
    if (n_out == 0) return OSN_OK;
    OSN_REQUIRE(in && Wp && out, OSN_E_ARG, "osn_spconv_fwd_x6: null pointer");
    OSN_REQUIRE(nbr || K == 1, OSN_E_ARG, "osn_spconv_fwd_x6: nbr may be null only for K == 1");
    OSN_REQUIRE(aligned16(in) && aligned16(Wp) && aligned16(out), OSN_E_ARG, "osn_spconv_fwd_x6: pointers must be 16-byte aligned");
    FwdPlan p = plan_fwd(n_out, K, cin, cout);
    OSN_REQUIRE(p.kps <= 32, OSN_E_ARG, "osn_spconv_fwd_x6: more than 32 offsets per block (K=%d)", K);
    float* dst = out;
    UnitsArgs ua = {nullptr, nullptr, nullptr};
    const bool use_units = gmask && nbr && p.unit_parts > 1;        // same rule osn_spconv_fwd_ws_bytes sizes for
    if (use_units) {
        const size_t need = units_ws_bytes(p, n_out, cout);
        OSN_REQUIRE(ws && ws_bytes >= need, OSN_E_WS, "osn_spconv_fwd_x6: workspace %zu < %zu", ws_bytes, need);
        ua.gmask = gmask;
        ua.tile_parts = static_cast<int32_t*>(ws);
        ua.extra = reinterpret_cast<float*>(static_cast<char*>(ws) + align_up(size_t(p.gx) * 4, 256));
    } else if (p.S > 1) {
        const size_t need = size_t(p.S) * size_t(n_out) * size_t(cout) * 4;
        OSN_REQUIRE(ws && ws_bytes >= need, OSN_E_WS, "osn_spconv_fwd_x6: workspace %zu < %zu", ws_bytes, need);
        dst = static_cast<float*>(ws);
    }
    const int n = int(n_out);
    const int cinp = (cin + 31) / 32 * 32;
    const __bf16* wp = static_cast<const __bf16*>(Wp);
    switch (p.cfg * 10 + p.tn) {
        case 1: launch_fwd_x6<4, 1, 1>(p, st, in, wp, nbr, out_rows, dst, n, K, cin, cinp, cout, ua); break;
        case 2: launch_fwd_x6<4, 1, 2>(p, st, in, wp, nbr, out_rows, dst, n, K, cin, cinp, cout, ua); break;
        case 3: launch_fwd_x6<4, 1, 3>(p, st, in, wp, nbr, out_rows, dst, n, K, cin, cinp, cout, ua); break;
        case 4: launch_fwd_x6<4, 1, 4>(p, st, in, wp, nbr, out_rows, dst, n, K, cin, cinp, cout, ua); break;
        case 11: launch_fwd_x6<2, 2, 1>(p, st, in, wp, nbr, out_rows, dst, n, K, cin, cinp, cout, ua); break;
        case 12: launch_fwd_x6<2, 2, 2>(p, st, in, wp, nbr, out_rows, dst, n, K, cin, cinp, cout, ua); break;
        default: launch_fwd_x6<1, 4, 1>(p, st, in, wp, nbr, out_rows, dst, n, K, cin, cinp, cout, ua); break;
    }
    OSN_LAUNCH_CHECK();
    const int64_t total = n_out * cout;
    int g = int(cdiv(total, 256));
    if (g > 4096) g = 4096;
    if (use_units) {
        hipLaunchKernelGGL(fixup_units_kernel, dim3(g), dim3(256), 0, st, ua.extra, ua.tile_parts, p.bm, n_out, cout,
                           out_rows, out);
        OSN_LAUNCH_CHECK();
    } else if (p.S > 1) {
        hipLaunchKernelGGL(reduce_partial_rows_kernel, dim3(g), dim3(256), 0, st, dst, p.S, n_out, cout, out_rows, out);
        OSN_LAUNCH_CHECK();
    }
    return OSN_OK;
}

extern "C" int osn_weight_transpose(const float* W, int K, int cin, int cout, int flip, float* Wt,
                                    osn_stream_t stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    OSN_REQUIRE(W && Wt && K >= 1 && cin >= 1 && cout >= 1, OSN_E_ARG, "osn_weight_transpose: bad arguments");
    const int64_t total = int64_t(K) * cin * cout;
    int g = int(cdiv(total, 256));
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(weight_transpose_kernel, dim3(g), dim3(256), 0, st, W, K, cin, cout, flip, Wt);
    OSN_LAUNCH_CHECK();
    return OSN_OK;
}

extern "C" size_t osn_spconv_wgrad_ws_bytes(int64_t n_out, int K, int cin, int cout) {
    if (n_out <= 0) return 0;
    WgradPlan p = plan_wgrad(n_out, K, cin, cout);
    return p.items_bytes + p.partial_bytes;
}

extern "C" size_t osn_spconv_wgrad_items_bytes(int64_t n_out, int K, int cin, int cout) {
    if (n_out <= 0 || K < 1 || cin < 1 || cout < 1) return 0;
    return plan_wgrad(n_out, K, cin, cout).items_bytes;
}

extern "C" int osn_spconv_wgrad_plan(const int64_t* counts, int64_t n_out, int K, int cin, int cout, int32_t* items,
                                     osn_stream_t stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    OSN_REQUIRE(n_out > 0 && n_out < (int64_t(1) << 31) && K >= 1 && K <= 125 && cin >= 1 && cout >= 1 && items, OSN_E_ARG,
                "osn_spconv_wgrad_plan: bad arguments");
    WgradPlan p = plan_wgrad(n_out, K, cin, cout);
    hipLaunchKernelGGL(wgrad_plan_kernel, dim3(1), dim3(256), 0, st, reinterpret_cast<const long long*>(counts),
                       int(n_out), K, p.T, p.min_rows, items, items + size_t(p.T) * 4);
    OSN_LAUNCH_CHECK();
    return OSN_OK;
}

extern "C" int osn_spconv_wgrad(const float* in, const float* gout, const int32_t* nbr, const int64_t* counts,
                                const int32_t* plan_items, float* gW, int64_t n_out, int K, int cin, int cout, void* ws,
                                size_t ws_bytes, osn_stream_t stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    OSN_REQUIRE(n_out >= 0 && n_out < (int64_t(1) << 31), OSN_E_ARG, "osn_spconv_wgrad: n_out out of range");
    OSN_REQUIRE(K >= 1 && K <= 125 && cin >= 1 && cout >= 1 && gW, OSN_E_ARG, "osn_spconv_wgrad: bad arguments");
    const int64_t wtotal = int64_t(K) * cin * cout;
    if (n_out == 0) {
        OSN_HIP(hipMemsetAsync(gW, 0, size_t(wtotal) * 4, st));
        return OSN_OK;
    }
    OSN_REQUIRE(in && gout, OSN_E_ARG, "osn_spconv_wgrad: null pointer");
    OSN_REQUIRE(nbr || K == 1, OSN_E_ARG, "osn_spconv_wgrad: nbr may be null only for K == 1");
    OSN_REQUIRE(aligned16(in) && aligned16(gout) && aligned16(gW), OSN_E_ARG, "osn_spconv_wgrad: pointers must be 16-byte aligned");
    WgradPlan p = plan_wgrad(n_out, K, cin, cout);
    OSN_REQUIRE(ws && ws_bytes >= p.items_bytes + p.partial_bytes, OSN_E_WS, "osn_spconv_wgrad: workspace %zu < %zu",
                ws_bytes, p.items_bytes + p.partial_bytes);
    const int* items = plan_items ? plan_items : static_cast<int*>(ws);
    const int* range = items + size_t(p.T) * 4;
    float* partial = reinterpret_cast<float*>(static_cast<char*>(ws) + p.items_bytes);
    if (!plan_items)
        hipLaunchKernelGGL(wgrad_plan_kernel, dim3(1), dim3(256), 0, st, reinterpret_cast<const long long*>(counts),
                           int(n_out), K, p.T, p.min_rows, static_cast<int*>(ws), static_cast<int*>(ws) + size_t(p.T) * 4);
    const dim3 grid(p.n_ci * p.n_co, p.T), block(256);
    const bool av = (cin & 3) == 0, gv = (cout & 3) == 0;
#define OSN_WG(T_, A_, G_)                                                                                          \
    hipLaunchKernelGGL((spconv_wgrad_kernel<T_, A_, G_>), grid, block, 0, st, in, gout, nbr, partial, items, int(n_out), K, \
                       cin, cout, p.n_co, p.ci_t, p.co_t)
#define OSN_WG_T(T_)                                     \
    do {                                                 \
        if (av && gv) OSN_WG(T_, true, true);            \
        else if (av) OSN_WG(T_, true, false);            \
        else if (gv) OSN_WG(T_, false, true);            \
        else OSN_WG(T_, false, false);                   \
    } while (0)
    switch (p.tpw) {
        case 1: OSN_WG_T(1); break;
        case 2: OSN_WG_T(2); break;
        case 3: OSN_WG_T(3); break;
        default: OSN_WG_T(4); break;
    }
#undef OSN_WG_T
#undef OSN_WG
    OSN_LAUNCH_CHECK();
    int g = int(cdiv(wtotal, 256));
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(reduce_items_kernel, dim3(g), dim3(256), 0, st, partial, range, K, int64_t(cin) * cout, gW);
    OSN_LAUNCH_CHECK();
    return OSN_OK;
}
