// EXPERIMENT (round 5; built by tools/micro_tr.py, not part of the library unless it wins): the tile-list convolution with its
// 32-pair step DECOUPLED -- one producer wave gathers, splits and stages the rows of the steps ahead into a ring of LDS slots, the
// MFMA waves consume the slots; the two sides meet through LDS flags (workgroup-scope acquire / release), not through the two
// workgroup barriers per step of spconv_tl_kernel.
//
// Why (counted in round 5): spconv_tl_kernel's step is a serial chain through two barriers (SQ_WAIT_ANY 55 % of the waves' life, MFMA
// pipe 23.5 % busy); removing the LDS hop altogether (spconv_pl.hip) is bitwise right and 56 % slower because three waves fetching the
// same rows saturate the CU's 64 B/clk vector-memory path -- but THAT kernel with its loads removed ran 88 us against ~105 us for
// this structure, i.e. a step without barriers is the faster shape as long as the gathered operand still enters the CU once and is
// broadcast through LDS.  So:
//   * waves 0 .. NW-1 (consumers) own 32 output columns each, keep the weight fragments of the current (offset, chunk) in
//     registers (re-loaded behind the MFMAs that last use them, as before), read ready-made A fragments from the ring slot of the
//     step and accumulate into the LDS output tile;
//   * wave NW (producer) walks the same step table GD steps ahead: 16-byte gathers through the pair list (GD register sets in flight),
//     three-way bf16 split, staged rows into slot (step mod R), then full[slot] = step + 1;
//   * a consumer publishes done[slot][wave] = step + 1 when it has read the slot's fragments; the producer re-uses a slot when
//     all consumers have published step - R + 1.  Step numbers count up over the workgroup's whole life: no flag is ever reset;
//   * workgroup barriers remain only where the lists change (per tile and per batch of offsets) and around the epilogue.
// Same pair lists, weight image, product order and summation order as spconv_tl_kernel: bitwise the same results.
#include "common.h"
#include "split.h"

namespace osn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int TR_BMAX = 128;
constexpr int TR_LCAP = 1024;
constexpr int TR_STEPS = TR_LCAP / 32 * 4;
constexpr int TR_KMAX = 128;

__device__ __forceinline__ int flag_load(const int* p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void flag_store(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }

template <int NW, int KS, int R, int GD, int NP, int DBG = 0>
__global__ __launch_bounds__(64 * (NW + NP)) __attribute__((amdgpu_waves_per_eu(NP == 2 ? 3 : 2, NP == 2 ? 3 : 2))) void spconv_tr_kernel(const float* __restrict__ in, const bf16x8* __restrict__ Wp,
                                                                      const int32_t* __restrict__ cnt, const int2* __restrict__ lst,
                                                                      const int32_t* __restrict__ out_rows, float* __restrict__ out,
                                                                      int32_t* __restrict__ counter, int n_out, int K, int cin, int cout,
                                                                      int bm, int n_tiles, int ns, int ncb, unsigned in_bytes) {
    constexpr int NT = 64 * (NW + NP);
    constexpr int CW = 32 * NW;
    constexpr int S = CW + 4;
    constexpr int CK = 32 * KS;
    constexpr int LDA = CK + 8;
    constexpr int QPR = CK / 4;
    constexpr int NQ = (32 * QPR + 64 * NP - 1) / (64 * NP);    // quads per PRODUCER lane and step
    constexpr int NL = (TR_LCAP + NT - 1) / NT;
    static_assert((32 * QPR) % (64 * NP) == 0, "a step's quads divide over the producers' lanes");
    extern __shared__ __attribute__((aligned(16))) float otile[];
    __shared__ __attribute__((aligned(16))) __bf16 ring[R][3][32][LDA];
    __shared__ uint32_t plist[TR_LCAP];
    __shared__ int klist[TR_KMAX];
    __shared__ int kcnt[TR_KMAX];
    __shared__ int lstart[TR_KMAX + 1];
    __shared__ unsigned char gowner[TR_LCAP / 32];
    __shared__ uint4 stab[TR_STEPS];
    __shared__ int orow_s[TR_BMAX];
    __shared__ int full_f[R][2];
    __shared__ int done_f[R][4];
    __shared__ int nact_s, bend_s, tile_s;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = wave >= NW;
    const int pw = producer ? wave - NW : 0;                 // producer index
    const int col0 = blockIdx.y * CW;
    const int cb0 = blockIdx.y * (2 * NW) + 2 * (producer ? 0 : wave);

    if (tid < 2 * R) full_f[tid >> 1][tid & 1] = 0;
    if (tid < 4 * R) done_f[tid >> 2][tid & 3] = 0;
    int gstep = 0;                                          // steps this workgroup has started (same count in every wave)
    int tile_iter = 0;

    // producer: staging coordinates of its quads
    int q_row[NQ], q_col[NQ];
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
        const int idx = lane + 64 * (NP * j + pw);
        q_row[j] = idx / QPR;
        q_col[j] = (idx % QPR) * 4;
    }
    const __amdgpu_buffer_rsrc_t insrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), 0, int(in_bytes), 0x00020000);
    const unsigned cin4 = unsigned(cin) * 4u;
    // consumers: weight fragments
    uint32_t boff[2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) boff[nb] = (cb0 + nb < ncb ? unsigned(cb0 + nb) : 0u) * 1024u + 16u * unsigned(lane);
    const uint32_t plane_bytes = unsigned(K) * unsigned(ns) * unsigned(ncb) * 1024u;
    const uint32_t kstep_bytes = unsigned(ncb) * 1024u;
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16x8*>(Wp), 0, int(3u * plane_bytes), 0x00020000);
    __syncthreads();

    for (;;) {
        if (!(DBG & 128)) {
            if (tid == 0) tile_s = atomicAdd(&counter[blockIdx.y], 1);
        } else if (tid == 0) {
            tile_s = (tile_iter++) * int(gridDim.x) + int(blockIdx.x);
        }
        __syncthreads();
        const int draw = __builtin_amdgcn_readfirstlane(tile_s);
        if (draw >= n_tiles) break;
        const int tile = n_tiles - 1 - draw;
        const int row0 = tile * bm;
        const int rows = min(bm, n_out - row0);
        for (int i = tid; i < rows; i += NT) orow_s[i] = out_rows ? out_rows[row0 + i] : row0 + i;
        {
            float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
            asm volatile("" : "+v"(z.x), "+v"(z.y), "+v"(z.z), "+v"(z.w));
            for (int i = tid; i < (bm + 1) * S / 4; i += NT) reinterpret_cast<float4*>(otile)[i] = z;
        }
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
        __syncthreads();
        const int nact = __builtin_amdgcn_readfirstlane(nact_s);

        int a0 = 0;
        while (a0 < nact) {
            if (tid == 0) {
                int tot = 0, a = a0;
                while (a < nact) {
                    const int np = (kcnt[a] + 31) & ~31;
                    if (tot + np > TR_LCAP) break;
                    lstart[a] = tot;
                    tot += np;
                    ++a;
                }
                lstart[a] = tot;
                bend_s = a;
            }
            __syncthreads();                               // (also: every wave is done with the previous batch's plist / stab)
            const int a1 = __builtin_amdgcn_readfirstlane(bend_s);
            const int E = __builtin_amdgcn_readfirstlane(lstart[a1]);
            for (int a = a0 + tid; a < a1; a += NT)
                for (int g = lstart[a] >> 5; g < (lstart[a + 1] >> 5); ++g) gowner[g] = (unsigned char)a;
            __syncthreads();
            {
                int2 x[NL];
                bool okv[NL];
#pragma unroll
                for (int j = 0; j < NL; ++j) {
                    const int e = tid + NT * j;
                    const int ec = e < E ? e : 0;
                    const int a = gowner[ec >> 5];
                    const int p = ec - lstart[a];
                    okv[j] = e < E && p < kcnt[a];
                    x[j] = lst[(int64_t(tile) * K + klist[a]) * bm + (okv[j] ? p : 0)];
                }
#pragma unroll
                for (int j = 0; j < NL; ++j) {
                    const uint32_t v = okv[j] ? ((uint32_t(x[j].y) << 24) | uint32_t(x[j].x)) : (uint32_t(bm) << 24);
                    if (tid + NT * j < E) plist[tid + NT * j] = v;
                }
            }
            const int nchunk = (ns + KS - 1) / KS;
            for (int a = a0 + tid; a < a1; a += NT) {
                const int np = kcnt[a], niter = (np + 31) >> 5, l0 = lstart[a];
                const int kk = klist[a];
                int t = nchunk * (l0 >> 5);
                for (int c = 0; c < nchunk; ++c)
                    for (int g = 0; g < niter; ++g, ++t)
                        stab[t] = make_uint4(uint32_t(l0 + 32 * g), uint32_t(min(32, np - 32 * g)) | (g == 0 ? 0x100u : 0u) | (uint32_t(c * KS) << 16),
                                             uint32_t((kk * ns + c * KS) * ncb), 0u);
            }
            const int T = (DBG & 512) ? 0 : nchunk * (E >> 5);
            __syncthreads();

            if (producer) {
                // ------------------------------------------------------------------------------ producer: steps 0 .. T-1 of the batch
                float4 P[GD][NQ];
                auto fetch = [&](int t, float4 (&Pb)[NQ]) {             // rows of step min(t, T - 1): unconditional, clamped
                    const uint4 v = stab[t < T ? t : T - 1];
                    const int base = __builtin_amdgcn_readfirstlane(int(v.x));
                    const unsigned soff = 128u * unsigned(__builtin_amdgcn_readfirstlane(int(v.y >> 16)));
#pragma unroll
                    for (int j = 0; j < NQ; ++j) {
                        const unsigned row = plist[base + q_row[j]] & 0xFFFFFFu;
                        Pb[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(insrc, __umul24(row, cin4) + 4u * unsigned(q_col[j]), soff, 0));
                    }
                };
#pragma unroll
                for (int d = 0; d < GD; ++d) {
                    fetch(d, P[d]);
                    __builtin_amdgcn_sched_barrier(0);
                }
                // (the loop body keeps every vector-memory instruction OUTSIDE branches -- loads of steps past the batch are clamped
                // duplicates -- so that the compiler's wait counts stay exact: the split of step t waits for ITS rows only and leaves
                // the GD - 1 younger sets in flight; only the LDS side (slot wait, staging, flag) is predicated on t < T)
                for (int t0 = 0; t0 < T; t0 += GD) {
#pragma unroll
                    for (int d = 0; d < GD; ++d) {
                        const int t = t0 + d;
                        const int g = gstep + t;
                        const int slot = g % R;
                        bf16x4 p1[NQ], p2[NQ], p3[NQ];
#pragma unroll
                        for (int j = 0; j < NQ; ++j) tl_split4(P[d][j], p1[j], p2[j], p3[j]);
                        if (t < T) {
                            // the slot is free when every consumer has read step g - R out of it
                            if (g >= R) {
                                const int want = g - R + 1;
#pragma unroll
                                for (int w = 0; w < NW; ++w)
                                    while (flag_load(&done_f[slot][w]) < want) __builtin_amdgcn_s_sleep(1);
                            }
#pragma unroll
                            for (int j = 0; j < NQ; ++j) {
                                if (DBG & 32) continue;
                                *reinterpret_cast<bf16x4*>(&ring[slot][0][q_row[j]][q_col[j]]) = p1[j];
                                *reinterpret_cast<bf16x4*>(&ring[slot][1][q_row[j]][q_col[j]]) = p2[j];
                                *reinterpret_cast<bf16x4*>(&ring[slot][2][q_row[j]][q_col[j]]) = p3[j];
                            }
                            if (lane == 0) flag_store(&full_f[slot][pw], g + 1);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        if (!(DBG & 1)) fetch(t + GD, P[d]);             // rows of step t + GD into the set just staged
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            } else {
                // ------------------------------------------------------------------------------ consumers
                struct Step {
                    int base, fl, blk0;
                    __device__ bool valid() const { return base >= 0; }
                    __device__ int np() const { return fl & 0xFF; }
                    __device__ bool first() const { return (fl & 0x100) != 0; }
                };
                auto entry = [&](int t) {
                    Step e;
                    if (t < T) {
                        const uint4 v = stab[t];
                        e.base = __builtin_amdgcn_readfirstlane(int(v.x));
                        e.fl = __builtin_amdgcn_readfirstlane(int(v.y));
                        e.blk0 = __builtin_amdgcn_readfirstlane(int(v.z));
                    } else {
                        e.base = -1; e.fl = 0; e.blk0 = 0;
                    }
                    return e;
                };
                bf16x8 B[KS][2][3];
                auto load_b = [&](int ks, const Step& u) {
                    const uint32_t ub = uint32_t(u.blk0) << 10;
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) {
                        const uint32_t so = ub + uint32_t(pl) * plane_bytes + uint32_t(ks) * kstep_bytes;
#pragma unroll
                        for (int nb = 0; nb < 2; ++nb)
                            B[ks][nb][pl] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, boff[nb], so, 0));
                    }
                };
                bool b_ahead = false;
                Step cur = entry(0), n1 = entry(1);
                for (int t = 0; cur.valid(); ++t) {
                    const int g = gstep + t;
                    const int slot = g % R;
                    if (cur.first() && !b_ahead && !((DBG & 2) && g > 0)) {
#pragma unroll
                        for (int ks = 0; ks < KS; ++ks) load_b(ks, cur);
                    }
                    const bool ahead = n1.valid() && n1.first();
                    const bool half1 = cur.np() > 16;
                    const int pbase = cur.base + (lane & 15);
                    int ocell[2];
                    f32x4 acc[2][2];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        if (h == 0 || half1) {
                            ocell[h] = (DBG & 4) ? (bm * S + 32 * wave + 4 * (lane >> 4)) : int(plist[pbase + 16 * h] >> 24) * S + 32 * wave + 4 * (lane >> 4);
                            const f32x4* cell = reinterpret_cast<const f32x4*>(__builtin_assume_aligned(&otile[ocell[h]], 16));
#pragma unroll
                            for (int nb = 0; nb < 2; ++nb) acc[h][nb] = (DBG & 4) ? f32x4{0.f, 0.f, 0.f, 0.f} : cell[4 * nb];
                        }
                    }
#pragma unroll
                    for (int pp = 0; pp < NP; ++pp)
                        while (flag_load(&full_f[slot][pp]) < g + 1) __builtin_amdgcn_s_sleep(1);      // the step's rows are staged
                    const int akq = 8 * (lane >> 4);
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
                        bf16x8 af[2][3];
#pragma unroll
                        for (int h = 0; h < 2; ++h)
#pragma unroll
                            for (int pl = 0; pl < 3; ++pl)
                                af[h][pl] = (DBG & 16) ? bf16x8{} : *reinterpret_cast<const bf16x8*>(&ring[slot][pl][(half1 ? h : 0) * 16 + (lane & 15)][ks * 32 + akq]);
#define TR_MFMA(H0, H1, AP, BP)                                                                             \
    _Pragma("unroll") for (int h = H0; h < H1; ++h) _Pragma("unroll") for (int nb = 0; nb < 2; ++nb)          \
        if (!(DBG & 8)) acc[h][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(B[ks][nb][BP], af[h][AP], acc[h][nb], 0, 0, 0);
                        if (half1) {
                            TR_MFMA(0, 2, 2, 0)
                            TR_MFMA(0, 2, 1, 1)
                            TR_MFMA(0, 2, 0, 2)
                            TR_MFMA(0, 2, 1, 0)
                            TR_MFMA(0, 2, 0, 1)
                            TR_MFMA(0, 2, 0, 0)
                        } else {
                            TR_MFMA(0, 1, 2, 0)
                            TR_MFMA(0, 1, 1, 1)
                            TR_MFMA(0, 1, 0, 2)
                            TR_MFMA(0, 1, 1, 0)
                            TR_MFMA(0, 1, 0, 1)
                            TR_MFMA(0, 1, 0, 0)
                        }
#undef TR_MFMA
                        if (ahead && !(DBG & 2)) load_b(ks, n1);
                    }
                    b_ahead = ahead;
                    if (lane == 0) flag_store(&done_f[slot][wave], g + 1);
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        if (h == 0 || half1) {
                            f32x4* cell = reinterpret_cast<f32x4*>(__builtin_assume_aligned(&otile[ocell[h]], 16));
                            if (DBG & 4) {
                                float sink = acc[h][0][0] + acc[h][1][0];
                                asm volatile("" ::"v"(sink));
                            } else {
#pragma unroll
                                for (int nb = 0; nb < 2; ++nb) cell[4 * nb] = acc[h][nb];
                            }
                        }
                    }
                    cur = n1;
                    n1 = entry(t + 2);
                }
            }
            gstep += T;
            a0 = a1;
        }
        __syncthreads();

        constexpr int V = CW / 4;
        if (!(DBG & 64))
        for (int idx = tid; idx < rows * V; idx += NT) {
            const int j = idx / V, c4 = idx - j * V;
            const int col = col0 + 4 * c4;
            if (col < cout) {
                const float4 v = *reinterpret_cast<const float4*>(&otile[j * S + 4 * c4]);
                *reinterpret_cast<float4*>(out + int64_t(orow_s[j]) * cout + col) = v;
            }
        }
        __syncthreads();
    }
    if (tid == 0) {
        const int done = atomicAdd(&counter[64 + blockIdx.y], 1);
        if (done == int(gridDim.x) - 1) {
            counter[blockIdx.y] = 0;
            counter[64 + blockIdx.y] = 0;
        }
    }
}

}  // namespace osn

using namespace osn;

// Same tile lists, weight image, counters and output as osn_spconv_fwd_tl_pc (full channel chunks, feature matrix below 2 GB, no
// offset split).  variant: ring slots R (2 / 3) | gather depth GD (1 / 2) << 4.
extern "C" int osn_dbg_spconv_fwd_tr(const float* in, int64_t n_in, const void* Wp, const void* tl, const int32_t* out_rows, float* out,
                                     int64_t n_out, int K, int cin, int cout, int bm, int32_t* counters, int variant, int wgs_per_cu,
                                     osn_stream_t stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    OSN_REQUIRE(in && Wp && tl && out && counters && n_out > 0 && bm >= 1 && bm <= TR_BMAX, OSN_E_ARG, "osn_dbg_spconv_fwd_tr: bad arguments");
    OSN_REQUIRE((cin & 31) == 0 && (cout & 31) == 0 && n_in < (int64_t(1) << 24) && uint64_t(n_in) * cin * 4 < (uint64_t(1) << 31), OSN_E_ARG,
                "osn_dbg_spconv_fwd_tr: shape");
    const int ns = cin / 32, ncb = cout / 16;
    const int ks = ns <= 4 ? ns : (ns % 4 == 0 ? 4 : (ns % 3 == 0 ? 3 : 0));
    OSN_REQUIRE(ks > 0 && ns % ks == 0, OSN_E_ARG, "osn_dbg_spconv_fwd_tr: channels");
    const int nw = cout % 128 == 0 ? 4 : (cout % 96 == 0 ? 3 : (cout % 64 == 0 ? 2 : 1));
    const int gy = cout / (32 * nw);
    const int64_t n_tiles = cdiv(n_out, bm);
    const size_t cb = align_up(size_t(n_tiles) * K * 4, 256);
    const int32_t* cnt = static_cast<const int32_t*>(tl);
    const int2* lst = reinterpret_cast<const int2*>(static_cast<const char*>(tl) + cb);
    const size_t tile_bytes = size_t(bm + 1) * size_t(32 * nw + 4) * 4;
    const int per_cu = wgs_per_cu >= 1 && wgs_per_cu <= 3 ? wgs_per_cu : 2;
    const unsigned gx = unsigned(n_tiles < int64_t(256) * per_cu ? n_tiles : int64_t(256) * per_cu);
    const dim3 grid(gx, unsigned(gy));
    const unsigned in_bytes = unsigned(uint64_t(n_in) * cin * 4);
    const int dbg = (variant >> 12) & 1023;
    const int R = variant & 15, GD = (variant >> 4) & 15, NPv = ((variant >> 8) & 3) ? ((variant >> 8) & 3) : 1;
    int rc_attr = OSN_OK;
#define OSN_TR(NW_, KS_, R_, GD_, NP_) OSN_TRD(NW_, KS_, R_, GD_, NP_, 0)
#define OSN_TRD(NW_, KS_, R_, GD_, NP_, DB_)                                                                                         \
    do {                                                                                                                   \
        auto kern = spconv_tr_kernel<NW_, KS_, R_, GD_, NP_, DB_>;                                                                 \
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,           \
                                int(size_t(TR_BMAX + 1) * size_t(32 * NW_ + 4) * 4)) != hipSuccess) {                      \
            rc_attr = OSN_E_HIP;                                                                                           \
            break;                                                                                                         \
        }                                                                                                                  \
        hipLaunchKernelGGL(kern, grid, dim3(64 * (NW_ + NP_)), tile_bytes, st, in, static_cast<const bf16x8*>(Wp), cnt, lst, out_rows, out, \
                           counters, int(n_out), K, cin, cout, bm, int(n_tiles), ns, ncb, in_bytes);                       \
    } while (0)
    if (nw == 3 && ks == 3) {
        if (R == 2 && GD == 1 && NPv == 1 && dbg == 0) OSN_TR(3, 3, 2, 1, 1);
        else if (R == 2 && GD == 1 && NPv == 1 && dbg == 1) OSN_TRD(3, 3, 2, 1, 1, 1);
        else if (R == 2 && GD == 1 && NPv == 1 && dbg == 2) OSN_TRD(3, 3, 2, 1, 1, 2);
        else if (R == 2 && GD == 1 && NPv == 1 && dbg == 3) OSN_TRD(3, 3, 2, 1, 1, 3);
        else if (R == 2 && GD == 1 && NPv == 1 && dbg == 7) OSN_TRD(3, 3, 2, 1, 1, 7);
        else if (R == 2 && GD == 1 && NPv == 1 && dbg == 11) OSN_TRD(3, 3, 2, 1, 1, 11);
        else if (R == 2 && GD == 1 && NPv == 1 && dbg == 15) OSN_TRD(3, 3, 2, 1, 1, 15);
        else if (R == 2 && GD == 1 && NPv == 1 && dbg == 31) OSN_TRD(3, 3, 2, 1, 1, 31);
        else if (R == 2 && GD == 1 && NPv == 1 && dbg == 47) OSN_TRD(3, 3, 2, 1, 1, 47);
        else if (R == 2 && GD == 1 && NPv == 1 && dbg == 63) OSN_TRD(3, 3, 2, 1, 1, 63);
        else if (R == 2 && GD == 1 && NPv == 1 && dbg == 127) OSN_TRD(3, 3, 2, 1, 1, 127);
        else if (R == 2 && GD == 1 && NPv == 1 && dbg == 191) OSN_TRD(3, 3, 2, 1, 1, 191);
        else if (R == 2 && GD == 1 && NPv == 1 && dbg == 255) OSN_TRD(3, 3, 2, 1, 1, 255);
        else if (R == 2 && GD == 1 && NPv == 1 && dbg == 128) OSN_TRD(3, 3, 2, 1, 1, 128);
        else if (R == 2 && GD == 1 && NPv == 1 && dbg == 639) OSN_TRD(3, 3, 2, 1, 1, 639);
        else if (R == 2 && GD == 2 && NPv == 2) OSN_TR(3, 3, 2, 2, 2);
        else if (R == 2 && GD == 3 && NPv == 2) OSN_TR(3, 3, 2, 3, 2);
        else if (R == 2 && GD == 4 && NPv == 2) OSN_TR(3, 3, 2, 4, 2);
        else if (R == 3 && GD == 4 && NPv == 2) OSN_TR(3, 3, 3, 4, 2);
        else OSN_REQUIRE(false, OSN_E_ARG, "osn_dbg_spconv_fwd_tr: variant");
    } else if (nw == 3 && ks == 4) {
        if (R == 2 && GD == 3 && NPv == 2) OSN_TR(3, 4, 2, 3, 2);
        else OSN_REQUIRE(false, OSN_E_ARG, "osn_dbg_spconv_fwd_tr: variant");
    } else {
        OSN_REQUIRE(false, OSN_E_ARG, "osn_dbg_spconv_fwd_tr: only 96 output columns in the probe");
    }
#undef OSN_TR
#undef OSN_TRD
    OSN_REQUIRE(rc_attr == OSN_OK, OSN_E_HIP, "osn_dbg_spconv_fwd_tr: LDS");
    OSN_LAUNCH_CHECK();
    return OSN_OK;
}

namespace osn {
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vfprintf(stderr, fmt, ap);
    va_end(ap);
    fputc('\n', stderr);
}
}  // namespace osn
