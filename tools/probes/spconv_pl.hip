// EXPERIMENT (round 5, REJECTED -- profiles/r05_s4_micro_planes_register_direct.txt; not part of the library, built by tools/micro_pl.py):
// the tile-list kernel's pair lists and LDS output tile, with the GATHERED operand read straight into MFMA fragments from
// PRE-SPLIT planes.  Bitwise the product kernel's results; 189 us against 121 us on the level-0 96 -> 96 convolution.
//
// Function parity (not a port) with MinkowskiEngine's convolution kernels:
//   out[o] = sum_k in[nbr[k, o]] @ W[k].
//
// What changed against spconv_tl.hip and why (VERDICT r4 "next" #1): that kernel's step is a serial chain -- gather 32 rows as
// fp32, split them into three bf16 pieces (VALU), stage them in LDS, two workgroup barriers, read the fragments back -- and its
// waves wait 55 % of their life (SQ_WAIT_ANY) with the MFMA pipe 23.5 % busy.  The three-way split is EXACT, so the producer of
// an activation can store it as three bf16 planes and every consumer sees the same numbers.  A row-major bf16 plane is
// already the 16 x 16 x 32 operand layout (a lane = 8 consecutive channels of one row = one 16-byte load), so here
//   * the input is `planes[row][3][cin]` bf16 (osn_split_planes writes it; 6 bytes per element);
//   * every MFMA wave loads ITS OWN A fragments of a 32-pair step with raw buffer loads through the pair list: no staging
//     buffer, no split, NO barrier inside a batch of offsets (a wave owns its 32 output columns of the LDS tile exclusively,
//     so waves never exchange data between the list load and the epilogue) and no staging-only fourth wave;
//   * a step's fragments are fetched one whole step ahead into a second register set (two waves per SIMD: the registers are
//     there), the weight fragments of the next offset behind the MFMAs that last use the current ones, as before;
//   * arithmetic, product order and summation order are those of spconv_tl_kernel: the result is bitwise the same.
// The price: the three waves of a workgroup each load the same rows (L1 serves two of three), and the planes are 1.5 x the
// bytes of the fp32 matrix.
#include "common.h"
#include "split.h"

namespace osn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int PL_BMAX = 128;      // rows per tile at most
constexpr int PL_LCAP = 1024;     // packed list entries resident in LDS per batch of offsets
constexpr int PL_STEPS = PL_LCAP / 32 * 4;
constexpr int PL_KMAXO = 128;     // kernel offsets
constexpr int PL_SLOTS = 512;

// planes[r][p][c] = piece p of in[r][c]
__global__ __launch_bounds__(256) void split_planes_kernel(const float4* __restrict__ in, int64_t total4, int c4,
                                                           split_bf16x4* __restrict__ planes) {
    for (int64_t e = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; e < total4; e += int64_t(gridDim.x) * blockDim.x) {
        const int64_t r = e / c4;
        const int q = int(e - r * c4);
        split_bf16x4 p1, p2, p3;
        tl_split4(in[e], p1, p2, p3);
        split_bf16x4* d = planes + r * (3 * c4) + q;
        d[0] = p1;
        d[c4] = p2;
        d[2 * c4] = p3;
    }
}

// NW waves, wave w owns output columns [32 w, 32 w + 32) of the workgroup's column group; channel chunks of KS x 32.
template <int NW, int KS, int OCC, int DBG = 0>
__global__ __launch_bounds__(64 * NW, OCC) void spconv_pl_kernel(const __bf16* __restrict__ planes, const bf16x8* __restrict__ Wp,
                                                                   const int32_t* __restrict__ cnt, const int2* __restrict__ lst,
                                                                   const int32_t* __restrict__ out_rows, float* __restrict__ out,
                                                                   double* __restrict__ bn_partial, int32_t* __restrict__ counter,
                                                                   int n_out, int K, int cin, int cout, int bm, int n_tiles, int ns,
                                                                   int ncb, int self_reset, unsigned planes_bytes) {
    constexpr int NT = 64 * NW;
    constexpr int CW = 32 * NW;
    constexpr int S = CW + 4;                              // fp32 row stride of the output tile
    constexpr int NL = (PL_LCAP + NT - 1) / NT;
    extern __shared__ __attribute__((aligned(16))) float otile[];      // (bm + 1) x S: the tile + one dump row for padded pairs
    __shared__ uint32_t plist[PL_LCAP];                    // (local output row << 24) | input row, 32-padded per offset
    __shared__ int klist[PL_KMAXO];
    __shared__ int kcnt[PL_KMAXO];
    __shared__ int lstart[PL_KMAXO + 1];
    __shared__ unsigned char gowner[PL_LCAP / 32];
    __shared__ uint4 stab[PL_STEPS];                       // x: first plist slot, y: pairs | first << 8 | first k-step << 16, z: weight block
    __shared__ int orow_s[PL_BMAX];
    __shared__ int nact_s, bend_s, tile_s;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col0 = blockIdx.y * CW;
    const int cb0 = blockIdx.y * (2 * NW) + 2 * wave;

    const unsigned row_bytes = 6u * unsigned(cin);         // one row of the planes: 3 x cin bf16
    const unsigned plane_step = 2u * unsigned(cin);
    const __amdgpu_buffer_rsrc_t prsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(planes), 0, int(planes_bytes), 0x00020000);
    uint32_t boff[2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) boff[nb] = (cb0 + nb < ncb ? unsigned(cb0 + nb) : 0u) * 1024u + 16u * unsigned(lane);
    const uint32_t wplane_bytes = unsigned(K) * unsigned(ns) * unsigned(ncb) * 1024u;
    const uint32_t kstep_bytes = unsigned(ncb) * 1024u;
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16x8*>(Wp), 0, int(3u * wplane_bytes), 0x00020000);

    bf16x8 B[KS][2][3];
    bf16x8 A0[KS][2][3], A1[KS][2][3];

    for (;;) {
        if (tid == 0) tile_s = atomicAdd(&counter[blockIdx.y], 1);
        __syncthreads();
        const int draw = __builtin_amdgcn_readfirstlane(tile_s);
        if (draw >= n_tiles) break;
        const int tile = n_tiles - 1 - draw;               // densest first (a tile-ordered table has the rows with most neighbours last)
        const int row0 = tile * bm;
        const int rows = min(bm, n_out - row0);
        for (int i = tid; i < rows; i += NT) orow_s[i] = out_rows ? out_rows[row0 + i] : row0 + i;
        {
            float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
            asm volatile("" : "+v"(z.x), "+v"(z.y), "+v"(z.z), "+v"(z.w));
            for (int i = tid; i < (bm + 1) * S / 4; i += NT) reinterpret_cast<float4*>(otile)[i] = z;
        }
        // ---- active offsets of the tile, ascending (=> fixed summation order)
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
            // ---- batch [a0, a1): as many consecutive offsets as fit the LDS list buffer
            if (tid == 0) {
                int tot = 0, a = a0;
                while (a < nact) {
                    const int np = (kcnt[a] + 31) & ~31;
                    if (tot + np > PL_LCAP) break;
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
                    const uint32_t v = okv[j] ? ((uint32_t(x[j].y) << 24) | uint32_t(x[j].x)) : (uint32_t(bm) << 24);   // padded: row 0, dump row
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
            const int T = nchunk * (E >> 5);
            __syncthreads();

            // ---- the batch's steps: no barrier from here to the end of the batch, and NO BRANCH inside a step.  The compiler's
            // wait-count pass is exact only inside one basic block: with a branch per half-step / per "weights ahead" the first MFMA
            // of a step waited for EVERY load in flight (s_waitcnt vmcnt(0 .. 5) in the first version's ISA), the prefetched next
            // step included.  So a step always multiplies both 16-pair halves (a half that is all padding accumulates into the
            // tile's dump row), always loads 18 A fragments for the step after it and always re-loads the weight fragments behind
            // each k-step's MFMAs -- the next (offset, chunk)'s, or the same ones again when the offset continues -- and a batch
            // runs an even number of steps, the surplus one being a copy of the last whose results go to the dump row.
            struct Step {
                int base, s0, blk0;                       // first plist slot, first k-step of the chunk, weight block
            };
            auto entry = [&](int t) {
                const uint4 v = stab[t < T ? t : T - 1];
                Step e;
                e.base = __builtin_amdgcn_readfirstlane(int(v.x));
                e.s0 = __builtin_amdgcn_readfirstlane(int(v.y >> 16));
                e.blk0 = __builtin_amdgcn_readfirstlane(int(v.z));
                return e;
            };
            // A fragments of a step: lane l = 8 channels (k-group l >> 4) of pair (l & 15) of each 16-pair half, per k-step and plane
            auto rows_of = [&](const Step& it, unsigned (&voff)[2]) {
#pragma unroll
                for (int h = 0; h < 2; ++h)
                    voff[h] = (plist[it.base + 16 * h + (lane & 15)] & 0xFFFFFFu) * row_bytes + 16u * unsigned(lane >> 4);
            };
            auto load_a = [&](int ks, const Step& it, const unsigned (&voff)[2], bf16x8 (&A)[KS][2][3]) {
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl)
                        A[ks][h][pl] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(
                                                                      prsrc, voff[h], unsigned(pl) * plane_step + 64u * unsigned(it.s0 + ks), 0));
            };
            auto load_b = [&](int ks, const Step& u) {
                const uint32_t ub = uint32_t(u.blk0) << 10;
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    const uint32_t so = ub + uint32_t(pl) * wplane_bytes + uint32_t(ks) * kstep_bytes;
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb)
                        B[ks][nb][pl] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, boff[nb], so, 0));
                }
            };
            // one step: accumulators start from the tile's cells, 72 MFMAs (KS = 3), cells written back; `live` false (the surplus
            // step of an odd batch): every lane accumulates into the dump row instead
            // The instruction ORDER of a step is pinned with scheduling barriers (left alone, the scheduler loaded the fragment the
            // first MFMA needs LAST and the step opened on s_waitcnt vmcnt(0)):
            //   A'(0) | MFMA(0) | B'(0) A'(1) | MFMA(1) | B'(1) A'(2) | MFMA(2) | B'(2)          ' = of the step after this one
            // A' goes to the other register set (free since the previous step), B' over the fragments MFMA(ks) just used.
            auto step = [&](const Step& it, bool live, bf16x8 (&A)[KS][2][3], const Step& nxt, bf16x8 (&An)[KS][2][3]) {
                unsigned nvoff[2];
                rows_of(nxt, nvoff);
                const int pbase = it.base + (lane & 15);
                int ocell[2];
                f32x4 acc[2][2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int lrow = live ? int(plist[pbase + 16 * h] >> 24) : bm;
                    ocell[h] = lrow * S + 32 * wave + 4 * (lane >> 4);
                    const f32x4* cell = reinterpret_cast<const f32x4*>(__builtin_assume_aligned(&otile[ocell[h]], 16));
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb) acc[h][nb] = cell[4 * nb];
                }
                __builtin_amdgcn_sched_barrier(0);
                if (!(DBG & 1)) load_a(0, nxt, nvoff, An);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    // product-major order, smallest terms first per accumulator: a3b1, a2b2, a1b3, a2b1, a1b2, a1b1 (= spconv_tl_kernel)
#define PL_MFMA(AP, BP)                                                                                     \
    _Pragma("unroll") for (int h = 0; h < 2; ++h) _Pragma("unroll") for (int nb = 0; nb < 2; ++nb)            \
        acc[h][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(B[ks][nb][BP], A[ks][h][AP], acc[h][nb], 0, 0, 0);
                    PL_MFMA(2, 0)
                    PL_MFMA(1, 1)
                    PL_MFMA(0, 2)
                    PL_MFMA(1, 0)
                    PL_MFMA(0, 1)
                    PL_MFMA(0, 0)
#undef PL_MFMA
                    __builtin_amdgcn_sched_barrier(0);
                    if (!(DBG & 2)) load_b(ks, nxt);       // behind the last use of B[ks]: the fragments of the step after this one
                    if (ks + 1 < KS && !(DBG & 1)) load_a(ks + 1, nxt, nvoff, An);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    f32x4* cell = reinterpret_cast<f32x4*>(__builtin_assume_aligned(&otile[ocell[h]], 16));
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb) cell[4 * nb] = acc[h][nb];
                }
            };

            if (T > 0) {
                Step cur = entry(0), n1 = entry(1);
                {
                    unsigned voff0[2];
                    rows_of(cur, voff0);
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {      // the order the loop leaves its own loads in: per k-step, A then B
                        load_a(ks, cur, voff0, A0);
                        if (DBG & 1) load_a(ks, cur, voff0, A1);
                        load_b(ks, cur);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                for (int t = 0; t < T; t += 2) {
                    step(cur, true, A0, n1, A1);
                    cur = n1; n1 = entry(t + 2);
                    step(cur, t + 1 < T, A1, n1, A0);
                    cur = n1; n1 = entry(t + 3);
                }
            }
            a0 = a1;
        }
        __syncthreads();                                   // every wave's columns of the tile are final

        // ---- epilogue: tile rows -> out[out_rows[row]] (16-byte stores), optional batch-norm partial sums
        constexpr int V = CW / 4;
        for (int idx = tid; idx < rows * V; idx += NT) {
            const int j = idx / V, c4 = idx - j * V;
            const int col = col0 + 4 * c4;
            if (col < cout) {
                const float4 v = *reinterpret_cast<const float4*>(&otile[j * S + 4 * c4]);
                *reinterpret_cast<float4*>(out + int64_t(orow_s[j]) * cout + col) = v;
            }
        }
        if (bn_partial) {
            for (int c = tid; c < CW; c += NT) {
                if (col0 + c < cout) {
                    double s1 = 0, s2 = 0;
                    for (int j = 0; j < rows; ++j) {
                        const double v = otile[j * S + c];
                        s1 += v;
                        s2 += v * v;
                    }
                    bn_partial[(int64_t(tile) * 2 + 0) * cout + col0 + c] = s1;
                    bn_partial[(int64_t(tile) * 2 + 1) * cout + col0 + c] = s2;
                }
            }
        }
        __syncthreads();                                   // the tile buffer is free for the next draw
    }
    if ((self_reset & 1) && tid == 0) {
        const int done = atomicAdd(&counter[64 + blockIdx.y], 1);
        if (done == int(gridDim.x) - 1) {
            counter[blockIdx.y] = 0;
            counter[64 + blockIdx.y] = 0;
        }
    }
}

}  // namespace osn

using namespace osn;

extern "C" size_t osn_dbg_split_planes_bytes(int64_t n, int c) { return size_t(n > 0 ? n : 0) * size_t(c) * 6; }

// (osn_dbg_*: tools only until the experiment is decided -- tools/micro_pl.py -- not part of include/openscene_amd.h)
// planes[r][p][c] (bf16) = piece p of in[r][c], p = 0 .. 2: in == planes[.][0] + planes[.][1] + planes[.][2] exactly
extern "C" int osn_dbg_split_planes(const float* in, int64_t n, int c, void* planes, osn_stream_t stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    OSN_REQUIRE(n >= 0 && c >= 4 && (c & 3) == 0, OSN_E_ARG, "osn_split_planes: need c %% 4 == 0 (c=%d)", c);
    if (n == 0) return OSN_OK;
    OSN_REQUIRE(in && planes && aligned16(in) && aligned16(planes), OSN_E_ARG, "osn_split_planes: null or unaligned pointer");
    const int64_t total4 = n * (c / 4);
    int64_t g = cdiv(total4, 256);
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(split_planes_kernel, dim3(unsigned(g)), dim3(256), 0, st, reinterpret_cast<const float4*>(in), total4, c / 4,
                       static_cast<split_bf16x4*>(planes));
    OSN_LAUNCH_CHECK();
    return OSN_OK;
}

// The tile-list convolution on pre-split planes.  Same tile lists, weight image, counters and output as osn_spconv_fwd_tl_pc; cin must be
// a multiple of 32 with a k-step count the chunking divides (every MinkUNet width), cout a multiple of 32 up to 128 per column group.
extern "C" int osn_dbg_spconv_fwd_pl(const void* planes, int64_t n_in, const void* Wp, const void* tl, const int32_t* out_rows, float* out,
                                 double* bn_partial, int64_t n_out, int K, int cin, int cout, int bm, int32_t* counters, int occ,
                                 osn_stream_t stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    OSN_REQUIRE(n_out >= 0 && n_out < (int64_t(1) << 31), OSN_E_ARG, "osn_spconv_fwd_pl: n_out out of range");
    OSN_REQUIRE(K >= 1 && K <= PL_KMAXO && cin >= 32 && (cin & 31) == 0 && cout >= 4 && (cout & 3) == 0, OSN_E_ARG,
                "osn_spconv_fwd_pl: needs K <= %d, cin %% 32 == 0, cout %% 4 == 0 (K=%d cin=%d cout=%d)", PL_KMAXO, K, cin, cout);
    OSN_REQUIRE(bm >= 1 && bm <= PL_BMAX, OSN_E_ARG, "osn_spconv_fwd_pl: bm=%d (at most %d rows per tile)", bm, PL_BMAX);
    OSN_REQUIRE(n_in >= 0 && n_in <= (int64_t(1) << 24) && uint64_t(n_in) * uint64_t(cin) * 6u < (uint64_t(1) << 32), OSN_E_RANGE,
                "osn_spconv_fwd_pl: %lld input rows x %d channels (24-bit rows, planes below 4 GB)", (long long)n_in, cin);
    if (n_out == 0) return OSN_OK;
    OSN_REQUIRE(planes && Wp && out && tl && counters, OSN_E_ARG, "osn_spconv_fwd_pl: null pointer");
    const int ns = cin / 32, ncb = (cout + 15) / 16;
    const int ks = ns <= 4 ? ns : (ns % 4 == 0 ? 4 : (ns % 3 == 0 ? 3 : 0));
    OSN_REQUIRE(ks > 0 && ns % ks == 0 && (ns / ks) * (PL_LCAP / 32) <= PL_STEPS, OSN_E_RANGE, "osn_spconv_fwd_pl: %d input channels", cin);
    int nw = 1, best_pad = 1 << 30;
    for (int w = 4; w >= 1; --w) {
        const int pad = int(cdiv(cout, 32 * w)) * 32 * w - cout;
        if (pad < best_pad) { best_pad = pad; nw = w; }
    }
    const int gy = int(cdiv(cout, 32 * nw));
    OSN_REQUIRE(gy <= 64, OSN_E_ARG, "osn_spconv_fwd_pl: more than 64 column groups");
    const int64_t n_tiles = cdiv(n_out, bm);
    const size_t cb = align_up(size_t(n_tiles) * K * 4, 256);
    const int32_t* cnt = static_cast<const int32_t*>(tl);
    const int2* lst = reinterpret_cast<const int2*>(static_cast<const char*>(tl) + cb);
    const size_t tile_bytes = size_t(bm + 1) * size_t(32 * nw + 4) * 4;
    const int dbg = (occ >> 8) & 3;                    // tools: 1 = no A loads in the step loop, 2 = no B reloads, 3 = neither
    occ &= 0xFF;
    const int per_cu = occ >= 2 && occ <= 4 ? occ : 2;
    unsigned gx = unsigned(n_tiles < int64_t(256) * per_cu ? n_tiles : int64_t(256) * per_cu);
    const dim3 grid(gx, unsigned(gy));
    const unsigned planes_bytes = unsigned(uint64_t(n_in) * uint64_t(cin) * 6u);
    int rc_attr = OSN_OK;
#define OSN_PL3(NW_, KS_, OC_)                                                                                              \
    do {                                                                                                                   \
        auto kern = spconv_pl_kernel<NW_, KS_, OC_>;                                                                       \
        if (tile_bytes > 32 * 1024 &&                                                                                      \
            hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,           \
                                int(size_t(PL_BMAX + 1) * size_t(32 * NW_ + 4) * 4)) != hipSuccess) {                      \
            rc_attr = OSN_E_HIP;                                                                                           \
            break;                                                                                                         \
        }                                                                                                                  \
        hipLaunchKernelGGL(kern, grid, dim3(64 * NW_), tile_bytes, st, static_cast<const __bf16*>(planes),                 \
                           static_cast<const bf16x8*>(Wp), cnt, lst, out_rows, out, bn_partial, counters, int(n_out), K, cin, cout, \
                           bm, int(n_tiles), ns, ncb, 1, planes_bytes);                                                    \
    } while (0)
#define OSN_PL2(NW_, KS_)                                                                                                   \
    do {                                                                                                                   \
        if (per_cu >= 3) OSN_PL3(NW_, KS_, 3);                                                                             \
        else OSN_PL3(NW_, KS_, 2);                                                                                         \
    } while (0)
#define OSN_PLD(D_)                                                                                                         \
    do {                                                                                                                   \
        auto kern = spconv_pl_kernel<3, 3, 2, D_>;                                                                         \
        hipLaunchKernelGGL(kern, grid, dim3(192), tile_bytes, st, static_cast<const __bf16*>(planes),                      \
                           static_cast<const bf16x8*>(Wp), cnt, lst, out_rows, out, bn_partial, counters, int(n_out), K, cin, cout, \
                           bm, int(n_tiles), ns, ncb, 1, planes_bytes);                                                    \
    } while (0)
    if (dbg && nw == 3 && ks == 3 && tile_bytes <= 32 * 1024) {
        if (dbg == 1) OSN_PLD(1);
        else if (dbg == 2) OSN_PLD(2);
        else OSN_PLD(3);
        OSN_LAUNCH_CHECK();
        return OSN_OK;
    }
#undef OSN_PLD
#define OSN_PL(NW_)                                                                                                         \
    do {                                                                                                                   \
        switch (ks) {                                                                                                      \
            case 1: OSN_PL2(NW_, 1); break;                                                                                \
            case 2: OSN_PL2(NW_, 2); break;                                                                                \
            case 3: OSN_PL2(NW_, 3); break;                                                                                \
            default: OSN_PL2(NW_, 4); break;                                                                               \
        }                                                                                                                  \
    } while (0)
    switch (nw) {
        case 4: OSN_PL(4); break;
        case 3: OSN_PL(3); break;
        case 2: OSN_PL(2); break;
        default: OSN_PL(1); break;
    }
#undef OSN_PL
#undef OSN_PL2
#undef OSN_PL3
    OSN_REQUIRE(rc_attr == OSN_OK, OSN_E_HIP, "osn_spconv_fwd_pl: cannot reserve %zu bytes of LDS for the output tile", tile_bytes);
    OSN_LAUNCH_CHECK();
    return OSN_OK;
}

// (the probe is its own shared object: the library's error sink, to stderr)
namespace osn {
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vfprintf(stderr, fmt, ap);
    va_end(ap);
    fputc('\n', stderr);
}
}  // namespace osn
