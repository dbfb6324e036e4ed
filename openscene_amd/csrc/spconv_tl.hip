// Sparse 3-D convolution on gfx950, second generation: per-tile COMPACTED pair lists, weights held in
// registers, output tile accumulated in LDS, work-balanced tiles handed out dynamically.
//
// Function parity (not a port) with MinkowskiEngine's convolution kernels (SURVEY.md section 2.1,
// appendix C item 5):  out[o] = sum_k in[nbr[k,o]] @ W[k].
//
// Why this shape (measured): on a 2 cm indoor scene a voxel has 5.3 of its 27 neighbours,
// so an output-stationary register tile pads every 32-row group to ~2x the real work and re-stages the weight
// chunk of every (tile, offset, channel chunk) through LDS.  Here:
//   * the kernel map of a tile of output rows is stored per offset as a dense list of (input row, local output
//     row) pairs (osn_tile_lists_build), so the MFMAs only see real pairs (padded to 32 per (tile, offset));
//   * a tile-ordered table puts the rows with many neighbours last, and the slowest tile takes 8x the fastest
//     (phase timers, profiles/): the tiles are handed to persistent workgroups through one atomic counter,
//     densest first (longest job first), instead of one workgroup per tile in launch order;
//   * a workgroup = NW waves, one per 32 output columns; a wave keeps the split-bf16 B fragments of W[k] for
//     ITS columns in registers for a whole (tile, offset), so weights enter the CU once per (tile, offset), are
//     loaded by exactly one wave, and never touch LDS (the load path into a CU, 64 B/clk, is what the first
//     version of this kernel saturated);
//   * gathered rows are split into three bf16 pieces ONCE per workgroup while they are staged to LDS; all waves
//     read ready-made MFMA A fragments with ds_read_b128;
//   * the pair lists of a batch of offsets are brought into LDS with one round of loads and the (offset, chunk,
//     32-pair step) sequence runs as one flat software pipeline whose gathers are issued ahead across offset
//     boundaries;
//   * the 16 x 16 result blocks are added into an fp32 output tile in LDS at the pairs' local output rows.
//     Within one offset an output row occurs at most once, offsets are walked in ascending order with
//     workgroup barriers in between, and a workgroup owns its rows exclusively => no atomics on data, fixed
//     summation order: bitwise reproducible whichever workgroup draws which tile;
//   * the epilogue streams the tile to HBM through the `out_rows` permutation (features stay in the caller's
//     row order) and can emit the per-tile column sums / sums of squares the following batch norm needs.
// Arithmetic: "bf16x6" -- x = x1 + x2 + x3 (bf16 pieces), a*b ~= a3b1 + a2b2 + a1b3 + a2b1 + a1b2 + a1b1
// accumulated in fp32 on v_mfma_f32_16x16x32_bf16 (dropped terms <= 2^-24 relative): fp32-class accuracy.
#include "common.h"
#include <atomic>
#include <type_traits>
#include "weight_prep.h"
#include "split.h"
#include "epilogue.h"

namespace osn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int TL_BMAX = 88;       // rows per tile at most (LDS budget of the 128-column instance, 2 workgroups / CU)
constexpr int TL_LIST_BMAX = 128; // rows per tile the LISTS can describe (a local row is 8 bits of a packed entry; spconv_pl.hip takes 128)
constexpr int TL_BMIN = 32;       // rows per tile of small tables
constexpr int TL_BOCC3 = 64;      // rows per tile that osn_tile_rows hands out at most (three workgroups per CU, see there)
constexpr int TL_LCAP = 1024;     // packed list entries resident in LDS per batch of offsets
constexpr int TL_STEPS = TL_LCAP / 32 * 4;   // step-table entries: 32-pair steps of a batch x channel chunks (<= 4: 512 channels)
constexpr int TL_KMAX = 128;      // kernel offsets a list-mode launch can take (5^3 = 125)
constexpr int TL_SLOTS = 512;     // persistent workgroups per column group (2 per CU)
constexpr int TL_MAX_DEVICES = 64; // per-device "LDS opt-in done" flags of every kernel instance

// ---- layout of a tile-list buffer ("tl"): tiles of bm = osn_tile_rows(n_out) consecutive table rows
//   int32 cnt[n_tiles][K]        pairs per (tile, offset)                 (256-byte aligned)
//   int2  lst[n_tiles][K][bm]    (input row, local output row), valid pairs first, ascending local row
struct TlView {
    int32_t* cnt;
    int2* lst;
    int64_t nt;
    size_t bytes;
};

static TlView tl_view(const void* base, int64_t n_out, int K, int bm) {
    TlView v;
    char* p = static_cast<char*>(const_cast<void*>(base));
    v.nt = cdiv(n_out > 0 ? n_out : 1, bm);
    const size_t cb = align_up(size_t(v.nt) * K * 4, 256);
    v.cnt = reinterpret_cast<int32_t*>(p);
    v.lst = reinterpret_cast<int2*>(p ? p + cb : nullptr);
    v.bytes = cb + size_t(v.nt) * K * bm * 8;
    return v;
}

// ------------------------------------------------------------------------------------------- lists
// cnt / lst of every tile.  One wave per offset (k = wave, wave + 4, ...): ballot compaction, no barriers.
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
__global__ void weight_prep_tl_kernel(const float* __restrict__ W, int K, int cin, int cout, int flip_b, int64_t per_plane_f,
                                      int64_t per_plane_b, __bf16* __restrict__ Wf, __bf16* __restrict__ Wb) {
    const int64_t total = per_plane_f + per_plane_b;
    for (int64_t e = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; e < total; e += int64_t(gridDim.x) * blockDim.x) {
        if (e < per_plane_f) weight_prep_tl_one(W, K, cin, cout, 0, 0, e, Wf);
        else weight_prep_tl_one(W, K, cin, cout, flip_b, 1, e - per_plane_f, Wb);
    }
}

// ------------------------------------------------------------------------------------------ conv
// PROF (tools only): wave 0 accumulates s_memtime deltas of the phases into prof[workgroup][10].
#define TL_TICK(slot)                                              \
    if (PROF) {                                                    \
        __builtin_amdgcn_sched_barrier(0);                         \
        const long long now_ = __builtin_amdgcn_s_memtime();       \
        tacc[slot] += now_ - tlast;                                \
        tlast = now_;                                              \
        __builtin_amdgcn_sched_barrier(0);                         \
    }

// 4 waves; wave w < NW owns output columns [32 w, 32 w + 32) of the workgroup's column group and BOTH 16-pair
// halves of a step (waves >= NW only help with the gathers and the staging); channel chunks of KS x 32 <= 128
// (B fragments of a chunk: KS k-steps x 2 column blocks x 3 planes = 24 KS VGPRs).  KS is the k-step count that
// tiles the input channels without a remainder where one exists (96 channels: KS = 3 -- with the fixed 128-channel
// chunk of round 2 a quarter of the weight-fragment loads and of the gather lanes of every 96-channel conv was padding).
// BUFG (full-chunk instances, feature matrices below 2 GB -- every launch of the benchmark configurations): the gathered rows come through
// a buffer resource over the feature matrix, like the weight fragments -- 32-bit offset row * (4 cin) + 4 col from one v_mad_u32_u24, the
// chunk's first channel in the scalar offset -- instead of a 64-bit multiply-add, two 64-bit adds and a channel clamp per quad (8 -> 2 VALU;
// round 5's A/B: step 8.80 -> 8.73 / 8.76 ms, bit-identical results, profiles/r05_s1_knobs_ab.txt).  Ragged chunks and matrices of 2 GB
// and more keep the 64-bit addresses.
// EPI (inference): the evaluation-mode batch norm of the stage in the epilogue (epilogue.h).  A template flag, not a run-time test: with
// the test in the training instances the 96 -> 96 kernel's spills went from 26 to 45 SGPRs and from 52 to 92 bytes of scratch per lane.
template <int NW, int KS, bool RAGGED, bool PROF = false, int OCC = 2, bool BUFG = false, bool EPI = false>
__global__ __launch_bounds__(256, OCC) void spconv_tl_kernel(const float* __restrict__ in, const bf16x8* __restrict__ Wp,
                                                               const int32_t* __restrict__ cnt, const int2* __restrict__ lst,
                                                               const int32_t* __restrict__ out_rows, float* __restrict__ out,
                                                               double* __restrict__ bn_partial, int32_t* __restrict__ counter,
                                                               float* __restrict__ partial, int nz, int n_out, int K, int cin,
                                                               int cout, int bm, int n_tiles, int ns, int ncb,
                                                               int self_reset, long long* __restrict__ prof, unsigned in_bytes,
                                                               const Epi epi) {
    static_assert(!(BUFG && RAGGED), "the buffer gather is for full channel chunks");
    constexpr int NT = 256;
    constexpr int CW = 32 * NW;               // output columns of the workgroup
    constexpr int S = CW + 4;                 // fp32 row stride of the output tile
    constexpr int CK = 32 * KS;               // input channels per chunk
    constexpr int LDA = CK + 8;               // bf16 row stride of a staged plane (16-byte aligned rows)
    constexpr int QPR = CK / 4;                           // 4-channel quads per staged row
    constexpr int NQ = (32 * QPR + NT - 1) / NT;          // quads per thread per step
    constexpr int NL = (TL_LCAP + NT - 1) / NT;           // list entries per thread per batch
    // the fp32 output tile: bm rows + one dump row for padded pairs.  Dynamic LDS (sized by the launch from the map's tile height):
    // a 64-row tile leaves room for a third workgroup per CU (the OCC = 3 instances)
    extern __shared__ __attribute__((aligned(16))) float otile[];
    __shared__ __attribute__((aligned(16))) __bf16 stage[3][32][LDA];
    __shared__ uint32_t plist[TL_LCAP];       // (local output row << 24) | input row, 32-padded per offset
    __shared__ int klist[TL_KMAX];
    __shared__ int kcnt[TL_KMAX];
    __shared__ int lstart[TL_KMAX + 1];       // first plist slot of each offset of the current batch
    __shared__ unsigned char gowner[TL_LCAP / 32];
    // Round 4: the batch's (offset, chunk, step) sequence as a TABLE, written once per batch by one thread per offset:
    //   x = first plist slot of the step   y = pairs of the step (1 .. 32) | first step of its (offset, chunk) << 8 | first
    //   k-step of the chunk << 16          z = 1 KB weight block of (offset, chunk): (k ns + s0) ncb
    // Round 3's loop advanced three iterators per step through LDS look-ups (pairs, list start and offset of the current
    // list entry, each a dependent read + readfirstlane) and rebuilt every weight-fragment address from scalars: 220 scalar
    // and 35 wait instructions per 72 MFMAs.  The kernel is bound by instruction issue (with every byte of memory traffic
    // removed it runs 149 instead of 173 us, profiles/r04_s2_*), so the control path is what there is to cut.
    __shared__ uint4 stab[TL_STEPS];
    __shared__ int nact_s, bend_s, tile_s;

    long long tacc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    long long tlast = PROF ? __builtin_amdgcn_s_memtime() : 0;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);         // wave-uniform => scalar registers
    const int col0 = blockIdx.y * CW;
    // (Rotating which wave of a workgroup is the staging-only one -- so that the staging waves of co-resident workgroups do not
    // all sit on one SIMD -- was measured: 1 - 7 % SLOWER, profiles/r04_s6_ab_wave_role_rotation.txt.  Waves keep their columns.)
    const int cw = wave;                                               // column-owner index of this wave (>= NW: staging only)
    const int cb0 = blockIdx.y * (2 * NW) + 2 * cw;                    // this wave's first 16-column block

    // loop-invariant staging coordinates of this thread's quads (full 128-channel rows; narrower chunks mask)
    int q_row[NQ], q_col[NQ];
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
        const int idx = tid + NT * j;
        q_row[j] = (idx / QPR) & 31;
        q_col[j] = (idx % QPR) * 4;
    }
    // BUFG: resource over the feature matrix (raw buffer, 32-bit byte offsets; in_bytes < 2^31 is the launch's condition)
    const __amdgpu_buffer_rsrc_t insrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), 0, int(BUFG ? in_bytes : 0u), 0x00020000);
    const unsigned cin4 = unsigned(cin) * 4u;

    bf16x8 B[KS][2][3];
    float4 P0[NQ], P1[NQ];
    // RAGGED = false: every channel chunk is full (every MinkUNet width) -- no channel masks, no fragment selects in the loop
    constexpr bool ragged = RAGGED;
    // a fragment load is ONE instruction: scalar base of (offset, chunk, k-step, plane) + this lane's byte offset inside the
    // column block (two registers: one per column block of the wave).  Column blocks past the weight are never stored: they
    // read block 0 instead (valid memory).
    uint32_t boff[2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) boff[nb] = (cb0 + nb < ncb ? unsigned(cb0 + nb) : 0u) * 1024u + 16u * unsigned(lane);
    const uint32_t plane_bytes = unsigned(K) * unsigned(ns) * unsigned(ncb) * 1024u;      // one bf16 piece of the whole weight
    const uint32_t kstep_bytes = unsigned(ncb) * 1024u;
    const __amdgpu_buffer_rsrc_t wrsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16x8*>(Wp), 0, int(3u * plane_bytes), 0x00020000);   // raw buffer, 32-bit offsets

    for (;;) {
        // ---- draw the next tile (densest first: a tile-ordered table has the rows with most neighbours last)
        if (tid == 0) tile_s = atomicAdd(&counter[blockIdx.y], 1);
        __syncthreads();
        // A unit of work = (tile, part z of nz): small tables cut the tile's active offsets into nz contiguous
        // parts handled by different workgroups (a tile's offsets are a serial chain: weights, then its steps);
        // part z writes its own partial tile, summed in part order by reduce_partial_rows_kernel.
        const int draw = __builtin_amdgcn_readfirstlane(tile_s);
        if (draw >= n_tiles * nz) break;
        const int tile = n_tiles - 1 - draw / nz;
        const int zpart = draw - (draw / nz) * nz;
        const int row0 = tile * bm;
        const int rows = min(bm, n_out - row0);
        // The tile's output row ids: loaded at the draw, parked before the epilogue in the unused tail of kcnt[] (K <= 40 offsets),
        // so that no load sits between the last MFMA and the stores: 96 -> 96 -1 %, 128 -> 96 -2 % (profiles/r04_s11_*).  NO new LDS
        // for it: 352 more bytes cross an allocation granule, the third workgroup per CU is gone and the kernel runs 138 instead
        // of 125 us (measured).  Also measured there: the next draw issued at the start of the epilogue (-1 % / 0), and a whole
        // tile of look-ahead -- draw and offset counts in flight across the steps -- 30 % SLOWER: the compiler's waitcnt pass sees
        // loads pending across the step loop and gives up its counted waits, and a reserved tile is lost to the dynamic balance.
        const bool park = K <= TL_KMAX - TL_BMAX;
        int* const orow_s = kcnt + (TL_KMAX - TL_BMAX);
        const int orow_pre = (park && tid < rows) ? (out_rows ? out_rows[row0 + tid] : row0 + tid) : 0;

        {
            // the zeros are materialised HERE (opaque to the optimiser): hoisted out of the persistent loop they stayed live across
            // every step, were spilled, had their registers borrowed as an MFMA temporary and were reloaded from scratch memory in
            // each step of the 96 -> 96 instance (ISA: one scratch_load_dwordx4 per step in the in-order vmcnt queue of the gathers)
            float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
            asm volatile("" : "+v"(z.x), "+v"(z.y), "+v"(z.z), "+v"(z.w));
            for (int i = tid; i < (bm + 1) * S / 4; i += NT) reinterpret_cast<float4*>(otile)[i] = z;
        }

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
        } else if (tid == 0) {              // identity map (K == 1): the tile's own rows, in order
            klist[0] = 0;
            kcnt[0] = rows;
            nact_s = 1;
        }
        __syncthreads();
        const int nact_all = __builtin_amdgcn_readfirstlane(nact_s);
        const int nact = nact_all * (zpart + 1) / nz;      // this part: active offsets [nact_all z / nz, nact_all (z + 1) / nz)
        TL_TICK(0)                                         // 0: tile draw, zeroing, active offsets

        int a0 = nact_all * zpart / nz;
        while (a0 < nact) {
            // ---- batch [a0, a1): as many consecutive offsets as fit the LDS list buffer
            if (tid == 0) {
                int tot = 0, a = a0;
                while (a < nact) {
                    const int np = (kcnt[a] + 31) & ~31;
                    if (tot + np > TL_LCAP) break;         // a single offset (<= 96 entries) always fits
                    lstart[a] = tot;
                    tot += np;
                    ++a;
                }
                lstart[a] = tot;
                bend_s = a;
            }
            __syncthreads();                               // (also: the previous batch is done with plist / lstart)
            const int a1 = __builtin_amdgcn_readfirstlane(bend_s);
            const int E = __builtin_amdgcn_readfirstlane(lstart[a1]);
            for (int a = a0 + tid; a < a1; a += NT)
                for (int g = lstart[a] >> 5; g < (lstart[a + 1] >> 5); ++g) gowner[g] = (unsigned char)a;
            __syncthreads();
            if (cnt) {
                // all loads first (unconditional, clamped addresses), then the selects and the LDS writes: one
                // exposed memory latency per batch instead of one per entry
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
                    // padded pair: input row 0 (valid address), dump row
                    const uint32_t v = okv[j] ? ((uint32_t(x[j].y) << 24) | uint32_t(x[j].x)) : (uint32_t(bm) << 24);
                    if (tid + NT * j < E) plist[tid + NT * j] = v;
                }
            } else {
                for (int e = tid; e < E; e += NT)
                    plist[e] = e < rows ? ((uint32_t(e) << 24) | uint32_t(row0 + e)) : (uint32_t(bm) << 24);
            }
            // step table of the batch: offset a owns entries [nchunk (lstart[a] / 32), + nchunk niter(a)), chunk-major
            const int nchunk = (ns + KS - 1) / KS;
            for (int a = a0 + tid; a < a1; a += NT) {
                const int np = kcnt[a], niter = (np + 31) >> 5, l0 = lstart[a];
                const int kk = cnt ? klist[a] : 0;
                int t = nchunk * (l0 >> 5);
                for (int c = 0; c < nchunk; ++c)
                    for (int g = 0; g < niter; ++g, ++t)
                        stab[t] = make_uint4(uint32_t(l0 + 32 * g), uint32_t(min(32, np - 32 * g)) | (g == 0 ? 0x100u : 0u) | (uint32_t(c * KS) << 16),
                                             uint32_t((kk * ns + c * KS) * ncb), 0u);
            }
            const int T = nchunk * (E >> 5);
            __syncthreads();
            TL_TICK(1)                                     // 1: batch list load

            // ---- flat pipeline over the step table of the batch
            struct Step {
                int base, fl, blk0;           // stab entry (scalars); base < 0: past the end
                __device__ bool valid() const { return base >= 0; }
                __device__ int np() const { return fl & 0xFF; }
                __device__ bool first() const { return (fl & 0x100) != 0; }
                __device__ int s0() const { return fl >> 16; }
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
            // gather of one step: 32 list entries x up to 128 channels, one 16-byte load per quad (unconditional,
            // clamped address; masked at conversion time)
            auto fetch = [&](const Step& it, float4 (&P)[NQ]) {
#pragma unroll
                for (int j = 0; j < NQ; ++j) {
                    const unsigned row = plist[it.base + q_row[j]] & 0xFFFFFFu;
                    if constexpr (BUFG) {
                        const unsigned voff = __umul24(row, cin4) + 4u * unsigned(q_col[j]);      // < in_bytes < 2^31 for every listed row
                        P[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(insrc, voff, 128u * unsigned(it.s0()), 0));
                    } else {
                        const int ch = 32 * it.s0() + q_col[j];
                        const unsigned cu = ch < cin ? unsigned(ch) : 0u;
                        P[j] = *reinterpret_cast<const float4*>(in + (uint64_t(row) * unsigned(cin) + cu));
                    }
                }
            };
            // fragments of k-step ks of the (offset, chunk) that step `u` belongs to: buffer loads -- resource = the whole weight
            // image, scalar offset = (offset, chunk, k-step, plane), vector offset = the lane's two precomputed registers
            auto load_b = [&](int ks, const Step& u) {
                const uint32_t ub = uint32_t(u.blk0) << 10;                                           // wave-uniform
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    // k-steps past the last chunk (ragged only) are never multiplied: any valid block will do (ks 0's)
                    const int kk = (!ragged || u.s0() + ks < ns) ? ks : 0;
                    const uint32_t so = ub + uint32_t(pl) * plane_bytes + uint32_t(kk) * kstep_bytes;  // scalar
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb)
                        B[ks][nb][pl] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, boff[nb], so, 0));
                }
            };
            bool b_ahead = false;          // this step's fragments were loaded during the previous step
            auto step = [&](const Step& it, float4 (&P)[NQ], const Step& n1s, const Step& nf) {
                // ---- split the fetched quads into three bf16 pieces (registers), two elements per conversion; channels past
                // the input (only when cin is not a multiple of the chunk) are zeroed first
                bf16x4 p1[NQ], p2[NQ], p3[NQ];
#pragma unroll
                for (int j = 0; j < NQ; ++j) {
                    float4 v = P[j];
                    if (ragged && !(32 * it.s0() + q_col[j] < cin)) v = make_float4(0.f, 0.f, 0.f, 0.f);
                    tl_split4(v, p1[j], p2[j], p3[j]);
                }
                TL_TICK(2)                                 // 2: wait for the gathered rows + split
                // ---- new (offset, chunk): B fragments of this wave's 32 columns, one coalesced 1 KB load each -- unless the
                // previous step already fetched them behind its own MFMAs (see the end of the k-step loop)
                if (it.first() && !b_ahead && cw < NW) {
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) load_b(ks, it);
                }
                // the NEXT step opens a new (offset, chunk): its fragments are loaded into the registers of k-step ks as soon as
                // this step's MFMAs of ks are issued (the last use of the current ones), i.e. beside the remaining MFMAs, the tile
                // write-back and the two barriers -- no second register set
                const bool ahead = n1s.valid() && n1s.first() && cw < NW;
                TL_TICK(8)                                 // 8: B-load issue
                __syncthreads();                           // every wave is done reading the previous stage
                TL_TICK(3)                                 // 3: barrier A
#pragma unroll
                for (int j = 0; j < NQ; ++j) {
                    if (NT * NQ == 32 * QPR || tid + NT * j < 32 * QPR) {
                        *reinterpret_cast<bf16x4*>(&stage[0][q_row[j]][q_col[j]]) = p1[j];
                        *reinterpret_cast<bf16x4*>(&stage[1][q_row[j]][q_col[j]]) = p2[j];
                        *reinterpret_cast<bf16x4*>(&stage[2][q_row[j]][q_col[j]]) = p3[j];
                    }
                }
                if (nf.valid()) fetch(nf, P);              // ahead of time; in flight during the MFMAs below
                TL_TICK(9)                                 // 9: stage write + next gather issue
                __syncthreads();                           // stage ready
                TL_TICK(4)                                 // 4: barrier B
                if (cw >= NW) { b_ahead = false; return; }     // staging-only wave
                // ---- 32 pairs x 32 columns per wave: accumulator blocks [pair half][column block]
                // (the last step of an offset may hold at most 16 pairs -- the average (tile, offset) of a 100 k-row map
                // has 36 -- : its second 16-pair half is all padding and is skipped: MFMAs, fragment reads, tile update)
                const bool half1 = it.np() > 16;                    // wave-uniform
                // Round 4: the accumulators START from the output tile's cells and are stored back after the last MFMA.  The
                // MFMAs take the WEIGHT fragment as their first operand and the staged rows as the second (same registers either
                // way round: both fragment layouts are [16 rows or columns][8 k per lane group]), i.e. they produce the block
                // TRANSPOSED: lane l holds columns 4 (l >> 4) .. + 3 of pair l & 15 -- four consecutive floats of one output row,
                // one 16-byte LDS access.  Round 3 added the finished block in a separate read-add-write phase, which the
                // compiler turned into a chain of eight dependent 8-byte LDS round trips (it could not prove the 16-byte
                // alignment): 920 of a step's 7 100 clocks.  Now the tile cells are read while the fragments are, and written
                // once.  (Within one offset an output row occurs at most once and a wave owns its columns: no other access to
                // these cells between the read and the write.)
                const int pbase = it.base + (lane & 15);
                int ocell[2];
                f32x4 acc[2][2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    if (h == 0 || half1) {
                        ocell[h] = int(plist[pbase + 16 * h] >> 24) * S + 32 * cw + 4 * (lane >> 4);
                        const f32x4* cell = reinterpret_cast<const f32x4*>(__builtin_assume_aligned(&otile[ocell[h]], 16));
#pragma unroll
                        for (int nb = 0; nb < 2; ++nb) acc[h][nb] = cell[4 * nb];
                    }
                }
                const int akq = 8 * (lane >> 4);
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    if (!ragged || it.s0() + ks < ns) {
                        bf16x8 af[2][3];
#pragma unroll
                        for (int h = 0; h < 2; ++h)
#pragma unroll
                            for (int pl = 0; pl < 3; ++pl)
                                af[h][pl] = *reinterpret_cast<const bf16x8*>(&stage[pl][(half1 ? h : 0) * 16 + (lane & 15)][ks * 32 + akq]);
                        // product-major order: consecutive MFMAs go to DIFFERENT accumulators (a dependent chain per
                        // accumulator would stall the in-order issue on every MFMA's result); per accumulator the
                        // order is still smallest terms first: a3b1, a2b2, a1b3, a2b1, a1b2, a1b1
#define TL_MFMA(H0, H1, AP, BP)                                                                             \
    _Pragma("unroll") for (int h = H0; h < H1; ++h) _Pragma("unroll") for (int nb = 0; nb < 2; ++nb)          \
        acc[h][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(B[ks][nb][BP], af[h][AP], acc[h][nb], 0, 0, 0);
                        if (half1) {
                            TL_MFMA(0, 2, 2, 0)
                            TL_MFMA(0, 2, 1, 1)
                            TL_MFMA(0, 2, 0, 2)
                            TL_MFMA(0, 2, 1, 0)
                            TL_MFMA(0, 2, 0, 1)
                            TL_MFMA(0, 2, 0, 0)
                        } else {
                            TL_MFMA(0, 1, 2, 0)
                            TL_MFMA(0, 1, 1, 1)
                            TL_MFMA(0, 1, 0, 2)
                            TL_MFMA(0, 1, 1, 0)
                            TL_MFMA(0, 1, 0, 1)
                            TL_MFMA(0, 1, 0, 0)
                        }
#undef TL_MFMA
                    }
                    if (ahead) load_b(ks, n1s);
                }
                b_ahead = ahead;
                if (PROF) {                                // force the MFMA results (and thus the B loads) before the tick
                    float sink = 0.f;
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int nb = 0; nb < 2; ++nb)
                            if (h == 0 || half1) sink += acc[h][nb][0];
                    asm volatile("" ::"v"(sink));
                }
                TL_TICK(5)                                 // 5: B-load wait + tile / fragment reads + MFMAs
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    if (h == 0 || half1) {
                        f32x4* cell = reinterpret_cast<f32x4*>(__builtin_assume_aligned(&otile[ocell[h]], 16));
#pragma unroll
                        for (int nb = 0; nb < 2; ++nb) cell[4 * nb] = acc[h][nb];
                    }
                }
                TL_TICK(6)                                 // 6: output-tile write-back
            };

            Step cur = entry(0), n1 = entry(1), n2 = entry(2);
            int tn = 3;
            if (cur.valid()) fetch(cur, P0);
            if (n1.valid()) fetch(n1, P1);
            while (cur.valid()) {
                step(cur, P0, n1, n2);
                cur = n1; n1 = n2; n2 = entry(tn++);
                if (!cur.valid()) break;
                step(cur, P1, n1, n2);
                cur = n1; n1 = n2; n2 = entry(tn++);
            }
            a0 = a1;
        }
        if (park && tid < rows) orow_s[tid] = orow_pre;
        __syncthreads();

        // ---- epilogue: tile rows -> out[out_rows[row]] (16-byte stores), optional batch-norm partial sums
        constexpr int V = CW / 4;
        if constexpr (EPI) {
            // a thread keeps ONE column quad and walks the rows (NT / V rows per pass): the stage's per-column constants -- four loads and
            // four reciprocal square roots -- once per tile and thread instead of once per stored quad
            constexpr int RPP = NT / V;
            const int c4 = tid % V, col = col0 + 4 * c4;
            if (tid < RPP * V && col < cout) {
                const EpiCols ec = epi_cols(epi, col);
                for (int j = tid / V; j < rows; j += RPP) {
                    const float4 v = *reinterpret_cast<const float4*>(&otile[j * S + 4 * c4]);
                    const int64_t orow = park ? orow_s[j] : (out_rows ? out_rows[row0 + j] : row0 + j);
                    *reinterpret_cast<float4*>(out + orow * cout + col) = epi_apply(epi, ec, v, orow, col, cout);
                }
            }
        } else
        for (int idx = tid; idx < rows * V; idx += NT) {
            const int j = idx / V, c4 = idx - j * V;
            const int col = col0 + 4 * c4;
            if (col < cout) {
                const float4 v = *reinterpret_cast<const float4*>(&otile[j * S + 4 * c4]);
                if (nz > 1) {                              // partial tile of part z, table row order
                    *reinterpret_cast<float4*>(partial + (int64_t(zpart) * n_out + row0 + j) * cout + col) = v;
                } else {
                    const int64_t orow = park ? orow_s[j] : (out_rows ? out_rows[row0 + j] : row0 + j);
                    // (inference: evaluation-mode batch norm + residual + ReLU on the finished row, epilogue.h)
                    *reinterpret_cast<float4*>(out + orow * cout + col) = v;
                }
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
        TL_TICK(7)                                         // 7: epilogue
    }
    if (PROF && tid == 0 && blockIdx.y == 0)
        for (int i = 0; i < 10; ++i) prof[int64_t(blockIdx.x) * 10 + i] = tacc[i];
    // caller-owned persistent counters: the last workgroup of a column group to leave (every other one has made its final
    // draw before it took its exit ticket) puts both counters back to zero for the next launch -- no memset per call
    if ((self_reset & 1) && tid == 0) {
        const int done = atomicAdd(&counter[64 + blockIdx.y], 1);
        if (done == int(gridDim.x) - 1) {
            counter[blockIdx.y] = 0;
            counter[64 + blockIdx.y] = 0;
        }
    }
}
#undef TL_TICK


// (Round 4 also tried this convolution with the gathered rows brought into LDS by LDS-DMA (global_load_lds_dwordx4), one
// barrier per step, the bf16 split done by every MFMA wave on its own fragments and the weight fragments of the next
// (offset, chunk) prefetched into a second register set: three variants, 139 - 173 us against 145 - 159 us for the kernel above
// at the time, and 149 us with EVERY byte of memory traffic removed -- which is how the kernel turned out to be bound by
// instruction issue, not by memory or latency.  The experiment is commits f129c6a .. 7a3f559 of this repository; its measurements are
// profiles/r04_s2_*.  What came out of it is the instruction diet of spconv_tl_kernel above.)

static int tl_waves(int cout) {
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

extern "C" int osn_tile_rows(int64_t n_out) {
    // rows per tile: about two rounds of the persistent workgroups, a multiple of 8 between 32 and 64.  Round 4: at most 64 rows
    // (was 88) -- with the output tile in dynamic LDS a <= 64-row tile of <= 96 columns lets THREE workgroups share a CU (the
    // OCC = 3 instances: level-0 96 -> 96 133 -> 126 us, level-1 105 -> 99 us), and the instances that stay at two workgroups per CU
    // lose nothing (128 -> 96: 162 / 163 us at 88 / 64 rows)
    int64_t bm = cdiv(n_out > 0 ? n_out : 1, 2 * (TL_SLOTS / 2 * 3));
    bm = (bm + 7) / 8 * 8;
    if (bm < TL_BMIN) bm = TL_BMIN;
    if (bm > TL_BOCC3) bm = TL_BOCC3;
    return int(bm);
}

extern "C" size_t osn_tile_lists_bytes(int64_t n_out, int K, int bm) {
    if (K < 1 || bm < 1) return 256;
    return tl_view(nullptr, n_out, K, bm).bytes;
}

extern "C" int osn_tile_lists_build(const int32_t* nbr, int64_t n_out, int K, int bm, void* tl, osn_stream_t stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    OSN_REQUIRE(n_out >= 0 && n_out < (int64_t(1) << 31), OSN_E_ARG, "osn_tile_lists_build: n_out out of range");
    OSN_REQUIRE(K >= 1 && K <= TL_KMAX, OSN_E_ARG, "osn_tile_lists_build: K=%d (at most %d offsets)", K, TL_KMAX);
    OSN_REQUIRE(bm >= 1 && bm <= TL_LIST_BMAX, OSN_E_ARG, "osn_tile_lists_build: bm=%d (at most %d rows per tile)", bm, TL_LIST_BMAX);
    if (n_out == 0) return OSN_OK;
    OSN_REQUIRE(nbr && tl, OSN_E_ARG, "osn_tile_lists_build: null pointer");
    TlView v = tl_view(tl, n_out, K, bm);
    hipLaunchKernelGGL(tile_lists_kernel, dim3(unsigned(v.nt)), dim3(256), 0, st, nbr, n_out, K, bm, v.cnt, v.lst);
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

// small tables: parts per tile such that a launch has ~1024 units of work (at most 16, at most K)
static int tl_split(int64_t n_out, int K, int cout, int bm) {
    if (K <= 1 || n_out <= 0) return 1;
    const int64_t units = cdiv(n_out, bm) * cdiv(cout, 32 * tl_waves(cout));
    int64_t nz = 1024 / (units > 0 ? units : 1);
    if (nz > 16) nz = 16;
    if (nz > K) nz = K;
    if (nz < 1) nz = 1;
    return int(nz);
}

// tile counters (one per column group; MUST be zero on entry, left dirty) + the partial tiles of a split launch
extern "C" size_t osn_spconv_fwd_tl_ws_bytes(int64_t n_out, int K, int cout, int bm) {
    if (bm < 1) return 512;
    const int nz = tl_split(n_out, K, cout, bm);
    return 512 + (nz > 1 ? size_t(nz) * size_t(n_out) * size_t(cout) * 4 : 0);
}

// out[out_rows ? out_rows[r] : r] = partial[0][r] + partial[1][r] + ...  (fixed order)
template <bool EPI>
__global__ void tl_reduce_parts_kernel(const float4* __restrict__ partial, int S, int64_t n_out, int c4,
                                       const int32_t* __restrict__ out_rows, float4* __restrict__ out, const Epi epi) {
    const int64_t total = n_out * c4;
    for (int64_t e = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; e < total; e += int64_t(gridDim.x) * blockDim.x) {
        float4 s = partial[e];
        for (int z = 1; z < S; ++z) {
            const float4 v = partial[int64_t(z) * total + e];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        const int64_t r = e / c4, c = e - r * c4;
        const int64_t orow = out_rows ? int64_t(out_rows[r]) : r;
        if constexpr (EPI) s = epi_quad(epi, s, orow, int(4 * c), 4 * c4);     // evaluation-mode batch norm (epilogue.h)
        out[orow * c4 + c] = s;
    }
}

static int spconv_fwd_tl_impl(const float* in, int64_t n_in, const void* Wp, const void* tl, const int32_t* out_rows,
                              float* out, double* bn_partial, int64_t n_out, int K, int cin, int cout, int bm, void* ws,
                              size_t ws_bytes, int32_t* counters, long long* prof, osn_stream_t stream, const Epi& epi = epi_none()) {
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
    const int nw = tl_waves(cout);
    const int gy = int(cdiv(cout, 32 * nw));
    const int nz = tl ? tl_split(n_out, K, cout, bm) : 1;
    const size_t need = osn_spconv_fwd_tl_ws_bytes(n_out, tl ? K : 1, cout, bm);
    OSN_REQUIRE(ws && ws_bytes >= need && gy <= 64, OSN_E_WS, "osn_spconv_fwd_tl: workspace %zu < %zu (or more than 64 column groups)", ws_bytes, need);
    OSN_REQUIRE(!(bn_partial && nz > 1), OSN_E_ARG, "osn_spconv_fwd_tl: bn_partial is not available on split (small-table) launches");
    int32_t* counter = counters;
    if (!counter) {                                  // no persistent counters: the head of the workspace, zeroed here
        OSN_HIP(hipMemsetAsync(ws, 0, 512, st));
        counter = static_cast<int32_t*>(ws);
    }
    const int self_reset = counters ? 1 : 0;
    float* partial = reinterpret_cast<float*>(static_cast<char*>(ws) + 512);
    const int32_t* cnt = nullptr;
    const int2* lst = nullptr;
    const int64_t n_tiles = cdiv(n_out, bm);
    if (tl) {
        TlView v = tl_view(tl, n_out, K, bm);
        cnt = v.cnt; lst = v.lst;
    }
    const int ns = (cin + 31) / 32, ncb = (cout + 15) / 16;
    const int64_t units = n_tiles * nz;
    unsigned gx = unsigned(units < TL_SLOTS ? units : TL_SLOTS);
    dim3 grid(gx, unsigned(gy));
    const bf16x8* wp = static_cast<const bf16x8*>(Wp);
    // k-steps per channel chunk: the whole contraction when it fits (<= 4 k-steps), else the divisor of the k-step count
    // that leaves no padded chunk (192 channels: 2 x 3), else 4
    const int ks = ns <= 4 ? ns : (ns % 4 == 0 ? 4 : (ns % 3 == 0 ? 3 : 4));
    OSN_REQUIRE(cdiv(ns, ks) * (TL_LCAP / 32) <= TL_STEPS, OSN_E_RANGE,
                "osn_spconv_fwd_tl: %d input channels need more than %d channel chunks per list batch (the kernel's step table); "
                "use osn_dense_fwd (K == 1) or osn_spconv_fwd_x6", cin, TL_STEPS / (TL_LCAP / 32));
    const bool ragged = (cin & 31) != 0 || ns % ks != 0;
    const bool ragged_host = ragged;
    const int ks_host = ks;
    // dynamic LDS = the output tile; beyond 64 KB per workgroup in total the kernel needs the opt-in attribute (once per instance)
    const size_t tile_bytes = size_t(bm + 1) * size_t(32 * nw + 4) * 4;
    // three workgroups per CU when the tile is low enough (<= 64 rows at <= 96 output columns: 3 x 53.7 KB)
    const bool occ3 = !prof && !ragged_host && nw <= 3 && ks_host <= 3 && tile_bytes + 28 * 1024 <= 54 * 1024;
    // gathers through a buffer resource (full-chunk instances, feature matrices below 2 GB)
    const uint64_t in_bytes64 = uint64_t(n_in) * uint64_t(cin) * 4u;
    // (the gather offset is __umul24(row, 4 cin): both factors below 2^24 -- rows by the check above, 4 cin stated here so that a
    //  later relaxation of either limit cannot silently corrupt addresses)
    const bool bufg = !prof && !ragged_host && in_bytes64 < (uint64_t(1) << 31) && n_in < (int64_t(1) << 24) && int64_t(4) * cin < (int64_t(1) << 24);
    const unsigned in_bytes = bufg ? unsigned(in_bytes64) : 0u;
    int rc_attr = OSN_OK;
    int dev_id = 0;
    OSN_HIP(hipGetDevice(&dev_id));
    const bool dev_slot_ok = dev_id >= 0 && dev_id < TL_MAX_DEVICES;      // (beyond: the attribute is set on every tall launch)
    const int dev_slot = dev_slot_ok ? dev_id : 0;
    if (occ3) {                                       // three persistent workgroups per CU
        gx = unsigned(units < TL_SLOTS / 2 * 3 ? units : TL_SLOTS / 2 * 3);
        grid = dim3(gx, unsigned(gy));
    }
    const bool use_epi = epi.mean != nullptr && nz == 1;        // (split launches: the reduction of the parts applies it)
#define OSN_TL4(NW_, KS_, RG_, PF_, OC_) OSN_TL5(NW_, KS_, RG_, PF_, OC_, false)
#define OSN_TL5(NW_, KS_, RG_, PF_, OC_, BG_)                                                                              \
    do {                                                                                                                   \
        if (use_epi) OSN_TL6(NW_, KS_, RG_, PF_, OC_, BG_, true);                                                          \
        else OSN_TL6(NW_, KS_, RG_, PF_, OC_, BG_, false);                                                                 \
    } while (0)
#define OSN_TL6(NW_, KS_, RG_, PF_, OC_, BG_, EP_)                                                                         \
    do {                                                                                                                   \
        auto kern = spconv_tl_kernel<NW_, KS_, RG_, PF_, OC_, BG_, EP_>;                                                   \
        /* dynamic LDS beyond the default limit needs the opt-in attribute: once per (instance, DEVICE), the largest tile any   \
           launch can ask for; relaxed atomics: a racing second thread (autograd's, the map prefetcher's) sets it again */       \
        static std::atomic<unsigned char> attr_set[TL_MAX_DEVICES];                                                        \
        if (tile_bytes > 36 * 1024 && !(dev_slot_ok && attr_set[dev_slot].load(std::memory_order_relaxed))) {              \
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,       \
                                    int(size_t(TL_BMAX + 1) * size_t(32 * NW_ + 4) * 4)) != hipSuccess) {                  \
                rc_attr = OSN_E_HIP;                                                                                       \
                break;                                                                                                     \
            }                                                                                                              \
            if (dev_slot_ok) attr_set[dev_slot].store(1, std::memory_order_relaxed);                                       \
        }                                                                                                                  \
        hipLaunchKernelGGL(kern, grid, dim3(256), tile_bytes, st, in, wp, cnt, lst, out_rows, out, bn_partial, counter, partial, nz, \
                           int(n_out), K, cin, cout, bm, int(n_tiles), ns, ncb, self_reset, prof, in_bytes,                \
                           use_epi ? epi : epi_none());                                                                    \
    } while (0)
#define OSN_TL3(NW_, KS_, RG_, PF_) OSN_TL4(NW_, KS_, RG_, PF_, 2)
#define OSN_TL2(NW_, KS_)                                                                                                  \
    do {                                                                                                                   \
        if (prof) {                                                                                                        \
            if (ragged) OSN_TL3(NW_, KS_, true, true);                                                                     \
            else OSN_TL3(NW_, KS_, false, true);                                                                           \
        } else if (ragged) {                                                                                               \
            OSN_TL3(NW_, KS_, true, false);                                                                                \
        } else if (occ3 && NW_ <= 3 && KS_ <= 3) {                                                                         \
            if (bufg) OSN_TL5((NW_ <= 3 ? NW_ : 3), (KS_ <= 3 ? KS_ : 3), false, false, 3, true);                          \
            else OSN_TL4((NW_ <= 3 ? NW_ : 3), (KS_ <= 3 ? KS_ : 3), false, false, 3);                                     \
        } else {                                                                                                           \
            if (bufg) OSN_TL5(NW_, KS_, false, false, 2, true);                                                            \
            else OSN_TL3(NW_, KS_, false, false);                                                                          \
        }                                                                                                                  \
    } while (0)
#define OSN_TL(NW_)                                                                                                        \
    do {                                                                                                                   \
        switch (ks) {                                                                                                      \
            case 1: OSN_TL2(NW_, 1); break;                                                                                \
            case 2: OSN_TL2(NW_, 2); break;                                                                                \
            case 3: OSN_TL2(NW_, 3); break;                                                                                \
            default: OSN_TL2(NW_, 4); break;                                                                               \
        }                                                                                                                  \
    } while (0)
    switch (nw) {
        case 4: OSN_TL(4); break;
        case 3: OSN_TL(3); break;
        case 2: OSN_TL(2); break;
        default: OSN_TL(1); break;
    }
#undef OSN_TL
#undef OSN_TL2
#undef OSN_TL3
#undef OSN_TL4
#undef OSN_TL5
#undef OSN_TL6
    OSN_REQUIRE(rc_attr == OSN_OK, OSN_E_HIP, "osn_spconv_fwd_tl: cannot reserve %zu bytes of LDS for the output tile", tile_bytes);
    OSN_LAUNCH_CHECK();
    if (nz > 1) {
        const int64_t total4 = n_out * (cout / 4);
        int g = int(cdiv(total4, 256));
        if (g > 4096) g = 4096;
        hipLaunchKernelGGL(epi.mean ? tl_reduce_parts_kernel<true> : tl_reduce_parts_kernel<false>, dim3(g), dim3(256), 0, st,
                           reinterpret_cast<const float4*>(partial), nz, n_out, cout / 4, out_rows, reinterpret_cast<float4*>(out), epi);
        OSN_LAUNCH_CHECK();
    }
    return OSN_OK;
}

int osn::spconv_fwd_tl_epi(const float* in, int64_t n_in, const void* Wp, const void* tl, const int32_t* out_rows, float* out,
                           int64_t n_out, int K, int cin, int cout, int bm, void* ws, size_t ws_bytes, int32_t* counters, const Epi& epi,
                           osn_stream_t stream) {
    OSN_REQUIRE(counters, OSN_E_ARG, "spconv_fwd_tl_epi: null counters");
    return spconv_fwd_tl_impl(in, n_in, Wp, tl, out_rows, out, nullptr, n_out, K, cin, cout, bm, ws, ws_bytes, counters, nullptr, stream, epi);
}

extern "C" int osn_spconv_fwd_tl(const float* in, int64_t n_in, const void* Wp, const void* tl, const int32_t* out_rows,
                                 float* out, double* bn_partial, int64_t n_out, int K, int cin, int cout, int bm, void* ws,
                                 size_t ws_bytes, osn_stream_t stream) {
    return spconv_fwd_tl_impl(in, n_in, Wp, tl, out_rows, out, bn_partial, n_out, K, cin, cout, bm, ws, ws_bytes, nullptr, nullptr,
                              stream);
}

extern "C" int osn_spconv_fwd_tl_pc(const float* in, int64_t n_in, const void* Wp, const void* tl, const int32_t* out_rows,
                                    float* out, double* bn_partial, int64_t n_out, int K, int cin, int cout, int bm, void* ws,
                                    size_t ws_bytes, int32_t* counters, osn_stream_t stream) {
    OSN_REQUIRE(counters, OSN_E_ARG, "osn_spconv_fwd_tl_pc: null counters (128 int32, zero before the first call)");
    return spconv_fwd_tl_impl(in, n_in, Wp, tl, out_rows, out, bn_partial, n_out, K, cin, cout, bm, ws, ws_bytes, counters, nullptr,
                              stream);
}

// Tools only (not part of include/openscene_amd.h, NOT in the product library: built with OSN_BUILD_TOOLS=1 python -m openscene_amd.build):
// the same launch with the phase timers of wave 0 of every workgroup written to prof[512][10] (s_memtime ticks; tools/prof_tl.py).
#ifdef OSN_BUILD_TOOLS
extern "C" int osn_dbg_spconv_fwd_tl_prof(const float* in, int64_t n_in, const void* Wp, const void* tl,
                                          const int32_t* out_rows, float* out, int64_t n_out, int K, int cin, int cout,
                                          int bm, void* ws, size_t ws_bytes, long long* prof, osn_stream_t stream) {
    return spconv_fwd_tl_impl(in, n_in, Wp, tl, out_rows, out, nullptr, n_out, K, cin, cout, bm, ws, ws_bytes, nullptr, prof, stream);
}
#endif
