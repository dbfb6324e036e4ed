// Tile-order optimisation for the implicit-GEMM convolution: order the output rows of a
// kernel map by their offset-occupancy bit mask so that the rows of one tile use (nearly)
// the same kernel offsets, and the convolution kernel's per-tile offset skip removes most of
// the empty (tile, offset) work.  Features stay in the caller's row order: the convolution
// takes the permutation as `out_rows` and the row-permuted table produced here.
//
// On the canonical 2 cm scene (SURVEY.md 8(d), 5.3 of 27 offsets occupied per voxel) a random
// row order leaves 27.0 active offsets per 128-row tile; mask order leaves 15.0 (11.7 per 32 rows).
// The sort key puts the RAREST offset in the most significant bit (given the per-offset pair
// counts): rows that use a rare offset end up together, so few tiles pay for it -- another
// -10 % active (tile, offset) pairs on that scene (13.5 per tile) compared to offset-index bit order.
#include "common.h"

#include <string.h>
#include <limits.h>
#include <rocprim/device/device_radix_sort.hpp>

namespace osn {

// key bit of offset k: its rank by pair count, most frequent = bit 0 ... rarest = bit K-1 (ties: lower k
// first); offset-index order when no counts are given.
__global__ void kmap_mask_kernel(const int32_t* __restrict__ nbr, int64_t n_out, int K,
                                 const long long* __restrict__ counts, uint32_t* __restrict__ mask,
                                 int32_t* __restrict__ iota) {
    __shared__ int bitpos[32];
    if (threadIdx.x < 32) {
        const int k = threadIdx.x;
        int pos = k;
        if (counts && k < K) {
            const long long c = counts[k];
            pos = 0;
            for (int j = 0; j < K; ++j) {
                const long long cj = counts[j];
                pos += (j != k && (cj > c || (cj == c && j < k))) ? 1 : 0;
            }
        }
        bitpos[k] = pos;
    }
    __syncthreads();
    const int64_t o = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (o >= n_out) return;
    uint32_t m = 0;
    for (int k = 0; k < K; ++k) m |= (nbr[int64_t(k) * n_out + o] >= 0 ? 1u : 0u) << bitpos[k];
    mask[o] = m;
    iota[o] = int(o);
}

__global__ void kmap_permute_kernel(const int32_t* __restrict__ nbr, const int32_t* __restrict__ order, int64_t n_out,
                                    int32_t* __restrict__ out) {
    const int64_t j = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    const int k = blockIdx.y;
    if (j >= n_out) return;
    out[int64_t(k) * n_out + j] = nbr[int64_t(k) * n_out + order[j]];
}

// gmask[g] = OR of the occupancy masks (bit k = offset k) of rows 32g .. 32g+31 of the sorted table
__global__ void kmap_group_mask_kernel(const int32_t* __restrict__ nbr_sorted, int64_t n_out, int K,
                                       uint32_t* __restrict__ gmask) {
    const int64_t j = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    uint32_t m = 0u;
    if (j < n_out)
        for (int k = 0; k < K; ++k) m |= (nbr_sorted[int64_t(k) * n_out + j] >= 0 ? 1u : 0u) << k;
#pragma unroll
    for (int d = 1; d < 32; d <<= 1) m |= __shfl_xor(m, d, 64);
    if ((threadIdx.x & 31) == 0 && j < n_out) gmask[j >> 5] = m;
}

// rocPRIM picks a merge sort (1 block sort + ~8 merge launches) below 1 M items -- 98 merge launches per scene
// (profiles/r02_s4), 174 us for the level-0 3^3 table of S100k.
using SortConfig = rocprim::default_config;      // measured: forcing the onesweep path (merge limit 4096) made the maps 0.25 ms slower per scene

struct SortWs {
    uint32_t *mask, *mask_sorted;
    int32_t* iota;
    void* tmp;
    size_t tmp_bytes, bytes;
};

static hipError_t sort32_tmp_bytes(int64_t n, size_t* out) {
    size_t b = 0;
    hipError_t e = rocprim::radix_sort_pairs<SortConfig>(nullptr, b, (uint32_t*)nullptr, (uint32_t*)nullptr, (int32_t*)nullptr,
                                             (int32_t*)nullptr, size_t(n > 0 ? n : 1), 0u, 32u, (hipStream_t)0);
    *out = b;
    return e;
}

static SortWs carve_sort(void* ws, int64_t n, size_t tmp_bytes) {
    SortWs w;
    char* base = static_cast<char*>(ws);
    size_t off = 0;
    auto take = [&](size_t b) { char* q = base ? base + off : nullptr; off += align_up(b, 256); return q; };
    const size_t m = size_t(n > 0 ? n : 1);
    w.mask = reinterpret_cast<uint32_t*>(take(m * 4));
    w.mask_sorted = reinterpret_cast<uint32_t*>(take(m * 4));
    w.iota = reinterpret_cast<int32_t*>(take(m * 4));
    w.tmp = take(tmp_bytes);
    w.tmp_bytes = tmp_bytes;
    w.bytes = off;
    return w;
}

}  // namespace osn

using namespace osn;

extern "C" size_t osn_kmap_sort_ws_bytes(int64_t n_out) {
    size_t tb = 0;
    if (sort32_tmp_bytes(n_out, &tb) != hipSuccess) return 0;
    return carve_sort(nullptr, n_out, tb).bytes;
}

extern "C" int osn_kmap_sort(const int32_t* nbr, int64_t n_out, int K, const int64_t* counts, int32_t* order,
                             int32_t* nbr_sorted, uint32_t* gmask, void* ws, size_t ws_bytes, osn_stream_t stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    OSN_REQUIRE(K >= 1 && K <= 32, OSN_E_ARG, "osn_kmap_sort: K=%d (only K <= 32 offsets fit the 32-bit occupancy mask)", K);
    OSN_REQUIRE(n_out >= 0 && n_out < (int64_t(1) << 31), OSN_E_ARG, "osn_kmap_sort: n_out out of range");
    if (n_out == 0) return OSN_OK;
    OSN_REQUIRE(nbr && order && nbr_sorted, OSN_E_ARG, "osn_kmap_sort: null pointer");
    size_t tb = 0;
    OSN_HIP(sort32_tmp_bytes(n_out, &tb));
    SortWs w = carve_sort(ws, n_out, tb);
    OSN_REQUIRE(ws && ws_bytes >= w.bytes, OSN_E_WS, "osn_kmap_sort: workspace %zu < %zu", ws_bytes, w.bytes);
    const int T = 256;
    hipLaunchKernelGGL(kmap_mask_kernel, dim3(cdiv(n_out, T)), dim3(T), 0, st, nbr, n_out, K,
                       reinterpret_cast<const long long*>(counts), w.mask, w.iota);
    OSN_LAUNCH_CHECK();
    size_t t2 = w.tmp_bytes;
    OSN_HIP(rocprim::radix_sort_pairs<SortConfig>(w.tmp, t2, w.mask, w.mask_sorted, w.iota, order, size_t(n_out), 0u, unsigned(K), st));
    hipLaunchKernelGGL(kmap_permute_kernel, dim3(cdiv(n_out, T), K), dim3(T), 0, st, nbr, order, n_out, nbr_sorted);
    if (gmask)
        hipLaunchKernelGGL(kmap_group_mask_kernel, dim3(cdiv(n_out, T)), dim3(T), 0, st, nbr_sorted, n_out, K, gmask);
    OSN_LAUNCH_CHECK();
    return OSN_OK;
}
