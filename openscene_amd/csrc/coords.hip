// Coordinate maps and kernel maps on gfx950 (integer / HBM-bound work).
//
// Function parity (not a port) with MinkowskiEngine's CoordinateManager
// (insert_and_map, stride, kernel_map) -- SURVEY.md section 2.1 / appendix C 1-4.
// All results are bit-exact against oracle/coords.py: unique rows are numbered
// in first-occurrence order, which is obtained deterministically with an
// atomicMin on the row index followed by a prefix scan (no order-dependent
// atomics on the numbering itself).
#include "common.h"
#include "coords_impl.h"

namespace osn {

// ------------------------------------------------------------------ scan ----
// Exclusive scan of int32 flags, 1024 elements per 256-thread block.
constexpr int SCAN_TPB = 256;
constexpr int SCAN_IPT = 4;
constexpr int SCAN_BLOCK = SCAN_TPB * SCAN_IPT;

__device__ inline int wave_incl_scan(int v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    return v;
}

// (every kernel of the unique pipeline takes the row count either from the host (n) or, when n_dev is given, from device
// memory: a level of the coordinate pyramid can then be queued before the previous level's count has reached the host;
// the grid is sized by an upper bound and the surplus threads leave)
// FLAG (the pyramid's batched builder): `in` is not read -- the unique flag of row i (its table slot holds i: the first
// occurrence) is computed here from (vals, slot_of) and written to flag_out: unique_flag_kernel and its launch folded in
template <bool FLAG>
__global__ __launch_bounds__(SCAN_TPB) void scan_block_kernel(const int* __restrict__ in, int* __restrict__ out,
                                                              int* __restrict__ sums, int64_t n,
                                                              const int32_t* __restrict__ n_dev, const int32_t* __restrict__ vals,
                                                              const int32_t* __restrict__ slot_of, int* __restrict__ flag_out) {
    if (n_dev) n = *n_dev;
    __shared__ int wave_tot[SCAN_TPB / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t base = int64_t(blockIdx.x) * SCAN_BLOCK + int64_t(tid) * SCAN_IPT;
    int v[SCAN_IPT];
    int s = 0;
#pragma unroll
    for (int j = 0; j < SCAN_IPT; ++j) {
        if constexpr (FLAG) {
            v[j] = 0;
            if (base + j < n) {
                const int sl = slot_of[base + j];
                v[j] = (sl >= 0 && vals[sl] == int(base + j)) ? 1 : 0;
                flag_out[base + j] = v[j];
            }
        } else {
            v[j] = (base + j < n) ? in[base + j] : 0;
        }
        s += v[j];
    }
    int incl = wave_incl_scan(s, lane);
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    int woff = 0;
#pragma unroll
    for (int w = 0; w < SCAN_TPB / 64; ++w) woff += (w < wave) ? wave_tot[w] : 0;
    int run = woff + incl - s;
#pragma unroll
    for (int j = 0; j < SCAN_IPT; ++j) {
        if (base + j < n) out[base + j] = run;
        run += v[j];
    }
    if (tid == SCAN_TPB - 1) sums[blockIdx.x] = woff + incl;
}

// Single block: exclusive scan of the block sums in place; total -> sums[nb].
__global__ __launch_bounds__(1024) void scan_sums_kernel(int* __restrict__ sums, int nb, int32_t* __restrict__ total_out) {
    __shared__ int wave_tot[16];
    __shared__ int carry_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < nb; base += 1024) {
        int i = base + tid;
        int v = (i < nb) ? sums[i] : 0;
        int incl = wave_incl_scan(v, lane);
        if (lane == 63) wave_tot[wave] = incl;
        __syncthreads();
        int woff = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) woff += (w < wave) ? wave_tot[w] : 0;
        int carry = carry_s;
        if (i < nb) sums[i] = carry + woff + incl - v;
        __syncthreads();
        if (tid == 1023) carry_s = carry + woff + incl;
        __syncthreads();
    }
    if (tid == 0) {
        sums[nb] = carry_s;
        if (total_out) *total_out = carry_s;
    }
}

__global__ __launch_bounds__(SCAN_TPB) void scan_add_kernel(int* __restrict__ out, const int* __restrict__ sums,
                                                            int64_t n, const int32_t* __restrict__ n_dev) {
    if (n_dev) n = *n_dev;
    const int64_t base = int64_t(blockIdx.x) * SCAN_BLOCK + int64_t(threadIdx.x) * SCAN_IPT;
    const int off = sums[blockIdx.x];
#pragma unroll
    for (int j = 0; j < SCAN_IPT; ++j)
        if (base + j < n) out[base + j] += off;
}

// in -> out (exclusive), total written to sums[nb] (device); sums has nb+1 ints
// (nb + 1 = exclusive_scan_sums_count(n)).
size_t exclusive_scan_sums_count(int64_t n) { return size_t(cdiv(n > 0 ? n : 1, SCAN_BLOCK)) + 1; }

// n = (upper bound of) the element count; n_dev (nullable) the actual count in device memory; total_out (nullable) also
// receives the total
// flag_vals / flag_slot_of (both or neither): scan the unique flags computed from them on the fly; they are also written to `in`
int exclusive_scan_i32_dev(const int* in, int* out, int* sums, int64_t n, hipStream_t st, const int32_t* n_dev,
                           int32_t* total_out, const int32_t* flag_vals = nullptr, const int32_t* flag_slot_of = nullptr) {
    const int nb = int(cdiv(n, SCAN_BLOCK));
    if (flag_vals)
        hipLaunchKernelGGL(scan_block_kernel<true>, dim3(nb), dim3(SCAN_TPB), 0, st, in, out, sums, n, n_dev, flag_vals, flag_slot_of,
                           const_cast<int*>(in));
    else
        hipLaunchKernelGGL(scan_block_kernel<false>, dim3(nb), dim3(SCAN_TPB), 0, st, in, out, sums, n, n_dev, nullptr, nullptr, nullptr);
    hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(1024), 0, st, sums, nb, total_out);
    hipLaunchKernelGGL(scan_add_kernel, dim3(nb), dim3(SCAN_TPB), 0, st, out, sums, n, n_dev);
    OSN_LAUNCH_CHECK();
    return OSN_OK;
}

int exclusive_scan_i32(const int* in, int* out, int* sums, int64_t n, hipStream_t st) {
    return exclusive_scan_i32_dev(in, out, sums, n, st, nullptr, nullptr);
}

// ---------------------------------------------------------------- unique ----
__device__ inline int4 quantise(int4 c, int stride) {
    if (stride > 1) {
        c.y = floor_div(c.y, stride) * stride;
        c.z = floor_div(c.z, stride) * stride;
        c.w = floor_div(c.w, stride) * stride;
    }
    return c;
}

// int4 = (b, x, y, z) in (.x, .y, .z, .w)
__global__ void hash_insert_kernel(const int4* __restrict__ coords, int64_t n, int stride,
                                   uint64_t* __restrict__ keys, int32_t* __restrict__ vals, uint32_t mask,
                                   int32_t* __restrict__ slot_of, int* __restrict__ err,
                                   const int32_t* __restrict__ n_dev) {
    if (n_dev) n = *n_dev;
    const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int4 c = quantise(coords[i], stride);
    const int lim = COORD_BIAS - 1;
    if (c.x < 0 || c.x >= 0xFFFF || c.y < -lim || c.y > lim || c.z < -lim || c.z > lim || c.w < -lim || c.w > lim) {
        *err = 1;
        slot_of[i] = -1;
        return;
    }
    const uint64_t key = pack_key(c.x, c.y, c.z, c.w);
    uint32_t slot = hash_key(key) & mask;
    while (true) {
        unsigned long long prev = atomicCAS(reinterpret_cast<unsigned long long*>(&keys[slot]),
                                            (unsigned long long)KEY_EMPTY, (unsigned long long)key);
        if (prev == KEY_EMPTY || prev == key) break;
        slot = (slot + 1) & mask;
    }
    atomicMin(&vals[slot], int(i));
    slot_of[i] = int(slot);
}

__global__ void unique_flag_kernel(const int32_t* __restrict__ vals, const int32_t* __restrict__ slot_of,
                                   int* __restrict__ flag, int64_t n, const int32_t* __restrict__ n_dev) {
    if (n_dev) n = *n_dev;
    const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int s = slot_of[i];
    flag[i] = (s >= 0 && vals[s] == int(i)) ? 1 : 0;
}

__global__ void unique_emit_kernel(const int4* __restrict__ coords, int64_t n, int stride,
                                   const int32_t* __restrict__ vals, const int32_t* __restrict__ slot_of,
                                   const int* __restrict__ flag, const int* __restrict__ rank,
                                   int4* __restrict__ out_coords, int32_t* __restrict__ inverse,
                                   int32_t* __restrict__ first, const int32_t* __restrict__ n_dev) {
    if (n_dev) n = *n_dev;
    const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int s = slot_of[i];
    if (s < 0) { inverse[i] = -1; return; }
    const int head = vals[s];            // lowest row holding this key
    inverse[i] = rank[head];
    if (flag[i]) {
        const int u = rank[i];
        out_coords[u] = quantise(coords[i], stride);
        first[u] = int(i);
    }
}

// After the emit pass: table value = unique row number.
__global__ void table_renumber_kernel(int32_t* __restrict__ vals, const int32_t* __restrict__ slot_of,
                                      const int* __restrict__ flag, const int* __restrict__ rank, int64_t n,
                                      const int32_t* __restrict__ n_dev) {
    if (n_dev) n = *n_dev;
    const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (flag[i]) vals[slot_of[i]] = rank[i];
}

// ------------------------------------------------------------ kernel maps ----
__global__ void kmap_build_kernel(const uint64_t* __restrict__ keys, const int32_t* __restrict__ vals, uint32_t mask,
                                  const int4* __restrict__ out_coords, int64_t n_out, int ksize, int scale,
                                  int32_t* __restrict__ nbr, unsigned long long* __restrict__ counts) {
    const int64_t o = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    const int k = blockIdx.y;
    int r = -1;
    if (o < n_out) {
        const int ix = k % ksize, iy = (k / ksize) % ksize, iz = k / (ksize * ksize);
        const int c = (ksize & 1) ? ksize / 2 : 0;
        const int4 p = out_coords[o];
        const int x = p.y + (ix - c) * scale, y = p.z + (iy - c) * scale, z = p.w + (iz - c) * scale;
        const int lim = COORD_BIAS - 1;
        if (x >= -lim && x <= lim && y >= -lim && y <= lim && z >= -lim && z <= lim)
            r = table_find(keys, vals, mask, pack_key(p.x, x, y, z));
        nbr[int64_t(k) * n_out + o] = r;
    }
    if (counts) {                                   // pairs per offset: one atomic per workgroup
        const int c = __syncthreads_count(r >= 0);
        if (threadIdx.x == 0 && c) atomicAdd(&counts[k], (unsigned long long)c);
    }
}

// Map of an ODD kernel over the table's OWN rows (stride-1 convolution: out_coords = the coordinates the table was built
// from, in row order).  Such a map is its own mirror -- nbr[k][o] = i  <=>  nbr[K-1-k][i] = o -- so only the offsets
// below the centre are probed (blockIdx.y < K/2); a hit also writes its mirrored entry (the upper half is preset to -1),
// and the centre offset is the identity.  Same table, bit for bit, as kmap_build_kernel; half the random probes.
__global__ void kmap_build_self_kernel(const uint64_t* __restrict__ keys, const int32_t* __restrict__ vals, uint32_t mask,
                                       const int4* __restrict__ coords, int64_t n, int ksize, int scale,
                                       int32_t* __restrict__ nbr, unsigned long long* __restrict__ counts) {
    const int64_t o = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    const int k = blockIdx.y;
    const int K = ksize * ksize * ksize;
    if (k == K / 2) {                               // centre: every row is its own neighbour
        if (o < n) nbr[int64_t(k) * n + o] = int(o);
        if (counts && blockIdx.x == 0 && threadIdx.x == 0) counts[k] = (unsigned long long)n;
        return;
    }
    int r = -1;
    if (o < n) {
        const int ix = k % ksize, iy = (k / ksize) % ksize, iz = k / (ksize * ksize);
        const int c = ksize / 2;
        const int4 p = coords[o];
        const int x = p.y + (ix - c) * scale, y = p.z + (iy - c) * scale, z = p.w + (iz - c) * scale;
        const int lim = COORD_BIAS - 1;
        if (x >= -lim && x <= lim && y >= -lim && y <= lim && z >= -lim && z <= lim)
            r = table_find(keys, vals, mask, pack_key(p.x, x, y, z));
        nbr[int64_t(k) * n + o] = r;
        if (r >= 0) nbr[int64_t(K - 1 - k) * n + r] = int(o);
    }
    if (counts) {
        const int c = __syncthreads_count(r >= 0);
        if (threadIdx.x == 0 && c) {
            atomicAdd(&counts[k], (unsigned long long)c);
            atomicAdd(&counts[K - 1 - k], (unsigned long long)c);
        }
    }
}

__global__ void kmap_transpose_kernel(const int32_t* __restrict__ nbr, int64_t n_out, int64_t n_in,
                                      int32_t* __restrict__ tbl) {
    const int64_t o = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    const int k = blockIdx.y;
    if (o >= n_out) return;
    const int i = nbr[int64_t(k) * n_out + o];
    if (i >= 0) tbl[int64_t(k) * n_in + i] = int(o);
}

__global__ void kmap_count_kernel(const int32_t* __restrict__ nbr, int64_t n_out, unsigned long long* counts) {
    const int k = blockIdx.y;
    int64_t c = 0;
    for (int64_t o = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; o < n_out; o += int64_t(gridDim.x) * blockDim.x)
        c += nbr[int64_t(k) * n_out + o] >= 0;
    __shared__ long long wsum[4];
    for (int d = 32; d >= 1; d >>= 1) c += __shfl_down(c, d, 64);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        const long long t = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        if (t) atomicAdd(&counts[k], (unsigned long long)t);
    }
}

// ---- presets of many regions in one launch (the map builder's ~20 hipMemsetAsync calls per scene were 3 - 8 us launches each)
__global__ __launch_bounds__(256) void fill_batch_kernel(const FillJobs jobs) {
    const FillJob& jb = jobs.j[blockIdx.y];
    char* p = static_cast<char*>(jb.ptr);
    const uint64_t bytes = jb.bytes;
    const uint32_t v = jb.value;
    // head up to the first 16-byte boundary and tail behind the last one: 4-byte stores by the first threads of block 0
    const uint64_t head = (16u - (reinterpret_cast<uintptr_t>(p) & 15u)) & 15u;
    const uint64_t h = head < bytes ? head : bytes;
    const uint64_t mid = (bytes - h) & ~uint64_t(15);
    const uint64_t tail = bytes - h - mid;
    if (blockIdx.x == 0) {
        if (threadIdx.x < h / 4) reinterpret_cast<uint32_t*>(p)[threadIdx.x] = v;
        if (threadIdx.x >= 64 && threadIdx.x - 64 < tail / 4) reinterpret_cast<uint32_t*>(p + h + mid)[threadIdx.x - 64] = v;
    }
    uint4* q = reinterpret_cast<uint4*>(p + h);
    const uint4 vv = make_uint4(v, v, v, v);
    const uint64_t n16 = mid >> 4;
    for (uint64_t i = uint64_t(blockIdx.x) * 256 + threadIdx.x; i < n16; i += uint64_t(gridDim.x) * 256) q[i] = vv;
}

int fill_batch(const FillJob* jobs, int n, hipStream_t st) {
    for (int b0 = 0; b0 < n; b0 += FILL_BATCH) {
        FillJobs fj;
        const int nb = n - b0 < FILL_BATCH ? n - b0 : FILL_BATCH;
        uint64_t big = 0;
        for (int i = 0; i < nb; ++i) {
            fj.j[i] = jobs[b0 + i];
            OSN_REQUIRE((reinterpret_cast<uintptr_t>(fj.j[i].ptr) & 3u) == 0 && (fj.j[i].bytes & 3u) == 0, OSN_E_ARG,
                        "fill_batch: region %d is not 4-byte aligned", b0 + i);
            if (fj.j[i].bytes > big) big = fj.j[i].bytes;
        }
        for (int i = nb; i < FILL_BATCH; ++i) fj.j[i] = FillJob{nullptr, 0, 0, 0};
        int64_t gx = cdiv(int64_t(big >> 4), 256 * 8);
        if (gx < 1) gx = 1;
        if (gx > 2048) gx = 2048;
        hipLaunchKernelGGL(fill_batch_kernel, dim3(unsigned(gx), unsigned(nb)), dim3(256), 0, st, fj);
        OSN_LAUNCH_CHECK();
    }
    return OSN_OK;
}

}  // namespace osn

using namespace osn;

extern "C" int64_t osn_hash_capacity(int64_t n) {
    int64_t cap = 1024;
    while (cap < 2 * n) cap <<= 1;
    return cap;
}

namespace {
struct UniqueWs {
    int32_t* slot_of;
    int* flag;
    int* rank;
    int* sums;
    int* err;
    size_t bytes;
};
UniqueWs carve_unique(void* ws, int64_t n) {
    UniqueWs w;
    char* p = static_cast<char*>(ws);
    size_t off = 0;
    auto take = [&](size_t b) { char* q = p ? p + off : nullptr; off += align_up(b, 256); return q; };
    const int64_t m = n > 0 ? n : 1;
    w.slot_of = reinterpret_cast<int32_t*>(take(size_t(m) * 4));
    w.flag = reinterpret_cast<int*>(take(size_t(m) * 4));
    w.rank = reinterpret_cast<int*>(take(size_t(m) * 4));
    w.sums = reinterpret_cast<int*>(take(size_t(cdiv(m, SCAN_BLOCK) + 2) * 4));
    w.err = reinterpret_cast<int*>(take(256));
    w.bytes = off;
    return w;
}
}  // namespace

extern "C" size_t osn_coords_unique_ws_bytes(int64_t n) { return carve_unique(nullptr, n).bytes; }

namespace {
// The unique pipeline of one level, queued on `st` without any host synchronisation.  n = (upper bound of) the row count,
// n_dev (nullable) = the actual count in device memory, count_out (nullable) receives the number of unique rows, err = a
// device flag (set to 1 on a coordinate outside the packable range; zeroed by the caller).
int queue_unique(const int32_t* coords4, int64_t n, const int32_t* n_dev, int stride, uint64_t* table_keys,
                 int32_t* table_vals, int64_t cap, int32_t* out_coords4, int32_t* inverse, int32_t* first,
                 const UniqueWs& w, int* err, int32_t* count_out, hipStream_t st, bool batched = false) {
    // batched (osn_coords_pyramid_async): the tables were preset by the caller's one fill launch, the flag pass rides in the scan
    if (!batched) {
        OSN_HIP(hipMemsetAsync(table_keys, 0xFF, size_t(cap) * 8, st));
        OSN_HIP(hipMemsetAsync(table_vals, 0x7F, size_t(cap) * 4, st));
    }
    const int T = 256;
    const dim3 grid(cdiv(n, T));
    const int4* c4 = reinterpret_cast<const int4*>(coords4);
    hipLaunchKernelGGL(hash_insert_kernel, grid, dim3(T), 0, st, c4, n, stride, table_keys, table_vals,
                       uint32_t(cap - 1), w.slot_of, err, n_dev);
    if (!batched) hipLaunchKernelGGL(unique_flag_kernel, grid, dim3(T), 0, st, table_vals, w.slot_of, w.flag, n, n_dev);
    OSN_LAUNCH_CHECK();
    int rc = batched ? exclusive_scan_i32_dev(w.flag, w.rank, w.sums, n, st, n_dev, count_out, table_vals, w.slot_of)
                     : exclusive_scan_i32_dev(w.flag, w.rank, w.sums, n, st, n_dev, count_out);
    if (rc) return rc;
    hipLaunchKernelGGL(unique_emit_kernel, grid, dim3(T), 0, st, c4, n, stride, table_vals, w.slot_of, w.flag, w.rank,
                       reinterpret_cast<int4*>(out_coords4), inverse, first, n_dev);
    hipLaunchKernelGGL(table_renumber_kernel, grid, dim3(T), 0, st, table_vals, w.slot_of, w.flag, w.rank, n, n_dev);
    OSN_LAUNCH_CHECK();
    return OSN_OK;
}
}  // namespace

extern "C" int osn_coords_unique(const int32_t* coords4, int64_t n, int stride, uint64_t* table_keys,
                                 int32_t* table_vals, int64_t cap, int32_t* out_coords4, int32_t* inverse,
                                 int32_t* first, int64_t* n_unique_host, void* ws, size_t ws_bytes,
                                 osn_stream_t stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    OSN_REQUIRE(n >= 0 && n < (int64_t(1) << 31), OSN_E_ARG, "osn_coords_unique: n=%lld out of range", (long long)n);
    OSN_REQUIRE(n_unique_host, OSN_E_ARG, "osn_coords_unique: n_unique_host is null");
    OSN_REQUIRE(stride >= 1, OSN_E_ARG, "osn_coords_unique: stride must be >= 1");
    OSN_REQUIRE(cap >= 2 * n && cap >= 2 && (cap & (cap - 1)) == 0, OSN_E_ARG,
                "osn_coords_unique: cap=%lld must be a power of two >= 2n", (long long)cap);
    OSN_REQUIRE(table_keys && table_vals, OSN_E_ARG, "osn_coords_unique: null table");
    *n_unique_host = 0;
    if (n == 0) {
        OSN_HIP(hipMemsetAsync(table_keys, 0xFF, size_t(cap) * 8, st));
        OSN_HIP(hipMemsetAsync(table_vals, 0x7F, size_t(cap) * 4, st));
        return OSN_OK;
    }
    OSN_REQUIRE(coords4 && out_coords4 && inverse && first, OSN_E_ARG, "osn_coords_unique: null pointer");
    OSN_REQUIRE(aligned16(coords4) && aligned16(out_coords4), OSN_E_ARG, "osn_coords_unique: coords must be 16-byte aligned");
    UniqueWs w = carve_unique(ws, n);
    OSN_REQUIRE(ws && ws_bytes >= w.bytes, OSN_E_WS, "osn_coords_unique: workspace %zu < %zu", ws_bytes, w.bytes);
    OSN_HIP(hipMemsetAsync(w.err, 0, 4, st));
    int rc = queue_unique(coords4, n, nullptr, stride, table_keys, table_vals, cap, out_coords4, inverse, first, w, w.err,
                          nullptr, st);
    if (rc) return rc;
    int host[2] = {0, 0};
    const int nb = int(cdiv(n, SCAN_BLOCK));
    OSN_HIP(hipMemcpyAsync(&host[0], w.sums + nb, 4, hipMemcpyDeviceToHost, st));
    OSN_HIP(hipMemcpyAsync(&host[1], w.err, 4, hipMemcpyDeviceToHost, st));
    OSN_HIP(hipStreamSynchronize(st));
    OSN_REQUIRE(host[1] == 0, OSN_E_RANGE,
                "osn_coords_unique: coordinate outside the packable range (|x|,|y|,|z| < 32767, 0 <= batch < 65535)");
    *n_unique_host = host[0];
    return OSN_OK;
}

// The same pipeline WITHOUT the host synchronisation, for chaining the levels of a coordinate pyramid: n_max bounds the
// row count (grid and buffer sizes), n_dev (nullable) holds the actual count in device memory (e.g. the previous
// level's count_dev), count_dev receives this level's unique count, err_dev is a sticky device flag the caller zeroed.
// Everything is sized for n_max rows; the caller reads the counts (and err) back once for the whole pyramid.
extern "C" int osn_coords_unique_async(const int32_t* coords4, int64_t n_max, const int32_t* n_dev, int stride,
                                       uint64_t* table_keys, int32_t* table_vals, int64_t cap, int32_t* out_coords4,
                                       int32_t* inverse, int32_t* first, int32_t* count_dev, int32_t* err_dev, void* ws,
                                       size_t ws_bytes, osn_stream_t stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    OSN_REQUIRE(n_max >= 1 && n_max < (int64_t(1) << 31), OSN_E_ARG, "osn_coords_unique_async: n_max=%lld out of range", (long long)n_max);
    OSN_REQUIRE(stride >= 1, OSN_E_ARG, "osn_coords_unique_async: stride must be >= 1");
    OSN_REQUIRE(cap >= 2 * n_max && (cap & (cap - 1)) == 0, OSN_E_ARG,
                "osn_coords_unique_async: cap=%lld must be a power of two >= 2 n_max", (long long)cap);
    OSN_REQUIRE(table_keys && table_vals && coords4 && out_coords4 && inverse && first && count_dev && err_dev, OSN_E_ARG,
                "osn_coords_unique_async: null pointer");
    OSN_REQUIRE(aligned16(coords4) && aligned16(out_coords4), OSN_E_ARG, "osn_coords_unique_async: coords must be 16-byte aligned");
    UniqueWs w = carve_unique(ws, n_max);
    OSN_REQUIRE(ws && ws_bytes >= w.bytes, OSN_E_WS, "osn_coords_unique_async: workspace %zu < %zu", ws_bytes, w.bytes);
    return queue_unique(coords4, n_max, n_dev, stride, table_keys, table_vals, cap, out_coords4, inverse, first, w, err_dev,
                        count_dev, st);
}

// The whole pyramid from one call (round 6): what coords_pyramid() issued as one osn_coords_unique_async per level -- per level two
// memsets and seven launches, 45 dependent launches of 2 - 9 us for five levels -- with ONE preset launch for every table and the
// counters, and the unique-flag pass folded into the scan: 1 + 6 per level.  Level i is built from level i - 1's output with its row
// count still in device memory; results are bit for bit those of the per-level calls.
extern "C" int osn_coords_pyramid_async(const int32_t* coords4, int64_t n0, const int32_t* strides, int n_levels,
                                        uint64_t* const* table_keys, int32_t* const* table_vals, int64_t cap,
                                        int32_t* const* out_coords4, int32_t* const* inverse, int32_t* const* first,
                                        int32_t* counts_dev, void* ws, size_t ws_bytes, osn_stream_t stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    OSN_REQUIRE(n0 >= 1 && n0 < (int64_t(1) << 31) && n_levels >= 1 && n_levels <= 16, OSN_E_ARG, "osn_coords_pyramid_async: n0=%lld, %d levels", (long long)n0, n_levels);
    OSN_REQUIRE(cap >= 2 * n0 && (cap & (cap - 1)) == 0, OSN_E_ARG, "osn_coords_pyramid_async: cap=%lld must be a power of two >= 2 n0", (long long)cap);
    OSN_REQUIRE(coords4 && strides && table_keys && table_vals && out_coords4 && inverse && first && counts_dev, OSN_E_ARG, "osn_coords_pyramid_async: null pointer");
    UniqueWs w = carve_unique(ws, n0);
    OSN_REQUIRE(ws && ws_bytes >= w.bytes, OSN_E_WS, "osn_coords_pyramid_async: workspace %zu < %zu", ws_bytes, w.bytes);
    FillJob fills[2 * 16 + 1];
    int nf = 0;
    fills[nf++] = FillJob{counts_dev, uint64_t(n_levels + 1) * 4u, 0u, 0u};
    for (int i = 0; i < n_levels; ++i) {
        OSN_REQUIRE(strides[i] >= 1 && table_keys[i] && table_vals[i] && out_coords4[i] && inverse[i] && first[i] && aligned16(out_coords4[i]), OSN_E_ARG,
                    "osn_coords_pyramid_async: level %d: null / unaligned pointer or stride < 1", i);
        fills[nf++] = FillJob{table_keys[i], uint64_t(cap) * 8u, 0xFFFFFFFFu, 0u};
        fills[nf++] = FillJob{table_vals[i], uint64_t(cap) * 4u, 0x7F7F7F7Fu, 0u};
    }
    OSN_REQUIRE(aligned16(coords4), OSN_E_ARG, "osn_coords_pyramid_async: coords must be 16-byte aligned");
    int rc = fill_batch(fills, nf, st);
    if (rc) return rc;
    const int32_t* prev = coords4;
    const int32_t* n_dev = nullptr;
    for (int i = 0; i < n_levels; ++i) {
        rc = queue_unique(prev, n0, n_dev, strides[i], table_keys[i], table_vals[i], cap, out_coords4[i], inverse[i], first[i], w,
                          counts_dev + n_levels, counts_dev + i, st, true);
        if (rc) return rc;
        prev = out_coords4[i];
        n_dev = counts_dev + i;
    }
    return OSN_OK;
}

extern "C" int osn_kmap_build(const uint64_t* in_table_keys, const int32_t* in_table_vals, int64_t cap,
                              const int32_t* out_coords4, int64_t n_out, int ksize, int offset_scale, int32_t* nbr,
                              int64_t* counts, osn_stream_t stream) {
    return kmap_build_impl(in_table_keys, in_table_vals, cap, out_coords4, n_out, ksize, offset_scale, nbr, counts, false,
                           static_cast<hipStream_t>(stream));
}

int osn::kmap_build_impl(const uint64_t* in_table_keys, const int32_t* in_table_vals, int64_t cap, const int32_t* out_coords4,
                         int64_t n_out, int ksize, int offset_scale, int32_t* nbr, int64_t* counts, bool prefilled, hipStream_t st) {
    OSN_REQUIRE(ksize >= 1 && ksize <= 7, OSN_E_ARG, "osn_kmap_build: ksize=%d unsupported", ksize);
    OSN_REQUIRE(cap >= 2 && (cap & (cap - 1)) == 0, OSN_E_ARG, "osn_kmap_build: cap must be a power of two");
    OSN_REQUIRE(n_out >= 0, OSN_E_ARG, "osn_kmap_build: n_out < 0");
    if (counts && !prefilled) OSN_HIP(hipMemsetAsync(counts, 0, size_t(ksize) * ksize * ksize * 8, st));
    if (n_out == 0) return OSN_OK;
    OSN_REQUIRE(in_table_keys && in_table_vals && out_coords4 && nbr, OSN_E_ARG, "osn_kmap_build: null pointer");
    OSN_REQUIRE(aligned16(out_coords4), OSN_E_ARG, "osn_kmap_build: out_coords4 must be 16-byte aligned");
    const int K = ksize * ksize * ksize;
    const int T = 256;
    hipLaunchKernelGGL(kmap_build_kernel, dim3(cdiv(n_out, T), K), dim3(T), 0, st, in_table_keys, in_table_vals,
                       uint32_t(cap - 1), reinterpret_cast<const int4*>(out_coords4), n_out, ksize, offset_scale, nbr,
                       reinterpret_cast<unsigned long long*>(counts));
    OSN_LAUNCH_CHECK();
    return OSN_OK;
}

extern "C" int osn_kmap_build_self(const uint64_t* table_keys, const int32_t* table_vals, int64_t cap,
                                   const int32_t* coords4, int64_t n, int ksize, int offset_scale, int32_t* nbr,
                                   int64_t* counts, osn_stream_t stream) {
    return kmap_build_self_impl(table_keys, table_vals, cap, coords4, n, ksize, offset_scale, nbr, counts, false, static_cast<hipStream_t>(stream));
}

int osn::kmap_build_self_impl(const uint64_t* table_keys, const int32_t* table_vals, int64_t cap, const int32_t* coords4, int64_t n,
                              int ksize, int offset_scale, int32_t* nbr, int64_t* counts, bool prefilled, hipStream_t st) {
    OSN_REQUIRE(ksize >= 1 && ksize <= 7 && (ksize & 1) == 1, OSN_E_ARG, "osn_kmap_build_self: ksize=%d must be odd, <= 7", ksize);
    OSN_REQUIRE(cap >= 2 && (cap & (cap - 1)) == 0, OSN_E_ARG, "osn_kmap_build_self: cap must be a power of two");
    OSN_REQUIRE(n >= 0, OSN_E_ARG, "osn_kmap_build_self: n < 0");
    const int K = ksize * ksize * ksize;
    if (counts && !prefilled) OSN_HIP(hipMemsetAsync(counts, 0, size_t(K) * 8, st));
    if (n == 0) return OSN_OK;
    OSN_REQUIRE(table_keys && table_vals && coords4 && nbr, OSN_E_ARG, "osn_kmap_build_self: null pointer");
    OSN_REQUIRE(aligned16(coords4), OSN_E_ARG, "osn_kmap_build_self: coords4 must be 16-byte aligned");
    if (K > 1 && !prefilled) OSN_HIP(hipMemsetAsync(nbr + int64_t(K / 2 + 1) * n, 0xFF, size_t(K / 2) * size_t(n) * 4, st));
    const int T = 256;
    hipLaunchKernelGGL(kmap_build_self_kernel, dim3(cdiv(n, T), K / 2 + 1), dim3(T), 0, st, table_keys, table_vals,
                       uint32_t(cap - 1), reinterpret_cast<const int4*>(coords4), n, ksize, offset_scale, nbr,
                       reinterpret_cast<unsigned long long*>(counts));
    OSN_LAUNCH_CHECK();
    return OSN_OK;
}

extern "C" int osn_kmap_transpose(const int32_t* nbr, int64_t n_out, int K, int64_t n_in, int32_t* tbl,
                                  osn_stream_t stream) {
    return kmap_transpose_impl(nbr, n_out, K, n_in, tbl, false, static_cast<hipStream_t>(stream));
}

int osn::kmap_transpose_impl(const int32_t* nbr, int64_t n_out, int K, int64_t n_in, int32_t* tbl, bool prefilled, hipStream_t st) {
    OSN_REQUIRE(K >= 1 && n_out >= 0 && n_in >= 0, OSN_E_ARG, "osn_kmap_transpose: bad sizes");
    if (n_in == 0) return OSN_OK;
    OSN_REQUIRE(tbl, OSN_E_ARG, "osn_kmap_transpose: null tbl");
    if (!prefilled) OSN_HIP(hipMemsetAsync(tbl, 0xFF, size_t(K) * size_t(n_in) * 4, st));
    if (n_out == 0) return OSN_OK;
    OSN_REQUIRE(nbr, OSN_E_ARG, "osn_kmap_transpose: null nbr");
    const int T = 256;
    hipLaunchKernelGGL(kmap_transpose_kernel, dim3(cdiv(n_out, T), K), dim3(T), 0, st, nbr, n_out, n_in, tbl);
    OSN_LAUNCH_CHECK();
    return OSN_OK;
}

extern "C" int osn_kmap_count(const int32_t* nbr, int64_t n_out, int K, int64_t* counts, osn_stream_t stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    OSN_REQUIRE(K >= 1 && n_out >= 0 && counts, OSN_E_ARG, "osn_kmap_count: bad arguments");
    OSN_HIP(hipMemsetAsync(counts, 0, size_t(K) * 8, st));
    if (n_out == 0) return OSN_OK;
    OSN_REQUIRE(nbr, OSN_E_ARG, "osn_kmap_count: null nbr");
    const int T = 256;
    int gx = int(cdiv(n_out, T));
    if (gx > 256) gx = 256;
    hipLaunchKernelGGL(kmap_count_kernel, dim3(gx, K), dim3(T), 0, st, nbr, n_out,
                       reinterpret_cast<unsigned long long*>(counts));
    OSN_LAUNCH_CHECK();
    return OSN_OK;
}
