// Hash voxelisation on gfx950 -- bit-exact with the reference's numpy path
// (dataset/voxelizer.py:117-129, dataset/voxelization_utils.py:9-22,112-132):
//   grid  = floor([xyz,1] @ T^T[:, :3])        float64
//   grid  = floor(grid - grid.min(0))
//   key   = FNV-1 64-bit over the three integer columns (multiply, then xor)
//   inds, inverse = np.unique(key, return_index=True, return_inverse=True)
//
// Rounding contract of the transform: numpy hands the [N,4]x[4,3] product to
// BLAS dgemm, whose micro-kernels accumulate k = 0..3 with fused multiply-adds:
//   acc = x*m0; acc = fma(y, m1, acc); acc = fma(z, m2, acc); acc = fma(1, m3, acc)
// (verified bit-for-bit against numpy on 60 000 values when this was written;
// tests/golden/voxelize_*.npz pin it).  The same chain is evaluated here with
// explicit round-to-nearest intrinsics so the compiler cannot re-associate.
//
// np.unique's "first occurrence in ascending key order" = a STABLE sort of
// (key, point index) followed by run heads; the sort is rocPRIM's LSD radix sort
// (the one library primitive in this file), everything else is hand-written.
#include "common.h"

#include <string.h>
#include <limits.h>
#include <rocprim/device/device_radix_sort.hpp>

namespace osn {

int exclusive_scan_i32(const int* in, int* out, int* sums, int64_t n, hipStream_t st);  // coords.hip
size_t exclusive_scan_sums_count(int64_t n);

constexpr unsigned long long FNV_OFFSET = 14695981039346656037ull;
constexpr unsigned long long FNV_PRIME = 1099511628211ull;

struct Mat34 {
    double m[12];
};

__global__ __launch_bounds__(256) void vox_transform_kernel(const double* __restrict__ xyz, int64_t n, Mat34 T,
                                                            double* __restrict__ grid, long long* __restrict__ gmin) {
    __shared__ long long smin[3][4];
    const int64_t p = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    long long mn[3] = {LLONG_MAX, LLONG_MAX, LLONG_MAX};
    if (p < n) {
        const double x = xyz[p * 3 + 0], y = xyz[p * 3 + 1], z = xyz[p * 3 + 2];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            double acc = __dmul_rn(x, T.m[j * 4 + 0]);
            acc = __fma_rn(y, T.m[j * 4 + 1], acc);
            acc = __fma_rn(z, T.m[j * 4 + 2], acc);
            acc = __dadd_rn(acc, T.m[j * 4 + 3]);
            const double g = floor(acc);
            grid[p * 3 + j] = g;
            mn[j] = (long long)g;
        }
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        long long v = mn[j];
        for (int d = 32; d >= 1; d >>= 1) {
            const long long o = __shfl_xor(v, d, 64);
            v = o < v ? o : v;
        }
        if ((threadIdx.x & 63) == 0) smin[j][threadIdx.x >> 6] = v;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        long long v = smin[threadIdx.x][0];
        for (int w = 1; w < 4; ++w) v = smin[threadIdx.x][w] < v ? smin[threadIdx.x][w] : v;
        if (v != LLONG_MAX) atomicMin(&gmin[threadIdx.x], v);
    }
}

__global__ void vox_key_kernel(double* __restrict__ grid, int64_t n, const long long* __restrict__ gmin,
                               unsigned long long* __restrict__ keys, int* __restrict__ idx) {
    const int64_t p = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (p >= n) return;
    unsigned long long h = FNV_OFFSET;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const double g = floor(grid[p * 3 + j] - double(gmin[j]));
        grid[p * 3 + j] = g;
        h *= FNV_PRIME;
        h ^= (unsigned long long)g;
    }
    keys[p] = h;
    idx[p] = int(p);
}

__global__ void fnv_hash_kernel(const double* __restrict__ grid, int64_t n, int ncol,
                                unsigned long long* __restrict__ keys) {
    const int64_t p = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (p >= n) return;
    unsigned long long h = FNV_OFFSET;
    for (int j = 0; j < ncol; ++j) {
        h *= FNV_PRIME;
        h ^= (unsigned long long)grid[p * ncol + j];
    }
    keys[p] = h;
}

__global__ void vox_head_kernel(const unsigned long long* __restrict__ ks, int64_t n, int* __restrict__ flag) {
    const int64_t j = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (j >= n) return;
    flag[j] = (j == 0 || ks[j] != ks[j - 1]) ? 1 : 0;
}

__global__ void vox_emit_kernel(const int* __restrict__ idx_sorted, const int* __restrict__ flag,
                                const int* __restrict__ excl, int64_t n, int64_t* __restrict__ inds,
                                int64_t* __restrict__ inverse) {
    const int64_t j = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const int f = flag[j];
    const int rank = excl[j] + f - 1;
    const int p = idx_sorted[j];
    inverse[p] = rank;
    if (f) inds[rank] = p;   // stable sort: the run head is the lowest point index
}

struct VoxWs {
    unsigned long long *keys, *keys_sorted;
    int *idx, *idx_sorted, *flag, *excl, *sums;
    long long* gmin;
    void* sort_tmp;
    size_t sort_bytes;
    size_t bytes;
};

static hipError_t sort_tmp_bytes(int64_t n, size_t* out) {
    size_t b = 0;
    hipError_t e = rocprim::radix_sort_pairs(nullptr, b, (unsigned long long*)nullptr, (unsigned long long*)nullptr,
                                             (int*)nullptr, (int*)nullptr, size_t(n > 0 ? n : 1), 0u, 64u,
                                             (hipStream_t)0);
    *out = b;
    return e;
}

static VoxWs carve_vox(void* ws, int64_t n, size_t sort_bytes) {
    VoxWs w;
    char* base = static_cast<char*>(ws);
    size_t off = 0;
    auto take = [&](size_t b) { char* q = base ? base + off : nullptr; off += align_up(b, 256); return q; };
    const size_t m = size_t(n > 0 ? n : 1);
    w.keys = reinterpret_cast<unsigned long long*>(take(m * 8));
    w.keys_sorted = reinterpret_cast<unsigned long long*>(take(m * 8));
    w.idx = reinterpret_cast<int*>(take(m * 4));
    w.idx_sorted = reinterpret_cast<int*>(take(m * 4));
    w.flag = reinterpret_cast<int*>(take(m * 4));
    w.excl = reinterpret_cast<int*>(take(m * 4));
    w.sums = reinterpret_cast<int*>(take(exclusive_scan_sums_count(n) * 4));
    w.gmin = reinterpret_cast<long long*>(take(256));
    w.sort_tmp = take(sort_bytes);
    w.sort_bytes = sort_bytes;
    w.bytes = off;
    return w;
}

}  // namespace osn

using namespace osn;

extern "C" size_t osn_voxelize_ws_bytes(int64_t n) {
    size_t sb = 0;
    if (sort_tmp_bytes(n, &sb) != hipSuccess) return 0;
    return carve_vox(nullptr, n, sb).bytes;
}

extern "C" int osn_voxelize_fnv(const double* xyz, int64_t n, const double* T12_host, double* grid_out, int64_t* inds,
                                int64_t* inverse, int64_t* n_vox_host, void* ws, size_t ws_bytes,
                                osn_stream_t stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    OSN_REQUIRE(n >= 0 && n < (int64_t(1) << 31), OSN_E_ARG, "osn_voxelize_fnv: n out of range");
    OSN_REQUIRE(n_vox_host && T12_host, OSN_E_ARG, "osn_voxelize_fnv: null host pointer");
    *n_vox_host = 0;
    if (n == 0) return OSN_OK;
    OSN_REQUIRE(xyz && grid_out && inds && inverse, OSN_E_ARG, "osn_voxelize_fnv: null pointer");
    size_t sb = 0;
    OSN_HIP(sort_tmp_bytes(n, &sb));
    VoxWs w = carve_vox(ws, n, sb);
    OSN_REQUIRE(ws && ws_bytes >= w.bytes, OSN_E_WS, "osn_voxelize_fnv: workspace %zu < %zu", ws_bytes, w.bytes);
    Mat34 T;
    for (int i = 0; i < 12; ++i) T.m[i] = T12_host[i];
    OSN_HIP(hipMemsetAsync(w.gmin, 0x7F, 24, st));   // 0x7F7F... = large positive
    const int TPB = 256;
    const dim3 grid(cdiv(n, TPB));
    hipLaunchKernelGGL(vox_transform_kernel, grid, dim3(TPB), 0, st, xyz, n, T, grid_out, w.gmin);
    hipLaunchKernelGGL(vox_key_kernel, grid, dim3(TPB), 0, st, grid_out, n, w.gmin, w.keys, w.idx);
    OSN_LAUNCH_CHECK();
    size_t tb = w.sort_bytes;
    OSN_HIP(rocprim::radix_sort_pairs(w.sort_tmp, tb, w.keys, w.keys_sorted, w.idx, w.idx_sorted, size_t(n), 0u, 64u, st));
    hipLaunchKernelGGL(vox_head_kernel, grid, dim3(TPB), 0, st, w.keys_sorted, n, w.flag);
    OSN_LAUNCH_CHECK();
    int rc = exclusive_scan_i32(w.flag, w.excl, w.sums, n, st);
    if (rc) return rc;
    hipLaunchKernelGGL(vox_emit_kernel, grid, dim3(TPB), 0, st, w.idx_sorted, w.flag, w.excl, n, inds, inverse);
    OSN_LAUNCH_CHECK();
    int total = 0;
    OSN_HIP(hipMemcpyAsync(&total, w.sums + (exclusive_scan_sums_count(n) - 1), 4, hipMemcpyDeviceToHost, st));
    OSN_HIP(hipStreamSynchronize(st));
    *n_vox_host = total;
    return OSN_OK;
}

namespace osn {
// ravel_hash_vec (dataset/voxelization_utils.py:25-41): key = ((c0 - min0) * (max1 - min1 + 1) + (c1 - min1)) * ... + (c_last - min_last)
__global__ void ravel_init_kernel(long long* mm, int ncol) {
    if (int(threadIdx.x) < ncol) {
        mm[threadIdx.x] = 0x7FFFFFFFFFFFFFFFll;          // column minimum
        mm[8 + threadIdx.x] = -0x7FFFFFFFFFFFFFFFll - 1;   // column maximum
    }
}

__global__ __launch_bounds__(256) void ravel_minmax_kernel(const double* __restrict__ grid, int64_t n, int ncol,
                                                           long long* __restrict__ mm) {
    long long lo[4], hi[4];
    for (int j = 0; j < 4; ++j) { lo[j] = 0x7FFFFFFFFFFFFFFFll; hi[j] = -0x7FFFFFFFFFFFFFFFll - 1; }
    for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x)
        for (int j = 0; j < ncol; ++j) {
            const long long v = (long long)grid[i * ncol + j];
            lo[j] = v < lo[j] ? v : lo[j];
            hi[j] = v > hi[j] ? v : hi[j];
        }
    for (int j = 0; j < ncol; ++j) {
        for (int d = 32; d >= 1; d >>= 1) {
            const long long a = __shfl_down(lo[j], d, 64), b = __shfl_down(hi[j], d, 64);
            lo[j] = a < lo[j] ? a : lo[j];
            hi[j] = b > hi[j] ? b : hi[j];
        }
        if ((threadIdx.x & 63) == 0) {          // min / max are order-independent: atomics stay deterministic
            atomicMin(&mm[j], lo[j]);
            atomicMax(&mm[8 + j], hi[j]);
        }
    }
}

__global__ void ravel_keys_kernel(const double* __restrict__ grid, int64_t n, int ncol, const long long* __restrict__ mm,
                                  unsigned long long* __restrict__ keys) {
    const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned long long key = 0;
    for (int j = 0; j < ncol - 1; ++j) {
        key += (unsigned long long)((long long)grid[i * ncol + j] - mm[j]);
        key *= (unsigned long long)(mm[8 + j + 1] - mm[j + 1] + 1);
    }
    key += (unsigned long long)((long long)grid[i * ncol + ncol - 1] - mm[ncol - 1]);
    keys[i] = key;
}
}  // namespace osn

extern "C" int osn_ravel_hash(const double* grid, int64_t n, int ncol, uint64_t* keys, void* ws, size_t ws_bytes,
                              osn_stream_t stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    OSN_REQUIRE(n >= 0 && ncol >= 1 && ncol <= 4, OSN_E_ARG, "osn_ravel_hash: bad sizes (ncol <= 4)");
    if (n == 0) return OSN_OK;
    OSN_REQUIRE(grid && keys && ws && ws_bytes >= 128, OSN_E_ARG, "osn_ravel_hash: null pointer or workspace < 128 bytes");
    long long* mm = static_cast<long long*>(ws);
    hipLaunchKernelGGL(osn::ravel_init_kernel, dim3(1), dim3(64), 0, st, mm, ncol);
    int g = int(osn::cdiv(n, 256));
    if (g > 1024) g = 1024;
    hipLaunchKernelGGL(osn::ravel_minmax_kernel, dim3(g), dim3(256), 0, st, grid, n, ncol, mm);
    hipLaunchKernelGGL(osn::ravel_keys_kernel, dim3(unsigned(osn::cdiv(n, 256))), dim3(256), 0, st, grid, n, ncol, mm,
                       reinterpret_cast<unsigned long long*>(keys));
    OSN_LAUNCH_CHECK();
    return OSN_OK;
}

extern "C" int osn_fnv_hash(const double* grid, int64_t n, int ncol, uint64_t* keys, osn_stream_t stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    OSN_REQUIRE(n >= 0 && ncol >= 1, OSN_E_ARG, "osn_fnv_hash: bad sizes");
    if (n == 0) return OSN_OK;
    OSN_REQUIRE(grid && keys, OSN_E_ARG, "osn_fnv_hash: null pointer");
    hipLaunchKernelGGL(fnv_hash_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, grid, n, ncol,
                       reinterpret_cast<unsigned long long*>(keys));
    OSN_LAUNCH_CHECK();
    return OSN_OK;
}
