// GPU-resident batch assembly for the distillation loader:
// after the voxeliser has picked one point per voxel, find for every voxel whether it carries a
// fused 2-D feature and which row of the compact feature matrix belongs to it
// (dataset/feature_loader.py:124-143, the index1 / chunk_ind / cumsum chain; :107-113,:166-171 for
// val/test), and write the [batch | x y z] coordinate rows of the collated batch
// (dataset/feature_loader.py:177-178,193-209).  Integer work, HBM/latency-bound, bit-exact.
#include "common.h"

namespace osn {

int exclusive_scan_i32(const int* in, int* out, int* sums, int64_t n, hipStream_t st);  // coords.hip
size_t exclusive_scan_sums_count(int64_t n);

__global__ void point_flag_kernel(const uint8_t* __restrict__ mask, int64_t n, int* __restrict__ flag) {
    const int64_t p = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (p < n) flag[p] = mask[p] ? 1 : 0;
}

// mask_vox[v] = mask[vox_ind[v]]; src_row[v] = rank[vox_ind[v]] or -1
__global__ void voxel_flag_kernel(const uint8_t* __restrict__ mask, const int* __restrict__ rank,
                                  const int64_t* __restrict__ vox_ind, int64_t n_vox, int64_t n_points,
                                  uint8_t* __restrict__ mask_vox, int64_t* __restrict__ src_row,
                                  int* __restrict__ flagv, int* __restrict__ err) {
    const int64_t v = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (v >= n_vox) return;
    const int64_t p = vox_ind[v];
    if (p < 0 || p >= n_points) {
        atomicOr(err, 1);
        mask_vox[v] = 0; src_row[v] = -1; flagv[v] = 0;
        return;
    }
    const int m = mask[p] ? 1 : 0;
    mask_vox[v] = uint8_t(m);
    src_row[v] = m ? int64_t(rank[p]) : int64_t(-1);
    flagv[v] = m;
}

__global__ void compact_rows_kernel(const int* __restrict__ flagv, const int* __restrict__ dest,
                                    const int64_t* __restrict__ src_row, int64_t n_vox, int64_t* __restrict__ indices) {
    const int64_t v = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (v < n_vox && flagv[v]) indices[dest[v]] = src_row[v];
}

__global__ void batch_coords_kernel(const int32_t* __restrict__ xyz3, int64_t n, int batch_index,
                                    int4* __restrict__ out4) {
    const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < n) out4[i] = make_int4(batch_index, xyz3[3 * i], xyz3[3 * i + 1], xyz3[3 * i + 2]);
}

struct RemapWs {
    int *flag, *rank, *sums, *flagv, *dest, *sumsv, *err;
    size_t bytes;
};

static RemapWs carve_remap(void* ws, int64_t n_points, int64_t n_vox) {
    RemapWs w;
    char* base = static_cast<char*>(ws);
    size_t off = 0;
    auto take = [&](size_t b) { char* q = base ? base + off : nullptr; off += align_up(b, 256); return q; };
    const size_t np = size_t(n_points > 0 ? n_points : 1), nv = size_t(n_vox > 0 ? n_vox : 1);
    w.flag = reinterpret_cast<int*>(take(np * 4));
    w.rank = reinterpret_cast<int*>(take(np * 4));
    w.sums = reinterpret_cast<int*>(take(exclusive_scan_sums_count(n_points) * 4));
    w.flagv = reinterpret_cast<int*>(take(nv * 4));
    w.dest = reinterpret_cast<int*>(take(nv * 4));
    w.sumsv = reinterpret_cast<int*>(take(exclusive_scan_sums_count(n_vox) * 4));
    w.err = reinterpret_cast<int*>(take(4));
    w.bytes = off;
    return w;
}

}  // namespace osn

using namespace osn;

extern "C" size_t osn_feature_remap_ws_bytes(int64_t n_points, int64_t n_vox) {
    return carve_remap(nullptr, n_points, n_vox).bytes;
}

extern "C" int osn_feature_remap(const uint8_t* mask_chunk, const int64_t* vox_ind, int64_t n_points, int64_t n_vox,
                                 uint8_t* mask_vox, int64_t* src_row, int64_t* indices, int64_t* n_sel_host, void* ws,
                                 size_t ws_bytes, osn_stream_t stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    OSN_REQUIRE(n_points >= 0 && n_points < (int64_t(1) << 31) && n_vox >= 0 && n_vox < (int64_t(1) << 31), OSN_E_ARG,
                "osn_feature_remap: sizes out of range");
    OSN_REQUIRE(n_sel_host, OSN_E_ARG, "osn_feature_remap: n_sel_host is null");
    *n_sel_host = 0;
    if (n_vox == 0) return OSN_OK;
    OSN_REQUIRE(mask_chunk && vox_ind && mask_vox && src_row && indices, OSN_E_ARG, "osn_feature_remap: null pointer");
    OSN_REQUIRE(n_points > 0, OSN_E_ARG, "osn_feature_remap: voxels without points");
    RemapWs w = carve_remap(ws, n_points, n_vox);
    OSN_REQUIRE(ws && ws_bytes >= w.bytes, OSN_E_WS, "osn_feature_remap: workspace %zu < %zu", ws_bytes, w.bytes);
    const int T = 256;
    OSN_HIP(hipMemsetAsync(w.err, 0, 4, st));
    hipLaunchKernelGGL(point_flag_kernel, dim3(cdiv(n_points, T)), dim3(T), 0, st, mask_chunk, n_points, w.flag);
    OSN_LAUNCH_CHECK();
    int rc = exclusive_scan_i32(w.flag, w.rank, w.sums, n_points, st);          // rank[p] = #True before p
    if (rc) return rc;
    hipLaunchKernelGGL(voxel_flag_kernel, dim3(cdiv(n_vox, T)), dim3(T), 0, st, mask_chunk, w.rank, vox_ind, n_vox,
                       n_points, mask_vox, src_row, w.flagv, w.err);
    OSN_LAUNCH_CHECK();
    rc = exclusive_scan_i32(w.flagv, w.dest, w.sumsv, n_vox, st);               // voxel order is kept
    if (rc) return rc;
    hipLaunchKernelGGL(compact_rows_kernel, dim3(cdiv(n_vox, T)), dim3(T), 0, st, w.flagv, w.dest, src_row, n_vox,
                       indices);
    OSN_LAUNCH_CHECK();
    int host[2] = {0, 0};
    OSN_HIP(hipMemcpyAsync(&host[0], w.sumsv + (exclusive_scan_sums_count(n_vox) - 1), 4, hipMemcpyDeviceToHost, st));
    OSN_HIP(hipMemcpyAsync(&host[1], w.err, 4, hipMemcpyDeviceToHost, st));
    OSN_HIP(hipStreamSynchronize(st));
    OSN_REQUIRE(host[1] == 0, OSN_E_RANGE, "osn_feature_remap: vox_ind outside [0, n_points)");
    *n_sel_host = host[0];
    return OSN_OK;
}

extern "C" int osn_batch_coords(const int32_t* xyz3, int64_t n, int batch_index, int32_t* out_coords4,
                                osn_stream_t stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    OSN_REQUIRE(n >= 0 && batch_index >= 0 && batch_index < 65535, OSN_E_ARG, "osn_batch_coords: bad arguments");
    if (n == 0) return OSN_OK;
    OSN_REQUIRE(xyz3 && out_coords4 && aligned16(out_coords4), OSN_E_ARG, "osn_batch_coords: null or unaligned pointer");
    hipLaunchKernelGGL(batch_coords_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, xyz3, n, batch_index,
                       reinterpret_cast<int4*>(out_coords4));
    OSN_LAUNCH_CHECK();
    return OSN_OK;
}
