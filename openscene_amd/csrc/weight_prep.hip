// All weight images of a network in ONE launch.
//
// A training step changes every convolution weight once (optimizer step), and every convolution needs one or two
// bf16-split images of its weight (forward, input gradient; plane layout of the first-generation kernel or fragment
// layout of the tile-list kernel).  Per-weight launches are 49 x 5-7 us of launch-bound work per step; here the host
// keeps a table of jobs in device memory (built once per model) and one launch walks all of them.  In eval mode the
// weights do not change and no launch is needed at all (openscene_amd/ops.py keys the images on the parameter version).
//
// Reference seam: the weights are MinkowskiConvolution.kernel / MinkowskiConvolutionTranspose.kernel
// (models/mink_unet.py:47-113, models/resnet_base.py); the images are private to this library.
#include "common.h"
#include "weight_prep.h"

namespace osn {

constexpr int WPB_THREADS = 256;
constexpr int WPB_PER_THREAD = 8;
constexpr int WPB_BLOCK = WPB_THREADS * WPB_PER_THREAD;      // elements of one plane per workgroup

__global__ __launch_bounds__(WPB_THREADS) void weight_prep_batch_kernel(const osn_prep_job* __restrict__ jobs, int n_jobs) {
    // job of this workgroup: last j with first_block[j] <= blockIdx.x  (uniform => scalar loads)
    int lo = 0, hi = n_jobs - 1;
    const int64_t b = blockIdx.x;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].first_block <= b) lo = mid; else hi = mid - 1;
    }
    const osn_prep_job j = jobs[lo];
    const int64_t e0 = (b - j.first_block) * WPB_BLOCK;
    const int nn = j.for_dgrad ? j.cin : j.cout, nc = j.for_dgrad ? j.cout : j.cin;
    const int64_t per_plane = j.layout == OSN_PREP_TL
                                  ? int64_t(j.K) * ((nc + 31) >> 5) * ((nn + 15) >> 4) * 512
                                  : int64_t(j.K) * nn * ((nc + 31) / 32 * 32);
    __bf16* out = static_cast<__bf16*>(j.out);
    if (j.layout == OSN_PREP_TL) {                 // fragment layout: a thread's 8 elements are one lane's 16 bytes of a fragment
        const int64_t e = e0 + int64_t(threadIdx.x) * WPB_PER_THREAD;
        if (e < per_plane) weight_prep_tl_group(j.W, j.K, j.cin, j.cout, j.flip, j.for_dgrad, e >> 3, out);
        return;
    }
#pragma unroll
    for (int i = 0; i < WPB_PER_THREAD; ++i) {
        const int64_t e = e0 + threadIdx.x + i * WPB_THREADS;
        if (e < per_plane) {
            if (j.layout == OSN_PREP_TL) weight_prep_tl_one(j.W, j.K, j.cin, j.cout, j.flip, j.for_dgrad, e, out);
            else weight_prep_x6_one(j.W, j.K, j.cin, j.cout, j.flip, j.for_dgrad, e, out);
        }
    }
}

}  // namespace osn

using namespace osn;

extern "C" int64_t osn_weight_prep_job_blocks(int K, int cin, int cout, int for_dgrad, int layout) {
    if (K < 1 || cin < 1 || cout < 1) return 0;
    const int nn = for_dgrad ? cin : cout, nc = for_dgrad ? cout : cin;
    const int64_t per_plane = layout == OSN_PREP_TL ? int64_t(K) * ((nc + 31) >> 5) * ((nn + 15) >> 4) * 512
                                                    : int64_t(K) * nn * ((nc + 31) / 32 * 32);
    return cdiv(per_plane, WPB_BLOCK);
}

extern "C" int osn_weight_prep_batch(const osn_prep_job* jobs_dev, int n_jobs, int64_t total_blocks, osn_stream_t stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    OSN_REQUIRE(n_jobs >= 0 && total_blocks >= 0 && total_blocks < (int64_t(1) << 31), OSN_E_ARG,
                "osn_weight_prep_batch: n_jobs=%d total_blocks=%lld", n_jobs, (long long)total_blocks);
    if (n_jobs == 0 || total_blocks == 0) return OSN_OK;
    OSN_REQUIRE(jobs_dev, OSN_E_ARG, "osn_weight_prep_batch: null job table");
    hipLaunchKernelGGL(weight_prep_batch_kernel, dim3(unsigned(total_blocks)), dim3(WPB_THREADS), 0, st, jobs_dev, n_jobs);
    OSN_LAUNCH_CHECK();
    return OSN_OK;
}
