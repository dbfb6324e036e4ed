// Internal entry points of coords.hip for the batched map builder (maps.hip): the same launches as the extern "C" functions, with the
// option of leaving out their memsets ("prefilled": osn_maps_build clears the regions of ALL its jobs with one launch, fill_batch).
#pragma once
#include "common.h"

namespace osn {

// Regions a launch presets: byte pattern `value` over [ptr, ptr + bytes); ptr and bytes multiples of 4.
struct FillJob {
    void* ptr;
    uint64_t bytes;
    uint32_t value;          // the byte, replicated (0x00000000 / 0xFFFFFFFF / 0x7F7F7F7F)
    uint32_t pad;
};
constexpr int FILL_BATCH = 48;
struct FillJobs {
    FillJob j[FILL_BATCH];
};
// every job of `jobs[0 .. n)` in ONE launch per FILL_BATCH jobs (n == 0: nothing)
int fill_batch(const FillJob* jobs, int n, hipStream_t st);

int kmap_build_impl(const uint64_t* in_table_keys, const int32_t* in_table_vals, int64_t cap, const int32_t* out_coords4, int64_t n_out,
                    int ksize, int offset_scale, int32_t* nbr, int64_t* counts, bool prefilled, hipStream_t st);
int kmap_build_self_impl(const uint64_t* table_keys, const int32_t* table_vals, int64_t cap, const int32_t* coords4, int64_t n, int ksize,
                         int offset_scale, int32_t* nbr, int64_t* counts, bool prefilled, hipStream_t st);
int kmap_transpose_impl(const int32_t* nbr, int64_t n_out, int K, int64_t n_in, int32_t* tbl, bool prefilled, hipStream_t st);

}  // namespace osn
