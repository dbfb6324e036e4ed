// Per-offset pair arrays of a kernel map ("pl" buffers, osn_pair_lists_build): layout shared by the weight gradient
// (wgrad_tl.hip) and the weight-stationary convolution of the small maps (spconv_ws.hip).
#pragma once
#include "common.h"

namespace osn {

constexpr int PL_KMAX = 128;          // offsets per map (5^3 = 125)
constexpr int PL_ITEMS = 512;         // work items per map: one round of 2 workgroups per CU
constexpr int PL_MIN_QUOTA = 256;     // pairs per item at least (bounds the partial-sum traffic of small maps)

// ---- layout of a pair-list buffer ("pl")
constexpr size_t PL_OFF_POFF = 0;                                   // int32 [PL_KMAX + 1] first pair of each offset
constexpr size_t PL_OFF_TOTAL = 1024;                               // int32 [PL_KMAX]     pairs of each offset
constexpr size_t PL_OFF_ITEMS = 2048;                               // int4  [PL_ITEMS]    (k, p0, p1, 0), k = -1 unused
constexpr size_t PL_OFF_RANGE = PL_OFF_ITEMS + size_t(PL_ITEMS) * 16;   // int2 [PL_KMAX]  items of each offset [first, last)
constexpr size_t PL_OFF_PAIRS = 16384;                              // int32 pin[cap], pout[cap], then the tile-prefix scratch

struct PlView {
    int32_t *poff, *total;
    int4* items;
    int2* range;
    int32_t *pin, *pout, *pref;
    size_t bytes;
};

static PlView pl_view(void* base, int64_t n_out, int K, int bm) {
    PlView v;
    char* p = static_cast<char*>(base);
    const size_t cap = size_t(K) * size_t(n_out > 0 ? n_out : 1);
    const size_t nt = size_t(cdiv(n_out > 0 ? n_out : 1, bm > 0 ? bm : 1));
    v.poff = reinterpret_cast<int32_t*>(p + PL_OFF_POFF);
    v.total = reinterpret_cast<int32_t*>(p + PL_OFF_TOTAL);
    v.items = reinterpret_cast<int4*>(p + PL_OFF_ITEMS);
    v.range = reinterpret_cast<int2*>(p + PL_OFF_RANGE);
    v.pin = reinterpret_cast<int32_t*>(p + PL_OFF_PAIRS);
    v.pout = v.pin + cap;
    v.pref = v.pout + cap;
    v.bytes = PL_OFF_PAIRS + (2 * cap + size_t(K) * nt) * 4;
    return v;
}

}  // namespace osn
