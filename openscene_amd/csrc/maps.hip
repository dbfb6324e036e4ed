// Every kernel map of a scene from ONE call, on several streams.
//
// Replaces the Python-level walk over [ME] CoordinateManager.kernel_map for the convolutions of a MinkUNet forward
// (models/mink_unet.py:47-113: one 5^3 map, five 3^3 maps, four 2^3 stride-2 maps and their transposes), i.e. what
// CoordinateManager.prebuild issued as ~90 separate C-ABI calls: per map the neighbour table (+ per-offset pair counts),
// the transposed table of a strided map, the tile-ordered copies, the tile lists and the pair lists.  The maps of
// different pyramid levels do not depend on each other, and each is a chain of latency-bound launches that leaves most of
// the chip idle -- so the chains are dealt to up to OSN_MAPS_MAX_STREAMS streams (fork after the pyramid, join at the end;
// nothing outside sees the extra streams).  Host-only code: it calls the library's own entry points, so every table is
// bit for bit what the per-map calls produce.
#include "common.h"
#include "events.h"
#include "coords_impl.h"
#include <vector>

using namespace osn;

extern "C" int osn_maps_build(const osn_map_level* levels, int n_levels, const osn_map_job* jobs, int n_jobs,
                              const osn_stream_t* streams, void* const* ws, const uint64_t* ws_bytes, int n_streams,
                              osn_events_t* events) {
    OSN_REQUIRE(levels && jobs && streams && ws && ws_bytes && n_levels >= 1 && n_jobs >= 0 && n_streams >= 1 &&
                    n_streams <= OSN_MAPS_MAX_STREAMS,
                OSN_E_ARG, "osn_maps_build: bad arguments (%d levels, %d jobs, %d streams)", n_levels, n_jobs, n_streams);
    OSN_REQUIRE(n_streams == 1 || (events && events_count(events) >= n_streams + 1), OSN_E_ARG,
                "osn_maps_build: %d streams need an event pool of at least %d events", n_streams, n_streams + 1);
    hipStream_t main = static_cast<hipStream_t>(streams[0]);
    // Round 6: what the per-map entry points preset with one hipMemsetAsync each (pair counts: zero; the upper half of a self map's
    // table and a transposed table: -1) is cleared for ALL jobs by one launch in front of the fork (~20 launches of 3 - 8 us per scene)
    {
        std::vector<FillJob> fills;
        for (int j = 0; j < n_jobs; ++j) {
            const osn_map_job& q = jobs[j];
            if (q.lvl_in < 0 || q.lvl_in >= n_levels || q.lvl_out < 0 || q.lvl_out >= n_levels || q.ksize < 1 || q.ksize > 7 || !q.nbr_fwd) continue;
            const int K = q.ksize * q.ksize * q.ksize;
            const int64_t n_in = levels[q.lvl_in].rows;
            if (q.counts) fills.push_back(FillJob{q.counts, uint64_t(K) * 8u, 0u, 0u});
            if (q.self_map) {
                if (K > 1 && n_in > 0) fills.push_back(FillJob{q.nbr_fwd + int64_t(K / 2 + 1) * n_in, uint64_t(K / 2) * uint64_t(n_in) * 4u, 0xFFFFFFFFu, 0u});
            } else if (q.nbr_bwd && n_in > 0) {
                fills.push_back(FillJob{q.nbr_bwd, uint64_t(K) * uint64_t(n_in) * 4u, 0xFFFFFFFFu, 0u});
            }
        }
        const int rc = fill_batch(fills.data(), int(fills.size()), main);
        if (rc) return rc;
    }
    // fork: the side streams see the pyramid (coordinates + hash tables) the main stream has queued
    if (n_streams > 1) {
        hipEvent_t e0 = events_get(events, 0);
        OSN_HIP(hipEventRecord(e0, main));
        for (int s = 1; s < n_streams; ++s) OSN_HIP(hipStreamWaitEvent(static_cast<hipStream_t>(streams[s]), e0, 0));
    }
    bool used[OSN_MAPS_MAX_STREAMS] = {false};
    for (int j = 0; j < n_jobs; ++j) {
        const osn_map_job& q = jobs[j];
        OSN_REQUIRE(q.lvl_in >= 0 && q.lvl_in < n_levels && q.lvl_out >= 0 && q.lvl_out < n_levels && q.ksize >= 1 && q.ksize <= 7,
                    OSN_E_ARG, "osn_maps_build: job %d: levels / kernel size", j);
        const int si = q.stream >= 0 && q.stream < n_streams ? q.stream : 0;
        used[si] = true;
        osn_stream_t st = streams[si];
        const osn_map_level& li = levels[q.lvl_in];
        const osn_map_level& lo = levels[q.lvl_out];
        const int K = q.ksize * q.ksize * q.ksize;
        OSN_REQUIRE(q.nbr_fwd && li.rows >= 1 && lo.rows >= 1, OSN_E_ARG, "osn_maps_build: job %d: null table or empty level", j);
        int rc;
        if (q.self_map) {
            OSN_REQUIRE(q.lvl_in == q.lvl_out && (q.ksize & 1), OSN_E_ARG, "osn_maps_build: job %d: a self map needs one level and an odd kernel", j);
            rc = kmap_build_self_impl(li.keys, li.vals, li.cap, li.coords4, li.rows, q.ksize, q.scale, q.nbr_fwd, q.counts, true,
                                      static_cast<hipStream_t>(st));
        } else {
            rc = kmap_build_impl(li.keys, li.vals, li.cap, lo.coords4, lo.rows, q.ksize, q.scale, q.nbr_fwd, q.counts, true,
                                 static_cast<hipStream_t>(st));
            if (!rc && q.nbr_bwd) rc = kmap_transpose_impl(q.nbr_fwd, lo.rows, K, li.rows, q.nbr_bwd, true, static_cast<hipStream_t>(st));
        }
        if (rc) return rc;
        // tile-ordered copies (rows sorted by offset-occupancy mask) of the forward / input-gradient tables
        if (q.sorted_fwd) {
            rc = osn_kmap_sort(q.nbr_fwd, lo.rows, K, q.counts, q.order_fwd, q.sorted_fwd, q.gmask_fwd, ws[si], size_t(ws_bytes[si]), st);
            if (rc) return rc;
        }
        if (q.sorted_bwd) {
            OSN_REQUIRE(q.nbr_bwd, OSN_E_ARG, "osn_maps_build: job %d: tile order of a missing input-gradient table", j);
            rc = osn_kmap_sort(q.nbr_bwd, li.rows, K, q.counts, q.order_bwd, q.sorted_bwd, q.gmask_bwd, ws[si], size_t(ws_bytes[si]), st);
            if (rc) return rc;
        }
        // tile lists (of the tile-ordered table where one exists) and the pair lists of the forward lists
        if (q.tl_fwd) {
            rc = osn_tile_lists_build(q.sorted_fwd ? q.sorted_fwd : q.nbr_fwd, lo.rows, K, q.bm_fwd, q.tl_fwd, st);
            if (rc) return rc;
            if (q.pl_fwd) {
                rc = osn_pair_lists_build(q.tl_fwd, q.sorted_fwd ? q.order_fwd : nullptr, lo.rows, K, q.bm_fwd, q.pl_fwd, st);
                if (rc) return rc;
            }
        }
        if (q.tl_bwd) {
            OSN_REQUIRE(q.nbr_bwd, OSN_E_ARG, "osn_maps_build: job %d: tile lists of a missing input-gradient table", j);
            rc = osn_tile_lists_build(q.sorted_bwd ? q.sorted_bwd : q.nbr_bwd, li.rows, K, q.bm_bwd, q.tl_bwd, st);
            if (rc) return rc;
        }
    }
    // join
    for (int s = 1; s < n_streams; ++s) {
        if (!used[s]) continue;
        hipEvent_t e = events_get(events, s);
        OSN_HIP(hipEventRecord(e, static_cast<hipStream_t>(streams[s])));
        OSN_HIP(hipStreamWaitEvent(main, e, 0));
    }
    return OSN_OK;
}
