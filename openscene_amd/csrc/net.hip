// Network executor: every launch of a MinkUNet forward / backward pass from ONE C call.
//
// Replaces the Python-level walk of models/mink_unet.py:116-174 and of its autograd graph: the host
// side compiles the module tree once into a linear program of stages (include/openscene_amd.h: osn_net_op) and this
// file plays it -- per stage the same kernel choice the per-module path makes (stem / tile-list / split-bf16
// output-stationary kernel; pair-array or table weight gradient), training-mode statistics kept for the backward
// pass, gradient sums formed inside the batch-norm backward (osn_bn_backward_multi), ME.cat written in place by its
// two producers (osn_bn_apply2).  Host-only code: no kernels here, only calls into the library's own entry points,
// so results are those of the per-module path (same kernels, same order; only the gradient SUMS are associated
// differently: (a + b) inside the consumer instead of a separate add).
#include "common.h"
#include <stdlib.h>
#include "events.h"
#include "epilogue.h"
#include <vector>

namespace osn {

constexpr uint64_t NO_OFF = ~uint64_t(0);

struct Events {
    std::vector<hipEvent_t> ev;
};

int events_count(const osn_events_t* e) { return e ? int(reinterpret_cast<const Events*>(e)->ev.size()) : 0; }
hipEvent_t events_get(const osn_events_t* e, int i) { return reinterpret_cast<const Events*>(e)->ev[size_t(i)]; }

struct Prof {
    std::vector<hipEvent_t> ev;        // 2 per record
    std::vector<int32_t> tag;
    std::vector<int32_t> only;         // tags to bracket; empty = every launch
    int cap = 0, n = 0;
};

static inline uint64_t up256(uint64_t x) { return (x + 255) & ~uint64_t(255); }

static bool stem_eligible(int K, int cin, int cout) { return cin <= 4 && cout == 32 && K > 1 && K <= 125; }
static bool tl_eligible(int K, int cin, int cout, int64_t n_in) {
    return (cin & 3) == 0 && cin >= 8 && cin <= 512 && (cout & 3) == 0 && K <= 128 && n_in <= (int64_t(1) << 24);
}
// functional.tl_rows_ok: a table big enough for the tile-list kernel (always from tl_min_rows rows on; from tl_mid_rows on
// when both channel counts are at least 96)
static bool tl_rows_ok(const osn_net_desc* net, int64_t rows, int ca, int cb) {
    return rows >= net->tl_min_rows || (net->tl_mid_rows > 0 && rows >= net->tl_mid_rows && ca >= 96 && cb >= 96);
}
// functional.ws_kernel: the weight-stationary kernel for a launch that gathers n_src rows of ca channels and writes n_dst
// rows of cb channels -- direct (no partial rows) when the destination is the fine side of a 2^3 stride-2 map, else on the
// maps of at most ws_max_rows destination rows.  0 = neither.
static int ws_kernel(const osn_net_desc* net, const osn_net_op& o, int ca, int cb, int64_t n_src, int64_t n_dst, bool dst_fine) {
    if (o.K <= 1 || !tl_eligible(o.K, ca, cb, n_src)) return 0;
    if (o.fine_unique && dst_fine) return OSN_NET_K_WS_DIRECT;
    if (net->ws_max_rows > 0 && n_dst <= net->ws_max_rows) return OSN_NET_K_WS;
    return 0;
}
static bool dense_eligible(int cin, int cout) { return (cin & 3) == 0 && cin >= 8 && (cout & 3) == 0; }
// functional.rg_kernel: the register-gather kernel for the narrow layers (32 / 64 channels on both sides) that neither the tile-list
// kernel nor a direct weight-stationary launch takes -- before the partial-row weight-stationary kernel and the first-generation one
static bool rg_eligible(int K, int ca, int cb, int64_t n_src) { return osn_spconv_fwd_rg_ok(n_src > 0 ? n_src : 1, K, ca, cb) != 0; }
// ... and AHEAD of the tile-list kernel when one side has 32 channels (measured at 101 k rows, tools/micro_rg.py: 32 -> 32 38.8 us against
// 49.3, 32 -> 64 49.6 / 53.7; 64 -> 64 stays with the tile-list kernel there: 72 against 112)
static bool rg_first(int K, int ca, int cb, int64_t n_src) { return (ca == 32 || cb == 32) && rg_eligible(K, ca, cb, n_src); }
static bool x6_eligible(int K, int cin, int cout, int64_t n_out) {
    if ((cin & 3) || cin < 8) return false;
    if (int64_t(3) * K * cout * ((cin + 31) / 32 * 32) >= (int64_t(1) << 30)) return false;
    int32_t plan[6];
    if (osn_spconv_fwd_plan(n_out, K, cin, cout, plan) != OSN_OK) return false;
    const int S = plan[4] > 0 ? plan[4] : 1;
    return (K + S - 1) / S <= 32;
}

// Everything the passes derive from (program, level sizes): kernel per stage, arena offsets, scratch size.
struct Layout {
    std::vector<uint64_t> x_off, stat_off, y_off, gx_off, gres_off, gin_off, gpart_off;
    uint64_t rows_in_off = NO_OFF, rows_gin_off = NO_OFF;   // output op only: gathered input rows / their input gradients (row-sparse goutput)
    std::vector<int32_t> fwd_k, dgrad_k, wgrad_k, images;
    std::vector<int32_t> producer;       // buffer -> op with dst == buffer
    std::vector<uint8_t> side;           // stage may run beside the main chain: a block's 1x1 shortcut (conv + BN), see below
    uint64_t fwd_bytes = 0, bwd_bytes = 0, ws_bytes = 0;
};

static int check_desc(const osn_net_desc* net, const int64_t* rows) {
    OSN_REQUIRE(net && rows && net->ops && net->bufs, OSN_E_ARG, "osn_net: null descriptor");
    OSN_REQUIRE(net->n_ops >= 1 && net->n_bufs >= 0 && net->n_levels >= 1 && net->n_levels <= OSN_NET_MAX_LEVELS, OSN_E_ARG,
                "osn_net: %d ops, %d buffers, %d levels", net->n_ops, net->n_bufs, net->n_levels);
    for (int l = 0; l < net->n_levels; ++l)
        OSN_REQUIRE(rows[l] >= 1 && rows[l] < (int64_t(1) << 31), OSN_E_ARG, "osn_net: level %d has %lld rows", l, (long long)rows[l]);
    for (int b = 0; b < net->n_bufs; ++b)
        OSN_REQUIRE(net->bufs[b].level >= 0 && net->bufs[b].level < net->n_levels && net->bufs[b].channels >= 4 &&
                        (net->bufs[b].channels & 3) == 0,
                    OSN_E_ARG, "osn_net: buffer %d (level %d, %d channels)", b, net->bufs[b].level, net->bufs[b].channels);
    for (int i = 0; i < net->n_ops; ++i) {
        const osn_net_op& o = net->ops[i];
        OSN_REQUIRE(o.K >= 1 && o.K <= 125 && o.cin >= 1 && o.cout >= 4 && (o.cout & 3) == 0, OSN_E_ARG, "osn_net: op %d shape (K=%d, %d -> %d)", i, o.K, o.cin, o.cout);
        OSN_REQUIRE(o.lvl_in >= 0 && o.lvl_in < net->n_levels && o.lvl_out >= 0 && o.lvl_out < net->n_levels, OSN_E_ARG, "osn_net: op %d levels", i);
        OSN_REQUIRE((o.K == 1) == (o.map < 0) && o.map < net->n_maps, OSN_E_ARG, "osn_net: op %d map index %d for K=%d", i, o.map, o.K);
        OSN_REQUIRE(o.K > 1 || o.lvl_in == o.lvl_out, OSN_E_ARG, "osn_net: op %d is a 1x1 conv across levels", i);
        OSN_REQUIRE(o.src >= -1 && o.src < net->n_bufs && o.dst >= -1 && o.dst < net->n_bufs && o.res >= -1 && o.res < net->n_bufs &&
                        o.copy_buf >= -1 && o.copy_buf < net->n_bufs,
                    OSN_E_ARG, "osn_net: op %d buffer index", i);
        OSN_REQUIRE(o.weight >= 0 && o.weight < net->n_weights && o.bn >= -1 && o.bn < net->n_bns, OSN_E_ARG, "osn_net: op %d weight / bn index", i);
        OSN_REQUIRE((o.dst >= 0) == (o.bn >= 0), OSN_E_ARG, "osn_net: op %d: exactly the stages with a batch norm write an activation buffer", i);
        if (o.src >= 0)
            OSN_REQUIRE(net->bufs[o.src].level == o.lvl_in && net->bufs[o.src].channels == o.cin, OSN_E_ARG, "osn_net: op %d reads buffer %d of another shape", i, o.src);
        if (o.dst >= 0)
            OSN_REQUIRE(net->bufs[o.dst].level == o.lvl_out && net->bufs[o.dst].channels == o.cout, OSN_E_ARG, "osn_net: op %d writes buffer %d of another shape", i, o.dst);
        if (o.res >= 0)
            OSN_REQUIRE(net->bufs[o.res].level == o.lvl_out && net->bufs[o.res].channels == o.cout, OSN_E_ARG, "osn_net: op %d residual buffer %d of another shape", i, o.res);
        if (o.copy_buf >= 0)
            OSN_REQUIRE(net->bufs[o.copy_buf].level == o.lvl_out && o.copy_col >= 0 && (o.copy_col & 3) == 0 &&
                            o.copy_col + o.cout <= net->bufs[o.copy_buf].channels,
                        OSN_E_ARG, "osn_net: op %d second store outside buffer %d", i, o.copy_buf);
    }
    return OSN_OK;
}

static int make_layout(const osn_net_desc* net, const int64_t* rows, int training, Layout& L) {
    int rc = check_desc(net, rows);
    if (rc) return rc;
    const int n = net->n_ops;
    L.x_off.assign(n, NO_OFF); L.stat_off.assign(n, NO_OFF); L.gx_off.assign(n, NO_OFF); L.gres_off.assign(n, NO_OFF);
    L.gin_off.assign(n, NO_OFF); L.gpart_off.assign(n, NO_OFF);
    L.y_off.assign(net->n_bufs, NO_OFF);
    L.fwd_k.assign(n, 0); L.dgrad_k.assign(n, 0); L.wgrad_k.assign(n, 0); L.images.assign(n, 0);
    L.producer.assign(net->n_bufs, -1);
    uint64_t f = 0, b = 0, ws = 4096;
    auto need_ws = [&](uint64_t v) { if (v > ws) ws = v; };
    for (int i = 0; i < n; ++i) {
        const osn_net_op& o = net->ops[i];
        const int64_t n_in = rows[o.lvl_in], n_out = rows[o.lvl_out];
        if (o.dst >= 0) {
            OSN_REQUIRE(L.producer[o.dst] < 0, OSN_E_ARG, "osn_net: buffer %d has two producers", o.dst);
            L.producer[o.dst] = i;
            L.x_off[i] = f; f += up256(uint64_t(n_out) * o.cout * 4);
            L.stat_off[i] = f; f += up256(uint64_t(2) * o.cout * 4);
            L.gx_off[i] = b; b += up256(uint64_t(n_out) * o.cout * 4);
            need_ws(osn_bn_ws_bytes(n_out, o.cout));
        }
        if (o.res >= 0) { L.gres_off[i] = b; b += up256(uint64_t(n_out) * o.cout * 4); }
        if (o.need_dgrad) { L.gin_off[i] = b; b += up256(uint64_t(n_in) * o.cin * 4); }
        if (o.dst < 0 && o.K == 1 && training) {          // room for osn_net_run.goutput_rows (at most every row)
            L.rows_in_off = b; b += up256(uint64_t(n_in) * o.cin * 4);
            L.rows_gin_off = b; b += up256(uint64_t(n_in) * o.cin * 4);
            need_ws(osn_spconv_wgrad_ws_bytes(n_out, 1, o.cin, o.cout));      // the row-compacted weight gradient runs on the table kernel
        }
        // ---- forward kernel
        if (stem_eligible(o.K, o.cin, o.cout)) {
            OSN_REQUIRE(!o.transposed, OSN_E_ARG, "osn_net: op %d: transposed stem", i);
            L.fwd_k[i] = OSN_NET_K_STEM;
        } else if (o.K == 1 && dense_eligible(o.cin, o.cout)) {
            L.fwd_k[i] = OSN_NET_K_DENSE;
            L.images[i] |= OSN_NET_IMG_TL_FWD;
        } else if (ws_kernel(net, o, o.cin, o.cout, n_in, n_out, o.transposed != 0) == OSN_NET_K_WS_DIRECT) {
            L.fwd_k[i] = OSN_NET_K_WS_DIRECT;
            L.images[i] |= OSN_NET_IMG_TL_FWD;
            need_ws(osn_spconv_fwd_ws_ws_bytes(n_out, o.K, o.cout, 1));
        } else if (o.K > 1 && tl_eligible(o.K, o.cin, o.cout, n_in) && tl_rows_ok(net, n_out, o.cin, o.cout) && !rg_first(o.K, o.cin, o.cout, n_in)) {
            L.fwd_k[i] = OSN_NET_K_TL;
            L.images[i] |= OSN_NET_IMG_TL_FWD;
            need_ws(osn_spconv_fwd_tl_ws_bytes(n_out, o.K, o.cout, osn_tile_rows(n_out)));
        } else if (rg_eligible(o.K, o.cin, o.cout, n_in)) {
            L.fwd_k[i] = OSN_NET_K_RG;
            L.images[i] |= OSN_NET_IMG_TL_FWD;
        } else if (ws_kernel(net, o, o.cin, o.cout, n_in, n_out, o.transposed != 0) == OSN_NET_K_WS) {
            L.fwd_k[i] = OSN_NET_K_WS;
            L.images[i] |= OSN_NET_IMG_TL_FWD;
            need_ws(osn_spconv_fwd_ws_ws_bytes(n_out, o.K, o.cout, 0));
        } else if (x6_eligible(o.K, o.cin, o.cout, n_out)) {
            L.fwd_k[i] = OSN_NET_K_X6;
            L.images[i] |= OSN_NET_IMG_X6_FWD;
            need_ws(osn_spconv_fwd_ws_bytes(n_out, o.K, o.cin, o.cout));
        } else {
            OSN_REQUIRE(false, OSN_E_ARG, "osn_net: op %d (K=%d, %d -> %d) fits none of the executor's forward kernels", i, o.K, o.cin, o.cout);
        }
        if (!training) continue;
        // ---- input gradient: a convolution of the output gradient with the transposed weights, [n_in, cin]
        if (o.need_dgrad) {
            const int wsk = tl_eligible(o.K, o.cin, o.cout, n_in) ? ws_kernel(net, o, o.cout, o.cin, n_out, n_in, o.transposed == 0) : 0;
            if (o.K == 1 && dense_eligible(o.cin, o.cout) && dense_eligible(o.cout, o.cin)) {
                L.dgrad_k[i] = OSN_NET_K_DENSE;
                L.images[i] |= OSN_NET_IMG_TL_DGRAD;
            } else if (wsk == OSN_NET_K_WS_DIRECT) {
                L.dgrad_k[i] = OSN_NET_K_WS_DIRECT;
                L.images[i] |= OSN_NET_IMG_TL_DGRAD;
                need_ws(osn_spconv_fwd_ws_ws_bytes(n_in, o.K, o.cin, 1));
            } else if (o.K > 1 && tl_eligible(o.K, o.cout, o.cin, n_out) && tl_eligible(o.K, o.cin, o.cout, n_in) && tl_rows_ok(net, n_in, o.cin, o.cout) &&
                       !rg_first(o.K, o.cout, o.cin, n_out)) {
                L.dgrad_k[i] = OSN_NET_K_TL;
                L.images[i] |= OSN_NET_IMG_TL_DGRAD;
                need_ws(osn_spconv_fwd_tl_ws_bytes(n_in, o.K, o.cin, osn_tile_rows(n_in)));
            } else if (rg_eligible(o.K, o.cout, o.cin, n_out)) {
                L.dgrad_k[i] = OSN_NET_K_RG;
                L.images[i] |= OSN_NET_IMG_TL_DGRAD;
            } else if (wsk == OSN_NET_K_WS) {
                L.dgrad_k[i] = OSN_NET_K_WS;
                L.images[i] |= OSN_NET_IMG_TL_DGRAD;
                need_ws(osn_spconv_fwd_ws_ws_bytes(n_in, o.K, o.cin, 0));
            } else if (x6_eligible(o.K, o.cout, o.cin, n_in)) {
                L.dgrad_k[i] = OSN_NET_K_X6;
                L.images[i] |= OSN_NET_IMG_X6_DGRAD;
                need_ws(osn_spconv_fwd_ws_bytes(n_in, o.K, o.cout, o.cin));
            } else {
                OSN_REQUIRE(false, OSN_E_ARG, "osn_net: op %d (K=%d, %d -> %d) fits none of the executor's input-gradient kernels", i, o.K, o.cin, o.cout);
            }
        }
        // ---- weight gradient
        // pair-array kernel on every 3^3 / 2^3 map, and -- identity map -- for the 1x1 shortcut convs up to 256 x 256
        // channels (their partial tiles are summed by the pass's batched reductions; the 96 -> 768 head stays on the
        // table kernel: 188 us against 215 us measured).  Round 6: the limit was 128 x 128, which left the three wide shortcuts of
        // MinkUNet18A (128 -> 256 at 730 rows, 256 -> 128 at 3.3 k, 192 -> 128 at 13 k) on the first-generation table kernel:
        // 78 - 85 us each for microseconds of work (profiles/r05_s20_bench_detail.json, ops 20 / 26 / 32).
        const bool wg_tl = (o.K > 1 || (o.cin <= 256 && o.cout <= 256)) && tl_eligible(o.K, o.cin, o.cout, n_in) &&
                           !stem_eligible(o.K, o.cin, o.cout);
        if (stem_eligible(o.K, o.cin, o.cout)) {
            L.wgrad_k[i] = OSN_NET_K_WGRAD_STEM;
            need_ws(osn_stem_conv_wgrad_ws_bytes(o.K, o.cin));
        } else if (wg_tl) {
            L.wgrad_k[i] = OSN_NET_K_WGRAD_TL;
            // partial sums per work item live in the backward arena until the ONE batched reduction at the end of the pass
            L.gpart_off[i] = b; b += up256(osn_spconv_wgrad_tl_ws_bytes(o.K, o.cin, o.cout));
        } else {
            L.wgrad_k[i] = OSN_NET_K_WGRAD;
            need_ws(osn_spconv_wgrad_ws_bytes(n_out, o.K, o.cin, o.cout));
        }
    }
    // A BasicBlock's shortcut (1x1 conv + batch norm without ReLU, models/resnet_base.py:101-107) only meets the main chain
    // again as the residual of the block's last batch norm: it can run on a second stream beside conv1 - BN - conv2.
    L.side.assign(n, 0);
    for (int i = 0; i < n; ++i) {
        const osn_net_op& o = net->ops[i];
        if (o.K != 1 || o.bn < 0 || o.relu || o.src < 0 || o.dst < 0 || o.copy_buf >= 0 || o.res >= 0) continue;
        int as_res = 0, other = 0;
        for (int j = 0; j < n; ++j) {
            if (net->ops[j].res == o.dst) ++as_res;
            if (net->ops[j].src == o.dst || net->ops[j].copy_buf == o.dst) ++other;
        }
        L.side[i] = (as_res == 1 && other == 0) ? 1 : 0;
    }
    for (int bf = 0; bf < net->n_bufs; ++bf) {
        L.y_off[bf] = f;
        f += up256(uint64_t(rows[net->bufs[bf].level]) * net->bufs[bf].channels * 4);
    }
    L.fwd_bytes = f + 256;
    L.bwd_bytes = b + 256;
    L.ws_bytes = up256(ws);
    return OSN_OK;
}

struct Bracket {
    Prof* p;
    hipStream_t st;
    int slot;
    Bracket(osn_prof_t* prof, int op, int phase, hipStream_t s) : p(reinterpret_cast<Prof*>(prof)), st(s), slot(-1) {
        if (!p || p->n >= p->cap) return;
        if (!p->only.empty()) {
            bool hit = false;
            for (int32_t t : p->only) hit = hit || t == op * 4 + phase;
            if (!hit) return;
        }
        slot = p->n++;
        p->tag[slot] = op * 4 + phase;
        (void)hipEventRecord(p->ev[2 * slot], st);
    }
    ~Bracket() {
        if (slot >= 0) (void)hipEventRecord(p->ev[2 * slot + 1], st);
    }
};

// effective tables of a stage: a transposed convolution runs on the mirrored tables of the strided conv's map
struct MapView {
    const int32_t *nbr_f, *nbr_b, *tf_rows, *tf_tbl, *tb_rows, *tb_tbl;
    const uint32_t *tf_g, *tb_g;
    const void *tl_f, *tl_b;
    const int32_t *tl_f_rows, *tl_b_rows;
    int tl_f_bm, tl_b_bm;
};
static MapView view_of(const osn_net_map& m, bool transposed) {
    MapView v;
    if (!transposed) {
        v.nbr_f = m.nbr_fwd; v.nbr_b = m.nbr_bwd;
        v.tf_rows = m.tiles_fwd_rows; v.tf_tbl = m.tiles_fwd_tbl; v.tf_g = m.tiles_fwd_gmask;
        v.tb_rows = m.tiles_bwd_rows; v.tb_tbl = m.tiles_bwd_tbl; v.tb_g = m.tiles_bwd_gmask;
        v.tl_f = m.tl_fwd; v.tl_f_rows = m.tl_fwd_rows; v.tl_f_bm = m.tl_fwd_bm;
        v.tl_b = m.tl_bwd; v.tl_b_rows = m.tl_bwd_rows; v.tl_b_bm = m.tl_bwd_bm;
    } else {
        v.nbr_f = m.nbr_bwd; v.nbr_b = m.nbr_fwd;
        v.tf_rows = m.tiles_bwd_rows; v.tf_tbl = m.tiles_bwd_tbl; v.tf_g = m.tiles_bwd_gmask;
        v.tb_rows = m.tiles_fwd_rows; v.tb_tbl = m.tiles_fwd_tbl; v.tb_g = m.tiles_fwd_gmask;
        v.tl_f = m.tl_bwd; v.tl_f_rows = m.tl_bwd_rows; v.tl_f_bm = m.tl_bwd_bm;
        v.tl_b = m.tl_fwd; v.tl_b_rows = m.tl_fwd_rows; v.tl_b_bm = m.tl_fwd_bm;
    }
    return v;
}

// out[n_out, cout] = conv(in[n_in, cin]) over table side `fwd` of the view, with weight images `x6` / `tl`
// epi (evaluation-mode forward pass only): the stage's batch norm applied by the kernel that stores the final rows (epilogue.h);
// every kernel family but the first-generation one takes it (epi_fusable)
static bool epi_fusable(int kernel) {
    return kernel == OSN_NET_K_DENSE || kernel == OSN_NET_K_WS || kernel == OSN_NET_K_WS_DIRECT || kernel == OSN_NET_K_RG ||
           kernel == OSN_NET_K_STEM || kernel == OSN_NET_K_TL;
}
static int run_conv(int kernel, const float* in, int64_t n_in, float* out, int64_t n_out, int K, int cin, int cout,
                    const float* W, const void* img_x6, const void* img_tl, const int32_t* nbr, const int32_t* t_rows,
                    const int32_t* t_tbl, const uint32_t* t_g, const void* tl, const int32_t* tl_rows, int tl_bm,
                    const void* pl, int64_t pl_rows, int pl_swap, const osn_net_run* run, osn_stream_t stream, int op,
                    const Epi& epi = epi_none()) {
    OSN_REQUIRE(!epi.mean || epi_fusable(kernel), OSN_E_ARG, "osn_net: op %d: kernel %d takes no epilogue", op, kernel);
    switch (kernel) {
        case OSN_NET_K_DENSE:
            OSN_REQUIRE(img_tl && K == 1 && n_in == n_out, OSN_E_ARG, "osn_net: op %d: the 1x1 kernel needs the fragment-order weight image", op);
            return dense_fwd_epi(in, img_tl, out, n_out, cin, cout, epi, stream);
        case OSN_NET_K_WS:
        case OSN_NET_K_WS_DIRECT:
            OSN_REQUIRE(pl && img_tl && nbr, OSN_E_ARG, "osn_net: op %d: pair arrays / tile-list weight image / destination table missing", op);
            return spconv_fwd_ws_epi(in, n_in, img_tl, pl, pl_rows, pl_swap, kernel == OSN_NET_K_WS_DIRECT ? 1 : 0, nbr, out, n_out, K,
                                     cin, cout, run->ws, size_t(run->ws_bytes), epi, stream);
        case OSN_NET_K_RG:
            OSN_REQUIRE(img_tl && nbr, OSN_E_ARG, "osn_net: op %d: the register-gather kernel needs the tile-list weight image and the table", op);
            return spconv_fwd_rg_epi(in, n_in, img_tl, t_tbl ? t_tbl : nbr, t_tbl ? t_rows : nullptr, out, n_out, K, cin, cout, epi, stream);
        case OSN_NET_K_STEM:
            OSN_REQUIRE(nbr && W, OSN_E_ARG, "osn_net: op %d: the stem kernel needs the plain table and the fp32 weight", op);
            return stem_conv_fwd_epi(in, W, nbr, out, n_out, K, cin, cout, epi, stream);
        case OSN_NET_K_TL:
            OSN_REQUIRE(tl && img_tl && run->tl_counters, OSN_E_ARG, "osn_net: op %d: tile lists / tile-list weight image / counters missing", op);
            return spconv_fwd_tl_epi(in, n_in, img_tl, tl, tl_rows, out, n_out, K, cin, cout, tl_bm, run->ws, size_t(run->ws_bytes),
                                     run->tl_counters, epi, stream);
        case OSN_NET_K_X6: {
            OSN_REQUIRE(img_x6 && (K == 1 || nbr), OSN_E_ARG, "osn_net: op %d: split-bf16 weight image / table missing", op);
            const int32_t* tbl = K == 1 ? nullptr : (t_tbl ? t_tbl : nbr);
            const int32_t* rows = (K > 1 && t_tbl) ? t_rows : nullptr;
            const uint32_t* gm = (K > 1 && t_tbl) ? t_g : nullptr;
            return osn_spconv_fwd_x6(in, img_x6, tbl, rows, gm, out, n_out, K, cin, cout, run->ws, size_t(run->ws_bytes), stream);
        }
        default:
            OSN_REQUIRE(false, OSN_E_ARG, "osn_net: op %d has no kernel", op);
    }
    return OSN_OK;
}

static int check_run(const osn_net_desc* net, const osn_net_run* run, const Layout& L, bool backward) {
    OSN_REQUIRE(run && run->level_rows && run->weights && (net->n_bns == 0 || run->bns) && (net->n_maps == 0 || run->maps) && run->input,
                OSN_E_ARG, "osn_net: null run descriptor member");
    OSN_REQUIRE(run->first_op >= 0 && run->first_op <= run->end_op && run->end_op <= net->n_ops, OSN_E_ARG, "osn_net: op range [%d, %d)", run->first_op, run->end_op);
    OSN_REQUIRE(run->fwd_arena && run->fwd_arena_bytes >= L.fwd_bytes && aligned16(run->fwd_arena), OSN_E_WS,
                "osn_net: forward arena %llu < %llu bytes", (unsigned long long)run->fwd_arena_bytes, (unsigned long long)L.fwd_bytes);
    OSN_REQUIRE(run->ws && run->ws_bytes >= L.ws_bytes, OSN_E_WS, "osn_net: workspace %llu < %llu bytes",
                (unsigned long long)run->ws_bytes, (unsigned long long)L.ws_bytes);
    if (backward)
        OSN_REQUIRE(run->bwd_arena && run->bwd_arena_bytes >= L.bwd_bytes && aligned16(run->bwd_arena), OSN_E_WS,
                    "osn_net: backward arena %llu < %llu bytes", (unsigned long long)run->bwd_arena_bytes, (unsigned long long)L.bwd_bytes);
    return OSN_OK;
}

}  // namespace osn

using namespace osn;

extern "C" int osn_net_plan_query(const osn_net_desc* net, const int64_t* level_rows, int training, osn_net_plan* plan) {
    OSN_REQUIRE(plan, OSN_E_ARG, "osn_net_plan_query: null plan");
    Layout L;
    int rc = make_layout(net, level_rows, training, L);
    if (rc) return rc;
    plan->fwd_arena_bytes = L.fwd_bytes;
    plan->bwd_arena_bytes = L.bwd_bytes;
    plan->ws_bytes = L.ws_bytes;
    for (int i = 0; i < net->n_ops; ++i) {
        if (plan->x_off) plan->x_off[i] = L.x_off[i];
        if (plan->stat_off) plan->stat_off[i] = L.stat_off[i];
        if (plan->fwd_kernel) plan->fwd_kernel[i] = L.fwd_k[i];
        if (plan->dgrad_kernel) plan->dgrad_kernel[i] = L.dgrad_k[i];
        if (plan->wgrad_kernel) plan->wgrad_kernel[i] = L.wgrad_k[i];
        if (plan->images) plan->images[i] = L.images[i];
    }
    if (plan->y_off)
        for (int b = 0; b < net->n_bufs; ++b) plan->y_off[b] = L.y_off[b];
    return OSN_OK;
}

extern "C" int osn_net_forward(const osn_net_desc* net, const osn_net_run* run, osn_stream_t stream) {
    OSN_REQUIRE(net && run, OSN_E_ARG, "osn_net_forward: null argument");
    Layout L;
    int rc = make_layout(net, run->level_rows, run->training, L);
    if (rc) return rc;
    rc = check_run(net, run, L, false);
    if (rc) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    char* A = static_cast<char*>(run->fwd_arena);
    const int64_t* rows = run->level_rows;
    // optional second stream: the blocks' shortcut stages run beside conv1 - BN - conv2 (events: [i] fork, [n_ops + 1 + i] done)
    Events* evs = reinterpret_cast<Events*>(run->events);
    hipStream_t side = static_cast<hipStream_t>(run->side_stream);
    const bool forked = side && side != st && evs && int(evs->ev.size()) >= 2 * net->n_ops + 2 && run->ws_side &&
                        run->ws_side_bytes >= L.ws_bytes;
    std::vector<uint8_t> pending(size_t(net->n_ops), 0);       // side stages the main stream has not joined yet
    for (int i = run->first_op; i < run->end_op; ++i) {
        const osn_net_op& o = net->ops[i];
        const int64_t n_in = rows[o.lvl_in], n_out = rows[o.lvl_out];
        const osn_net_weight& w = run->weights[o.weight];
        const bool on_side = forked && L.side[i];
        // A shortcut stage reads the BLOCK INPUT, like the conv1 queued just before it: its fork event is recorded in front of that
        // conv1 (look-ahead), not behind it -- the stage then runs beside conv1 - BN - conv2 instead of starting after conv1's batch
        // norm, finishing after conv2 and stalling the main stream at the residual's join (35 - 49 us on the level-0 / level-1 blocks,
        // tools/gap_census.py; the event behind conv1 was round 4's A/B loser, profiles/r04_s15_to_s19_stream_queueing_ab.txt).
        if (forked && !L.side[i] && i + 1 < run->end_op && L.side[i + 1] && net->ops[i + 1].src == o.src && !pending[i + 1]) {
            OSN_HIP(hipEventRecord(evs->ev[i + 1], st));
            pending[i + 1] = 2;                                // (2 = fork event already recorded; becomes 1 when the stage is queued)
        }
        if (on_side) {                                         // fork: everything queued so far (the producer of src) is visible
            if (pending[i] != 2) OSN_HIP(hipEventRecord(evs->ev[i], st));
            pending[i] = 0;
            OSN_HIP(hipStreamWaitEvent(side, evs->ev[i], 0));
        }
        osn_net_run side_run = *run;                           // scratch of the stream the stage runs on
        if (on_side) { side_run.ws = run->ws_side; side_run.ws_bytes = run->ws_side_bytes; }
        const osn_net_run* r = on_side ? &side_run : run;
        const osn_stream_t sstream = on_side ? run->side_stream : stream;

        const float* in = o.src < 0 ? run->input : reinterpret_cast<const float*>(A + L.y_off[o.src]);
        float* x = o.dst < 0 ? run->output : reinterpret_cast<float*>(A + L.x_off[i]);
        OSN_REQUIRE(x, OSN_E_ARG, "osn_net_forward: op %d writes the network output but run->output is null", i);
        MapView v{};
        if (o.map >= 0) {
            OSN_REQUIRE(run->maps[o.map].K == o.K, OSN_E_ARG, "osn_net_forward: op %d (K=%d) got a map with K=%d", i, o.K, run->maps[o.map].K);
            v = view_of(run->maps[o.map], o.transposed != 0);
        }
        // Inference: the stage's batch norm (+ residual) (+ ReLU) (+ cat store) in the epilogue of the convolution's last kernel
        // (epilogue.h: bitwise the separate launch): the convolution writes y, the x buffer and the osn_bn_apply2 launch are skipped
        Epi epi = epi_none();
        if (!run->training && o.bn >= 0 && !(run->flags & OSN_NET_RUN_NO_BN_EPILOGUE) && epi_fusable(L.fwd_k[i])) {
            const osn_net_bn& bn = run->bns[o.bn];
            OSN_REQUIRE(bn.running_mean && bn.running_var, OSN_E_ARG, "osn_net_forward: op %d: evaluation-mode batch norm without running statistics", i);
            if (o.res >= 0 && forked) {                        // the residual is read by the convolution's kernel: join in front of it
                const int p = L.producer[o.res];
                if (p >= 0 && pending[p]) {
                    OSN_HIP(hipStreamWaitEvent(st, evs->ev[net->n_ops + 1 + p], 0));
                    pending[p] = 0;
                }
            }
            epi.mean = bn.running_mean; epi.var = bn.running_var; epi.gamma = bn.gamma; epi.beta = bn.beta; epi.eps = bn.eps;
            epi.res = o.res >= 0 ? reinterpret_cast<const float*>(A + L.y_off[o.res]) : nullptr;
            epi.relu = o.relu;
            if (o.copy_buf >= 0) {
                epi.ld2 = net->bufs[o.copy_buf].channels;
                epi.y2 = reinterpret_cast<float*>(A + L.y_off[o.copy_buf]) + o.copy_col;
            }
            x = reinterpret_cast<float*>(A + L.y_off[o.dst]);
        }
        {
            Bracket br(run->prof, i, 0, on_side ? side : st);
            // pair arrays: those of the map's own (strided / self) direction; a transposed conv walks them the other way
            rc = run_conv(L.fwd_k[i], in, n_in, x, n_out, o.K, o.cin, o.cout, w.W, w.x6_fwd, w.tl_fwd, v.nbr_f, v.tf_rows, v.tf_tbl,
                          v.tf_g, v.tl_f, v.tl_f_rows, v.tl_f_bm, o.map >= 0 ? run->maps[o.map].pl_fwd : nullptr,
                          o.transposed ? n_in : n_out, o.transposed ? 1 : 0, r, sstream, i, epi);
        }
        if (rc) return rc;
        if (o.bn < 0) continue;
        if (epi.mean) {                                        // the batch norm ran in the convolution's epilogue
            if (on_side) {
                OSN_HIP(hipEventRecord(evs->ev[net->n_ops + 1 + i], side));
                pending[i] = 1;
            }
            continue;
        }
        if (o.res >= 0 && forked) {                            // join: the residual may come from a side stage
            const int p = L.producer[o.res];
            if (p >= 0 && pending[p]) {
                OSN_HIP(hipStreamWaitEvent(st, evs->ev[net->n_ops + 1 + p], 0));
                pending[p] = 0;
            }
        }
        const osn_net_bn& bn = run->bns[o.bn];
        float* y = reinterpret_cast<float*>(A + L.y_off[o.dst]);
        const float* res = o.res >= 0 ? reinterpret_cast<const float*>(A + L.y_off[o.res]) : nullptr;
        float* y2 = nullptr;
        int64_t ld2 = 0;
        if (o.copy_buf >= 0) {
            ld2 = net->bufs[o.copy_buf].channels;
            y2 = reinterpret_cast<float*>(A + L.y_off[o.copy_buf]) + o.copy_col;
        }
        if (run->training) {
            float* mv = reinterpret_cast<float*>(A + L.stat_off[i]);
            rc = osn_bn_forward_train2(x, n_out, o.cout, bn.gamma, bn.beta, bn.eps, res, o.relu, bn.momentum, mv, mv + o.cout,
                                       bn.running_mean, bn.running_var, y, y2, ld2, r->ws, size_t(r->ws_bytes), sstream);
        } else {
            OSN_REQUIRE(bn.running_mean && bn.running_var, OSN_E_ARG, "osn_net_forward: op %d: evaluation-mode batch norm without running statistics", i);
            rc = osn_bn_apply2(x, bn.running_mean, bn.running_var, bn.gamma, bn.beta, bn.eps, res, o.relu, y, y2, ld2, n_out, o.cout, sstream);
        }
        if (rc) return rc;
        if (on_side) {
            OSN_HIP(hipEventRecord(evs->ev[net->n_ops + 1 + i], side));
            pending[i] = 1;
        }
    }
    for (int i = run->first_op; i < run->end_op; ++i)          // a side stage nobody consumed inside the executed range
        if (pending[i]) OSN_HIP(hipStreamWaitEvent(st, evs->ev[net->n_ops + 1 + i], 0));
    return OSN_OK;
}

extern "C" int osn_net_backward(const osn_net_desc* net, const osn_net_run* run, osn_stream_t stream) {
    OSN_REQUIRE(net && run, OSN_E_ARG, "osn_net_backward: null argument");
    Layout L;
    int rc = make_layout(net, run->level_rows, 1, L);
    if (rc) return rc;
    rc = check_run(net, run, L, true);
    if (rc) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    char* A = static_cast<char*>(run->fwd_arena);
    char* B = static_cast<char*>(run->bwd_arena);
    const int64_t* rows = run->level_rows;
    std::vector<osn_wgrad_job> jobs;
    jobs.reserve(size_t(net->n_ops));
    // optional second stream for the weight gradients
    Events* evs = reinterpret_cast<Events*>(run->events);
    hipStream_t side = static_cast<hipStream_t>(run->side_stream);
    const bool forked = side && side != st && evs && int(evs->ev.size()) >= 2 * net->n_ops + 2 && run->ws_side &&
                        run->ws_side_bytes >= L.ws_bytes;
    osn_stream_t wstream = forked ? run->side_stream : stream;
    void* wws = forked ? run->ws_side : run->ws;
    const size_t wws_bytes = size_t(forked ? run->ws_side_bytes : run->ws_bytes);
    for (int i = run->end_op - 1; i >= run->first_op; --i) {
        const osn_net_op& o = net->ops[i];
        const int64_t n_in = rows[o.lvl_in], n_out = rows[o.lvl_out];
        const osn_net_weight& w = run->weights[o.weight];
        // ---- the gradient arriving at this stage's output: every consumer of dst (and of its copy inside a cat buffer)
        const float* gsrc[4];
        int64_t gld[4];
        int ng = 0;
        if (o.dst < 0) {
            // (the dense gradient may be absent when the caller hands over only the rows the loss saw -- round 6: the head itself ran
            //  on those rows, there is no [N, cout] output -- ; the row-compacted path below then has to apply)
            OSN_REQUIRE(run->goutput || (run->goutput_rows && run->grows_pos && run->grows_idx && run->n_grows > 0), OSN_E_ARG,
                        "osn_net_backward: null output gradient");
            gsrc[ng] = run->goutput; gld[ng] = o.cout; ++ng;
        } else {
            // (every later op, not only those of this call's range: a backward pass may be played in segments, highest ops
            // first -- executor.py does when a gradient exchange wants the finished slices early -- and the consumers of an
            // earlier segment left their input gradients in the arena)
            for (int j = i + 1; j < net->n_ops && ng < 4; ++j) {
                const osn_net_op& c = net->ops[j];
                if (c.src == o.dst && c.need_dgrad) { gsrc[ng] = reinterpret_cast<const float*>(B + L.gin_off[j]); gld[ng] = c.cin; ++ng; }
                if (ng < 4 && c.res == o.dst) { gsrc[ng] = reinterpret_cast<const float*>(B + L.gres_off[j]); gld[ng] = c.cout; ++ng; }
                if (ng < 4 && o.copy_buf >= 0 && c.src == o.copy_buf && c.need_dgrad) {
                    gsrc[ng] = reinterpret_cast<const float*>(B + L.gin_off[j]) + o.copy_col; gld[ng] = c.cin; ++ng;
                }
            }
            OSN_REQUIRE(ng >= 1 && ng <= 3, OSN_E_ARG, "osn_net_backward: the output of op %d has %s consumers inside the executed range (1 .. 3 supported)",
                        i, ng == 0 ? "no" : "more than three");
        }
        // Round 4: a shortcut stage's batch-norm backward and input gradient stay on the MAIN stream.  Queued on the side stream they
        // sat behind its backlog of weight gradients, and the main stream stalled 24 - 112 us at each of the pass's joins waiting for
        // them (tools/gap_census.py; -0.09 ms per step, profiles/r04_s15_to_s19_stream_queueing_ab.txt).  Only the weight gradients
        // fork.  (In the FORWARD pass the side stream has no backlog and the shortcut stages do run there: +0.04 ms with them on the
        // main stream.)
        const osn_net_run* r = run;
        const osn_stream_t sstream = stream;
        const float* gx;
        if (o.bn >= 0) {
            const osn_net_bn& bn = run->bns[o.bn];
            const float* x = reinterpret_cast<const float*>(A + L.x_off[i]);
            const float* y = reinterpret_cast<const float*>(A + L.y_off[o.dst]);
            const float *mean = bn.running_mean, *var = bn.running_var;
            if (run->training) {
                const float* mv = reinterpret_cast<const float*>(A + L.stat_off[i]);
                mean = mv; var = mv + o.cout;
            }
            float* gxw = reinterpret_cast<float*>(B + L.gx_off[i]);
            float* gres = o.res >= 0 ? reinterpret_cast<float*>(B + L.gres_off[i]) : nullptr;
            OSN_REQUIRE(bn.ggamma && bn.gbeta, OSN_E_ARG, "osn_net_backward: op %d: null batch-norm gradient pointers", i);
            // bn -> relu without a residual: the mask is recomputed from x, y is not read (one tensor less per backward kernel)
            const bool from_x = o.relu && o.res < 0 && bn.beta;
            rc = osn_bn_backward_multi2(x, from_x ? nullptr : y, gsrc, gld, ng, mean, var, bn.gamma, from_x ? bn.beta : nullptr, bn.eps,
                                        o.relu, run->training, gxw, gres, bn.ggamma, bn.gbeta, n_out, o.cout, r->ws, size_t(r->ws_bytes),
                                        sstream);
            if (rc) return rc;
            gx = gxw;
        } else {
            OSN_REQUIRE(ng == 1 && gld[0] == o.cout, OSN_E_ARG, "osn_net_backward: op %d without a batch norm needs one contiguous gradient", i);
            gx = gsrc[0];
        }
        const float* in = o.src < 0 ? run->input : reinterpret_cast<const float*>(A + L.y_off[o.src]);
        MapView v{};
        const osn_net_map* m = o.map >= 0 ? &run->maps[o.map] : nullptr;
        if (m) v = view_of(*m, o.transposed != 0);
        // the network output's gradient with known zero rows (the loss saw n_grows of the n_out rows): both gradients of the
        // head run on the compacted rows -- in[idx]^T @ g[idx] and scatter(g[idx] @ W^T) -- instead of all n_out
        const bool sparse_rows = o.dst < 0 && o.K == 1 && o.bn < 0 && run->goutput_rows && run->grows_pos && run->grows_idx &&
                                 run->n_grows > 0 && run->n_grows < n_out && L.rows_in_off != NO_OFF &&
                                 (L.wgrad_k[i] == OSN_NET_K_WGRAD || L.wgrad_k[i] == OSN_NET_K_WGRAD_TL) &&
                                 (!o.need_dgrad || L.dgrad_k[i] == OSN_NET_K_DENSE);
        OSN_REQUIRE(o.dst >= 0 || sparse_rows || run->goutput, OSN_E_ARG,
                    "osn_net_backward: op %d: only the supervised rows of the output gradient were given, but the row-compacted head "
                    "backward does not apply to this configuration", i);
        // ---- weight gradient (fork: the side stream sees everything the main stream has queued up to gx)
        OSN_REQUIRE(w.gW, OSN_E_ARG, "osn_net_backward: op %d: null weight-gradient pointer", i);
        // the stem is the LAST weight gradient of a pass and nothing is left to hide it behind: the reduction of the pair-array
        // gradients still pending depends on none of the stem's inputs, so it is queued in FRONT of the side stream's wait for the
        // stem's output gradient -- it runs beside the main stream's last batch-norm backward instead of in the pass's tail
        if (forked && L.wgrad_k[i] == OSN_NET_K_WGRAD_STEM && !jobs.empty()) {
            rc = osn_wgrad_tl_reduce_batch(jobs.data(), int(jobs.size()), wstream);
            jobs.clear();
            if (rc) return rc;
        }
        // ... and the stem's own weight gradient runs on the MAIN stream: when the pass reaches it the main stream has nothing else
        // left, while the side stream may still be draining its backlog -- the two now overlap instead of queueing up
        // (on the side stream it queued up behind the backlog: round 4's A/B, profiles/r04_s15_to_s19_stream_queueing_ab.txt)
        const bool wg_main = forked && L.wgrad_k[i] == OSN_NET_K_WGRAD_STEM;
        if (forked && !wg_main) {
            OSN_HIP(hipEventRecord(evs->ev[i], st));
            OSN_HIP(hipStreamWaitEvent(side, evs->ev[i], 0));
        }
        {
            Bracket br(run->prof, i, 2, (forked && !wg_main) ? side : st);
            if (sparse_rows) {
                float* in_rows = reinterpret_cast<float*>(B + L.rows_in_off);
                rc = osn_rows_gather(in, run->grows_idx, run->n_grows, o.cin, in_rows, wstream);
                if (!rc)
                    rc = osn_spconv_wgrad(in_rows, run->goutput_rows, nullptr, nullptr, nullptr, w.gW, run->n_grows, 1, o.cin, o.cout, wws,
                                          wws_bytes, wstream);
            } else if (L.wgrad_k[i] == OSN_NET_K_WGRAD_TL) {
                // pair arrays of the map's forward table; a transposed conv uses the strided conv's arrays, roles swapped
                OSN_REQUIRE(o.K == 1 || (m && m->pl_fwd), OSN_E_ARG, "osn_net_backward: op %d: pair lists missing", i);
                osn_wgrad_job job;
                rc = osn_spconv_wgrad_tl_partial(in, gx, m ? m->pl_fwd : nullptr, o.transposed ? 1 : 0, w.gW, n_in, n_out, o.K, o.cin, o.cout,
                                                 B + L.gpart_off[i], osn_spconv_wgrad_tl_ws_bytes(o.K, o.cin, o.cout), &job, wstream);
                if (!rc) jobs.push_back(job);
                // reduce in batches ON the stream the partial sums were computed on (stream order is all the ordering it needs):
                // only the last, small batch is left for the tail of the pass
                if (!rc && forked && jobs.size() >= 12) {          // (6 / 24 measured: within noise, profiles/r04_s15_to_s19_*)
                    rc = osn_wgrad_tl_reduce_batch(jobs.data(), int(jobs.size()), wstream);
                    jobs.clear();
                }
            } else if (L.wgrad_k[i] == OSN_NET_K_WGRAD_STEM) {
                OSN_REQUIRE(v.nbr_f, OSN_E_ARG, "osn_net_backward: op %d: the stem weight gradient needs the plain table", i);
                rc = wg_main ? osn_stem_conv_wgrad(in, gx, v.nbr_f, w.gW, n_out, o.K, o.cin, o.cout, run->ws, size_t(run->ws_bytes), stream)
                             : osn_stem_conv_wgrad(in, gx, v.nbr_f, w.gW, n_out, o.K, o.cin, o.cout, wws, wws_bytes, wstream);
            } else {
                rc = osn_spconv_wgrad(in, gx, o.K > 1 ? v.nbr_f : nullptr, o.K > 1 && m ? m->counts : nullptr, nullptr, w.gW, n_out, o.K,
                                      o.cin, o.cout, wws, wws_bytes, wstream);
            }
        }
        if (rc) return rc;
        // ---- input gradient
        if (o.need_dgrad) {
            float* gin = reinterpret_cast<float*>(B + L.gin_off[i]);
            {
                Bracket br(run->prof, i, 1, st);
                // a self map (flip) is its own mirror: same direction of the pair arrays with the mirrored weight image;
                // otherwise the input gradient walks them the other way round
                const int swap_f = o.transposed ? 1 : 0;
                if (sparse_rows) {
                    float* gin_rows = reinterpret_cast<float*>(B + L.rows_gin_off);
                    rc = osn_dense_fwd(run->goutput_rows, w.tl_dgrad, gin_rows, run->n_grows, o.cout, o.cin, sstream);
                    if (!rc) rc = osn_rows_scatter_zero(gin_rows, run->grows_pos, n_in, o.cin, gin, sstream);
                } else
                rc = run_conv(L.dgrad_k[i], gx, n_out, gin, n_in, o.K, o.cout, o.cin, nullptr, w.x6_dgrad, w.tl_dgrad, v.nbr_b, v.tb_rows,
                              v.tb_tbl, v.tb_g, v.tl_b, v.tl_b_rows, v.tl_b_bm, m ? m->pl_fwd : nullptr, o.transposed ? n_in : n_out,
                              (m && m->flip) ? swap_f : 1 - swap_f, r, sstream, i);
            }
            if (rc) return rc;
        }
    }
    // the pair-array weight gradients still waiting for their reduction: one launch (instead of one per convolution)
    rc = osn_wgrad_tl_reduce_batch(jobs.data(), int(jobs.size()), wstream);
    if (rc) return rc;
    // join: everything after the pass sees the side stream's work.  (Not after an inner segment of a segmented pass when the
    // caller says so: the main stream would stall three more times per pass on weight gradients nothing on it reads --
    // 0.27 ms per step, profiles/r04_s9_dist_readiness.txt; the exchange of those gradients queues behind the side stream.)
    if (forked && !(run->flags & OSN_NET_RUN_NO_JOIN)) {
        OSN_HIP(hipEventRecord(evs->ev[net->n_ops], side));
        OSN_HIP(hipStreamWaitEvent(st, evs->ev[net->n_ops], 0));
    }
    return OSN_OK;
}

extern "C" osn_events_t* osn_events_create(int n) {
    if (n < 1) return nullptr;
    Events* e = new (std::nothrow) Events();
    if (!e) return nullptr;
    e->ev.resize(size_t(n), nullptr);
    for (int i = 0; i < n; ++i) {
        if (hipEventCreateWithFlags(&e->ev[i], hipEventDisableTiming) != hipSuccess) {
            for (int j = 0; j < i; ++j) (void)hipEventDestroy(e->ev[j]);
            delete e;
            return nullptr;
        }
    }
    return reinterpret_cast<osn_events_t*>(e);
}

extern "C" void osn_events_destroy(osn_events_t* h) {
    Events* e = reinterpret_cast<Events*>(h);
    if (!e) return;
    for (hipEvent_t x : e->ev) (void)hipEventDestroy(x);
    delete e;
}

// ------------------------------------------------------------------------------------------- launch timer
extern "C" osn_prof_t* osn_prof_create(int capacity) {
    if (capacity < 1) return nullptr;
    Prof* p = new (std::nothrow) Prof();
    if (!p) return nullptr;
    p->ev.resize(size_t(2) * capacity, nullptr);
    p->tag.assign(capacity, 0);
    for (int i = 0; i < 2 * capacity; ++i) {
        if (hipEventCreate(&p->ev[i]) != hipSuccess) {
            for (int j = 0; j < i; ++j) (void)hipEventDestroy(p->ev[j]);
            delete p;
            return nullptr;
        }
    }
    p->cap = capacity;
    return reinterpret_cast<osn_prof_t*>(p);
}

extern "C" void osn_prof_destroy(osn_prof_t* h) {
    Prof* p = reinterpret_cast<Prof*>(h);
    if (!p) return;
    for (hipEvent_t e : p->ev) (void)hipEventDestroy(e);
    delete p;
}

extern "C" int osn_prof_filter(osn_prof_t* h, const int32_t* tags, int n_tags) {
    Prof* p = reinterpret_cast<Prof*>(h);
    OSN_REQUIRE(p && n_tags >= 0 && (n_tags == 0 || tags), OSN_E_ARG, "osn_prof_filter: bad arguments");
    p->only.assign(tags, tags + n_tags);
    return OSN_OK;
}

extern "C" int osn_prof_read(osn_prof_t* h, int32_t* tags, float* ms, int capacity, int reset) {
    Prof* p = reinterpret_cast<Prof*>(h);
    if (!p || !tags || !ms) return 0;
    int n = p->n < capacity ? p->n : capacity;
    for (int i = 0; i < n; ++i) {
        float t = 0.f;
        if (hipEventSynchronize(p->ev[2 * i + 1]) != hipSuccess || hipEventElapsedTime(&t, p->ev[2 * i], p->ev[2 * i + 1]) != hipSuccess)
            t = -1.f;
        tags[i] = p->tag[i];
        ms[i] = t;
    }
    if (reset) p->n = 0;
    return n;
}
