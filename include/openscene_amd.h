/* openscene_amd.h -- C ABI of libopenscene_amd.so (MI355X / gfx950 only).
 *
 * Drop-in boundary for ONE hot path of pengsongyou/openscene: hash voxelisation,
 * coordinate / kernel maps, sparse 3-D convolution (fwd, dgrad, wgrad), batch-norm
 * (+ReLU +residual), and the per-point feature x CLIP-text query.  Every entry
 * point names the reference interface it replaces (paths relative to the
 * reference tree; [ME] = NVIDIA/MinkowskiEngine v0.5.4, the un-vendored
 * dependency that holds the reference's arithmetic -- SURVEY.md appendix C).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (torch), unless the
 *     parameter is documented as "host";
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*),
 *     except the few that return a count to the host (documented);
 *   - return value: 0 = OK, negative = error (OSN_E_*); text via osn_last_error()
 *     (thread-local); no C++ exception crosses the boundary;
 *   - the library allocates nothing: scratch comes in through `ws` arguments
 *     whose size the matching *_ws_bytes() helper reports;
 *   - callable from any host thread (autograd's backward thread included); no
 *     global mutable state.
 *   - feature matrices are row-major float32 [rows, channels]; coordinates are
 *     int32 rows (batch, x, y, z) exactly as the reference builds them
 *     (dataset/feature_loader.py:178-179,204-205).
 *   - a kernel map is a dense k-major neighbour table nbr[K, n_out] of int32,
 *     nbr[k, o] = input row found at coord[o] + offset_k, or -1.
 */
#ifndef OPENSCENE_AMD_H
#define OPENSCENE_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OSN_OK 0
#define OSN_E_ARG (-1)     /* bad argument (null pointer, size, alignment, range) */
#define OSN_E_HIP (-2)     /* a HIP runtime call or kernel launch failed          */
#define OSN_E_WS (-3)      /* workspace too small                                  */
#define OSN_E_RANGE (-4)   /* coordinate outside the packable range                */

typedef void* osn_stream_t; /* hipStream_t */

/* ---- library ------------------------------------------------------------- */
int osn_version(void);                 /* ABI version, bumped on any signature change */
const char* osn_last_error(void);      /* message of this thread's last failing call   */
int osn_device_ok(void);               /* 1 if the current HIP device is gfx950        */

/* ---- coordinate maps ----------------------------------------------------- *
 * Replaces [ME] CoordinateManager.insert_and_map / stride (called implicitly by
 *   SparseTensor(features, coordinates)   run/distill.py:316-317, run/evaluate.py:284
 *   stride-2 convolutions                 models/mink_unet.py:52-53,59-60,66-67,73-74).
 * The hash table is open addressing over a 64-bit packed key
 * (b<<48 | (x+2^15)<<32 | (y+2^15)<<16 | (z+2^15)); capacity = power of two.    */
int64_t osn_hash_capacity(int64_t n);                                  /* host helper */
size_t osn_coords_unique_ws_bytes(int64_t n);                          /* host helper */

/* Quantise rows to multiples of `stride` (floor; stride 1 = as is), de-duplicate.
 * out_coords4[u] = u-th distinct row in FIRST-OCCURRENCE order, inverse[i] = u,
 * first[u] = lowest input row of u.  On return the table (keys, vals) maps a packed
 * key to u and is what osn_kmap_build() probes.  *n_unique_host is written on the
 * HOST; the call synchronises `stream` once to deliver it.                      */
int osn_coords_unique(const int32_t* coords4, int64_t n, int stride,
                      uint64_t* table_keys, int32_t* table_vals, int64_t cap,
                      int32_t* out_coords4, int32_t* inverse, int32_t* first,
                      int64_t* n_unique_host, void* ws, size_t ws_bytes, osn_stream_t stream);

/* The same unique pipeline WITHOUT the host synchronisation, for chaining the levels of a coordinate pyramid
 * (models/mink_unet.py:52-74: four stride-2 levels are created per forward): everything is sized for n_max rows;
 * n_dev (nullable) = the actual row count in device memory (the previous level's count_dev), count_dev receives
 * this level's number of unique rows, err_dev is a sticky flag (zeroed by the caller) set on a coordinate outside
 * the packable range.  The caller reads all counts and the flag back ONCE.  Identical results to
 * osn_coords_unique on the first `count` rows.                                                                */
int osn_coords_unique_async(const int32_t* coords4, int64_t n_max, const int32_t* n_dev, int stride,
                            uint64_t* table_keys, int32_t* table_vals, int64_t cap, int32_t* out_coords4,
                            int32_t* inverse, int32_t* first, int32_t* count_dev, int32_t* err_dev,
                            void* ws, size_t ws_bytes, osn_stream_t stream);

/* The whole coordinate pyramid of a scene from ONE call (models/mink_unet.py:52-74: conv1p1s2 ... conv4p8s2 create four stride-2
 * levels below the input's): level i = the unique rows of level i - 1 quantised to strides[i], queued back to back with the row
 * counts in device memory -- osn_coords_unique_async per level, with one preset launch for all tables and counters and the flag pass
 * folded into the scan (31 instead of 45 dependent launches for five levels).  Every per-level array is sized for n0 rows / `cap`
 * slots; counts_dev[0 .. n_levels) receive the unique rows per level, counts_dev[n_levels] the sticky range-error flag (all zeroed
 * here); the caller reads them back ONCE.  Pointer arrays are HOST arrays of device pointers.  Results: bit for bit the per-level calls'. */
int osn_coords_pyramid_async(const int32_t* coords4, int64_t n0, const int32_t* strides, int n_levels,
                             uint64_t* const* table_keys, int32_t* const* table_vals, int64_t cap,
                             int32_t* const* out_coords4, int32_t* const* inverse, int32_t* const* first,
                             int32_t* counts_dev, void* ws, size_t ws_bytes, osn_stream_t stream);

/* Replaces [ME] CoordinateManager.kernel_map (HYPER_CUBE region).  Offsets
 * enumerate with x fastest; odd ksize centred, even ksize spans [0,ksize); all
 * multiplied by `offset_scale` (= dilation * tensor stride of the INPUT map).  */
int osn_kmap_build(const uint64_t* in_table_keys, const int32_t* in_table_vals, int64_t cap,
                   const int32_t* out_coords4, int64_t n_out, int ksize, int offset_scale,
                   int32_t* nbr, int64_t* counts /* nullable: int64 [K] pairs per offset */,
                   osn_stream_t stream);

/* The same map when the output coordinates ARE the table's own rows in row order (every stride-1
 * convolution of the path: models/mink_unet.py:47-113 BasicBlock convs, conv0p1s1) and ksize is odd:
 * such a map is its own mirror (nbr[k][o] = i <=> nbr[K-1-k][i] = o), so only the offsets below the centre
 * are probed and each hit also writes its mirrored entry.  Bit-identical output to osn_kmap_build. */
int osn_kmap_build_self(const uint64_t* table_keys, const int32_t* table_vals, int64_t cap,
                        const int32_t* coords4, int64_t n, int ksize, int offset_scale,
                        int32_t* nbr, int64_t* counts /* nullable */, osn_stream_t stream);

/* tbl[k, i] = o  <=>  nbr[k, o] = i : the map of the transposed operator
 * ([ME] MinkowskiConvolutionTranspose, models/mink_unet.py:77-78,84-85,91-92,98-99,
 * and the input-gradient of every strided convolution).                          */
int osn_kmap_transpose(const int32_t* nbr, int64_t n_out, int K, int64_t n_in, int32_t* tbl,
                       osn_stream_t stream);

/* Tile-order optimisation (no reference counterpart; ME's GPU kernels are unordered):
 * order[j] = output rows sorted (stably) by their offset-occupancy bit mask, and
 * nbr_sorted[k, j] = nbr[k, order[j]].  Feed both to osn_spconv_fwd (nbr_sorted as the
 * table, order as out_rows): rows of one tile then share their offsets and the per-tile
 * offset skip removes the empty work; results are unchanged.  K <= 32.
 * counts (nullable, int64 [K] = osn_kmap_count): orders the key bits by rarity (rarest offset = most
 * significant bit) instead of offset index, so the tiles that pay for a rare offset are few.        */
size_t osn_kmap_sort_ws_bytes(int64_t n_out);
int osn_kmap_sort(const int32_t* nbr, int64_t n_out, int K, const int64_t* counts, int32_t* order,
                  int32_t* nbr_sorted,
                  uint32_t* gmask /* nullable: uint32 [ceil(n_out/32)], OR of the occupancy masks of
                                     each 32-row group of the sorted table */,
                  void* ws, size_t ws_bytes, osn_stream_t stream);

/* counts[k] = #valid entries of nbr[k, :]  (int64 [K], device).                  */
int osn_kmap_count(const int32_t* nbr, int64_t n_out, int K, int64_t* counts, osn_stream_t stream);

/* ---- sparse convolution -------------------------------------------------- *
 * Replaces [ME] MinkowskiConvolution / MinkowskiConvolutionTranspose forward and
 * backward (constructed at models/mink_unet.py:47-113, models/resnet_base.py:92-97).
 *   out[o, :] = sum_k in[nbr[k, o], :] @ W[k]        W: [K, cin, cout] float32
 * fp32 in / fp32 accumulate on v_mfma_f32_32x32x2_f32 (bit-for-bit an fmaf chain).
 * out_rows (nullable): out row of tile slot j is out_rows[j] instead of j.       */
size_t osn_spconv_fwd_ws_bytes(int64_t n_out, int K, int cin, int cout);
int osn_spconv_fwd(const float* in, const float* W, const int32_t* nbr, const int32_t* out_rows,
                   const uint32_t* gmask /* nullable: osn_kmap_sort's group masks of `nbr`; lets the
                       kernel split each tile's ACTIVE offsets into equal work units (load balance) */,
                   float* out, int64_t n_out, int K, int cin, int cout,
                   void* ws, size_t ws_bytes, osn_stream_t stream);

/* Split-bf16 ("bf16x6") variant of osn_spconv_fwd: same result contract (fp32 in / fp32 out, fp32-level
 * accuracy: every operand is split into three bf16 pieces and the six significant cross products are
 * accumulated in fp32 on v_mfma_f32_32x32x16_bf16, 2.7x the fp32 MFMA throughput).  Wp comes from
 * osn_weight_prep_x6: bf16 [3][K][n_out_channels][c_padded]; for_dgrad = 1 prepares the (optionally
 * mirrored) transposed weights of the input gradient, replacing osn_weight_transpose.
 * Here `cin` = contraction channels of `in`, `cout` = output channels.  Needs cin % 4 == 0.          */
size_t osn_weight_prep_x6_bytes(int K, int cin, int cout, int for_dgrad);
int osn_weight_prep_x6(const float* W, int K, int cin, int cout, int flip, int for_dgrad, void* Wp,
                       osn_stream_t stream);
/* Both layouts of one weight in ONE launch (training: the forward planes now, the input-gradient planes
 * kept for the backward pass): Wp_fwd = prep(flip 0, for_dgrad 0), Wp_dgrad = prep(flip, for_dgrad 1). */
int osn_weight_prep_x6_pair(const float* W, int K, int cin, int cout, int flip, void* Wp_fwd, void* Wp_dgrad,
                            osn_stream_t stream);
int osn_spconv_fwd_x6(const float* in, const void* Wp, const int32_t* nbr, const int32_t* out_rows,
                      const uint32_t* gmask, float* out, int64_t n_out, int K, int cin, int cout,
                      void* ws, size_t ws_bytes, osn_stream_t stream);

/* ---- second-generation convolution: per-tile compacted pair lists ------------------------- *
 * Same result contract as osn_spconv_fwd / osn_spconv_fwd_x6 ([ME] MinkowskiConvolution[Transpose]
 * forward, models/mink_unet.py:47-113; fp32 in / fp32 out, split-bf16 arithmetic of fp32-class accuracy,
 * bitwise reproducible), other data layout: the kernel map of every tile of `bm` consecutive rows of a
 * neighbour table is stored per offset as a dense list of (input row, local output row) pairs -- what
 * [ME] calls the kernel map's in/out index lists, cut per tile -- so the matrix units only see real
 * pairs, the weights of an offset stay in registers for a whole (tile, offset), the output tile is
 * accumulated in LDS, and persistent workgroups draw the tiles from an atomic counter (densest first).
 *   osn_tile_rows(n_out)            rows per tile for a table of n_out rows (host helper, 32 .. 88)
 *   osn_tile_lists_build(nbr, ..)   tl = { int32 cnt[n_tiles][K] (256-byte aligned), int2 lst[n_tiles][K][bm] }
 *                                   from a (possibly osn_kmap_sort-ordered) table; lists keep row order
 *   osn_weight_prep_tl(W, ..)       MFMA-fragment images of a weight: Wp_fwd for the forward, Wp_dgrad
 *                                   (transposed, mirrored in k if flip) for the input gradient; either
 *                                   may be null; sizes from osn_weight_prep_tl_bytes
 *   osn_spconv_fwd_tl(..)           out[out_rows ? out_rows[j] : j] = sum_k in[list rows] @ W[k] for table
 *                                   row j; tl = null <=> K == 1 identity map.  bn_partial (nullable):
 *                                   double [n_tiles][2][cout] per-tile column sums / sums of squares of
 *                                   `out` (the following batch norm's statistics without a pass over out).
 *                                   ws: osn_spconv_fwd_tl_ws_bytes(..) bytes: tile counters (512 bytes, zeroed by the call) and, on
 *                                   small tables, the partial tiles of a launch that splits each tile's offsets over
 *                                   several workgroups (summed in a fixed order; bn_partial must be null there).
 *                                   Needs cin % 4 == 0, cout % 4 == 0 and n_in <= 2^24 (OSN_E_RANGE otherwise).  */
int osn_tile_rows(int64_t n_out);
size_t osn_tile_lists_bytes(int64_t n_out, int K, int bm);
int osn_tile_lists_build(const int32_t* nbr, int64_t n_out, int K, int bm, void* tl, osn_stream_t stream);
size_t osn_weight_prep_tl_bytes(int K, int cin, int cout, int for_dgrad);
int osn_weight_prep_tl(const float* W, int K, int cin, int cout, int flip, void* Wp_fwd, void* Wp_dgrad,
                       osn_stream_t stream);
size_t osn_spconv_fwd_tl_ws_bytes(int64_t n_out, int K, int cout, int bm);
/* osn_spconv_fwd_tl_pc: the same call with caller-owned PERSISTENT tile counters (128 int32 in device memory, zero before
 * the first call, one buffer per stream): the kernel's last workgroup puts them back to zero, so the call needs no memset
 * launch.  A launch that fails leaves them undefined (zero them again).                                                  */
int osn_spconv_fwd_tl(const float* in, int64_t n_in, const void* Wp, const void* tl, const int32_t* out_rows,
                      float* out, double* bn_partial, int64_t n_out, int K, int cin, int cout, int bm,
                      void* ws, size_t ws_bytes, osn_stream_t stream);
int osn_spconv_fwd_tl_pc(const float* in, int64_t n_in, const void* Wp, const void* tl, const int32_t* out_rows,
                         float* out, double* bn_partial, int64_t n_out, int K, int cin, int cout, int bm,
                         void* ws, size_t ws_bytes, int32_t* counters, osn_stream_t stream);

/* ---- every weight image of a model in one launch ----------------------------------------------- *
 * [ME] keeps one `kernel` parameter per MinkowskiConvolution / MinkowskiConvolutionTranspose
 * (models/mink_unet.py:47-113, models/resnet_base.py:38-60); an optimizer step changes all of them at once.
 * Instead of one osn_weight_prep_x6 / osn_weight_prep_tl launch per convolution and step, the host keeps a
 * table of jobs in DEVICE memory (built once per model: pointers and shapes do not change) and one launch
 * fills every image.  jobs[i].first_block = sum of osn_weight_prep_job_blocks(...) of the jobs before i,
 * total_blocks = the sum over all jobs.  Image contents are identical to the per-weight entry points
 * (layout OSN_PREP_X6: osn_weight_prep_x6(W, .., flip, for_dgrad); OSN_PREP_TL: the Wp_fwd (for_dgrad = 0)
 * or Wp_dgrad (for_dgrad = 1) of osn_weight_prep_tl).                                                      */
enum { OSN_PREP_X6 = 0, OSN_PREP_TL = 1 };
typedef struct osn_prep_job {
    const float* W;       /* [K][cin][cout] fp32 weight                                   */
    void* out;            /* image: osn_weight_prep_x6_bytes / osn_weight_prep_tl_bytes    */
    int64_t first_block;  /* exclusive prefix of the jobs' block counts                    */
    int32_t K, cin, cout; /* weight shape                                                  */
    int32_t flip;         /* mirror the offsets (input gradient of stride-1 odd kernels)   */
    int32_t for_dgrad;    /* 1: image of the transposed weight                             */
    int32_t layout;       /* OSN_PREP_X6 | OSN_PREP_TL                                     */
} osn_prep_job;           /* 48 bytes */
int64_t osn_weight_prep_job_blocks(int K, int cin, int cout, int for_dgrad, int layout);
int osn_weight_prep_batch(const osn_prep_job* jobs_dev, int n_jobs, int64_t total_blocks, osn_stream_t stream);

/* Which kernel instance / launch shape osn_spconv_fwd() uses for a problem (host helper, for
 * profiling): plan6 = {WM, WN, TN, BK, S (offset splits), workgroups};
 * the kernel symbol is spconv_fwd_kernel<WM, WN, TN, BK>.                          */
int osn_spconv_fwd_plan(int64_t n_out, int K, int cin, int cout, int32_t* plan6);

/* Wt[k] = W[flip ? K-1-k : k]^T   ([K, cout, cin]).  The input gradient is
 * osn_spconv_fwd(gout, Wt, table, ...) with table = nbr and flip = 1 for a
 * stride-1 odd kernel (the map is its own mirror), or the transposed table and
 * flip = 0 otherwise.                                                             */
int osn_weight_transpose(const float* W, int K, int cin, int cout, int flip, float* Wt,
                         osn_stream_t stream);

/* gW[k] = sum_{o : nbr[k,o] >= 0} in[nbr[k,o], :]^T (x) gout[o, :]   ([K, cin, cout]).
 * counts (nullable, device int64 [K], = osn_kmap_count of nbr): pair count per offset, used
 * to split each offset's rows into work items of equal pair count on the device (the centre
 * offset of a 3^3 map holds 19 % of the pairs, a corner offset < 1 %).  Deterministic: the
 * split is a pure function of (counts, sizes); partial sums are reduced in item order.
 * plan_items (nullable): the work-item table osn_spconv_wgrad_plan() wrote for the same
 * (counts, n_out, K) and a (cin, cout) with the same osn_spconv_wgrad_items_bytes -- the table
 * depends on the map only, so the convs of one map share it and skip the per-call plan launch. */
size_t osn_spconv_wgrad_ws_bytes(int64_t n_out, int K, int cin, int cout);
size_t osn_spconv_wgrad_items_bytes(int64_t n_out, int K, int cin, int cout);
int osn_spconv_wgrad_plan(const int64_t* counts, int64_t n_out, int K, int cin, int cout,
                          int32_t* items /* osn_spconv_wgrad_items_bytes */, osn_stream_t stream);
int osn_spconv_wgrad(const float* in, const float* gout, const int32_t* nbr, const int64_t* counts,
                     const int32_t* plan_items, float* gW, int64_t n_out, int K, int cin, int cout,
                     void* ws, size_t ws_bytes, osn_stream_t stream);

/* Second-generation weight gradient: the pairs of every offset compacted ONCE per map into arrays
 * (input row, output row) in tile order -- [ME]'s kernel-map in/out index lists -- shared by every
 * convolution and every step of the map's life; split-bf16 arithmetic on the bf16 MFMA with LDS transpose
 * reads (fp32-class accuracy, bitwise reproducible: work items are a pure function of the map, partial sums
 * are reduced in item order).
 *   osn_pair_lists_build(tl, ..)    pl = per-offset pair arrays + device-planned work items, from the tile
 *                                   lists `tl` of a table with n_out rows (bm as given to osn_tile_lists_build;
 *                                   out_rows = the table's row permutation or null); osn_pair_lists_bytes
 *   osn_spconv_wgrad_tl(..)         gW[k] = sum over the pairs of offset k of in[i]^T (x) gout[o]  ([K, cin, cout]).
 *                                   swap = 1: the arrays belong to the strided convolution this TRANSPOSED
 *                                   convolution mirrors (in rows are indexed by the arrays' output rows).
 *                                   pl = null <=> K == 1 identity map.  Needs cin % 4 == 0 and cout % 4 == 0.   */
size_t osn_pair_lists_bytes(int64_t n_out, int K, int bm);
int osn_pair_lists_build(const void* tl, const int32_t* out_rows, int64_t n_out, int K, int bm, void* pl,
                         osn_stream_t stream);
size_t osn_spconv_wgrad_tl_ws_bytes(int K, int cin, int cout);
int osn_spconv_wgrad_tl(const float* in, const float* gout, const void* pl, int swap, float* gW, int64_t n_in,
                        int64_t n_out, int K, int cin, int cout, void* ws, size_t ws_bytes, osn_stream_t stream);

/* The two halves of osn_spconv_wgrad_tl for callers that run many weight gradients back to back (the network executor):
 * osn_spconv_wgrad_tl_partial launches the kernel only -- partial sums per work item into `partial`
 * (osn_spconv_wgrad_tl_ws_bytes bytes, must stay untouched until the reduction ran) -- and fills `job` (HOST);
 * osn_wgrad_tl_reduce_batch reduces any number of such jobs in ONE launch per 32 jobs (same summation order as
 * osn_spconv_wgrad_tl: bitwise the same gradient).  A job whose gW is null (empty map: gW was zeroed) is skipped.  */
typedef struct osn_wgrad_job {
    const float* partial;        /* device */
    const void* range;           /* device: items of each offset, or null (identity map: items [0, ident_items))     */
    float* gW;                   /* device, [K, cin, cout]                                                         */
    int32_t K, cin, cout, ident_items;
} osn_wgrad_job;                 /* 40 bytes */
int osn_spconv_wgrad_tl_partial(const float* in, const float* gout, const void* pl, int swap, float* gW, int64_t n_in,
                                int64_t n_out, int K, int cin, int cout, void* partial, size_t partial_bytes,
                                osn_wgrad_job* job_host, osn_stream_t stream);
int osn_wgrad_tl_reduce_batch(const osn_wgrad_job* jobs_host, int n_jobs, osn_stream_t stream);

/* 1x1 convolution = row-wise matrix product ([ME] MinkowskiConvolution(kernel_size=1): the head `final`,
 * models/mink_unet.py:108-113, and the BasicBlock shortcuts, models/resnet_base.py:101-107; with the input-gradient image
 * their backward):  out[n, cout] = in[n, cin] @ B,  B given as an osn_weight_prep_tl image with K = 1 (forward image:
 * B = W; input-gradient image: B = W^T, `cin` then counts the conv's OUTPUT channels).  Same split-bf16 arithmetic as
 * osn_spconv_fwd_tl; no workspace.  Needs cin % 4 == 0, cin >= 8, cout % 4 == 0.                                    */
int osn_dense_fwd(const float* in, const void* Wp, float* out, int64_t n, int cin, int cout, osn_stream_t stream);

/* Convolution of a NARROW layer (32 or 64 channels on both sides) straight from the neighbour table ([ME]
 * MinkowskiConvolution forward, models/mink_unet.py:59-93: conv1p1s2 .. block2 -- the 32- and 64-channel stages of the encoder --
 * and, with the input-gradient image and the mirrored table, their backward): every lane of a wave gathers the 32 bytes of the
 * fp32 row that ARE its MFMA operand (two 16-byte loads through nbr; an absent neighbour reads zeros and moves no data), splits
 * them in registers and multiplies -- no staging through LDS, no barrier in the loop.  A workgroup owns 64 table rows, its four
 * waves take the offsets w, w + 4, ..., partial tiles are summed in wave order (bitwise reproducible).  Same arithmetic and
 * weight image as osn_spconv_fwd_tl (osn_weight_prep_tl).  nbr: int32 [K, n_out], plain or osn_kmap_sort-ordered (then out_rows =
 * its permutation: table row j is written to out[out_rows[j]]).  osn_spconv_fwd_rg_ok: 1 for the shapes it takes (K > 1, cin and
 * cout in {32, 64}, n_in < 2^24, the feature matrix below 2 GB).                                                           */
int osn_spconv_fwd_rg_ok(int64_t n_in, int K, int cin, int cout);
int osn_spconv_fwd_rg(const float* in, int64_t n_in, const void* Wp, const int32_t* nbr, const int32_t* out_rows, float* out,
                      int64_t n_out, int K, int cin, int cout, osn_stream_t stream);

/* Convolution of a SMALL map (<= 16 k rows) from its per-offset pair arrays, weight-stationary ([ME]
 * MinkowskiConvolution[Transpose] forward, models/mink_unet.py:51-113 on the 1/4 .. 1/16 levels; with the input-gradient
 * weight image also their backward):  a workgroup keeps W[k] of ONE offset in registers and multiplies a chunk of that
 * offset's pairs, writes the result rows to partial[k][dst row]; a second launch sums out[r] over the offsets k the
 * destination table has at r, ascending (fixed order, bitwise reproducible).  Same arithmetic and weight images as
 * osn_spconv_fwd_tl (osn_weight_prep_tl: forward image, or the input-gradient image for a backward launch).
 *   pl / pl_rows  pair arrays (osn_pair_lists_build) of a map with pl_rows output rows;
 *   swap = 0      gathers pin -> writes pout: the map's own direction (pl_rows == n_dst);
 *   swap = 1      gathers pout -> writes pin: the transposed direction (pl_rows == n_in) -- input gradient of a strided
 *                 convolution, forward of the transposed convolution that mirrors it;
 *   nbr_dst       int32 [K, n_dst] neighbour table of the DESTINATION side in plain row order (>= 0 <=> partial[k][r] was
 *                 written): nbr_fwd for swap = 0, the transposed table (osn_kmap_transpose) for swap = 1.
 *   direct = 1    the caller guarantees that every destination row has EXACTLY ONE pair in the whole map -- the fine side
 *                 of a 2^3 stride-2 map (a voxel has one parent cell): forward of [ME] MinkowskiConvolutionTranspose
 *                 (kernel 2, stride 2) and input gradient of the strided convolution.  The result rows then go straight to
 *                 `out`: no partial rows, no second launch, any map size; nbr_dst may be null.
 * ws: osn_spconv_fwd_ws_ws_bytes(n_dst, K, cout, direct) bytes (the partial rows).  Needs 2 <= K <= 128,
 * cin % 4 == cout % 4 == 0. */
size_t osn_spconv_fwd_ws_ws_bytes(int64_t n_dst, int K, int cout, int direct);
int osn_spconv_fwd_ws(const float* in, int64_t n_in, const void* Wp, const void* pl, int64_t pl_rows, int swap, int direct,
                      const int32_t* nbr_dst, float* out, int64_t n_dst, int K, int cin, int cout, void* ws,
                      size_t ws_bytes, osn_stream_t stream);

/* The 3-channel stem convolution (conv0p1s1: 5^3 kernel, 3 -> 32, models/mink_unet.py:47-50): same contract as
 * osn_spconv_fwd on the plain (unordered) table, for cin <= 4 and cout == 32.  Below 32 768 output rows: exact fp32 FMA
 * chains in ascending offset order.  From 32 768 rows on (round 6): a matrix product on the MFMA units -- two table entries
 * x four padded channels per lane, the library's split-bf16 arithmetic (fp32-class: three bf16 pieces per operand, six
 * products, fp32 accumulate), partial tiles of four waves summed in wave order.  Bitwise reproducible either way.           */
int osn_stem_conv_fwd(const float* in, const float* W, const int32_t* nbr, float* out, int64_t n_out, int K,
                      int cin, int cout, osn_stream_t stream);
/* Its weight gradient gW[k][ci][n] = sum_o in[nbr[k][o]][ci] * gout[o][n] ([ME] MinkowskiConvolutionFunction backward for
 * the same layer): dense over the table (an absent neighbour is a zero row), rows in ascending order, per-workgroup partial
 * sums added in a fixed order -- bitwise reproducible.  Below 32 768 rows fp32 fmaf chains; from 32 768 rows on the
 * contraction over the rows runs on the MFMA units (split-bf16, 32 rows per step, one accumulator set per input channel,
 * 512 partial gradients).  ws: osn_stem_conv_wgrad_ws_bytes.                                                                */
size_t osn_stem_conv_wgrad_ws_bytes(int K, int cin);
int osn_stem_conv_wgrad(const float* in, const float* gout, const int32_t* nbr, float* gW, int64_t n_out, int K, int cin,
                        int cout, void* ws, size_t ws_bytes, osn_stream_t stream);

/* ---- batch norm (+ReLU, +residual) -------------------------------------- *
 * Replaces [ME] MinkowskiBatchNorm (= torch.nn.BatchNorm1d on .F), MinkowskiReLU
 * and the BasicBlock residual add (models/mink_unet.py:50-114, resnet_base.py:98).
 * Training statistics: biased variance for normalisation, unbiased for the
 * running estimate, momentum as torch.                                            */
size_t osn_bn_ws_bytes(int64_t n, int c);
/* mean[c], var[c] (biased) of x; if running_mean/var non-null update them in place:
 * r = (1-momentum) r + momentum * stat  (var unbiased).                           */
int osn_bn_stats(const float* x, int64_t n, int c, float* mean, float* var,
                 float* running_mean, float* running_var, float momentum,
                 void* ws, size_t ws_bytes, osn_stream_t stream);
/* y = act( (x - mean) * rsqrt(var + eps) * gamma + beta  [+ residual] ),  act = ReLU if relu */
int osn_bn_apply(const float* x, const float* mean, const float* var, const float* gamma,
                 const float* beta, float eps, const float* residual, int relu, float* y,
                 int64_t n, int c, osn_stream_t stream);
/* osn_bn_stats + osn_bn_apply in one call (training-mode forward of MinkowskiBatchNorm [+ residual] [+ MinkowskiReLU],
 * models/mink_unet.py:116-174; mean / var are outputs kept for the backward).                                      */
int osn_bn_forward_train(const float* x, int64_t n, int c, const float* gamma, const float* beta, float eps,
                         const float* residual, int relu, float momentum, float* mean, float* var,
                         float* running_mean, float* running_var, float* y, void* ws, size_t ws_bytes,
                         osn_stream_t stream);
/* Backward of osn_bn_apply.  g = relu ? gy * (y > 0) : gy.
 * training != 0 (batch statistics were used):
 *   gx = gamma * invstd * (g - mean_rows(g) - xhat * mean_rows(g * xhat))
 * training == 0 (running statistics): gx = gamma * invstd * g.
 * ggamma = sum_rows(g * xhat), gbeta = sum_rows(g); gres (nullable) = g.          */
int osn_bn_backward(const float* x, const float* y, const float* gy, const float* mean,
                    const float* var, const float* gamma, float eps, int relu, int training,
                    float* gx, float* gres, float* ggamma, float* gbeta,
                    int64_t n, int c, void* ws, size_t ws_bytes, osn_stream_t stream);

/* osn_bn_apply with a SECOND destination: the same rows also stored as a column window of a wider matrix
 * (y2 = first element of the window, ld2 = that matrix's row stride in floats) -- ME.cat (models/mink_unet.py:147,155,
 * 163,171) written in place by the producers of its two halves.  y2 null = osn_bn_apply.                           */
int osn_bn_apply2(const float* x, const float* mean, const float* var, const float* gamma,
                  const float* beta, float eps, const float* residual, int relu, float* y,
                  float* y2, int64_t ld2, int64_t n, int c, osn_stream_t stream);
/* osn_bn_forward_train with the second destination of osn_bn_apply2.  Maps of at most 4096 rows (the deep U-Net levels)
 * run as ONE launch (statistics and apply from registers); osn_bn_forward_train does the same.                         */
int osn_bn_forward_train2(const float* x, int64_t n, int c, const float* gamma, const float* beta, float eps,
                          const float* residual, int relu, float momentum, float* mean, float* var,
                          float* running_mean, float* running_var, float* y, float* y2, int64_t ld2,
                          void* ws, size_t ws_bytes, osn_stream_t stream);
/* osn_bn_backward whose incoming gradient is the SUM of n_gy (1..3) row-aligned matrices, each with its own row
 * stride (HOST arrays of device pointers / strides in floats): a block input feeds conv1 and the residual, an encoder
 * output feeds the next stride-2 convolution and -- through ME.cat -- two convolutions of the decoder
 * (models/mink_unet.py:116-174); the sum autograd would form with separate add kernels is formed while reading.   */
int osn_bn_backward_multi(const float* x, const float* y, const float* const* gy_host, const int64_t* gy_ld_host,
                          int n_gy, const float* mean, const float* var, const float* gamma, float eps, int relu,
                          int training, float* gx, float* gres, float* ggamma, float* gbeta,
                          int64_t n, int c, void* ws, size_t ws_bytes, osn_stream_t stream);
/* the same with `beta`: for a batch norm + ReLU WITHOUT a residual, y may be null -- the mask (y > 0) is recomputed from x
 * with the forward pass's own expression, so neither backward pass reads y (models/resnet_base.py:98: bn -> relu)       */
int osn_bn_backward_multi2(const float* x, const float* y, const float* const* gy_host, const int64_t* gy_ld_host, int n_gy,
                           const float* mean, const float* var, const float* gamma, const float* beta, float eps, int relu,
                           int training, float* gx, float* gres, float* ggamma, float* gbeta, int64_t n, int c, void* ws,
                           size_t ws_bytes, osn_stream_t stream);

/* ---- distillation loss on the supervised rows ------------------------------- *
 * Replaces run/distill.py:322-328 and its autograd chain:
 *     output_3d = output_3d[mask]
 *     loss = (1 - torch.nn.CosineSimilarity()(output_3d, feat_3d)).mean()      kind 0 ('cosine')
 *     loss = torch.nn.L1Loss()(output_3d, feat_3d)                             kind 1 ('l1')
 *   out     float [n, d]       the network output, input row order
 *   sel     int64 [n_sel]      rows the loss sees (= mask.nonzero(): distinct, any order), target[j] belongs to out[sel[j]]
 *   target  float [n_sel, d]   feat_3d
 *   loss    float [1]  (device)
 *   state   osn_distill_loss_state_bytes(n, n_sel) bytes kept by the caller from the forward to the backward call
 *   gloss   float [1] (device) gradient of the loss, or null (= 1)
 *   gout    float [n, d]       receives d loss / d out: the closed-form row gradient on the selected rows, ZEROS elsewhere
 *                              (what autograd's index backward produces with an index_add into a zero-filled tensor)
 * osn_distill_loss_check synchronises the stream and reports an index outside [0, n) or a row selected twice
 * (OSN_E_ARG); the kernels themselves never read or write out of bounds.  d % 4 == 0.                          */
size_t osn_distill_loss_state_bytes(int64_t n, int64_t n_sel);
int osn_distill_loss_fwd(const float* out, const int64_t* sel, const float* target, int64_t n, int64_t n_sel, int d, int kind,
                         float* loss, void* state, size_t state_bytes, osn_stream_t stream);
int osn_distill_loss_bwd(const float* out, const float* target, const float* gloss, int64_t n, int64_t n_sel, int d, int kind,
                         float* gout, const void* state, size_t state_bytes, osn_stream_t stream);
/* the same pass; additionally grows[j, :] = gout[sel[j], :] (nullable): the non-zero rows of the gradient once more, compacted,
 * for osn_net_run.goutput_rows (the first n int32 of `state` are osn_net_run.grows_pos)                                  */
int osn_distill_loss_bwd_rows(const float* out, const float* target, const float* gloss, int64_t n, int64_t n_sel, int d, int kind,
                              float* gout, float* grows, const void* state, size_t state_bytes, osn_stream_t stream);
int osn_distill_loss_check(const void* state, int64_t n, int64_t n_sel, osn_stream_t stream);

/* ---- optimizer step over one flat buffer ------------------------------------------------------------ *
 * Replaces optimizer.step() of torch.optim.Adam (run/distill.py:170-178 builds it, :333 steps it) for parameters,
 * gradients and moment estimates laid out as four flat fp32 arrays of n elements (n % 4 == 0; the network executor's
 * gradient buffer has that layout).  step = the 1-based count of this update.  The update rule of torch's Adam
 * (amsgrad = False, maximize = False; L2 weight decay added to the gradient): one launch, 28 bytes per parameter.   */
int osn_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, int64_t step, float lr,
                  float beta1, float beta2, float eps, float weight_decay, osn_stream_t stream);

/* ---- row-aligned elementwise pieces --------------------------------------- *
 * Replaces the stand-alone [ME] MinkowskiReLU (models/mink_unet.py:114, used un-fused by the reference's own module
 * chain), the BasicBlock residual `out += residual` when it is not fused into a batch norm, and ME.cat of two tensors
 * on one coordinate map (models/mink_unet.py:147,155,163,171) with its backward split.  `total` = rows x channels.
 *   osn_relu_fwd   y = max(x, 0)                       osn_relu_bwd   gx = y > 0 ? gy : 0
 *   osn_add        out = a + b  (out may alias a or b)
 *   osn_cat2       out[r] = (a[r, 0:ca], b[r, 0:cb])   osn_cat2_bwd   ga[r] = gout[r, 0:ca], gb[r] = gout[r, ca:ca+cb]
 * cat: ca and cb multiples of 4 (every width of the MinkUNet family is).                                          */
int osn_relu_fwd(const float* x, float* y, int64_t total, osn_stream_t stream);
int osn_relu_bwd(const float* y, const float* gy, float* gx, int64_t total, osn_stream_t stream);
int osn_add(const float* a, const float* b, float* out, int64_t total, osn_stream_t stream);
int osn_cat2(const float* a, int ca, const float* b, int cb, float* out, int64_t n, osn_stream_t stream);
int osn_cat2_bwd(const float* gout, float* ga, int ca, float* gb, int cb, int64_t n, osn_stream_t stream);

/* ---- open-vocabulary query ---------------------------------------------- *
 * Replaces run/evaluate.py:290-292 (and run/distill.py:423-425):
 *   pred = feats[inds_reverse].half() @ text.t();  label = argmax(pred, 1)
 * X float32 [n_rows_x, d]; gather (nullable) int64 [n]; text fp16 [c, d];
 * scores (nullable) fp16 [n, c]; argmax int64 [n].  fp16 MFMA, fp32 accumulate,
 * one rounding to fp16; argmax over the rounded scores, lowest index on ties.    */
int osn_cosine_query(const float* X, const int64_t* gather, const void* text_f16,
                     void* scores_f16, int64_t* argmax, int64_t n, int d, int c,
                     osn_stream_t stream);
/* run/evaluate.py:302-324 (ensemble): per point normalise both feature sources
 * (x / (|x| + 1e-5)), take the source whose best fp16 score is larger (fusion wins
 * only if strictly larger), re-score the selected UN-normalised fp16 features.
 * Each source has its own (nullable) gather: the reference gathers the distilled
 * voxel features with inds_reverse while the fused features are already per point.
 * sel (nullable) uint8 [n] = 1 where the fusion feature was selected.            */
/* labels[p] = argmax over the first c columns of row (gather ? gather[p] : p) of a float32 score matrix with row stride ld
 * (first maximum wins): `torch.max(pred, 1)[1]` of run/evaluate.py:292 fused with the point -> voxel gather `[inds_reverse]`,
 * for scores that exist per VOXEL (the fused-head query).  Indices outside [0, n_rows) read row 0.  */
int osn_rows_argmax(const float* scores, int64_t ld, int c, const int64_t* gather, int64_t n_pts, int64_t n_rows,
                    int64_t* labels, osn_stream_t stream);
size_t osn_query_ensemble_ws_bytes(int64_t n);
int osn_query_ensemble(const float* X_distill, const int64_t* gather_distill,
                       const float* X_fusion, const int64_t* gather_fusion,
                       const void* text_f16, void* scores_f16, int64_t* argmax, uint8_t* sel,
                       int64_t n, int d, int c, void* ws, size_t ws_bytes, osn_stream_t stream);

/* ---- hash voxelisation --------------------------------------------------- *
 * Replaces Voxelizer.voxelize (dataset/voxelizer.py:117-129) +
 * sparse_quantize / fnv_hash_vec (dataset/voxelization_utils.py:9-22,112-132):
 *   grid = floor([xyz,1] @ T^T[:, :3]); grid -= min(grid); key = FNV64(grid)
 *   inds = first occurrence per distinct key in ascending key order,
 *   inverse[p] = rank of key(p).
 * xyz float64 [n,3] device; T12 = first three rows of the 4x4 transform, HOST,
 * row-major (12 doubles).  Outputs: grid_out float64 [n,3] (integral, shifted),
 * inds int64 [<=n], inverse int64 [n]; *n_vox_host on the HOST (one stream sync). */
size_t osn_voxelize_ws_bytes(int64_t n);
int osn_voxelize_fnv(const double* xyz, int64_t n, const double* T12_host,
                     double* grid_out, int64_t* inds, int64_t* inverse, int64_t* n_vox_host,
                     void* ws, size_t ws_bytes, osn_stream_t stream);
/* fnv_hash_vec alone (dataset/voxelization_utils.py:9-22): keys[i] of integral rows. */
int osn_fnv_hash(const double* grid, int64_t n, int ncol, uint64_t* keys, osn_stream_t stream);
/* ravel_hash_vec (dataset/voxelization_utils.py:25-41, the alternative key of sparse_quantize; the path itself
 * uses the FNV key): key = Fortran-style ravel of (coords - column minimum) over the column extents.
 * grid: float64 [n, ncol <= 4] integral values; ws: 128 bytes.                                              */
int osn_ravel_hash(const double* grid, int64_t n, int ncol, uint64_t* keys, void* ws, size_t ws_bytes,
                   osn_stream_t stream);

/* ---- loader-side batch assembly --------------------- *
 * Replaces the index1 / chunk_ind / cumsum chain of FusedFeatureLoader.__getitem__
 * (dataset/feature_loader.py:124-143; val/test :107-113,:166-171): after the voxeliser kept one
 * point per voxel (vox_ind = its `inds`), which voxels carry a fused 2-D feature and which row of
 * the compact feature matrix (one row per True of mask_chunk, in point order) belongs to each:
 *   mask_vox[v] = mask_chunk[vox_ind[v]]          uint8 [n_vox]
 *   src_row[v]  = #True in mask_chunk[0 .. vox_ind[v])  if mask_vox[v] else -1     int64 [n_vox]
 *   indices     = src_row[mask_vox] (voxel order kept; the reference's `indices`)  int64 [<= n_vox]
 * *n_sel_host = number of selected voxels, on the HOST (one stream sync).           */
size_t osn_feature_remap_ws_bytes(int64_t n_points, int64_t n_vox);
int osn_feature_remap(const uint8_t* mask_chunk, const int64_t* vox_ind, int64_t n_points, int64_t n_vox,
                      uint8_t* mask_vox, int64_t* src_row, int64_t* indices, int64_t* n_sel_host,
                      void* ws, size_t ws_bytes, osn_stream_t stream);
/* One scene's rows of the collated coordinate matrix (dataset/feature_loader.py:177-178,204-205):
 * out_coords4[i] = (batch_index, xyz3[i,0], xyz3[i,1], xyz3[i,2]); the caller passes the row offset
 * of the scene inside the batch tensor.                                             */
int osn_batch_coords(const int32_t* xyz3, int64_t n, int batch_index, int32_t* out_coords4,
                     osn_stream_t stream);

/* ---- multi-view feature fusion (the producer of the fused features the distillation trains on) ---- *
 * osn_fusion_project: scripts/feature_fusion/fusion_util.py:93-139 (PointCloudToImageMapper.compute_mapping).
 *   coords3 double [n, 3] (device); world_to_camera16 = np.linalg.inv(camera_to_world), row-major 4 x 4 (HOST);
 *   intrinsic4 = {fx, fy, cx, cy} (HOST); depth double [H, W] in metres (device) or NULL (then: in front of the camera);
 *   image is W x H pixels; cut_bound pixels of border are excluded; mapping int64 [n, 3] = (row, column, visible),
 *   zeros where not visible -- bit-identical to the numpy code (fp64, same operation order and rounding).
 * osn_fusion_accumulate: scripts/feature_fusion/scannet_openseg.py:93-106, one view: for visible points
 *   counter[p] += 1, sum_features[p, :] += feat2d[:, row, column]; feat2d float [D, H, W], sum float [n, D], counter float [n].
 * osn_fusion_finish: scannet_openseg.py:108-109: feat_bank = sum_features / (counter == 0 ? 1e-5 : counter).        */
int osn_fusion_project(const double* coords3, int64_t n, const double* world_to_camera16, const double* intrinsic4,
                       const double* depth, int H, int W, int cut_bound, double vis_thres, int64_t* mapping,
                       osn_stream_t stream);
int osn_fusion_accumulate(const float* feat2d, int D, int H, int W, const int64_t* mapping, int64_t n,
                          float* sum_features, float* counter, osn_stream_t stream);
int osn_fusion_finish(const float* sum_features, const float* counter, int64_t n, int D, float* feat_bank,
                      osn_stream_t stream);

/* ---- network executor: one call per forward / backward pass of a MinkUNet -------------------------------- *
 * Replaces the Python-level walk of models/mink_unet.py:116-174 (MinkUNetBase.forward: conv0 .. block8, final) and
 * of its autograd graph.  The module tree is compiled ONCE into a linear program of stages
 *   x = conv(src)  [-> BN (batch or running statistics) (+ residual) (+ ReLU)]  -> dst  [-> second store into a cat buffer]
 * and the library issues every launch of a pass from one call: kernel choice per stage, scratch, the training-mode
 * statistics kept for the backward pass, the gradient routing (sums of up to three sources are formed inside the
 * batch-norm backward, ME.cat is written in place by its producers).  Same kernels and same results as the per-module
 * entry points above; what disappears is the host work between launches (~40 us per stage of Python + autograd).
 * All `const T*` members of the descriptors are HOST arrays unless noted; activations live in caller-provided arenas. */
#define OSN_NET_MAX_LEVELS 8
typedef struct osn_net_op {
    int32_t K, cin, cout;        /* kernel volume (1, 8, 27, 125), channels                                      */
    int32_t lvl_in, lvl_out;     /* pyramid levels (index into level_rows) of the input / output rows           */
    int32_t map;                 /* index into osn_net_run.maps, -1 <=> K == 1 (identity map)                     */
    int32_t transposed;          /* 1: MinkowskiConvolutionTranspose (runs on the mirrored tables of `map`)       */
    int32_t src;                 /* activation buffer read, -1 = the network input                                */
    int32_t dst;                 /* activation buffer written, -1 = the network output (no batch norm: `final`)   */
    int32_t bn;                  /* index into osn_net_run.bns, -1 = none                                         */
    int32_t relu;                /* ReLU after the batch norm (+ residual)                                        */
    int32_t res;                 /* residual buffer added before the ReLU, -1 = none                              */
    int32_t copy_buf, copy_col;  /* second store of dst: columns [copy_col, copy_col + cout) of buffer copy_buf   */
    int32_t weight;              /* index into osn_net_run.weights                                                */
    int32_t need_dgrad;          /* 0: the input needs no gradient (stem)                                         */
    int32_t fine_unique;         /* 1: `map` is a 2^3 stride-2 map -- every row of its fine side has exactly one pair
                                  *    (osn_spconv_fwd_ws direct mode for the launches that write the fine side)   */
    int32_t reserved;
} osn_net_op;                    /* 72 bytes */
typedef struct osn_net_buf { int32_t level, channels; } osn_net_buf;
typedef struct osn_net_desc {
    int32_t n_ops, n_bufs, n_bns, n_weights, n_maps, n_levels;
    int32_t tl_min_rows;         /* tile-list forward / input gradient on tables of at least this many rows       */
    int32_t tl_mid_rows;         /* ... and from this many rows on when both channel counts are >= 96 (0 = never)  */
    int32_t ws_max_rows;         /* weight-stationary kernel (osn_spconv_fwd_ws) for launches writing at most this
                                  *    many rows (0 = never; the direct mode of fine_unique maps is not limited)   */
    int32_t reserved;
    const osn_net_op* ops;
    const osn_net_buf* bufs;
} osn_net_desc;
enum { OSN_NET_K_NONE = 0, OSN_NET_K_STEM = 1, OSN_NET_K_TL = 2, OSN_NET_K_X6 = 3, OSN_NET_K_WGRAD_TL = 4, OSN_NET_K_WGRAD = 5,
       OSN_NET_K_WS = 6, OSN_NET_K_WS_DIRECT = 7, OSN_NET_K_WGRAD_STEM = 8, OSN_NET_K_DENSE = 9, OSN_NET_K_RG = 10 };
enum { OSN_NET_IMG_X6_FWD = 1, OSN_NET_IMG_X6_DGRAD = 2, OSN_NET_IMG_TL_FWD = 4, OSN_NET_IMG_TL_DGRAD = 8 };
typedef struct osn_net_plan {                 /* every array is caller-provided HOST memory                       */
    uint64_t fwd_arena_bytes, bwd_arena_bytes, ws_bytes;
    uint64_t* x_off;             /* [n_ops]  conv output (pre batch norm) in the forward arena                    */
    uint64_t* stat_off;          /* [n_ops]  mean | var (2 x cout floats) of the training-mode statistics          */
    uint64_t* y_off;             /* [n_bufs] activation buffers in the forward arena                              */
    int32_t* fwd_kernel;         /* [n_ops]  OSN_NET_K_*                                                          */
    int32_t* dgrad_kernel;       /* [n_ops]                                                                       */
    int32_t* wgrad_kernel;       /* [n_ops]                                                                       */
    int32_t* images;             /* [n_ops]  OSN_NET_IMG_* bit mask of the weight images the pass reads            */
} osn_net_plan;
typedef struct osn_net_map {                  /* one kernel map of the coordinate manager (device pointers)       */
    const int32_t* nbr_fwd;      /* [K, rows_out]                                                                 */
    const int32_t* nbr_bwd;      /* [K, rows_in] table of the input gradient (= nbr_fwd for odd stride-1 kernels)  */
    const int32_t* tiles_fwd_rows; const int32_t* tiles_fwd_tbl; const uint32_t* tiles_fwd_gmask;   /* osn_kmap_sort, or null */
    const int32_t* tiles_bwd_rows; const int32_t* tiles_bwd_tbl; const uint32_t* tiles_bwd_gmask;
    const int64_t* counts;       /* [K] pairs per offset                                                          */
    const void* tl_fwd; const int32_t* tl_fwd_rows;     /* osn_tile_lists_build of the forward table (+ permutation) */
    const void* tl_bwd; const int32_t* tl_bwd_rows;
    const void* pl_fwd;          /* osn_pair_lists_build of tl_fwd: weight gradient, weight-stationary convolutions  */
    int32_t K, flip, tl_fwd_bm, tl_bwd_bm;
} osn_net_map;
typedef struct osn_net_weight {
    const float* W;              /* [K, cin, cout]                                                                */
    const void* x6_fwd; const void* x6_dgrad; const void* tl_fwd; const void* tl_dgrad;    /* images (OSN_NET_IMG_*) */
    float* gW;                   /* backward: receives the weight gradient                                        */
} osn_net_weight;
typedef struct osn_net_bn {
    const float* gamma; const float* beta;
    float* running_mean; float* running_var;
    float* ggamma; float* gbeta; /* backward                                                                      */
    float eps, momentum;
} osn_net_bn;
typedef struct osn_prof osn_prof_t;           /* optional launch timer, see osn_prof_create                       */
typedef struct osn_events osn_events_t;       /* pool of HIP events for the fork / join of the backward pass      */
/* osn_net_run.flags.  NO_JOIN (backward pass with a side stream, a call that is NOT the last segment of the pass): return
 * without making `stream` wait for the side stream.  The weight gradients of the executed ops are then final in SIDE-stream
 * order only -- a consumer (the gradient exchange of a segmented pass) must queue behind the side stream; the call that
 * plays the last segment (flags 0) joins everything.  Input gradients a later segment reads are joined by their own events
 * either way.                                                                                                      */
#define OSN_NET_RUN_NO_JOIN 1
/* osn_net_forward, evaluation mode: by default a stage's batch norm (+ residual) (+ ReLU) (+ cat store) is applied in the EPILOGUE of
 * the kernel that finishes the stage's convolution (csrc/epilogue.h: same expression, bitwise the separate launch); this flag keeps
 * the separate osn_bn_apply2 launch (A/B, tests).                                                                         */
#define OSN_NET_RUN_NO_BN_EPILOGUE 2
typedef struct osn_net_run {
    const int64_t* level_rows;   /* [n_levels] rows of every pyramid level                                        */
    const osn_net_map* maps;
    const osn_net_weight* weights;
    const osn_net_bn* bns;
    const float* input;          /* [rows(level of op 0), cin of op 0]                                            */
    float* output;               /* forward: [rows, cout] of the op with dst == -1 (may be null if not run)       */
    const float* goutput;        /* backward: gradient of `output`; may be null when goutput_rows (below) carries every
                                  * non-zero row of it -- the head then ran on those rows only (round 6)               */
    void* fwd_arena; uint64_t fwd_arena_bytes;
    void* bwd_arena; uint64_t bwd_arena_bytes;
    void* ws; uint64_t ws_bytes;
    int32_t* tl_counters;        /* 128 persistent tile counters of this stream (osn_spconv_fwd_tl_pc)            */
    int32_t training;            /* batch statistics + running update (1) or running statistics (0)               */
    int32_t first_op, end_op;    /* ops [first_op, end_op) are executed                                           */
    int32_t flags;               /* OSN_NET_RUN_*                                                                 */
    osn_prof_t* prof;            /* nullable                                                                      */
    /* all nullable: work that is off the main dependency chain runs on `side_stream` -- in both passes the BasicBlock
     * shortcut stages (1x1 conv + batch norm, beside conv1 - BN - conv2), in the backward pass also every weight
     * gradient (needed only at the end of the pass).  The passes fork and join with events of `events`, so nothing
     * outside sees the second stream; results are bitwise those of a single stream.                               */
    osn_stream_t side_stream;
    void* ws_side; uint64_t ws_side_bytes;     /* scratch of the side stream's launches (same size rule as ws)       */
    osn_events_t* events;        /* osn_events_create(2 * n_ops + 2)                                              */
    /* backward, all nullable: `goutput` is zero outside `n_grows` rows (the distillation loss supervises ~20 % of the
     * voxels: run/distill.py:321-326 indexes the output with a mask before the loss, so autograd's gradient of the full
     * output has zero rows everywhere else).  grows_pos[r] = j for the j-th such row, -1 otherwise (osn_distill_loss_* keeps
     * it); grows_idx[j] = r; goutput_rows [n_grows, cout] = goutput[grows_idx].  Given all three, the input and weight
     * gradients of the op that produced `output` run on the n_grows rows only (same values: the other rows contribute
     * exact zeros).                                                                                                */
    const int32_t* grows_pos;
    const int64_t* grows_idx;
    const float* goutput_rows;
    int64_t n_grows;
} osn_net_run;
int osn_net_plan_query(const osn_net_desc* net, const int64_t* level_rows, int training, osn_net_plan* plan);
/* out[j, :] = in[idx[j], :] (c % 4 == 0) and its inverse out[r, :] = pos[r] >= 0 ? in[pos[r], :] : 0 over all n rows: the row
 * compaction around the head's gradients (`output_3d[mask]`, run/distill.py:322, and torch's index backward)        */
int osn_rows_gather(const float* in, const int64_t* idx, int64_t n_idx, int c, float* out, osn_stream_t stream);
int osn_rows_scatter_zero(const float* in, const int32_t* pos, int64_t n, int c, float* out, osn_stream_t stream);
int osn_net_forward(const osn_net_desc* net, const osn_net_run* run, osn_stream_t stream);
int osn_net_backward(const osn_net_desc* net, const osn_net_run* run, osn_stream_t stream);
osn_events_t* osn_events_create(int n);       /* n timing-free HIP events on the current device; null on failure   */
void osn_events_destroy(osn_events_t* e);

/* ---- every kernel map of a scene from one call ------------------------------------------------------- *
 * Replaces the per-map calls behind [ME] CoordinateManager.kernel_map for the convolutions of one forward pass
 * (models/mink_unet.py:47-113: a 5^3 map, five 3^3 maps, four 2^3 stride-2 maps + their transposes).  A job = one map
 * with everything derived from it; null output pointers skip a product.  Jobs name the stream they run on (index into
 * `streams`, [0] = the caller's stream): the maps of different levels are independent chains of latency-bound launches,
 * so they run side by side; the call forks after what is queued on streams[0] (the coordinate pyramid) and joins before
 * it returns.  ws[s] = scratch of stream s (osn_kmap_sort_ws_bytes of the largest table sorted there).  Tables are bit
 * for bit those of osn_kmap_build[_self] / osn_kmap_transpose / osn_kmap_sort / osn_tile_lists_build / osn_pair_lists_build. */
#define OSN_MAPS_MAX_STREAMS 4
typedef struct osn_map_level {               /* one pyramid level (osn_coords_unique[_async])                     */
    const int32_t* coords4; const uint64_t* keys; const int32_t* vals;
    int64_t cap, rows;
} osn_map_level;
typedef struct osn_map_job {
    int32_t lvl_in, lvl_out;     /* table of lvl_in probed at the coordinates of lvl_out                          */
    int32_t ksize, scale;        /* kernel size, offset scale (dilation x tensor stride of the input map)         */
    int32_t self_map;            /* 1: odd kernel over the table's own rows (osn_kmap_build_self; no nbr_bwd)     */
    int32_t stream;              /* index into `streams`                                                          */
    int32_t bm_fwd, bm_bwd;      /* tile rows of the lists (osn_tile_rows)                                        */
    int32_t* nbr_fwd; int32_t* nbr_bwd; int64_t* counts;
    int32_t* order_fwd; int32_t* sorted_fwd; uint32_t* gmask_fwd;
    int32_t* order_bwd; int32_t* sorted_bwd; uint32_t* gmask_bwd;
    void* tl_fwd; void* tl_bwd; void* pl_fwd;
} osn_map_job;                               /* 128 bytes */
int osn_maps_build(const osn_map_level* levels, int n_levels, const osn_map_job* jobs, int n_jobs,
                   const osn_stream_t* streams, void* const* ws, const uint64_t* ws_bytes, int n_streams,
                   osn_events_t* events);

/* Launch timer for the executor (bench.py's roofline entry): HIP events recorded on the launch stream around the
 * convolution launches of selected stages.  tag = op * 4 + phase (0 forward, 1 input gradient, 2 weight gradient).
 * osn_prof_filter: bracket only the listed tags (HOST array; n_tags = 0: every launch).  osn_prof_read synchronises
 * the events and returns the number of records (tags / ms: HOST arrays of `capacity` entries).                     */
osn_prof_t* osn_prof_create(int capacity);
void osn_prof_destroy(osn_prof_t* p);
int osn_prof_filter(osn_prof_t* p, const int32_t* tags, int n_tags);
int osn_prof_read(osn_prof_t* p, int32_t* tags, float* ms, int capacity, int reset);

#ifdef __cplusplus
}
#endif
#endif /* OPENSCENE_AMD_H */
