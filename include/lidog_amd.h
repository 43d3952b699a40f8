/*
 * lidog_amd.h -- C ABI of the MI355X-native sparse-voxel engine (liblidog_amd.so).
 *
 * This is the drop-in boundary for LiDOG's hot path.  The reference reaches
 * this functionality through the python package MinkowskiEngine 0.5.4 and
 * torch.nn; each entry point below cites the reference call site it replaces
 * (paths relative to the reference repository root).  INTEGRATION.md shows the
 * ctypes binding a maintainer adds on the reference side.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HBM) unless its name ends in _host;
 *   - the caller owns all memory; the library never allocates or frees
 *     caller-visible buffers and keeps no state between calls;
 *   - `stream` is a hipStream_t passed as void*; calls are asynchronous on it
 *     unless documented as synchronising;
 *   - return value 0 = success, non-zero = failure with a message in
 *     lidog_last_error() (thread-local);
 *   - coordinates are int32 rows (batch, x, y, z); features are float32
 *     row-major [rows, channels]; all index tables are int32.
 *   - coordinate components must lie in [-65536, 65535], batch in [0, 4095]
 *     (63-bit packed hash key); violations set *err_flag (device int32) to 1.
 */
#ifndef LIDOG_AMD_H
#define LIDOG_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

const char *lidog_last_error(void);
int lidog_abi_version(void);

/* ------------------------------------------------------------------ coordinate maps
 * Replaces MinkowskiEngine's CoordinateManager (insert_and_map / stride / kernel map),
 * reached from ME.SparseTensor(...) at utils/pipelines/trainer_lighting_2d.py:151 and
 * implicitly from every ME.MinkowskiConvolution call in utils/models/minkunet_bev.py:307-371. */

/* slots needed for n rows (power of two, load factor <= 0.5) */
int64_t lidog_hash_capacity(int64_t n);

/* Build the hash table of a coordinate map: keys[cap] (uint64), vals[cap] (int32 row of the FIRST
 * occurrence of that coordinate).  first_row[i] = vals of row i's key (== i for unique input).
 * n_unique_dev receives the number of distinct coordinates (device int64). */
int lidog_coords_insert(const int32_t *coords, int64_t n, uint64_t *keys, int32_t *vals, int64_t cap,
                        int32_t *first_row, int64_t *n_unique_dev, int32_t *err_flag, void *stream);
/* the same, and in the same pass what a caller needs to size occupancy bitmaps and batch loops:
 * info [9] int64 (device) = (unique rows, error flag, largest batch index, lowest x, y, z, highest x, y, z) -- one
 * read-back instead of n_unique + err + three device-wide min / max reductions of the caller's own. */
int lidog_coords_insert_info(const int32_t *coords, int64_t n, uint64_t *keys, int32_t *vals, int64_t cap,
                             int32_t *first_row, int64_t *info, int32_t *err_flag, void *stream);

/* Compact a map with duplicates: unique_rows[j] = first-occurrence row of output row j (ascending),
 * inverse[i] = output row of input row i; rewrites vals to output rows.  scan_ws: int32[n + 2048]. */
int lidog_coords_compact(const int32_t *first_row, int64_t n, const uint64_t *keys, int32_t *vals, int64_t cap,
                         const int32_t *coords, int32_t *unique_rows, int32_t *inverse, int32_t *scan_ws,
                         void *stream);

/* Strided map (conv kernel 2 stride 2, minkunet_bev.py:62,69,76,83): out coordinate =
 * floor(c / new_stride) * new_stride per spatial dim, duplicates collapsed, rows in first-occurrence
 * order of the parent.  coords_out has room for n_in rows; n_out_dev (device int64) gets the count.
 * parent2child[i] = output row of parent row i.  ws: int32[2 * n_in + 2048]. */
int lidog_coords_stride(const int32_t *coords_in, int64_t n_in, int32_t new_stride, uint64_t *keys,
                        int32_t *vals, int64_t cap, int32_t *parent2child, int32_t *coords_out,
                        int64_t *n_out_dev, int32_t *ws, int32_t *err_flag, void *stream);

/* Occupancy bitmap of a coordinate map over its bounding box (one bit per cell, x fastest, then y, z, batch; x0 / y0 / z0
 * and every coordinate multiples of `stride`): lidog_kernel_map_bits tests the bit of a neighbour's cell before it
 * probes the hash table -- 84 % (3^3) to 90 % (5^3) of the neighbours of a LiDAR voxel do not exist, and each of those
 * probes was a random 64-byte fetch.  lidog_bitmap_words: uint32 words of the bitmap, -1 if it would exceed max_bytes
 * (the caller then passes bits = NULL: plain probes).  lidog_bitmap_set: bits zeroed by the caller; *err_flag = 2 if a
 * coordinate lies outside the box.  The neighbour table is the same with or without the bitmap. */
int64_t lidog_bitmap_words(int32_t nx, int32_t ny, int32_t nz, int32_t nb, int64_t max_bytes);
int lidog_bitmap_set(const int32_t *coords, int64_t n, int32_t x0, int32_t y0, int32_t z0, int32_t nx, int32_t ny,
                     int32_t nz, int32_t stride, int32_t nb, uint32_t *bits, int32_t *err_flag, void *stream);
int lidog_kernel_map_bits(const int32_t *coords_out, int64_t n_out, const uint64_t *in_keys, const int32_t *in_vals,
                          int64_t in_cap, const int32_t *offsets_host, int32_t K, const uint32_t *bits, int32_t x0,
                          int32_t y0, int32_t z0, int32_t nx, int32_t ny, int32_t nz, int32_t stride, int32_t nb,
                          int32_t *nbr, void *stream);
/* Neighbour table of a kernel map whose offsets are a SUBSET of another map's over the same coordinate map (same tensor
 * stride and dilation): the 3^3 map of the stride-1 BasicBlocks (minkunet_bev.py:371, block8) from the stem's 5^3 table
 * (:57, conv0p1s1).  nbr [K, n] row k = nbr_big [K_big, n] row sel_host[k]; no hash probes, same table bit for bit. */
int lidog_kernel_map_subset(const int32_t *nbr_big, int64_t n, int32_t K_big, const int32_t *sel_host, int32_t K,
                            int32_t *nbr, void *stream);

/* Kernel map, neighbour-table form, k-major: nbr[k * n_out + o] = row of the input map holding
 * coordinate out[o] + offsets[k], or -1.  offsets_host is a HOST array [K,3] (x,y,z), K <= 125. */
int lidog_kernel_map(const int32_t *coords_out, int64_t n_out, const uint64_t *in_keys, const int32_t *in_vals,
                     int64_t in_cap, const int32_t *offsets_host, int32_t K, int32_t *nbr, void *stream);

/* Rule book from the neighbour table (wavefront ballot + prefix-sum compaction): for each k the pairs
 * (pair_in, pair_out) in ascending out-row order, k_off_dev[K+1] segment offsets (device int64),
 * pos_out[k*n_out+o] = pair position or -1, pos_in[k*n_in+i] = pair position or -1 (both may be NULL when the map
 * is never walked by row: the 5^3 stem).
 * pair_* need room for n_out*K entries unless the caller knows P.  ws: int32[(blocks+1)*K + K + 2],
 * blocks = ceil(n_out / 1024). */
int lidog_kernel_map_pairs(const int32_t *nbr, int64_t n_out, int64_t n_in, int32_t K, int64_t *k_off_dev,
                           int32_t *pair_in, int32_t *pair_out, int32_t *pos_out, int32_t *pos_in, int32_t *ws,
                           void *stream);
/* Per-row lists of a rule book: row_ptr [n+1], row_list [P] = the pair positions of row o (entries of pos [K][n] that
 * are >= 0) in ascending offset order; what lidog_sconv_reduce_rows[_stats] walk.  mark_k >= 0: the entry of that
 * offset is stored as -1 (a consumer that computes that offset's product itself; unused by this library's path).  ws: ceil((n+1)/1024)+1 ints. */
int lidog_kernel_map_rows(const int32_t *pos, int64_t n, int32_t K, int32_t mark_k, int32_t *row_ptr,
                          int32_t *row_list, int32_t *ws, void *stream);

/* ------------------------------------------------------------------ sparse convolution
 * Replaces ME.MinkowskiConvolution / MinkowskiConvolutionTranspose forward and backward
 * (minkunet_bev.py:57-123 construction, :307-371 calls; kernel layout [K,Cin,Cout], :404). */

/* Gathered GEMM over the rule book: for every tile t (tile_k[t], tile_row0[t], tile_rows[t] <= 128)
 *   T[dst(p)][:] = A[gather[p]][:] . B[tile_k]      p = tile_row0 .. tile_row0 + tile_rows
 * B is [K, Cin, Cout] row-major.  dst(p) = p when scatter == NULL, else scatter[p] (each destination
 * written once).  The product of one row is an fmaf chain over ci ascending starting from 0
 * (bit-identical to oracle/me_oracle.c:orc_conv_fwd before its scatter-add).
 * bias (may be NULL) [Cout] is added after the chain.
 * a_rows: rows of A (every gather index, or every p when gather == NULL, is below it); 0 = not known.  Only a hint for
 * the choice of kernel (an A below 4 GiB lets a workgroup walk several tiles with 32-bit row offsets); never read as a bound. */
int lidog_sconv_gemm(const float *A, const int32_t *gather, const float *B, const float *bias,
                     const int32_t *tile_k, const int32_t *tile_row0, const int32_t *tile_rows, int32_t n_tiles,
                     int32_t Cin, int32_t Cout, float *T, const int32_t *scatter, int64_t a_rows, void *stream);
/* The matrix-core gathered GEMM gives a workgroup several (tile, column tile) units -- the next unit's rows are fetched
 * while the current one is multiplied -- when a launch is more than one round of resident workgroups, A is below 4 GiB
 * (a_rows known) and there is neither a scatter index nor a bias; results are bit-identical either way.  multi: 1 / 0 =
 * on / off (< 0: unchanged; default on, LIDOG_GEMM_MULTI=0 in the environment turns it off); slots > 0: plan as if that
 * many workgroups were resident (0: ask the device).  Returns the previous `multi`. */
int32_t lidog_sconv_gemm_units(int32_t multi, int32_t slots);
/* T = addend + product of the same gathered GEMM, narrow inputs only (Cin <= 8, Cout % 4 == 0): the data gradient of the
 * classifier `final` (minkunet_bev.py:118-123, 7 -> 96) lands on rows that already hold the BEV head's gradient.  Same
 * bits as lidog_sconv_gemm + lidog_add(addend, product).  Returns 3 for other shapes (nothing launched). */
int lidog_sconv_gemm_addend(const float *A, const int32_t *gather, const float *B, const int32_t *tile_k,
                            const int32_t *tile_row0, const int32_t *tile_rows, int32_t n_tiles, int32_t Cin,
                            int32_t Cout, const float *addend, float *T, const int32_t *scatter, void *stream);

/* out[o][:] = sum over k ascending of T[pos[k*n + o]][:] (skipping -1), plus bias if not NULL.
 * The gather->GEMM->scatter-add order of the ME CPU algorithm, without atomics. */
int lidog_sconv_reduce(const float *T, const int32_t *pos, int64_t n, int32_t K, int32_t C, const float *bias,
                       const float *addend /* [n,C] added last (may be NULL): a second gradient of the same tensor */,
                       float *out, void *stream);

/* lidog_sconv_reduce with the BatchNorm statistics of `out` folded into the same pass: sums[0..C) = sum over
 * rows of out, sums[C..2C) = sum of squares (fp64, added in a fixed order: bit-reproducible; overwritten).
 * partial_ws: lidog_sconv_reduce_stats_ws(n, C) doubles.  C % 4 == 0. */
int64_t lidog_sconv_reduce_stats_ws(int64_t n, int32_t C);
int lidog_sconv_reduce_stats(const float *T, const int32_t *pos, int64_t n, int32_t K, int32_t C, const float *bias,
                             float *out, double *sums, double *partial_ws, double count, float eps, float momentum,
                             float *mean, float *invstd, float *running_mean, float *running_var, void *stream);
/* The same two reductions over the per-row lists of lidog_kernel_map_rows (only the offsets a voxel really has are
 * read; same additions in the same order: bit-identical results). */
int lidog_sconv_reduce_rows(const float *T, const int32_t *row_ptr, const int32_t *row_list, int64_t n, int32_t C,
                            const float *bias, const float *addend, float *out, void *stream);
int lidog_sconv_reduce_rows_stats(const float *T, const int32_t *row_ptr, const int32_t *row_list, int64_t n,
                                  int32_t C, const float *bias, float *out, double *sums, double *partial_ws,
                                  double count, float eps, float momentum, float *mean, float *invstd,
                                  float *running_mean, float *running_var, void *stream);
/* Data-gradient reduction (lidog_sconv_reduce_rows without bias) whose epilogue is lidog_bn_bwd_reduce of the layer that
 * produced the rows: `out` is the complete gradient dy of that layer's BatchNorm (+ ReLU) output
 * (ME.MinkowskiBatchNorm backward, minkunet_bev.py:60), `pre` / mean / invstd its saved input and statistics,
 * relu_y or (relu_w, relu_b) its ReLU mask as in lidog_bn_bwd_reduce.  sums / count / dw / db as there; the sums are
 * bit-identical to calling lidog_bn_bwd_reduce on `out` afterwards (same partial sums in the same order), one pass
 * over dy less.  relu_bits: the mask as written by lidog_bn_apply_bits instead of relu_y.
 * partial_ws: lidog_bn_reduce_ws(C, 1) doubles. */
int lidog_sconv_reduce_rows_bwdstats(const float *T, const int32_t *row_ptr, const int32_t *row_list, int64_t n,
                                     int32_t C, const float *addend, float *out, const float *pre,
                                     const float *relu_y, const uint32_t *relu_bits, const float *mean,
                                     const float *invstd, const float *relu_w, const float *relu_b, double *sums,
                                     double *partial_ws, double count, float *dw, float *db, void *stream);
/* Validation path (running statistics, minkunet_bev.py:376-393): the reduction with the evaluation-mode BatchNorm
 * (+ residual + ReLU) in its epilogue, same expression and order as lidog_bn_apply. */
int lidog_sconv_reduce_rows_bn(const float *T, const int32_t *row_ptr, const int32_t *row_list, int64_t n, int32_t C,
                               const float *bias, const float *mean, const float *invstd, const float *w,
                               const float *b, const float *residual, int32_t relu, float *out, void *stream);
/* count / eps / momentum / mean / invstd / running_*: as for lidog_bn_stats below (the last kernel of the
 * reduction also stores the row count behind the sums and, when mean != NULL, finalises the statistics). */

/* Cin == 1 convolution (the 5^3 stem) straight from the neighbour table nbr [K][n] of lidog_kernel_map, no product
 * rows: out[o] = sum_k x[nbr[k][o]] * W[k][:] (+ bias), bit-identical to lidog_sconv_gemm + lidog_sconv_reduce.
 * Cout in {16, 32, 64}. */
int lidog_sconv_cin1(const float *x, const int32_t *nbr, const float *W, const float *bias, int64_t n, int32_t K,
                     int32_t C, float *out, void *stream);

/* gW[k] = sum over the pairs p of offset k of A[pair_a[p]]^T . G[pair_g[p]]   ([Cin,Cout] per k).
 * The pair list is cut on the host into work items of (nearly) equal length that never straddle an offset:
 * items [3, n_items] int32 = (k, first pair, end pair), ordered by k; item_off [K+1] = first item of every k.
 * Every item writes its own partial [Cin,Cout] slot(s); the slots of one offset are then summed in order.
 * partial: lidog_sconv_wgrad_slabs(Cin, Cout, n_items) * Cin * Cout floats. */
int lidog_sconv_wgrad(const float *A, const int32_t *pair_a, const float *G, const int32_t *pair_g,
                      const int32_t *items, int32_t n_items, const int32_t *item_off, int32_t K, int32_t Cin,
                      int32_t Cout, float *partial, float *gW, void *stream);

/* slots of Cin*Cout floats that `partial` must hold for lidog_sconv_wgrad with n_items work items */
int lidog_sconv_wgrad_slabs(int32_t Cin, int32_t Cout, int32_t n_items);
/* workgroups of lidog_sconv_wgrad (fold != 0: lidog_sconv_wgrad_in_bn) for Cin x Cout that are resident on the chip at a
 * time (occupancy x CUs): the caller cuts the rule book so that a launch is a whole number of rounds (me._wgrad_items;
 * ME's own weight gradient, MinkowskiConvolution backward, has no such knob).  0: not a matrix-core shape. */
int32_t lidog_sconv_wgrad_slots(int32_t Cin, int32_t Cout, int32_t fold);

/* Arithmetic core of lidog_sconv_gemm / lidog_sconv_wgrad: 1 = exact-f32 MFMA (default), 0 = vector FMA.
 * Both produce bit-identical results (an f32 MFMA is a k-ordered fmaf chain); kept selectable for A/B. */
int lidog_set_sparse_core(int32_t core);
int lidog_get_sparse_core(void);

/* Wt[k][co][ci] = W[k][ci][co] */
int lidog_transpose_kernel(const float *W, int32_t K, int32_t Cin, int32_t Cout, float *Wt, void *stream);
/* all kernels of a model at once (after the optimiser step): desc int64 [n_mats][6] = (src offset, dst offset, K, Cin,
 * Cout, first 32x32 tile), offsets in floats into src / dst */
int lidog_transpose_batched(const float *src, float *dst, const int64_t *desc, int32_t n_mats, int64_t total_tiles,
                            void *stream);

/* ------------------------------------------------------------------ BatchNorm / ReLU on COO features
 * Replaces ME.MinkowskiBatchNorm (nn.BatchNorm1d over all rows, minkunet_bev.py:60,406-408) and
 * ME.MinkowskiReLU (:124).  layout: x[n, C] when hw == 1; NCHW with hw = H*W otherwise
 * (nn.BatchNorm2d of utils/models/conv2d.py:18,21). */

/* per-channel sums in double: sums[0..2C) = (sum x, sum x^2), overwritten.  ws: lidog_bn_reduce_ws(C, hw) doubles of
 * scratch -- for NCHW input (hw > 1) that many PER IMAGE (n x) -- holding per-workgroup partials that are added in a
 * fixed order: no atomics anywhere, every sum is run-to-run reproducible (0 = not needed, ws may be NULL).
 * count > 0: also stored at sums[2*C] (SyncBatchNorm all-reduces the row count together with the sums).
 * mean != NULL: lidog_bn_finalize(sums, count, ...) is folded into the same launch (local BatchNorm). */
int64_t lidog_bn_reduce_ws(int32_t C, int64_t hw);
int lidog_bn_stats(const float *x, int64_t n, int32_t C, int64_t hw, double *sums, double *ws, double count,
                   float eps, float momentum, float *mean, float *invstd, float *running_mean, float *running_var,
                   void *stream);
/* mean/invstd from sums and count; updates running stats (momentum, unbiased var) when not NULL.
 * count <= 0: the count is read from sums[2*C] on the device (SyncBatchNorm all-reduces it with the sums);
 * the same convention holds for lidog_bn_bwd_apply. */
int lidog_bn_finalize(const double *sums, double count, int32_t C, float eps, float momentum, float *mean,
                      float *invstd, float *running_mean, float *running_var, void *stream);
/* y = (x - mean) * invstd * w + b (+ residual) (relu).  In-place (y == x) allowed. */
int lidog_bn_apply(const float *x, int64_t n, int32_t C, int64_t hw, const float *mean, const float *invstd,
                   const float *w, const float *b, const float *residual, int32_t relu, float *y, void *stream);
/* backward reduce: sums[0..2C) = (sum dy', sum dy'*xhat) with dy' = dy * (y > 0) when relu_y != NULL; ws as above;
 * count > 0: stored at sums[2*C]; db / dw != NULL: the LOCAL parameter gradients (float copies of the two sums) */
/* relu_w / relu_b (both or neither; relu_y must then be NULL; [rows, C] with C % 4 == 0 only): the ReLU mask is
 * recomputed from x with the forward pass's own expression ((x - mean) * invstd * w + b > 0, same bits) instead of
 * being read from the saved output -- for a BatchNorm + ReLU WITHOUT residual, one tensor less to stream. */
/* upper bound of the workgroups (= rows of fp64 partial sums) of the per-row statistics reductions */
int32_t lidog_stats_max_blocks(void);
/* workgroups (= rows of partial sums) lidog_bn_bwd_reduce uses for [n, C] rows, C % 4 == 0 */
int64_t lidog_bn_bwd_reduce_blocks(int64_t n, int32_t C);
int lidog_bn_bwd_reduce(const float *dy, const float *x, const float *relu_y, int64_t n, int32_t C, int64_t hw,
                        const float *mean, const float *invstd, double *sums, double *ws, double count, float *dw,
                        float *db, const float *relu_w, const float *relu_b, void *stream);
/* dx = w*invstd*(dy' - s0/count - xhat*s1/count); dres = dy' when dres != NULL; dw = s1, db = s0 when != NULL;
 * relu_b: as above (w is the BatchNorm weight already) */
int lidog_bn_bwd_apply(const float *dy, const float *x, const float *relu_y, int64_t n, int32_t C, int64_t hw,
                       const float *mean, const float *invstd, const float *w, const double *sums, double count,
                       float *dx, float *dres, float *dw, float *db, const float *relu_b, void *stream);
/* ReLU masks as bits ([rows, C] with C % 4 == 0; MinkowskiReLU after a residual add, resnet_block.py:8-56): the apply
 * pass also writes bit 4 * (q % 8) + j of word q / 8 = (element j of float4 number q of y) > 0 into relu_bits
 * [lidog_relu_bits_words(n, C)], and the two backward passes read that instead of the saved output y (1/32 of the
 * bytes; same mask, same results).  relu_bits == NULL: exactly the functions above. */
int64_t lidog_relu_bits_words(int64_t n, int32_t C);
int lidog_bn_apply_bits(const float *x, int64_t n, int32_t C, int64_t hw, const float *mean, const float *invstd,
                        const float *w, const float *b, const float *residual, int32_t relu, float *y,
                        uint32_t *relu_bits, void *stream);
/* SyncBatchNorm forward in one launch: mean / invstd / running statistics from the ALL-REDUCED sums (the row count
 * behind them) + the apply pass; = lidog_bn_finalize(sums, -1, ...) followed by lidog_bn_apply_bits, bit for bit */
int lidog_bn_apply_sync(const float *x, int64_t n, int32_t C, const double *sums, float eps, float momentum, float *mean,
                        float *invstd, float *running_mean, float *running_var, const float *w, const float *b,
                        const float *residual, int32_t relu, float *y, uint32_t *relu_bits, void *stream);
int lidog_bn_bwd_reduce_bits(const float *dy, const float *x, const float *relu_y, const uint32_t *relu_bits, int64_t n,
                             int32_t C, int64_t hw, const float *mean, const float *invstd, double *sums, double *ws,
                             double count, float *dw, float *db, const float *relu_w, const float *relu_b,
                             void *stream);
int lidog_bn_bwd_apply_bits(const float *dy, const float *x, const float *relu_y, const uint32_t *relu_bits, int64_t n,
                            int32_t C, int64_t hw, const float *mean, const float *invstd, const float *w,
                            const double *sums, double count, float *dx, float *dres, float *dw, float *db,
                            const float *relu_b, void *stream);
/* out[c] = sum over the n rows of x[., c] for a narrow matrix (C <= 16; the classifier's bias gradient); ws: 512*C
 * doubles */
int lidog_colsum(const float *x, int64_t n, int32_t C, float *out, double *ws, void *stream);
int64_t lidog_colsum_ws(int32_t C); /* doubles of workspace lidog_colsum needs */
/* evaluation-mode BatchNorm (running statistics, minkunet_bev.py:376-393 validation path): invstd = 1/sqrt(var+eps) */
int lidog_bn_eval_invstd(const float *running_var, float eps, int32_t C, float *invstd, void *stream);
/* ME.cat(a, b) on one coordinate map (minkunet_bev.py:337,348,359,370) and its backward (two contiguous gradients) */
int lidog_cat2(const float *a, int32_t Ca, const float *b, int32_t Cb, int64_t n, float *out, void *stream);
int lidog_split2(const float *g, int32_t Ca, int32_t Cb, int64_t n, float *ga, float *gb, void *stream);
int lidog_relu_fwd(const float *x, int64_t n, float *y, void *stream);
int lidog_relu_bwd(const float *dy, const float *y, int64_t n, float *dx, void *stream);
int lidog_add(const float *a, const float *b, int64_t n, float *out, void *stream);

/* ------------------------------------------------------------------ BEV projection
 * Replaces MinkUNetBaseBEV.sparse2super (minkunet_bev.py:158-230) without materialising the
 * [H,W,C] tensor: a [B,H,W] int32 winner map (last row wins) + fused view-scramble + MaxPool2d. */
int lidog_bev_winner(const int32_t *coords, int64_t n, const int32_t *lut_x, const int32_t *lut_y, int32_t lut_lo,
                     int32_t lut_n, int32_t H, int32_t W, int32_t *winner /*[B,H,W], pre-filled -1*/,
                     int32_t *pixel /*[n] linear b*H*W+py*W+px or -1*/, void *stream);
/* winner/pixel/n: as produced by lidog_bev_winner for the same rows (only windows that contain a cell of an
 * occupied pixel are computed; the rest of out/argsrc is filled with 0 / -1).
 * rowbits (may be NULL; when given, argsrc is written for the computed windows only and NOT filled elsewhere):
 * uint64 [B*C*Ho][ceil(Wo/64)], bit x of row (b,c,yo) set for every computed window: the
 * structural support of `out` (a superset of argsrc >= 0) in the form lidog_conv2d_support keeps at the start of
 * its `act` buffer -- pass that buffer here and call lidog_conv2d_support(NULL, ...). */
int lidog_bev_pool_fwd(const float *feats, int32_t C, const int32_t *winner, const int32_t *pixel, int64_t n,
                       int32_t B, int32_t H, int32_t W, int32_t pk, int32_t ps, int32_t pp, int32_t Ho, int32_t Wo,
                       float *out /*[B,C,Ho,Wo]*/,
                       int32_t *argsrc /*[B,C,Ho,Wo] row*C+c of the arg-max cell or -1*/, uint64_t *rowbits,
                       void *stream);
/* Backward of lidog_bev_pool_fwd as a gather over (voxel row, channel), no atomics (bit-reproducible): every row that
 * targets a pixel receives the gradient of the pixel's winning row's cells (index_put's backward, minkunet_bev.py:217),
 * a cell's gradient is the sum over the windows whose arg-max it was, in ascending window order. */
int lidog_bev_pool_bwd(const float *gout /*[B,C,Ho,Wo]*/, const int32_t *argsrc, const int32_t *winner,
                       const int32_t *pixel, int64_t n, int32_t C, int32_t B, int32_t H, int32_t W, int32_t pk,
                       int32_t ps, int32_t pp, int32_t Ho, int32_t Wo, float *gfeats /*[n,C]*/, void *stream);

/* ------------------------------------------------------------------ dense 2-D BEV head (MFMA)
 * Replaces nn.Conv2d(k3,s2,p1,bias=False) x2 and nn.Conv2d(k1) of Encoder2D
 * (utils/models/conv2d.py:16-22,116,184-185).  NCHW float32, exact-f32 MFMA. */
int lidog_conv2d_fwd(const float *x, const float *w, const float *bias, int32_t B, int32_t Cin, int32_t H,
                     int32_t W, int32_t Cout, int32_t ksize, int32_t stride, int32_t pad, float *y, void *stream);
/* ws: 9*Cin*Cout floats (parity-class repacked weights) for k3 s2 p1; unused for k1 */
int lidog_conv2d_dgrad(const float *gy, const float *w, int32_t B, int32_t Cin, int32_t H, int32_t W,
                       int32_t Cout, int32_t ksize, int32_t stride, int32_t pad, float *gx, float *ws,
                       void *stream);
/* ws: split-K partial slabs, summed in order (no atomics).  k3 s2 p1: as many slabs of Cout*Cin*9 floats as fit
 * (32 are enough); k1: 4*B slabs of Cout*Cin + Cout floats. */
int lidog_conv2d_wgrad(const float *x, const float *gy, int32_t B, int32_t Cin, int32_t H, int32_t W,
                       int32_t Cout, int32_t ksize, int32_t stride, int32_t pad, float *gw, float *gbias,
                       float *ws, int64_t ws_floats, void *stream);

/* Conv2d(k3,s2,p1,bias=False) over a structurally sparse input (the image made by sparse2super: ~95 % empty cells).
 * support [B,Cin,H,W] int32: >= 0 where the cell can be non-zero / its gradient is needed (the arg-max source map
 * of lidog_bev_pool_fwd), < 0 where x is exactly 0 and its gradient is never read.
 * lidog_conv2d_support turns it into per-tile lists of active input channels (act: lidog_conv2d_support_ws int32;
 * support == NULL: the row bitmasks at the start of act were already written by lidog_bev_pool_fwd);
 * fwd_sparse = lidog_conv2d_fwd restricted to them (bit-identical result); dgrad_sparse writes gx ONLY for the
 * active channels of every 128-pixel tile (a superset of the cells with support >= 0), the rest of gx is untouched;
 * wgrad_sparse = lidog_conv2d_wgrad visiting, for every group of 3 input channels, only the pixel tiles in which
 * one of them is active (equal up to the order of the split sums).
 * ws: Cin*9*Cout floats (transposed / parity-class repacked weights); wgrad: lidog_conv2d_wgrad_sparse_ws floats
 * (one slab of Cout*Cin*9 per work item of a channel group).  Cin <= 128, Cout % 128 == 0. */
int64_t lidog_conv2d_support_ws(int32_t B, int32_t Cin, int32_t H, int32_t W);
int lidog_conv2d_support(const int32_t *support, int32_t B, int32_t Cin, int32_t H, int32_t W, int32_t *act,
                         void *stream);
int lidog_conv2d_fwd_sparse(const float *x, const float *w, const int32_t *act, int32_t B, int32_t Cin, int32_t H,
                            int32_t W, int32_t Cout, float *y, float *ws, void *stream);
int lidog_conv2d_dgrad_sparse(const float *gy, const float *w, const int32_t *act, int32_t B, int32_t Cin, int32_t H,
                              int32_t W, int32_t Cout, float *gx, float *ws, void *stream);
/* FLOPs the three support-restricted kernels execute for the lists in `act` (out[0] forward, [1] data gradient,
 * [2] weight gradient; padded stages included): measurement only, reads the list headers back (synchronises) */
int lidog_conv2d_support_work(const int32_t *act, int32_t B, int32_t Cin, int32_t H, int32_t W, int32_t Cout,
                              double *out, void *stream);
int64_t lidog_conv2d_wgrad_sparse_ws(int32_t B, int32_t Cin, int32_t H, int32_t W, int32_t Cout); /* floats of ws */
int lidog_conv2d_wgrad_sparse(const float *x, const float *gy, const int32_t *act, int32_t B, int32_t Cin, int32_t H,
                              int32_t W, int32_t Cout, float *gw, float *ws, int64_t ws_floats, void *stream);

/* ------------------------------------------------------------------ DICE losses
 * utils/losses/losses.py:56-97 (DICELoss: soft = 0) and :100-187 (SoftDICELoss: soft = 1, label smoothing eps,
 * powerize, present-class mask), as called by trainer_lighting_2d.py:172-190.  logits [n, C] float32 (C in
 * {2, 7, 8, 16}), target [n] int64; rows with target == ignore_label are skipped when has_ignore != 0.
 * fwd: loss[0] = 1 - mean present-class DICE + offset (offset = -1 for neg_range); coef [2C] receives the
 * per-class gradient coefficients bwd needs; ws: lidog_dice_ws(C) doubles.  bwd: glogits = d loss / d logits * gout[0]. */
int64_t lidog_dice_ws(int32_t C);
int lidog_dice_fwd(const float *logits, const int64_t *target, int64_t n, int32_t C, int64_t ignore_label,
                   int32_t has_ignore, float eps, int32_t soft, int32_t powerize, int32_t use_tmask, float offset,
                   double *ws, float *loss, float *coef, void *stream);
int lidog_dice_bwd(const float *logits, const int64_t *target, int64_t n, int32_t C, int64_t ignore_label,
                   int32_t has_ignore, float eps, int32_t soft, int32_t powerize, const float *coef, const float *gout,
                   float *glogits, void *stream);

/* ------------------------------------------------------------------ data path next to the hot path
 * ME.utils.sparse_quantize (utils/datasets/semantickitti_bev.py:232-238) = lidog_voxel_floor +
 * lidog_coords_insert + lidog_coords_compact + lidog_label_vote; PC2ImgConverter.getBEVImageNew
 * (utils/datasets/semantickitti_bev.py:433-464) = lidog_bev_label_raster. */
/* rows[i] = (batch, floor(x/qx), floor(y/qy), floor(z/qz)) in float32 arithmetic; points [n,3] */
int lidog_voxel_floor(const float *points, int64_t n, float qx, float qy, float qz, int32_t batch, int32_t *rows,
                      void *stream);
/* voxel_labels[j] = label of the voxel's first point, or ignore_label when its points disagree */
int lidog_label_vote(const int32_t *labels, const int32_t *unique_rows, const int32_t *inverse, int64_t n, int64_t m,
                     int32_t ignore_label, int32_t *voxel_labels, void *stream);
/* point_idx [B,S,S] (pre-filled with -1) = index inside its scan of the last labelled in-bounds voxel per pixel,
 * img_labels [B,S,S] int64 = its label or -1.  lut_x/lut_y: pixel per integer coordinate or -1, lut_z: 1 inside
 * the z range.  batch_start [B+1]: first row of every scan. */
int lidog_bev_label_raster(const int32_t *coords, const int32_t *labels, int64_t n, const int32_t *lut_x,
                           const int32_t *lut_y, const int32_t *lut_z, int32_t lut_lo, int32_t lut_n, int32_t B,
                           int32_t S, const int64_t *batch_start, int32_t *point_idx, int64_t *img_labels,
                           void *stream);

/* ------------------------------------------------------------------ optimiser
 * Replaces torch.optim.Adam(lr, weight_decay) of trainer_lighting_2d.py:356-358 on a flat buffer. */
int lidog_adam_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, int64_t n, float lr,
                    float beta1, float beta2, float eps, float weight_decay, int32_t step, float grad_scale,
                    void *stream);
/* torch.optim.SGD(lr, momentum=0.98, weight_decay, nesterov=True) of trainer_lighting_2d.py:351-355 (and
 * trainer_lighting.py:337-341) on a slice of the flat buffers; momentum_buf starts at zero. */
int lidog_sgd_step(float *param, const float *grad, float *momentum_buf, int64_t n, float lr, float momentum,
                   float weight_decay, int32_t nesterov, float grad_scale, void *stream);

/* ------------------------------------------------------------------ collectives (RCCL over xGMI)
 * The two collectives of the data-parallel path for hosts that do not go through torch.distributed: the gradient
 * all-reduce of Lightning DDP (train_lidog.py:227-231, strategy='ddp'; sum, the caller folds 1/world into the
 * optimiser step) and the SyncBatchNorm statistics all-reduce (ME.MinkowskiSyncBatchNorm, train_lidog.py:228; the
 * (sum, sum, rows) vectors of lidog_bn_stats / lidog_bn_bwd_reduce are fp64).  librccl is loaded at first use.
 * One process per GPU: rank 0 calls lidog_comm_unique_id and ships the bytes to the others (any host channel),
 * every rank calls lidog_comm_init_rank with its device current.  In place, asynchronous on `stream`. */
int32_t lidog_comm_unique_id_bytes(void);
int lidog_comm_unique_id(void *id_out);
int lidog_comm_init_rank(const void *id, int32_t nranks, int32_t rank, void **comm_out);
int lidog_comm_destroy(void *comm);
int32_t lidog_comm_count(void *comm);   /* ranks of the communicator as RCCL reports them (ncclCommCount); -1 on error */
int lidog_allreduce_f32(float *buf, int64_t n, void *comm, void *stream);
int lidog_allreduce_f64(double *buf, int64_t n, void *comm, void *stream);

/* One-shot peer all-reduce for the small SyncBatchNorm statistics messages (241 messages of
 * <= 4 KB per step, all on the dependent chain): every rank pushes its vector into a mailbox in every peer's memory
 * over its direct xGMI link, waits for the N senders of its own mailbox and adds them in rank order (same bits on
 * every rank) -- one single-workgroup launch on the caller's stream instead of a ring / tree collective.
 * Set-up: every rank allocates a mailbox (lidog_peer_mailbox_alloc: fine-grained device memory + its hipIpc handle of
 * lidog_peer_handle_bytes() bytes), the handles travel through whatever the host has (torch.distributed here), every
 * rank opens the others' (lidog_peer_mailbox_open) and builds the communicator.  Waits are bounded: lidog_peer_status
 * != 0 means a sender's flag never arrived (the caller falls back to lidog_allreduce_f64). */
int32_t lidog_peer_handle_bytes(void);
int64_t lidog_peer_mailbox_bytes(int32_t nranks, int32_t max_doubles);
int lidog_peer_mailbox_alloc(int64_t bytes, void **ptr_out, void *handle_out);
int lidog_peer_mailbox_open(const void *handle, void **ptr_out);
int lidog_peer_comm_create(int32_t rank, int32_t nranks, int32_t max_doubles, void *local, void *const *peer_ptrs,
                           void **comm_out);
int32_t lidog_peer_max_doubles(void *comm);
int lidog_peer_set_spin_limit(void *comm, int64_t polls /* ~1-2 us each; 0 = default, several minutes */);
int lidog_peer_allreduce_f64(void *comm, double *buf, int64_t n, void *stream);
int32_t lidog_peer_status(void *comm);   /* 0 ok, 1 a sender's flag never arrived, 2 a sender was ahead (desynchronised) */
/* Frees the local mailbox and (close_peers != 0) unmaps the peers'.  PRECONDITION the library cannot check: every rank
 * has synchronised its stream and ALL ranks have passed a barrier after their last call -- a peer may otherwise still
 * store into (or poll) this rank's mailbox through its hipIpc mapping when the memory goes away. */
int lidog_peer_comm_destroy(void *comm, int32_t close_peers);
/* A communicator's calls must all be queued on one stream (the mailbox slots are reused in stream order): the first
 * call binds it, a call on another stream is refused; lidog_peer_rebind_stream waits for the old stream and moves it. */
int lidog_peer_rebind_stream(void *comm, void *stream);
int64_t lidog_peer_calls(void *comm);    /* calls made so far (= the sequence number of the last one) */
/* test hook of the failure path: the call with this (1-based) sequence number raises no flags on this rank, so every
 * rank's wait for it runs into the limit, sets the error word and -- from then on -- waits for nothing */
int lidog_peer_inject_skip_flag(void *comm, int64_t seq);
/* frees a mailbox that never became part of a communicator (set-up failed half way) */
int lidog_peer_mailbox_free(void *ptr);
int lidog_peer_mailbox_close(void *peer_ptr);

/* ------------------------------------------------------------------ output-stationary 3^3 convolution (csrc/sconv_os.hip)
 * MinkowskiConvolution(kernel_size=3, stride=1) without product rows: a workgroup owns 128 output rows, walks the
 * kernel offsets in ascending order and adds each offset's product -- the same fmaf chain from zero as a row of
 * lidog_sconv_gemm's T -- to the rows' running sum in the order lidog_sconv_reduce_rows adds them: identical results,
 * no 4 P Cout bytes written and read back (utils/models/minkunet_bev.py:425-439, resnet_block.py:8-56).
 * lidog_kernel_map_sorted: rows sorted by their neighbour mask (rarest offset = most significant bit) so that the rows
 * of a tile share their offsets; perm [pad128(n)] (-1 behind the last row), wave_masks [pad128(n) / 32]: OR of the
 * masks of every 32 sorted rows; tile_order [pad128(n) / 128]: the 128-row tiles in launch order (most work first).
 * nbr / k_off: lidog_kernel_map(_bits) / lidog_kernel_map_pairs of the same map.
 * lidog_sconv_os: out [n, Cout] = sum_k A[nbr[k][row]] W[k] (+ bias) (+ addend); reverse != 0 = the data gradient over
 * the (symmetric) map: offsets from the top, weights W[K-1-k] (pass the transposed kernels, Cin / Cout swapped). */
int64_t lidog_kernel_map_sorted_ws(int64_t n);
int lidog_kernel_map_sorted(const int32_t *nbr, int64_t n, int32_t K, const int64_t *k_off, int32_t *perm,
                            uint32_t *wave_masks, int32_t *tile_order, void *ws, int64_t ws_bytes, void *stream);
int lidog_sconv_os(const float *A, const int32_t *nbr, int64_t n, int32_t K, const int32_t *perm,
                   const uint32_t *wave_masks, const int32_t *tile_order, const float *W, int32_t reverse,
                   const float *bias, const float *addend, int32_t Cin, int32_t Cout, float *out, void *stream);
/* lidog_sconv_os_stats: the forward form with the BatchNorm statistics of its result in the epilogue (one partial row
 * per tile, finished in-kernel; arguments as lidog_sconv_reduce_rows_stats).  ws: lidog_sconv_os_stats_ws(n, Cout) doubles. */
int64_t lidog_sconv_os_stats_ws(int64_t n, int32_t C);
/* lidog_sconv_os_bn: the forward form with an evaluation-mode BatchNorm (+ residual + ReLU) in the epilogue -- the
 * validation path (arguments as lidog_sconv_reduce_rows_bn) */
int lidog_sconv_os_bn(const float *A, const int32_t *nbr, int64_t n, int32_t K, const int32_t *perm,
                      const uint32_t *wave_masks, const int32_t *tile_order, const float *W, const float *bias, int32_t Cin,
                      int32_t Cout, const float *mean, const float *invstd, const float *w, const float *b,
                      const float *residual, int32_t relu, float *out, void *stream);
int lidog_sconv_os_stats(const float *A, const int32_t *nbr, int64_t n, int32_t K, const int32_t *perm,
                         const uint32_t *wave_masks, const int32_t *tile_order, const float *W, const float *bias,
                         int32_t Cin, int32_t Cout, float *out, double *sums, double *ws, double count, float eps,
                         float momentum, float *mean, float *invstd, float *running_mean, float *running_var,
                         void *stream);

/* ------------------------------------------------------------------ BatchNorm + ReLU applied in the consumer's staging
 * conv1 -> BatchNorm -> ReLU -> conv2 inside a BasicBlock (ME modules/resnet_block.py as called from
 * utils/models/minkunet_bev.py:312-371): conv2 is the only reader of the normalised rows, so they need not exist in
 * memory.  The three forms below take the RAW output of conv1 as A plus conv1's BatchNorm vectors in_* [Cin] (batch mean,
 * 1/sqrt(var + eps), weight, bias) and apply ((x - mean) * invstd) * w + b (and max(., 0) when in_relu) to every
 * gathered row on its way into LDS -- lidog_bn_apply_bits' expression and operation order, so each form equals
 * lidog_bn_apply_bits followed by the plain entry point bit for bit.  Matrix-core kernels only (channel counts multiples
 * of 32, lidog_set_sparse_core(1)).  Other arguments as lidog_sconv_gemm / lidog_sconv_os_stats / lidog_sconv_wgrad. */
int lidog_sconv_gemm_in_bn(const float *A, const int32_t *gather, const float *B, const float *bias,
                           const int32_t *tile_k, const int32_t *tile_row0, const int32_t *tile_rows, int32_t n_tiles,
                           int32_t Cin, int32_t Cout, float *T, const int32_t *scatter, const float *in_mean,
                           const float *in_invstd, const float *in_w, const float *in_b, int32_t in_relu, int64_t a_rows,
                           void *stream);
int lidog_sconv_os_stats_in_bn(const float *A, const int32_t *nbr, int64_t n, int32_t K, const int32_t *perm,
                               const uint32_t *wave_masks, const int32_t *tile_order, const float *W, const float *bias,
                               int32_t Cin, int32_t Cout, float *out, double *sums, double *ws, double count, float eps,
                               float momentum, float *mean, float *invstd, float *running_mean, float *running_var,
                               const float *in_mean, const float *in_invstd, const float *in_w, const float *in_b,
                               int32_t in_relu, void *stream);
int lidog_sconv_wgrad_in_bn(const float *A, const int32_t *pair_a, const float *G, const int32_t *pair_g,
                            const int32_t *items, int32_t n_items, const int32_t *item_off, int32_t K, int32_t Cin,
                            int32_t Cout, float *partial, float *gW, const float *in_mean, const float *in_invstd,
                            const float *in_w, const float *in_b, int32_t in_relu, void *stream);

/* ------------------------------------------------------------------ host-side tables of a kernel map (csrc/hostprep.hip)
 * Pure host code.  k_off_host [K+1]: the rule book's offsets (lidog_kernel_map_pairs' k_off copied to the host).
 * lidog_tiles_host: the (tile_k, tile_row0, tile_rows) descriptors lidog_sconv_gemm takes, `tile_rows` (128) pairs per
 * tile, never straddling an offset, in launch order; skip_k >= 0 leaves that offset out.  Written as a contiguous
 * [3][n_tiles] int32 array at `out` (capacity `cap` tiles); returns n_tiles or -1.
 * lidog_wgrad_items_host: the work items lidog_sconv_wgrad takes ([4][n]: k, first pair, end pair, launch order;
 * item_off [K+1]); order_mode 0 = by offset, 1 = same relative position together, 2 = 1 in XCD-sized groups of
 * `group`.  Returns n or -1. */
int64_t lidog_tiles_host(const int64_t *k_off_host, int32_t K, int32_t skip_k, int32_t tile_rows, int32_t *out,
                         int64_t cap);
int64_t lidog_wgrad_items_host(const int64_t *k_off_host, int32_t K, int64_t chunk, int32_t order_mode, int32_t group,
                               int32_t *items, int32_t *item_off, int64_t cap);

/* ------------------------------------------------------------------ trunk executor (csrc/trunk.hip)
 * The whole MinkUNet encoder-decoder -- stem, 4 x (strided convolution + BasicBlocks), 4 x (transposed convolution +
 * ME.cat + BasicBlocks), classifier: utils/models/minkunet_bev.py:302-374, utils/models/minkunet.py:97-158 -- forward
 * and backward in ONE call each.  The reference's forward is ~190 python-level ME calls per pass; the operator path of
 * this library (one entry point per operator) costs ~12 ms of host dispatch per training step.  Here the launch
 * sequence comes from tables: the same entry points above are called in the same order with the same arguments, so
 * results are bit-identical to calling them one by one.
 *
 * Tables (int64 rows, host memory; pointers are device pointers stored as integers):
 *   convs [n_convs][20]: kind (0 = 3^3 stride 1, 1 = 2^3 stride 2, 2 = transposed 2^3 stride 2, 3 = 1x1, 4 = C_in 1
 *         stem through the neighbour table), map row, C_in, C_out, K, W [K,C_in,C_out], Wt [K,C_out,C_in] or 0
 *         (transposed here), gW, bias or 0, g_bias or 0, BatchNorm weight, bias, running_mean, running_var, g_weight,
 *         g_bias (0 for a convolution without BatchNorm), weight-gradient work items, their count, item_off
 *         (lidog_sconv_wgrad), 0;   conv_f [n_convs][2]: BatchNorm eps, momentum
 *   maps  [n_maps][16]: K, n_in, n_out, pairs, pair_in, pair_out, row_ptr / row_list of the output rows, of the input
 *         rows (lidog_kernel_map_rows; 0 where unused), tile descriptors [3][n_tiles], n_tiles, neighbour table
 *         (stem) or 0, identity rows 0..n-1 (1x1) or 0
 *   ops   [n_ops][8]: type (0 = convolution + BatchNorm, 1 = concatenation, 2 = convolution), convolution row, input
 *         buffer, output buffer, ReLU, residual buffer or -1, fold (the gradient already accumulated for the input
 *         buffer -- the residual branch of a BasicBlock -- enters the data gradient's reduction as its addend), second
 *         input of a concatenation
 *   bufs  [n_bufs][4]: level (index into level_rows), channels, external slot or -1 (lives in the arena)
 *   ext / ext_grad [slots]: the caller's tensors: slot 0 = input features (no gradient), the others outputs; incoming
 *         gradients (0 = none)
 * Memory: see the notes on arena / garena / scratch / lane_scratch in csrc/trunk.hip.  dry != 0: nothing is launched,
 * need[] receives the bytes of each region for this batch.  rec [n_ops * 4 + n_bufs]: written by forward, read by
 * backward (arena offsets).  Training-mode BatchNorm.
 *
 * Data parallelism (train_lidog.py:227-231: Lightning DDP + MinkowskiSyncBatchNorm.convert_sync_batchnorm) --
 * dp [12] int64, NULL = single process:
 *   [0] SyncBatchNorm on: every BatchNorm's (sum, sum, rows) message is summed over the ranks between the kernel that
 *       produces it and the kernel that consumes it (forward: statistics -> finalise + apply, one message for conv1 +
 *       downsample of a block; backward: reduce -> apply), IN ORDER ON `stream`
 *   [1] communicator for those messages (lidog_comm_init_rank) or 0
 *   [2] host callback `int cb(int32_t what, int64_t a, int64_t b)` or 0, used where a communicator is 0:
 *       what 0 = all-reduce (sum) of b doubles at device address a, ordered on `stream`; what 1 = gradient bucket a
 *       has all its gradients queued (the callee reduces it)
 *   [3] communicator of the gradient buckets, [4] its hipStream_t, [5] base of the flat fp32 gradient buffer
 *   [6] number of buckets (0 = none), [7] int64 [n][2] element ranges (lo, hi) of the buckets in that buffer,
 *   [8] int32 [n] HOST countdown per bucket (gradients still missing; shared with the caller, who counts the
 *       parameters that are not the trunk's), [9] int32 [n_convs][4] HOST: bucket of (kernel, bias, BatchNorm weight,
 *       BatchNorm bias) of every convolution or -1,
 *   [10] peer all-reduce communicator (lidog_peer_comm_create) or 0: takes the statistics messages that fit its mailbox.
 * A bucket whose countdown reaches 0 inside lidog_trunk_backward is all-reduced (sum) on its stream behind events
 * recorded on `stream` and `lane`; the caller makes its optimiser step wait for that stream. */
int lidog_trunk_forward(const int64_t *convs, const double *conv_f, int32_t n_convs, const int64_t *maps,
                        int32_t n_maps, const int64_t *ops, int32_t n_ops, const int64_t *bufs, int32_t n_bufs,
                        const int64_t *level_rows, const int64_t *ext, void *arena, int64_t arena_bytes,
                        void *scratch, int64_t scratch_bytes, int64_t *rec, int64_t *need /*[2]*/, int32_t dry,
                        const int64_t *dp /*[12] or NULL*/, void *stream, void *lane /* or NULL */);
/* forward, lane: second stream (NULL = everything in line): the 1x1 downsample convolution + BatchNorm of a layer's first
 * block (minkunet_bev.py:414-420) runs on it next to conv1 / conv2 of that block and is joined in front of the residual add
 * (local BatchNorm only; same results).  Its scratch comes from the arena, so dry and real calls must pass the same lane.
 * backward, lane: second stream for the weight gradients (NULL = in line), forked behind each data gradient's GEMM
 * (wgrad_first = 0) or before it; joined into `stream` before the call returns. */
int lidog_trunk_backward(const int64_t *convs, const double *conv_f, int32_t n_convs, const int64_t *maps,
                         int32_t n_maps, const int64_t *ops, int32_t n_ops, const int64_t *bufs, int32_t n_bufs,
                         const int64_t *level_rows, const int64_t *ext, const int64_t *ext_grad, void *arena,
                         const int64_t *rec, void *garena, int64_t garena_bytes, void *scratch,
                         int64_t scratch_bytes, void *lane_scratch, int64_t lane_bytes, int64_t *need /*[3]*/,
                         int32_t *conv_done_host /*[n_convs]: 1 = this convolution's parameter gradients were written*/,
                         int32_t dry, int32_t wgrad_first, const int64_t *dp /*[12] or NULL*/, void *stream,
                         void *lane);
/* Fusions the executor applies on top of the operator path's launch sequence (bit mask; results are bit-identical
 * either way): 1 = BatchNorm-backward statistics in the epilogue of the producing data-gradient reduction
 * (lidog_sconv_reduce_rows_bwdstats); 2 = the ReLU masks of BatchNorm + residual + ReLU layers kept as bits
 * (lidog_bn_apply_bits); 4 = the BatchNorm + ReLU between the two convolutions of a block applied in the second one's
 * staging (lidog_sconv_*_in_bn) instead of by a pass of its own.  mask >= 0 sets it; returns the previous mask.  Set it between passes, not between a forward
 * pass and its backward pass. */
int32_t lidog_trunk_fusions(int32_t mask);
/* readers [n_ops]: which op (a 3^3 convolution + BatchNorm) applies op o's BatchNorm + ReLU in its staging under fusion 4,
 * -1 where the BatchNorm keeps its own pass -- conv1 of every BasicBlock in MinkUNet34.  Host only (tables as above). */
int lidog_trunk_in_bn_readers(const int64_t *convs, int32_t n_convs, const int64_t *maps, int32_t n_maps,
                              const int64_t *ops, int32_t n_ops, const int64_t *bufs, int32_t n_bufs, int32_t *readers);
/* Timing of the executor's gathered-GEMM launches for the roofline figure of bench.py: on != 0 brackets every such
 * launch with HIP events on its stream; _read waits for the recorded launches, returns (launches, total ms, algorithmic
 * FLOPs, algorithmic bytes: SURVEY.md 8(d)) in out[0..3] and forgets them.  One timing client per process. */
int lidog_trunk_gemm_timing(int32_t on);
int lidog_trunk_gemm_timing_read(double *out /*[4]*/);
/* algorithmic bytes of the executor's MFMA weight-gradient and per-row reduction launches while the timing is on:
 * out[0] launches / out[1] bytes (weight gradient), out[2] / out[3] (reduction), out[4] / out[5] (output-stationary
 * convolution); reading resets the counters */
int lidog_trunk_work_read(double *out /*[6]*/);

#ifdef __cplusplus
}
#endif
#endif /* LIDOG_AMD_H */
