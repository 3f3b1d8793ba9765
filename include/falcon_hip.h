/*
 * falcon_hip.h -- C ABI of libfalcon_hip.so: falcon's vectorise -> ANN -> DBSCAN hot
 * path as hand-written HIP kernels for gfx950 (MI355X).
 *
 * The reference (bittremieux/falcon @ 2024-12-23) is pure Python and has no FFI; the seam
 * this library sits behind is the one call `cluster.generate_clusters(...)`
 * (reference falcon/cluster/cluster.py:24-156, call site falcon/falcon.py:178-188).
 * Each entry point below names the reference code (file:line) whose work it does.
 * INTEGRATION.md shows the ctypes binding a falcon maintainer would add.
 *
 * Conventions
 *   - every function returns 0 on success or a negative FAL_E* code and never throws or
 *     aborts across the boundary; fal_last_error() returns a thread-local message;
 *   - plain C types only; pointers marked [dev] are device (HBM) pointers valid on the
 *     context's device (e.g. torch tensor .data_ptr()), [host] are host pointers;
 *   - the CALLER allocates every input/output array; the library owns only the opaque
 *     handles (fal_ctx, fal_ivf) and grow-only scratch inside the context;
 *   - all device work is enqueued on the context's stream; outputs are valid after
 *     fal_ctx_sync() (or any later call on the same context that reads them);
 *   - one fal_ctx per device, not shared between threads;
 *   - there is NO CPU fallback: a context can only be created on a HIP device.
 */
#ifndef FALCON_HIP_H
#define FALCON_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FAL_OK            0
#define FAL_EINVAL       -1   /* bad argument */
#define FAL_EHIP         -2   /* a HIP runtime call failed */
#define FAL_ENOMEM       -3   /* device allocation failed */
#define FAL_ENODEV       -4   /* no usable gfx950 device */
#define FAL_EUNSUPPORTED -5   /* parameter outside the compiled limits */
#define FAL_EINTERNAL    -6   /* a library invariant failed (only raised under FALCON_DEBUG_POISON=1) */

#define FAL_DTYPE_F32 0
#define FAL_DTYPE_F16 1
#define FAL_DTYPE_SPLIT16 2   /* per row: low_dim f16 'hi' = f16(x), then low_dim f16 'lo' = f16((x - hi) * 2048) */
#define FAL_OUT_F32_F16   3   /* fal_vectorize_rows: float32 rows to `out` + their float16 rounding to `out2` (= fal_vectorize_pair) */
#define FAL_OUT_F16_IMAGE 4   /* fal_vectorize_rows: float16 VECTORS to `out2` + their float32 image to `out` (= fal_vectorize_f16_image) */

#define FAL_MAX_LOW_DIM   1024     /* multiple of 8 */
#define FAL_MAX_K_ANN      256
#define FAL_MAX_N_PROBE    128
#define FAL_MAX_N_LIST  131072

typedef struct fal_ctx fal_ctx;
typedef struct fal_ivf fal_ivf;

/* ---- library / context ------------------------------------------------------------- */
int         fal_version(void);
const char* fal_last_error(void);
int fal_device_count(int* count);
/* own_stream != 0: the library creates (and later destroys) a private stream and `stream`
 * is ignored.  own_stream == 0: the context enqueues on the caller's hipStream_t `stream`
 * (e.g. torch.cuda.current_stream().cuda_stream; NULL is the legacy default stream). */
int fal_ctx_create(int device, void* stream, int own_stream, fal_ctx** out);
int fal_ctx_destroy(fal_ctx* ctx);
int fal_ctx_sync(fal_ctx* ctx);
/* The one-shot call pattern of the reference (falcon.py:153-193: generate_clusters ONCE per precursor charge, in a fresh
 * process) makes the FIRST pass the only one.  fal_ctx_plan: before it, load the code objects of every kernel on the context's
 * device (otherwise loaded one translation unit at a time between the first pass's kernels) and size the scratch slots that follow
 * from the shape alone (n spectra, k_ann, n_probe, batch_size); a host-side call, safe at any time, never shrinks anything.
 * fal_ctx_trim: return the context's cached device memory (scratch slots, free pool blocks) to the driver -- between jobs of very
 * different sizes; drains the stream. */
int fal_ctx_plan(fal_ctx* ctx, int64_t n, int low_dim, int k_ann, int n_probe, int64_t batch_size);
int fal_ctx_trim(fal_ctx* ctx);
/* Elapsed milliseconds (HIP events on the context's stream) of the kernels the LAST
 * call of the named stage enqueued; used by bench.py for the roofline figure.
 * stage: 0 vectorize, 1 kmeans/ivf build, 2 coarse probe, 3 fine scan (cosine kernel),
 * 4 top-k select, 5 filter, 6 dbscan, 7 tail, 8 the launches of the cosine kernel alone (dense4_kernel / dense_kernel /
 * scan16_kernel / list16_kernel / ivf_list4_kernel; a subset of stage 3: the exact pair chains and the one-block
 * buckets' dense_tiny4_kernel are not in it). */
int fal_ctx_stage_ms(fal_ctx* ctx, int stage, float* ms, int64_t* launches);
int fal_ctx_enable_timing(fal_ctx* ctx, int on);
/* Work counters of the LAST fal_ivf_search_topk on this context (for roofline accounting):
 * which = 0: (query, candidate) inner products the fine scan produced results for
 *            (= sum over queries of candidates in their probed lists; n_b^2 per flat bucket);
 * which = 1: (query, centroid) inner products of the coarse quantiser;
 * which = 2: number of scan kernel launches (batches);  3: bytes of the sims scratch buffer;
 * which = 4: inner products the matrix cores actually computed for the flat buckets (tile padding
 *            included; the fp32 kernel computes only the blocks on/above each bucket's diagonal);
 * which = 5: queries of the last prefiltered search that took the exact fallback (after a sync);
 * which = 6: precondition check of the float16 prefilters for the LAST index built on this context: bit 0 = rows of
 *            an indexed bucket, bit 1 = rows handed to the flat prefilter hold negative / non-finite components, so
 *            that index is built / searched with the exact kernels (see fal_ivf_build_x16);
 * which = 7, 8: the part of counters 0 and 4 that belongs to flat buckets of up to 32 rows, whose kernel is not
 *            timed under stage 8 (subtract them to price stage 8's launches). */
int fal_ctx_counter(fal_ctx* ctx, int which, int64_t* value);

/* ---- a1  bin geometry: reference spectrum.py:172-199 `get_dim` (float32) --- [host] */
int fal_get_dim(float min_mz, float max_mz, float bin_size,
                uint32_t* dim, float* start_dim, float* end_dim);

/* ---- a3  feature-hash lookup table: MurmurHash3_x86_32(int32 bin, seed) % low_dim;
 *          spec reference README.md:124-131 ------------------------------------ [host] */
int fal_hash_lookup(uint32_t n_bins, uint32_t low_dim, uint32_t seed, uint32_t* out_table);

/* ---- a2  m/z -> bin index (the CSR `indices` of reference spectrum.py:250-296
 *          `_to_vector`): floor(((double)mz - min_mz) / bin_size) as int32 ------ [dev] */
int fal_to_vector_indices(fal_ctx* ctx, const float* mz, int64_t nnz,
                          double min_mz, double bin_size, int32_t* out_indices);

/* ---- a2+a3  CSR peaks -> dense low_dim vectors, L2-normalised: reference
 *          spectrum.py:202-247 `to_vector` with the projection realised as feature
 *          hashing.  Row r of `out` is spectrum row_order[r] (row_order may be NULL).
 *          out: [n, low_dim] float32 or float16, or [n, 2, low_dim] float16 hi/lo planes of the
 *          float32 result (out_dtype, FAL_DTYPE_*).  Peaks whose bin lies outside [0, n_bins)
 *          are ignored; an all-zero row stays zero. ------------------------------- [dev] */
int fal_vectorize(fal_ctx* ctx, const float* mz, const float* intensity,
                  const int64_t* indptr, const int64_t* row_order, int64_t n,
                  double min_mz, double bin_size, uint32_t n_bins,
                  uint32_t low_dim, uint32_t seed, int normalize, int out_dtype, void* out);
/* The same with TWO outputs from one pass over the peaks: the float32 rows and their float16 rounding (the copy the
 * float16 prefilters of the index build and of the search read; same values as fal_vectorize with FAL_DTYPE_F16). [dev] */
int fal_vectorize_pair(fal_ctx* ctx, const float* mz, const float* intensity, const int64_t* indptr,
                       const int64_t* row_order, int64_t n, double min_mz, double bin_size, uint32_t n_bins,
                       uint32_t low_dim, uint32_t seed, int normalize, float* out_f32, void* out_f16);
/* float16 VECTORS (BASELINE configs[4]: low_dim 800 fp16): out_f16 = the rows rounded to float16 (= fal_vectorize with
 * FAL_DTYPE_F16), out_f32_image = the float32 image of those rounded values.  The similarity of two float16 vectors is
 * defined as the k-ordered float32 fmaf chain over their images (products of float16 values are exact in float32), so an
 * index built on the image with out_f16 as its prefilter copy (fal_ivf_build_x16 / fal_ivf_attach_prefilter_ex) searches
 * float16 vectors exactly and reproducibly; the oracle restates it as "round to float16, then the float32 path". [dev] */
int fal_vectorize_f16_image(fal_ctx* ctx, const float* mz, const float* intensity, const int64_t* indptr,
                            const int64_t* row_order, int64_t n, double min_mz, double bin_size, uint32_t n_bins,
                            uint32_t low_dim, uint32_t seed, int normalize, float* out_f32_image, void* out_f16);

/* `--low_dim` is a free integer in the reference (README.md:114-117, config.py: "low_dim" of the nearest-neighbour options);
 * the cosine kernels are instantiated for rows of 64, 128, 256, 400 and 800 columns.  fal_row_width: the row width the path
 * stores `low_dim`-dimensional vectors in = the smallest of those widths >= low_dim (FAL_EUNSUPPORTED beyond 800) -- [host].
 * fal_vectorize_rows: fal_vectorize / _pair / _f16_image (out_mode = FAL_DTYPE_* / FAL_OUT_*) with the hash taken modulo
 * `low_dim` (any integer in [1, row_width]) and rows of `row_width` columns (a multiple of 8) whose columns >= low_dim are
 * zero.  A zero column adds fma(0 * 0, acc) = acc to every k-ordered inner-product chain, so every kernel downstream runs at
 * d = row_width unchanged and the similarities are those of the low_dim-dimensional vectors summed in the order of the
 * row_width-wide chain (the oracle pads the same way). ---------------------------------------- [dev] */
int fal_row_width(uint32_t low_dim, uint32_t* row_width);
int fal_vectorize_rows(fal_ctx* ctx, const float* mz, const float* intensity, const int64_t* indptr,
                       const int64_t* row_order, int64_t n, double min_mz, double bin_size, uint32_t n_bins,
                       uint32_t low_dim, uint32_t row_width, uint32_t seed, int normalize, int out_mode,
                       void* out, void* out2 /*second output of FAL_OUT_*, else NULL*/);

/* ---- a5  precursor-m/z bucket boundaries: reference cluster.py:159-209
 *          `_get_precursor_mz_splits` over the m/z-SORTED float32 precursor array,
 *          plus (flags) the build's two extra rules: chunk the last block
 *          too, and cut at fixed windows floor(mz / mz_interval).
 *          splits_out [host] receives at most max_splits int64 boundaries
 *          (first 0, last n); *n_splits = how many. -------------------------------- */
int fal_precursor_splits(fal_ctx* ctx, const float* precursor_mz_sorted /*[dev]*/, int64_t n,
                         double tol, int tol_is_da, int64_t batch_size,
                         double mz_interval, int chunk_last,
                         int64_t* splits_out /*[host]*/, int64_t max_splits, int64_t* n_splits);

/* ---- e   the multi-GPU front end: whole precursor windows floor(mz / mz_interval) are the unit dealt to GPUs (buckets never
 *          cross a window and a window's buckets depend on its own spectra only; reference analogue: blocks are
 *          clustered independently, cluster.py:107-141).  fal_window_counts: all n_parts partitions (precursor charges) of a
 *          job in one call -- precursor_mz[p] [host array of device pointers], n[p] [host] spectra of partition p,
 *          counts[p * n_windows + w] [dev] = spectra of partition p in window w (windows >= n_windows - 1 share the last
 *          counter; the call zeroes counts first); counts_host (optional, pinned host memory, n_parts * (n_windows + 2)
 *          int32): the same table, then (first, last) occupied window of every partition (INT32_MAX, 0 for an empty
 *          one), copied stream-ordered (valid after fal_ctx_sync).  fal_window_select: the spectra whose
 *          window belongs to `rank` (owner[w], int32 per window) -- rows_out i64 ascending dataset rows, mz_out their
 *          precursor m/z; *count [host] how many (synchronises).  Buffers need room for n entries. ------------- [dev] */
int fal_window_counts(fal_ctx* ctx, const float* const* precursor_mz /*[host]*/, const int64_t* n /*[host]*/, int n_parts,
                      double mz_interval, int64_t n_windows, int32_t* counts, int32_t* counts_host /*[pinned host] or NULL*/);
int fal_window_select(fal_ctx* ctx, const float* precursor_mz, int64_t n, double mz_interval, int64_t n_windows,
                      const int32_t* owner, int rank, int64_t* rows_out, float* mz_out, int64_t* count /*[host]*/);

/* ---- a6  IVF build per bucket (k-means + inverted lists); the reference's only
 *          statement is README.md:134-136 (Faiss IndexIVFFlat, un-vendored dep
 *          setup.cfg:25).  X [dev] is [n, low_dim] float32 in precursor-sorted row order;
 *          bucket_off [host] has n_buckets+1 row offsets; n_list [host] lists per bucket
 *          (1 = flat).  Deterministic: init rows floor(i*n_b/n_list), `kmeans_iters`
 *          x (argmax-IP assign, spherical mean in row order), final assign. ---------- */
int fal_ivf_build(fal_ctx* ctx, const float* X, int64_t n, int low_dim,
                  const int64_t* bucket_off, int64_t n_buckets, const int32_t* n_list,
                  int kmeans_iters, fal_ivf** out);
/* The same build with float16 copies [n, low_dim] of X (fal_vectorize FAL_DTYPE_F16 on the same peaks) as a PREFILTER of
 * the k-means / final assignment of buckets with <= 512 lists: the arg-max runs on the f16 matrix cores and only the
 * rows whose two best centroids are closer than the float16 error bound are re-evaluated exactly in float32 -- every
 * assignment, centroid and list is identical to fal_ivf_build's (low_dim in {64, 128, 256, 400}; X16 NULL = fal_ivf_build).
 * PRECONDITION of every float16 prefilter of this library (here and fal_ivf_attach_prefilter[_ex]): X16 is the float16
 * rounding of X, and no component of X is negative, infinite or NaN -- the error bound that makes the prefiltered results
 * exact, |f16-MFMA(x.y) - fp32 chain(x.y)| <= 1.3e-3 (x.y) + 2e-6, is relative to the similarity and holds for
 * non-negative rows only (hashed spectra are: intensities >= 0).  The precondition is CHECKED, not assumed: the build's
 * pass over the rows of the indexed buckets looks at every component (one stream synchronisation when X16 is given), and
 * an index whose rows fail it is built and searched with the exact float32 kernels -- same results, no error. */
int fal_ivf_build_x16(fal_ctx* ctx, const float* X, const void* X16, int64_t n, int low_dim,
                      const int64_t* bucket_off, int64_t n_buckets, const int32_t* n_list,
                      int kmeans_iters, fal_ivf** out);
/* Optional: float16 copies of the vectors, [n, planes, low_dim] in the same (sorted) row order, that
 * the FLAT buckets are then scanned with on the f16 matrix cores: planes = 1 plain float16 rows
 * (fal_vectorize FAL_DTYPE_F16; BASELINE config 5), planes = 2 the hi/lo split of the float32
 * rows (FAL_DTYPE_SPLIT16; float32-accurate to ~3e-7).  With planes = 1 fal_ivf_build may be given
 * X = NULL as long as every bucket is flat.  The buffer is borrowed. ------------------- [dev] */
int fal_ivf_attach_f16(fal_ivf* ivf, const void* X16, int planes);
/* Optional: float16 copies [n, low_dim] of the float32 vectors (fal_vectorize FAL_DTYPE_F16 on the same
 * peaks) used ONLY as a PREFILTER by fal_ivf_search_neighbors on flat buckets: the bucket is scanned on the
 * f16 matrix cores to bracket every query's n_neighbors_ann-th best similarity, the candidates inside the
 * precursor window are then evaluated exactly in float32 and the bracket is resolved exactly where it
 * matters -- the neighbour lists are BIT-IDENTICAL to the ones computed without the prefilter, the
 * [n, candidates] similarity matrix never exists in HBM.  low_dim in {64, 128, 256, 400}; other sizes
 * ignore the prefilter.  fal_ivf_search_topk never uses it.  The buffer is borrowed.  Precondition as stated at
 * fal_ivf_build_x16 (non-negative finite components); the call checks it with one pass over X16 and one stream
 * synchronisation, and rows that fail it make the searches ignore the prefilter (exact staged scan). ---- [dev] */
int fal_ivf_attach_prefilter(fal_ivf* ivf, const void* X16);
/* The same with a choice of where the prefilter is used: which & 1 = flat buckets (as above), which & 2 = buckets
 * with an index: their fine scan runs on the f16 matrix cores over X16 itself (gathered through the index's row
 * permutation; the buffer is borrowed and must outlive the index for this part too), the k-th best key
 * of every query is bracketed from 16-bit keys, and the exact float32 work is limited to the precursor window
 * and to the candidates that can decide the k-th key.  BIT-IDENTICAL neighbour lists under the same precondition
 * (the build has already looked at the rows of the indexed buckets: if any has a negative / non-finite component
 * no float16 copy is made and the searches run the exact fine scan).  Reference: the n_probe query of README.md:107-113
 * (faiss IndexIVFFlat.search, an un-vendored dependency: setup.cfg:25). ------------------------------------ [dev] */
int fal_ivf_attach_prefilter_ex(fal_ivf* ivf, const void* X16, int which);
int fal_ivf_destroy(fal_ivf* ivf);
int fal_ivf_total_lists(const fal_ivf* ivf, int64_t* total_lists);
/* Copy the index out for inspection (any pointer may be NULL): centroids
 * [total_lists, low_dim] f32, list id of every row (bucket-local) i32[n], perm i32[n]
 * (row ids in (bucket, list, row) order), list_off i64[total_lists+1]. --------- [dev] */
int fal_ivf_export(fal_ctx* ctx, const fal_ivf* ivf, float* centroids, int32_t* assign,
                   int32_t* perm, int64_t* list_off);

/* ---- a7  n_probe query of every row against its bucket's index + top-k_ann by
 *          (inner product desc, row id asc); reference spec README.md:107-113,137-142.
 *          sim f32[n,k_ann] (pad -inf), idx i32[n,k_ann] (pad -1, ids = sorted rows). */
int fal_ivf_search_topk(fal_ctx* ctx, const fal_ivf* ivf, int n_probe, int k_ann,
                        float* sim, int32_t* idx);

/* ---- a8  neighbour filter + distance: drop self / pads / neighbours outside the
 *          precursor (and RT) tolerance, keep the first n_neighbors,
 *          dist = clip(1 - sim, 0, 1): reference cluster.py:190-195 (mass_diff usage),
 *          cluster.py:626, similarity.py:78.  rt may be NULL / rt_tol < 0 = no RT filter.
 *          nb_idx i32[n,k] (pad -1), nb_dist f32[n,k] (pad +inf). ---------------- [dev] */
int fal_filter_neighbors(fal_ctx* ctx, const float* sim, const int32_t* idx, int64_t n,
                         int k_ann, const float* precursor_mz_sorted, const float* rt_sorted,
                         double tol, int tol_is_da, double rt_tol, int n_neighbors,
                         int32_t* nb_idx, float* nb_dist);

/* ---- a7+a8 in one call: the same search with the neighbour filter applied to the k_ann
 *          selected candidates inside the selection kernel (only the survivors are sorted;
 *          the [n, k_ann] result never goes to HBM).  Output identical to
 *          fal_ivf_search_topk followed by fal_filter_neighbors; nb_count (optional)
 *          receives the number of stored neighbours of every row. ------------------ [dev] */
int fal_ivf_search_neighbors(fal_ctx* ctx, const fal_ivf* ivf, int n_probe, int k_ann,
                             const float* precursor_mz_sorted, const float* rt_sorted,
                             double tol, int tol_is_da, double rt_tol, int n_neighbors,
                             int32_t* nb_idx, float* nb_dist, int32_t* nb_count /*[n] or NULL*/);

/* ---- f4  exact re-scoring of the stored neighbours with the matched-peak cosine the
 *          reference ships: similarity.py:17-80 `cosine_fast` (pairs of peaks within
 *          fragment_tol, optimal assignment, sum of the positive pair scores), used as
 *          cluster.py:593-639 does: nb_dist = 1 - sim, sim = 0 when fewer than min_matches
 *          peaks match.  In place on nb_dist; entry order is not changed.  mz / intensity /
 *          indptr: the preprocessed spectra (dataset rows), row_order: sorted position ->
 *          dataset row.  Synchronises (FAL_EUNSUPPORTED if more than 32 peaks of one
 *          spectrum chain inside the tolerance). ------------------------------- [dev] */
int fal_rescore_neighbors(fal_ctx* ctx, const int32_t* nb_idx, float* nb_dist, int64_t n, int k,
                          const float* mz, const float* intensity, const int64_t* indptr,
                          const int64_t* row_order, double fragment_tol, int min_matches);

/* ---- e   neighbour lists ELL -> CSR, ids shifted by id_offset to global rows: the
 *          payload of the one multi-GPU exchange step (all-gatherv of the
 *          sparse neighbour lists; the reference's per-block results are likewise
 *          concatenated with offsets, cluster.py:115-141).  Entry order within a row is
 *          kept.  Segments (e.g. the charge partitions of falcon.py:151-160) chain on
 *          the device: a call writes rows [row0, row0 + n) of indptr_out and continues
 *          at nnz = indptr_out[row0] (row0 = 0 starts a new graph).  indptr i64[rows+1];
 *          idx_out / dist_out need room for every segment's n*k entries. ---------- [dev] */
int fal_neighbors_to_csr(fal_ctx* ctx, const int32_t* nb_idx, const float* nb_dist,
                         const int32_t* nb_count /*[n] lengths of front-packed rows, or NULL*/,
                         int64_t n, int k, int64_t id_offset, int64_t row0, int64_t* indptr_out,
                         int32_t* idx_out, float* dist_out);
/*          Same with an id map: stored id -> id_map[id] + id_offset.  A rank that ran the path on a
 *          SUBSET of a dataset's buckets (its share of one precursor-sorted dataset) maps
 *          the subset positions back to dataset rows here (id_map = the subset's row order). -- [dev] */
int fal_neighbors_to_csr_mapped(fal_ctx* ctx, const int32_t* nb_idx, const float* nb_dist,
                                const int32_t* nb_count, int64_t n, int k,
                                const int64_t* id_map /*[ids] or NULL*/, int64_t id_offset, int64_t row0,
                                int64_t* indptr_out, int32_t* idx_out, float* dist_out);

/* ---- a9  DBSCAN(eps, min_samples = 2 as reference cluster.py:66) on the sparse
 *          neighbour graph; spec README.md:143-146.  Order-independent form:
 *          clusters = components of core points, border -> lowest-index core
 *          in-neighbour, clusters numbered by lowest core row, noise = -1. -------- [dev] */
int fal_dbscan(fal_ctx* ctx, const int32_t* nb_idx, const float* nb_dist, int64_t n, int k,
               float eps, int32_t* labels, int64_t* n_clusters /*[host]*/);

/* ---- a10 precursor / RT refinement of every DBSCAN cluster: reference
 *          cluster.py:362-455 `_postprocess_cluster` + cluster.py:458-509 `_linkage`
 *          (1-D complete linkage cut at the tolerance; groups < 2 -> noise);
 *          final clusters numbered in (DBSCAN label, first member) order like
 *          cluster.py:293-313.  labels in/out i32[n] (sorted-row space). ---------- [dev] */
int fal_refine_clusters(fal_ctx* ctx, int32_t* labels, int64_t n,
                        const float* precursor_mz_sorted, const float* rt_sorted,
                        double tol, int tol_is_da, double rt_tol,
                        int64_t* n_clusters /*[host]*/);

/* ---- a11+a12 medoids (reference cluster.py:512-553 on the sparse graph) and label
 *          globalisation (cluster.py:556-590, 144-155): labels_sorted -> labels by DATASET
 *          row with noise renumbered n_clusters.. in dataset-row order; medoids[c] =
 *          dataset row of cluster c's medoid (c < n_clusters), then the noise rows. [dev] */
int fal_finalize(fal_ctx* ctx, const int32_t* labels_sorted, int64_t n, int64_t n_clusters,
                 const int64_t* row_order, const int32_t* nb_idx, const float* nb_dist, int k,
                 int32_t* labels_out, int32_t* medoids_out, int64_t* n_labels /*[host]*/);

/* ---- a9 + a10 + a11 + a12 in one call (same kernels; all intermediate counts stay on the
 *          device, a single host synchronisation at the end).  labels_sorted_scratch i32[n]
 *          receives the refined labels in sorted-row space (-1 = noise). ------------- [dev] */
int fal_cluster_graph(fal_ctx* ctx, const int32_t* nb_idx, const float* nb_dist, int64_t n, int k,
                      float eps, const float* precursor_mz_sorted, const float* rt_sorted,
                      double tol, int tol_is_da, double rt_tol, const int64_t* row_order,
                      int32_t* labels_sorted_scratch, int32_t* labels_out, int32_t* medoids_out,
                      int64_t* n_clusters /*[host]*/, int64_t* n_labels /*[host]*/);

/* The same for FRONT-PACKED neighbour rows of known length (what fal_ivf_search_neighbors leaves: nb_count[i] stored
 * neighbours in slots 0 .. nb_count[i] - 1, padding behind): the graph passes read the stored slots only. --- [dev] */
int fal_cluster_graph_counted(fal_ctx* ctx, const int32_t* nb_idx, const float* nb_dist, const int32_t* nb_count,
                              int64_t n, int k, float eps, const float* precursor_mz_sorted, const float* rt_sorted,
                              double tol, int tol_is_da, double rt_tol, const int64_t* row_order,
                              int32_t* labels_sorted_scratch, int32_t* labels_out, int32_t* medoids_out,
                              int64_t* n_clusters /*[host]*/, int64_t* n_labels /*[host]*/);

/* ---- f4  hierarchical clustering of the neighbour graph: the clustering the reference snapshot ships,
 *          fcluster(fastcluster.linkage(pdist, linkage), distance_threshold, "distance") (cluster.py:283-290), on the
 *          sparse graph with "missing pair = distance 1" (cluster.py:621-626), normally after fal_rescore_neighbors
 *          put the exact matched-peak distances into it.  method: 0 single, 1 complete, 2 average; threshold < 1.
 *          Same output contract as fal_dbscan (clusters numbered by lowest row, groups of one row = -1), so
 *          fal_refine_clusters / fal_finalize follow unchanged; fal_cluster_graph_linkage is the fused a9..a12
 *          form.  Tie order between equal merge heights is the build's own (PARITY UNPINNED: fastcluster absent).
 *          The complete / average forms synchronise the stream once more than fal_dbscan does (the sizes of the connected
 *          groups decide their scratch) and agglomerate one group per wave in O(m^3 / 64): a connected group of more than
 *          2,048 rows within the threshold is refused with FAL_EUNSUPPORTED (single linkage and DBSCAN have no such limit). [dev] */
int fal_linkage_cluster(fal_ctx* ctx, const int32_t* nb_idx, const float* nb_dist, int64_t n, int k,
                        float threshold, int method, int32_t* labels, int64_t* n_clusters /*[host]*/);
int fal_cluster_graph_linkage(fal_ctx* ctx, const int32_t* nb_idx, const float* nb_dist, int64_t n, int k,
                              float threshold, int method, const float* precursor_mz_sorted,
                              const float* rt_sorted, double tol, int tol_is_da, double rt_tol,
                              const int64_t* row_order, int32_t* labels_sorted_scratch, int32_t* labels_out,
                              int32_t* medoids_out, int64_t* n_clusters /*[host]*/, int64_t* n_labels /*[host]*/);

/* ---- f1  spectrum preprocessing, the step in front of the path: reference
 *          spectrum.py:73-169 `process_spectrum` over a CSR of raw peaks (m/z float64
 *          sorted per spectrum, intensity float32): m/z range cut (135), precursor-peak
 *          removal for charge states z..1 (139-149; charge 0 = unknown -> 1), base-peak
 *          intensity filter + the max_peaks_used most intense (151-155), validity check
 *          after every step (27-52), scaling 0 none / 1 root / 2 log / 3 rank (157), L2
 *          normalisation (55-70, 158).  Unset options: mz_min / mz_max = NaN,
 *          remove_precursor_tol < 0, min_intensity < 0, max_peaks_used = 0.
 *          valid_out i32[n]; out_indptr i64[n+1] (invalid spectra hold 0 peaks);
 *          out_mz / out_intensity f32 with room for nnz peaks. ---------------- [dev] */
int fal_process_spectra(fal_ctx* ctx, const double* mz, const float* intensity,
                        const int64_t* indptr, int64_t n, int64_t nnz,
                        const double* precursor_mz, const int32_t* precursor_charge,
                        int min_peaks, double min_mz_range, double mz_min, double mz_max,
                        double remove_precursor_tol, double min_intensity, int max_peaks_used,
                        int scaling, int32_t* valid_out, int64_t* out_indptr, float* out_mz,
                        float* out_intensity);

/* ---- sort by precursor m/z (reference cluster.py:73-85 `.sort_values`): stable.
 *          order_out i64[n] (dataset row of sorted position), mz_sorted_out f32[n]. [dev] */
int fal_sort_by_precursor(fal_ctx* ctx, const float* precursor_mz, int64_t n,
                          int64_t* order_out, float* mz_sorted_out);
/* out[i] = src[order[i]] for float arrays (retention times). -------------------- [dev] */
int fal_gather_f32(fal_ctx* ctx, const float* src, const int64_t* order, int64_t n, float* out);

#ifdef __cplusplus
}
#endif
#endif /* FALCON_HIP_H */
