/*
 * gulon_hip.h -- C ABI of libgulon_hip.so: the MI355X (gfx950) implementation of
 * tixxit/gulon's ANN hot path (KMeans train/assign, ProductQuantizer encode +
 * distance tables, Index.query ADC scan + top-k, partial top-k merge).
 *
 * The reference has NO native boundary (it is 100 % Scala); every entry point
 * below replaces the body of one public Scala method and cites it.  Paths are
 * relative to /root/reference/core/src/main/scala/net/tixxit/gulon/.  The JNI /
 * Scala stubs that bind these are shown in INTEGRATION.md.
 *
 * Conventions
 *  - plain pointers and sizes only; the caller owns every host buffer for the
 *    duration of a call; handles own device memory until *_destroy.
 *  - matrices are flat row-major float32 (Matrix.data flattened, ld = cols).
 *  - codebooks are ONE flat array of k*d floats: quantizer j's k x s_j block
 *    starts at cents + k*from_j (from_j, s_j from gulon_subvectors).
 *  - every function returns a status: 0 ok, <0 error class (below);
 *    gulon_last_error() gives the thread-local message.  The JNI glue maps
 *    INVALID_ARGUMENT -> IllegalArgumentException (the reference's `require`),
 *    ILLEGAL_STATE -> IllegalStateException, the rest -> RuntimeException.
 *  - all arithmetic is IEEE binary32, unfused, in the reference's evaluation
 *    order; results are bit-identical to the reference algorithm (see DESIGN.md
 *    for the two documented tie rules).
 *  - "_dev" variants take DEVICE pointers and a hipStream_t (as void*), do not
 *    synchronise, and are what bench.py times (inputs resident in HBM).
 */
#ifndef GULON_HIP_H
#define GULON_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GULON_ABI_VERSION 3

#define GULON_OK 0
#define GULON_ERR_INVALID_ARGUMENT (-1) /* reference: require(...) / IllegalArgumentException */
#define GULON_ERR_ILLEGAL_STATE (-2)    /* reference: IllegalStateException("heap is empty") */
#define GULON_ERR_UNSUPPORTED (-3)      /* valid in the reference, not yet on this path */
#define GULON_ERR_DEVICE (-4)           /* HIP runtime failure */
#define GULON_ERR_OOM (-5)

/* flags written per query by the top-k kernels (see DESIGN.md "tie rule") */
#define GULON_FLAG_BOUNDARY_TIE 1 /* K-th and (K+1)-th smallest distances are equal */
#define GULON_FLAG_INTERIOR_TIE 2 /* two equal distances inside the top K */
#define GULON_FLAG_EXACT_REPLAY 4 /* tie resolved by replaying the reference heap's insertion history:
                                     ids and order are exactly TopKHeap's (single, unsharded index only) */

#define GULON_FLAG_NONFINITE 8    /* the query's distances can be NaN / +inf (NaN or huge query components, NaN
                                     centroids): result = the literal TopKHeap over all rows, TopKHeap.scala:69-79 */

#define GULON_MAX_K 63 /* neighbours per query held by one wavefront list: fast path, exact tie replay,
                          sharded merge */
#define GULON_MAX_K_PEELED 8191 /* larger k_nn (flat index, sharded or not; exact kNN): the result is peeled 64
                                   entries per scan (a shard returns k_nn + 1 <= 8191 entries); ties keep the
                                   (distance, row id) order + flags */

typedef struct gulon_dataset gulon_dataset; /* device-resident Matrix            */
typedef struct gulon_index gulon_index;     /* device-resident PQIndex (codes+PQ) */
typedef struct gulon_sharded_index gulon_sharded_index; /* PQIndex row-sharded over the GPUs of one node */

/* One KMeans.ProgressReport (KMeans.scala:119-127) as plain numbers. */
typedef struct {
  int32_t num_iterations;
  int32_t converged;
  int32_t step_count; /* SummaryStats.count */
  float step_mean;    /* SummaryStats.mean  */
  float step_s;       /* SummaryStats.s     */
} gulon_kmeans_report;

/* ---- runtime ---------------------------------------------------------- */
const char *gulon_last_error(void);
int32_t gulon_abi_version(void);
int32_t gulon_device_count(int32_t *out);
int32_t gulon_set_device(int32_t device);
int32_t gulon_device_synchronize(void);
int32_t gulon_dev_malloc(void **out, size_t bytes);
int32_t gulon_dev_free(void *p);
int32_t gulon_memcpy_h2d(void *dst_dev, const void *src_host, size_t bytes);
int32_t gulon_memcpy_d2h(void *dst_host, const void *src_dev, size_t bytes);

/* ---- Vectors.subvectors (Vectors.scala:84-104) -------------------------- */
int32_t gulon_subvectors(int32_t d, int32_t m, int32_t *from, int32_t *until);

/* ---- Coder (Coder.scala) ------------------------------------------------ */
/* ProductQuantizer.coderFactory (ProductQuantizer.scala:11-16): width for k
 * clusters after Coder.factoryFor rounding (Coder.scala:35-45); -1 if > 16 bit. */
int32_t gulon_coder_width(int32_t num_clusters, int32_t *width_out);
/* BytePackedCoder.bytesPerCode / BytePlus (Coder.scala:82-83,153) */
int32_t gulon_coder_bytes(int32_t width, int32_t length, int32_t *bytes_out);
/* Coder.buildCode (Coder.scala:85-89,100-108,115-123,130-136,147-161) */
int32_t gulon_coder_build(int32_t width, const int32_t *indices, int32_t length, uint8_t *code_out);
/* Coder.getIndex for all i (Coder.scala:91-92,110-111,125-126,138-139,163-167) */
int32_t gulon_coder_unpack(int32_t width, const uint8_t *code, int32_t length, int32_t *indices_out);

/* ---- Matrix on the device ------------------------------------------------ */
/* Copies an n x d row-major host matrix to HBM (Matrix.scala:3). */
int32_t gulon_dataset_create(const float *x_host, int32_t n, int32_t d, gulon_dataset **out);
/* Synthetic data generated on the device, bit-identical to the CPU test
 * generator: kind 0 iid N(0,1)~Irwin-Hall, 1 clustered, 2 U[0,1), 3 overlapping clusters. */
int32_t gulon_dataset_create_synth(int32_t n, int32_t d, int32_t kind, uint64_t seed,
                                   int32_t ncentres, gulon_dataset **out);
int32_t gulon_dataset_destroy(gulon_dataset *ds);
int32_t gulon_dataset_shape(const gulon_dataset *ds, int32_t *n, int32_t *d);
/* device pointer to the n x d row-major float32 data (for *_dev entry points) */
int32_t gulon_dataset_device_ptr(const gulon_dataset *ds, const float **out);
/* copy rows[0..nrows) (row-major) back to the host */
int32_t gulon_dataset_get_rows(const gulon_dataset *ds, const int32_t *rows, int32_t nrows, float *out_host);

/* ---- KMeans (KMeans.scala) ------------------------------------------------ */
/* KMeans.init (KMeans.scala:188-196): k rows drawn with java.util.Random(seed),
 * with replacement.  c_out: k x s host floats; rows_out (nullable): k ints. */
int32_t gulon_kmeans_init(const gulon_dataset *ds, int32_t from, int32_t s, int32_t k, int32_t seed,
                          float *c_out, int32_t *rows_out);
/* KMeans.assign (serial, KMeans.scala:18-22,70-98; rng_batch = 0: one Random(0)
 * stream over all rows) and KMeans.parAssign (KMeans.scala:57-68; rng_batch =
 * 25000: a fresh Random(0) per 25 000-row batch).  assignments: n host ints,
 * written for every row (the reference leaves NaN-distance rows untouched; see
 * DESIGN.md). */
int32_t gulon_kmeans_assign(const gulon_dataset *ds, int32_t from, int32_t s, const float *centroids,
                            int32_t k, int32_t rng_batch, int32_t *assignments);
/* KMeans.fromAssignment (KMeans.scala:198-226): order-dependent fp32 running mean. */
int32_t gulon_kmeans_update(const gulon_dataset *ds, int32_t from, int32_t s, int32_t k,
                            const int32_t *assignments, float *c_out);
/* KMeans.iterate (KMeans.scala:100-106). */
int32_t gulon_kmeans_iterate(const gulon_dataset *ds, int32_t from, int32_t s, const float *c_in,
                             int32_t k, int32_t iters, float *c_out);
/* KMeans.computeClusters (KMeans.scala:134-157) with Config(numClusters = k,
 * maxIterations, seed) (KMeans.scala:129-132).  reports (nullable) receives the
 * ProgressReports in order (first = the init report); *n_reports the count. */
int32_t gulon_kmeans_train(const gulon_dataset *ds, int32_t from, int32_t s, int32_t k,
                           int32_t max_iterations, int32_t seed, float *c_out,
                           gulon_kmeans_report *reports, int32_t max_reports, int32_t *n_reports);

#ifdef GULON_TEST_HOOKS   /* libgulon_hip_testhooks.so only: the product library does not export these */
/* Self-test of the running mean's division (kmeans.hip, update_chains): the three-operation quotient
 * q0 = RN(a y), r = fma(-n, q0, a), q = fma(r, y, q0) with y = RN(1/n) against the correctly rounded division
 * for every divisor n in [1, n_max] (n_max < 2^24) and `numerators_per_divisor` numerators each (random, and
 * next to rounding boundaries).  *mismatches must come back 0. */
int32_t gulon_selftest_mean_division(int32_t n_max, int32_t numerators_per_divisor, uint64_t seed, int64_t *mismatches);
/* KMeans.fromAssignment (KMeans.scala:198-226) of one slice -- columns [from, from + s) of the host matrix x (n rows,
 * leading dimension ld), host assignments in [0, k) -- through the STREAMED update of PQ training (kmeans_stream.hip),
 * which the library proper reaches inside gulon_pq_train only.  centroids_out: k x s. */
int32_t gulon_selftest_stream_update(const float *x, int32_t n, int32_t ld, int32_t from, int32_t s, int32_t k,
                                     const int32_t *assign, float *centroids_out);
/* Self-test of the filter's conflict-ordered code copy (conflict_order.hip; no reference counterpart: where a row
 * sits inside the device copy is an implementation detail behind PQIndex.batchQuery, Index.scala:209-263).
 * codes: n_blocks x 64 rows x 16 code bytes; rows are re-dealt inside WINDOWS of four consecutive blocks (a last
 * window of fewer blocks: inside the blocks it has).  codes_out receives the re-dealt blocks, place_out[b * 64 + lane]
 * the place (0..255) in its window of the row that now sits in `lane` of block b: over a window the places are a
 * permutation and codes_out[b][lane] == codes[window rows][place].  rounds = passes of the ordering (0: identity). */
int32_t gulon_selftest_conflict_order(const uint8_t *codes, int64_t n_blocks, int32_t rounds, uint8_t *codes_out,
                                      uint8_t *place_out);
/* Self-test of the bf16-split MFMA filter of KMeans.assign (kmeans_mfma.hip, assign_bf16): largest
 * |d' - d| / band over 64 random rows x 32 random centroids of sub-dimension s (coordinates ~ scale * U(-1,1) with
 * outliers), d' = the matrix cores' value, d = the reference's unfused fp32 chain (KMeans.scala:42-47).  The filter
 * is sound while the result stays below 0.5 (the band is twice the error bound). */
int32_t gulon_selftest_assign_band(int32_t s, uint64_t seed, float scale, double *max_error_over_band);
#endif
/* Stage times of the training loop for bench.py's k-means record (BASELINE config 3): while enabled, every
 * stage of KMeans.computeClusters / ProductQuantizer.apply is closed by a device synchronisation and its wall
 * time accumulated over the iterations (process-wide; not for concurrent trainings):
 *   update   = KMeans.fromAssignment of all sub-quantizers (stable counting sort + sequential mean chains)
 *   assign   = first stage of parAssign of all sub-quantizers: centroid prep + the MFMA filter kernel
 *   recheck  = exact re-evaluation of the rows the filter flagged + java.util.Random tie replay
 *   converge = Arrays.equals(prev, next) + ProgressReport
 * mfma_flops = 2 n k s summed over the filtered sub-quantizers and iterations (2 n k d per PQ
 * iteration); update_bytes = 4 n s summed likewise (n d 4 B per PQ iteration). */
typedef struct {
  int32_t iterations;
  double update_ms, assign_ms, recheck_ms, converge_ms;
  double mfma_flops, update_bytes;
  double rows_rechecked, rows_total;
} gulon_kmeans_trace_totals;
int32_t gulon_kmeans_trace(int32_t enable);   /* resets the totals */
int32_t gulon_kmeans_trace_read(gulon_kmeans_trace_totals *out);

/* ---- ProductQuantizer (ProductQuantizer.scala) ----------------------------- */
/* ProductQuantizer.apply / fromSubvectors (ProductQuantizer.scala:121-153), Config
 * (:107-111): m independent computeClusters, seed = quantizer index.
 * cents_out: k*d floats.  reports (nullable): m x max_reports, n_reports: m. */
int32_t gulon_pq_train(const gulon_dataset *ds, int32_t m, int32_t k, int32_t max_iterations,
                       float *cents_out, gulon_kmeans_report *reports, int32_t max_reports,
                       int32_t *n_reports);
/* The same for quantizers [j_begin, j_end) only -- the unit of work when the m independent
 * sub-quantizers are trained on different GPUs (ProductQuantizer.scala:130-145 runs them with
 * parTraverse).  Only those quantizers' blocks of cents_out are written; reports/n_reports are
 * indexed from j_begin. */
int32_t gulon_pq_train_range(const gulon_dataset *ds, int32_t m, int32_t k, int32_t max_iterations,
                             int32_t j_begin, int32_t j_end, float *cents_out, gulon_kmeans_report *reports,
                             int32_t max_reports, int32_t *n_reports);
/* ProductQuantizer.encode (ProductQuantizer.scala:25-35): per quantizer the serial
 * assign, packed by the Coder for k (coderFactory :11-16).  codes_out: m arrays of
 * gulon_coder_bytes(width, n) bytes, back to back ([m][bytesPerCode], the
 * EncodedMatrix.encodings layout, EncodedMatrix.scala:11-23). */
int32_t gulon_pq_encode(const gulon_dataset *ds, int32_t m, int32_t k, const float *cents,
                        uint8_t *codes_out);
/* quantizers [j_begin, j_end) only: codes_out holds (j_end - j_begin) packed arrays back to back */
int32_t gulon_pq_encode_range(const gulon_dataset *ds, int32_t m, int32_t k, const float *cents,
                              int32_t j_begin, int32_t j_end, uint8_t *codes_out);

/* ---- Index (Index.scala) ---------------------------------------------------- */
/* Index.prepareQuery (Index.scala:352-383): t_out[B][m][k]. */
int32_t gulon_prepare_query(const float *cents, int32_t d, int32_t m, int32_t k, const float *queries,
                            int32_t b, float *t_out);
/* PQIndex(productQuantizer, data) (Index.scala:385-391): copies the m packed code
 * arrays (each gulon_coder_bytes(width,n) bytes, back to back) and the codebooks
 * to HBM.  row_base is added to every returned row id (row sharding).
 * Every code width of ProductQuantizer.coderFactory: k <= 256 (widths 0/2/4/8) runs the
 * byte-coded kernels; 256 < k <= 65536 (Coder.BytePlus, widths 10/12/16) the wide-code path
 * (exact scan, k_nn <= GULON_MAX_K, tie flags without GULON_FLAG_EXACT_REPLAY). */
int32_t gulon_index_create(const uint8_t *codes, int32_t n, int32_t d, int32_t m, int32_t k,
                           const float *cents, int32_t row_base, gulon_index **out);
int32_t gulon_index_destroy(gulon_index *idx);
/* Concurrency rules of a gulon_index handle (the reference's PQIndex is immutable and queried from many
 * threads at once, Tests.scala:109-122):
 *  - one handle = one workspace.  All per-query scratch (tables, partial lists, survivor queues, replay pools)
 *    belongs to the handle.  Calls on one handle may come from any thread and any stream: the host side is
 *    serialised by a mutex, and the DEVICE work of a call on another stream than the previous call's is
 *    ordered behind it with an event -- two batches "in flight" on one handle run one after the other,
 *    never on top of each other's scratch.
 *  - gulon_index_context_create gives another workspace over the SAME read-only codes and codebooks (no
 *    copy): one context per batch in flight (bench.py: one per stream) overlaps their device work.  A
 *    context is destroyed with gulon_index_destroy; the codes live until the index and all of its contexts
 *    are destroyed.  Create and use a context with the index's device current.
 *  - the host-pointer gulon_index_batch_query takes such a context internally (up to GULON_HOST_CONTEXTS,
 *    default 4, created on first use, each with a stream of its own), so concurrent callers overlap.
 *  - the two-call sequence scan_bounds -> scan_partial_bounded must not be interleaved with other calls on the
 *    same handle (any other query call drops the pending first half: GULON_ERR_INVALID_ARGUMENT). */
int32_t gulon_index_context_create(gulon_index *parent, gulon_index **out);
/* PQIndex.batchQuery(k, vectors, from, until) (Index.scala:417-440) followed by
 * Index.Result.fromHeap (Index.scala:83-94): per query the K nearest rows of
 * [from, until) by ADC distance, ascending.  out_idx/out_dist: [B][K];
 * out_count[B] = live entries (< K when until - from < K); out_flags (nullable)
 * [B] GULON_FLAG_*.  from/until are LOCAL row numbers (0..n). */
int32_t gulon_index_batch_query(gulon_index *idx, const float *queries, int32_t b, int32_t k_nn,
                                int32_t from, int32_t until, int32_t *out_idx, float *out_dist,
                                int32_t *out_count, int32_t *out_flags);
/* Device-resident form: queries/out_* are device pointers, work is enqueued on
 * `stream` (hipStream_t) and not synchronised. */
int32_t gulon_index_batch_query_dev(gulon_index *idx, const float *d_queries, int32_t b, int32_t k_nn,
                                    int32_t from, int32_t until, int32_t *d_out_idx, float *d_out_dist,
                                    int32_t *d_out_count, int32_t *d_out_flags, void *stream);
/* Per-shard partial top-(K+1) for the multi-GPU merge: d_part_dist/d_part_idx are
 * [B][K+1] device arrays, ascending by (distance, row id), padded with
 * (+inf, INT32_MAX). */
int32_t gulon_index_scan_partial_dev(gulon_index *idx, const float *d_queries, int32_t b, int32_t k_nn,
                                     int32_t from, int32_t until, float *d_part_dist,
                                     int32_t *d_part_idx, void *stream);
/* The same partial scan in two halves, so that ROW SHARDS can share their pruning bounds (one tiny
 * all-gather between the halves; sharded.py).  The reference scans every row of every partition
 * (Index.scala:424-437); the quantized filter in front of the exact arithmetic prunes against an
 * upper bound of the (K+1)-th distance, and a bound drawn from the samples of ALL shards is
 * `shards` times tighter than a shard's own -- results do not change, only the work.
 *   1. gulon_index_scan_bounds_dev: table build + sample scan of this shard; d_bounds [B][K+1] =
 *      the K+1 smallest sample distances per query, ascending, +inf padded (all +inf for ranges
 *      the filter does not take);
 *   2. the caller gathers the arrays of all shards: d_all_bounds [lists][B][K+1] (this shard's
 *      own among them, any order);
 *   3. gulon_index_scan_partial_bounded_dev: the rest of the scan against the (K+1)-th smallest
 *      value of the union; same arguments as step 1 on the same index (GULON_ERR_INVALID_ARGUMENT
 *      otherwise), outputs as gulon_index_scan_partial_dev.  k_nn <= GULON_MAX_K. */
int32_t gulon_index_scan_bounds_dev(gulon_index *idx, const float *d_queries, int32_t b, int32_t k_nn,
                                    int32_t from, int32_t until, float *d_bounds, void *stream);
int32_t gulon_index_scan_partial_bounded_dev(gulon_index *idx, const float *d_queries, int32_t b, int32_t k_nn,
                                             int32_t from, int32_t until, const float *d_all_bounds,
                                             int32_t lists, float *d_part_dist, int32_t *d_part_idx,
                                             void *stream);
/* Exact TopKHeap replay of tie-flagged queries across ROW SHARDS (the unsharded
 * gulon_index_batch_query* does this internally).  After gulon_topk_merge_dev every shard holds
 * the same flags; each shard then collects, for up to `max_flagged` flagged queries after the first
 * `skip` ones (ascending query number), the rows of its range that may insert into the reference's
 * heap (a superset, with global row ids, at most `pool` per query) into a buffer of
 * gulon_replay_pack_words(max_flagged, pool) int32 words; word [3] of the buffer receives the number
 * of flagged queries of the whole batch.  The buffers of all shards (in row order: shard 0 first),
 * laid out back to back, go to gulon_replay_apply_dev, which runs the literal TopKHeap
 * (TopKHeap.scala:57-79) over their union and overwrites idx/dist/count of those queries, setting
 * GULON_FLAG_EXACT_REPLAY.  A batch with more flagged queries than one round holds is finished by
 * further rounds with skip advanced (gulon_sharded_index_batch_query and gulon_amd/sharded.py loop
 * until skip >= word [3]).  A query with more than `pool` candidates in a shard (or more than 16 384
 * over all shards) keeps the (distance, row id) result and its tie flags.
 * max_flagged in [1, 1024], pool in [64, 8192]; the defaults below size the first, unconditional round. */
#define GULON_REPLAY_MAX_FLAGGED 16
#define GULON_REPLAY_POOL 2048
int64_t gulon_replay_pack_words(int32_t max_flagged, int32_t pool); /* -1: out of range */
int32_t gulon_index_replay_collect_dev(gulon_index *idx, const float *d_queries, int32_t b, int32_t k_nn,
                                       int32_t from, int32_t until, const int32_t *d_flags, int32_t skip,
                                       int32_t max_flagged, int32_t pool, int32_t *d_pack, void *stream);
int32_t gulon_replay_apply_dev(const int32_t *d_packs, int32_t lists, int32_t max_flagged, int32_t pool, int32_t b,
                               int32_t k_nn, int32_t *d_out_idx, float *d_out_dist, int32_t *d_out_count,
                               int32_t *d_out_flags, void *stream);
/* ---- PQIndex row-sharded over the GPUs of one node, ONE host process (sharded.hip) ---------------
 * The multi-GPU form of PQIndex.batchQuery for a caller that is one process (the JVM): shard s holds rows
 * [n*s/S, n*(s+1)/S) -- the from/until contract of Index.scala:417-419 -- on device devices[s] (a device
 * may hold several shards); every query batch runs bounds -> scan -> merge -> tie replay on all shards
 * with three RCCL all-gathers over xGMI in between (ncclCommInitAll over the distinct devices; librccl is
 * loaded on first use).  The merge is TopKHeap.merge (TopKHeap.scala:44-53) under the (distance, row id)
 * order and tie-flagged queries are replayed with the literal heap over the candidates of all shards, in
 * as many rounds as the batch needs: ids, order, distances and flags equal the unsharded
 * gulon_index_batch_query bit for bit, whatever the number of shards.  codes/cents as gulon_index_create
 * (all n rows).  GULON_MAX_K < k_nn < GULON_MAX_K_PEELED: peeled partial lists, pairwise merge, no tie replay
 * (as the unsharded index at such k_nn). */
int32_t gulon_sharded_index_create(const uint8_t *codes, int32_t n, int32_t d, int32_t m, int32_t k,
                                   const float *cents, const int32_t *devices, int32_t n_shards,
                                   gulon_sharded_index **out);
int32_t gulon_sharded_index_destroy(gulon_sharded_index *idx);
int32_t gulon_sharded_index_batch_query(gulon_sharded_index *idx, const float *queries, int32_t b, int32_t k_nn,
                                        int32_t *out_idx, float *out_dist, int32_t *out_count, int32_t *out_flags);
/* shards, distinct devices (= RCCL ranks), ncclGetVersion, and of the last batch: replay rounds, flagged queries
 * (every out pointer nullable) */
int32_t gulon_sharded_index_info(const gulon_sharded_index *idx, int32_t *n_shards, int32_t *n_devices,
                                 int32_t *rccl_version, int32_t *last_replay_rounds, int32_t *last_flagged_queries);
/* ---- GroupedIndex (Index.scala:231-308): coarse groups + product-quantized residuals ---------
 * Rows are in GROUPED order (WordVectors.grouped, WordVectors.scala:24-58): stably ordered by the
 * coarse cluster they were assigned to; group c covers rows [offsets[c-1], offsets[c]) (first group
 * from 0, last to n); group_centroids holds the centroids of the g groups as the reference's builder
 * loop emits them (the non-empty clusters in cluster order, preceded by a copy of original row 0's
 * centroid for an EMPTY group [0, 0) when row 0 is not in the first of them, :38-39); the codes are the ProductQuantizer codes of the RESIDUALS (row - its group's centroid,
 * WordVectors.Grouped.residuals :118-138).  Returned ids are grouped row positions.
 *
 * gulon_dataset_group_residuals builds that residual matrix on the device:
 *   out[i] = ds[perm[i]] - group_centroids[group_of[i]]                       (MathUtils.subtract)
 * gulon_grouped_index_batch_query = GroupedIndex.batchQuery (:254-282): per query searchSpace
 * (:285-299; strategy 0 = LimitGroups(limit), 1 = LimitVectors(limit)), then for every searched
 * group, nearest first, PQIndex.query on (query - centroid) over the group's rows and
 * TopKHeap.merge into the result heap, then Result.fromHeap.  The heaps are the reference's,
 * literally (array order included), so ids and order equal the JVM's also under distance ties --
 * equally distant coarse centroids included (they are common: WordVectors.grouped's leading empty
 * group repeats a centroid, WordVectors.scala:38-39): queries whose searched groups hang on such a tie
 * (or on a NaN distance) select their groups through the literal exactNearestNeighbours heap.
 * Groups may be empty (offsets may repeat).  k_nn <= GULON_MAX_K on the fast paths; up to 2048 (8-bit codes) through
 * the literal heaps alone, kept in LDS (Tests.scala asks for up to 1000 neighbours). */
typedef struct gulon_grouped_index gulon_grouped_index;
int32_t gulon_dataset_group_residuals(const gulon_dataset *ds, const int32_t *perm, const int32_t *group_of,
                                      const float *group_centroids, int32_t g, gulon_dataset **out);
int32_t gulon_grouped_index_create(const uint8_t *codes, int32_t n, int32_t d, int32_t m, int32_t k,
                                   const float *pq_cents, const float *group_centroids, const int32_t *offsets,
                                   int32_t g, gulon_grouped_index **out);
int32_t gulon_grouped_index_destroy(gulon_grouped_index *idx);
int32_t gulon_grouped_index_batch_query(gulon_grouped_index *idx, const float *queries, int32_t b, int32_t k_nn,
                                        int32_t strategy, int32_t limit, int32_t *out_idx, float *out_dist,
                                        int32_t *out_count);
int32_t gulon_grouped_index_batch_query_dev(gulon_grouped_index *idx, const float *d_queries, int32_t b,
                                            int32_t k_nn, int32_t strategy, int32_t limit, int32_t *d_out_idx,
                                            float *d_out_dist, int32_t *d_out_count, void *stream);

/* Kernel timing for the roofline line of bench.py: when enabled, every scan-kernel
 * launch of this index is bracketed by hipEvents on the launch stream;
 * gulon_index_profile_read synchronises them and returns the summed duration. */
int32_t gulon_index_profile(gulon_index *idx, int32_t enable);
int32_t gulon_index_profile_read(gulon_index *idx, double *scan_ms_total, int32_t *launches);
/* Same, plus the number of rows the bracketed launches covered (the dominant kernel is the
 * quantized filter when it is active, the exact scan otherwise). */
int32_t gulon_index_profile_read_ex(gulon_index *idx, double *ms_total, int32_t *launches, int64_t *rows_total);
/* Of the last batch this handle ran through the quantized filter (synchronises the device): its query tiles
 * and how many of them the device-side safety net redid with the exact scan (unusable bound or survivor-queue
 * overflow).  query_tiles = 0: the batch took the exact scan. */
int32_t gulon_index_filter_stats(gulon_index *idx, int32_t *query_tiles, int32_t *tiles_redone);
/* Launch-shape / algorithm knobs of the scan, per handle (tests and tuning experiments).  A handle takes its settings
 * from the ENVIRONMENT when it is created -- the variable names are the keys: "GULON_SCAN_FILTER" (0/1),
 * "GULON_FILTER_MIN_RB", "GULON_FILTER_PERIOD", "GULON_FILTER_STAGE0", "GULON_FILTER_STAGE1", "GULON_FILTER_SAMPLE",
 * "GULON_FILTER_CAP", "GULON_FILTER_NADD", "GULON_FILTER_BLOCKS", "GULON_FILTER_SHARED_STAGE1", "GULON_SCAN_BLOCKS",
 * "GULON_SCAN_PRUNE", "GULON_SCAN_PRUNE_FROM" -- and this call changes them for ONE handle (and the contexts created
 * from it afterwards); there is no process-wide setter.  Results never depend on them.
 * "GULON_FILTER_ORDER" (default 1): an index of one-word codes (m <= 16) created while the environment's value is
 * non-zero keeps a second, conflict-ordered copy of its codes for the filter kernel (+ 17 bytes per row; the value =
 * rounds of the ordering); per handle 0 makes the filter read the plain copy again. */
int32_t gulon_index_tuning(gulon_index *idx, const char *key, int32_t value);
/* TopKHeap.merge semantics (TopKHeap.scala:44-53, used at Index.scala:279) under
 * the deterministic (distance, row id) order: merges `lists` partial lists per
 * query, laid out [lists][B][K+1], into the final K.  list_stride = elements
 * between consecutive lists (0 = B*(K+1), i.e. dense).  k_nn > GULON_MAX_K (lists from
 * gulon_index_scan_partial_dev at that k_nn): pairwise merges through scratch; synchronises the stream. */
int32_t gulon_topk_merge_dev(const float *d_part_dist, const int32_t *d_part_idx, int32_t lists,
                             int64_t list_stride, int32_t b, int32_t k_nn, int32_t *d_out_idx,
                             float *d_out_dist, int32_t *d_out_count, int32_t *d_out_flags, void *stream);
/* All-NaN queries on a ROW-SHARDED index: a query with a NaN component has every distance NaN
 * (Index.scala:352-383), the reference's heap then keeps the first min(K, n_total) rows it is offered
 * (TopKHeap.scala:69-79) and Result.fromHeap returns them as [1, ..., c-1, 0] with NaN distances.  Run on the
 * merged result of the shards (gulon_amd/sharded.py, gulon_sharded_index_batch_query); the unsharded
 * gulon_index_batch_query* handles every non-finite case itself (GULON_FLAG_NONFINITE). */
int32_t gulon_nan_queries_fix_dev(const float *d_queries, int32_t b, int32_t d, int32_t k_nn, int32_t n_total,
                                  int32_t *d_out_idx, float *d_out_dist, int32_t *d_out_count, int32_t *d_out_flags,
                                  void *stream);
/* host-pointer convenience form of the same merge */
int32_t gulon_topk_merge(const float *part_dist, const int32_t *part_idx, int32_t lists, int32_t b,
                         int32_t k_nn, int32_t *out_idx, float *out_dist, int32_t *out_count,
                         int32_t *out_flags);
/* Index.exactNearestNeighbours (Index.scala:209-229) + Result.fromHeap for B
 * queries over rows [from, until) of the dataset. */
int32_t gulon_exact_knn(const gulon_dataset *ds, int32_t from, int32_t until, const float *queries,
                        int32_t b, int32_t k_nn, int32_t *out_idx, float *out_dist, int32_t *out_count,
                        int32_t *out_flags);
/* MathUtils.distanceSq(query_q, X[rows[q][i]]) (MathUtils.scala:85-95) for the
 * recall harness (Tests.scala:18-41): out[B][K]; rows < 0 are skipped (0). */
int32_t gulon_distance_sq_rows(const gulon_dataset *ds, const float *queries, int32_t b,
                               const int32_t *rows, int32_t k_nn, float *out);

#ifdef __cplusplus
}
#endif
#endif /* GULON_HIP_H */
