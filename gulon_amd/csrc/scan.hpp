// Internal declarations shared by the top-k users (scan, exact kNN).
#pragma once
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <memory>

#include "common.hpp"

namespace gulon {
// Merge `lists` sorted partial lists per query (element (l,q,e) at l*stride_l + q*stride_q + e,
// each list K+1 long, padded with (+inf, INT_MAX)).  final_out: write [B][K] idx/dist/count/flags,
// else write one (K+1)-list per query to out_pv/out_pi.
void launch_merge(bool final_out, const float *in_v, const int *in_i, int lists, long long stride_l,
                  long long stride_q, int B, int K, int *out_idx, float *out_dist, int *out_count,
                  int *out_flags, float *out_pv, int *out_pi, hipStream_t st);
bool replay_enabled();
// large-K peeling helpers (scan.hip): append a 64-entry round to [B][cap] and set the next lower bound;
// then cut the concatenated list to [B][K] + count + tie flags
void launch_peel_update(const float *tv, const int *ti, int B, int round, int cap, float *pv, int *pi, float *lbv,
                        int *lbi, hipStream_t st);
void launch_peel_finalize(const float *pv, const int *pi, int B, int cap, int K, int *out_idx, float *out_dist,
                          int *out_count, int *out_flags, hipStream_t st);
}  // namespace gulon

namespace gulon { struct ScanTuning; }
// PQIndex on the device (opaque to C callers)
using gulon::DevBuf;
struct gulon_index {
  // launch-shape / algorithm knobs of THIS handle: the environment at its creation, then gulon_index_tuning.  Contexts
  // inherit their parent's at creation.
  std::shared_ptr<gulon::ScanTuning> tune;
  int32_t n = 0, d = 0, m = 0, k = 0, row_base = 0;
  int vec = 16, ng = 1, m_pad = 16, nsub = 1, w = 4;   // w: queries interleaved per table entry
  DevBuf<uint8_t> codes;   // [n/64][ng][64][vec]
  // the filter's own copy of one-word codes (m <= 16), rows re-dealt inside every 64-row block for the LDS bank
  // conflicts of its table gathers (conflict_order.hip), and each lane's place in the block's row order
  DevBuf<uint8_t> fcodes, fperm;   // [n/64][64][16], [n/64][64]
  int fwindow = 1;                 // blocks per ordering window of fcodes (4: fperm = place in a 256-row window; 1: in the block)
  bool wide = false;       // k > 256: 16-bit codes, tables in HBM (wide.hip)
  DevBuf<uint16_t> wcodes; // wide: [n/64][m][64]
  DevBuf<float> wpartial;  // wide, sliced tables: running sums [queries of the sub-batch][rows]
  DevBuf<uint32_t> wpark;  // wide filter, sliced 8-bit tables: byte sums [query group][row block][64 lanes] (wide_filter.hip)
  DevBuf<float> cents;     // k*d
  DevBuf<int> from, sdim;  // m
  // scratch, grown on demand under `mu`
  DevBuf<float> tables;
  DevBuf<float> part_v;
  DevBuf<int> part_i;
  DevBuf<unsigned> gtau;   // per-query cross-workgroup pruning thresholds (float bits)
  DevBuf<float> stage_q;
  DevBuf<int> stage_oi, stage_oc, stage_of;
  DevBuf<float> stage_od;
  DevBuf<int> flags_scratch;
  DevBuf<unsigned long long> dbg;   // GULON_SCAN_TIMELINE stamps
  // large-K peeling rounds
  DevBuf<float> peel_v, peel_tv, peel_lbv;
  DevBuf<int> peel_i, peel_ti, peel_lbi;
  // exact tie replay (replay.hip)
  DevBuf<int> rp_pack, rp_segcnt, rp_segi, rp_precnt, rp_l0i, rp_l0c;
  DevBuf<float> rp_q, rp_tables, rp_segtop, rp_prefix, rp_l0v;
  DevBuf<int> rp_fb, rp_fini;           // level 2 of the replay through the quantized filter (filter.hip)
  DevBuf<int> rp_done;                  // [3][F]: served by rp_shortcut / left to the segment scan / last row that can insert
  DevBuf<int> rp_order;                 // the long level's order of the flagged queries (rp_filter_order)
  DevBuf<float> rp_tau, rp_finv, rp_mins;
  // quantized lower-bound filter (filter.hip)
  DevBuf<float> fin_v, qmins, tau0; // running exact (K+1)-lists [Bq][keff]; table minima [Bq][m_pad]; sample bounds
  DevBuf<int> fin_i, sv_cnt, sv_queue, fb_tile;
  DevBuf<uint8_t> qtab;             // [Bq/16][m_pad][256][16] quantized table entries
  DevBuf<float> nf_tables;          // literal.hip: one fp32 table per workgroup of the non-finite pass
  float cents_absmax = 0.f;         // max |centroid coordinate| (+inf if any is NaN / inf): overflow bound of a query
  // optional hipEvent bracketing of the dominant scan kernel (bench.py roofline line)
  bool profile = false;
  long long prof_rows = 0;          // rows covered by the bracketed launches
  std::vector<std::pair<hipEvent_t, hipEvent_t>> events;   // recorded pairs (events belong to ev_pool)
  std::vector<hipEvent_t> ev_pool;  // created once: hipEventCreate inside the timed region is slow
  size_t ev_next = 0;
  std::mutex mu;
  // gulon_index_scan_bounds_dev -> gulon_index_scan_partial_bounded_dev: arguments of the pending first half
  int pend_b = -1, pend_k = -1, pend_from = -1, pend_until = -1;
  // host-mapped word the filter's fallback launch sets when it had anything to do: sizes the next one
  int *fb_hint_h = nullptr, *fb_hint_d = nullptr;
  int *rp_hint_h = nullptr, *rp_hint_d = nullptr;   // host-mapped: flagged queries of a recent batch (replay.hip)
  int fb_wide_left = 0;   // launches that stay wide after the word was last seen set
  int last_filter_tiles = 0;   // query tiles of the last filtered batch on this handle (0: it took the exact scan)
  // ---- query contexts (gulon_index_context_create): a context BORROWS the read-only members of its parent
  // (codes, wcodes, cents, from, sdim) and owns every scratch buffer above, so batches in flight on
  // different streams / threads never share device scratch.  The parent stays alive until its last context
  // is destroyed (refs).
  gulon_index *parent = nullptr;
  std::atomic<int> refs{1};
  // one handle = one workspace: the device work of consecutive calls on DIFFERENT streams is ordered through
  // this event (two batches in flight on one handle are serialised on the device, never corrupted)
  hipEvent_t order_ev = nullptr;
  hipStream_t order_st = nullptr;
  bool order_set = false;
  // host-pointer calls (gulon_index_batch_query) from several threads -- the reference's recall harness does
  // that (Tests.scala:109-122) -- run concurrently on lazily created internal contexts, each with its own stream
  std::mutex host_mu;
  std::condition_variable host_cv;
  std::vector<gulon_index *> host_all, host_free;   // internal contexts (owned); the handle itself is the first
  bool host_self_busy = false;
  hipStream_t host_stream = nullptr;
  hipEvent_t take_event() {
    if (ev_next == ev_pool.size()) {
      hipEvent_t e = nullptr;
      HIP_CHECK(hipEventCreate(&e));
      ev_pool.push_back(e);
    }
    return ev_pool[ev_next++];
  }
  ~gulon_index() {
    for (auto *c : host_all) delete c;
    for (auto &e : ev_pool) (void)hipEventDestroy(e);
    if (fb_hint_h) (void)hipHostFree(fb_hint_h);
    if (rp_hint_h) (void)hipHostFree(rp_hint_h);
    if (order_ev) (void)hipEventDestroy(order_ev);
    if (host_stream) (void)hipStreamDestroy(host_stream);
  }
};

namespace gulon {
// a new workspace over `parent`'s read-only data (no reference counting: the caller keeps `parent` alive)
gulon_index *make_context(gulon_index *parent);
// RAII: order this call's device work after the previous call's on the same handle when the stream differs
struct StreamOrder {
  gulon_index *ix;
  hipStream_t st;
  StreamOrder(gulon_index *ix_, hipStream_t st_) : ix(ix_), st(st_) {
    if (!ix->order_ev) HIP_CHECK(hipEventCreateWithFlags(&ix->order_ev, hipEventDisableTiming));
    if (ix->order_set && ix->order_st != st) HIP_CHECK(hipStreamWaitEvent(st, ix->order_ev, 0));
  }
  void done() {
    HIP_CHECK(hipEventRecord(ix->order_ev, st));
    ix->order_st = st;
    ix->order_set = true;
  }
};
}  // namespace gulon


namespace gulon {
// Strided selection of row blocks: the e-th eligible block (relative to the first block of the
// scanned range) is (e / width) * period + lo + e % width.  {1, 0, 1} = every block.
struct RbMap { int period, lo, width; };
inline int rbmap_count(int rb_total, RbMap mp) {
  int extra = rb_total % mp.period - mp.lo;
  extra = extra < 0 ? 0 : extra > mp.width ? mp.width : extra;
  return (rb_total / mp.period) * mp.width + extra;
}

// launch shape knobs (the environment at index creation / gulon_index_tuning: for experiments and tests)
// filter.hip: of the 16 quantizers of a one-word code, the entries of the last GULON_FILTER_GLB are fetched through
// the vector L1 instead of LDS; conflict_order.hip orders rows for the bank conflicts of the others
#ifndef GULON_FILTER_GLB
#define GULON_FILTER_GLB 3
#endif
constexpr int FILTER_LDS_QUANTIZERS = 16 - GULON_FILTER_GLB;

struct ScanTuning {
  int threads = 1024;        // workgroup size (16 waves hide the pruning checkpoints' LDS drain)
  int target_blocks = 4096;  // workgroups per launch aimed for
  int prune = 1;             // exact early termination on/off
  int prune_from = -1;       // first quantizer index with a pruning checkpoint (-1: m_pad/2)
  int filter = 1;            // quantized lower-bound filter on/off
  int filter_min_rb = 512;   // smallest range (in 64-row blocks) the filter is used for
  int filter_period = 128;   // row blocks per sampling period
  int filter_sample = 65536; // most sample rows whose exact distances give the initial bounds
  int filter_stage0 = 0;     // blocks per period scanned by an extra first filter stage (0: none)
  int filter_stage1 = 10;    // blocks per period scanned by the second filter stage (6 until round 2: 2.44 -> 2.38 ms per
                             // batch at 10 M rows, 0.497 -> 0.469 at 1.25 M; flat from 10 to 20)
  int filter_cap = 32768;    // survivor queue entries per query and stage
  int filter_nadd = 0;       // table entries summed in 8 bits before widening (2: 7-bit, 4: 6-bit levels, 0: by m)
  int filter_blocks = 4096;         // workgroups aimed for by a filter launch
  int filter_shared_stage1 = -1;    // bounds shared across shards: run the short first stage? (-1: by sample size)
  int filter_order = 1;             // conflict-ordered code copy: built at index creation (the environment's value) / used (per handle)
  ScanTuning();
  bool set(const char *key, int v);
};
const ScanTuning &tuning_defaults();
inline const ScanTuning &tuning_of(const gulon_index *ix) { return ix && ix->tune ? *ix->tune : tuning_defaults(); }

// wide.hip: indexes with more than 256 centroids per quantizer (code widths 10/12/16)
void wide_store_codes(gulon_index *ix, const uint16_t *wide16 /* device, [m][n] */);
void launch_build_tables_wide(const float *cents, const int *from, const int *sdim, int d, int m, int k, const float *dQ,
                              int q0, int nq, float *tables /*[nq][m][k]*/, hipStream_t st);
// wide_filter.hip: the quantized lower-bound filter for 16-bit code words (k <= 1024 at m = 16)
bool wide_filter_eligible(const gulon_index *ix, int B, int K, int rb_total);
void run_wide_filter_query(gulon_index *ix, const float *dQ, int B, int K, int from, int until, bool final_out,
                           int *d_oi, float *d_od, int *d_oc, int *d_of, float *d_pv, int *d_pi, hipStream_t st);
void launch_scan_wide_range(gulon_index *ix, int B, int K, int from, int until, int rb_begin, int rb_total,
                            int rb_per_chunk, int nchunks, const int *enable, hipStream_t st);
void run_wide_query(gulon_index *ix, const float *dQ, int B, int K, int from, int until, bool final_out, int *d_oi,
                    float *d_od, int *d_oc, int *d_of, float *d_pv, int *d_pi, hipStream_t st);

// scan.hip: exact scan of the selected row blocks of [from, until) into ix->part_v/part_i
// ([query][nchunks][keff]); tile_enable (device, per query tile) skips disabled tiles.
void launch_scan(gulon_index *ix, int ntiles, int nchunks, int rb_begin, int e_count, int e_per_chunk, RbMap mp,
                 int from, int until, int keff, hipStream_t st, const float *lbv = nullptr, const int *lbi = nullptr,
                 const int *tile_enable = nullptr, bool one_sub = false, int grid_x = 0, int *hint = nullptr);
// merge_lists<false> restricted to the queries of enabled tiles (fallback of the filter)
void launch_merge_enabled(const float *in_v, const int *in_i, int lists, long long stride_l, long long stride_q, int B,
                          int K, float *out_pv, int *out_pi, const int *tile_enable, int qt, hipStream_t st);
// filter.hip: level 2 of the tie replay (replay.hip) through the quantized filter.  For the flagged queries f < *count
// (tables [f][m_pad][256] with their minima `mins`, bounds = the K-th of prefix_v[f][K] when prefix_c[f] >= K) every
// row of blocks [rb_lo, rb_hi) with an exact distance BELOW the bound is appended to the query's candidate pool
// (evv / evi / evcnt, `pool` entries per query).  only[f] is left 0 for the queries served here and 1 for those the
// caller's segment scan still has to do (fewer than `min_flagged` flagged queries, unusable bound, queue overflow).
// Returns false (nothing enqueued) when this index does not take the filter.
bool replay_level2_filtered(gulon_index *ix, int F, int K, int rb_lo, int rb_hi, int from, int until,
                            const float *tables, const float *mins, const float *prefix_v, const int *prefix_c,
                            const int *count, float *evv, int *evi, int *evcnt, int pool, int **only, hipStream_t st,
                            const int *done, const int *rlast, int *scanme);
// filter.hip: sample scan -> quantized filter stages -> exact re-evaluation of the survivors.
bool filter_eligible(const gulon_index *ix, int K, int rb_total);
// Bounds shared across row shards (sharded.py): phase 1 stops after the sample scan and writes the K+1
// smallest sample distances per query ([B][K+1], ascending, +inf padded) to bounds_out; phase 2 resumes
// with the `lists` gathered arrays ([lists][B][K+1]) -- the (K+1)-th smallest of their union bounds the
// (K+1)-th distance of the WHOLE index, so every shard filters against the global bound.
struct SharedBounds { int phase; float *bounds_out; const float *all_bounds; int lists; };
void run_filter_query(gulon_index *ix, const float *dQ, int B, int K, int from, int until, bool final_out, int *d_oi,
                      float *d_od, int *d_oc, int *d_of, float *d_pv, int *d_pi, hipStream_t st,
                      const SharedBounds *sb = nullptr);

#ifdef __HIPCC__
template <int VEC> struct CodeWord;
template <> struct CodeWord<4> { using type = uint32_t; };
template <> struct CodeWord<16> { using type = uint4; };
template <int VEC>
__device__ inline uint32_t code_byte(const typename CodeWord<VEC>::type &w, int b);
template <>
__device__ inline uint32_t code_byte<4>(const uint32_t &w, int b) { return (w >> (8 * b)) & 0xFFu; }
template <>
__device__ inline uint32_t code_byte<16>(const uint4 &w, int b) {
  uint32_t x = (b < 4) ? w.x : (b < 8) ? w.y : (b < 12) ? w.z : w.w;
  return (x >> (8 * (b & 3))) & 0xFFu;
}
#endif

// Index.prepareQuery tables, W queries interleaved (scan.hip)
void launch_build_tables(int W, gulon_index *ix, const float *dQ, int B, int Bpad, float *tables, hipStream_t st,
                         const int *live_queries = nullptr, float *mins = nullptr);
// literal.hip: queries whose distances can be NaN / +inf are redone through the literal TopKHeap over all rows
float centroid_absmax(const float *d_cents, long long count);
void run_nonfinite_literal(gulon_index *ix, const float *dQ, int B, int K, int from, int until, int *d_oi, float *d_od,
                           int *d_oc, int *d_of, hipStream_t st);
// For every query flagged with an exact distance tie, recompute the result with the
// reference's TopKHeap semantics (insertion history in row order) -- replay.hip
void run_tie_replay(gulon_index *ix, const float *dQ, int B, int K, int from, int until, int *d_oi, float *d_od,
                    int *d_oc, int *d_of, hipStream_t st);
void launch_conflict_order(const uint8_t *src, uint8_t *dst, uint8_t *perm, long long nblk, int nq, int rounds,
                           hipStream_t st);
bool conflict_order_windowed();   // the ordering spans windows of four blocks (default) or single blocks
}  // namespace gulon
