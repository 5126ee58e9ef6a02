// Internal declarations shared by the top-k users (scan, exact kNN).
#pragma once
#include <algorithm>
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

// PQIndex on the device (opaque to C callers)
using gulon::DevBuf;
struct gulon_index {
  int32_t n = 0, d = 0, m = 0, k = 0, row_base = 0;
  int vec = 16, ng = 1, m_pad = 16, nsub = 1, w = 4;   // w: queries interleaved per table entry
  DevBuf<uint8_t> codes;   // [n/64][ng][64][vec]
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
  DevBuf<int> rp_list, rp_count, rp_segcnt, rp_evcnt, rp_overflow, rp_evi, rp_precnt;
  DevBuf<float> rp_q, rp_tables, rp_segtop, rp_prefix, rp_evv;
  // optional hipEvent bracketing of the scan kernel (bench.py roofline line)
  bool profile = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
  std::mutex mu;
  ~gulon_index() {
    for (auto &e : events) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
  }
};


namespace gulon {
// Index.prepareQuery tables, W queries interleaved (scan.hip)
void launch_build_tables(int W, gulon_index *ix, const float *dQ, int B, int Bpad, float *tables, hipStream_t st,
                         const int *live_queries = nullptr);
// For every query flagged with an exact distance tie, recompute the result with the
// reference's TopKHeap semantics (insertion history in row order) -- replay.hip
void run_tie_replay(gulon_index *ix, const float *dQ, int B, int K, int from, int until, int *d_oi, float *d_od,
                    int *d_oc, int *d_of, hipStream_t st);
}  // namespace gulon
