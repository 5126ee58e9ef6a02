// Exact TopKHeap semantics under distance ties (TopKHeap.scala:57-79, Index.scala:431-434).
//
// The reference's heap rejects a candidate equal to its current maximum, and which of
// several equal maxima sits at the root depends on the whole insertion history -- including
// early, large entries that are evicted later.  So when a query has equal distances around
// its K-th neighbour, the returned ids are a function of the sequence of successful
// insertions ("events").  The fast path orders by (distance, row id) and flags such queries;
// this file recomputes them exactly:
//
//   an insertion happens at row r  <=>  fewer than K rows came before, or
//                                       dist(r) < K-th smallest distance among rows < r
//
// and the literal heap makes exactly that test itself.  So it is enough to feed it, in row
// order, any SUPERSET of the inserting rows: a row that would not insert is rejected by the
// heap exactly as in the reference and leaves it untouched.  A superset is cheap to produce in
// parallel: a segment of rows that starts from an upper bound of the true running threshold
// (the K smallest distances of some earlier rows) and tightens it with its own rows emits every
// row below its running threshold -- all true insertions and a few more.  Three levels:
//   level 0  rows [0, 4096)            one wave, cold start (exact events)
//   level 1  the next 128 K rows       32 segments, each starting from level 0's K smallest
//   level 2  the rest                  up to 2048 segments starting from the K smallest of levels 0+1
// (about K * (ln(4096/K) + 1 + 1 + n/132K) candidates per query), then
//   rp_heap  sorts the candidates by row id and pushes them through a literal TopKHeap,
//            drained like Result.fromHeap.
// Distances are the same bit-exact j-ordered sums as in the main scan.
#include "scan.hpp"

namespace gulon {

constexpr int RP_MAXF = 1024;    // flagged queries replayed per batch (all of a 1024-query batch)
constexpr int RP_POOL = 8192;    // candidate rows kept per flagged query (more: the flagged result stays)
constexpr int RP_LDS_POOL = 16384;   // candidates rp_heap can sort in LDS (all shards of a query together)
#ifndef GULON_RP_L1_BLOCKS
#define GULON_RP_L1_BLOCKS 2048
#endif
#ifndef GULON_RP_L1_SEG
#define GULON_RP_L1_SEG 64
#endif
#ifndef GULON_RP_L0_BLOCKS
#define GULON_RP_L0_BLOCKS 64
#endif
constexpr int RP_L0_BLOCKS = GULON_RP_L0_BLOCKS, RP_L1_BLOCKS = GULON_RP_L1_BLOCKS, RP_L1_SEG = GULON_RP_L1_SEG, RP_L2_SEGS = 2048, RP_L2_MIN = 16;

// Candidate buffer ("pack", int32 words) for F flagged queries with C candidates each:
//   [0] flagged queries in this pack (<= F)   [1] F   [2] C   [3] flagged queries of the whole batch
//   [4, 4+F) query ids, ascending          [4+F, 4+2F) candidates written per query
//   then F*C distances (float bits), then F*C global row ids.
// One pack per row shard; rp_heap consumes the packs of all shards of a query together.
struct Pack {
  int *base; int F, C;
  __host__ __device__ int *count() const { return base; }
  __host__ __device__ int *list() const { return base + 4; }
  __host__ __device__ int *evcnt() const { return base + 4 + F; }
  __host__ __device__ float *evv() const { return reinterpret_cast<float *>(base + 4 + 2 * F); }
  __host__ __device__ int *evi() const { return base + 4 + 2 * F + (size_t)F * C; }
};
size_t replay_pack_words(int F, int C) { return 4 + 2 * (size_t)F + 2 * (size_t)F * C; }

// flagged queries in ascending order (deterministic: every shard builds the same list), the first `skip`
// of them left out (they were replayed by an earlier round); one wave
__global__ __launch_bounds__(64) void rp_collect(const int *__restrict__ flags, int B, Pack pk, int skip,
                                                 int *__restrict__ hint /* host-mapped; may be null */) {
  const int lane = threadIdx.x;
  int base = 0;
  for (int q0 = 0; q0 < B; q0 += 64) {
    const int q = q0 + lane;
    const bool fl = q < B && (flags[q] & (GULON_FLAG_BOUNDARY_TIE | GULON_FLAG_INTERIOR_TIE));
    const unsigned long long mk = __ballot(fl);
    const int pos = base + __popcll(mk & ((1ull << lane) - 1ull)) - skip;
    if (fl && pos >= 0 && pos < pk.F) pk.list()[pos] = q;
    base += __popcll(mk);
  }
  const int here = max(0, min(base - skip, pk.F));
  for (int f = lane; f < pk.F; f += 64) {
    pk.evcnt()[f] = 0;
    if (f >= here) pk.list()[f] = -1;
  }
  if (lane == 0) { pk.count()[0] = here; pk.count()[1] = pk.F; pk.count()[2] = pk.C; pk.count()[3] = base; }
  if (lane == 0 && hint && skip == 0) *hint = base;
}

// (+ rlast[f], when the main pass's result is at hand: the last row that can still insert for flagged query f -- the
// largest row id among its K best.  At any later row all K of them are in the reference's heap, its root is at most
// the final K-th distance, and a row strictly below that distance would be one of the K best itself: no level of
// the replay has to look past it.  INT_MAX: unknown / fewer than K results)
__global__ void rp_gather_queries(const float *__restrict__ Q, int d, int maxf, const int *__restrict__ list,
                                  const int *__restrict__ count, float *__restrict__ Qf, int K,
                                  const int *__restrict__ fin_i, const int *__restrict__ fin_c, int *__restrict__ rlast) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  int f = t / d, c = t - f * d;
  if (f >= maxf) return;
  int nf = min(*count, maxf);
  Qf[t] = f < nf ? Q[(size_t)list[f] * d + c] : 0.f;
  if (rlast && c == 0) {
    int last = INT_MAX;
    if (f < nf && fin_i) {
      const int q = list[f];
      if (q >= 0 && fin_c[q] == K) {
        last = 0;
        for (int i = 0; i < K; i++) last = max(last, fin_i[(size_t)q * K + i]);
      }
    }
    rlast[f] = last;
  }
}

template <int VEC> struct RpWord;
template <> struct RpWord<4> { using type = uint32_t; };
template <> struct RpWord<16> { using type = uint4; };
__device__ inline uint32_t rp_byte(const uint32_t &w, int b) { return (w >> (8 * b)) & 0xFFu; }
__device__ inline uint32_t rp_byte(const uint4 &w, int b) {
  uint32_t x = (b < 4) ? w.x : (b < 8) ? w.y : (b < 12) ? w.z : w.w;
  return (x >> (8 * (b & 3))) & 0xFFu;
}

// One wave per (segment, flagged query); 4 segments per workgroup share the query's table in LDS.
// Starts from the K smallest distances of earlier rows (start_v, ascending; nullptr: cold) and
// emits every row below its running K-th smallest distance into the query's candidate pool.
template <int VEC>
__global__ __launch_bounds__(256) void rp_scan(const uint8_t *__restrict__ codes, int ng, int m_pad,
                                               const float *__restrict__ tables /*[f][m_pad][256]*/,
                                               const int *__restrict__ count, int maxf, int row_from, int row_until,
                                               int row_base, int rb_lo, int rb_hi, int rb_per_seg, int nseg, int K,
                                               const float *__restrict__ start_v, const int *__restrict__ start_c,
                                               float *__restrict__ out_v, int *__restrict__ out_i,
                                               int *__restrict__ out_c, int pool, float *__restrict__ evv,
                                               int *__restrict__ evi, int *__restrict__ evcnt,
                                               const int *__restrict__ only /* per flagged query; null: all */,
                                               const int *__restrict__ rlast /* per flagged query: last row that can insert; null: none known */) {
  using Word = typename RpWord<VEC>::type;
  extern __shared__ float tab[];   // m_pad * 256
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int s = blockIdx.x * 4 + wave;
  const int nf = min(*count, maxf);
  // the grid has a fixed, small y extent; each block walks the flagged queries f = y, y+Y, ...
  for (int f = blockIdx.y; f < nf; f += gridDim.y) {
    if (only && only[f] != 1) continue;     // (uniform) 0: this query's rows came through the quantized filter; 2: rp_shortcut
    // no row after rlast[f] inserts: a level that lies behind it has nothing to emit for this query (its seeds for
    // the next level are not needed either: that level lies behind it as well)
    int rb_stop = rb_hi;
    if (rlast && rlast[f] != INT_MAX) rb_stop = min(rb_hi, max(0, ((rlast[f] - row_base) >> 6) + 1));
    if (rb_lo >= rb_stop && !out_v) continue;          // (uniform; a level that feeds the next one still writes its lists)
    __syncthreads();
    {
      const float *src = tables + (size_t)f * m_pad * 256;
      for (int e = tid; e < m_pad * 256; e += 256) tab[e] = src[e];
    }
    __syncthreads();
    if (s >= nseg) continue;
    WaveList wl;
    wl.init();
    int cnt = 0;
    if (start_v) {
      cnt = start_c[f];
      if (lane < cnt) { wl.v = start_v[(size_t)f * K + lane]; wl.i = -1; }   // earlier rows: sort before any tie
      if (cnt >= K) { wl.tau = readlane_f(wl.v, K - 1); wl.tau_i = -1; }
    }
    // pending candidates of this wave: lane j holds the j-th; flushed with one atomic per 64
    float pv = 0.f;
    int pi = 0, npend = 0;
    auto flush = [&]() {
      int base = 0;
      if (lane == 0) base = atomicAdd(&evcnt[f], npend);
      base = readlane_i(base, 0);
      if (lane < npend && base + lane < pool) {
        evv[(size_t)f * pool + base + lane] = pv;
        evi[(size_t)f * pool + base + lane] = pi;
      }
      npend = 0;
    };
    const Word *cw = reinterpret_cast<const Word *>(codes);
    const int rb0 = rb_lo + s * rb_per_seg;
    const int rb1 = min(rb_stop, rb0 + rb_per_seg);
    Word w_first{};
    if (rb0 < rb1) w_first = cw[((size_t)rb0 * ng) * 64 + lane];
    for (int rb = rb0; rb < rb1; rb++) {
      float acc = 0.f;
      Word w0 = w_first;
      if (rb + 1 < rb1) w_first = cw[((size_t)(rb + 1) * ng) * 64 + lane];
      for (int g = 0; g < ng; g++) {
        Word w = g == 0 ? w0 : cw[((size_t)rb * ng + g) * 64 + lane];
        const float *tj = tab + g * VEC * 256;
#pragma unroll
        for (int b = 0; b < VEC; b++) acc += tj[b * 256 + rp_byte(w, b)];
      }
      const int row = rb * 64 + lane;
      const bool valid = row >= row_from && row < row_until;
      unsigned long long mk = __ballot(valid && (cnt < K || acc < wl.tau));
      while (mk) {
        int l = __ffsll((long long)mk) - 1;
        mk &= mk - 1;
        float cv = readlane_f(acc, l);
        int cr = rb * 64 + l + row_base;
        if (cnt < K || cv < wl.tau) {   // TopKHeap.update: not full, or root > v (strict)
          if (lane == npend) { pv = cv; pi = cr; }
          if (++npend == 64) flush();
          wl.insert(cv, cr, K, lane);
          if (cnt < K) cnt++;
        }
      }
    }
    if (npend) flush();
    if (out_v) {
      const size_t fs = (size_t)f * nseg + s;
      if (lane < K) { out_v[fs * K + lane] = wl.v; out_i[fs * K + lane] = wl.i; }
      if (lane == 0) out_c[fs] = cnt;
    }
  }
}

// The same segment scan over the 16-bit codes of a wide index (k > 256, wide.hip): codes [n/64][m][64], the
// flagged query's table [m][k] gathered from global memory (L2) -- the reference's j-ordered unfused sum.
__global__ __launch_bounds__(256) void rp_scan_wide(const uint16_t *__restrict__ codes, int m, int k,
                                                    const float *__restrict__ tables /*[f][m][k]*/,
                                                    const int *__restrict__ count, int maxf, int row_from, int row_until,
                                                    int row_base, int rb_lo, int rb_hi, int rb_per_seg, int nseg, int K,
                                                    const float *__restrict__ start_v, const int *__restrict__ start_c,
                                                    float *__restrict__ out_v, int *__restrict__ out_i,
                                                    int *__restrict__ out_c, int pool, float *__restrict__ evv,
                                                    int *__restrict__ evi, int *__restrict__ evcnt) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int s = blockIdx.x * 4 + wave;
  const int nf = min(*count, maxf);
  if (s >= nseg) return;
  for (int f = blockIdx.y; f < nf; f += gridDim.y) {
    const float *T = tables + (size_t)f * m * k;
    WaveList wl;
    wl.init();
    int cnt = 0;
    if (start_v) {
      cnt = start_c[f];
      if (lane < cnt) { wl.v = start_v[(size_t)f * K + lane]; wl.i = -1; }
      if (cnt >= K) { wl.tau = readlane_f(wl.v, K - 1); wl.tau_i = -1; }
    }
    float pv = 0.f;
    int pi = 0, npend = 0;
    auto flush = [&]() {
      int base = 0;
      if (lane == 0) base = atomicAdd(&evcnt[f], npend);
      base = readlane_i(base, 0);
      if (lane < npend && base + lane < pool) {
        evv[(size_t)f * pool + base + lane] = pv;
        evi[(size_t)f * pool + base + lane] = pi;
      }
      npend = 0;
    };
    const int rb0 = rb_lo + s * rb_per_seg;
    const int rb1 = min(rb_hi, rb0 + rb_per_seg);
    for (int rb = rb0; rb < rb1; rb++) {
      const uint16_t *p = codes + (size_t)rb * m * 64 + lane;
      float acc = 0.f;
      for (int j = 0; j < m; j++) acc += T[(size_t)j * k + p[(size_t)j * 64]];
      const int row = rb * 64 + lane;
      const bool valid = row >= row_from && row < row_until;
      unsigned long long mk = __ballot(valid && (cnt < K || acc < wl.tau));
      while (mk) {
        int l = __ffsll((long long)mk) - 1;
        mk &= mk - 1;
        float cv = readlane_f(acc, l);
        int cr = rb * 64 + l + row_base;
        if (cnt < K || cv < wl.tau) {   // TopKHeap.update: not full, or root > v (strict)
          if (lane == npend) { pv = cv; pi = cr; }
          if (++npend == 64) flush();
          wl.insert(cv, cr, K, lane);
          if (cnt < K) cnt++;
        }
      }
    }
    if (npend) flush();
    if (out_v) {
      const size_t fs = (size_t)f * nseg + s;
      if (lane < K) { out_v[fs * K + lane] = wl.v; out_i[fs * K + lane] = wl.i; }
      if (lane == 0) out_c[fs] = cnt;
    }
  }
}

// K smallest distances of (start list) + (the new entries of nseg segment lists); one wave per query.
__global__ __launch_bounds__(64) void rp_merge(const float *__restrict__ start_v, const int *__restrict__ start_c,
                                               const float *__restrict__ seg_v, const int *__restrict__ seg_i,
                                               const int *__restrict__ count, int maxf, int nseg, int K,
                                               float *__restrict__ out_v, int *__restrict__ out_c) {
  const int lane = threadIdx.x;
  const int nf = min(*count, maxf);
  for (int f = blockIdx.x; f < nf; f += gridDim.x) {
    WaveList wl;
    wl.init();
    int cnt = start_c[f];
    if (lane < cnt) { wl.v = start_v[(size_t)f * K + lane]; wl.i = -1; }
    if (cnt >= K) { wl.tau = readlane_f(wl.v, K - 1); wl.tau_i = -1; }
    for (int s = 0; s < nseg; s++) {
      const size_t fs = (size_t)f * nseg + s;
      const float cur_v = lane < K ? seg_v[fs * K + lane] : INFINITY;
      const int cur_i = lane < K ? seg_i[fs * K + lane] : INT_MAX;
      for (int e = 0; e < K; e++) {
        const float cv = readlane_f(cur_v, e);
        const int ci = readlane_i(cur_i, e);
        if (ci == INT_MAX) break;          // padding: the list is shorter than K
        if (ci == -1) continue;            // inherited from the start list
        if (cnt < K || cv < wl.tau) {
          wl.insert(cv, s * 64 + e, K, lane);
          if (cnt < K) cnt++;
        } else {
          break;                           // ascending: nothing smaller follows
        }
      }
    }
    if (lane < K) out_v[(size_t)f * K + lane] = wl.v;
    if (lane == 0) out_c[f] = cnt;
  }
}

// The K smallest distances among a flagged query's candidates so far (the bound of the long level) when level 1 went
// through the quantized filter: every row of the earlier levels that can be among them is in the pool -- level 0's scan
// pushes every row that entered its running list, the filter emits every level-1 row below level 0's K-th distance.
// One wave per query.  A pool that overflowed gives no bound (the query's replay is abandoned anyway).
__global__ __launch_bounds__(64) void rp_prefix_pool(const float *__restrict__ evv, const int *__restrict__ evcnt, int pool,
                                                     const int *__restrict__ count, int maxf, int K,
                                                     float *__restrict__ out_v, int *__restrict__ out_c) {
  const int lane = threadIdx.x;
  const int nf = min(*count, maxf);
  for (int f = blockIdx.x; f < nf; f += gridDim.x) {
    const int total = evcnt[f];
    const int n = total > pool ? 0 : total;
    WaveList wl;
    wl.init();
    int cnt = 0;
    for (int e0 = 0; e0 < n; e0 += 64) {
      const bool valid = e0 + lane < n;
      const float v = valid ? evv[(size_t)f * pool + e0 + lane] : INFINITY;
      unsigned long long mk = __ballot(valid && (cnt < K || v < wl.tau));
      while (mk) {
        const int l = __ffsll((long long)mk) - 1;
        mk &= mk - 1;
        const float cv = readlane_f(v, l);
        if (cnt < K || cv < wl.tau) {
          wl.insert(cv, e0 + l, K, lane);
          if (cnt < K) cnt++;
        }
      }
    }
    if (lane < K) out_v[(size_t)f * K + lane] = wl.v;
    if (lane == 0) out_c[f] = cnt;
  }
}

// literal TopKHeap (TopKHeap.scala) over the candidates in row order, then Result.fromHeap
// (Index.scala:83-94).  One workgroup per flagged query:
//   1. the candidate pool is sorted by row id (bitonic, LDS);
//   2. wave 0 walks it with a running K-smallest list and keeps only the rows that really
//      insert (distance below the K-th smallest so far) -- the rest would be rejected by the heap;
//   3. wave 0 pushes those through the heap (kept in registers, lane = slot) and drains it.
constexpr int RP_KEEP = 2048;
__global__ __launch_bounds__(256) void rp_heap(const int *__restrict__ packs, int lists, long long stride_words, int F,
                                               int C, int lds_pool, int K, int *__restrict__ out_idx,
                                               float *__restrict__ out_dist, int *__restrict__ out_count,
                                               int *__restrict__ out_flags, unsigned long long *__restrict__ dbg,
                                               int min_total /* queries with fewer candidates belong to another launch */) {
  extern __shared__ float rp_lds[];
  float *cv = rp_lds;                                          // [lds_pool] candidate distances
  int *ci = reinterpret_cast<int *>(rp_lds + lds_pool);        // [lds_pool] candidate rows
  float *sv = rp_lds + 2 * lds_pool;                           // [RP_KEEP] inserting rows, in row order
  int *si = reinterpret_cast<int *>(rp_lds + 2 * lds_pool + RP_KEEP);
  __shared__ int nkeep;
  const int tid = threadIdx.x;
  const Pack p0{const_cast<int *>(packs), F, C};
  const int nf = min(p0.count()[0], F);
  for (int f = blockIdx.x; f < nf; f += gridDim.x) {
    __syncthreads();
    // candidates of query f from every shard (list): concatenate, then sort by row id
    int total = 0;
    bool ok = true;
    for (int l = 0; l < lists; l++) {
      const Pack pk{const_cast<int *>(packs) + (size_t)l * stride_words, F, C};
      const int c = pk.evcnt()[f];
      ok = ok && c <= C;
      total += min(c, C);
    }
    if (!ok || total > lds_pool || total < min_total) continue;   // too many candidates: keeps the (distance, row id) result and its flags
    if (dbg && tid == 0 && f == 0) dbg[0] = wall_clock64();
    int n2 = 64;
    while (n2 < total) n2 <<= 1;
    {
      int off = 0;
      for (int l = 0; l < lists; l++) {
        const Pack pk{const_cast<int *>(packs) + (size_t)l * stride_words, F, C};
        const int c = pk.evcnt()[f];
        for (int e = tid; e < c; e += 256) {
          ci[off + e] = pk.evi()[(size_t)f * C + e];
          cv[off + e] = pk.evv()[(size_t)f * C + e];
        }
        off += c;
      }
      for (int e = total + tid; e < n2; e += 256) { ci[e] = INT_MAX; cv[e] = INFINITY; }
    }
    __syncthreads();
    if (dbg && tid == 0 && f == 0) dbg[1] = wall_clock64();
    for (int k = 2; k <= n2; k <<= 1)
      for (int j = k >> 1; j >= 1; j >>= 1) {
        for (int i = tid; i < n2; i += 256) {
          const int l = i ^ j;
          if (l > i) {
            const int a = ci[i], b = ci[l];
            if ((a > b) == ((i & k) == 0)) {
              const float fa = cv[i], fb = cv[l];
              ci[i] = b; ci[l] = a; cv[i] = fb; cv[l] = fa;
            }
          }
        }
        __syncthreads();
      }
    if (dbg && tid == 0 && f == 0) dbg[2] = wall_clock64();
    if (tid < 64) {
      const int lane = tid;
      WaveList wl;
      wl.init();
      int cnt = 0, nk = 0;
      for (int base = 0; base < total; base += 64) {
        const bool have = base + lane < total;
        const float v = have ? cv[base + lane] : INFINITY;
        const int r = have ? ci[base + lane] : INT_MAX;
        unsigned long long mk = __ballot(have && (cnt < K || v < wl.tau));
        while (mk) {
          const int l = __ffsll((long long)mk) - 1;
          mk &= mk - 1;
          const float x = readlane_f(v, l);
          const int xr = readlane_i(r, l);
          if (cnt < K || x < wl.tau) {   // TopKHeap.update inserts: not full, or root > x (strict)
            if (lane == 0 && nk < RP_KEEP) { sv[nk] = x; si[nk] = xr; }
            nk++;
            wl.insert(x, xr, K, lane);
            if (cnt < K) cnt++;
          }
        }
      }
      if (lane == 0) nkeep = nk;
    }
    __syncthreads();
    const int kept = nkeep;
    if (dbg && tid == 0 && f == 0) dbg[5] = kept;
    if (kept > RP_KEEP) continue;
    if (dbg && tid == 0 && f == 0) dbg[3] = wall_clock64();
    if (tid < 64) {
      // the heap lives in registers: lane i = slot i (K <= 63); every index below is wave-uniform
      const int lane = tid;
      float hv = 0.f;
      int hk = 0;
      int size = 0;
      // percolateDown from the root with every lane working (grouped.hip's RegHeap::down_root): lane l decides from its
      // own two children where an entry of value `cur` standing at slot l would go next -- the reference's two
      // comparisons in its order --, the path from the root is a chase through those answers and the slots on it take
      // their chosen child's entry at once
      auto down_root = [&](float cur, int curk) {
        const int lc = 2 * lane + 1, rc = 2 * lane + 2;
        const float a0 = __int_as_float(__builtin_amdgcn_ds_bpermute(4 * (lc & 63), __float_as_int(hv)));
        const float b0 = __int_as_float(__builtin_amdgcn_ds_bpermute(4 * (rc & 63), __float_as_int(hv)));
        int nxt = -1;
        float nv = cur;
        if (lc < size && nv < a0) { nv = a0; nxt = lc; }
        if (rc < size && nv < b0) { nv = b0; nxt = rc; }
        const int kc = __builtin_amdgcn_ds_bpermute(4 * (max(nxt, 0) & 63), hk);
        unsigned long long path = 0ull;
        int node = 0;
        for (;;) {
          const int n2 = readlane_i(nxt, node);
          if (n2 < 0) break;
          path |= 1ull << node;
          node = n2;
        }
        if ((path >> lane) & 1ull) { hv = nv; hk = kc; }
        if (lane == node) { hv = cur; hk = curk; }
      };
      auto del = [&]() {                                        // delete, TopKHeap.scala:57-67
        size -= 1;
        const float lv = readlane_f(hv, size);
        const int lk = readlane_i(hk, size);
        down_root(lv, lk);
      };
      for (int base = 0; base < kept; base += 64) {
        const float ev = base + lane < kept ? sv[base + lane] : 0.f;
        const int ek = base + lane < kept ? si[base + lane] : 0;
        const int ne = min(64, kept - base);
        for (int e = 0; e < ne; e++) {
          const float v = readlane_f(ev, e);
          const int kk = readlane_i(ek, e);
          if (size == K && readlane_f(hv, 0) > v) del();        // update, TopKHeap.scala:69-79
          if (size < K) {
            int i = size;
            while (i > 0) {                                     // percolateUp, TopKHeap.scala:21-28
              const int p = (i - 1) / 2;
              const float pv = readlane_f(hv, p);
              if (v > pv) {
                const int pk = readlane_i(hk, p);
                if (lane == i) { hv = pv; hk = pk; }
                i = p;
              } else break;
            }
            if (lane == i) { hv = v; hk = kk; }
            size += 1;
          }
        }
      }
      const int q = p0.list()[f];
      const int live = size;
      for (int i = live - 1; i >= 0; i--) {                     // Result.fromHeap: max first, fill from the back
        const float tv = readlane_f(hv, 0);
        const int tk = readlane_i(hk, 0);
        if (lane == 0) { out_idx[(size_t)q * K + i] = tk; out_dist[(size_t)q * K + i] = tv; }
        del();
      }
      if (lane >= live && lane < K) { out_idx[(size_t)q * K + lane] = -1; out_dist[(size_t)q * K + lane] = INFINITY; }
      if (lane == 0) {
        if (out_count) out_count[q] = live;
        out_flags[q] |= GULON_FLAG_EXACT_REPLAY;
        if (dbg && f == 0) dbg[4] = wall_clock64();
      }
    }
  }
}

// Candidates of the flagged queries over rows [from, until) of this index (cold start at `from`).
// The long level without a scan.  If the K-th smallest distance of the earlier rows (the bound level 2 starts from) IS
// the K-th smallest distance of the whole range -- what the main pass returned -- then every later row that can
// still insert (distance strictly below the bound; TopKHeap.update rejects an equal one) is one of the fewer than K
// rows below the final K-th distance, and all of those are in the main pass's list.  That is the all-ties case of the
// reference-shaped data (thousands of rows share the query's code: the first K of them fill the heap within the
// first levels); the query's level-2 candidates are then read off its result.  done[f] = 2 / scanme[f] = 0 for
// such a query (1 where the scan still has to walk).
__global__ void rp_shortcut(const int *__restrict__ count, const int *__restrict__ list, int F, int K,
                            const float *__restrict__ prefix_v, const int *__restrict__ prefix_c,
                            const float *__restrict__ fin_d, const int *__restrict__ fin_i, const int *__restrict__ fin_c,
                            int first_row /* global id of level 2's first row */, int pool, float *__restrict__ evv,
                            int *__restrict__ evi, int *__restrict__ evcnt, int *__restrict__ done,
                            int *__restrict__ scanme, int *__restrict__ rlast /* last global row id that can insert */) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= F) return;
  const int nf = min(*count, F);
  bool hit = false;
  {   // no row after the last of the main pass's K best can insert: all K are in the heap by then, its root is at most
      // the final K-th distance, and a row strictly below that distance is one of the K best itself
    int last = INT_MAX;
    if (f < nf) {
      const int q = list[f];
      if (q >= 0 && fin_c[q] == K) {
        last = 0;
        for (int i = 0; i < K; i++) last = max(last, fin_i[(size_t)q * K + i]);
      }
    }
    rlast[f] = last;
  }
  if (f < nf && prefix_c[f] >= K) {
    const int q = list[f];
    const float bound = prefix_v[(size_t)f * K + K - 1];
    if (q >= 0 && fin_c[q] == K && bound < INFINITY &&
        __float_as_uint(bound) == __float_as_uint(fin_d[(size_t)q * K + K - 1])) {
      hit = true;
      int n = 0;
      for (int i = 0; i < K; i++)
        if (fin_d[(size_t)q * K + i] < bound && fin_i[(size_t)q * K + i] >= first_row) n++;
      int pos = n > 0 ? atomicAdd(&evcnt[f], n) : 0;
      for (int i = 0; i < K; i++) {
        const float d = fin_d[(size_t)q * K + i];
        const int r = fin_i[(size_t)q * K + i];
        if (d < bound && r >= first_row) {
          if (pos < pool) { evv[(size_t)f * pool + pos] = d; evi[(size_t)f * pool + pos] = r; }
          pos++;
        }
      }
    }
  }
  done[f] = hit ? 2 : 0;
  scanme[f] = hit ? 0 : 1;
}

void replay_collect(gulon_index *ix, const float *dQ, int B, int K, int from, int until, const int *d_flags, int F,
                    int C, int *pack, hipStream_t st, int skip = 0, const float *fin_d = nullptr,
                    const int *fin_i = nullptr, const int *fin_c = nullptr /* the main pass's [B][K] result, if at hand */) {
  const Pack pk{pack, F, C};
  // how many queries recent batches of this handle had flagged: read without synchronisation (the word lags by the
  // batches in flight), it only decides whether the long level is worth the quantized filter's launches -- a batch
  // with one flagged query (the usual case) pays ~80 us for them, a batch in which every query ties saves 25 ms
  if (!ix->rp_hint_h) {
    HIP_CHECK(hipHostMalloc(reinterpret_cast<void **>(&ix->rp_hint_h), sizeof(int), hipHostMallocMapped));
    *ix->rp_hint_h = 0;
    HIP_CHECK(hipHostGetDevicePointer(reinterpret_cast<void **>(&ix->rp_hint_d), ix->rp_hint_h, 0));
  }
  const int recently_flagged = *reinterpret_cast<volatile int *>(ix->rp_hint_h);
  hipLaunchKernelGGL(rp_collect, dim3(1), dim3(64), 0, st, d_flags, B, pk, skip, ix->rp_hint_d);
  HIP_CHECK(hipGetLastError());
  if (until <= from) return;
  const int rb_begin = from / 64, rb_end = ceil_div(until, 64), rb_total = rb_end - rb_begin;
  // y extent of the scan grids (a block walks the flagged queries y, y + Y, ...): the two short levels have few
  // segments, so every flagged query gets blocks of its own once the handle has seen batches with many of them
  // (rp_hint, below) -- a batch in which all 1024 queries tie spent 3.7 ms in them with 16; the long level keeps 16
  // (its x extent fills the chip; blocks without a query cost a launch slot)
  const int gy = std::min(F, 16);
  const int gy_short = recently_flagged >= 32 ? std::min(F, 256) : gy;
  // level geometry (in 64-row blocks)
  // (level 0 is one wave per query walking its blocks one after the other: short where a batch has a flagged query or
  // two -- its latency is the replay's -- and four times as long where most of the batch ties: level 1's bound, the K-th
  // distance of level 0's rows, then lets a quarter as many rows through the filter)
  const int l0 = std::min(rb_total, recently_flagged >= 32 ? 4 * RP_L0_BLOCKS : RP_L0_BLOCKS);
  const int l1 = std::min(rb_total - l0, RP_L1_BLOCKS);
  const int l2 = rb_total - l0 - l1;
  const int segs1 = ceil_div(l1, RP_L1_SEG);
  const int per2 = std::max(RP_L2_MIN, ceil_div(l2, RP_L2_SEGS));
  const int segs2 = ceil_div(l2, per2);
  ix->rp_q.ensure((size_t)F * ix->d);
  ix->rp_tables.ensure(ix->wide ? (size_t)F * ix->m * ix->k : (size_t)F * ix->m_pad * 256);
  ix->rp_segtop.ensure((size_t)F * std::max(segs1, 1) * K); ix->rp_segi.ensure((size_t)F * std::max(segs1, 1) * K);
  ix->rp_segcnt.ensure((size_t)F * std::max(segs1, 1));
  ix->rp_l0v.ensure((size_t)F * K); ix->rp_l0i.ensure((size_t)F * K); ix->rp_l0c.ensure(F);
  ix->rp_prefix.ensure((size_t)F * K); ix->rp_precnt.ensure(F);
  ix->rp_mins.ensure((size_t)(F + 16) * std::max(ix->m_pad, ix->m));
  ix->rp_done.ensure((size_t)4 * F);
  int *rlast_all = fin_i && !ix->wide ? ix->rp_done.p + 2 * F : nullptr;   // (rp_shortcut writes the same values again)
  hipLaunchKernelGGL(rp_gather_queries, dim3(ceil_div((long long)F * ix->d, 256)), dim3(256), 0, st, dQ, ix->d, F,
                     pk.list(), pk.count(), ix->rp_q.p, K, fin_i, fin_c, rlast_all);
  if (ix->wide)   // (tables of all F slots: the flagged-query count stays on the device)
    launch_build_tables_wide(ix->cents.p, ix->from.p, ix->sdim.p, ix->d, ix->m, ix->k, ix->rp_q.p, 0, F, ix->rp_tables.p, st);
  else
    launch_build_tables(1, ix, ix->rp_q.p, F, F, ix->rp_tables.p, st, pk.count(), ix->rp_mins.p);
  const size_t lds = (size_t)ix->m_pad * 256 * sizeof(float);
  auto scan = [&](int rb_lo, int rb_hi, int per_seg, int nseg, const float *sv, const int *sc, float *ov, int *oi,
                  int *oc, int gy, const int *only) {
    if (nseg <= 0) return;
    if (ix->wide) {
      hipLaunchKernelGGL(rp_scan_wide, dim3(ceil_div(nseg, 4), gy), dim3(256), 0, st, ix->wcodes.p, ix->m, ix->k,
                         ix->rp_tables.p, pk.count(), F, from, until, ix->row_base, rb_lo, rb_hi, per_seg, nseg, K, sv, sc,
                         ov, oi, oc, C, pk.evv(), pk.evi(), pk.evcnt());
      HIP_CHECK(hipGetLastError());
      return;
    }
    auto kern = ix->vec == 16 ? rp_scan<16> : rp_scan<4>;
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds));
    hipLaunchKernelGGL(kern, dim3(ceil_div(nseg, 4), gy), dim3(256), lds, st, ix->codes.p, ix->ng, ix->m_pad,
                       ix->rp_tables.p, pk.count(), F, from, until, ix->row_base, rb_lo, rb_hi, per_seg, nseg, K, sv, sc,
                       ov, oi, oc, C, pk.evv(), pk.evi(), pk.evcnt(), only, rlast_all);
    HIP_CHECK(hipGetLastError());
  };
  // level 0: cold start over the first rows -> their K smallest distances
  scan(rb_begin, rb_begin + l0, l0, 1, nullptr, nullptr, ix->rp_l0v.p, ix->rp_l0i.p, ix->rp_l0c.p, gy_short, nullptr);
  if (segs1 > 0) {
    // Level 1 (the next 131 K rows) tightens the bound for the long level.  As an exact fp32 scan it was the replay's
    // largest kernel when every query of a batch ties (0.6 ms of LDS gathers); with level 0's K-th distance as a fixed
    // bound it is a threshold query over 2048 row blocks -- 1 % of a main stage for the quantized filter -- and the long
    // level's bound is then read off the candidate pool.  Batches with few flagged queries keep the segment scan.
    int *only1 = nullptr;
    bool l1_filtered = false;
    static const bool l1_off = [] { const char *e = getenv("GULON_REPLAY_L1_FILTER"); return e && atoi(e) == 0; }();
    if (!l1_off && recently_flagged >= 32 && segs2 > 0 && !ix->wide)
      l1_filtered = replay_level2_filtered(ix, F, K, rb_begin + l0, rb_begin + l0 + l1, from,
                                           std::min(until, (rb_begin + l0 + l1) * 64), ix->rp_tables.p, ix->rp_mins.p,
                                           ix->rp_l0v.p, ix->rp_l0c.p, pk.count(), pk.evv(), pk.evi(), pk.evcnt(), C, &only1, st,
                                           nullptr, rlast_all, ix->rp_done.p + 3 * F);
    if (l1_filtered)   // (the queries the filter left over: overflowing queues, no bound yet)
      scan(rb_begin + l0, rb_begin + l0 + l1, RP_L1_SEG, segs1, ix->rp_l0v.p, ix->rp_l0c.p, nullptr, nullptr, nullptr, gy_short,
           only1);
    else
      scan(rb_begin + l0, rb_begin + l0 + l1, RP_L1_SEG, segs1, ix->rp_l0v.p, ix->rp_l0c.p, ix->rp_segtop.p,
           ix->rp_segi.p, ix->rp_segcnt.p, gy_short, nullptr);
    if (segs2 > 0) {
      if (l1_filtered)
        hipLaunchKernelGGL(rp_prefix_pool, dim3(std::min(F, 1024)), dim3(64), 0, st, pk.evv(), pk.evcnt(), C, pk.count(), F, K,
                           ix->rp_prefix.p, ix->rp_precnt.p);
      else
        hipLaunchKernelGGL(rp_merge, dim3(std::min(F, 1024)), dim3(64), 0, st, ix->rp_l0v.p, ix->rp_l0c.p, ix->rp_segtop.p,
                           ix->rp_segi.p, pk.count(), F, segs1, K, ix->rp_prefix.p, ix->rp_precnt.p);
      HIP_CHECK(hipGetLastError());
      // The long level.  Its segment scans are an exact fp32 scan of (nearly) all rows per flagged query, one query's
      // table per workgroup: 27 us per query, 28 ms when all 1024 queries of a batch tie (the reference-shaped data).
      // With the K-th distance of the earlier rows as a FIXED bound the rows that can insert are a threshold query
      // -- what the quantized filter answers sixteen queries at a time (the candidates only have to be a superset:
      // the literal heap applies TopKHeap's own test, in row order); batches with enough flagged queries take that
      // road, and the segment scan below only walks the queries left to it (`only`).
      int *only = nullptr;
      const int *done = nullptr, *rlast = nullptr;
      if (fin_d && !ix->wide) {
        hipLaunchKernelGGL(rp_shortcut, dim3(ceil_div(F, 256)), dim3(256), 0, st, pk.count(), pk.list(), F, K, ix->rp_prefix.p,
                           ix->rp_precnt.p, fin_d, fin_i, fin_c, (rb_begin + l0 + l1) * 64 + ix->row_base, C, pk.evv(), pk.evi(),
                           pk.evcnt(), ix->rp_done.p, ix->rp_done.p + F, ix->rp_done.p + 2 * F);
        HIP_CHECK(hipGetLastError());
        done = ix->rp_done.p;
        only = ix->rp_done.p + F;
        rlast = ix->rp_done.p + 2 * F;
      }
      if (recently_flagged >= 32)
        replay_level2_filtered(ix, F, K, rb_begin + l0 + l1, rb_end, from, until, ix->rp_tables.p, ix->rp_mins.p,
                             ix->rp_prefix.p, ix->rp_precnt.p, pk.count(), pk.evv(), pk.evi(), pk.evcnt(), C, &only, st, done,
                             rlast, ix->rp_done.p + F);
      scan(rb_begin + l0 + l1, rb_end, per2, segs2, ix->rp_prefix.p, ix->rp_precnt.p, nullptr, nullptr, nullptr, gy, only);
    }
  }
  if (getenv("GULON_REPLAY_STATS")) {   // debugging aid: synchronous candidate counts
    HIP_CHECK(hipStreamSynchronize(st));
    std::vector<int> h(4 + 2 * (size_t)F);
    HIP_CHECK(hipMemcpy(h.data(), pack, sizeof(int) * h.size(), hipMemcpyDeviceToHost));
    fprintf(stderr, "[replay] %d flagged queries; levels %d + %d x %d + %d x %d blocks; candidates:", h[0], l0, segs1,
            RP_L1_SEG, segs2, per2);
    for (int i = 0; i < h[0] && i < 16; i++) fprintf(stderr, " %d", h[4 + F + i]);
    if (fin_d && !ix->wide && segs2 > 0) {
      std::vector<int> dn((size_t)F);
      HIP_CHECK(hipMemcpy(dn.data(), ix->rp_done.p, sizeof(int) * (size_t)F, hipMemcpyDeviceToHost));
      int hits = 0;
      for (int i = 0; i < h[0]; i++) hits += dn[i] != 0;
      fprintf(stderr, "; long level read off the result for %d of them", hits);
      std::vector<int> rl((size_t)F);
      HIP_CHECK(hipMemcpy(rl.data(), ix->rp_done.p + 2 * F, sizeof(int) * (size_t)F, hipMemcpyDeviceToHost));
      double sum_q = 0, sum_t = 0; int nq = 0, nt = 0;
      for (int t0 = 0; t0 < h[0]; t0 += 16) {
        int tmax = -1;
        for (int i = t0; i < std::min(h[0], t0 + 16); i++) {
          if (dn[i]) continue;
          const double fr = rl[i] == INT_MAX ? 1.0 : std::min(1.0, std::max(0.0, (double)(rl[i] - ix->row_base - from) / (double)(until - from)));
          sum_q += fr; nq++;
          tmax = std::max(tmax, (int)(fr * 1000));
        }
        if (tmax >= 0) { sum_t += tmax / 1000.0; nt++; }
      }
      fprintf(stderr, "; last inserting row: mean %.3f of the range over %d queries, mean of the 16-query tile maxima %.3f over %d tiles",
              nq ? sum_q / nq : 0.0, nq, nt ? sum_t / nt : 0.0, nt);
      if (recently_flagged >= 32 && ix->rp_fb.n >= (size_t)F + F / 16 && ix->vec == 16 && ix->ng == 1) {   // (tiles of 16 queries)
        std::vector<int> lim((size_t)F / 16);
        HIP_CHECK(hipMemcpy(lim.data(), ix->rp_fb.p + F, sizeof(int) * lim.size(), hipMemcpyDeviceToHost));
        double cover = 0;
        for (int l : lim) cover += std::max(0.0, std::min(1.0, (double)(l - (rb_begin + l0 + l1) + 1) / (double)std::max(1, l2)));
        fprintf(stderr, "; in the served order the tiles walk %.3f of (tiles x long-level rows)", cover / std::max<size_t>(1, lim.size()));
      }
    }
    fprintf(stderr, "\n");
  }
}

// Literal heap over the candidates of `lists` packs (one per row shard, same flagged-query list).
void replay_apply(const int *packs, int lists, long long stride_words, int F, int C, int K, int *d_oi, float *d_od,
                  int *d_oc, int *d_of, unsigned long long *dbgp, hipStream_t st, bool many_flagged = false) {
  int lds_pool = 64;   // a power of two: the bitonic sort pads the candidates to one
  while (lds_pool < RP_LDS_POOL && lds_pool < (long long)lists * C) lds_pool <<= 1;
  const size_t heap_lds = (size_t)(2 * lds_pool + 2 * RP_KEEP) * sizeof(float);
  HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(rp_heap), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)heap_lds));
  // a block per flagged query only when the handle has recently seen many of them: every block wants 80+ KiB of LDS,
  // which the other batch's filter kernel does not leave free -- 1024 blocks waiting for it took 1.2 ms for the
  // usual single flagged query (64 blocks: 0.29 ms)
  // ... and then most of them have few candidates: those go first, with a pool of 2048 -- 32 KiB of LDS, every block
  // resident at once (0.51 -> 0.1 ms for a batch of 1024 tied queries) -- and the large launch takes what is left
  int min_total = 0;
  if (many_flagged && lds_pool > 2048) {
    const size_t small_lds = (size_t)(2 * 2048 + 2 * RP_KEEP) * sizeof(float);
    hipLaunchKernelGGL(rp_heap, dim3(std::min(F, 1024)), dim3(256), small_lds, st, packs, lists, stride_words, F, C, 2048, K, d_oi,
                       d_od, d_oc, d_of, dbgp, 0);
    min_total = 2049;
  }
  hipLaunchKernelGGL(rp_heap, dim3(std::min(F, many_flagged ? 1024 : 64)), dim3(256), heap_lds, st, packs, lists, stride_words, F, C, lds_pool,
                     K, d_oi, d_od, d_oc, d_of, dbgp, min_total);
  HIP_CHECK(hipGetLastError());
  if (dbgp) {
    HIP_CHECK(hipStreamSynchronize(st));
    unsigned long long h[6];
    HIP_CHECK(hipMemcpy(h, dbgp, sizeof(h), hipMemcpyDeviceToHost));
    fprintf(stderr, "[replay] rp_heap phases (us): load %.1f sort %.1f events %.1f heap %.1f; inserting rows %llu\n",
            (h[1] - h[0]) / 100.0, (h[2] - h[1]) / 100.0, (h[3] - h[2]) / 100.0, (h[4] - h[3]) / 100.0, h[5]);
  }
}

void run_tie_replay(gulon_index *ix, const float *dQ, int B, int K, int from, int until, int *d_oi, float *d_od,
                    int *d_oc, int *d_of, hipStream_t st) {
  if (B <= 0 || K <= 0 || until <= from) return;
  // wide codes: a flagged query's table is m * k floats in HBM -- at most 256 MiB of them per round, and as many
  // rounds as the batch could need (the flagged-query count stays on the device: later rounds find nothing to do)
  int F = std::min(B, RP_MAXF);
  if (ix->wide) F = (int)std::max<size_t>(1, std::min<size_t>((size_t)F, (256ull << 20) / ((size_t)ix->m * ix->k * sizeof(float))));
  const int C = RP_POOL;
  ix->rp_pack.ensure(replay_pack_words(F, C));
  unsigned long long *dbgp = nullptr;
  if (getenv("GULON_REPLAY_STATS")) { ix->dbg.ensure(8); dbgp = ix->dbg.p; }
  for (int skip = 0; skip < B; skip += F) {
    replay_collect(ix, dQ, B, K, from, until, d_of, F, C, ix->rp_pack.p, st, skip, d_od, d_oi, d_oc);
    replay_apply(ix->rp_pack.p, 1, 0, F, C, K, d_oi, d_od, d_oc, d_of, dbgp, st,
                 ix->rp_hint_h && *reinterpret_cast<volatile int *>(ix->rp_hint_h) >= 32);
  }   // (byte codes: F = min(B, 1024) -- one round unless the batch is larger than that)
}

}  // namespace gulon

using namespace gulon;

// ---- row-sharded replay (one pack per shard, exchanged by the caller) ------------------------
namespace {
void check_pack_shape(int F, int C) {
  GULON_REQUIRE(F >= 1 && F <= RP_MAXF, "max_flagged must be in [1, %d]", RP_MAXF);
  GULON_REQUIRE(C >= 64 && C <= RP_POOL, "pool must be in [64, %d]", RP_POOL);
}
}  // namespace

GULON_API int64_t gulon_replay_pack_words(int32_t max_flagged, int32_t pool) {
  if (max_flagged < 1 || max_flagged > RP_MAXF || pool < 64 || pool > RP_POOL) return -1;
  return (int64_t)replay_pack_words(max_flagged, pool);
}

GULON_API int32_t gulon_index_replay_collect_dev(gulon_index *idx, const float *d_queries, int32_t b, int32_t k_nn,
                                                 int32_t from, int32_t until, const int32_t *d_flags, int32_t skip,
                                                 int32_t max_flagged, int32_t pool, int32_t *d_pack, void *stream) {
  return guarded([&] {
    GULON_REQUIRE(idx != nullptr && d_pack != nullptr && d_flags != nullptr, "null argument");
    GULON_REQUIRE(from <= until && from >= 0 && until <= idx->n, "expected: 0 <= from <= until <= length");
    GULON_REQUIRE(b >= 0 && k_nn >= 1 && skip >= 0, "bad shape");
    check_pack_shape(max_flagged, pool);
    GULON_UNSUPPORTED(k_nn > GULON_MAX_K, "k_nn = %d > GULON_MAX_K = %d", k_nn, GULON_MAX_K);
    std::lock_guard<std::mutex> lock(idx->mu);
    StreamOrder so(idx, (hipStream_t)stream);
    replay_collect(idx, d_queries, b, k_nn, from, until, d_flags, max_flagged, pool, d_pack, (hipStream_t)stream, skip);
    so.done();
  });
}

GULON_API int32_t gulon_replay_apply_dev(const int32_t *d_packs, int32_t lists, int32_t max_flagged, int32_t pool,
                                         int32_t b, int32_t k_nn, int32_t *d_out_idx, float *d_out_dist,
                                         int32_t *d_out_count, int32_t *d_out_flags, void *stream) {
  return guarded([&] {
    GULON_REQUIRE(d_packs != nullptr && lists >= 1 && b >= 0 && k_nn >= 1, "bad arguments");
    check_pack_shape(max_flagged, pool);
    GULON_UNSUPPORTED(k_nn > GULON_MAX_K, "k_nn = %d > GULON_MAX_K = %d", k_nn, GULON_MAX_K);
    GULON_REQUIRE(d_out_idx && d_out_dist && d_out_flags, "null output");
    replay_apply(d_packs, lists, (long long)replay_pack_words(max_flagged, pool), max_flagged, pool, k_nn, d_out_idx,
                 d_out_dist, d_out_count, d_out_flags, nullptr, (hipStream_t)stream);
  });
}
