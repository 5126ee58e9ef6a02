// GroupedIndex (Index.scala:231-308): coarse groups + product-quantized residuals.
//
//   query(k, q):  nn = searchSpace(q)                       nearest coarse centroids (TopKHeap + deleteAll)
//                 for c in nn:  residual = q - centroid(c)   MathUtils.subtract
//                               heap.merge(vectorIndex.query(k, residual, from(c), until(c)))
//                 Result.fromHeap(heap)
//
// The per-group heaps are merged with TopKHeap.merge, i.e. update() of the other heap's slots in
// ARRAY order, so under distance ties the answer depends on the literal heaps.  Kernels:
//   gq_cdist            query x centroid distances (MathUtils.distanceSq order), centroids transposed
//   gq_nearest_groups   LimitGroups(<= 63): literal TopKHeap in registers (lane = slot) + deleteAll
//   gq_sorted_groups    LimitVectors / larger limits: (distance, id) bitonic sort, cut by rows or count
//   gq_group_scan       workgroup = 4 (query, searched group) pairs, one wave each: the residual's
//                       m x k table in LDS (Index.prepareQuery arithmetic; each quantizer's codebook
//                       staged once per workgroup, transposed), then the group's rows 64 at a time
//                       (PQIndex.distances order).  Two instantiations:
//                         fast     ascending (distance, row) list of K+1 entries per pair
//                         literal  the reference's TopKHeap, fed in row order, stored in array order
//   gq_merge_fast       merges the lists; a query whose K+1 best contain equal distances (or that
//                       saw a NaN) is appended to a list ...
//   gq_group_scan<literal> + gq_merge   ... and redone literally: TopKHeap.merge of the group heaps
//                       in search order + Result.fromHeap.  Ids and order therefore equal the
//                       reference's also under ties (GULON_GROUPED_LITERAL=1: every query).
#include "scan.hpp"

#include "grouped_filter.hpp"
using gulon::DevBuf;

struct gulon_grouped_index {
  gulon_index *pq = nullptr;       // residual codes + residual codebooks (row-blocked layout of scan.hip)
  int32_t n = 0, d = 0, g = 0;
  DevBuf<float> gcent, gcent_t;    // [g][d] centroids of the non-empty groups, and the [d][g] transpose
  DevBuf<int> bounds;              // [g+1] first row of every group, then n
  DevBuf<uint16_t> codes2;         // [n/64][ceil(m/2)][64]: the codes of quantizers 2h (low byte) and 2h+1 (gq_scan_qm)
  // scratch (grown on demand under mu)
  DevBuf<float> q_dev, cdist, hv, od;
  DevBuf<int> nn, nn_cnt, hk, hs, oi, oc, qlist, qcount, sel_ok, nn_sized, lit_flag;
  DevBuf<float> wide_tables;       // k > 256: residual tables, one slot per workgroup of gq_group_scan_wide
  // approximate pre-selection (gq_approx_scan): |g + decode(codes_i)|^2 per row, its maximum, per-query tables, lists
  DevBuf<float> xnorm, ptab, apv, amv;
  DevBuf<int> api, ami, anan;
  gulon::GroupFilter gfilter;     // the same pre-selection batched by group with 8-bit bound tables (grouped_filter.hip)
  float xnmax = 0.f;
  int n_empty = 0;                 // groups without rows (the reference's leading empty group, WordVectors.scala:38-39)
  std::mutex mu;
  ~gulon_grouped_index() { if (pq) gulon_index_destroy(pq); }
};

namespace gulon {
namespace {

// TopKHeap.scala with lane = slot storage (K <= 63); every index is wave-uniform.
struct RegHeap {
  float v = 0.f;
  int k = 0;
  int size = 0;
  int cap;
  int lane;
  __device__ RegHeap(int cap_, int lane_) : cap(cap_), lane(lane_) {}
  __device__ float val(int i) const { return readlane_f(v, i); }
  __device__ int key(int i) const { return readlane_i(k, i); }
  // The reference's chains of swaps move ONE entry down (or up) the tree; here that entry travels in registers and
  // the entries it passes are shifted into the hole it leaves -- the same comparisons in the same order, the same final
  // arrangement, and three lane reads per level instead of eight (a batch's group selection with LimitGroups(50) over
  // 1001 groups is ~200 serial updates per query on one wavefront: 234 us of a 0.51 ms batch with the swaps).
  // percolateDown from the ROOT with every lane working: lane l looks at its own two children and decides where an
  // entry of value `cur` standing at slot l would go next (the reference's two comparisons, in its order); the path from
  // the root is then a chase through those answers -- one lane read per level -- and every slot on the path takes its
  // chosen child's entry at once.
  __device__ void down_root(float cur, int curk) {
    const int lc = 2 * lane + 1, rc = 2 * lane + 2;
    const float a0 = __int_as_float(__builtin_amdgcn_ds_bpermute(4 * (lc & 63), __float_as_int(v)));
    const float b0 = __int_as_float(__builtin_amdgcn_ds_bpermute(4 * (rc & 63), __float_as_int(v)));
    const bool ha = lc < size, hb = rc < size;
    // top = l; if (lc < size && val(top) < val(lc)) top = lc; if (rc < size && val(top) < val(rc)) top = rc;
    int nxt = -1;
    float nv = cur;
    if (ha && nv < a0) { nv = a0; nxt = lc; }
    if (hb && nv < b0) { nv = b0; nxt = rc; }
    const int kc = __builtin_amdgcn_ds_bpermute(4 * (max(nxt, 0) & 63), k);
    unsigned long long path = 0ull;
    int node = 0;
    for (;;) {
      const int n2 = readlane_i(nxt, node);
      if (n2 < 0) break;
      path |= 1ull << node;
      node = n2;
    }
    if ((path >> lane) & 1ull) { v = nv; k = kc; }
    if (lane == node) { v = cur; k = curk; }
  }
  __device__ void down(int i, float cur, int curk) {        // percolateDown, TopKHeap.scala:30-42; (cur, curk) = entry i
    for (;;) {
      int top = i;
      float best = cur;
      const int lc = 2 * i + 1, rc = 2 * i + 2;
      if (lc < size) { const float a = val(lc); if (best < a) { best = a; top = lc; } }
      if (rc < size) { const float b = val(rc); if (best < b) { best = b; top = rc; } }
      if (top == i) break;
      const int tk = key(top);
      if (lane == i) { v = best; k = tk; }
      i = top;
    }
    if (lane == i) { v = cur; k = curk; }
  }
  __device__ int del() {                                    // delete, TopKHeap.scala:57-67
    size -= 1;
    const int removed = key(0);
    const float lv = val(size);
    const int lk = key(size);
#ifdef GULON_REGHEAP_SERIAL_DOWN
    down(0, lv, lk);
#else
    down_root(lv, lk);
#endif
    return removed;
  }
  __device__ bool would_insert(float x) const { return size < cap || val(0) > x; }
  __device__ void update(int kk, float x) {                 // update, TopKHeap.scala:69-79
    if (size == cap && val(0) > x) del();
    if (size < cap) {
      int i = size;
      while (i > 0) {                                       // percolateUp, TopKHeap.scala:21-28
        const int p = (i - 1) / 2;
        const float pv = val(p);
        if (x > pv) {
          const int pk = key(p);
          if (lane == i) { v = pv; k = pk; }
          i = p;
        } else break;
      }
      if (lane == i) { v = x; k = kk; }
      size += 1;
    }
  }
};

// TopKHeap.scala for k_nn > 63: the arrays in LDS (one heap per wave).  Every lane runs the same wave-uniform code
// and reads the same entries; lane 0 stores.  (LDS operations of a wave execute in order.)  A fallback: Tests.scala
// asks for up to 1000 neighbours, the benchmarks for 10.
struct LdsHeap {
  volatile float *hv;
  volatile int *hk;
  int size = 0;
  int cap;
  int lane;
  __device__ LdsHeap(float *v_, int *k_, int cap_, int lane_) : hv(v_), hk(k_), cap(cap_), lane(lane_) {}
  __device__ float val(int i) const { return hv[i]; }
  __device__ int key(int i) const { return hk[i]; }
  __device__ void put(int i, int kk, float x) { if (lane == 0) { hv[i] = x; hk[i] = kk; } }
  __device__ void swp(int a, int b) {
    const float va = val(a), vb = val(b);
    const int ka = key(a), kb = key(b);
    put(a, kb, vb);
    put(b, ka, va);
  }
  __device__ void down(int i) {                             // percolateDown, TopKHeap.scala:30-42
    for (;;) {
      int top = i;
      const int lc = 2 * i + 1, rc = 2 * i + 2;
      if (lc < size && val(top) < val(lc)) top = lc;
      if (rc < size && val(top) < val(rc)) top = rc;
      if (top == i) break;
      swp(i, top);
      i = top;
    }
  }
  __device__ int del() {                                    // delete, TopKHeap.scala:57-67
    size -= 1;
    const int removed = key(0);
    put(0, key(size), val(size));
    down(0);
    return removed;
  }
  __device__ bool would_insert(float x) const { return size < cap || val(0) > x; }
  __device__ void update(int kk, float x) {                 // update, TopKHeap.scala:69-79
    if (size == cap && val(0) > x) del();
    if (size < cap) {
      put(size, kk, x);
      int i = size;
      while (i > 0) {                                       // percolateUp, TopKHeap.scala:21-28
        const int p = (i - 1) / 2;
        if (val(i) > val(p)) { swp(i, p); i = p; } else break;
      }
      size += 1;
    }
  }
};

// ---- coarse search: distances of every query to every group centroid -------------------------
// MathUtils.distanceSq(centroid, query): sum of (q_e - c_e)^2, e ascending, unfused.
// gcent_t is the [d][g] transpose: consecutive threads (centroids) read consecutive addresses.
// A thread takes one centroid and CD_Q queries: the centroid's coordinates are read once for all of them (one query
// per thread re-read the 5 MB of centroids per query: 5 GB through L2 per batch, 0.33 ms at 10 001 groups).
constexpr int CD_Q = 8;
__global__ __launch_bounds__(256) void gq_cdist(const float *__restrict__ gcent_t, int g, int d,
                                                const float *__restrict__ Q, int B, float *__restrict__ out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x, q0 = blockIdx.y * CD_Q;
  if (c >= g) return;
  const int nq = min(CD_Q, B - q0);
  float sum[CD_Q];
#pragma unroll
  for (int i = 0; i < CD_Q; i++) sum[i] = 0.f;
  for (int e = 0; e < d; e++) {
    const float ce = gcent_t[(size_t)e * g + c];
#pragma unroll
    for (int i = 0; i < CD_Q; i++) {
      const float dx = Q[(size_t)(q0 + min(i, nq - 1)) * d + e] - ce;     // (wave-uniform address: a scalar load)
      sum[i] += dx * dx;
    }
  }
#pragma unroll
  for (int i = 0; i < CD_Q; i++)
    if (i < nq) out[(size_t)(q0 + i) * g + c] = sum[i];
}

// LimitGroups(limit <= 63): literal exactNearestNeighbours(centroids, query, limit).deleteAll()
__global__ __launch_bounds__(64) void gq_nearest_groups(const float *__restrict__ cdist, int g, int limit,
                                                        int *__restrict__ nn /*[B][stride]*/, int stride,
                                                        int *__restrict__ nn_cnt) {
  const int q = blockIdx.x, lane = threadIdx.x;
  RegHeap h(limit, lane);
  const float *dq = cdist + (size_t)q * g;
  for (int base = 0; base < g; base += 64) {
    const bool have = base + lane < g;
    const float dv = have ? dq[base + lane] : 0.f;
    // centroids in index order through heap.update; the ballot only skips those the heap would reject
    unsigned long long mk = __ballot(have && h.would_insert(dv));
    while (mk) {
      const int l = __ffsll((long long)mk) - 1;
      mk &= mk - 1;
      const float x = readlane_f(dv, l);
      if (h.would_insert(x)) h.update(base + l, x);
    }
  }
  const int live = h.size;
  for (int i = live - 1; i >= 0; i--) {                      // deleteAll(): fill from the back
    const int kk = h.del();
    if (lane == 0) nn[(size_t)q * stride + i] = kk;
  }
  if (lane == 0) nn_cnt[q] = live;
}

// LimitVectors(limit) (and LimitGroups beyond 63): all centroids in ascending (distance, id)
// order -- the reference's heap order wherever no two centroid distances are equal -- cut after
// enough groups to cover `limit` rows (or after `limit` groups).  One workgroup per query, bitonic
// sort in LDS.
__global__ __launch_bounds__(256) void gq_sorted_groups(const float *__restrict__ cdist, int g, int n2,
                                                        const int *__restrict__ bounds, int by_vectors, int limit,
                                                        int *__restrict__ nn, int stride, int *__restrict__ nn_cnt,
                                                        const int *__restrict__ done, int *__restrict__ lit) {
  extern __shared__ float gs_lds[];
  if (done && done[blockIdx.x]) return;   // gq_select_groups already answered this query
  float *sv = gs_lds;
  int *si = reinterpret_cast<int *>(gs_lds + n2);
  __shared__ int s_lit;
  const int q = blockIdx.x, tid = threadIdx.x;
  if (tid == 0) s_lit = 0;
  __syncthreads();
  for (int e = tid; e < n2; e += 256) {
    float v = e < g ? cdist[(size_t)q * g + e] : INFINITY;
    if (v != v) s_lit = 2;            // a NaN distance: only the literal heap knows what the reference does with it
    sv[e] = v != v ? INFINITY : v;    // NaN distances order last
    si[e] = e < g ? e : INT_MAX;
  }
  __syncthreads();
  for (int k = 2; k <= n2; k <<= 1)
    for (int j = k >> 1; j >= 1; j >>= 1) {
      for (int i = tid; i < n2; i += 256) {
        const int l = i ^ j;
        if (l > i) {
          const float a = sv[i], b = sv[l];
          const int ai = si[i], bi = si[l];
          const bool gt = a > b || (a == b && ai > bi);
          if (gt == ((i & k) == 0)) { sv[i] = b; sv[l] = a; si[i] = bi; si[l] = ai; }
        }
      }
      __syncthreads();
    }
  if (tid == 0) {
    int i = 0;
    if (by_vectors) {
      int count = 0;
      while (i < g && count < limit) { const int c = si[i]; count += bounds[c + 1] - bounds[c]; i++; }
    } else {
      i = min(limit, g);
    }
    nn_cnt[q] = i;
  }
  __syncthreads();
  const int cnt = nn_cnt[q];
  for (int e = tid; e < cnt; e += 256) nn[(size_t)q * stride + e] = si[e];
  // (distance, id) order is the reference's heap order only while the entries that decide the answer -- the
  // searched ones and the first one left out -- have pairwise different distances; otherwise gq_literal_groups
  // redoes this query (equal centroid distances are common: the reference's leading empty group repeats a centroid)
  // level 2: the tie sits AT the cut (or a NaN is around): the searched SET hangs on the heap's order -- redone before
  // the scan; level 1: only the ORDER of searched groups does, which matters solely to queries whose RESULT is later
  // redone literally (TopKHeap.merge runs in search order) -- those are redone then
  for (int e = tid; e + 1 < min(cnt + 1, g); e += 256)
    if (sv[e] == sv[e + 1]) atomicMax(&s_lit, e + 1 == cnt ? 2 : 1);
  __syncthreads();
  if (tid == 0) lit[q] = s_lit;
}

// LimitGroups(limit) for limit > 63 (the CLI's default is 5 % of the groups): only the `limit` nearest
// centroids are needed, so instead of sorting all g distances the limit-th smallest key is found
// by a 4-pass radix select on the float bits (distances are >= +0: unsigned order), everything at or
// below it is compacted (limit + ties entries) and only that is sorted by (distance, id).
// ok[q] = 0 if more than `cap` entries tie at the threshold (the caller then sorts everything).
// KR > 0: the query's g <= 256 KR keys are read ONCE and kept in registers through the four counting passes and the
// compaction (the kernel is one workgroup's latency -- all B of them are resident at once: five trips through the
// 40 KB of a query's distances and a 256-step serial walk over the bins per pass were most of its 134 us); KR = 0: any g,
// from memory.
template <int KR>
__global__ __launch_bounds__(256) void gq_select_groups(const float *__restrict__ cdist, int g, int limit, int cap,
                                                        int *__restrict__ nn, int stride,
                                                        int *__restrict__ nn_cnt, int *__restrict__ ok,
                                                        int *__restrict__ lit) {
  extern __shared__ float sel_lds[];
  float *sv = sel_lds;                                     // [cap]
  int *si = reinterpret_cast<int *>(sel_lds + cap);        // [cap]
  __shared__ unsigned hist[256], hsub[256 * 8];   // (eight sub-counters per bin: centroid distances share their high bytes,
  __shared__ unsigned s_prefix, s_remaining;      //  and 10 000 atomics on one or two LDS words serialise)
  __shared__ int s_count, s_lit;
  const int q = blockIdx.x, tid = threadIdx.x;
  const float *dq = cdist + (size_t)q * g;
  auto keyof = [&](int c) { const float v = dq[c]; return v != v ? 0x7F800000u : __float_as_uint(v); };   // NaN orders last
  const int want = min(limit, g);
  if (tid == 0) { s_prefix = 0u; s_remaining = (unsigned)want; s_count = 0; s_lit = 0; }
  constexpr int KRN = KR > 0 ? KR : 1;
  unsigned kreg[KRN];                                      // key of centroid tid + 256 r (0xFFFFFFFF: none)
  bool any_nan = false;
  if (KR > 0) {
#pragma unroll
    for (int r = 0; r < KRN; r++) {
      const int c = tid + 256 * r;
      const float v = c < g ? dq[c] : 0.f;
      any_nan = any_nan || v != v;
      kreg[r] = c < g ? (v != v ? 0x7F800000u : __float_as_uint(v)) : 0xFFFFFFFFu;
    }
  }
  __syncthreads();
  unsigned mask = 0u;
  for (int shift = 24; shift >= 0; shift -= 8) {
    for (int e = tid; e < 256 * 8; e += 256) hsub[e] = 0u;
    __syncthreads();
    const unsigned prefix = s_prefix;
    if (KR > 0) {
#pragma unroll
      for (int r = 0; r < KRN; r++) {
        const unsigned key = kreg[r];
        if (key != 0xFFFFFFFFu && (key & mask) == prefix) atomicAdd(&hsub[((key >> shift) & 255u) * 8 + (tid & 7)], 1u);
      }
    } else {
      for (int c = tid; c < g; c += 256) {
        const unsigned key = keyof(c);
        if ((key & mask) == prefix) atomicAdd(&hsub[((key >> shift) & 255u) * 8 + (tid & 7)], 1u);
      }
    }
    __syncthreads();
    {
      unsigned h = 0;
#pragma unroll
      for (int x = 0; x < 8; x++) h += hsub[tid * 8 + x];
      hist[tid] = h;
    }
    __syncthreads();
    if (tid < 64) {
      // the bin in which the running count reaches `remaining`: four bins per lane, a prefix sum over the lanes
      const unsigned h0 = hist[4 * tid], h1 = hist[4 * tid + 1], h2 = hist[4 * tid + 2], h3 = hist[4 * tid + 3];
      const unsigned mine = h0 + h1 + h2 + h3;
      unsigned incl = mine;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const unsigned up = __shfl_up(incl, o);
        if (tid >= o) incl += up;
      }
      const unsigned rem = s_remaining;
      const unsigned long long reach = __ballot(incl >= rem);
      // (the serial walk stopped at bin 255 at the latest: a total below `remaining` cannot happen -- every pass keeps
      // at least `remaining` keys -- but the last lane takes it then, as the walk did)
      const int first = reach ? __ffsll((long long)reach) - 1 : 63;
      if (tid == first) {
        unsigned cum = incl - mine;
        int b = 4 * tid;
        if (cum + h0 >= rem) { }
        else if (cum + h0 + h1 >= rem) { cum += h0; b += 1; }
        else if (cum + h0 + h1 + h2 >= rem) { cum += h0 + h1; b += 2; }
        else { cum += h0 + h1 + h2; b += 3; }
        s_remaining = rem - cum;
        s_prefix = prefix | ((unsigned)b << shift);
      }
    }
    mask |= 255u << shift;
    __syncthreads();
  }
  const unsigned thr = s_prefix;                           // key of the want-th smallest distance
  if (KR > 0) {
    if (any_nan) s_lit = 2;                                // NaN distance: literal heap (gq_literal_groups)
#pragma unroll
    for (int r = 0; r < KRN; r++) {
      const unsigned key = kreg[r];
      if (key != 0xFFFFFFFFu && key <= thr) {
        const int p = atomicAdd(&s_count, 1);
        if (p < cap) { sv[p] = __uint_as_float(key); si[p] = tid + 256 * r; }
      }
    }
  } else {
    for (int c = tid; c < g; c += 256) {
      const unsigned key = keyof(c);
      if (dq[c] != dq[c]) s_lit = 2;                       // NaN distance: literal heap (gq_literal_groups)
      if (key <= thr) {
        const int p = atomicAdd(&s_count, 1);
        if (p < cap) { sv[p] = __uint_as_float(key); si[p] = c; }
      }
    }
  }
  __syncthreads();
  const int cnt = s_count;
  if (cnt > cap) {                                         // a huge tie at the threshold
    if (tid == 0) ok[q] = 0;
    return;
  }
  int n2 = 64;
  while (n2 < cnt) n2 <<= 1;
  for (int e = cnt + tid; e < n2; e += 256) { sv[e] = INFINITY; si[e] = INT_MAX; }
  __syncthreads();
  for (int k = 2; k <= n2; k <<= 1)
    for (int j = k >> 1; j >= 1; j >>= 1) {
      for (int i = tid; i < n2; i += 256) {
        const int l = i ^ j;
        if (l > i) {
          const float a = sv[i], b = sv[l];
          const int ai = si[i], bi = si[l];
          const bool gt = a > b || (a == b && ai > bi);
          if (gt == ((i & k) == 0)) { sv[i] = b; sv[l] = a; si[i] = bi; si[l] = ai; }
        }
      }
      __syncthreads();
    }
  for (int e = tid; e < want; e += 256) nn[(size_t)q * stride + e] = si[e];
  // equal distances among the searched groups or at the cut (every entry at the threshold was compacted, so a tie
  // there shows as cnt > want): the reference's heap order decides, not (distance, id) -- gq_literal_groups
  for (int e = tid; e + 1 < min(want + 1, cnt); e += 256)
    if (sv[e] == sv[e + 1]) atomicMax(&s_lit, e + 1 == want ? 2 : 1);   // levels: see gq_sorted_groups
  __syncthreads();
  if (tid == 0) { nn_cnt[q] = want; ok[q] = 1; lit[q] = s_lit; }
}

// exactNearestNeighbours(centroids, query, cap).deleteAll() (Index.scala:209-229, :287,:290) LITERALLY, for the
// queries gq_sorted_groups / gq_select_groups flagged: a TopKHeap of capacity `cap` (LimitGroups: the limit;
// LimitVectors: all g groups) in LDS, fed the centroid distances in index order, drained in place like heapsort
// (deleteAll fills its result from the back: slot `size` is free the moment delete() returns).  One wavefront per
// query; every lane runs the same scalar heap code on the same LDS words (stores of equal values, broadcast reads)
// and only the scan for centroids the heap would take is spread over the lanes.
__global__ __launch_bounds__(64) void gq_literal_groups(const float *__restrict__ cdist, int g, int cap,
                                                        const int *__restrict__ bounds, int by_vectors, int limit,
                                                        const int *__restrict__ lit, int level, const int *__restrict__ qlist,
                                                        const int *__restrict__ qcount, int *__restrict__ nn, int stride,
                                                        int *__restrict__ nn_cnt) {
  extern __shared__ float lg_lds[];
  const int lane = threadIdx.x;
  // level 2: every query whose searched set hangs on a tie (grid = B); level 1: the listed queries (redone literally)
  // whose group ORDER does (grid = a few workgroups walking qlist)
  const int nq = qlist ? *qcount : (int)gridDim.x;
  for (int fy = blockIdx.x; fy < nq; fy += gridDim.x) {
  const int q = qlist ? qlist[fy] : fy;
  if (lit[q] != level) continue;
  volatile float *hv = lg_lds;                                  // [cap]
  volatile int *hk = reinterpret_cast<volatile int *>(lg_lds + cap);   // [cap]
  const float *dq = cdist + (size_t)q * g;
  bool has_nan = false;
  for (int c = lane; c < g; c += 64) has_nan = has_nan || dq[c] != dq[c];
  has_nan = __any(has_nan);
  int size = 0;
  auto swp = [&](int a, int b) {
    const float va = hv[a], vb = hv[b];
    const int ka = hk[a], kb = hk[b];
    hv[a] = vb; hk[a] = kb; hv[b] = va; hk[b] = ka;
  };
  auto down = [&](int i) {                                      // percolateDown, TopKHeap.scala:30-42
    for (;;) {
      int top = i;
      const int lc = 2 * i + 1, rc = 2 * i + 2;
      if (lc < size && hv[top] < hv[lc]) top = lc;
      if (rc < size && hv[top] < hv[rc]) top = rc;
      if (top == i) break;
      swp(i, top);
      i = top;
    }
  };
  auto del = [&]() {                                            // delete, TopKHeap.scala:57-67
    size -= 1;
    const int removed = hk[0];
    const float lv = hv[size];
    const int lk = hk[size];
    hv[0] = lv; hk[0] = lk;
    down(0);
    return removed;
  };
  for (int base = 0; base < g; base += 64) {
    const bool have = base + lane < g;
    const float dv = have ? dq[base + lane] : 0.f;
    // with a NaN around the heap is not ordered and every centroid goes through update; otherwise its root only falls
    unsigned long long mk = __ballot(have && (has_nan || size < cap || hv[0] > dv));
    while (mk) {
      const int l = __ffsll((long long)mk) - 1;
      mk &= mk - 1;
      const float x = readlane_f(dv, l);
      if (size == cap && hv[0] > x) del();                      // update, TopKHeap.scala:69-79
      if (size < cap) {
        hv[size] = x; hk[size] = base + l;
        int i = size;
        while (i > 0) {                                         // percolateUp, TopKHeap.scala:21-28
          const int p = (i - 1) / 2;
          if (hv[i] > hv[p]) { swp(i, p); i = p; } else break;
        }
        size += 1;
      }
    }
  }
  const int live = size;
  for (int j = live - 1; j >= 0; j--) {                         // deleteAll(): result(j) = delete(), j descending
    const int kk = del();
    hk[j] = kk;                                                 // slot j == size: outside the heap now
  }
  int cnt = live;
  if (by_vectors) {                                             // searchSpace, Index.scala:289-298
    int i = 0, count = 0;
    while (i < live && count < limit) { const int c = hk[i]; count += bounds[c + 1] - bounds[c]; i++; }
    cnt = i;
  }
  cnt = min(cnt, stride);
  for (int e = lane; e < cnt; e += 64) nn[(size_t)q * stride + e] = hk[e];
  if (lane == 0) nn_cnt[q] = cnt;
  }
}

// ---- one searched group of one query -------------------------------------------------------------
// LITERAL = true: the group's literal TopKHeap, stored in array order (hk/hv/hs).  With a query list
// (qlist/qcount on the device) only the listed queries are processed -- the tie-flagged ones.
// LITERAL = false: the K+1 smallest (distance, row) pairs of the group as an ascending list
// (hv/hk hold K+1 entries per pair, padded with (+inf, INT_MAX)); hs = 1 if a NaN distance was seen.
constexpr int GQ_WAVES = 4;    // (query, group) pairs per workgroup: they share the staged codebook slices
constexpr int GQ_RPT = 8;      // centroid components prefetched per thread (sub-vectors up to 8 wide are fully overlapped)
constexpr int GQ_PD = 4;       // quantizers whose codebooks are in flight
template <int VEC, bool LITERAL, bool BIG = false /* k_nn > 63: the literal heap in LDS */>
__global__ __launch_bounds__(64 * GQ_WAVES) void gq_group_scan(const uint8_t *__restrict__ codes, int ng, int m, int m_pad, int k,
                                                    int d, const float *__restrict__ pq_cents,
                                                    const int *__restrict__ from, const int *__restrict__ sdim,
                                                    const float *__restrict__ gcent, const int *__restrict__ bounds,
                                                    const float *__restrict__ Q, const int *__restrict__ nn,
                                                    int nn_stride, const int *__restrict__ nn_cnt, int stride, int K,
                                                    int *__restrict__ hk, float *__restrict__ hv,
                                                    int *__restrict__ hs, const int *__restrict__ qlist,
                                                    const int *__restrict__ qcount, int slice_floats) {
  using Word = typename CodeWord<VEC>::type;
  // per wave: m_pad * 256 table entries + d residual components; then one codebook slice (k * smax floats)
  extern __shared__ float gq_lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int t = blockIdx.x * GQ_WAVES + wave;           // searched-group slot of this wave
  float *gtab = gq_lds + (size_t)wave * (m_pad * 256 + d);
  float *res = gtab + m_pad * 256;
  float *slice = gq_lds + (size_t)GQ_WAVES * (m_pad * 256 + d);
  const int nq = qlist ? *qcount : (int)gridDim.y;
  for (int fy = blockIdx.y; fy < nq; fy += gridDim.y) {
  const int q = qlist ? qlist[fy] : fy;
  const bool live = t < nn_cnt[q];                       // wave-uniform; dead waves still help staging
  const int c = live ? nn[(size_t)q * nn_stride + t] : 0;
  __syncthreads();
  if (live)
    for (int e = lane; e < d; e += 64) res[e] = Q[(size_t)q * d + e] - gcent[(size_t)c * d + e];   // MathUtils.subtract
  // Index.prepareQuery on the residual: T[j][c'] = sum_e (r[from_j+e] - cent_j[c'][e])^2, e ascending, unfused.
  // The codebook of quantizer j (k * s_j contiguous floats) is staged once for the four pairs.
  // The codebooks of the next GQ_PD quantizers are in flight in registers (GQ_RPT floats per thread
  // each) while this one is used: at two workgroups per CU a single global round trip costs more
  // than a quantizer's 256 x s table entries (LDS holds one slice at a time).
  static_assert(64 * GQ_WAVES == 256, "thread = centroid staging assumes 256 threads");
  float pre[GQ_PD][GQ_RPT];
  // thread = centroid (k <= 256 = GQ_THREADS): component x of its centroid, no index arithmetic
  auto fetch = [&](int j, float (&dst)[GQ_RPT]) {
    const int fr = j < m ? from[j] : 0, sj = j < m ? sdim[j] : 0;
    const float *cent = pq_cents + (size_t)k * fr + (size_t)tid * sj;
#pragma unroll
    for (int u = 0; u < GQ_RPT; u++) dst[u] = (tid < k && u < sj) ? cent[u] : 0.f;
  };
#pragma unroll
  for (int p = 0; p < GQ_PD; p++) fetch(p, pre[p]);
  for (int j0 = 0; j0 < m_pad; j0 += GQ_PD) {            // m_pad is a multiple of 4
#pragma unroll
    for (int p = 0; p < GQ_PD; p++) {
      const int j = j0 + p;
      const int fr = j < m ? from[j] : 0, sj = j < m ? sdim[j] : 0;
      float *sl = slice;
      __syncthreads();                                   // slice j-1 is no longer read
      // stored transposed, [x][centroid]: the lanes of a wave then read consecutive addresses
#pragma unroll
      for (int u = 0; u < GQ_RPT; u++)
        if (u < sj) sl[u * 256 + tid] = pre[p][u];
      for (int x = GQ_RPT; x < sj; x++)                  // long sub-vectors: the rest straight from memory
        sl[x * 256 + tid] = tid < k ? pq_cents[(size_t)k * fr + (size_t)tid * sj + x] : 0.f;
      if (j + GQ_PD < m_pad) fetch(j + GQ_PD, pre[p]);
      __syncthreads();                                   // slice j complete
      if (live) {
        float acc[4] = {0.f, 0.f, 0.f, 0.f};             // centroids lane, lane + 64, lane + 128, lane + 192
        for (int x = 0; x < sj; x++) {
          const float rx = res[fr + x];
#pragma unroll
          for (int i = 0; i < 4; i++) {
            const float dd = rx - sl[x * 256 + lane + 64 * i];
            acc[i] += dd * dd;
          }
        }
#pragma unroll
        for (int i = 0; i < 4; i++) gtab[j * 256 + lane + 64 * i] = lane + 64 * i < k ? acc[i] : 0.f;
      }
    }
  }
  __syncthreads();
  if (!live) continue;
  const int row_from = bounds[c], row_until = bounds[c + 1];
  const int keff = K + 1;
  // (BIG: the heaps sit behind the tables and the codebook slice)
  float *bigv = gq_lds + (size_t)GQ_WAVES * (m_pad * 256 + d) + slice_floats + (size_t)wave * (BIG ? K : 0);
  int *bigk = reinterpret_cast<int *>(gq_lds + (size_t)GQ_WAVES * (m_pad * 256 + d) + slice_floats + (size_t)GQ_WAVES * (BIG ? K : 0)) +
              (size_t)wave * (BIG ? K : 0);
  typename std::conditional<BIG, LdsHeap, RegHeap>::type h = [&] {
    if constexpr (BIG) return LdsHeap(bigv, bigk, K, lane);
    else return RegHeap(K, lane);
  }();
  WaveList wl;
  wl.init();
  int cnt = 0, saw_nan = 0;
  const Word *cw = reinterpret_cast<const Word *>(codes);
  // the first code words of the next GQ_PD row blocks stay in flight
  const int rb_first = row_from / 64, rb_end = (row_until + 63) / 64;
  Word wq[GQ_PD];
#pragma unroll
  for (int p = 0; p < GQ_PD; p++) wq[p] = rb_first + p < rb_end ? cw[((size_t)(rb_first + p) * ng) * 64 + lane] : Word{};
  for (int rb0 = rb_first; rb0 < rb_end; rb0 += GQ_PD) {
#pragma unroll
  for (int p = 0; p < GQ_PD; p++) {
    const int rb = rb0 + p;
    if (rb >= rb_end) break;
    float acc = 0.f;                 // PQIndex.distances: j ascending, unfused fp32
    const Word w0 = wq[p];
    if (rb + GQ_PD < rb_end) wq[p] = cw[((size_t)(rb + GQ_PD) * ng) * 64 + lane];
    for (int gi = 0; gi < ng; gi++) {
      const Word w = gi == 0 ? w0 : cw[((size_t)rb * ng + gi) * 64 + lane];
      const float *tj = gtab + gi * VEC * 256;
#pragma unroll
      for (int b = 0; b < VEC; b++) acc += tj[b * 256 + code_byte<VEC>(w, b)];
    }
    const int row = rb * 64 + lane;
    const bool valid = row >= row_from && row < row_until;
    if (LITERAL) {
      // rows in ascending order through heap.update; the ballot only skips rows the heap would
      // reject anyway (full and root <= value -- NaN compares false and is rejected like there)
      unsigned long long mk = __ballot(valid && (h.size < K || h.val(0) > acc));
      while (mk) {
        const int l = __ffsll((long long)mk) - 1;
        mk &= mk - 1;
        const float x = readlane_f(acc, l);
        if (h.would_insert(x)) h.update(rb * 64 + l, x);
      }
    } else {
      if (__ballot(valid && acc != acc) != 0ull) saw_nan = 1;   // the heap keeps NaNs while it is not full
      unsigned long long mk = __ballot(valid && (cnt < keff || wl.accepts(acc, row)));
      while (mk) {
        const int l = __ffsll((long long)mk) - 1;
        mk &= mk - 1;
        const float x = readlane_f(acc, l);
        const int r = rb * 64 + l;
        if (cnt < keff || wl.accepts(x, r)) {
          wl.insert(x, r, keff, lane);
          if (cnt < keff) cnt++;
        }
      }
    }
  }
  }
  if (LITERAL) {
    const size_t o = ((size_t)q * stride + t) * K;
    if constexpr (BIG) {
      for (int i = lane; i < h.size; i += 64) { hk[o + i] = h.key(i); hv[o + i] = h.val(i); }
    } else {
      if (lane < h.size) { hk[o + lane] = h.k; hv[o + lane] = h.v; }
    }
    if (lane == 0) hs[(size_t)q * stride + t] = h.size;
  } else {
    const size_t o = ((size_t)q * stride + t) * keff;
    if (lane < keff) { hk[o + lane] = wl.i; hv[o + lane] = wl.v; }
    if (lane == 0) hs[(size_t)q * stride + t] = saw_nan;
  }
  }
}

// ---- the same (LITERAL = false) result, quantizer-major ------------------------------------------
// gq_group_scan keeps a whole 16 KiB table per (query, group) pair in LDS: 8 waves per CU, one
// workgroup barrier per quantizer for 4 pairs, and most of the time is latency.  Here a workgroup is 8
// pairs (one wave each) and walks the quantizers two at a time: the two codebook slices are staged once
// for all pairs of the workgroup (double-buffered, one barrier per step), every wave builds only the 2 x 256 table
// entries of its pair for these two quantizers (2 KiB) and adds them straight onto per-row partial sums
// held in registers -- up to 32 row blocks (2048 rows) of the group; longer groups take further passes.
// The sums still grow in the reference's order (j ascending, unfused fp32).  52 KiB of LDS per workgroup
// at d = 128: three workgroups = 24 waves per CU.  Codes come from a second layout with the bytes of
// quantizers 2h and 2h + 1 of a row side by side (one 2-byte load per row and step).
constexpr int QM_WAVES = 8;       // 76 VGPRs allow 6 waves per SIMD: three workgroups of 8 waves (52 KiB each at d = 128) per CU
constexpr int QM_PARTS = QM_WAVES * 64 / 256;   // staging threads per centroid
constexpr int QM_RB = 32;      // row blocks whose partial sums a wave holds
constexpr int QM_LB = 8;       // code words in flight

// The pairs of a workgroup walk the quantizers in lockstep (one barrier per step), so a workgroup is as slow
// as its largest group: give it groups of similar size.  Per query, the searched groups sorted by their
// number of row blocks, largest first (any order of the groups gives the same merged list).
__global__ __launch_bounds__(256) void gq_sort_by_size(const int *__restrict__ nn, int nn_stride,
                                                       const int *__restrict__ nn_cnt, const int *__restrict__ bounds,
                                                       int npad /* power of two >= every count, <= 2048 */,
                                                       int *__restrict__ out) {
  extern __shared__ unsigned long long gs_keys[];
  const int q = blockIdx.x, tid = threadIdx.x;
  const int cnt = nn_cnt[q];
  for (int e = tid; e < npad; e += 256) {
    unsigned long long key = ~0ull;                      // padding sorts last
    if (e < cnt) {
      const int c = nn[(size_t)q * nn_stride + e];
      const unsigned size = (unsigned)(bounds[c + 1] - bounds[c]);
      key = ((unsigned long long)(0xFFFFFFFFu - size) << 32) | (unsigned)c;
    }
    gs_keys[e] = key;
  }
  __syncthreads();
  for (int k2 = 2; k2 <= npad; k2 <<= 1)
    for (int j = k2 >> 1; j >= 1; j >>= 1) {
      for (int e = tid; e < npad; e += 256) {
        const int p = e ^ j;
        if (p > e) {
          const unsigned long long a = gs_keys[e], b = gs_keys[p];
          const bool up = (e & k2) == 0;
          if ((a > b) == up) { gs_keys[e] = b; gs_keys[p] = a; }
        }
      }
      __syncthreads();
    }
  for (int e = tid; e < cnt; e += 256) out[(size_t)q * nn_stride + e] = (int)(gs_keys[e] & 0xFFFFFFFFull);
}

__global__ void gq_pair_codes(const uint8_t *__restrict__ codes /*[n/64][ng][64][vec]*/, int ng, int vec, int m, int mh,
                              uint16_t *__restrict__ out /*[n/64][mh][64]*/, long long total) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const int lane = (int)(t & 63);
  const long long bh = t >> 6;
  const int h = (int)(bh % mh);
  const long long rb = bh / mh;
  auto byte_of = [&](int j) -> uint32_t {
    if (j >= m) return 0u;
    return codes[(((size_t)rb * ng + j / vec) * 64 + lane) * vec + j % vec];
  };
  out[t] = (uint16_t)(byte_of(2 * h) | (byte_of(2 * h + 1) << 8));
}

template <int XS>   // centroid components per staging thread and quantizer: sub-vectors up to QM_PARTS * XS wide
__global__ __launch_bounds__(64 * QM_WAVES) void gq_scan_qm(
    const uint16_t *__restrict__ codes2, int mh, int m, int k, int d, const float *__restrict__ pq_cents,
    const int *__restrict__ from, const int *__restrict__ sdim, const float *__restrict__ gcent,
    const int *__restrict__ bounds, const float *__restrict__ Q, const int *__restrict__ nn, int nn_stride,
    const int *__restrict__ nn_cnt, int stride, int K, int *__restrict__ hk, float *__restrict__ hv,
    int *__restrict__ hs, int smax) {
  extern __shared__ float qm_lds[];
  __shared__ int s_npass;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cbq = smax * 256;                       // floats of one staged codebook slice, [x][256]
  float *cb = qm_lds;                               // [2 buffers][2 quantizers][cbq]
  float *T2 = qm_lds + 4 * (size_t)cbq + (size_t)wave * (512 + d);
  float *res = T2 + 512;
  const int q = blockIdx.y;
  const int t = blockIdx.x * QM_WAVES + wave;       // searched-group slot of this wave
  const bool live = t < nn_cnt[q];                  // wave-uniform; dead waves still help staging
  const int c = live ? nn[(size_t)q * nn_stride + t] : 0;
  const int row_from = live ? bounds[c] : 0, row_until = live ? bounds[c + 1] : 0;
  const int rb_first = row_from / 64;
  const int nrb_total = live ? (row_until + 63) / 64 - rb_first : 0;
  if (tid == 0) s_npass = 0;
  __syncthreads();
  if (lane == 0 && nrb_total > 0) atomicMax(&s_npass, (nrb_total + QM_RB - 1) / QM_RB);
  if (live)
    for (int e = lane; e < d; e += 64) res[e] = Q[(size_t)q * d + e] - gcent[(size_t)c * d + e];   // MathUtils.subtract
  __syncthreads();
  const int npass = s_npass;
  const int nsteps = (m + 1) / 2;
  const int cc = tid & 255, part = tid >> 8;        // staging: centroid cc, components part, part + QM_PARTS, ...
  auto fetch = [&](int step, float (&dst)[2][XS]) {
#pragma unroll
    for (int jj = 0; jj < 2; jj++) {
      const int j = 2 * step + jj;
      const int fr = j < m ? from[j] : 0, sj = j < m ? sdim[j] : 0;
      const float *cent = pq_cents + (size_t)k * fr + (size_t)cc * sj;
#pragma unroll
      for (int u = 0; u < XS; u++) {
        const int x = part + QM_PARTS * u;
        dst[jj][u] = (cc < k && x < sj) ? cent[x] : 0.f;
      }
    }
  };
  auto store = [&](int buf, const float (&src)[2][XS]) {
#pragma unroll
    for (int jj = 0; jj < 2; jj++)
#pragma unroll
      for (int u = 0; u < XS; u++) {
        const int x = part + QM_PARTS * u;
        if (x < smax) cb[(size_t)(buf * 2 + jj) * cbq + x * 256 + cc] = src[jj][u];
      }
  };
  const int keff = K + 1;
  WaveList wl;
  wl.init();
  int cnt = 0, saw_nan = 0;
  for (int pass = 0; pass < npass; pass++) {
    const int rb0 = rb_first + pass * QM_RB;
    const int nrb = min(max(nrb_total - pass * QM_RB, 0), QM_RB);
    float acc[QM_RB];
#pragma unroll
    for (int i = 0; i < QM_RB; i++) acc[i] = 0.f;    // PQIndex.distances: j ascending, unfused fp32
    float pre[2][XS];
    fetch(0, pre);
    store(0, pre);             // (every wave left the previous pass through its last barrier)
    __syncthreads();
    for (int step = 0; step < nsteps; step++) {
      const int buf = step & 1;
      if (step + 1 < nsteps) fetch(step + 1, pre);
      if (nrb > 0) {
        // Index.prepareQuery on the residual, quantizers 2 step and 2 step + 1:
        // T[c'] = sum_e (r[from_j + e] - cent_j[c'][e])^2, e ascending, unfused
#pragma unroll
        for (int jj = 0; jj < 2; jj++) {
          const int j = 2 * step + jj;
          if (j < m) {
            const int fr = from[j], sj = sdim[j];
            const float *sl = cb + (size_t)(buf * 2 + jj) * cbq;
            // centroids lane, lane + 64 | lane + 128, lane + 192: two per packed-fp32 instruction
            // (v_pk_add_f32 / v_pk_mul_f32 round each half like the scalar instruction: still unfused)
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            f32x2 a01 = {0.f, 0.f}, a23 = {0.f, 0.f};
            for (int x = 0; x < sj; x++) {
              const float rx = res[fr + x];
              const f32x2 r2 = {rx, rx};
              const f32x2 c01 = {sl[x * 256 + lane], sl[x * 256 + lane + 64]};
              const f32x2 c23 = {sl[x * 256 + lane + 128], sl[x * 256 + lane + 192]};
              const f32x2 d01 = r2 - c01, d23 = r2 - c23;
              a01 += d01 * d01;
              a23 += d23 * d23;
            }
            T2[jj * 256 + lane] = a01.x; T2[jj * 256 + lane + 64] = a01.y;
            T2[jj * 256 + lane + 128] = a23.x; T2[jj * 256 + lane + 192] = a23.y;
          }
        }
        const bool two = 2 * step + 1 < m;
#pragma unroll
        for (int b0 = 0; b0 < QM_RB; b0 += QM_LB) {
          if (b0 < nrb) {
            uint32_t w[QM_LB];
#pragma unroll
            for (int i = 0; i < QM_LB; i++)
              w[i] = b0 + i < nrb ? (uint32_t)codes2[((size_t)(rb0 + b0 + i) * mh + step) * 64 + lane] : 0u;
#pragma unroll
            for (int i = 0; i < QM_LB; i++) {
              if (b0 + i < nrb) {
                acc[b0 + i] += T2[w[i] & 255u];
                if (two) acc[b0 + i] += T2[256 + (w[i] >> 8)];
              }
            }
          }
        }
      }
      if (step + 1 < nsteps) store((step + 1) & 1, pre);
      __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < QM_RB; i++) {
      if (i < nrb) {
        const int rb = rb0 + i;
        const int row = rb * 64 + lane;
        const bool valid = row >= row_from && row < row_until;
        const float a = acc[i];
        if (__ballot(valid && a != a) != 0ull) saw_nan = 1;   // the heap keeps NaNs while it is not full
        unsigned long long mk = __ballot(valid && (cnt < keff || wl.accepts(a, row)));
        while (mk) {
          const int l = __ffsll((long long)mk) - 1;
          mk &= mk - 1;
          const float x = readlane_f(a, l);
          const int r = rb * 64 + l;
          if (cnt < keff || wl.accepts(x, r)) {
            wl.insert(x, r, keff, lane);
            if (cnt < keff) cnt++;
          }
        }
      }
    }
  }
  if (live) {
    const size_t o = ((size_t)q * stride + t) * keff;
    if (lane < keff) { hk[o + lane] = wl.i; hv[o + lane] = wl.v; }
    if (lane == 0) hs[(size_t)q * stride + t] = saw_nan;
  }
}

// ---- one searched group of one query over 16-bit codes (k > 256: Coder.BytePlus, wide.hip) ----------------
// The residual's table is m * k floats (64 KiB at k = 1024, 4 MiB at k = 65 536): it is built in a slot of
// global scratch (one slot per workgroup, which walks the (query, group) pairs), and the group's rows go through
// the LITERAL TopKHeap in row order -- the same heaps, stored in array order, as gq_group_scan<.., true>, so
// that gq_merge folds them exactly as GroupedIndex.query does (Index.scala:265-282).  Generality over speed.
__global__ __launch_bounds__(64) void gq_group_scan_wide(const uint16_t *__restrict__ wcodes, int m, int k, int d,
                                                         const float *__restrict__ pq_cents,
                                                         const int *__restrict__ from, const int *__restrict__ sdim,
                                                         const float *__restrict__ gcent, const int *__restrict__ bounds,
                                                         const float *__restrict__ Q, const int *__restrict__ nn,
                                                         int nn_stride, const int *__restrict__ nn_cnt, int stride, int B,
                                                         int K, float *__restrict__ scratch /*[gridDim.x][m][k]*/,
                                                         int *__restrict__ hk, float *__restrict__ hv, int *__restrict__ hs) {
  extern __shared__ float gw_res[];   // d
  const int lane = threadIdx.x;
  float *T = scratch + (size_t)blockIdx.x * m * k;
  const long long pairs = (long long)B * stride;
  for (long long pr = blockIdx.x; pr < pairs; pr += gridDim.x) {
    const int q = (int)(pr / stride), t = (int)(pr - (long long)q * stride);
    if (t >= nn_cnt[q]) continue;
    const int c = nn[(size_t)q * nn_stride + t];
    for (int e = lane; e < d; e += 64) gw_res[e] = Q[(size_t)q * d + e] - gcent[(size_t)c * d + e];   // MathUtils.subtract
    // Index.prepareQuery on the residual (Index.scala:352-383): e ascending, unfused
    for (int j = 0; j < m; j++) {
      const int fr = from[j], sj = sdim[j];
      for (int cc = lane; cc < k; cc += 64) {
        const float *cent = pq_cents + (size_t)k * fr + (size_t)cc * sj;
        float acc = 0.f;
        for (int x = 0; x < sj; x++) {
          const float dd = gw_res[fr + x] - cent[x];
          acc += dd * dd;
        }
        T[(size_t)j * k + cc] = acc;
      }
    }
    __threadfence();   // the table is read back by other lanes through the vector L1
    const int row_from = bounds[c], row_until = bounds[c + 1];
    RegHeap h(K, lane);
    for (int rb = row_from / 64; rb < (row_until + 63) / 64; rb++) {
      const uint16_t *p = wcodes + (size_t)rb * m * 64 + lane;
      float acc = 0.f;                 // PQIndex.distances: j ascending, unfused fp32
      for (int j = 0; j < m; j++) acc += T[(size_t)j * k + p[(size_t)j * 64]];
      const int row = rb * 64 + lane;
      const bool valid = row >= row_from && row < row_until;
      unsigned long long mk = __ballot(valid && (h.size < K || h.val(0) > acc));
      while (mk) {
        const int l = __ffsll((long long)mk) - 1;
        mk &= mk - 1;
        const float x = readlane_f(acc, l);
        if (h.would_insert(x)) h.update(rb * 64 + l, x);
      }
    }
    const size_t o = ((size_t)q * stride + t) * K;
    if (lane < h.size) { hk[o + lane] = h.k; hv[o + lane] = h.v; }
    if (lane == 0) hs[(size_t)q * stride + t] = h.size;
    __threadfence();   // the next pair overwrites the table slot
  }
}

// ---- approximate pre-selection + exact re-ranking -------------------------------------------------------
// The reference scores a row of a searched group through the residual's own table: one m x 256 table per (query,
// group) pair -- 512 K tables of 98 Kflop per 1024-query batch at 10 M rows / LimitGroups(500), more work than
// the scan they serve (8.4 ms per batch in round 1, against 3 ms for the flat index scanning 20 times the rows).
// But the distance it computes is  |q - (g + r^_i)|^2  with  r^_i = decode(codes_i),  and
//     |q - x^_i|^2 = |q|^2 - 2 q.g - 2 sum_j q_j . c_j[code_ij] + |x^_i|^2 ,      x^_i = g + r^_i ,
// needs ONE table per QUERY (P[j][c] = -2 q_j . c_j[c], 16 KiB, shared by all its groups), one dot product per
// (query, group) and one precomputed number per row.  That value D~ is not the reference's arithmetic, so it only
// SELECTS: every query keeps the 64 rows with the smallest D~ of its searched groups (gq_approx_scan), those are
// re-scored with the reference's arithmetic (gq_rerank: MathUtils.subtract, Index.prepareQuery's and
// PQIndex.distances' summation order, bit for bit) and ordered by (distance, row); the answer is certified when
// the (K+1)-th exact distance lies below the 64th D~ minus an error margin that bounds |D~ - exact| -- then no
// row outside the 64 can reach the K+1 best.  Uncertified queries, queries with equal distances among their K+1
// best and queries that saw a NaN go to the literal kernels as before, so results stay the reference's.
__global__ void gq_row_norms(const uint8_t *__restrict__ codes, int ng, int vec, int m, int k, int d,
                             const float *__restrict__ pq_cents, const int *__restrict__ from, const int *__restrict__ sdim,
                             const float *__restrict__ gcent, const int *__restrict__ bounds, int g, int n,
                             float *__restrict__ xnorm, unsigned *__restrict__ xnmax_bits) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int lo = 0, hi = g;                       // group of row i: largest c with bounds[c] <= i (empty groups skipped)
  while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (bounds[mid] <= i) lo = mid; else hi = mid; }
  const float *gc = gcent + (size_t)lo * d;
  float acc = 0.f;
  for (int j = 0; j < m; j++) {
    const int code = codes[(((size_t)(i >> 6) * ng + j / vec) * 64 + (i & 63)) * vec + j % vec];
    const int fr = from[j], sj = sdim[j];
    const float *c = pq_cents + (size_t)k * fr + (size_t)code * sj;
    for (int t = 0; t < sj; t++) { const float v = gc[fr + t] + c[t]; acc += v * v; }
  }
  xnorm[i] = acc;
  atomicMax(xnmax_bits, __float_as_uint(acc == acc && acc < INFINITY ? acc : INFINITY));
}

// P[q][j][c] = -2 * (q_j . c_j[c]); entries beyond k (and padding quantizers) are 0
__global__ __launch_bounds__(256) void gq_ptables(const float *__restrict__ pq_cents, const int *__restrict__ from,
                                                  const int *__restrict__ sdim, int d, int m, int m_pad, int k,
                                                  const float *__restrict__ Q, float *__restrict__ P) {
  const int c = threadIdx.x, j = blockIdx.x, q = blockIdx.y;
  float acc = 0.f;
  if (j < m && c < k) {
    const int fr = from[j], sj = sdim[j];
    const float *cc = pq_cents + (size_t)k * fr + (size_t)c * sj;
    for (int t = 0; t < sj; t++) acc += Q[(size_t)q * d + fr + t] * cc[t];
    acc *= -2.0f;
  }
  P[((size_t)q * m_pad + j) * 256 + c] = acc;
}

constexpr int GA_WAVES = 16;    // waves of one query's workgroup: each takes every 16th searched group
constexpr int GA_C = 64;        // candidates kept per query
template <int VEC>
__global__ __launch_bounds__(64 * GA_WAVES) void gq_approx_scan(const uint8_t *__restrict__ codes, int ng, int m_pad, int d,
                                                                const float *__restrict__ P, const float *__restrict__ xnorm,
                                                                const float *__restrict__ gcent,
                                                                const int *__restrict__ bounds, const float *__restrict__ Q,
                                                                const int *__restrict__ nn, int nn_stride,
                                                                const int *__restrict__ nn_cnt, int gmax /* searched groups taken at most */,
                                                                int split /* waves that share a group's row blocks (1, 2, 4 ..) */,
                                                                float *__restrict__ lv,
                                                                int *__restrict__ li, int *__restrict__ nanflag) {
  using Word = typename CodeWord<VEC>::type;
  extern __shared__ float ga_lds[];           // m_pad * 256 table entries, then d query coordinates
  float *tab = ga_lds, *qv = ga_lds + m_pad * 256;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = blockIdx.x;
  for (int e = tid; e < m_pad * 256; e += 64 * GA_WAVES) tab[e] = P[(size_t)q * m_pad * 256 + e];
  for (int e = tid; e < d; e += 64 * GA_WAVES) qv[e] = Q[(size_t)q * d + e];
  __syncthreads();
  float qq = 0.f;
  for (int e = lane; e < d; e += 64) qq += qv[e] * qv[e];
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) qq += __shfl_xor(qq, o);
  WaveList wl;
  wl.init();
  int cnt = 0, saw_nan = 0;
  const int ngroups = min(nn_cnt[q], gmax);
  const Word *cw = reinterpret_cast<const Word *>(codes);
  for (int t = wave / split; t < ngroups; t += GA_WAVES / split) {
    const int c = nn[(size_t)q * nn_stride + t];
    const int row_from = bounds[c], row_until = bounds[c + 1];
    float qg = 0.f;
    for (int e = lane; e < d; e += 64) qg += qv[e] * gcent[(size_t)c * d + e];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) qg += __shfl_xor(qg, o);
    const float base = qq - 2.0f * qg;
    const int rb_first = (row_from >> 6) + wave % split, rb_end = (row_until + 63) >> 6;
    Word wnext{};
    if (rb_first < rb_end) wnext = cw[((size_t)rb_first * ng) * 64 + lane];
    for (int rb = rb_first; rb < rb_end; rb += split) {
      const Word w0 = wnext;
      if (rb + split < rb_end) wnext = cw[((size_t)(rb + split) * ng) * 64 + lane];
      const int row = rb * 64 + lane;
      const bool valid = row >= row_from && row < row_until;
      float acc = base + (valid ? xnorm[row] : 0.f);
      for (int gi = 0; gi < ng; gi++) {
        const Word w = gi == 0 ? w0 : cw[((size_t)rb * ng + gi) * 64 + lane];
        const float *tj = tab + gi * VEC * 256;
#pragma unroll
        for (int b = 0; b < VEC; b++) acc += tj[b * 256 + code_byte<VEC>(w, b)];
      }
      if (__ballot(valid && acc != acc) != 0ull) saw_nan = 1;
      unsigned long long mk = __ballot(valid && (cnt < GA_C || wl.accepts(acc, row)));
      while (mk) {
        const int l = __ffsll((long long)mk) - 1;
        mk &= mk - 1;
        const float x = readlane_f(acc, l);
        const int r = rb * 64 + l;
        if (cnt < GA_C || wl.accepts(x, r)) {
          wl.insert(x, r, GA_C, lane);
          if (cnt < GA_C) cnt++;
        }
      }
    }
  }
  const size_t o = ((size_t)q * GA_WAVES + wave) * GA_C;
  lv[o + lane] = wl.v;
  li[o + lane] = wl.i;
  if (lane == 0) nanflag[q * GA_WAVES + wave] = saw_nan;
}

// 64-lane bitonic sort of (value, id) pairs, ascending by (value, id); padding = (+inf, INT_MAX)
__device__ inline void sort64_pairs(float &v, int &id, int lane) {
#pragma unroll
  for (int k = 2; k <= 64; k <<= 1)
#pragma unroll
    for (int j = k >> 1; j >= 1; j >>= 1) {
      const float ov = __shfl_xor(v, j);
      const int oi = __shfl_xor(id, j);
      const bool up = (lane & k) == 0, lower = (lane & j) == 0;
      const bool other_less = ov < v || (ov == v && oi < id);
      const bool take = (lower == up) ? other_less : !other_less && !(ov == v && oi == id);
      if (take) { v = ov; id = oi; }
    }
}

// exact re-scoring of one query's candidates (lane = candidate) + certificate; see the block comment above
__global__ __launch_bounds__(64) void gq_rerank(const uint8_t *__restrict__ codes, int ng, int vec, int m, int k, int d,
                                                const float *__restrict__ pq_cents, const int *__restrict__ from,
                                                const int *__restrict__ sdim, const float *__restrict__ gcent,
                                                const int *__restrict__ bounds, int g, const float *__restrict__ Q,
                                                const float *__restrict__ cv, const int *__restrict__ ci,
                                                const int *__restrict__ nanflag, float xnmax, int K,
                                                int *__restrict__ out_idx, float *__restrict__ out_dist,
                                                int *__restrict__ out_count, int *__restrict__ qlist,
                                                int *__restrict__ qcount) {
  const int q = blockIdx.x, lane = threadIdx.x;
  const float approx = cv[(size_t)q * GA_C + lane];
  const int row = ci[(size_t)q * GA_C + lane];
  const bool have = row != INT_MAX;
  const int ncand = __popcll(__ballot(have));
  const float a_last = readlane_f(approx, GA_C - 1);          // the largest kept D~ (every other row's is >= it)
  float qq = 0.f;
  for (int e = lane; e < d; e += 64) { const float x = Q[(size_t)q * d + e]; qq += x * x; }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) qq += __shfl_xor(qq, o);
  float D = INFINITY;
  if (have) {
    int lo = 0, hi = g;
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (bounds[mid] <= row) lo = mid; else hi = mid; }
    const float *gc = gcent + (size_t)lo * d;
    const float *qp = Q + (size_t)q * d;
    D = 0.f;                                                  // PQIndex.distances: 0 + T[0] + T[1] + ... (unfused fp32)
    for (int j = 0; j < m; j++) {
      const int code = codes[(((size_t)(row >> 6) * ng + j / vec) * 64 + (row & 63)) * vec + j % vec];
      const int fr = from[j], sj = sdim[j];
      const float *c = pq_cents + (size_t)k * fr + (size_t)code * sj;
      float tj = 0.f;                                         // Index.prepareQuery on the residual (MathUtils.subtract)
      for (int t = 0; t < sj; t++) { const float dd = (qp[fr + t] - gc[fr + t]) - c[t]; tj += dd * dd; }
      D += tj;
    }
  }
  const bool nan_d = __ballot(have && D != D) != 0ull;
  float sv = have && D == D ? D : INFINITY;
  int si = have && D == D ? row : INT_MAX;
  sort64_pairs(sv, si, lane);
  const int nfin = __popcll(__ballot(si != INT_MAX));
  const int live = min(K, nfin);
  if (lane < K) {
    out_idx[(size_t)q * K + lane] = lane < live ? si : -1;
    out_dist[(size_t)q * K + lane] = lane < live ? sv : INFINITY;
  }
  const float nv = __shfl_down(sv, 1);
  const int ni = __shfl_down(si, 1);
  const bool tie = lane < K && si != INT_MAX && ni != INT_MAX && sv == nv;      // equal neighbours among the K+1 best
  // |D~ - exact| <= margin: both are sums of ~d + m terms of magnitude <= (|q| + |x^|)^2
  const float xm = __fsqrt_rn(qq) + __fsqrt_rn(xnmax);
  const float margin = 4.0f * (float)(d + 2 * m + 16) * 5.9604645e-8f * xm * xm;
  bool certified = true;
  if (ncand == GA_C) {                                        // rows were left out: the K+1 best must clear their bound
    const float ek = readlane_f(sv, min(K, GA_C - 1));        // (K+1)-th exact distance (K <= 63)
    certified = nfin > K && ek < a_last - margin && margin == margin && a_last == a_last;
  }
  int nanany = lane < GA_WAVES ? nanflag[q * GA_WAVES + lane] : 0;
  const bool redo = !certified || nan_d || __ballot(tie) != 0ull || __ballot(nanany != 0) != 0ull;
  if (lane == 0) {
    if (out_count) out_count[q] = live;
    if (redo) qlist[atomicAdd(qcount, 1)] = q;
  }
}

// ---- TopKHeap.merge of the group heaps in search order, Result.fromHeap ---------------------------
template <bool BIG>
__global__ __launch_bounds__(64) void gq_merge(const int *__restrict__ hk, const float *__restrict__ hv,
                                               const int *__restrict__ hs, const int *__restrict__ nn_cnt, int stride,
                                               int K, int *__restrict__ out_idx, float *__restrict__ out_dist,
                                               int *__restrict__ out_count, const int *__restrict__ qlist,
                                               const int *__restrict__ qcount) {
  extern __shared__ float gm_lds[];            // BIG: K values, then K keys
  const int lane = threadIdx.x;
  const int nq = qlist ? *qcount : (int)gridDim.x;
  for (int fx = blockIdx.x; fx < nq; fx += gridDim.x) {
  const int q = qlist ? qlist[fx] : fx;
  typename std::conditional<BIG, LdsHeap, RegHeap>::type h = [&] {
    if constexpr (BIG) return LdsHeap(gm_lds, reinterpret_cast<int *>(gm_lds + K), K, lane);
    else return RegHeap(K, lane);
  }();
  const int cnt = nn_cnt[q];
  for (int t = 0; t < cnt; t++) {
    const size_t o = ((size_t)q * stride + t) * K;
    const int sz = hs[(size_t)q * stride + t];
    if constexpr (BIG) {
      for (int i = 0; i < sz; i++) h.update(hk[o + i], hv[o + i]);                    // array order
    } else {
      const int kk = lane < sz ? hk[o + lane] : 0;
      const float vv = lane < sz ? hv[o + lane] : 0.f;
      for (int i = 0; i < sz; i++) h.update(readlane_i(kk, i), readlane_f(vv, i));   // array order
    }
  }
  const int live = h.size;
  for (int i = live - 1; i >= 0; i--) {                     // Result.fromHeap: max first, fill from the back
    const float tv = h.val(0);
    const int tk = h.del();
    if (lane == 0) { out_idx[(size_t)q * K + i] = tk; out_dist[(size_t)q * K + i] = tv; }
  }
  for (int i = live + lane; i < K; i += 64) { out_idx[(size_t)q * K + i] = -1; out_dist[(size_t)q * K + i] = INFINITY; }
  if (lane == 0 && out_count) out_count[q] = live;
  }
}

// Tie-free fast path: merge the groups' ascending (K+1)-lists under the (distance, row) order.
// Wherever the K+1 smallest distances of the searched rows are pairwise different this IS the
// reference's answer; a query with equal neighbours among them (or a NaN distance anywhere) is
// appended to qlist and redone by the literal kernels.
__global__ __launch_bounds__(64) void gq_merge_fast(const int *__restrict__ li, const float *__restrict__ lv,
                                                    const int *__restrict__ nanflag,
                                                    const int *__restrict__ nn_cnt, int stride, int K,
                                                    int *__restrict__ out_idx, float *__restrict__ out_dist,
                                                    int *__restrict__ out_count, int *__restrict__ qlist,
                                                    int *__restrict__ qcount) {
  const int q = blockIdx.x, lane = threadIdx.x;
  const int keff = K + 1;
  WaveList wl;
  wl.init();
  const int cnt = nn_cnt[q];
  const int total = cnt * keff;
  int nanany = 0;
  for (int t = lane; t < cnt; t += 64) nanany |= nanflag[(size_t)q * stride + t];
  nanany = __ballot(nanany != 0) != 0ull;
  const size_t o = (size_t)q * stride * keff;
  for (int base = 0; base < total; base += 64) {
    const int e = base + lane;
    const float cv = e < total ? lv[o + e] : INFINITY;
    const int cr = e < total ? li[o + e] : INT_MAX;
    unsigned long long mk = __ballot(cr != INT_MAX && wl.accepts(cv, cr));
    while (mk) {
      const int l = __ffsll((long long)mk) - 1;
      mk &= mk - 1;
      const float v = readlane_f(cv, l);
      const int r = readlane_i(cr, l);
      if (wl.accepts(v, r)) wl.insert(v, r, keff, lane);
    }
  }
  const int live = __popcll(__ballot(lane < K && wl.i != INT_MAX));
  if (lane < K) {
    const bool ok = wl.i != INT_MAX;
    out_idx[(size_t)q * K + lane] = ok ? wl.i : -1;
    out_dist[(size_t)q * K + lane] = ok ? wl.v : INFINITY;
  }
  const float nv = __shfl_down(wl.v, 1);
  const int ni = __shfl_down(wl.i, 1);
  const bool tie = wl.i != INT_MAX && ni != INT_MAX && wl.v == nv && lane + 1 < keff;
  const bool flagged = __ballot(tie) != 0ull || nanany;
  if (lane == 0) {
    if (out_count) out_count[q] = live;
    if (flagged) qlist[atomicAdd(qcount, 1)] = q;
  }
}

// residual dataset in grouped order: out[i] = X[perm[i]] - gcent[group_of[i]]
__global__ void gq_residuals(const float *__restrict__ X, int d, const int *__restrict__ perm,
                             const int *__restrict__ group_of, const float *__restrict__ gcent, long long total,
                             float *__restrict__ out) {
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (long long)gridDim.x * blockDim.x) {
    const long long i = t / d;
    const int e = (int)(t - i * d);
    out[t] = X[(size_t)perm[i] * d + e] - gcent[(size_t)group_of[i] * d + e];
  }
}

void run_grouped_query(gulon_grouped_index *gx, const float *dQ, int B, int K, int strategy, int limit, int *d_oi,
                       float *d_od, int *d_oc, hipStream_t st) {
  GULON_REQUIRE(B >= 0 && K >= 0, "k and batch size must be non-negative");
  GULON_REQUIRE(strategy == 0 || strategy == 1, "strategy must be 0 (LimitGroups) or 1 (LimitVectors)");
  GULON_REQUIRE(limit >= 0, "limit must be non-negative");
  // k_nn > GULON_MAX_K (Tests.scala asks for up to 1000): the literal kernels with the heaps in LDS, for every query
  constexpr int GROUPED_MAX_K_BIG = 2048;
  const bool big_k = K > GULON_MAX_K;
  GULON_UNSUPPORTED(K > GROUPED_MAX_K_BIG, "k_nn = %d > %d is not supported by the grouped index", K, GROUPED_MAX_K_BIG);
  if (B == 0) return;
  gulon_index *ix = gx->pq;
  const int g = gx->g;
  GULON_UNSUPPORTED(big_k && ix->wide, "k_nn = %d > GULON_MAX_K = %d is not supported by a grouped index with 16-bit codes", K,
                    GULON_MAX_K);
  if (big_k) {   // the per-group heaps are B x groups x k_nn entries: batches of queries that keep them under 2 GiB
    const long long per_query = (long long)std::max(1, std::min(strategy == 1 ? g : limit, g)) * K;
    const int sub = (int)std::max<long long>(1, std::min<long long>(B, (1ll << 28) / std::max<long long>(1, per_query)));
    if (sub < B) {
      for (int q0 = 0; q0 < B; q0 += sub) {
        const int nb = std::min(sub, B - q0);
        run_grouped_query(gx, dQ + (size_t)q0 * gx->d, nb, K, strategy, limit, d_oi + (size_t)q0 * K, d_od + (size_t)q0 * K,
                          d_oc ? d_oc + q0 : nullptr, st);
      }
      return;
    }
  }
  if (K == 0) {
    if (d_oc) HIP_CHECK(hipMemsetAsync(d_oc, 0, sizeof(int) * (size_t)B, st));
    return;
  }
  // groups searched per query: at most `nn_stride` (LimitVectors: every non-empty group holds >= 1 row; the
  // reference's leading empty group adds nothing to the count and is searched on top)
  const int nn_stride = std::max(1, (int)std::min<long long>((long long)limit + (strategy == 1 ? gx->n_empty : 0), g));
  int stride = nn_stride;
  int lit_cap = 0;          // > 0: group selection went through the (distance, id) sorts; capacity of the literal heap
  gx->cdist.ensure((size_t)B * g);
  gx->nn.ensure((size_t)B * nn_stride);
  gx->nn_cnt.ensure((size_t)B);
  hipLaunchKernelGGL(gq_cdist, dim3(ceil_div(g, 256), ceil_div(B, CD_Q)), dim3(256), 0, st, gx->gcent_t.p, g, gx->d, dQ, B,
                     gx->cdist.p);
  if (strategy == 0 && limit <= GULON_MAX_K && limit >= 1) {
    hipLaunchKernelGGL(gq_nearest_groups, dim3(B), dim3(64), 0, st, gx->cdist.p, g, limit, gx->nn.p, nn_stride,
                       gx->nn_cnt.p);
  } else {
    int n2 = 64;
    while (n2 < g) n2 <<= 1;
    const size_t lds = (size_t)n2 * 8;
    GULON_UNSUPPORTED(lds > 144 * 1024, "%d groups: ordering all of them needs %zu B of LDS (> 144 KiB)", g, lds);
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(gq_sorted_groups),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int *done = nullptr;
    gx->lit_flag.ensure((size_t)B);
    if (strategy == 0 && limit >= 1 && limit * 4 <= g) {
      // few of many: radix-select + sort of the selected; the full sort only runs for queries it gave up on
      int cap = 256;
      while (cap < limit + 128) cap <<= 1;
      gx->sel_ok.ensure((size_t)B);
      const size_t sel_lds = (size_t)cap * 8;
      auto kern = g <= 256 * 8 ? gq_select_groups<8> : g <= 256 * 40 ? gq_select_groups<40> : gq_select_groups<0>;
      HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sel_lds));
      hipLaunchKernelGGL(kern, dim3(B), dim3(256), sel_lds, st, gx->cdist.p, g, limit, cap, gx->nn.p,
                         nn_stride, gx->nn_cnt.p, gx->sel_ok.p, gx->lit_flag.p);
      done = gx->sel_ok.p;
    }
    hipLaunchKernelGGL(gq_sorted_groups, dim3(B), dim3(256), lds, st, gx->cdist.p, g, n2, gx->bounds.p, strategy == 1,
                       limit, gx->nn.p, nn_stride, gx->nn_cnt.p, done, gx->lit_flag.p);
    {
      // queries whose answer hangs on equally distant centroids (or NaN distances): the reference's heap, literally
      const int hcap = strategy == 1 ? g : std::min(limit, g);
      if (hcap >= 1) {
        const size_t hl = (size_t)hcap * 8;
        GULON_UNSUPPORTED(hl > 144 * 1024, "%d groups: the literal heap needs %zu B of LDS (> 144 KiB)", hcap, hl);
        HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(gq_literal_groups),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)hl));
        hipLaunchKernelGGL(gq_literal_groups, dim3(B), dim3(64), hl, st, gx->cdist.p, g, hcap, gx->bounds.p, strategy == 1,
                           limit, gx->lit_flag.p, 2, (const int *)nullptr, (const int *)nullptr, gx->nn.p, nn_stride,
                           gx->nn_cnt.p);
        lit_cap = hcap;
        if (ix->wide || big_k || getenv("GULON_GROUPED_LITERAL") != nullptr) {
          // every query's result comes from the literal heaps, merged in search order: the order-only ties as well
          hipLaunchKernelGGL(gq_literal_groups, dim3(B), dim3(64), hl, st, gx->cdist.p, g, hcap, gx->bounds.p, strategy == 1,
                             limit, gx->lit_flag.p, 1, (const int *)nullptr, (const int *)nullptr, gx->nn.p, nn_stride,
                             gx->nn_cnt.p);
          lit_cap = 0;
        }
      }
    }
    if (strategy == 1 && nn_stride > 64) {
      // LimitVectors rarely needs more than a handful of groups: size the per-group heaps by the
      // largest count of this batch (one small read-back) instead of by the worst case
      std::vector<int> h((size_t)B);
      HIP_CHECK(hipMemcpyAsync(h.data(), gx->nn_cnt.p, sizeof(int) * (size_t)B, hipMemcpyDeviceToHost, st));
      HIP_CHECK(hipStreamSynchronize(st));
      int mx = 1;
      for (int v : h) mx = std::max(mx, v);
      stride = mx;
    }
  }
  HIP_CHECK(hipGetLastError());
  const int keff = K + 1;
  if (ix->wide) {
    // 16-bit codes: the literal heaps for every (query, group) pair, residual tables in global scratch (<= 1 GiB)
    gx->hk.ensure((size_t)B * stride * K);
    gx->hv.ensure((size_t)B * stride * K);
    gx->hs.ensure((size_t)B * stride);
    const size_t slot = (size_t)ix->m * ix->k * sizeof(float);
    const long long pairs = (long long)B * stride;
    const int blocks = (int)std::max<long long>(1, std::min<long long>(std::min<long long>(pairs, 4096),
                                                                      std::max<long long>(64, (1ll << 30) / (long long)slot)));
    gx->wide_tables.ensure((size_t)blocks * ix->m * ix->k);
    HIP_CHECK(hipMemsetAsync(gx->hs.p, 0, sizeof(int) * (size_t)B * stride, st));
    hipLaunchKernelGGL(gq_group_scan_wide, dim3(blocks), dim3(64), sizeof(float) * (size_t)ix->d, st, ix->wcodes.p, ix->m,
                       ix->k, ix->d, ix->cents.p, ix->from.p, ix->sdim.p, gx->gcent.p, gx->bounds.p, dQ, gx->nn.p, nn_stride,
                       gx->nn_cnt.p, stride, B, K, gx->wide_tables.p, gx->hk.p, gx->hv.p, gx->hs.p);
    hipLaunchKernelGGL(gq_merge<false>, dim3(B), dim3(64), 0, st, gx->hk.p, gx->hv.p, gx->hs.p, gx->nn_cnt.p, stride, K, d_oi,
                       d_od, d_oc, (const int *)nullptr, (const int *)nullptr);
    HIP_CHECK(hipGetLastError());
    return;
  }
  const bool literal_only = big_k || getenv("GULON_GROUPED_LITERAL") != nullptr;   // (the variable: a testing aid)
  gx->hk.ensure((size_t)B * stride * keff);
  gx->hv.ensure((size_t)B * stride * keff);
  gx->hs.ensure((size_t)B * stride);
  gx->qlist.ensure((size_t)B);
  gx->qcount.ensure(1);
  int smax = 1;
  {
    std::vector<int> fr, un;
    subvectors(ix->d, ix->m, fr, un);
    for (int j = 0; j < ix->m; j++) smax = std::max(smax, un[j] - fr[j]);
  }
  const int slice_floats = 256 * smax;               // [x][256], transposed
  const size_t lds = ((size_t)GQ_WAVES * (ix->m_pad * 256 + ix->d) + (size_t)slice_floats + (big_k ? (size_t)GQ_WAVES * 2 * K : 0)) *
                     sizeof(float);
  GULON_UNSUPPORTED(lds > 160 * 1024, "grouped query needs %zu B of LDS (m = %d, d = %d, k_nn = %d)", lds, ix->m, ix->d, K);
  auto scan = [&](bool literal, int gy, const int *qlist, const int *qcount) {
#define GS3(V) GS_(V, true, true)
#define GS(V, L) GS_(V, L, false)
#define GS_(V, L, BG)                                                                                               \
    {                                                                                                               \
      auto kern = gq_group_scan<V, L, BG>;                                                                             \
      HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),                                           \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                         \
      hipLaunchKernelGGL(kern, dim3(ceil_div(stride, GQ_WAVES), gy), dim3(64 * GQ_WAVES), lds, st, ix->codes.p, ix->ng, \
                         ix->m, ix->m_pad, ix->k,                                                                  \
                         ix->d, ix->cents.p, ix->from.p, ix->sdim.p, gx->gcent.p, gx->bounds.p, dQ, gx->nn.p,       \
                         nn_stride, gx->nn_cnt.p, stride, K, gx->hk.p, gx->hv.p, gx->hs.p, qlist, qcount,          \
                         slice_floats);                                                                            \
    }
    if (big_k) { if (ix->vec == 16) GS3(16) else GS3(4) }
    else if (ix->vec == 16) { if (literal) GS(16, true) else GS(16, false) }
    else               { if (literal) GS(4, true) else GS(4, false) }
#undef GS
#undef GS3
    HIP_CHECK(hipGetLastError());
  };
  if (literal_only) {
    scan(true, B, nullptr, nullptr);
    if (big_k)
      hipLaunchKernelGGL(gq_merge<true>, dim3(B), dim3(64), sizeof(float) * 2 * (size_t)K, st, gx->hk.p, gx->hv.p, gx->hs.p,
                         gx->nn_cnt.p, stride, K, d_oi, d_od, d_oc, (const int *)nullptr, (const int *)nullptr);
    else
    hipLaunchKernelGGL(gq_merge<false>, dim3(B), dim3(64), 0, st, gx->hk.p, gx->hv.p, gx->hs.p, gx->nn_cnt.p, stride, K, d_oi,
                       d_od, d_oc, (const int *)nullptr, (const int *)nullptr);
    HIP_CHECK(hipGetLastError());
    return;
  }
  // fast path for every query, then the literal heaps for the tie-flagged ones (usually none)
  HIP_CHECK(hipMemsetAsync(gx->qcount.p, 0, sizeof(int), st));
  static const bool approx_off = [] { const char *e = getenv("GULON_GROUPED_APPROX"); return e && atoi(e) == 0; }();
  const size_t lds_ga = ((size_t)ix->m_pad * 256 + ix->d) * sizeof(float);
  if (!approx_off && gx->xnorm.n > 0 && lds_ga <= 150 * 1024) {
    // approximate pre-selection with one table per query, exact re-ranking of 64 candidates, certificate
    gx->ptab.ensure((size_t)B * ix->m_pad * 256);
    gx->apv.ensure((size_t)B * GA_WAVES * GA_C); gx->api.ensure((size_t)B * GA_WAVES * GA_C);
    gx->amv.ensure((size_t)B * GA_C); gx->ami.ensure((size_t)B * GA_C);
    gx->anan.ensure((size_t)B * GA_WAVES);
    {
      auto kern = ix->vec == 16 ? gq_approx_scan<16> : gq_approx_scan<4>;
      HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)lds_ga));
      // by group with 8-bit bound tables (grouped_filter.hip) where a query searches more than a handful of groups; else
      // every searched row through this kernel
      const bool by_group = gx->gfilter.built && group_filter_applies(ix->m, ix->m_pad, ix->ng, ix->vec, ix->k, ix->d) &&
                            nn_stride > GF_SAMPLE_GROUPS && B <= 65535 &&   // (a grid's y extent carries the query)
                            (long long)B * nn_stride <= (1ll << 28) &&      // (the pairs' tiles)
                            (long long)B * g <= (1ll << 27);                // (the groups' query lists, room for all: 512 MiB at most)
      if (!by_group)   // (the by-group path builds the tables where it quantizes them)
        hipLaunchKernelGGL(gq_ptables, dim3(ix->m_pad, B), dim3(256), 0, st, ix->cents.p, ix->from.p, ix->sdim.p, ix->d, ix->m,
                           ix->m_pad, ix->k, dQ, gx->ptab.p);
      if (by_group)
        group_filter_run(gx->gfilter, ix->codes.p, ix->ng, ix->vec, ix->m, ix->m_pad, ix->k, ix->d, gx->ptab.p, ix->cents.p,
                         ix->from.p, ix->sdim.p, gx->xnorm.p, gx->xnmax,
                         gx->gcent.p, gx->bounds.p, g, dQ, gx->cdist.p, gx->nn.p, nn_stride, gx->nn_cnt.p, B, gx->amv.p, gx->ami.p,
                         gx->anan.p, st);
      else {
        hipLaunchKernelGGL(kern, dim3(B), dim3(64 * GA_WAVES), lds_ga, st, ix->codes.p, ix->ng, ix->m_pad, ix->d, gx->ptab.p,
                           gx->xnorm.p, gx->gcent.p, gx->bounds.p, dQ, gx->nn.p, nn_stride, gx->nn_cnt.p, INT_MAX, 1, gx->apv.p,
                           gx->api.p, gx->anan.p);
        launch_merge(false, gx->apv.p, gx->api.p, GA_WAVES, (long long)GA_C, (long long)GA_WAVES * GA_C, B, GA_C - 1, nullptr,
                     nullptr, nullptr, nullptr, gx->amv.p, gx->ami.p, st);
      }
    }
    hipLaunchKernelGGL(gq_rerank, dim3(B), dim3(64), 0, st, ix->codes.p, ix->ng, ix->vec, ix->m, ix->k, ix->d, ix->cents.p,
                       ix->from.p, ix->sdim.p, gx->gcent.p, gx->bounds.p, g, dQ, gx->amv.p, gx->ami.p, gx->anan.p, gx->xnmax,
                       K, d_oi, d_od, d_oc, gx->qlist.p, gx->qcount.p);
    HIP_CHECK(hipGetLastError());
    if (getenv("GULON_GROUPED_STATS")) {   // debugging aid
      HIP_CHECK(hipStreamSynchronize(st));
      int nfl = 0;
      HIP_CHECK(hipMemcpy(&nfl, gx->qcount.p, sizeof(int), hipMemcpyDeviceToHost));
      fprintf(stderr, "[grouped] approximate pre-selection: %d of %d queries redone with literal heaps\n", nfl, B);
    }
    if (lit_cap > 0) {   // listed queries whose group ORDER hangs on equal centroid distances: the reference's heap order
      hipLaunchKernelGGL(gq_literal_groups, dim3(std::min(B, 16)), dim3(64), (size_t)lit_cap * 8, st, gx->cdist.p, g, lit_cap,
                         gx->bounds.p, strategy == 1, limit, gx->lit_flag.p, 1, gx->qlist.p, gx->qcount.p, gx->nn.p, nn_stride,
                         gx->nn_cnt.p);
      HIP_CHECK(hipGetLastError());
    }
    const int fy = std::min(B, 16);
    scan(true, fy, gx->qlist.p, gx->qcount.p);
    hipLaunchKernelGGL(gq_merge<false>, dim3(fy), dim3(64), 0, st, gx->hk.p, gx->hv.p, gx->hs.p, gx->nn_cnt.p, stride, K, d_oi,
                       d_od, d_oc, gx->qlist.p, gx->qcount.p);
    HIP_CHECK(hipGetLastError());
    return;
  }
  const size_t lds_qm = (4 * (size_t)smax * 256 + (size_t)QM_WAVES * (512 + ix->d)) * sizeof(float);
  const char *qm_env = getenv("GULON_GROUPED_QM");   // testing aid: 0 = never, 1 = whenever it applies
  const bool qm_off = qm_env && atoi(qm_env) == 0, qm_force = qm_env && atoi(qm_env) == 1;
  // (from two workgroups of pairs per query on: 10 M rows / LimitGroups(500) 12.6 -> 8.4 ms per batch,
  //  1 M rows / LimitGroups(50) 2.06 -> 1.92 ms)
  if (!qm_off && (stride >= 2 * QM_WAVES || qm_force) && smax <= 16 && lds_qm <= 150 * 1024) {
    const int mh = (ix->m + 1) / 2;
    if (gx->codes2.n == 0) {   // second code layout, built once
      const size_t nblk = (size_t)ceil_div(ix->n, 64);
      const long long total = (long long)nblk * mh * 64;
      const size_t padded = (size_t)total + (size_t)QM_RB * mh * 64;   // gq_scan_qm reads whole batches of row blocks
      gx->codes2.alloc(padded);
      HIP_CHECK(hipMemsetAsync(gx->codes2.p, 0, padded * sizeof(uint16_t), st));
      if (total > 0) {
        hipLaunchKernelGGL(gq_pair_codes, dim3((unsigned)ceil_div(total, 256LL)), dim3(256), 0, st, ix->codes.p, ix->ng,
                           ix->vec, ix->m, mh, gx->codes2.p, total);
        HIP_CHECK(hipGetLastError());
      }
    }
    const int *nn_qm = gx->nn.p;
    if (stride >= 2 * QM_WAVES && stride <= 2048) {   // several workgroups per query: size-sorted groups
      int npad = 1;
      while (npad < stride) npad <<= 1;
      gx->nn_sized.ensure((size_t)B * nn_stride);
      hipLaunchKernelGGL(gq_sort_by_size, dim3(B), dim3(256), (size_t)npad * sizeof(unsigned long long), st, gx->nn.p,
                         nn_stride, gx->nn_cnt.p, gx->bounds.p, npad, gx->nn_sized.p);
      HIP_CHECK(hipGetLastError());
      nn_qm = gx->nn_sized.p;
    }
#define QM(X)                                                                                                        \
    {                                                                                                               \
      auto kern = gq_scan_qm<X>;                                                                                    \
      HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),                                           \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_qm));                      \
      hipLaunchKernelGGL(kern, dim3(ceil_div(stride, QM_WAVES), B), dim3(64 * QM_WAVES), lds_qm, st, gx->codes2.p, mh, \
                         ix->m, ix->k, ix->d, ix->cents.p, ix->from.p, ix->sdim.p, gx->gcent.p, gx->bounds.p, dQ,    \
                         nn_qm, nn_stride, gx->nn_cnt.p, stride, K, gx->hk.p, gx->hv.p, gx->hs.p, smax);             \
    }
    if (smax <= 4) QM(2) else if (smax <= 8) QM(4) else if (smax <= 12) QM(6) else QM(8)
#undef QM
    HIP_CHECK(hipGetLastError());
  } else {
    scan(false, B, nullptr, nullptr);
  }
  hipLaunchKernelGGL(gq_merge_fast, dim3(B), dim3(64), 0, st, gx->hk.p, gx->hv.p, gx->hs.p, gx->nn_cnt.p, stride, K,
                     d_oi, d_od, d_oc, gx->qlist.p, gx->qcount.p);
  HIP_CHECK(hipGetLastError());
  if (getenv("GULON_GROUPED_STATS")) {   // debugging aid
    HIP_CHECK(hipStreamSynchronize(st));
    int nfl = 0;
    HIP_CHECK(hipMemcpy(&nfl, gx->qcount.p, sizeof(int), hipMemcpyDeviceToHost));
    fprintf(stderr, "[grouped] %d of %d queries redone with literal heaps; %d groups searched per query at most\n", nfl, B,
            stride);
  }
  {
    if (lit_cap > 0) {   // listed queries whose group ORDER hangs on equal centroid distances: the reference's heap order
      hipLaunchKernelGGL(gq_literal_groups, dim3(std::min(B, 16)), dim3(64), (size_t)lit_cap * 8, st, gx->cdist.p, g, lit_cap,
                         gx->bounds.p, strategy == 1, limit, gx->lit_flag.p, 1, gx->qlist.p, gx->qcount.p, gx->nn.p, nn_stride,
                         gx->nn_cnt.p);
      HIP_CHECK(hipGetLastError());
    }
  }
  const int fy = std::min(B, 16);
  scan(true, fy, gx->qlist.p, gx->qcount.p);
  hipLaunchKernelGGL(gq_merge<false>, dim3(fy), dim3(64), 0, st, gx->hk.p, gx->hv.p, gx->hs.p, gx->nn_cnt.p, stride, K, d_oi,
                     d_od, d_oc, gx->qlist.p, gx->qcount.p);
  HIP_CHECK(hipGetLastError());
}

}  // namespace
}  // namespace gulon

using namespace gulon;

GULON_API int32_t gulon_dataset_group_residuals(const gulon_dataset *ds, const int32_t *perm,
                                                const int32_t *group_of, const float *group_centroids, int32_t g,
                                                gulon_dataset **out) {
  return guarded([&] {
    GULON_REQUIRE(ds != nullptr && out != nullptr && (ds->n == 0 || (perm && group_of)) && group_centroids && g >= 1,
                  "bad arguments");
    *out = nullptr;
    const int n = ds->n, d = ds->d;
    for (int i = 0; i < n; i++) {
      GULON_REQUIRE(perm[i] >= 0 && perm[i] < n, "perm[%d] = %d out of range", i, perm[i]);
      GULON_REQUIRE(group_of[i] >= 0 && group_of[i] < g, "group_of[%d] = %d out of range", i, group_of[i]);
    }
    std::unique_ptr<gulon_dataset> r(new gulon_dataset());
    r->n = n; r->d = d;
    const long long total = (long long)n * d;
    r->x.alloc(std::max<size_t>((size_t)total, 1));
    if (total) {
      DevBuf<int> dp, dg;
      DevBuf<float> dc;
      dp.upload(perm, n); dg.upload(group_of, n); dc.upload(group_centroids, (size_t)g * d);
      hipLaunchKernelGGL(gq_residuals, dim3((unsigned)std::min<long long>(ceil_div(total, 256), 1 << 20)), dim3(256), 0,
                         0, ds->x.p, d, dp.p, dg.p, dc.p, total, r->x.p);
      HIP_CHECK(hipGetLastError());
      HIP_CHECK(hipDeviceSynchronize());
    }
    *out = r.release();
  });
}

GULON_API int32_t gulon_grouped_index_create(const uint8_t *codes, int32_t n, int32_t d, int32_t m, int32_t k,
                                             const float *pq_cents, const float *group_centroids,
                                             const int32_t *offsets, int32_t g, gulon_grouped_index **out) {
  return guarded([&] {
    GULON_REQUIRE(out != nullptr, "out is null");
    *out = nullptr;
    GULON_REQUIRE(g >= 1 && group_centroids != nullptr && (g == 1 || offsets != nullptr), "bad grouping");
    // GroupedIndex asserts centroids.length == offsets.length + 1 (Index.scala:240-241); offsets ascend
    std::vector<int> bounds((size_t)g + 1);
    bounds[0] = 0;
    for (int c = 1; c < g; c++) {
      GULON_REQUIRE(offsets[c - 1] >= bounds[c - 1] && offsets[c - 1] <= n, "group offsets must ascend within [0, n]");
      bounds[c] = offsets[c - 1];
    }
    bounds[g] = n;
    std::unique_ptr<gulon_grouped_index> gx(new gulon_grouped_index());
    for (int c = 0; c < g; c++) gx->n_empty += bounds[c + 1] == bounds[c];
    gulon_index *pq = nullptr;
    int32_t rc = gulon_index_create(codes, n, d, m, k, pq_cents, 0, &pq);
    if (rc != GULON_OK) throw DeviceError{rc};
    gx->pq = pq;
    gx->n = n; gx->d = d; gx->g = g;
    gx->gcent.upload(group_centroids, (size_t)g * d);
    {
      std::vector<float> tr((size_t)g * d);
      for (int c = 0; c < g; c++)
        for (int e = 0; e < d; e++) tr[(size_t)e * g + c] = group_centroids[(size_t)c * d + e];
      gx->gcent_t.upload(tr.data(), tr.size());
      HIP_CHECK(hipDeviceSynchronize());   // tr goes out of scope
    }
    gx->bounds.upload(bounds.data(), bounds.size());
    HIP_CHECK(hipDeviceSynchronize());
    if (!pq->wide && n > 0) {   // |g + decode(codes_i)|^2 per row: the per-row term of the approximate pre-selection
      gx->xnorm.alloc((size_t)n);
      DevBuf<unsigned> mx(1);
      HIP_CHECK(hipMemset(mx.p, 0, sizeof(unsigned)));
      hipLaunchKernelGGL(gq_row_norms, dim3(ceil_div(n, 256)), dim3(256), 0, 0, pq->codes.p, pq->ng, pq->vec, pq->m, pq->k, d,
                         pq->cents.p, pq->from.p, pq->sdim.p, gx->gcent.p, gx->bounds.p, g, n, gx->xnorm.p, mx.p);
      HIP_CHECK(hipGetLastError());
      unsigned h = 0;
      HIP_CHECK(hipMemcpy(&h, mx.p, sizeof(h), hipMemcpyDeviceToHost));
      memcpy(&gx->xnmax, &h, sizeof(float));
      if (gx->xnmax < INFINITY && group_filter_applies(pq->m, pq->m_pad, pq->ng, pq->vec, pq->k, d))
        group_filter_build(gx->gfilter, gx->xnorm.p, n, gx->gcent.p, gx->bounds.p, g, d);
    }
    *out = gx.release();
  });
}

GULON_API int32_t gulon_grouped_index_destroy(gulon_grouped_index *idx) {
  return guarded([&] { delete idx; });
}

GULON_API int32_t gulon_grouped_index_batch_query_dev(gulon_grouped_index *idx, const float *d_queries, int32_t b,
                                                      int32_t k_nn, int32_t strategy, int32_t limit,
                                                      int32_t *d_out_idx, float *d_out_dist, int32_t *d_out_count,
                                                      void *stream) {
  return guarded([&] {
    GULON_REQUIRE(idx != nullptr, "index is null");
    std::lock_guard<std::mutex> lock(idx->mu);
    run_grouped_query(idx, d_queries, b, k_nn, strategy, limit, d_out_idx, d_out_dist, d_out_count,
                      (hipStream_t)stream);
  });
}

GULON_API int32_t gulon_grouped_index_batch_query(gulon_grouped_index *idx, const float *queries, int32_t b,
                                                  int32_t k_nn, int32_t strategy, int32_t limit, int32_t *out_idx,
                                                  float *out_dist, int32_t *out_count) {
  return guarded([&] {
    GULON_REQUIRE(idx != nullptr, "index is null");
    GULON_REQUIRE(b >= 0 && k_nn >= 0, "k and batch size must be non-negative");
    std::lock_guard<std::mutex> lock(idx->mu);
    const size_t bk = (size_t)b * (size_t)k_nn;
    idx->q_dev.ensure((size_t)b * idx->d + 1);
    idx->oi.ensure(bk + 1); idx->od.ensure(bk + 1); idx->oc.ensure((size_t)b + 1);
    hipStream_t st = nullptr;
    if (b > 0) HIP_CHECK(hipMemcpyAsync(idx->q_dev.p, queries, sizeof(float) * (size_t)b * idx->d, hipMemcpyHostToDevice, st));
    run_grouped_query(idx, idx->q_dev.p, b, k_nn, strategy, limit, idx->oi.p, idx->od.p, idx->oc.p, st);
    if (bk) { idx->oi.download(out_idx, bk, st); idx->od.download(out_dist, bk, st); }
    if (b > 0 && out_count) idx->oc.download(out_count, b, st);
    HIP_CHECK(hipStreamSynchronize(st));
  });
}
