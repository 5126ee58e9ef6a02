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
// which depends only on the multiset of earlier distances, not on the heap's shape.  Hence
//   1. rp_seg_scan<0>: every segment of rows computes its K smallest distances        (parallel)
//   2. rp_prefix:      exclusive prefix-merge over segments -> K smallest before each segment
//   3. rp_seg_scan<1>: every segment replays its rows in order against that prefix and
//                      emits its events (row, distance)                                  (parallel)
//   4. rp_heap:        one thread per query pushes the ~K ln(n/K) events, in row order,
//                      through a literal TopKHeap and drains it like Result.fromHeap.
// Distances are the same bit-exact j-ordered sums as in the main scan.
#include "scan.hpp"

namespace gulon {

constexpr int RP_MAXF = 1024;   // flagged queries replayed per batch (all of a 1024-query batch)
constexpr int RP_MAXSEG = 512;

__global__ void rp_collect(const int *__restrict__ flags, int B, int maxf, int *__restrict__ list,
                           int *__restrict__ count) {
  int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q < B && (flags[q] & (GULON_FLAG_BOUNDARY_TIE | GULON_FLAG_INTERIOR_TIE))) {
    int p = atomicAdd(count, 1);
    if (p < maxf) list[p] = q;
  }
}

__global__ void rp_gather_queries(const float *__restrict__ Q, int d, int maxf, const int *__restrict__ list,
                                  const int *__restrict__ count, float *__restrict__ Qf) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  int f = t / d, c = t - f * d;
  if (f >= maxf) return;
  int nf = min(*count, maxf);
  Qf[t] = f < nf ? Q[(size_t)list[f] * d + c] : 0.f;
}

template <int VEC> struct RpWord;
template <> struct RpWord<4> { using type = uint32_t; };
template <> struct RpWord<16> { using type = uint4; };
__device__ inline uint32_t rp_byte(const uint32_t &w, int b) { return (w >> (8 * b)) & 0xFFu; }
__device__ inline uint32_t rp_byte(const uint4 &w, int b) {
  uint32_t x = (b < 4) ? w.x : (b < 8) ? w.y : (b < 12) ? w.z : w.w;
  return (x >> (8 * (b & 3))) & 0xFFu;
}

// One wave per (segment, flagged query).  PHASE 0: K smallest distances of the segment.
// PHASE 1: events of the segment given the K smallest distances of all earlier rows.
template <int VEC, int PHASE>
__global__ __launch_bounds__(64) void rp_seg_scan(const uint8_t *__restrict__ codes, int ng, int m_pad,
                                                  const float *__restrict__ tables /*[f][m_pad][256]*/,
                                                  const int *__restrict__ count, int maxf, int row_from, int row_until,
                                                  int row_base, int rb_begin, int rb_end, int rb_per_seg, int nseg,
                                                  int K, float *__restrict__ segtop, int *__restrict__ segcnt,
                                                  const float *__restrict__ prefix, const int *__restrict__ precnt,
                                                  int evcap, float *__restrict__ evv, int *__restrict__ evi,
                                                  int *__restrict__ evcnt, int *__restrict__ overflow) {
  using Word = typename RpWord<VEC>::type;
  extern __shared__ float tab[];   // m_pad * 256
  const int s = blockIdx.x, lane = threadIdx.x;
  const int nf = min(*count, maxf);
  // the grid has a fixed, small y extent; each block walks the flagged queries f = y, y+Y, ...
  // (an empty launch then costs a few thousand blocks instead of nseg * maxf)
  for (int f = blockIdx.y; f < nf; f += gridDim.y) {
  __syncthreads();
  {
    const float *src = tables + (size_t)f * m_pad * 256;
    for (int e = lane; e < m_pad * 256; e += 64) tab[e] = src[e];
  }
  __syncthreads();
  const size_t fs = (size_t)f * nseg + s;
  WaveList wl;
  wl.init();
  int cnt = 0;
  if (PHASE == 1) {
    cnt = precnt[fs];
    if (lane < cnt) { wl.v = prefix[fs * K + lane]; wl.i = -1; }   // earlier rows: sort before any tie
    if (cnt >= K) { wl.tau = readlane_f(wl.v, K - 1); wl.tau_i = -1; }
  }
  int nev = 0;
  const Word *cw = reinterpret_cast<const Word *>(codes);
  const int rb0 = rb_begin + s * rb_per_seg;
  const int rb1 = min(rb_end, rb0 + rb_per_seg);
  // one wave walks the segment alone: keep the next row block's first code word in flight
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
        if (PHASE == 1) {
          if (nev < evcap) {
            if (lane == 0) { evv[fs * evcap + nev] = cv; evi[fs * evcap + nev] = cr; }
          } else if (lane == 0) {
            overflow[f] = 1;
          }
          nev++;
        }
        wl.insert(cv, cr, K, lane);
        if (cnt < K) cnt++;
      }
    }
  }
  if (PHASE == 0) {
    if (lane < K) segtop[fs * K + lane] = wl.v;
    if (lane == 0) segcnt[fs] = cnt;
  } else if (lane == 0) {
    evcnt[fs] = min(nev, evcap);
  }
  }
}

// exclusive prefix over segments of "the K smallest distances so far"; one wave per query.
// Segment s+1's list is prefetched (one coalesced load, lane e = entry e) while s is merged.
__global__ __launch_bounds__(64) void rp_prefix(const float *__restrict__ segtop, const int *__restrict__ segcnt,
                                                const int *__restrict__ count, int maxf, int nseg, int K,
                                                float *__restrict__ prefix, int *__restrict__ precnt) {
  const int lane = threadIdx.x;
  const int nf = min(*count, maxf);
  for (int f = blockIdx.x; f < nf; f += gridDim.x) {
    WaveList wl;
    wl.init();
    int cnt = 0;
    const size_t f0 = (size_t)f * nseg;
    float nxt_v = lane < K ? segtop[f0 * K + lane] : INFINITY;
    int nxt_c = segcnt[f0];
    for (int s = 0; s < nseg; s++) {
      const size_t fs = f0 + s;
      const float cur_v = nxt_v;
      const int sc = nxt_c;
      if (s + 1 < nseg) {
        nxt_v = lane < K ? segtop[(fs + 1) * K + lane] : INFINITY;
        nxt_c = segcnt[fs + 1];
      }
      if (lane < K) prefix[fs * K + lane] = wl.v;
      if (lane == 0) precnt[fs] = cnt;
      for (int e = 0; e < sc; e++) {
        float cv = readlane_f(cur_v, e);
        if (cnt < K || cv < wl.tau) {
          wl.insert(cv, s * 64 + e, K, lane);
          if (cnt < K) cnt++;
        } else {
          break;   // segtop is ascending: nothing smaller follows
        }
      }
    }
  }
}

// literal TopKHeap (TopKHeap.scala) over the events, then Result.fromHeap (Index.scala:83-94).
// One wave per flagged query: the lanes first pack the per-segment event lists into one
// contiguous LDS array (wave prefix sum over the segment counts), then lane 0 runs the heap.
constexpr int RP_LDS_EVENTS = 4096;
__global__ __launch_bounds__(64) void rp_heap(const float *__restrict__ evv, const int *__restrict__ evi,
                                              const int *__restrict__ evcnt, const int *__restrict__ overflow,
                                              const int *__restrict__ list, const int *__restrict__ count, int maxf,
                                              int nseg, int evcap, int K, int *__restrict__ out_idx,
                                              float *__restrict__ out_dist, int *__restrict__ out_count,
                                              int *__restrict__ out_flags) {
  __shared__ float sv[RP_LDS_EVENTS];
  __shared__ int si[RP_LDS_EVENTS];
  __shared__ int hk[GULON_MAX_K + 1];
  __shared__ float hv[GULON_MAX_K + 1];
  const int lane = threadIdx.x;
  const int nf = min(*count, maxf);
  for (int f = blockIdx.x; f < nf; f += gridDim.x) {
    __syncthreads();
    if (overflow[f]) continue;   // keeps the (distance, row id) result and its tie flags
    // pack events: segment order = row order
    int total = 0;
    for (int s0 = 0; s0 < nseg; s0 += 64) {
      const int s = s0 + lane;
      const size_t fs = (size_t)f * nseg + s;
      const int c = s < nseg ? evcnt[fs] : 0;
      int inc = c;
      for (int o = 1; o < 64; o <<= 1) {
        int u = __shfl_up(inc, o);
        if (lane >= o) inc += u;
      }
      const int base = total + inc - c;
      for (int e = 0; e < c; e++) {
        if (base + e < RP_LDS_EVENTS) { sv[base + e] = evv[fs * evcap + e]; si[base + e] = evi[fs * evcap + e]; }
      }
      total += __shfl(inc, 63);
    }
    __syncthreads();
    if (total > RP_LDS_EVENTS) continue;   // too many events for the LDS staging: leave the flagged result
    if (lane == 0) {
      int size = 0;
      auto swp = [&](int a, int b) {
        int tk = hk[a]; float tv = hv[a];
        hk[a] = hk[b]; hv[a] = hv[b];
        hk[b] = tk; hv[b] = tv;
      };
      auto down = [&](int i) {                                  // percolateDown, TopKHeap.scala:30-42
        for (;;) {
          int top = i, lc = 2 * i + 1, rc = 2 * i + 2;
          if (lc < size && hv[top] < hv[lc]) top = lc;
          if (rc < size && hv[top] < hv[rc]) top = rc;
          if (top == i) break;
          swp(i, top);
          i = top;
        }
      };
      auto del = [&]() {                                        // delete, TopKHeap.scala:57-67
        size -= 1;
        hk[0] = hk[size];
        hv[0] = hv[size];
        down(0);
      };
      for (int e = 0; e < total; e++) {
        const float v = sv[e];
        const int kk = si[e];
        if (size == K && hv[0] > v) del();                      // update, TopKHeap.scala:69-79
        if (size < K) {
          hk[size] = kk;
          hv[size] = v;
          int i = size;
          while (i > 0) {                                       // percolateUp, TopKHeap.scala:21-28
            int p = (i - 1) / 2;
            if (hv[i] > hv[p]) { swp(i, p); i = p; } else break;
          }
          size += 1;
        }
      }
      const int q = list[f];
      const int live = size;
      for (int i = live - 1; i >= 0; i--) {                     // Result.fromHeap: max first, fill from the back
        out_idx[(size_t)q * K + i] = hk[0];
        out_dist[(size_t)q * K + i] = hv[0];
        del();
      }
      for (int i = live; i < K; i++) { out_idx[(size_t)q * K + i] = -1; out_dist[(size_t)q * K + i] = INFINITY; }
      if (out_count) out_count[q] = live;
      out_flags[q] |= GULON_FLAG_EXACT_REPLAY;
    }
  }
}

void run_tie_replay(gulon_index *ix, const float *dQ, int B, int K, int from, int until, int *d_oi, float *d_od,
                    int *d_oc, int *d_of, hipStream_t st) {
  if (B <= 0 || K <= 0 || until <= from) return;
  const int rb_begin = from / 64, rb_end = ceil_div(until, 64), rb_total = rb_end - rb_begin;
  const int maxf = std::min(B, RP_MAXF);
  int nseg = std::min(maxf <= 64 ? RP_MAXSEG : 256, std::max(1, rb_total / 4));
  const int gy = std::min(maxf, 16);   // y extent of the segment-scan grids
  const int rb_per_seg = ceil_div(rb_total, nseg);
  nseg = ceil_div(rb_total, rb_per_seg);
  const int evcap = 64 + 8 * K;
  const size_t FS = (size_t)maxf * nseg;
  ix->rp_list.ensure(maxf); ix->rp_count.ensure(1); ix->rp_overflow.ensure(maxf);
  ix->rp_q.ensure((size_t)maxf * ix->d);
  ix->rp_tables.ensure((size_t)maxf * ix->m_pad * 256);
  ix->rp_segtop.ensure(FS * K); ix->rp_segcnt.ensure(FS);
  ix->rp_prefix.ensure(FS * K); ix->rp_precnt.ensure(FS);
  ix->rp_evv.ensure(FS * evcap); ix->rp_evi.ensure(FS * evcap); ix->rp_evcnt.ensure(FS);
  HIP_CHECK(hipMemsetAsync(ix->rp_count.p, 0, sizeof(int), st));
  HIP_CHECK(hipMemsetAsync(ix->rp_overflow.p, 0, sizeof(int) * maxf, st));
  hipLaunchKernelGGL(rp_collect, dim3(ceil_div(B, 256)), dim3(256), 0, st, d_of, B, maxf, ix->rp_list.p,
                     ix->rp_count.p);
  hipLaunchKernelGGL(rp_gather_queries, dim3(ceil_div((long long)maxf * ix->d, 256)), dim3(256), 0, st, dQ, ix->d,
                     maxf, ix->rp_list.p, ix->rp_count.p, ix->rp_q.p);
  launch_build_tables(1, ix, ix->rp_q.p, maxf, maxf, ix->rp_tables.p, st, ix->rp_count.p);
  const size_t lds = (size_t)ix->m_pad * 256 * sizeof(float);
#define SEG(V, PH)                                                                                                 \
  do {                                                                                                             \
    auto kern = rp_seg_scan<V, PH>;                                                                                \
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                  (int)lds));                                                                      \
    hipLaunchKernelGGL(kern, dim3(nseg, gy), dim3(64), lds, st, ix->codes.p, ix->ng, ix->m_pad,                  \
                       ix->rp_tables.p, ix->rp_count.p, maxf, from, until, ix->row_base, rb_begin, rb_end, rb_per_seg,   \
                       nseg, K, ix->rp_segtop.p, ix->rp_segcnt.p, ix->rp_prefix.p, ix->rp_precnt.p, evcap,          \
                       ix->rp_evv.p, ix->rp_evi.p, ix->rp_evcnt.p, ix->rp_overflow.p);                             \
  } while (0)
  if (ix->vec == 16) SEG(16, 0); else SEG(4, 0);
  hipLaunchKernelGGL(rp_prefix, dim3(std::min(maxf, 64)), dim3(64), 0, st, ix->rp_segtop.p, ix->rp_segcnt.p, ix->rp_count.p,
                     maxf, nseg, K, ix->rp_prefix.p, ix->rp_precnt.p);
  if (ix->vec == 16) SEG(16, 1); else SEG(4, 1);
#undef SEG
  hipLaunchKernelGGL(rp_heap, dim3(std::min(maxf, 64)), dim3(64), 0, st, ix->rp_evv.p, ix->rp_evi.p, ix->rp_evcnt.p,
                     ix->rp_overflow.p, ix->rp_list.p, ix->rp_count.p, maxf, nseg, evcap, K, d_oi, d_od, d_oc, d_of);
  HIP_CHECK(hipGetLastError());
}

}  // namespace gulon
