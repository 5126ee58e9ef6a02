// Non-finite distances: the literal TopKHeap (TopKHeap.scala:57-79) over ALL rows of the range.
//
// The fast paths order candidates by (distance, row id), which is the reference's answer whenever
// distances are ordinary numbers (ties: replay.hip).  A NaN distance breaks that: TopKHeap.update
// inserts a NaN while the heap is not full (`values(0) > v` is false, so nothing is deleted, and
// `size < capacity` appends it), percolateUp never moves it, and from then on comparisons against it
// are all false -- the heap is no longer a heap and what it returns is a function of the whole
// update sequence.  A query with a NaN component has EVERY distance NaN (each table entry of that
// quantizer is NaN, Index.scala:352-383), and the reference then returns the first K rows of the
// range in the order [from+1, ..., from+K-1, from]; +inf distances (a huge query) behave the same.
// NaN / inf centroids give a mix of NaN and ordinary distances.
//
// literal_nonfinite runs after the regular pipeline of an UNSHARDED final query: one wavefront per
// query tests whether any distance of that query CAN be non-finite (a non-finite query component, a
// non-finite centroid, or a magnitude bound that allows overflow); only such queries (normally none:
// the wave returns after reading the query) are redone literally -- the query's table is rebuilt with
// Index.prepareQuery's arithmetic, every row of [from, until) gets its j-ordered fp32 sum, and the
// updates go through the reference's heap (lane = slot) in row order; Result.fromHeap drains it.
// Exit as soon as the heap is full and its root is not above a lower bound of every remaining
// distance (for an all-NaN or all-inf query: after K rows).
//
// Row shards: the literal heap needs the rows in global order, so across shards only the all-NaN
// case is reproduced (nan_query_fix: closed form, needs nothing but the query) -- NaN centroids on a
// sharded index keep the (distance, row id) rule (deviations).
#include "scan.hpp"

namespace gulon {

namespace {

// the reference's heap in registers: lane i = slot i (K <= 63); every index below is wave-uniform
struct RegHeap {
  float hv = 0.f;
  int hk = 0;
  int size = 0;
  int lane;
  __device__ float val(int i) const { return readlane_f(hv, i); }
  __device__ void swp(int a, int b) {
    const float va = readlane_f(hv, a), vb = readlane_f(hv, b);
    const int ka = readlane_i(hk, a), kb = readlane_i(hk, b);
    if (lane == a) { hv = vb; hk = kb; }
    if (lane == b) { hv = va; hk = ka; }
  }
  __device__ void down(int i) {                       // percolateDown, TopKHeap.scala:30-42
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
  __device__ void del() {                             // delete, TopKHeap.scala:57-67
    size -= 1;
    const float lv = readlane_f(hv, size);
    const int lk = readlane_i(hk, size);
    if (lane == 0) { hv = lv; hk = lk; }
    down(0);
  }
  __device__ void update(int key, float v, int K) {   // update, TopKHeap.scala:69-79
    if (size == K && val(0) > v) del();
    if (size < K) {
      if (lane == size) { hv = v; hk = key; }
      int i = size;
      while (i > 0) {                                 // percolateUp, TopKHeap.scala:21-28
        const int p = (i - 1) / 2;
        if (val(i) > val(p)) { swp(i, p); i = p; } else break;
      }
      size += 1;
    }
  }
};

template <int VEC>
__global__ __launch_bounds__(64) void literal_nonfinite(
    const uint8_t *__restrict__ codes, int ng, int m_pad, int m, int k, int d, const float *__restrict__ cents,
    const int *__restrict__ from, const int *__restrict__ sdim, const float *__restrict__ Q, int B, int K,
    int row_from, int row_until, int row_base, float cmax /* max |centroid coordinate|, +inf if any is not finite */,
    float *__restrict__ scratch /* [gridDim.x][m_pad * 256] */, int *__restrict__ out_idx, float *__restrict__ out_dist,
    int *__restrict__ out_count, int *__restrict__ out_flags) {
  using Word = typename CodeWord<VEC>::type;
  const int lane = threadIdx.x;
  float *T = scratch + (size_t)blockIdx.x * m_pad * 256;
  for (int q = blockIdx.x; q < B; q += gridDim.x) {
    // can any distance of this query be NaN or +inf?  |q_t - c_t| <= |q_t| + cmax, so every j-ordered fp32 sum
    // stays below sum_t (|q_t| + cmax)^2 * (1 + d 2^-23): finite if that is < 1e38
    bool bad = !(cmax < INFINITY);
    double bound = 0.0;
    for (int t = lane; t < d; t += 64) {
      const float x = Q[(size_t)q * d + t];
      if (!(fabsf(x) < INFINITY)) bad = true;
      const double a = (double)fabsf(x) + (double)cmax;
      bound += a * a;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) bound += __shfl_xor(bound, o);
    if (!(bound < 1e38)) bad = true;
    if (!__any(bad)) continue;

    // Index.prepareQuery (Index.scala:352-383) for this one query: T[j][c], sequential, unfused
    for (int e = lane; e < m_pad * 256; e += 64) {
      const int j = e >> 8, c = e & 255;
      float acc = 0.f;
      if (j < m && c < k) {
        const int fr = from[j], s = sdim[j];
        const float *cc = cents + (size_t)k * fr + (size_t)c * s;
        for (int t = 0; t < s; t++) {
          const float dd = Q[(size_t)q * d + fr + t] - cc[t];
          acc += dd * dd;
        }
      }
      T[e] = acc;
    }
    __threadfence();   // the table is read back by other lanes of this wave through the vector L1
    // lower bound of every non-NaN distance: j-ordered sum of the per-quantizer minima (fp32 addition of
    // non-negative terms is monotone); NaN when some quantizer has only NaN entries (then every distance is NaN)
    float lb = 0.f;
    for (int j = 0; j < m; j++) {
      float mn = INFINITY;
      bool any = false;
      for (int c = lane; c < k; c += 64) {
        const float v = T[j * 256 + c];
        if (v == v) { mn = fminf(mn, v); any = true; }
      }
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) mn = fminf(mn, __shfl_xor(mn, o));
      lb = __any(any) ? lb + mn : NAN;
    }

    RegHeap h;
    h.lane = lane;
    bool heap_has_nan = false;
    const Word *cw = reinterpret_cast<const Word *>(codes);
    const int rb0 = row_from / 64, rb1 = (row_until + 63) / 64;
    for (int rb = rb0; rb < rb1; rb++) {
      if (h.size == K && !(h.val(0) > lb)) break;   // no remaining row can satisfy `values(0) > v`
      float acc = 0.f;
      for (int g = 0; g < ng; g++) {
        const Word w = cw[((size_t)rb * ng + g) * 64 + lane];
        const float *tj = T + g * VEC * 256;
#pragma unroll
        for (int b = 0; b < VEC; b++) acc += tj[b * 256 + code_byte<VEC>(w, b)];
      }
      const int row = rb * 64 + lane;
      const bool valid = row >= row_from && row < row_until;
      // rows that can change the heap: all of them while it is not full or holds a NaN (its root then follows no
      // order); otherwise the root only falls, so `root > v` with the root of this moment is a superset
      unsigned long long mk = __ballot(valid && (h.size < K || heap_has_nan || h.val(0) > acc));
      while (mk) {
        const int l = __ffsll((long long)mk) - 1;
        mk &= mk - 1;
        const float v = readlane_f(acc, l);
        const int before = h.size;
        const bool ins = before < K || h.val(0) > v;
        h.update(rb * 64 + l + row_base, v, K);
        if (ins && !(v == v)) heap_has_nan = true;
        if (h.size == K && !heap_has_nan && before == K) {
          // full, ordered heap: drop the lanes the new (lower) root already rules out
          mk &= __ballot(valid && h.val(0) > acc);
        }
      }
    }
    // Result.fromHeap (Index.scala:83-94): max first, filled from the back
    const int live = h.size;
    for (int i = live - 1; i >= 0; i--) {
      const float tv = h.val(0);
      const int tk = readlane_i(h.hk, 0);
      if (lane == 0) { out_idx[(size_t)q * K + i] = tk; out_dist[(size_t)q * K + i] = tv; }
      h.del();
    }
    if (lane >= live && lane < K) { out_idx[(size_t)q * K + lane] = -1; out_dist[(size_t)q * K + lane] = INFINITY; }
    if (lane == 0) {
      if (out_count) out_count[q] = live;
      if (out_flags) out_flags[q] = GULON_FLAG_EXACT_REPLAY | GULON_FLAG_NONFINITE;
    }
  }
}

// All-NaN queries on a row-sharded index: a query with a NaN component has every distance NaN, and the
// reference's heap then holds the first min(K, n) rows of the index, drained as [1, ..., c-1, 0].
__global__ __launch_bounds__(64) void nan_query_fix(const float *__restrict__ Q, int B, int d, int K, int n_total,
                                                    int *__restrict__ out_idx, float *__restrict__ out_dist,
                                                    int *__restrict__ out_count, int *__restrict__ out_flags) {
  const int lane = threadIdx.x;
  for (int q = blockIdx.x; q < B; q += gridDim.x) {
    bool nan = false;
    for (int t = lane; t < d; t += 64) { const float x = Q[(size_t)q * d + t]; nan = nan || !(x == x); }
    if (!__any(nan)) continue;
    const int c = min(K, n_total);
    for (int e = lane; e < K; e += 64) {
      out_idx[(size_t)q * K + e] = e < c ? (e == c - 1 ? 0 : e + 1) : -1;
      out_dist[(size_t)q * K + e] = e < c ? NAN : INFINITY;
    }
    if (lane == 0) {
      if (out_count) out_count[q] = c;
      if (out_flags) out_flags[q] = GULON_FLAG_EXACT_REPLAY | GULON_FLAG_NONFINITE;
    }
  }
}

__global__ void absmax_kernel(const float *__restrict__ x, long long n, unsigned *__restrict__ out) {
  // max |x| as float bits (non-negative floats order like unsigned ints; NaN / inf map to +inf)
  unsigned best = 0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float a = fabsf(x[i]);
    const unsigned bits = (a < INFINITY) ? __float_as_uint(a) : 0x7F800000u;
    best = max(best, bits);
  }
  atomicMax(out, best);
}

}  // namespace

float centroid_absmax(const float *d_cents, long long count) {
  DevBuf<unsigned> out(1);
  HIP_CHECK(hipMemset(out.p, 0, sizeof(unsigned)));
  if (count > 0) {
    hipLaunchKernelGGL(absmax_kernel, dim3(256), dim3(256), 0, 0, d_cents, count, out.p);
    HIP_CHECK(hipGetLastError());
  }
  unsigned h = 0;
  HIP_CHECK(hipMemcpy(&h, out.p, sizeof(unsigned), hipMemcpyDeviceToHost));
  float f;
  memcpy(&f, &h, sizeof(f));
  return f;
}

void run_nonfinite_literal(gulon_index *ix, const float *dQ, int B, int K, int from, int until, int *d_oi, float *d_od,
                           int *d_oc, int *d_of, hipStream_t st) {
  if (B <= 0 || K <= 0 || K > GULON_MAX_K || ix->wide) return;
  const int grid = std::min(B, 1024);
  ix->nf_tables.ensure((size_t)grid * ix->m_pad * 256);
  auto kern = ix->vec == 16 ? literal_nonfinite<16> : literal_nonfinite<4>;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(64), 0, st, ix->codes.p, ix->ng, ix->m_pad, ix->m, ix->k, ix->d, ix->cents.p,
                     ix->from.p, ix->sdim.p, dQ, B, K, from, until, ix->row_base, ix->cents_absmax, ix->nf_tables.p, d_oi,
                     d_od, d_oc, d_of);
  HIP_CHECK(hipGetLastError());
}

}  // namespace gulon

using namespace gulon;

GULON_API int32_t gulon_nan_queries_fix_dev(const float *d_queries, int32_t b, int32_t d, int32_t k_nn, int32_t n_total,
                                            int32_t *d_out_idx, float *d_out_dist, int32_t *d_out_count,
                                            int32_t *d_out_flags, void *stream) {
  return guarded([&] {
    GULON_REQUIRE(b >= 0 && d >= 1 && k_nn >= 0 && n_total >= 0, "bad shape");
    if (b == 0 || k_nn == 0) return;
    GULON_REQUIRE(d_queries && d_out_idx && d_out_dist, "null argument");
    hipLaunchKernelGGL(nan_query_fix, dim3(std::min(b, 1024)), dim3(64), 0, (hipStream_t)stream, d_queries, b, d, k_nn,
                       n_total, d_out_idx, d_out_dist, d_out_count, d_out_flags);
    HIP_CHECK(hipGetLastError());
  });
}
