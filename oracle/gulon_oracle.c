/*
 * gulon_oracle.c -- CPU restatement of tixxit/gulon's ANN hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under gulon_amd/ may import, link or
 * call this file; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and only as the checker / the timed CPU baseline.
 *
 * PARITY UNPINNED vs the JVM: the reference is pure Scala, no JVM/scalac/sbt
 * exists in this image, and the reference's tests hold no golden vectors or
 * fixtures (they are 100 % ScalaCheck properties).  This file is therefore a
 * loop-for-loop, rounding-for-rounding restatement written from the Scala
 * source; it is pinned only by (i) the JDK-specified java.util.Random known
 * answers, (ii) CoderSpec's packed-length known answers, (iii) the reference's
 * own properties ported to tests/, and (iv) an independent numpy.float32
 * restatement (oracle/py_oracle.py) that must agree bit for bit.
 *
 * Arithmetic contract: IEEE binary32 everywhere, no FMA contraction, strict
 * left-to-right evaluation -- build with  gcc -O2 -ffp-contract=off  and
 * never -ffast-math.  Paths below are relative to
 * /root/reference/core/src/main/scala/net/tixxit/gulon/.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

#define GO_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------ */
/* java.util.Random (JDK spec; used via scala.util.Random at            */
/* KMeans.scala:28,71,189 and Tests.scala:82)                           */
/* ------------------------------------------------------------------ */
typedef struct { uint64_t seed; } go_jrandom;

#define JR_MULT 0x5DEECE66DULL
#define JR_ADD  0xBULL
#define JR_MASK ((1ULL << 48) - 1)

GO_API void go_jr_init(go_jrandom *r, int64_t seed) {
  r->seed = ((uint64_t)seed ^ JR_MULT) & JR_MASK;
}
static inline int32_t jr_next(go_jrandom *r, int bits) {
  r->seed = (r->seed * JR_MULT + JR_ADD) & JR_MASK;
  /* (int)(seed >>> (48 - bits)): truncating cast of the 48-bit value */
  return (int32_t)(uint32_t)(r->seed >> (48 - bits));
}
GO_API int32_t go_jr_next_int(go_jrandom *r) { return jr_next(r, 32); }
GO_API int32_t go_jr_next_int_bound(go_jrandom *r, int32_t bound) {
  int32_t rr = jr_next(r, 31);
  int32_t m = bound - 1;
  if ((bound & m) == 0) {
    rr = (int32_t)(((int64_t)bound * (int64_t)rr) >> 31);
  } else {
    int32_t u = rr;
    for (;;) {
      rr = u % bound;
      /* Java int arithmetic wraps: u - rr + m < 0 on overflow */
      int32_t t = (int32_t)((uint32_t)u - (uint32_t)rr + (uint32_t)m);
      if (t >= 0) break;
      u = jr_next(r, 31);
    }
  }
  return rr;
}
GO_API int32_t go_jr_next_boolean(go_jrandom *r) { return jr_next(r, 1) != 0; }
GO_API float go_jr_next_float(go_jrandom *r) {
  return (float)jr_next(r, 24) / (float)(1 << 24);
}

/* ------------------------------------------------------------------ */
/* Vectors.subvectors  (Vectors.scala:84-104)                           */
/* ------------------------------------------------------------------ */
GO_API void go_subvectors(int32_t d, int32_t m, int32_t *from, int32_t *until) {
  int32_t ideal = (d + m - 1) / m;
  int32_t shortfall = ideal * m - d;
  int32_t full = m - shortfall;
  for (int32_t i = 0; i < m; i++) {
    if (i < full) {
      from[i] = i * ideal;
      until[i] = from[i] + ideal;
    } else {
      from[i] = full * ideal + (i - full) * (ideal - 1);
      until[i] = from[i] + ideal - 1;
    }
  }
}

/* ------------------------------------------------------------------ */
/* MathUtils.distanceSq  (MathUtils.scala:85-95)                        */
/* ------------------------------------------------------------------ */
GO_API float go_distance_sq(const float *x, const float *y, int32_t len) {
  float sum = 0.0f;
  for (int32_t i = 0; i < len; i++) {
    float dx = y[i] - x[i];
    sum += dx * dx;
  }
  return sum;
}
/* MathUtils.distance(x,y) = math.sqrt(distanceSq).toFloat (MathUtils.scala:97-98) */
static float go_distance(const float *x, const float *y, int32_t len) {
  return (float)sqrt((double)go_distance_sq(x, y, len));
}
/* MathUtils.normalize (MathUtils.scala:100-120) */
GO_API void go_normalize(const float *xs, int32_t len, float *out) {
  float sum = 0.0f;
  for (int32_t i = 0; i < len; i++) { float x = xs[i]; sum += x * x; }
  float dist = (float)sqrt((double)sum);
  for (int32_t i = 0; i < len; i++) out[i] = xs[i] / dist;
}

/* ------------------------------------------------------------------ */
/* SummaryStats builder (MathUtils.scala:43-57): fp32 Welford           */
/* ------------------------------------------------------------------ */
typedef struct { int32_t n; float m; float s; } go_stats;
static void stats_update(go_stats *st, float x) {
  st->n += 1;
  float m0 = st->m;
  st->m = m0 + (x - m0) / (float)st->n;
  st->s = st->s + (x - m0) * (x - st->m);
}

/* ------------------------------------------------------------------ */
/* TopKHeap (TopKHeap.scala:3-94)                                       */
/* ------------------------------------------------------------------ */
typedef struct {
  int32_t cap;
  int32_t size;
  int32_t *keys;
  float *values;
} go_heap;

GO_API go_heap *go_heap_new(int32_t k) {
  go_heap *h = (go_heap *)malloc(sizeof(go_heap));
  h->cap = k; h->size = 0;
  h->keys = (int32_t *)calloc(k > 0 ? k : 1, sizeof(int32_t));
  h->values = (float *)calloc(k > 0 ? k : 1, sizeof(float));
  return h;
}
GO_API void go_heap_free(go_heap *h) { if (h) { free(h->keys); free(h->values); free(h); } }
GO_API int32_t go_heap_size(const go_heap *h) { return h->size; }
GO_API const int32_t *go_heap_keys(const go_heap *h) { return h->keys; }
GO_API const float *go_heap_values(const go_heap *h) { return h->values; }

static void heap_swap(go_heap *h, int32_t i, int32_t j) {
  int32_t tk = h->keys[i]; float tv = h->values[i];
  h->keys[i] = h->keys[j]; h->values[i] = h->values[j];
  h->keys[j] = tk; h->values[j] = tv;
}
static void heap_up(go_heap *h, int32_t i) {          /* TopKHeap.scala:21-28 */
  while (i > 0) {
    int32_t p = (i - 1) / 2;
    if (h->values[i] > h->values[p]) { heap_swap(h, i, p); i = p; } else break;
  }
}
static void heap_down(go_heap *h, int32_t i) {        /* TopKHeap.scala:30-42 */
  for (;;) {
    int32_t top = i, lc = 2 * i + 1, rc = 2 * i + 2;
    if (lc < h->size && h->values[top] < h->values[lc]) top = lc;
    if (rc < h->size && h->values[top] < h->values[rc]) top = rc;
    if (top == i) break;
    heap_swap(h, i, top);
    i = top;
  }
}
/* returns removed key, or INT32_MIN when empty (reference throws) */
GO_API int32_t go_heap_delete(go_heap *h) {            /* TopKHeap.scala:57-67 */
  if (h->size <= 0) return INT32_MIN;
  h->size -= 1;
  int32_t removed = h->keys[0];
  h->keys[0] = h->keys[h->size];
  h->values[0] = h->values[h->size];
  heap_down(h, 0);
  return removed;
}
GO_API void go_heap_update(go_heap *h, int32_t k, float v) {  /* TopKHeap.scala:69-79 */
  if (h->size == h->cap && h->cap > 0 && h->values[0] > v) go_heap_delete(h);
  if (h->size < h->cap) {
    h->keys[h->size] = k;
    h->values[h->size] = v;
    heap_up(h, h->size);
    h->size += 1;
  }
}
GO_API void go_heap_merge(go_heap *h, const go_heap *that) {  /* TopKHeap.scala:44-53 */
  for (int32_t i = 0; i < that->size; i++) go_heap_update(h, that->keys[i], that->values[i]);
}
/* Index.Result.fromHeap (Index.scala:83-94) / TopKHeap.deleteAll (:81-89):
 * drains max-first filling from the back => ascending.  Destroys the heap.
 * Returns the number of entries written. */
GO_API int32_t go_heap_drain(go_heap *h, int32_t *keys_out, float *values_out) {
  int32_t n = h->size;
  for (int32_t i = n - 1; i >= 0; i--) {
    keys_out[i] = h->keys[0];
    if (values_out) values_out[i] = h->values[0];
    go_heap_delete(h);
  }
  return n;
}

/* ------------------------------------------------------------------ */
/* KMeans (KMeans.scala)                                                */
/* Data are flat row-major: row i = X + i*ld, columns [from, from+s).   */
/* ------------------------------------------------------------------ */
/* KMeans.apply: offsets(i) = sum_j c_ij^2  (KMeans.scala:170-186) */
GO_API void go_kmeans_offsets(const float *C, int32_t k, int32_t s, float *off) {
  for (int32_t i = 0; i < k; i++) {
    float acc = 0.0f;
    for (int32_t j = 0; j < s; j++) { float x = C[(size_t)i * s + j]; acc += x * x; }
    off[i] = acc;
  }
}

/* KMeans.init (KMeans.scala:188-196): k draws with replacement */
GO_API void go_kmeans_init(const float *X, int32_t n, int32_t ld, int32_t from, int32_t s,
                           int32_t k, int32_t seed, float *C_out, int32_t *rows_out) {
  go_jrandom rng; go_jr_init(&rng, (int64_t)seed);
  for (int32_t c = 0; c < k; c++) {
    int32_t i = go_jr_next_int_bound(&rng, n);
    if (rows_out) rows_out[c] = i;
    if (C_out) memcpy(C_out + (size_t)c * s, X + (size_t)i * ld + from, sizeof(float) * s);
  }
}

/* private ranged assign (KMeans.scala:24-55): fresh Random(0) per call.
 * The public serial assign (:70-98) is the same loop over [0, n).
 * `assignments` is written only when a candidate wins, exactly like the
 * reference (a NaN distance never wins, leaving the slot untouched). */
GO_API void go_kmeans_assign_range(const float *X, int32_t ld, int32_t from, int32_t s,
                                   const float *C, const float *off, int32_t k,
                                   int32_t start, int32_t end, int32_t *assignments) {
  go_jrandom rng; go_jr_init(&rng, 0);
  for (int32_t i = start; i < end; i++) {
    const float *row = X + (size_t)i * ld + from;
    float min = FLT_MAX;
    for (int32_t c = 0; c < k; c++) {
      const float *cc = C + (size_t)c * s;
      float d = 0.0f;
      for (int32_t j = 0; j < s; j++) d += row[j] * cc[j];
      d = off[c] - 2 * d;
      if (d < min || (d == min && go_jr_next_boolean(&rng))) {
        assignments[i] = c;
        min = d;
      }
    }
  }
}
/* KMeans.parAssign (KMeans.scala:57-68): 25 000-row batches, RNG restarts per batch.
 * rng_batch <= 0 means one stream over all rows (= serial assign). */
GO_API void go_kmeans_assign(const float *X, int32_t n, int32_t ld, int32_t from, int32_t s,
                             const float *C, int32_t k, int32_t rng_batch, int32_t *assignments) {
  float *off = (float *)malloc(sizeof(float) * (k > 0 ? k : 1));
  go_kmeans_offsets(C, k, s, off);
  if (rng_batch <= 0) {
    go_kmeans_assign_range(X, ld, from, s, C, off, k, 0, n, assignments);
  } else {
    for (int32_t b = 0; b < n; b += rng_batch) {
      int32_t e = b + rng_batch < n ? b + rng_batch : n;
      go_kmeans_assign_range(X, ld, from, s, C, off, k, b, e, assignments);
    }
  }
  free(off);
}

/* KMeans.fromAssignment (KMeans.scala:198-226): running mean in row order */
GO_API void go_kmeans_from_assignment(const float *X, int32_t n, int32_t ld, int32_t from, int32_t s,
                                      int32_t k, const int32_t *assignments, float *C_out) {
  int32_t *counts = (int32_t *)calloc(k > 0 ? k : 1, sizeof(int32_t));
  memset(C_out, 0, sizeof(float) * (size_t)k * s);
  for (int32_t i = 0; i < n; i++) {
    const float *v = X + (size_t)i * ld + from;
    int32_t a = assignments[i];
    float *c = C_out + (size_t)a * s;
    int32_t cnt = counts[a] + 1;
    for (int32_t j = 0; j < s; j++) {
      float p = c[j];
      c[j] = p + ((v[j] - p) / (float)cnt);
    }
    counts[a] = cnt;
  }
  free(counts);
}

/* KMeans.iterate (KMeans.scala:100-106): serial assign + fromAssignment, one
 * zero-initialised assignments array reused across iterations. */
GO_API void go_kmeans_iterate(const float *X, int32_t n, int32_t ld, int32_t from, int32_t s,
                              const float *C_in, int32_t k, int32_t iters, float *C_out) {
  int32_t *assign = (int32_t *)calloc(n > 0 ? n : 1, sizeof(int32_t));
  float *cur = (float *)malloc(sizeof(float) * (size_t)k * s + 4);
  memcpy(cur, C_in, sizeof(float) * (size_t)k * s);
  for (int32_t it = 0; it < iters; it++) {
    go_kmeans_assign(X, n, ld, from, s, cur, k, 0, assign);
    go_kmeans_from_assignment(X, n, ld, from, s, k, assign, cur);
  }
  memcpy(C_out, cur, sizeof(float) * (size_t)k * s);
  free(cur); free(assign);
}

/* One KMeans.ProgressReport (KMeans.scala:119-127) as plain numbers. */
typedef struct {
  int32_t num_iterations;
  int32_t converged;
  int32_t step_count;
  float step_mean;
  float step_s;
} go_kmeans_report;

/* KMeans.computeClusters (KMeans.scala:134-157).  Returns the number of
 * reports written (first report is the init one).  C_out = `next` of the last
 * executed iteration. */
GO_API int32_t go_kmeans_compute_clusters(const float *X, int32_t n, int32_t ld, int32_t from,
                                          int32_t s, int32_t k, int32_t max_iterations,
                                          int32_t seed, float *C_out,
                                          go_kmeans_report *reports, int32_t max_reports) {
  size_t csz = sizeof(float) * (size_t)k * s + 4;
  float *prev = (float *)malloc(csz), *next = (float *)malloc(csz);
  int32_t *pa = (int32_t *)calloc(n > 0 ? n : 1, sizeof(int32_t));
  int32_t *na = (int32_t *)calloc(n > 0 ? n : 1, sizeof(int32_t));
  int32_t nrep = 0;
  go_kmeans_init(X, n, ld, from, s, k, seed, prev, NULL);
  go_kmeans_assign(X, n, ld, from, s, prev, k, 25000, pa);
  if (reports && nrep < max_reports) {
    go_kmeans_report r = {0, 0, 0, 0.0f, 0.0f};
    reports[nrep] = r;
  }
  nrep++;
  int32_t i = 0;
  while (i <= max_iterations) {
    go_kmeans_from_assignment(X, n, ld, from, s, k, pa, next);
    /* a fresh Array[Int] per parAssign: zero-filled */
    memset(na, 0, sizeof(int32_t) * (size_t)n);
    go_kmeans_assign(X, n, ld, from, s, next, k, 25000, na);
    int32_t converged = memcmp(pa, na, sizeof(int32_t) * (size_t)n) == 0;
    /* stepSize (KMeans.scala:160-168) */
    go_stats st = {0, 0.0f, 0.0f};
    for (int32_t c = 0; c < k; c++)
      stats_update(&st, go_distance(prev + (size_t)c * s, next + (size_t)c * s, s));
    if (reports && nrep < max_reports) {
      go_kmeans_report r = {i, converged, st.n, st.m, st.s};
      reports[nrep] = r;
    }
    nrep++;
    i = converged ? max_iterations + 1 : i + 1;
    float *tf = prev; prev = next; next = tf;
    int32_t *ti = pa; pa = na; na = ti;
  }
  memcpy(C_out, prev, sizeof(float) * (size_t)k * s);
  free(prev); free(next); free(pa); free(na);
  return nrep;
}

/* ------------------------------------------------------------------ */
/* Coder (Coder.scala)                                                  */
/* ------------------------------------------------------------------ */
/* ProductQuantizer.coderFactory width rule (ProductQuantizer.scala:11-16)
 * followed by Coder.factoryFor rounding (Coder.scala:35-45).  -1 = none. */
GO_API int32_t go_coder_width_for_clusters(int32_t num_clusters) {
  uint32_t x = (uint32_t)(num_clusters - 1);
  int32_t nlz = x == 0 ? 32 : __builtin_clz(x);
  int32_t w = 32 - nlz;
  if (w < 0) return -1;
  if (w == 0) return 0;
  if (w <= 2) return 2;
  if (w <= 4) return 4;
  if (w <= 8) return 8;
  if (w <= 10) return 10;
  if (w <= 12) return 12;
  if (w <= 16) return 16;
  return -1;
}
/* Coder.factoryFor applied to an explicit width (Coder.apply, Coder.scala:54-58) */
GO_API int32_t go_coder_round_width(int32_t w) {
  if (w < 0) return -1;
  if (w == 0) return 0;
  if (w <= 2) return 2;
  if (w <= 4) return 4;
  if (w <= 8) return 8;
  if (w <= 10) return 10;
  if (w <= 12) return 12;
  if (w <= 16) return 16;
  return -1;
}
static int32_t packed_bytes(int32_t width, int32_t length) {   /* Coder.scala:82-83 */
  int32_t per = 8 / width;
  return (length + per - 1) / per;
}
GO_API int32_t go_coder_bytes(int32_t width, int32_t length) {
  switch (width) {
    case 0: return 0;
    case 2: case 4: case 8: return packed_bytes(width, length);
    case 10: return length + packed_bytes(2, length);   /* BytePlus, Coder.scala:153 */
    case 12: return length + packed_bytes(4, length);
    case 16: return length + packed_bytes(8, length);
    default: return -1;
  }
}
static void packed_build(int32_t width, uint8_t *code, const int32_t *idx, int32_t n, int32_t offset) {
  for (int32_t i = 0; i < n; i++) {
    if (width == 2) {                                   /* Coder.scala:100-108 */
      int32_t id = idx[i] & 0x3; int32_t j = offset + (i >> 2);
      code[j] = (uint8_t)(code[j] | (id << ((i & 0x3) * 2)));
    } else if (width == 4) {                            /* Coder.scala:115-123 */
      int32_t id = idx[i] & 0xF; int32_t j = offset + (i >> 1);
      code[j] = (uint8_t)(code[j] | (id << ((i & 0x1) * 4)));
    } else {                                            /* Coder.scala:130-136 */
      code[offset + i] = (uint8_t)idx[i];
    }
  }
}
static int32_t packed_get(int32_t width, const uint8_t *b, int32_t offset, int32_t i) {
  /* bytes are signed on the JVM; `>>>` after int promotion of a negative byte
   * followed by the mask gives the same bits as the unsigned view. */
  if (width == 2) return (b[offset + (i >> 2)] >> ((i & 0x3) * 2)) & 0x3;   /* :110-111 */
  if (width == 4) return (b[offset + (i >> 1)] >> ((i & 0x1) * 4)) & 0xF;   /* :125-126 */
  return b[offset + i] & 0xFF;                                              /* :138-139 */
}
GO_API int32_t go_coder_build(int32_t width, const int32_t *idx, int32_t n, uint8_t *code) {
  int32_t nb = go_coder_bytes(width, n);
  if (nb < 0) return -1;
  memset(code, 0, (size_t)nb);
  if (width == 0) return 0;
  if (width <= 8) { packed_build(width, code, idx, n, 0); return nb; }
  int32_t lw = width - 8;                               /* BytePlus.buildCode :147-161 */
  for (int32_t i = 0; i < n; i++) code[i] = (uint8_t)((uint32_t)idx[i] >> lw);
  packed_build(lw, code, idx, n, n);
  return nb;
}
GO_API int32_t go_coder_get(int32_t width, const uint8_t *code, int32_t n, int32_t i) {
  if (width == 0) return 0;
  if (width <= 8) return packed_get(width, code, 0, i);
  int32_t lw = width - 8;                               /* BytePlus.getIndex :163-167 */
  int32_t b1 = (code[i] & 0xFF) << lw;
  int32_t b0 = packed_get(lw, code, n, i) & 0xFF;
  return b1 | b0;
}

/* ------------------------------------------------------------------ */
/* ProductQuantizer (ProductQuantizer.scala)                            */
/* Codebooks are one flat array of k*d floats: quantizer j's k x s_j     */
/* block starts at cents + k*from_j.                                    */
/* ------------------------------------------------------------------ */
/* ProductQuantizer.apply/fromSubvectors (:121-153): m independent
 * computeClusters, seed = quantizer index.  iters_out[j]/converged_out[j]
 * describe the last report of quantizer j. */
GO_API void go_pq_train(const float *X, int32_t n, int32_t d, int32_t m, int32_t k,
                        int32_t max_iterations, float *cents,
                        int32_t *iters_out, int32_t *converged_out) {
  int32_t *from = (int32_t *)malloc(sizeof(int32_t) * m), *until = (int32_t *)malloc(sizeof(int32_t) * m);
  go_subvectors(d, m, from, until);
  int32_t maxrep = max_iterations + 3;
  go_kmeans_report *rep = (go_kmeans_report *)malloc(sizeof(go_kmeans_report) * maxrep);
  for (int32_t j = 0; j < m; j++) {
    int32_t s = until[j] - from[j];
    int32_t nrep = go_kmeans_compute_clusters(X, n, d, from[j], s, k, max_iterations, j,
                                              cents + (size_t)k * from[j], rep, maxrep);
    if (iters_out) iters_out[j] = rep[nrep - 1].num_iterations;
    if (converged_out) converged_out[j] = rep[nrep - 1].converged;
  }
  free(rep); free(from); free(until);
}

/* ProductQuantizer.encode (:25-35): per quantizer the SERIAL assign (one RNG
 * stream over all rows).  idx_out is [m][n] centroid indices. */
GO_API void go_pq_encode(const float *X, int32_t n, int32_t d, int32_t m, int32_t k,
                         const float *cents, int32_t *idx_out) {
  int32_t *from = (int32_t *)malloc(sizeof(int32_t) * m), *until = (int32_t *)malloc(sizeof(int32_t) * m);
  go_subvectors(d, m, from, until);
  memset(idx_out, 0, sizeof(int32_t) * (size_t)m * n);
  for (int32_t j = 0; j < m; j++) {
    int32_t s = until[j] - from[j];
    go_kmeans_assign(X, n, d, from[j], s, cents + (size_t)k * from[j], k, 0, idx_out + (size_t)j * n);
  }
  free(from); free(until);
}

/* ProductQuantizer.decode (:37-78) */
GO_API void go_pq_decode(const int32_t *idx, int32_t n, int32_t d, int32_t m, int32_t k,
                         const float *cents, float *X_out) {
  int32_t *from = (int32_t *)malloc(sizeof(int32_t) * m), *until = (int32_t *)malloc(sizeof(int32_t) * m);
  go_subvectors(d, m, from, until);
  for (int32_t j = 0; j < m; j++) {
    int32_t s = until[j] - from[j];
    const float *cb = cents + (size_t)k * from[j];
    for (int32_t i = 0; i < n; i++)
      memcpy(X_out + (size_t)i * d + from[j], cb + (size_t)idx[(size_t)j * n + i] * s, sizeof(float) * s);
  }
  free(from); free(until);
}

/* ------------------------------------------------------------------ */
/* Index (Index.scala)                                                  */
/* ------------------------------------------------------------------ */
/* Index.prepareQuery (:352-383): T[q][j][c], direct diff-square-add form */
GO_API void go_prepare_query(const float *cents, int32_t d, int32_t m, int32_t k,
                             const float *Q, int32_t B, float *T) {
  int32_t *from = (int32_t *)malloc(sizeof(int32_t) * m), *until = (int32_t *)malloc(sizeof(int32_t) * m);
  go_subvectors(d, m, from, until);
  for (int32_t j = 0; j < m; j++) {
    int32_t s = until[j] - from[j];
    const float *cb = cents + (size_t)k * from[j];
    for (int32_t c = 0; c < k; c++) {
      const float *cc = cb + (size_t)c * s;
      for (int32_t q = 0; q < B; q++) {
        const float *query = Q + (size_t)q * d + from[j];
        float sum = 0.0f;
        for (int32_t t = 0; t < s; t++) {
          float dd = query[t] - cc[t];
          sum += dd * dd;
        }
        T[((size_t)q * m + j) * k + c] = sum;
      }
    }
  }
  free(from); free(until);
}

/* PQIndex.batchQuery (:417-440) with PQIndex.distances (:393-409).
 * idx is [m][n] unpacked centroid indices (what coder.getIndex returns).
 * Returns -1 when the reference's `require`s fail.  Results are drained as
 * Index.Result.fromHeap does: ascending, out_count[q] live entries per query,
 * out arrays are [B][K]. */
GO_API int32_t go_pq_batch_query(const int32_t *idx, int32_t n, int32_t d, int32_t m, int32_t k,
                                 const float *cents, const float *Q, int32_t B, int32_t K,
                                 int32_t from_row, int32_t until_row,
                                 int32_t *out_idx, float *out_dist, int32_t *out_count) {
  if (!(from_row <= until_row)) return -1;
  if (!(from_row >= 0 && until_row <= n)) return -1;
  float *T = (float *)malloc(sizeof(float) * (size_t)(B > 0 ? B : 1) * m * k + 4);
  go_prepare_query(cents, d, m, k, Q, B, T);
  go_heap **heaps = (go_heap **)malloc(sizeof(go_heap *) * (B > 0 ? B : 1));
  for (int32_t q = 0; q < B; q++) heaps[q] = go_heap_new(K);
  float *ds = (float *)malloc(sizeof(float) * 4096);
  for (int32_t i = from_row; i < until_row;) {
    int32_t bs = until_row - i < 4096 ? until_row - i : 4096;
    for (int32_t q = 0; q < B; q++) {
      memset(ds, 0, sizeof(float) * bs);
      for (int32_t j = 0; j < m; j++) {
        const float *qds = T + ((size_t)q * m + j) * k;
        const int32_t *code = idx + (size_t)j * n + i;
        for (int32_t r = 0; r < bs; r++) ds[r] += qds[code[r]];
      }
      go_heap *h = heaps[q];
      for (int32_t r = 0; r < bs; r++) go_heap_update(h, i + r, ds[r]);
    }
    i += bs;
  }
  for (int32_t q = 0; q < B; q++) {
    int32_t cnt = go_heap_drain(heaps[q], out_idx + (size_t)q * K, out_dist + (size_t)q * K);
    if (out_count) out_count[q] = cnt;
    go_heap_free(heaps[q]);
  }
  free(ds); free(heaps); free(T);
  return 0;
}

/* Same scan with uint8 codes [m][n] (Coder8, k<=256): the form the CPU
 * baseline times -- identical arithmetic and blocking to the function above. */
GO_API int32_t go_pq_batch_query_u8(const uint8_t *codes, int32_t n, int32_t d, int32_t m, int32_t k,
                                    const float *cents, const float *Q, int32_t B, int32_t K,
                                    int32_t from_row, int32_t until_row,
                                    int32_t *out_idx, float *out_dist, int32_t *out_count) {
  if (!(from_row <= until_row)) return -1;
  if (!(from_row >= 0 && until_row <= n)) return -1;
  float *T = (float *)malloc(sizeof(float) * (size_t)(B > 0 ? B : 1) * m * k + 4);
  go_prepare_query(cents, d, m, k, Q, B, T);
  go_heap **heaps = (go_heap **)malloc(sizeof(go_heap *) * (B > 0 ? B : 1));
  for (int32_t q = 0; q < B; q++) heaps[q] = go_heap_new(K);
  float *ds = (float *)malloc(sizeof(float) * 4096);
  for (int32_t i = from_row; i < until_row;) {
    int32_t bs = until_row - i < 4096 ? until_row - i : 4096;
    for (int32_t q = 0; q < B; q++) {
      memset(ds, 0, sizeof(float) * bs);
      for (int32_t j = 0; j < m; j++) {
        const float *qds = T + ((size_t)q * m + j) * k;
        const uint8_t *code = codes + (size_t)j * n + i;
        for (int32_t r = 0; r < bs; r++) ds[r] += qds[code[r] & 0xFF];
      }
      go_heap *h = heaps[q];
      for (int32_t r = 0; r < bs; r++) go_heap_update(h, i + r, ds[r]);
    }
    i += bs;
  }
  for (int32_t q = 0; q < B; q++) {
    int32_t cnt = go_heap_drain(heaps[q], out_idx + (size_t)q * K, out_dist + (size_t)q * K);
    if (out_count) out_count[q] = cnt;
    go_heap_free(heaps[q]);
  }
  free(ds); free(heaps); free(T);
  return 0;
}

/* Index.exactNearestNeighbours (:209-229) + Result.fromHeap for B queries.
 * X is n x d row-major (ld = d). */
GO_API int32_t go_exact_knn(const float *X, int32_t n, int32_t d, int32_t from_row, int32_t until_row,
                            const float *Q, int32_t B, int32_t K,
                            int32_t *out_idx, float *out_dist, int32_t *out_count) {
  if (!(from_row <= until_row)) return -1;
  if (!(until_row <= n)) return -1;
  for (int32_t q = 0; q < B; q++) {
    go_heap *h = go_heap_new(K);
    const float *query = Q + (size_t)q * d;
    for (int32_t i = from_row; i < until_row; i++)
      go_heap_update(h, i, go_distance_sq(X + (size_t)i * d, query, d));
    int32_t cnt = go_heap_drain(h, out_idx + (size_t)q * K, out_dist + (size_t)q * K);
    if (out_count) out_count[q] = cnt;
    go_heap_free(h);
  }
  return 0;
}

/* GroupedIndex.query (Index.scala:265-282) + searchSpace (:285-299) + getBounds (:259-263)
 * for B queries (batchQuery = one query at a time, :254-257).
 *   idx      residual PQ codes [m][n], rows in grouped order (WordVectors.grouped, :24-58)
 *   gcent    the g non-empty coarse centroids [g][d], offsets[g-1] = first row of groups 1..g-1
 *   strategy 0 = LimitGroups(limit), 1 = LimitVectors(limit)
 * Every searched group gets its own TopKHeap from PQIndex.query on the residual
 * (query - centroid, MathUtils.subtract :74-83); the heaps are folded into the result with
 * TopKHeap.merge (:44-53), i.e. update() of the other heap's slots in ARRAY order. */
GO_API int32_t go_grouped_query(const int32_t *idx, int32_t n, int32_t d, int32_t m, int32_t k,
                                const float *pq_cents, const float *gcent, const int32_t *offsets, int32_t g,
                                const float *Q, int32_t B, int32_t K, int32_t strategy, int32_t limit,
                                int32_t *out_idx, float *out_dist, int32_t *out_count) {
  if (g < 1 || limit < 0) return -1;
  float *T = (float *)malloc(sizeof(float) * (size_t)m * k + 4);
  float *ds = (float *)malloc(sizeof(float) * 4096);
  float *res = (float *)malloc(sizeof(float) * (size_t)d);
  int32_t *order = (int32_t *)malloc(sizeof(int32_t) * (size_t)g);
  float *odist = (float *)malloc(sizeof(float) * (size_t)g);
  for (int32_t q = 0; q < B; q++) {
    const float *query = Q + (size_t)q * d;
    /* searchSpace */
    int32_t hk = strategy == 0 ? limit : g;
    go_heap *hc = go_heap_new(hk);
    for (int32_t i = 0; i < g; i++) go_heap_update(hc, i, go_distance_sq(gcent + (size_t)i * d, query, d));
    int32_t nn = go_heap_drain(hc, order, odist);          /* deleteAll(): ascending by distance */
    go_heap_free(hc);
    if (strategy == 1) {
      int32_t i = 0, count = 0;
      while (i < nn && count < limit) {
        int32_t c = order[i];
        int32_t start = c == 0 ? 0 : offsets[c - 1];
        int32_t end = c == g - 1 ? n : offsets[c];
        count += end - start;
        i++;
      }
      nn = i;
    }
    go_heap *heap = go_heap_new(K);
    for (int32_t t = 0; t < nn; t++) {
      int32_t c = order[t];
      int32_t from_row = c == 0 ? 0 : offsets[c - 1];
      int32_t until_row = c == g - 1 ? n : offsets[c];
      for (int32_t e = 0; e < d; e++) res[e] = query[e] - gcent[(size_t)c * d + e];
      go_prepare_query(pq_cents, d, m, k, res, 1, T);
      go_heap *hg = go_heap_new(K);
      for (int32_t i = from_row; i < until_row;) {
        int32_t bs = until_row - i < 4096 ? until_row - i : 4096;
        memset(ds, 0, sizeof(float) * bs);
        for (int32_t j = 0; j < m; j++) {
          const float *qds = T + (size_t)j * k;
          const int32_t *code = idx + (size_t)j * n + i;
          for (int32_t r = 0; r < bs; r++) ds[r] += qds[code[r]];
        }
        for (int32_t r = 0; r < bs; r++) go_heap_update(hg, i + r, ds[r]);
        i += bs;
      }
      go_heap_merge(heap, hg);
      go_heap_free(hg);
    }
    int32_t cnt = go_heap_drain(heap, out_idx + (size_t)q * K, out_dist + (size_t)q * K);
    if (out_count) out_count[q] = cnt;
    go_heap_free(heap);
  }
  free(odist); free(order); free(res); free(ds); free(T);
  return 0;
}

/* Tests.recallOf (Tests.scala:18-41) for ONE k (eps = 0): fraction of the
 * first K returned rows whose exact distanceSq(query, X[row]) <= cutoff,
 * cutoff = exact K-th neighbour distance.  Returns mean recall; *sd_out the
 * SummaryStats stdDev (sqrt(s/count)). */
GO_API float go_recall(const float *X, int32_t d, const float *Q, int32_t B, int32_t K,
                       const int32_t *ann_idx, const int32_t *ann_count,
                       const float *exact_dist, const int32_t *exact_count, float *sd_out) {
  /* Monoid.combineAll over SummaryStats(tp/k) == pairwise ++; the mean of
   * the merged stats is order dependent only in rounding; we follow the
   * left fold ((s0 ++ s1) ++ s2) ... that combineAll performs. */
  int32_t cnt = 0; float mean = 0.0f, ss = 0.0f;
  for (int32_t q = 0; q < B; q++) {
    if (exact_count[q] < K) continue;                 /* ks.filter(_ <= result.length) */
    float cutoff = exact_dist[(size_t)q * K + (K - 1)];
    int32_t tp = 0;
    int32_t lim = ann_count[q] < K ? ann_count[q] : K;
    for (int32_t i = 0; i < lim; i++) {
      float dd = go_distance_sq(Q + (size_t)q * d, X + (size_t)ann_idx[(size_t)q * K + i] * d, d);
      if (dd <= cutoff) tp++;
    }
    float x = (float)tp / (float)K;
    /* SummaryStats ++ (MathUtils.scala:11-22) with that = SummaryStats(1, x, 0) */
    if (cnt == 0) { cnt = 1; mean = x; ss = 0.0f; }
    else {
      int32_t nn = cnt + 1;
      float dlt = mean - x;
      float nm = mean + ((float)1 / (float)nn) * (x - mean);
      float ns = ss + 0.0f + dlt * dlt * (float)cnt * (float)1 / (float)nn;
      cnt = nn; mean = nm; ss = ns;
    }
  }
  if (sd_out) *sd_out = cnt > 0 ? (float)sqrt((double)(ss / (float)cnt)) : 0.0f;
  return mean;
}

/* ------------------------------------------------------------------ */
/* Synthetic data (NOT part of the reference): counter-based integer     */
/* generator shared bit-for-bit with the HIP generator (synth.hip).      */
/* ------------------------------------------------------------------ */
static inline uint64_t go_mix64(uint64_t z) {             /* splitmix64 finaliser */
  z += 0x9E3779B97F4A7C15ULL;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}
static inline float go_u24(uint64_t h) { return (float)(uint32_t)(h >> 40) * (1.0f / 16777216.0f); }
GO_API float go_synth_uniform(uint64_t seed, uint64_t stream, uint64_t idx) {
  return go_u24(go_mix64(go_mix64(seed ^ (stream * 0xD1B54A32D192ED03ULL)) + idx));
}
/* Irwin-Hall(12) - 6: libm-free approximate N(0,1); 12 24-bit uniforms from 6 hashes */
GO_API float go_synth_gauss(uint64_t seed, uint64_t stream, uint64_t idx) {
  uint64_t base = go_mix64(seed ^ (stream * 0xD1B54A32D192ED03ULL));
  float acc = 0.0f;
  for (int t = 0; t < 6; t++) {
    uint64_t h = go_mix64(base + idx * 6 + (uint64_t)t);
    acc += (float)(uint32_t)(h >> 40) * (1.0f / 16777216.0f);
    acc += (float)(uint32_t)((h >> 16) & 0xFFFFFF) * (1.0f / 16777216.0f);
  }
  return acc - 6.0f;
}
/* kind 0: iid N(0,1); kind 1: clustered (ncentres centres U(-5,5), per-dim
 * scale U(0.1,1), row = centre + g*scale -- the shape of the reference's test
 * generator, Generators.scala:18-60); kind 2: U[0,1); kind 3: overlapping
 * clusters (centres U(-2,2), scale U(0.5,1.5)): PQ codes stay (almost) unique */
GO_API void go_synth_fill(float *X, int64_t row0, int64_t nrows, int32_t d, int32_t kind,
                          uint64_t seed, int32_t ncentres) {
  for (int64_t r = 0; r < nrows; r++) {
    uint64_t row = (uint64_t)(row0 + r);
    for (int32_t c = 0; c < d; c++) {
      uint64_t idx = row * (uint64_t)d + (uint64_t)c;
      float v;
      if (kind == 0) v = go_synth_gauss(seed, 1, idx);
      else if (kind == 2) v = go_synth_uniform(seed, 1, idx);
      else {
        uint64_t ce = go_mix64(go_mix64(seed ^ 0x5851F42D4C957F2DULL) + row) % (uint64_t)ncentres;
        uint64_t cidx = ce * (uint64_t)d + (uint64_t)c;
        float centre, scale;
        if (kind == 1) {
          centre = go_synth_uniform(seed, 2, cidx) * 10.0f - 5.0f;
          scale = go_synth_uniform(seed, 3, cidx) * 0.9f + 0.1f;
        } else {
          centre = go_synth_uniform(seed, 2, cidx) * 4.0f - 2.0f;
          scale = go_synth_uniform(seed, 3, cidx) + 0.5f;
        }
        float g = go_synth_gauss(seed, 1, idx);
        v = centre + g * scale;
      }
      X[(size_t)r * d + c] = v;
    }
  }
}
