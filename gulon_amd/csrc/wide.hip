// PQ indexes with more than 256 centroids per quantizer: the 10-, 12- and 16-bit codes of
// Coder.BytePlus (Coder.scala:99-127,142-168; ProductQuantizer.coderFactory,
// ProductQuantizer.scala:11-16, picks them for 256 < numClusters <= 65 536).
//
// The byte-coded kernels keep a query's m x 256 table (or its 8-bit image) in LDS; with k entries per
// quantizer a table is m * k * 4 B -- 64 KiB at m = 16, k = 1024, 4 MiB at k = 65 536.  This path keeps
// the reference's arithmetic (Index.scala:352-383 table, Index.scala:424-437 j-ordered unfused fp32
// sum, (distance, row) order of the lists) and trades speed for generality:
//
//   codes    uint16 per (row, quantizer), row-blocked [n / 64][m][64]: a wave reads one quantizer's
//            64 codes with one coalesced 128-byte load;
//   tables   fp32 [query][m][k] in HBM (built per sub-batch of queries, <= 1 GiB at a time);
//   scan     one workgroup (8 waves) per (query, chunk of row blocks): the query's table in LDS when
//            m * k * 4 B fits 128 KiB; larger tables in slices of as many quantizers as fit, one launch per
//            slice with the running sums parked in HBM in between; k > 32 768 (one quantizer = 256 KiB)
//            gathered from L2/HBM; lane = row; every wave keeps its own sorted top-(K+1) list in
//            registers (WaveList), written out as one partial list per wave;
//   merge    the common merge_lists of scan.hip; queries flagged with exact distance ties are then replayed with
//            the literal TopKHeap like those of a byte-coded index (replay.hip: rp_scan_wide gathers the flagged
//            query's table from global memory).
#include "scan.hpp"

namespace gulon {

namespace {

constexpr int WIDE_THREADS = 512;
constexpr int WIDE_NW = WIDE_THREADS / 64;
constexpr size_t WIDE_LDS_TABLE = 128 * 1024;
constexpr size_t WIDE_TABLE_BYTES = 1ull << 30;   // tables of one sub-batch of queries

__global__ void relayout_wide(const uint16_t *__restrict__ src /*[m][n]*/, int n, int m,
                              uint16_t *__restrict__ dst /*[n/64][m][64]*/, long long total) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const int lane = (int)(t & 63);
  const long long bj = t >> 6;
  const int j = (int)(bj % m);
  const long long rb = bj / m;
  const long long row = rb * 64 + lane;
  dst[t] = row < n ? src[(size_t)j * n + row] : (uint16_t)0;
}

// Index.prepareQuery (Index.scala:352-383) for any k: T[q][j][c] = sum_t (q[from_j + t] - c_j[c][t])^2,
// t ascending, unfused.  Workgroup = 256 centroids of one quantizer x TQ queries: a centroid coordinate is read once
// for all of them, the query values are block-uniform scalar loads (one workgroup per query: 65 536 workgroups of a
// few hundred cycles each for a 1024-query batch at k = 1024 -- 0.45 ms; eight queries each: 0.1 ms).
constexpr int WIDE_TQ = 8;
__global__ __launch_bounds__(256) void build_tables_wide(const float *__restrict__ cents, const int *__restrict__ from,
                                                         const int *__restrict__ sdim, int d, int m, int k,
                                                         const float *__restrict__ Q, int q0, int nq,
                                                         float *__restrict__ T /*[queries of the sub-batch][m][k]*/) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  const int j = blockIdx.y;
  const int ql0 = blockIdx.z * WIDE_TQ;
  if (c >= k) return;
  const int fr = from[j], s = sdim[j];
  const float *cc = cents + (size_t)k * fr + (size_t)c * s;
  float acc[WIDE_TQ];
#pragma unroll
  for (int u = 0; u < WIDE_TQ; u++) acc[u] = 0.f;
  for (int t = 0; t < s; t++) {
    const float cv = cc[t];
#pragma unroll
    for (int u = 0; u < WIDE_TQ; u++) {
      const int ql = min(ql0 + u, nq - 1);                       // (uniform; clamped: always a valid query)
      const float dd = Q[(size_t)(q0 + ql) * d + fr + t] - cv;
      acc[u] += dd * dd;
    }
  }
#pragma unroll
  for (int u = 0; u < WIDE_TQ; u++)
    if (ql0 + u < nq) T[((size_t)(ql0 + u) * m + j) * k + c] = acc[u];
}

// Quantizers [j0, j1) of the table are used by this launch.  LDS_T: that slice is staged in LDS.
// A table too large for LDS as a whole is walked in slices, one launch per slice: the running sums of
// every (query, row) go through `partial` ([query][rows of the range], HBM) between the launches --
// still the reference's order, j ascending -- and only the last launch selects.
template <bool LDS_T, bool FIRST, bool LAST>
__global__ __launch_bounds__(WIDE_THREADS) void scan_wide(const uint16_t *__restrict__ codes, int m, int k,
                                                          const float *__restrict__ tables, int row_from,
                                                          int row_until, int row_base, int rb_begin, int rb_total,
                                                          int rb_per_chunk, int nchunks, int keff,
                                                          float *__restrict__ part_v, int *__restrict__ part_i,
                                                          int j0, int j1, float *__restrict__ partial,
                                                          const int *__restrict__ enable /* per query; null: all */) {
  extern __shared__ float wide_lds[];
  if (enable && enable[blockIdx.x] == 0) return;   // (wide_filter.hip: only the queries the filter gave up on)
  if (FIRST) j0 = 0;       // (constants for the optimiser: the one-slice instantiation is the plain scan)
  if (LAST) j1 = m;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = blockIdx.x, chunk = blockIdx.y;
  const float *T = tables + (size_t)q * m * k;
  if (LDS_T) {
    const float *src = T + (size_t)j0 * k;
    for (int e = tid; e < (j1 - j0) * k; e += WIDE_THREADS) wide_lds[e] = src[e];
    __syncthreads();
  }
  const float *tab = LDS_T ? wide_lds : T + (size_t)j0 * k;     // tab[(j - j0) * k + c] for j in [j0, j1)
  constexpr bool first = FIRST, last = LAST;      // slice [0, ..) starts the sums, slice [.., m) selects
  float *pq = (first && last) ? nullptr : partial + (size_t)q * rb_total * 64;
  WaveList wl;
  wl.init();
  int cnt = 0;
  const int e0 = chunk * rb_per_chunk, e1 = min(rb_total, e0 + rb_per_chunk);
  for (int e = e0 + wave; e < e1; e += WIDE_NW) {
    const int rb = rb_begin + e;
    const uint16_t *p = codes + (size_t)rb * m * 64 + lane;
    float acc = first ? 0.f : pq[(size_t)e * 64 + lane];   // the reference's order: j ascending, unfused fp32
    int j = j0;
    for (; j + 4 <= j1; j += 4) {
      const int c0 = p[(size_t)(j + 0) * 64], c1 = p[(size_t)(j + 1) * 64];
      const int c2 = p[(size_t)(j + 2) * 64], c3 = p[(size_t)(j + 3) * 64];
      const float t0 = tab[(size_t)(j - j0 + 0) * k + c0], t1 = tab[(size_t)(j - j0 + 1) * k + c1];
      const float t2 = tab[(size_t)(j - j0 + 2) * k + c2], t3 = tab[(size_t)(j - j0 + 3) * k + c3];
      acc += t0; acc += t1; acc += t2; acc += t3;
    }
    for (; j < j1; j++) acc += tab[(size_t)(j - j0) * k + p[(size_t)j * 64]];
    if (!last) {
      pq[(size_t)e * 64 + lane] = acc;
      continue;
    }
    const int row = rb * 64 + lane;
    const bool valid = row >= row_from && row < row_until;
    unsigned long long mk = __ballot(valid && acc <= wl.tau);
    while (mk) {
      const int l = __ffsll((long long)mk) - 1;
      mk &= mk - 1;
      const float cv = readlane_f(acc, l);
      const int cr = rb * 64 + l + row_base;
      if (cnt < keff || wl.accepts(cv, cr)) {
        wl.insert(cv, cr, keff, lane);
        if (cnt < keff) cnt++;
      }
    }
  }
  if (last && lane < keff) {
    const size_t o = (((size_t)q * nchunks + chunk) * WIDE_NW + wave) * keff + lane;
    part_v[o] = wl.v;
    part_i[o] = wl.i;
  }
}

}  // namespace

// codes of a wide index: `wide16` = [m][n] uint16 on the device -> row-blocked layout
void wide_store_codes(gulon_index *ix, const uint16_t *wide16) {
  const size_t nblk = (size_t)ceil_div(ix->n, 64);
  ix->wcodes.alloc(std::max<size_t>(nblk * ix->m * 64, 64));
  if (ix->n <= 0) return;
  const long long total = (long long)nblk * ix->m * 64;
  hipLaunchKernelGGL(relayout_wide, dim3((unsigned)ceil_div(total, 256LL)), dim3(256), 0, 0, wide16, ix->n, ix->m,
                     ix->wcodes.p, total);
  HIP_CHECK(hipGetLastError());
  HIP_CHECK(hipDeviceSynchronize());
}

void launch_build_tables_wide(const float *cents, const int *from, const int *sdim, int d, int m, int k, const float *dQ,
                              int q0, int nq, float *tables, hipStream_t st) {
  constexpr int ZMAX = 32768;                      // grid.z <= 65535
  for (int z0 = 0; z0 < nq; z0 += ZMAX * WIDE_TQ) {
    const int nz = std::min(ZMAX * WIDE_TQ, nq - z0);
    hipLaunchKernelGGL(build_tables_wide, dim3(ceil_div(k, 256), m, ceil_div(nz, WIDE_TQ)), dim3(256), 0, st, cents, from,
                       sdim, d, m, k, dQ, q0 + z0, nz, tables + (size_t)z0 * m * k);
    HIP_CHECK(hipGetLastError());
  }
}

// The one-slice exact scan (table in LDS) of `B` queries over row blocks [rb_begin, rb_begin + rb_total), rows
// [from, until): partial lists [B][nchunks * 8][K + 1] into ix->part_v / part_i (wide_filter.hip: its sample scan and
// its fallback; `enable` selects the queries).  The tables are in ix->tables.
void launch_scan_wide_range(gulon_index *ix, int B, int K, int from, int until, int rb_begin, int rb_total,
                            int rb_per_chunk, int nchunks, const int *enable, hipStream_t st) {
  const int m = ix->m, k = ix->k;
  const size_t table_bytes = (size_t)m * k * sizeof(float);
  auto go = [&](auto kern, size_t lds_bytes, int j0, int j1, float *partial) {
    if (lds_bytes)
      HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)lds_bytes));
    hipLaunchKernelGGL(kern, dim3(B, nchunks), dim3(WIDE_THREADS), lds_bytes, st, ix->wcodes.p, m, k, ix->tables.p,
                       from, until, ix->row_base, rb_begin, rb_total, rb_per_chunk, nchunks, K + 1, ix->part_v.p,
                       ix->part_i.p, j0, j1, partial, enable);
    HIP_CHECK(hipGetLastError());
  };
  if (table_bytes <= WIDE_LDS_TABLE) {
    go(scan_wide<true, true, true>, table_bytes, 0, m, nullptr);
    return;
  }
  // a table beyond LDS (wide_filter.hip's sliced form, k >= 4096 at m = 16): slices of as many quantizers as fit, the
  // running sums of every (query, row) parked in between -- while that buffer stays small (the filter's sample scan);
  // else gathered through L2 (its fallback over the whole range: only the queries `enable` selects do any work)
  const int jp = (int)(WIDE_LDS_TABLE / ((size_t)k * sizeof(float)));
  const size_t rows_pad = (size_t)rb_total * 64;
  if (jp >= 1 && (size_t)B * rows_pad * sizeof(float) <= (256ull << 20)) {
    ix->wpartial.ensure((size_t)B * rows_pad);
    for (int j0 = 0; j0 < m; j0 += jp) {
      const int j1 = std::min(m, j0 + jp);
      const size_t lds_bytes = (size_t)(j1 - j0) * k * sizeof(float);
      if (j0 == 0) go(scan_wide<true, true, false>, lds_bytes, j0, j1, ix->wpartial.p);
      else if (j1 == m) go(scan_wide<true, false, true>, lds_bytes, j0, j1, ix->wpartial.p);
      else go(scan_wide<true, false, false>, lds_bytes, j0, j1, ix->wpartial.p);
    }
    return;
  }
  go(scan_wide<false, true, true>, 0, 0, m, nullptr);
}

// Table build + scan + merge of one batch over rows [from, until) of a wide index (run_query's contract).
void run_wide_query(gulon_index *ix, const float *dQ, int B, int K, int from, int until, bool final_out, int *d_oi,
                    float *d_od, int *d_oc, int *d_of, float *d_pv, int *d_pi, hipStream_t st) {
  GULON_UNSUPPORTED(K > GULON_MAX_K, "k_nn = %d > GULON_MAX_K = %d is not supported for k = %d centroids", K, GULON_MAX_K,
                    ix->k);
  const int keff = K + 1, m = ix->m, k = ix->k;
  const int rb_begin = from / 64, rb_total = ceil_div(until, 64) - rb_begin;
  if (wide_filter_eligible(ix, B, K, rb_total)) {   // 8-bit lower bounds in front of the exact arithmetic (wide_filter.hip)
    run_wide_filter_query(ix, dQ, B, K, from, until, final_out, d_oi, d_od, d_oc, d_of, d_pv, d_pi, st);
    return;
  }
  const size_t table_bytes = (size_t)m * k * sizeof(float);
  // quantizers per launch: the whole table if it fits LDS; else as many as fit, the running sums going through
  // HBM between the launches; a single quantizer's k entries above 128 KiB (k > 32 768): gathered through L2
  const int jp_fit = (int)(WIDE_LDS_TABLE / ((size_t)k * sizeof(float)));
  const bool lds_t = jp_fit >= 1;
  const int jp = lds_t ? std::min(jp_fit, m) : m;
  const int passes = ceil_div(m, jp);
  const size_t rows_pad = (size_t)rb_total * 64;
  size_t qb_cap = WIDE_TABLE_BYTES / table_bytes;
  if (passes > 1) qb_cap = std::min(qb_cap, WIDE_TABLE_BYTES / (rows_pad * sizeof(float)));
  const int qb = (int)std::max<size_t>(1, std::min<size_t>((size_t)B, qb_cap));
  // chunks: ~2048 workgroups per launch, at least 2 row blocks per wave
  int nchunks = std::max(1, std::min(ceil_div(2048, std::min(qb, B)), rb_total / (2 * WIDE_NW)));
  const int rb_per_chunk = ceil_div(rb_total, nchunks);
  nchunks = ceil_div(rb_total, rb_per_chunk);
  const int lists = nchunks * WIDE_NW;
  ix->tables.ensure((size_t)qb * m * k);
  ix->part_v.ensure((size_t)qb * lists * keff);
  ix->part_i.ensure((size_t)qb * lists * keff);
  if (passes > 1) ix->wpartial.ensure((size_t)qb * rows_pad);
  int *flags = d_of;
  if (final_out && replay_enabled() && flags == nullptr) {   // the replay needs the tie flags even if the caller does not
    ix->flags_scratch.ensure((size_t)B);
    flags = ix->flags_scratch.p;
  }
  for (int q0 = 0; q0 < B; q0 += qb) {
    const int nq = std::min(qb, B - q0);
    launch_build_tables_wide(ix->cents.p, ix->from.p, ix->sdim.p, ix->d, m, k, dQ, q0, nq, ix->tables.p, st);
    for (int j0 = 0; j0 < m; j0 += jp) {
      const int j1 = std::min(m, j0 + jp);
      float *partial = passes > 1 ? ix->wpartial.p : nullptr;
      const size_t lds_bytes = lds_t ? (size_t)(j1 - j0) * k * sizeof(float) : 0;
      const bool first = j0 == 0, last = j1 == m;
#define WIDE_GO(L, F, LA)                                                                                          \
      {                                                                                                             \
        auto kern = scan_wide<L, F, LA>;                                                                            \
        if (lds_bytes)                                                                                              \
          HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),                                       \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));               \
        hipLaunchKernelGGL(kern, dim3(nq, nchunks), dim3(WIDE_THREADS), lds_bytes, st, ix->wcodes.p, m, k,           \
                           ix->tables.p, from, until, ix->row_base, rb_begin, rb_total, rb_per_chunk, nchunks, keff, \
                           ix->part_v.p, ix->part_i.p, j0, j1, partial, (const int *)nullptr);                      \
      }
      if (!lds_t) WIDE_GO(false, true, true)
      else if (first && last) WIDE_GO(true, true, true)
      else if (first) WIDE_GO(true, true, false)
      else if (last) WIDE_GO(true, false, true)
      else WIDE_GO(true, false, false)
#undef WIDE_GO
      HIP_CHECK(hipGetLastError());
    }
    HIP_CHECK(hipGetLastError());
    launch_merge(final_out, ix->part_v.p, ix->part_i.p, lists, (long long)keff, (long long)lists * keff, nq, K,
                 final_out ? d_oi + (size_t)q0 * K : nullptr, final_out ? d_od + (size_t)q0 * K : nullptr,
                 final_out && d_oc ? d_oc + q0 : nullptr, final_out && flags ? flags + q0 : nullptr,
                 final_out ? nullptr : d_pv + (size_t)q0 * keff, final_out ? nullptr : d_pi + (size_t)q0 * keff, st);
  }
  // queries with exact distance ties: the reference heap's insertion history (replay.hip, rp_scan_wide)
  if (final_out && replay_enabled()) run_tie_replay(ix, dQ, B, K, from, until, d_oi, d_od, d_oc, flags, st);
}

}  // namespace gulon
