// GroupedIndex.query's pre-selection (Index.scala:265-299), batched BY GROUP with 8-bit bound tables.
//
// grouped.hip's approximate pre-selection scores every row of a query's searched groups with
//     D~ = |q|^2 - 2 q.g + |x^|^2 + sum_j P_q[j][code_j] ,     P_q[j][c] = -2 q_j . c_j[c] ,   x^ = g + decode(codes),
// keeps the 64 smallest per query and re-ranks those with the reference's own arithmetic (gq_rerank, with a
// certificate).  gq_approx_scan does that one query per workgroup: 16 fp32 LDS gathers per (query, row) and every
// group's code blocks fetched again by each of the ~50 queries that search it -- 1.85 ms of the 2.46 ms batch at 10 M
// rows / LimitGroups(500), as long as the flat index takes to scan twenty times the rows.
//
// Here the (query, group) pairs are inverted -- for every group the list of queries that search it -- and a workgroup
// takes one group and SIXTEEN of its queries (gf_filter): the group's code blocks are read once per tile, and one 16-byte
// LDS gather returns the table bytes of all sixteen queries (filter.hip's layout).  The tables are 8-bit LOWER-BOUND
// levels of P_q (gf_quant; one set per query, shared by all its groups) plus a seventeenth table for an 8-bit level of
// the row's |x^|^2; a row survives for a query when the summed levels fit the budget the query's threshold leaves in
// that group.  The threshold is the 64th smallest D~ over the rows of the query's nearest groups (gf_quant scores a
// few hundred of them itself): real rows, so the 64 smallest of everything lie at or below it.  Survivors
// (a few hundred per query) are scored with gq_approx_scan's exact D~ arithmetic (gf_survivors), and from there the
// pipeline is unchanged: merge to the 64 smallest, gq_rerank, certificate, literal kernels for what it rejects.
//
// Nothing here is the reference's arithmetic; it only decides which 64 rows are re-scored with it, and it must not lose
// a row whose D~ is among the 64 smallest.  Every rounding goes the safe way: levels are rounded down (and saturate at
// 127, which alone exceeds any budget), budgets are rounded up, the threshold is raised by gq_rerank's error margin (the
// distance between a computed D~ and its real-number value), and a query whose inputs are not finite, whose threshold
// is missing (fewer than 64 rows) or whose survivors overflow their queue keeps everything / is flagged for the
// literal kernels.
#include "grouped_filter.hpp"

#include "scan.hpp"

namespace gulon {
namespace {

constexpr int GF_THREADS = 512;
constexpr int GF_NW = GF_THREADS / 64;
constexpr int GF_NT = 17;                       // tables per query: 16 quantizers and the row norm
constexpr float GF_SHRINK = 0.99999905f;        // 1 - 2^-20: a product of two roundings stays below the real product
#ifndef GULON_GF_NADD
#define GULON_GF_NADD 1
#endif
constexpr int GF_NADD = GULON_GF_NADD;          // table entries summed as bytes before a widening: 4 (6-bit levels), 2 (7-bit) or 1 (8-bit)
constexpr int GF_SAT = 255 / GF_NADD;           // a saturated level: alone it exceeds any budget (63 / 127 / 255)
constexpr float GF_LEVELS = (float)(GF_SAT - 3); // the largest budget in steps (two steps of slack below saturation)

__device__ inline uint32_t gf_pk_sub_sat_u16(uint32_t a, uint32_t b) {   // per 16-bit half: max(a - b, 0)
  uint32_t d;
  asm("v_pk_sub_u16 %0, %1, %2 clamp" : "=v"(d) : "v"(a), "v"(b));
  return d;
}

__device__ inline float gf_wave_sum(float x) {                          // gq_approx_scan's reduction order
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) x += __shfl_xor(x, o);
  return x;
}

// the 16 code bytes of row `row` as four words: one 16-byte word per row (VEC = 16), or ng <= 4 four-byte words in
// scan.hip's [row block][word][lane] layout (VEC = 4; quantizers from 4 ng on read as code 0: their tables are zero)
template <int VEC>
__device__ inline uint4 gf_row_words(const uint8_t *__restrict__ codes, int ng, int row) {
  if constexpr (VEC == 16) return reinterpret_cast<const uint4 *>(codes)[row];
  const uint32_t *cw = reinterpret_cast<const uint32_t *>(codes);
  const size_t o = ((size_t)(row >> 6) * ng) * 64 + (row & 63);
  uint4 w = uint4{cw[o], 0u, 0u, 0u};
  if (ng > 1) w.y = cw[o + 64];
  if (ng > 2) w.z = cw[o + 128];
  if (ng > 3) w.w = cw[o + 192];
  return w;
}

// ---- per index: 8-bit levels of the row norms above their group's smallest ----------------------------
// (per group: |x^|^2 = |g|^2 + 2 g.r^ + |r^|^2 moves with the group; against one floor for the whole index most of a
// row's budget would go to the distance between its group's norms and the smallest norm anywhere)
__global__ void gf_group_lo(const float *__restrict__ xnorm, const int *__restrict__ bounds, int g, float *__restrict__ xnlo,
                            unsigned *__restrict__ range_bits) {
  const int c = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (c >= g) return;
  float lo = INFINITY, hi = 0.f;
  for (int r = bounds[c] + lane; r < bounds[c + 1]; r += 64) { const float v = xnorm[r]; lo = fminf(lo, v); hi = fmaxf(hi, v); }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) { lo = fminf(lo, __shfl_xor(lo, o)); hi = fmaxf(hi, __shfl_xor(hi, o)); }
  if (lane == 0) {
    const bool any = bounds[c + 1] > bounds[c];
    xnlo[c] = any ? lo : 0.f;
    if (any) atomicMax(range_bits, __float_as_uint(hi - lo));     // non-negative: the bit patterns order like the values
  }
}

__global__ void gf_xcode(const float *__restrict__ xnorm, const int *__restrict__ bounds, int g, const float *__restrict__ xnlo,
                         int n, int npad, float inv, uint8_t *__restrict__ xcode) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npad) return;
  int c = 0;
  if (i < n) {
    int lo = 0, hi = g;                        // group of row i: largest c with bounds[c] <= i (empty groups skipped)
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (bounds[mid] <= i) lo = mid; else hi = mid; }
    c = min(255, max(0, (int)((xnorm[i] - xnlo[lo]) * inv)));     // xnlo + c * step <= xnorm[i]
  }
  xcode[i] = (uint8_t)c;
}

__global__ void gf_gnorm(const float *__restrict__ gcent, int g, int d, float *__restrict__ gnorm, unsigned *__restrict__ mx) {
  const int c = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (c >= g) return;
  float a = 0.f;
  for (int e = lane; e < d; e += 64) { const float v = gcent[(size_t)c * d + e]; a += v * v; }
  a = gf_wave_sum(a);
  if (lane == 0) { gnorm[c] = a; atomicMax(mx, __float_as_uint(a == a && a < INFINITY ? a : INFINITY)); }
}

// ---- per batch: for every group the queries that search it -------------------------------------------
// (Leaving out the pairs whose budget is negative even for the group's best row -- threshold - table minima -
// (|q - g|^2 - |g|^2 + the group's smallest row norm) < 0 -- was tried: under 1 % of the pairs at 10 M rows /
// LimitGroups(500), for two scattered loads per pair.  The lists themselves: every group has room for all B queries
// (pairs[c][B]), so a query is appended where it is counted, in gf_quant -- no offsets, no second pass over the pairs.)

// everything a tile's workgroup needs to start, in one record (one scalar round trip instead of four dependent ones)
// A group's first tile = the tiles of the groups before it: every workgroup adds those up itself (at most g counters,
// 40 KB from L2 -- a scan kernel of its own in front of this one cost a launch for 10 us of work); the last workgroup
// also leaves the totals (meta[0] = tiles, meta[1] = pairs).
__global__ __launch_bounds__(256) void gf_tiles(const int *__restrict__ gcnt, int g, int B, const int *__restrict__ bounds,
                                                const float *__restrict__ xnlo, const int *__restrict__ pairs,
                                                GfTile *__restrict__ tiles, int *__restrict__ meta) {
  // 64 groups per workgroup, four threads per group: a group's ~4 tiles are written side by side (one thread per group
  // wrote them one after the other, sixteen dependent loads each: 30 us at 1001 groups)
  __shared__ int s_w[4], s_p[4], s_t0[GF_TG], s_n[GF_TG];
  const int tid = threadIdx.x, lane = tid & 63;
  const int c0 = blockIdx.x * GF_TG;
  int before = 0, pbefore = 0;                 // tiles (and pairs) of the groups in front of this workgroup's
  for (int e = tid; e < c0; e += 256) { const int n = gcnt[e]; before += (n + GF_QT - 1) / GF_QT; pbefore += n; }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) { before += __shfl_xor(before, o); pbefore += __shfl_xor(pbefore, o); }
  if (lane == 0) { s_w[tid >> 6] = before; s_p[tid >> 6] = pbefore; }
  __syncthreads();
  if (tid < GF_TG) {                           // one wavefront: lane = group
    const int c = c0 + tid;
    const int cnt = c < g ? gcnt[c] : 0;
    const int mine = (cnt + GF_QT - 1) / GF_QT;
    int incl = mine, psum = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int up = __shfl_up(incl, o);
      if (tid >= o) incl += up;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) psum += __shfl_xor(psum, o);
    const int base = s_w[0] + s_w[1] + s_w[2] + s_w[3];
    s_t0[tid] = base + incl - mine;
    s_n[tid] = cnt;
    if (blockIdx.x == gridDim.x - 1 && tid == GF_TG - 1) {
      meta[0] = base + incl;
      meta[1] = s_p[0] + s_p[1] + s_p[2] + s_p[3] + psum;   // (pairs: counted only for the debugging aid's report)
    }
  }
  __syncthreads();
  const int gi = tid >> 2, c = c0 + gi;
  if (c >= g) return;
  const int cnt = s_n[gi], t0 = s_t0[gi], nt = (cnt + GF_QT - 1) / GF_QT;
  for (int ts = tid & 3; ts < nt; ts += 4) {
    const int first = ts * GF_QT;
    GfTile T;
    T.c = c; T.nq = min(GF_QT, cnt - first); T.r0 = bounds[c]; T.r1 = bounds[c + 1]; T.xl = xnlo[c];
    T.pad[0] = T.pad[1] = T.pad[2] = 0;
    for (int i = 0; i < GF_QT; i++) T.qid[i] = pairs[(size_t)c * B + first + min(i, T.nq - 1)];
    tiles[t0 + ts] = T;
  }
}

// ---- per query: threshold -> step, 8-bit levels of its tables ----------------------------------------
// qs[q] = {budget at base 0 (threshold + margin - sum of the tables' minima), 1 / step (0: keep
// every row), -, -}.  One step = (the largest budget any of the query's groups leaves) / GF_LEVELS.
template <int VEC>
__global__ __launch_bounds__(256) void gf_quant(float *__restrict__ P /* out: the query's table, gq_ptables' */,
                                                const float *__restrict__ pq_cents, const int *__restrict__ from,
                                                const int *__restrict__ sdim, int m, int m_pad, int k, int d,
                                                const float *__restrict__ Q, const float *__restrict__ cdist, int g,
                                                const float *__restrict__ gnorm, float gnmax, const float *__restrict__ xnlo,
                                                const int *__restrict__ nn, int stride, const int *__restrict__ nn_cnt,
                                                const uint8_t *__restrict__ codes, int ng, const float *__restrict__ xnorm,
                                                const float *__restrict__ gcent, const int *__restrict__ bounds,
                                                float xnmax, float xn_step, int *__restrict__ gcnt, int *__restrict__ pairs, int B,
                                                uint8_t *__restrict__ qb, float *__restrict__ qs) {
  extern __shared__ float gq_sm[];             // d query coordinates, the query's P table (m_pad x 256), GF_SAMPLE_ROWS values
  float *qv = gq_sm, *tab = gq_sm + d, *vals = tab + m_pad * 256;
  __shared__ float s_lo[16], s_mb[4], s_sbase[GF_SAMPLE_GROUPS];
  __shared__ int s_bad, s_sc[GF_SAMPLE_GROUPS], s_sr0[GF_SAMPLE_GROUPS], s_soff[GF_SAMPLE_GROUPS + 1], s_ns;
  __shared__ unsigned s_hist[256], s_prefix, s_remaining;
  const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float *Pq = P + (size_t)q * m_pad * 256;
  for (int e = tid; e < d; e += 256) qv[e] = Q[(size_t)q * d + e];
  if (tid == 0) s_bad = 0;
  __syncthreads();
  // the query's table P[j][c] = -2 (q_j . c_j[c]) -- gq_ptables' arithmetic, term by term -- into LDS and out to memory
  // (gf_survivors reads it there): thread = centroid
  for (int j = 0; j < m_pad; j++) {
    float acc = 0.f;
    if (j < m && tid < k) {
      const int fr = from[j], sj = sdim[j];
      const float *cc = pq_cents + (size_t)k * fr + (size_t)tid * sj;
      for (int t = 0; t < sj; t++) acc += qv[fr + t] * cc[t];
      acc *= -2.0f;
    }
    tab[j * 256 + tid] = acc;
    Pq[j * 256 + tid] = acc;
  }
  const int ngroups = nn_cnt[q];
  if (tid == 0) {
    // the threshold's sample: the rows of the query's nearest groups, as many groups as it takes to reach 256 rows (at
    // most GF_SAMPLE_GROUPS groups, at most GF_SAMPLE_ROWS rows: any rows do, they only have to be real ones)
    int ns = 0, total = 0;
    for (int t = 0; t < min(ngroups, GF_SAMPLE_GROUPS) && total < 256; t++) {
      const int c = nn[(size_t)q * stride + t];
      const int cnt = min(bounds[c + 1] - bounds[c], GF_SAMPLE_ROWS - total);
      if (cnt <= 0) continue;
      s_sc[ns] = c; s_sr0[ns] = bounds[c]; s_soff[ns] = total;
      total += cnt;
      ns++;
    }
    s_soff[ns] = total;
    s_ns = ns;
  }
  __syncthreads();
  float qq = 0.f;
  for (int e = lane; e < d; e += 64) qq += qv[e] * qv[e];
  qq = gf_wave_sum(qq);
  bool bad = false;
  for (int j = wave; j < 16; j += 4) {         // the smallest entry of every table (entries from k on are never looked up)
    float lo = INFINITY;
    if (j < m)
      for (int c = lane; c < k; c += 64) {
        const float v = tab[j * 256 + c];
        bad = bad || !(fabsf(v) < INFINITY);
        lo = fminf(lo, v);
      }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) lo = fminf(lo, __shfl_xor(lo, o));
    if (lane == 0) s_lo[j] = j < m ? lo : 0.f;
  }
  const int ns = s_ns, total = s_soff[ns];
  for (int t = wave; t < ns; t += 4) {         // the sampled groups' bases, gq_approx_scan's arithmetic
    float qg = 0.f;
    for (int e = lane; e < d; e += 64) qg += qv[e] * gcent[(size_t)s_sc[t] * d + e];
    qg = gf_wave_sum(qg);
    if (lane == 0) s_sbase[t] = qq - 2.0f * qg;
  }
  __syncthreads();
  // D~ of the sampled rows (gq_approx_scan's sum, term by term), as order-preserving keys in registers: up to
  // GF_SAMPLE_ROWS / 256 = 8 per thread
  constexpr int SR = GF_SAMPLE_ROWS / 256;
  unsigned skey[SR];
#pragma unroll
  for (int r = 0; r < SR; r++) {
    const int i = tid + 256 * r;
    skey[r] = 0xFFFFFFFFu;                     // (no row; a real key is never this: NaN values become +inf below)
    if (i < total) {
      int t = 0;
      while (t + 1 < ns && i >= s_soff[t + 1]) t++;
      const int row = s_sr0[t] + (i - s_soff[t]);
      const uint4 w = gf_row_words<VEC>(codes, ng, row);
      float acc = s_sbase[t] + xnorm[row];
      for (int j = 0; j < m_pad; j++) {
        const uint32_t x = j < 4 ? w.x : j < 8 ? w.y : j < 12 ? w.z : w.w;
        acc += tab[j * 256 + ((x >> (8 * (j & 3))) & 0xFFu)];
      }
      bad = bad || acc != acc;
      if (acc != acc) acc = INFINITY;
      const unsigned u = __float_as_uint(acc);
      skey[r] = (u & 0x80000000u) ? ~u : (u | 0x80000000u);     // unsigned order = float order
    }
  }
  // the 64th smallest of them (+inf with fewer than 64 rows: keep every row): a four-pass radix select over the keys in
  // registers -- 16 barriers where the bitonic sort of up to 2048 values in LDS took 66 (gq_select_groups' scheme: eight
  // sub-counters per bin, the bins walked by a prefix sum over one wavefront)
  float tq = INFINITY;
  if (total >= GF_LIST) {                      // (uniform over the workgroup)
    unsigned *hsub = reinterpret_cast<unsigned *>(vals);          // [256][8]: GF_SAMPLE_ROWS words
    if (tid == 0) { s_prefix = 0u; s_remaining = (unsigned)GF_LIST; }
    unsigned mask = 0u;
    for (int shift = 24; shift >= 0; shift -= 8) {
      for (int e = tid; e < 256 * 8; e += 256) hsub[e] = 0u;
      __syncthreads();
      const unsigned prefix = s_prefix;
#pragma unroll
      for (int r = 0; r < SR; r++) {
        const unsigned key = skey[r];
        if (key != 0xFFFFFFFFu && (key & mask) == prefix) atomicAdd(&hsub[((key >> shift) & 255u) * 8 + (tid & 7)], 1u);
      }
      __syncthreads();
      {
        unsigned h = 0;
#pragma unroll
        for (int x = 0; x < 8; x++) h += hsub[tid * 8 + x];
        s_hist[tid] = h;
      }
      __syncthreads();
      if (tid < 64) {
        const unsigned h0 = s_hist[4 * tid], h1 = s_hist[4 * tid + 1], h2 = s_hist[4 * tid + 2], h3 = s_hist[4 * tid + 3];
        const unsigned mine = h0 + h1 + h2 + h3;
        unsigned incl = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const unsigned up = __shfl_up(incl, o);
          if (tid >= o) incl += up;
        }
        const unsigned rem = s_remaining;
        const unsigned long long reach = __ballot(incl >= rem);
        const int first = reach ? __ffsll((long long)reach) - 1 : 63;
        if (tid == first) {
          unsigned cum = incl - mine;
          int bin = 4 * tid;
          if (cum + h0 >= rem) { }
          else if (cum + h0 + h1 >= rem) { cum += h0; bin += 1; }
          else if (cum + h0 + h1 + h2 >= rem) { cum += h0 + h1; bin += 2; }
          else { cum += h0 + h1 + h2; bin += 3; }
          s_remaining = rem - cum;
          s_prefix = prefix | ((unsigned)bin << shift);
        }
      }
      mask |= 255u << shift;
      __syncthreads();
    }
    const unsigned t = s_prefix;
    tq = __uint_as_float((t & 0x80000000u) ? (t & 0x7FFFFFFFu) : ~t);
  }
  // a lower bound of |q|^2 - 2 q.g over the searched groups, from the centroid distances the group selection already
  // has: |q - g|^2 - |g|^2, less what the two roundings can differ by (it only sizes the step; the budgets themselves
  // use gq_approx_scan's own base)
  float mb = INFINITY;
  for (int t = tid; t < ngroups; t += 256) {
    const int c = nn[(size_t)q * stride + t];
    if (bounds[c + 1] > bounds[c]) pairs[(size_t)c * B + atomicAdd(&gcnt[c], 1)] = q;   // the group's list of queries (room for all B) grows by this one
    const float base = (cdist[(size_t)q * g + c] - gnorm[c]) + xnlo[c];   // + the group's smallest row norm
    bad = bad || !(fabsf(base) < INFINITY);
    mb = fminf(mb, base);
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) mb = fminf(mb, __shfl_xor(mb, o));
  const float xg = __fsqrt_rn(qq) + __fsqrt_rn(gnmax);
  const float slack = 8.0f * (float)(d + 4) * 5.9604645e-8f * xg * xg;
  mb -= slack;
  if (lane == 0) s_mb[wave] = mb;
  if (__ballot(bad) != 0ull && lane == 0) atomicOr(&s_bad, 1);
  __syncthreads();
  const float minbase = fminf(fminf(s_mb[0], s_mb[1]), fminf(s_mb[2], s_mb[3]));
  float sumlo = 0.f;
  for (int j = 0; j < m; j++) sumlo += s_lo[j];
  const float xm = __fsqrt_rn(qq) + __fsqrt_rn(xnmax);
  const float margin = 4.0f * (float)(d + 2 * m + 16) * 5.9604645e-8f * xm * xm;     // gq_rerank's |D~ - real| bound
  const float budget0 = (tq + margin) - sumlo;
  const float rmax = (budget0 - minbase) * 1.001f;
  float inv = GF_LEVELS / fmaxf(rmax, 1e-30f);
  if (s_bad != 0 || !(tq < INFINITY) || !(margin < INFINITY) || !(fabsf(budget0) < INFINITY) || !(rmax < INFINITY) || !(inv < INFINITY) ||
      ngroups <= 0)
    inv = 0.f;                                 // keep every row (a short query: few rows; anything else overflows into the literal kernels)
  if (tid == 0) {
    qs[(size_t)q * 4] = budget0; qs[(size_t)q * 4 + 1] = inv; qs[(size_t)q * 4 + 2] = 0.f; qs[(size_t)q * 4 + 3] = 0.f;
  }
  uint8_t *out = qb + (size_t)q * GF_NT * 256;
  const float invs = inv * GF_SHRINK;
  for (int e = tid; e < GF_NT * 256; e += 256) {
    const int j = e >> 8, c = e & 255;
    int lv = 0;
    if (j < m && c < k) lv = min(GF_SAT, max(0, (int)((tab[j * 256 + c] - s_lo[j]) * invs)));
    else if (j == 16) lv = min(GF_SAT, max(0, (int)(((float)c * xn_step) * invs)));
    else if (j < m) lv = GF_SAT;
    out[e] = (uint8_t)lv;
  }
}

// ---- one group x sixteen of its queries --------------------------------------------------------------
// A tile is short -- a group's ~16 row blocks over 8 waves -- and everything it needs comes from memory: its record,
// the base inputs of its queries, its first code blocks, 68 KiB of tables.  A workgroup's time is the sum of those round
// trips unless they are all in flight at once: the record and the query ids are wave-uniform (one scalar round trip),
// and everything a wave will need is requested before anything is waited for.  Stamps at 10 M rows / LimitGroups(500),
// cycles per tile: record 1 400, vector round trip 3 000, byte transpose + LDS store 1 000-3 000, barrier 1 000, the
// scan of the rows 5 000 -- the last is the LDS gather floor (17 16-byte gathers per row, two workgroups per CU).
// (Persistent workgroups that request the next tile's loads before scanning the current tile's rows were tried: the
// second set of 50 registers does not fit beside the scan's at two workgroups per CU, and with the spills the kernel
// took 890 us against 580.)
template <int VEC>
__global__ __launch_bounds__(GF_THREADS) void gf_filter(const uint8_t *__restrict__ codes, int ng, const uint8_t *__restrict__ xcode, int d,
                                                        const float *__restrict__ Q, const float *__restrict__ gcent,
                                                        const GfTile *__restrict__ tiles, const int *__restrict__ meta,
                                                        const uint8_t *__restrict__ qb, const float *__restrict__ qs,
                                                        int *__restrict__ qcnt, uint2 *__restrict__ queue) {
  extern __shared__ uint4 tabs[];              // [GF_NT][256] entries: byte b of dword dd = the level for query 4 dd + b
  __shared__ uint32_t s_lim[GF_QT];
  __shared__ float s_base[GF_QT];
  const int tile = blockIdx.x;
  if (tile >= meta[0]) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool short_d = d <= 128;
  // the tile's record and query ids are wave-uniform (scalar loads); everything is requested before anything is used
  // the tile's inputs
  int n_c, n_nq, n_r0, n_r1;
  float n_xl;
  int n_qid[GF_QT];
  float n_bud[2], n_inv[2];                    // of queries wave and wave + 8
  float n_bq[2][2], n_bg[2];                   // their coordinates lane, lane + 64 and the centroid's (d <= 128)
  uint4 n_w;                                   // the wave's first code block
  uint32_t n_xc;
  uint32_t n_in0[GF_QT], n_in1[GF_QT];         // words of units tid, tid + 512 of the 16 x 64 of the quantizers' tables
  uint32_t n_inx[2];                           // eight row-norm levels of query tid & 15: codes 8 (tid >> 4) ..
  auto fetch = [&](int t) __attribute__((always_inline)) {
    const GfTile &T = tiles[t];
    n_c = T.c; n_nq = T.nq; n_r0 = T.r0; n_r1 = T.r1; n_xl = T.xl;
#pragma unroll
    for (int i = 0; i < GF_QT; i++) n_qid[i] = T.qid[i];
    const int qa = T.qid[wave], qb2 = T.qid[wave + GF_NW];     // (from memory: a register array indexed by the wave would live in scratch)
    n_bud[0] = qs[(size_t)qa * 4]; n_inv[0] = qs[(size_t)qa * 4 + 1];
    n_bud[1] = qs[(size_t)qb2 * 4]; n_inv[1] = qs[(size_t)qb2 * 4 + 1];
    if (short_d) {
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const int e = lane + 64 * h;
        const bool in = e < d;
        n_bq[0][h] = in ? Q[(size_t)qa * d + e] : 0.f;
        n_bq[1][h] = in ? Q[(size_t)qb2 * d + e] : 0.f;
        n_bg[h] = in ? gcent[(size_t)n_c * d + e] : 0.f;
      }
    }
    const int rb0 = (n_r0 >> 6) + wave;
    n_w = uint4{0u, 0u, 0u, 0u};
    n_xc = 0;
    if (rb0 < (n_r1 + 63) >> 6) { n_w = gf_row_words<VEC>(codes, ng, rb0 * 64 + lane); n_xc = xcode[rb0 * 64 + lane]; }
    auto load_unit = [&](int u, uint32_t (&in)[GF_QT]) __attribute__((always_inline)) {   // four consecutive codes of one table, all sixteen queries
      const int j = u >> 6, c4 = (u & 63) * 4;
#pragma unroll
      for (int i = 0; i < GF_QT; i++)
        in[i] = *reinterpret_cast<const uint32_t *>(qb + ((size_t)n_qid[i] * GF_NT + j) * 256 + c4);
    };
    load_unit(tid, n_in0);
    load_unit(tid + GF_THREADS, n_in1);
    {   // the seventeenth table (row-norm levels): every thread moves eight bytes of one query (a third 16-register unit
        // for the first wave alone cost every wave its registers: the unit lived in scratch, a wait behind every load)
      const uint2 v = *reinterpret_cast<const uint2 *>(qb + ((size_t)T.qid[tid & 15] * GF_NT + 16) * 256 + (tid >> 4) * 8);
      n_inx[0] = v.x; n_inx[1] = v.y;
    }
  };
  auto store_unit = [&](int u, const uint32_t (&in)[GF_QT]) __attribute__((always_inline)) {
    const int j = u >> 6, c4 = (u & 63) * 4;
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const uint32_t sel = 0x0C0C0000u | ((4u + e) << 8) | (uint32_t)e;   // byte e of the low word, byte e of the high word
      uint32_t o[4];
#pragma unroll
      for (int dd = 0; dd < 4; dd++) {
        const uint32_t t01 = __builtin_amdgcn_perm(in[4 * dd + 1], in[4 * dd], sel);
        const uint32_t t23 = __builtin_amdgcn_perm(in[4 * dd + 3], in[4 * dd + 2], sel);
        o[dd] = t01 | (t23 << 16);
      }
      tabs[j * 256 + c4 + e] = uint4{o[0], o[1], o[2], o[3]};
    }
  };
  fetch(tile);
  {
    // every query's base in this group (gq_approx_scan's arithmetic: the survivors' D~ starts from it) and how many
    // steps of budget it leaves
    const int c = n_c, nq = n_nq, r0 = n_r0, r1 = n_r1;
    int qid[GF_QT];
#pragma unroll
    for (int i = 0; i < GF_QT; i++) qid[i] = n_qid[i];
#pragma unroll
    for (int h2 = 0; h2 < 2; h2++) {
      const int i = wave + h2 * GF_NW;
      float qq = 0.f, qg = 0.f;
      if (short_d) {
#pragma unroll
        for (int h = 0; h < 2; h++)
          if (lane + 64 * h < d) { qq += n_bq[h2][h] * n_bq[h2][h]; qg += n_bq[h2][h] * n_bg[h]; }
      } else {
        const int q = tiles[tile].qid[i];
        for (int e = lane; e < d; e += 64) {
          const float x = Q[(size_t)q * d + e];
          qq += x * x;
          qg += x * gcent[(size_t)c * d + e];
        }
      }
      qq = gf_wave_sum(qq);
      qg = gf_wave_sum(qg);
      const float base = qq - 2.0f * qg;
      int lim;                                 // a row survives with a level sum below lim
      if (i >= nq) lim = 0;
      else if (n_inv[h2] == 0.f) lim = GF_SAT;
      else {
        const float f = floorf((n_bud[h2] - (base + n_xl)) * n_inv[h2]);
        lim = f >= (float)(GF_SAT - 1) ? GF_SAT : f >= -1.f ? (int)f + 2 : 0;   // (steps rounded down) + 1 for the rounding, + 1: "below"
      }
      if (lane == 0) { s_base[i] = base; s_lim[i] = (uint32_t)lim; }
    }
    store_unit(tid, n_in0);
    store_unit(tid + GF_THREADS, n_in1);
    {
      uint8_t *tb = reinterpret_cast<uint8_t *>(tabs + 16 * 256 + (tid >> 4) * 8) + (tid & 15);
#pragma unroll
      for (int e = 0; e < 8; e++) tb[e * 16] = (uint8_t)(n_inx[e >> 2] >> (8 * (e & 3)));
    }
    // the wave's code blocks: the first came with the tile
    const int rb_first = (r0 >> 6) + wave, rb_end = (r1 + 63) >> 6;
    uint4 wq[2] = {n_w, uint4{0u, 0u, 0u, 0u}};
    uint32_t xq[2] = {n_xc, 0u};
#pragma unroll
    for (int k = 1; k < 2; k++)
      if (rb_first + k * GF_NW < rb_end) {
        wq[k] = gf_row_words<VEC>(codes, ng, (rb_first + k * GF_NW) * 64 + lane);
        xq[k] = xcode[(rb_first + k * GF_NW) * 64 + lane];
      }
    __syncthreads();
    int kblk = 0;
    for (int rb = rb_first; rb < rb_end; rb += GF_NW, kblk++) {
      const int row = rb * 64 + lane;
      const bool valid = row >= r0 && row < r1;
      uint4 w;
      uint32_t xc;
      if (kblk == 0) { w = wq[0]; xc = xq[0]; }
      else if (kblk == 1) { w = wq[1]; xc = xq[1]; }
      else { w = gf_row_words<VEC>(codes, ng, row); xc = xcode[row]; }   // (a group of more than 1024 rows)
      uint32_t acc[8];
#pragma unroll
      for (int x = 0; x < 8; x++) acc[x] = 0;
#pragma unroll
      for (int b = 0; b < 16; b += GF_NADD) {
        uint32_t xs[4] = {0, 0, 0, 0};
#pragma unroll
        for (int a = 0; a < GF_NADD; a++) {    // bytes cannot carry: GF_NADD x GF_SAT <= 255
          const uint4 y = tabs[(b + a) * 256 + code_byte<16>(w, b + a)];
          xs[0] += y.x; xs[1] += y.y; xs[2] += y.z; xs[3] += y.w;
        }
#pragma unroll
        for (int dd = 0; dd < 4; dd++) {
          acc[2 * dd] += xs[dd];                                                   // (unmasked: corrected below)
          acc[2 * dd + 1] += __builtin_amdgcn_perm(0u, xs[dd], 0x0C030C01u);   // bytes 1 and 3
        }
      }
      {
        const uint4 y = tabs[16 * 256 + xc];
        const uint32_t ys[4] = {y.x, y.y, y.z, y.w};
#pragma unroll
        for (int dd = 0; dd < 4; dd++) {
          acc[2 * dd] += ys[dd];
          acc[2 * dd + 1] += __builtin_amdgcn_perm(0u, ys[dd], 0x0C030C01u);
        }
      }
      // filter.hip's running sum: acc[2 dd] holds W = S0 + 2^8 S1 + 2^16 S2 + 2^24 S3 (mod 2^32; S_i = the sum of byte i
      // over the 17 tables, < 2^13) and acc[2 dd + 1] = S1 + 2^16 S3 exactly, so W - 2^8 acc[2 dd + 1] = S0 + 2^16 S2 -- one
      // subtraction per dword here instead of a mask per table
#pragma unroll
      for (int dd = 0; dd < 4; dd++) acc[2 * dd] -= acc[2 * dd + 1] << 8;
      uint32_t left[8], any = 0;
#pragma unroll
      for (int x = 0; x < 8; x++) {
        // limits in the accumulators' layout: acc[2 dd] = queries 4 dd (low half) and 4 dd + 2, acc[2 dd + 1] = 4 dd + 1
        // and 4 dd + 3 (read per block: eight registers the next tile's loads need more)
        const int dd_ = x >> 1, o_ = x & 1;
        const uint32_t limp = s_lim[4 * dd_ + o_] | (s_lim[4 * dd_ + o_ + 2] << 16);
        left[x] = gf_pk_sub_sat_u16(limp, acc[x]);              // non-zero half <=> that query keeps this row
        any |= left[x];
      }
      if (__ballot(valid && any != 0) == 0ull) continue;
      // queue slots: one atomic per (query, row block) for all of the wave's survivors
      const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
      for (int x = 0; x < 8; x++) {
        const uint32_t l = valid ? left[x] : 0u;
        if (__ballot(l != 0) == 0ull) continue;
#pragma unroll
        for (int hf = 0; hf < 2; hf++) {
          const int qi = 4 * (x >> 1) + (x & 1) + 2 * hf;
          const unsigned long long mk = __ballot((hf ? l >> 16 : l & 0xFFFFu) != 0);
          if (mk == 0ull) continue;
          int slot = 0;
          if (lane == 0) slot = atomicAdd(&qcnt[qid[qi]], __popcll(mk));
          const int pos = __builtin_amdgcn_readfirstlane(slot) + __popcll(mk & lt);
          if (((mk >> lane) & 1ull) && pos < GF_CAP)
            queue[(size_t)qid[qi] * GF_CAP + pos] = uint2{(uint32_t)row, __float_as_uint(s_base[qi])};
        }
      }
    }
  }
}

// ---- the survivors' D~, gq_approx_scan's arithmetic; lists in its format --------------------------------
template <int VEC>
__global__ __launch_bounds__(64 * GF_WAVES) void gf_survivors(const uint8_t *__restrict__ codes, int ng, int m_pad,
                                                              const float *__restrict__ P, const float *__restrict__ xnorm,
                                                              const int *__restrict__ qcnt, const uint2 *__restrict__ queue,
                                                              float *__restrict__ lv, int *__restrict__ li,
                                                              int *__restrict__ nanflag) {
  // The GF_LIST smallest (D~, row) of the query's survivors, ascending.  Every survivor is scored into a register (an
  // order-preserving key and the row: up to GF_CAP / 1024 = 16 per thread, usually one); the GF_LIST-th smallest key is
  // found by a four-pass radix select over those registers (gf_quant's, gq_select_groups' scheme), the entries at or
  // below it are compacted into LDS and each of them finds its place by counting the entries below it.  (Sixteen wave
  // lists -- serial insertions -- and a bitonic sort of their 1024 entries, 55 barriers of a 1024-thread workgroup, before.)
  extern __shared__ float tab[];               // m_pad * 256 table entries; afterwards the select's counters and the compacted entries
  __shared__ unsigned s_hist[256], s_prefix, s_remaining;
  __shared__ int s_nan, s_cnt;
  constexpr int NT = 64 * GF_WAVES, PER = GF_CAP / NT;
  const int tid = threadIdx.x, lane = tid & 63, q = blockIdx.x;
  for (int e = tid; e < m_pad * 256; e += NT) tab[e] = P[(size_t)q * m_pad * 256 + e];
  if (tid == 0) { s_nan = 0; s_cnt = 0; s_prefix = 0u; }
  __syncthreads();
  const int total = qcnt[q], n_e = min(total, GF_CAP);
  const int want = min(GF_LIST, n_e);
  unsigned key[PER];
  int rowv[PER];
  bool nanv = false;
#pragma unroll
  for (int h = 0; h < PER; h++) {
    key[h] = 0xFFFFFFFFu;                      // (no entry; a real key is never this: NaN values become +inf)
    rowv[h] = INT_MAX;
    if (h * NT < n_e) {                        // (uniform)
      const int e = tid + h * NT;
      if (e < n_e) {
        const uint2 ent = queue[(size_t)q * GF_CAP + e];
        const int row = (int)ent.x;
        const uint4 w = gf_row_words<VEC>(codes, ng, row);
        float acc = __uint_as_float(ent.y) + xnorm[row];
        if constexpr (VEC == 16) {
#pragma unroll
          for (int b = 0; b < 16; b++) acc += tab[b * 256 + code_byte<16>(w, b)];
        } else {
          for (int gi = 0; gi < ng; gi++) {    // quantizers 0 .. m_pad - 1 in order, as gq_approx_scan<4> adds them
            const uint32_t x = gi == 0 ? w.x : gi == 1 ? w.y : gi == 2 ? w.z : w.w;
#pragma unroll
            for (int b = 0; b < 4; b++) acc += tab[(gi * 4 + b) * 256 + ((x >> (8 * b)) & 0xFFu)];
          }
        }
        if (acc != acc) { nanv = true; acc = INFINITY; }
        const unsigned u = __float_as_uint(acc);
        key[h] = (u & 0x80000000u) ? ~u : (u | 0x80000000u);      // unsigned order = float order
        rowv[h] = row;
      }
    }
  }
  if (tid == 0) s_remaining = (unsigned)max(want, 1);
  __syncthreads();                                               // (the tables have been read)
  unsigned *hsub = reinterpret_cast<unsigned *>(tab);            // [256][8]
  unsigned *ck = hsub + 256 * 8;                                 // [GF_PLACED] compacted keys
  int *ci = reinterpret_cast<int *>(ck + GF_PLACED);             // [GF_PLACED] ... and rows
  unsigned mask = 0u;
  if (want > 0) {                                                // (uniform)
    for (int shift = 24; shift >= 0; shift -= 8) {
      for (int e = tid; e < 256 * 8; e += NT) hsub[e] = 0u;
      __syncthreads();
      const unsigned prefix = s_prefix;
#pragma unroll
      for (int h = 0; h < PER; h++)
        if (h * NT < n_e) {
          const unsigned k = key[h];
          if (k != 0xFFFFFFFFu && (k & mask) == prefix) atomicAdd(&hsub[((k >> shift) & 255u) * 8 + (tid & 7)], 1u);
        }
      __syncthreads();
      if (tid < 256) {
        unsigned hh = 0;
#pragma unroll
        for (int x = 0; x < 8; x++) hh += hsub[tid * 8 + x];
        s_hist[tid] = hh;
      }
      __syncthreads();
      if (tid < 64) {
        const unsigned h0 = s_hist[4 * tid], h1 = s_hist[4 * tid + 1], h2 = s_hist[4 * tid + 2], h3 = s_hist[4 * tid + 3];
        const unsigned mine = h0 + h1 + h2 + h3;
        unsigned incl = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const unsigned up = __shfl_up(incl, o);
          if (tid >= o) incl += up;
        }
        const unsigned rem = s_remaining;
        const unsigned long long reach = __ballot(incl >= rem);
        const int first = reach ? __ffsll((long long)reach) - 1 : 63;
        if (tid == first) {
          unsigned cum = incl - mine;
          int bin = 4 * tid;
          if (cum + h0 >= rem) { }
          else if (cum + h0 + h1 >= rem) { cum += h0; bin += 1; }
          else if (cum + h0 + h1 + h2 >= rem) { cum += h0 + h1; bin += 2; }
          else { cum += h0 + h1 + h2; bin += 3; }
          s_remaining = rem - cum;
          s_prefix = prefix | ((unsigned)bin << shift);
        }
      }
      mask |= 255u << shift;
      __syncthreads();
    }
    const unsigned thr = s_prefix;                               // key of the want-th smallest value
#pragma unroll
    for (int h = 0; h < PER; h++)
      if (h * NT < n_e) {
        const unsigned k = key[h];
        if (k != 0xFFFFFFFFu && k <= thr) {
          const int p = atomicAdd(&s_cnt, 1);
          if (p < GF_PLACED) { ck[p] = k; ci[p] = rowv[h]; }
        }
      }
  }
  if (__ballot(nanv) != 0ull && lane == 0) atomicOr(&s_nan, 1);
  __syncthreads();
  const int c = s_cnt;                                           // >= want: the want smallest and whatever ties with the last
  if (c <= GF_PLACED) {
    for (int t = tid; t < c; t += NT) {
      const unsigned k = ck[t];
      const int r = ci[t];
      int place = 0;
      for (int j = 0; j < c; j++) {
        const unsigned kj = ck[j];
        const int rj = ci[j];
        place += (kj < k || (kj == k && rj < r)) ? 1 : 0;
      }
      if (place < GF_LIST) {
        lv[(size_t)q * GF_LIST + place] = __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
        li[(size_t)q * GF_LIST + place] = r;
      }
    }
  }
  if ((tid >= want || c > GF_PLACED) && tid < GF_LIST) { lv[(size_t)q * GF_LIST + tid] = INFINITY; li[(size_t)q * GF_LIST + tid] = INT_MAX; }
  // an overflowing queue has lost rows, a NaN value or more ties at the cut than the placing holds: the literal kernels
  if (tid < GF_WAVES) nanflag[q * GF_WAVES + tid] = tid == 0 ? ((s_nan != 0 || total > GF_CAP || c > GF_PLACED) ? 1 : 0) : 0;
}

}  // namespace

bool group_filter_applies(int m, int m_pad, int ng, int vec, int k, int d) {
  static const bool off = [] { const char *e = getenv("GULON_GROUPED_FILTER"); return e && atoi(e) == 0; }();
  return !off && m >= 1 && m <= 16 && m_pad <= 16 && m_pad == ng * vec && ((vec == 16 && ng == 1) || (vec == 4 && ng <= 4)) && k <= 256 &&
         d >= 1 && d <= 8192;
}

void group_filter_build(GroupFilter &gf, const float *xnorm, int n, const float *gcent, const int *bounds, int g, int d) {
  if (n <= 0) return;
  gf.xnlo.alloc((size_t)g);
  DevBuf<unsigned> rb(1);
  HIP_CHECK(hipMemset(rb.p, 0, sizeof(unsigned)));
  hipLaunchKernelGGL(gf_group_lo, dim3(ceil_div(g, 4)), dim3(256), 0, 0, xnorm, bounds, g, gf.xnlo.p, rb.p);
  HIP_CHECK(hipGetLastError());
  unsigned h = 0;
  HIP_CHECK(hipMemcpy(&h, rb.p, sizeof(h), hipMemcpyDeviceToHost));
  float range;
  memcpy(&range, &h, 4);
  if (!(range < INFINITY)) return;                              // (the caller keeps gq_approx_scan)
  gf.xn_step = range / 255.0f;
  const float inv = range > 0.f && gf.xn_step > 0.f ? (255.0f / range) * GF_SHRINK : 0.f;
  const int npad = ceil_div(n, 64) * 64;
  gf.xcode.alloc((size_t)npad);
  hipLaunchKernelGGL(gf_xcode, dim3(ceil_div(npad, 256)), dim3(256), 0, 0, xnorm, bounds, g, gf.xnlo.p, n, npad, inv, gf.xcode.p);
  HIP_CHECK(hipGetLastError());
  gf.gnorm.alloc((size_t)g);
  DevBuf<unsigned> gm(1);
  HIP_CHECK(hipMemset(gm.p, 0, sizeof(unsigned)));
  hipLaunchKernelGGL(gf_gnorm, dim3(ceil_div(g, 4)), dim3(256), 0, 0, gcent, g, d, gf.gnorm.p, gm.p);
  HIP_CHECK(hipGetLastError());
  unsigned hg = 0;
  HIP_CHECK(hipMemcpy(&hg, gm.p, sizeof(hg), hipMemcpyDeviceToHost));
  memcpy(&gf.gnmax, &hg, 4);
  if (!(gf.gnmax < INFINITY)) return;
  gf.built = true;
}

void group_filter_run(GroupFilter &gf, const uint8_t *codes, int ng, int vec, int m, int m_pad, int k, int d, float *P,
                      const float *pq_cents, const int *from, const int *sdim,
                      const float *xnorm, float xnmax, const float *gcent, const int *bounds, int g, const float *Q,
                      const float *cdist, const int *nn, int nn_stride, const int *nn_cnt, int B, float *amv,
                      int *ami, int *anan, hipStream_t st) {
  const size_t pairs_max = (size_t)B * nn_stride;
  const size_t tiles_max = pairs_max / GF_QT + (size_t)g + 1;
  GULON_UNSUPPORTED(tiles_max >= (1ull << 31), "too many (query, group) pairs");
  gf.gcnt.ensure((size_t)g + 1);
  gf.pairs.ensure((size_t)g * B); gf.tiles.ensure(tiles_max); gf.meta.ensure(4); gf.qcnt.ensure((size_t)B);
  gf.qb.ensure((size_t)B * GF_NT * 256); gf.qs.ensure((size_t)B * 4); gf.queue.ensure((size_t)B * GF_CAP);
  HIP_CHECK(hipMemsetAsync(gf.gcnt.p, 0, sizeof(int) * ((size_t)g + 1), st));
  HIP_CHECK(hipMemsetAsync(gf.qcnt.p, 0, sizeof(int) * (size_t)B, st));
  {
    const size_t lds_q = sizeof(float) * ((size_t)d + (size_t)m_pad * 256 + GF_SAMPLE_ROWS);
    auto kern = vec == 16 ? gf_quant<16> : gf_quant<4>;
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_q));
    hipLaunchKernelGGL(kern, dim3(B), dim3(256), lds_q, st, P, pq_cents, from, sdim, m, m_pad, k, d, Q, cdist, g, gf.gnorm.p, gf.gnmax, gf.xnlo.p, nn,
                       nn_stride, nn_cnt, codes, ng, xnorm, gcent, bounds, xnmax, gf.xn_step, gf.gcnt.p, gf.pairs.p, B, gf.qb.p, gf.qs.p);
  }
  hipLaunchKernelGGL(gf_tiles, dim3(ceil_div(g, GF_TG)), dim3(256), 0, st, gf.gcnt.p, g, B, bounds, gf.xnlo.p, gf.pairs.p, gf.tiles.p,
                     gf.meta.p);
  HIP_CHECK(hipGetLastError());
  const size_t lds_f = sizeof(uint4) * GF_NT * 256;
  static const bool attr = [&] {
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(gf_filter<16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_f));
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(gf_filter<4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_f));
    return true;
  }();
  (void)attr;
  {
    auto kern = vec == 16 ? gf_filter<16> : gf_filter<4>;
    hipLaunchKernelGGL(kern, dim3((unsigned)tiles_max), dim3(GF_THREADS), lds_f, st, codes, ng, gf.xcode.p, d, Q, gcent,
                       gf.tiles.p, gf.meta.p, gf.qb.p, gf.qs.p, gf.qcnt.p, gf.queue.p);
  }
  HIP_CHECK(hipGetLastError());
  if (getenv("GULON_GROUPED_STATS")) {   // debugging aid
    HIP_CHECK(hipStreamSynchronize(st));
    std::vector<int> h((size_t)B), meta(4);
    std::vector<float> hq((size_t)B * 4);
    HIP_CHECK(hipMemcpy(h.data(), gf.qcnt.p, sizeof(int) * (size_t)B, hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(meta.data(), gf.meta.p, sizeof(int) * 4, hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(hq.data(), gf.qs.p, sizeof(float) * 4 * (size_t)B, hipMemcpyDeviceToHost));
    long long sum = 0; int mx = 0, over = 0, keep = 0;
    for (int q = 0; q < B; q++) { sum += h[q]; mx = std::max(mx, h[q]); over += h[q] > GF_CAP; keep += hq[(size_t)q * 4 + 1] == 0.f; }
    fprintf(stderr, "[grouped] by-group filter: %d tiles over %d pairs; survivors per query mean %.1f max %d, %d queues overflowed, %d queries keep all\n",
            meta[0], meta[1], (double)sum / B, mx, over, keep);
  }
  {
    auto kern = vec == 16 ? gf_survivors<16> : gf_survivors<4>;
    hipLaunchKernelGGL(kern, dim3(B), dim3(64 * GF_WAVES), sizeof(float) * std::max((size_t)m_pad * 256, (size_t)256 * 8 + 2 * GF_PLACED), st,
                       codes, ng, m_pad, P, xnorm,
                       gf.qcnt.p, gf.queue.p, amv, ami, anan);
  }
  HIP_CHECK(hipGetLastError());
}

}  // namespace gulon
