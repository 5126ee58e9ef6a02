// Quantized lower-bound filter in front of the exact ADC scan (PQIndex.batchQuery,
// Index.scala:393-440).  Results are the exact path's, bit for bit; only the work changes.
//
// The exact scan gathers one fp32 table entry per (query, row, quantizer) from LDS, and the LDS
// gather bandwidth is what bounds it.  A table entry here is 8 bits, so one
// ds_read_b128 serves 16 queries instead of 4:
//
//   1. the exact distances of a few thousand strided sample rows give every query a valid upper
//      bound tau on its final (K+1)-th distance (the (K+1)-th smallest of a subset);
//   2. per query the fp32 table T_j[c] is quantized DOWNWARDS (min_j comes out of build_tables):
//        q_j[c] = min(QMAX, floor((T_j[c] - min_j) / delta)),  delta = (tau' - sum_j min_j) / QL,
//      where tau' = tau * (1 + 2 m u), u = 2^-24, covers the rounding of the reference's
//      sequential fp32 sum D against the real sum R (all terms >= 0: D >= R (1 - m u)).
//      Then  sum_j min_j + delta * sum_j q_j[c_j]  <=  R,  and a row with  sum_j q_j > QL  has
//      R > tau', hence D > tau: it cannot be among the K+1 smallest and is dropped;
//   3. the (few) surviving (query, row) pairs are queued and re-evaluated with the exact fp32
//      tables in the reference's summation order, and merged into the running (K+1)-lists under
//      the same (distance, row id) order as the exact scan.
//
// The rows go through three filter stages over disjoint, strided sets of row blocks (two short
// ones that tighten tau, then the rest).  A query whose bound is unusable (NaN/inf) or whose survivor queue overflows is redone
// by the exact scan (per query tile, decided on the device), so the filter never changes a result.
#include <atomic>
#include <map>
#include <type_traits>

#include "scan.hpp"

namespace gulon {

namespace {

constexpr size_t FILTER_LDS_BUDGET = 144 * 1024;
#ifndef GULON_FILTER_THREADS
#define GULON_FILTER_THREADS 1024
#endif
constexpr int FILTER_THREADS = GULON_FILTER_THREADS;   // filter_kernel's workgroup
constexpr int BOUND_THREADS = 1024;                     // bound_tables' workgroup
constexpr int NSLOT = 16;   // survivor sub-queues per query (workgroups of different chunks use different ones)

__device__ inline uint32_t pk_sub_sat_u16(uint32_t a, uint32_t b) {   // per 16-bit half: max(a - b, 0)
  uint32_t d;
  asm("v_pk_sub_u16 %0, %1, %2 clamp" : "=v"(d) : "v"(a), "v"(b));
  return d;
}

// fp32 table entry (j, c) of query q; W queries are interleaved per entry (scan.hip build_tables)
__device__ inline float table_at(const float *__restrict__ tables, int W, int m_pad, int q, int j, int c) {
  return tables[(((size_t)(q / W) * m_pad + j) * 256 + c) * W + q % W];
}

#ifdef GULON_FILTER_STAMPS
// experiment builds only (scripts/variant.sh ... -DGULON_FILTER_STAMPS): every wave of every main-stage workgroup records
// when it started, had its tables staged and left its row loop (100 MHz wall clock) and where it ran
// (HW_ID, XCC_ID) -> GULON_FILTER_STAMPS=<file> dumps [workgroup][52] uint64 after every main-stage launch
__device__ unsigned long long *g_filter_stamps;
__device__ unsigned long long *g_qt_stamps;   // qt_quantize: [workgroup][4] = entry, bounds ready, loop done, 0
__device__ unsigned long long *g_bt_stamps;   // bound_tables: [workgroup][8] = entry, tables built, scan done, end, + 4 inside the table build
#endif
// ---- quantize the tables of four queries (one dword of an entry) against the current bounds ----
__global__ __launch_bounds__(256) void qt_quantize(const float *__restrict__ tables, int W, int Bp, int m_pad, int k,
                                                   int B, const float *__restrict__ mins,
                                                   const float *__restrict__ fin_v, const int *__restrict__ fin_i,
                                                   const float *__restrict__ tau0, int keff, int qmax, int QW,
                                                   uint8_t *__restrict__ qtab /*[Bq/QW][m_pad][256][QW]*/,
                                                   int *__restrict__ fb_tile, int qt,
                                                   const int *__restrict__ slot = nullptr /* table / minima of query q: those of slot[q] */) {
  // One workgroup = the 256 centroids of ONE quantizer for FOUR queries (one dword of a table entry): 4 x the
  // workgroups of a 16-query form with a quarter of the serial work each -- the kernel is latency, not throughput
  // (20 us per launch on a 1.25 M-row shard as 1024 workgroups that walked their 16 queries in four dependent steps).
  __shared__ float s_inv32[4];
  __shared__ float s_min[4];
  __shared__ int s_dead[4];
  __shared__ float s_mins[4 * 144];                // the four queries' table minima (m_pad <= 144)
  const int g4 = blockIdx.x, j = blockIdx.y, c = threadIdx.x;
#ifdef GULON_FILTER_STAMPS
  const unsigned long long qs0 = __builtin_amdgcn_s_memrealtime();
#endif
  // everything the workgroup reads is requested up front -- its table entries, the 4 x m_pad minima (coalesced) and
  // each query's bound -- so that the memory round trips overlap
  float vv[4];
#pragma unroll
  for (int u = 0; u < 4; u++) {
    const int q = g4 * 4 + u;
    vv[u] = (q < Bp && c < k) ? table_at(tables, W, m_pad, slot ? slot[q] : q, j, c) : 0.f;
  }
  for (int e = c; e < 4 * m_pad; e += 256) {
    const int qq = e / m_pad, jj = e - qq * m_pad, q = g4 * 4 + qq;
    s_mins[e] = q < B ? mins[(size_t)(slot ? slot[q] : q) * m_pad + jj] : 0.f;
  }
  float tau = INFINITY;
  int fb = 1;
  if (c < 4 && g4 * 4 + c < B) {
    const int q = g4 * 4 + c;
    // bound = the sample's, tightened by the running list once that is full
    const float t0 = tau0[q], fv = fin_v[(size_t)q * keff + keff - 1];
    const int fi = fin_i[(size_t)q * keff + keff - 1];
    fb = fb_tile[q / qt];
    tau = fi != INT_MAX ? fminf(t0, fv) : t0;
  }
  __syncthreads();
  if (c < 4) {
    const int q = g4 * 4 + c;
    int dead = 1;
    double delta = 1.0;
    if (q < B) {
      if (fb != 0) {
        // this query tile already goes to the exact scan (unusable bound, queue overflow, or a
        // first stage that let too many rows through): nothing of it is filtered any more
      } else if (!(tau < INFINITY)) {
        if (j == 0) fb_tile[q / qt] = 1;          // no usable bound: this query is redone exactly
      } else {
        double sum_min = 0.0;
        for (int jj = 0; jj < m_pad; jj++) sum_min += (double)s_mins[c * m_pad + jj];
        const double taup = (double)tau * (1.0 + 2.0 * m_pad * 5.97e-8) * (1.0 + 1e-9);
        double budget = taup - sum_min;
        if (!(budget > 0.0)) budget = 0.0;
        delta = budget / (double)(qmax - 1);
        if (delta < 1e-290) delta = 1e-290;
        dead = 0;
      }
    }
    s_dead[c] = dead;
    // 1 / delta for the fp32 levels below: rounded to fp32 (<= 2^-24 relative), then shrunk by 2^-21; a delta so
    // small that the reciprocal overflows fp32 leaves +inf -> every entry above its minimum gets the top level (valid:
    // levels only ever round down from x / delta, and the top level is what min(QMAX, .) would give)
    float inv32 = (float)(1.0 / delta);
    inv32 = inv32 < INFINITY ? inv32 * (1.0f - 4.76837158e-7f) : 3.0e38f;
    s_inv32[c] = inv32;
    s_min[c] = s_mins[c * m_pad + j];
  }
  __syncthreads();
#ifdef GULON_FILTER_STAMPS
  const unsigned long long qs1 = __builtin_amdgcn_s_memrealtime();
#endif
  // The level of an entry in fp32, rounded DOWN for certain: x = T - min (one rounding, <= 2^-24 relative), times the
  // reciprocal of delta, which the prologue shrank by 2^-21 (fp64 -> fp32, the shrinking, the subtraction, the product: four roundings of
  // 2^-24 each = 2^-22) -- they cannot carry the product above the real quotient x / delta, so  level * delta <= x  holds as the bound
  // needs it (the fp64 form of this -- conversions, a guard loop -- was a dozen half- and quarter-rate instructions
  // per entry on a vector ALU that another batch's main stage keeps 90 % busy).  Levels are < 256: exact in fp32.
  uint32_t word = 0;
#pragma unroll
  for (int u = 0; u < 4; u++) {
    const int q = g4 * 4 + u;
    const bool have = q < Bp && c < k;
    const float v = vv[u];
    int qv = qmax;
    if (have && !s_dead[u] && v == v) {
      float x = v - s_min[u];
      if (!(x > 0.f)) x = 0.f;
      const float r = x * s_inv32[u];
      if (r < (float)qmax) qv = (int)r;                      // floor: r >= 0
    }
    word |= (uint32_t)qv << (8 * u);
  }
  // queries 4 g4 .. + 3 -> dword (4 g4 % QW) / 4 of entry [4 g4 / QW][j][c] (QW bytes per entry)
  const int grp = (g4 * 4) / QW, part = (g4 * 4 % QW) / 4;
  reinterpret_cast<uint32_t *>(qtab)[(((size_t)grp * m_pad + j) * 256 + c) * (QW / 4) + part] = word;
#ifdef GULON_FILTER_STAMPS
  if (c == 0 && g_qt_stamps) {
    unsigned long long *o = g_qt_stamps + 4 * ((size_t)blockIdx.y * gridDim.x + blockIdx.x);
    o[0] = qs0; o[1] = qs1; o[2] = __builtin_amdgcn_s_memrealtime(); o[3] = 0;
  }
#endif
}

// ---- initial bounds from a strided sample of row blocks ---------------------------------------
// One workgroup per W queries (bound_tables below), their W-interleaved fp32 table in LDS, lane = row as in the
// exact scan -- but no top-k lists in the loop: every lane only keeps the minimum exact distance
// of the rows it saw (1024 disjoint groups of rows per workgroup).  Any K+1 group minima belong
// to K+1 distinct rows, so the (K+1)-th smallest group minimum bounds the final (K+1)-th
// distance from above; with 1024 groups it is almost always the sample's own (K+1)-th distance.
// Selection = one 64-lane bitonic sort per wave and query, then a sorted merge of the 16 waves.
__device__ inline float sort64_asc(float x, int lane) {
#pragma unroll
  for (int k = 2; k <= 64; k <<= 1)
#pragma unroll
    for (int j = k >> 1; j >= 1; j >>= 1) {
      const float y = __shfl_xor(x, j);
      const bool up = (lane & k) == 0, lower = (lane & j) == 0;
      x = (lower == up) ? fminf(x, y) : fmaxf(x, y);
    }
  return x;
}
// a, b ascending: the 64 smallest of both, ascending
__device__ inline float merge64_asc(float a, float b, int lane) {
  float x = fminf(a, __shfl(b, 63 - lane));   // bitonic sequence holding the 64 smallest
#pragma unroll
  for (int j = 32; j >= 1; j >>= 1) {
    const float y = __shfl_xor(x, j);
    x = (lane & j) == 0 ? fminf(x, y) : fmaxf(x, y);
  }
  return x;
}

// the same two networks for 64-bit keys (the survivor pass orders (distance bits << 32) | row id)
__device__ inline unsigned long long shfl_u64(unsigned long long x, int src) {
  const unsigned lo = (unsigned)__shfl((int)(unsigned)x, src), hi = (unsigned)__shfl((int)(unsigned)(x >> 32), src);
  return ((unsigned long long)hi << 32) | lo;
}
__device__ inline unsigned long long shfl_xor_u64(unsigned long long x, int m) {
  const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)x, m), hi = (unsigned)__shfl_xor((int)(unsigned)(x >> 32), m);
  return ((unsigned long long)hi << 32) | lo;
}
__device__ inline unsigned long long sort64_u64(unsigned long long x, int lane) {
#pragma unroll
  for (int k = 2; k <= 64; k <<= 1)
#pragma unroll
    for (int j = k >> 1; j >= 1; j >>= 1) {
      const unsigned long long y = shfl_xor_u64(x, j);
      const bool up = (lane & k) == 0, lower = (lane & j) == 0;
      x = (lower == up) ? (x < y ? x : y) : (x < y ? y : x);
    }
  return x;
}
// a, b ascending: the 64 smallest of both, ascending
__device__ inline unsigned long long merge64_u64(unsigned long long a, unsigned long long b, int lane) {
  const unsigned long long br = shfl_u64(b, 63 - lane);
  unsigned long long x = a < br ? a : br;      // bitonic sequence holding the 64 smallest
#pragma unroll
  for (int j = 32; j >= 1; j >>= 1) {
    const unsigned long long y = shfl_xor_u64(x, j);
    x = (lane & j) == 0 ? (x < y ? x : y) : (x < y ? y : x);
  }
  return x;
}

template <int W> struct FTab;
template <> struct FTab<4> { using type = float4; };
template <> struct FTab<2> { using type = float2; };
template <> struct FTab<1> { using type = float; };

// ---- the batch's first kernel: reset + Index.prepareQuery + sample scan in ONE launch ---------------------------
// The three were separate launches (filter_reset, build_tables, bound_scan): 5 + 25 + 40 us on a 1.25 M-row shard, a
// quarter of what the batch spends outside its main-stage kernel, and build_tables' 4096 workgroups take the whole chip
// from another batch's filter kernel for their 20 us.  A sample-scan workgroup needs the fp32 tables of its W queries
// in LDS anyway, so it BUILDS them there -- Index.prepareQuery's arithmetic (Index.scala:352-383: d = q - c,
// sum += d * d, sequential, unfused), the same expression build_tables evaluates, hence the same bits -- writes them
// out for the survivor pass, reduces the per-(query, quantizer) minima the quantisation needs, and clears its share of
// the batch's counters on the way.  16 waves: wave w builds quantizers w, w + 16, ...; quantizer and
// query are wave-uniform, so the query values arrive through scalar loads.
template <int VEC, int W>
__global__ __launch_bounds__(BOUND_THREADS) void bound_tables(
    const float *__restrict__ cents, const int *__restrict__ from, const int *__restrict__ sdim, int d, int m, int k,
    const float *__restrict__ Q, float *__restrict__ tables, float *__restrict__ mins,
    const uint8_t *__restrict__ codes, const uint8_t *__restrict__ perm /* codes = the conflict-ordered copy: its row order */,
    int pwin /* blocks per ordering window of that copy - 1 */, int ng, int m_pad, int row_from, int row_until, int rb_begin, int e_count,
    RbMap mp, int B, int keff, float *__restrict__ tau0, float *__restrict__ fin_v, int *__restrict__ fin_i,
    float *__restrict__ bounds_out, unsigned *__restrict__ gtau, int n_gtau, int *__restrict__ fb_tile, int n_fb,
    int *__restrict__ sv_cnt, int n_cnt) {
  constexpr int NW = BOUND_THREADS / 64;
  using Word = typename CodeWord<VEC>::type;
  using TV = typename FTab<W>::type;
  extern __shared__ uint4 qlds[];
  TV *lds = reinterpret_cast<TV *>(qlds);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int qg = blockIdx.x;
#ifdef GULON_FILTER_STAMPS
  unsigned long long bt0 = __builtin_amdgcn_s_memrealtime(), bt1 = 0, bt2 = 0, bx[4] = {0, 0, 0, 0};
#endif
  // (1) this batch's counters: pruning thresholds of the fallback scan, tile flags, survivor sub-queue fill levels
  for (int t = qg * BOUND_THREADS + tid; t < max(n_gtau, max(n_fb, n_cnt)); t += gridDim.x * BOUND_THREADS) {
    if (t < n_gtau) gtau[t] = 0x7F800000u;   // +inf
    if (t < n_fb) fb_tile[t] = 0;
    if (t < n_cnt) sv_cnt[t] = 0;
  }
  if (qg * W >= B) return;                    // a padding group of the last query tile: no table, no bound
  // (2) tables of queries qg*W .. +W-1, W-interleaved: entry (j, c) at lds[j * 256 + c].  Wave w owns quantizers
  // w, w + 16, ...: sub-vector bounds and query values are wave-uniform, lane l evaluates centroids l, l + 64,
  // l + 128, l + 192 one after the other, and the table minimum of a (query, quantizer) pair is a reduction inside the
  // wave.  The loops are ROLLED on purpose: every wave runs through this code once per batch, out of a cold instruction
  // cache -- unrolled (6.5 KiB) the section took 17.8 us of which the arithmetic is < 1 us (per-phase stamps); what
  // such straight-line code costs is its size.
  for (int j = wave; j < m_pad; j += NW) {
    int fr = 0, s = 0;
    if (j < m) { fr = from[j]; s = sdim[j]; }
#ifdef GULON_FILTER_STAMPS
    asm volatile("" :: "s"(fr), "s"(s));
    if (j == wave) bx[0] = __builtin_amdgcn_s_memrealtime();     // sub-vector bounds arrived
#endif
    const float *cc = cents + (size_t)k * fr;
    float mnu[W];
#pragma unroll
    for (int u = 0; u < W; u++) mnu[u] = INFINITY;
#pragma unroll 1
    for (int i = 0; i < 4; i++) {
      const int c = lane + 64 * i;
      float acc[W];
#pragma unroll
      for (int u = 0; u < W; u++) acc[u] = 0.f;
#pragma unroll 1
      for (int t0 = 0; t0 < s; t0 += 8) {
        // eight dimensions per step.  The query values: lane 8 u + e loads component t0 + e of query u (one vector
        // load; 32 guarded scalar loads took eight memory round trips), every lane then reads them with v_readlane.
        // Dimensions beyond s contribute (0 - 0)^2 = +0 to a sum that is >= +0 or NaN: no change.
        float qreg = 0.f;
        {
          const int lu = lane >> 3, le = lane & 7;
          if (lu < W && t0 + le < s && qg * W + lu < B) qreg = Q[(size_t)(qg * W + lu) * d + fr + t0 + le];
        }
        // the centroid's eight components: two 16-byte loads where all eight exist (dword-aligned: s is arbitrary).
        // Eight dword loads per lane, 32 bytes apart from lane to lane, touched every cache line eight times over and
        // made this loop L1-bound: 3 us per centroid group (per-phase stamps)
        float cv[8];
        if (t0 + 8 <= s) {
          typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
          f32x4u lo = {0.f, 0.f, 0.f, 0.f}, hi = {0.f, 0.f, 0.f, 0.f};
          if (c < k) {
            const f32x4u *row = reinterpret_cast<const f32x4u *>(cc + (size_t)c * s + t0);
            lo = row[0]; hi = row[1];
          }
#pragma unroll
          for (int e = 0; e < 4; e++) { cv[e] = lo[e]; cv[4 + e] = hi[e]; }
        } else {
#pragma unroll
          for (int e = 0; e < 8; e++) cv[e] = (t0 + e < s && c < k) ? cc[(size_t)c * s + t0 + e] : 0.f;
        }
#pragma unroll
        for (int e = 0; e < 8; e++) {
#pragma unroll
          for (int u = 0; u < W; u++) {
            const float dd = readlane_f(qreg, 8 * u + e) - cv[e];
            acc[u] += dd * dd;
          }
        }
      }
#ifdef GULON_FILTER_STAMPS
      asm volatile("" :: "v"(acc[0]));
      if (j == wave && i == 0) bx[1] = __builtin_amdgcn_s_memrealtime();   // first centroid group evaluated
      if (j == wave && i == 3) bx[2] = __builtin_amdgcn_s_memrealtime();   // last centroid group evaluated
#endif
      TV tv;
#pragma unroll
      for (int u = 0; u < W; u++) {
        if (!(c < k && qg * W + u < B)) acc[u] = 0.f;         // entries beyond k / of padding queries: zero
        reinterpret_cast<float *>(&tv)[u] = acc[u];
        if (c < k && acc[u] == acc[u]) mnu[u] = fminf(mnu[u], acc[u]);   // NaN entries are ignored
      }
      lds[j * 256 + c] = tv;       // (to LDS only: see the copy-out below)
    }
#pragma unroll
    for (int u = 0; u < W; u++) {
      float x = mnu[u];
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) x = fminf(x, __shfl_xor(x, o));
      if (lane == 0) mins[(size_t)(qg * W + u) * m_pad + j] = x;
    }
  }
#ifdef GULON_FILTER_STAMPS
  bx[3] = __builtin_amdgcn_s_memrealtime();                      // this wave's quantizers done (before the barrier)
#endif
  __syncthreads();
#ifdef GULON_FILTER_STAMPS
  bt1 = __builtin_amdgcn_s_memrealtime();
#endif
  // (3) the sample scan (see above): lane = row, every lane keeps the minimum exact distance it saw
  const Word *cw = reinterpret_cast<const Word *>(codes);
  int mp_p = wave / mp.width, mp_r = wave - mp_p * mp.width;
  auto block_of = [&](int p, int r) { return rb_begin + p * mp.period + mp.lo + r; };
  auto advance = [&](int &p, int &r) { r += NW; while (r >= mp.width) { r -= mp.width; p++; } };
  float mn[W];
#pragma unroll
  for (int u = 0; u < W; u++) mn[u] = INFINITY;
  Word w_first{};
  if (wave < e_count) w_first = cw[((size_t)block_of(mp_p, mp_r) * ng) * 64 + lane];
  // The tables leave for global memory (the survivor pass and the quantisation read them) only now, in one burst
  // BEHIND the first code-word request.  vmcnt counts loads and stores together, in order: with the store of a
  // centroid group inside the build loop, the next group's loads waited for that store to be acknowledged (~3 us
  // each, per-phase stamps: first group after 2.7 us, fourth after 11.5); a wait for the code word requested before
  // the burst leaves the stores in flight.
  for (int e = tid; e < m_pad * 256; e += BOUND_THREADS)
    reinterpret_cast<TV *>(tables)[(size_t)qg * m_pad * 256 + e] = lds[e];
  for (int e = wave; e < e_count; e += NW) {
    const int rb = block_of(mp_p, mp_r);
    advance(mp_p, mp_r);
    const Word *p = cw + ((size_t)rb * ng) * 64 + lane;
    Word w = w_first;
    if (e + NW < e_count) w_first = cw[((size_t)block_of(mp_p, mp_r) * ng) * 64 + lane];
    float acc[W];                            // the reference's order: j ascending, unfused fp32
#pragma unroll
    for (int u = 0; u < W; u++) acc[u] = 0.f;
    for (int g = 0; g < ng; g++) {
      Word wn = w;
      if (g + 1 < ng) wn = p[(size_t)(g + 1) * 64];
      const TV *tj = lds + g * VEC * 256;
#pragma unroll
      for (int b = 0; b < VEC; b++) {
        const TV t = tj[b * 256 + code_byte<VEC>(w, b)];
        const float *tf = reinterpret_cast<const float *>(&t);
#pragma unroll
        for (int u = 0; u < W; u++) acc[u] += tf[u];
      }
      w = wn;
    }
    // which row a lane holds only matters in a window the range cuts (the conflict-ordered copy deals a window's rows to
    // the lanes of its blocks by `perm`; the sample only needs distances of DISTINCT rows of the range)
    bool valid = true;
    const int wrow = (rb & ~pwin) * 64;               // first row of the block's ordering window
    if (wrow < row_from || wrow + 64 * (pwin + 1) > row_until) {
      const int row = wrow + (perm ? (int)perm[(size_t)rb * 64 + lane] : lane);
      valid = row >= row_from && row < row_until;
    }
    if (valid) {
#pragma unroll
      for (int u = 0; u < W; u++) mn[u] = fminf(mn[u], acc[u]);   // NaN distances are ignored
    }
  }
  __syncthreads();                     // the table is dead: reuse LDS for the per-wave sorted minima
#ifdef GULON_FILTER_STAMPS
  bt2 = __builtin_amdgcn_s_memrealtime();
#endif
  float *sv = reinterpret_cast<float *>(qlds);
#pragma unroll
  for (int u = 0; u < W; u++) sv[(u * NW + wave) * 64 + lane] = mn[u];
#pragma unroll 1                       // (one copy of the sorting network, not W: code size again)
  for (int u = 0; u < W; u++) {
    float *slot = sv + (u * NW + wave) * 64 + lane;
    *slot = sort64_asc(*slot, lane);
  }
  __syncthreads();
  if (wave < W) {
    const int u = wave, q = qg * W + u;
    float best = sv[(u * NW) * 64 + lane];
#pragma unroll 1
    for (int w2 = 1; w2 < NW; w2++) best = merge64_asc(best, sv[(u * NW + w2) * 64 + lane], lane);
    if (q < B) {
      if (lane == keff - 1) tau0[q] = best;          // +inf when fewer than K+1 groups saw a row
      if (lane < keff) { fin_v[(size_t)q * keff + lane] = INFINITY; fin_i[(size_t)q * keff + lane] = INT_MAX; }
      if (bounds_out && lane < keff) bounds_out[(size_t)q * keff + lane] = best;   // for the other shards
    }
  }
#ifdef GULON_FILTER_STAMPS
  if (tid == 0 && g_bt_stamps) {
    unsigned long long *o = g_bt_stamps + 8 * (size_t)qg;
    o[0] = bt0; o[1] = bt1; o[2] = bt2; o[3] = __builtin_amdgcn_s_memrealtime();
    o[4] = bx[0]; o[5] = bx[1]; o[6] = bx[2]; o[7] = bx[3];
  }
#endif
}

// ---- bounds shared across shards: tau0[q] = the keff-th smallest of the union of `lists` ascending
// arrays of keff sample distances (one per shard, this shard's own among them).  Each of them belongs
// to a distinct row of the whole index, so the value bounds the index-wide keff-th distance from above.
// One wave per query; bisection over the ordered bit patterns (32 counting passes over <= 64 * lists values).
__global__ __launch_bounds__(256) void shared_tau(const float *__restrict__ all /*[lists][B][keff]*/, int lists, int B,
                                                  int keff, float *__restrict__ tau0) {
  const int lane = threadIdx.x & 63;
  const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (q >= B) return;
  const int n = lists * keff;
  auto key_at = [&](int i) {
    const int l = i / keff, e = i - l * keff;
    const uint32_t u = __float_as_uint(all[((size_t)l * B + q) * keff + e]);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);          // unsigned order = float order
  };
  constexpr int R = 4;                                           // values per lane held in registers
  uint32_t held[R];
#pragma unroll
  for (int r = 0; r < R; r++) held[r] = lane + 64 * r < n ? key_at(lane + 64 * r) : 0xFFFFFFFFu;
  uint32_t t = 0;
  for (int bit = 31; bit >= 0; bit--) {
    const uint32_t cand = t | (1u << bit);
    int c = 0;
#pragma unroll
    for (int r = 0; r < R; r++) c += held[r] < cand ? 1 : 0;
    for (int i = lane + 64 * R; i < n; i += 64) c += key_at(i) < cand ? 1 : 0;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) c += __shfl_xor(c, o);
    if (c < keff) t = cand;                                      // fewer than keff values below: go up
  }
  if (lane == 0) {
    const float v = __uint_as_float((t & 0x80000000u) ? (t & 0x7FFFFFFFu) : ~t);
    if (v == v) tau0[q] = fminf(tau0[q], v);
  }
}

// ---- the filter: lane = row, 16*NQG queries per workgroup, NADD entries summed per byte -------
// MAIN only tags the instantiation used for the last (large) stage, so that profilers list it apart
// from the short first stage.
#ifdef GULON_FILTER_STAMPS
__global__ void dummy_kernel(const int *p) { if (p == nullptr) __builtin_trap(); }
// experiment (timing only, results are stale): GULON_SKIP = 1 bound_tables | 2 qt_quantize | 4 survivors | 8 fallback + merges | 16 main stage | 32 short first stage
static std::atomic<int> skip_calls{0};   // batches seen; the mask applies from batch GULON_SKIP_AFTER on (default 16)
static int skip_mask() {
  static const int m = getenv("GULON_SKIP") ? atoi(getenv("GULON_SKIP")) : 0;
  static const int after = getenv("GULON_SKIP_AFTER") ? atoi(getenv("GULON_SKIP_AFTER")) : 16;
  return skip_calls.load() > after ? m : 0;
}
#define GULON_SKIPPED(bit) (skip_mask() & (bit))
#else
#define GULON_SKIPPED(bit) false
#endif
template <int QW> struct QEntry;                     // QW queries (one byte each) per table entry
template <> struct QEntry<16> { using type = uint4; };
template <> struct QEntry<8> { using type = uint2; };
template <> struct QEntry<4> { using type = uint32_t; };

template <int QW, int NQG, int VEC, int NADD, int MAIN,
          int NG1 /* 1: a single code word per row (ng == 1); 2: and `codes` is the conflict-ordered copy, `perm` its row order */>
__global__ __launch_bounds__(FILTER_THREADS) void filter_kernel(
    const uint8_t *__restrict__ codes, const uint8_t *__restrict__ perm, int ng, int m_pad, const uint8_t *__restrict__ qtab, int row_from, int row_until,
    int rb_begin, int e_count, int e_per_chunk, RbMap mp, int *__restrict__ cnt, int *__restrict__ queue, int cap /* entries per sub-queue */,
    const int *__restrict__ fb_tile, int qt, int B,
    int pwin /* ordered copy: blocks per ordering window - 1 (0 or 3) -- `perm` = a row's place in its window */) {
  constexpr int NW = FILTER_THREADS / 64;
  constexpr uint32_t QMAXP = (255u / NADD) * 0x00010001u;   // QMAX in both halves; survive <=> sum <= QMAX - 1
  constexpr int DW = QW / 4;     // dwords per entry
  // The kernel runs at the LDS random-gather floor, and the vector L1 is a second gather
  // pipe that sits idle: the entries of the last GLB quantizers of a code word are read from the quantized
  // table in global memory (4 KiB per tile and quantizer: L1-resident) instead of its LDS copy, requested at
  // the top of the row block so that their latency passes under the LDS gathers.  Measured on the 10 M-row
  // bench index (two batches in flight / main stage alone): 3.44 / 3.00 ms with every entry from LDS,
  // 3.06 / 2.66 with 2 from L1, 3.04 / 2.57 with 3, 3.06 / 2.66 with 4 (requested where they are used instead
  // of up front: 3.21, 3.23, 3.36; five 3.96, eight 5.58 -- the L1 path saturates quickly).  It pays only where
  // LDS is the one busy pipe: 16-byte entries, four entries summed per widening, two workgroups per CU (m <= 16);
  // with 4-byte code words, wider indexes (m = 32, 64, 100) or the 7-bit levels it measured 2-50 % slower.
  constexpr int GLB = (QW == 16 && NQG == 1 && VEC == 16 && (NADD == 4 || NADD == 8) && NG1) ? GULON_FILTER_GLB : 0;
  using Word = typename CodeWord<VEC>::type;
  using QE = typename QEntry<QW>::type;
  extern __shared__ uint4 qlds_raw[];
  QE *qlds = reinterpret_cast<QE *>(qlds_raw);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // (MAIN == 2: chunks along x -- consecutive workgroups go to different XCDs, and with the tiles served in the order of
  // their row limits every XCD then gets the same share of every tile)
  const int tile = MAIN == 2 ? blockIdx.y : blockIdx.x, chunk = MAIN == 2 ? blockIdx.x : blockIdx.y;
  const int ntile = MAIN == 2 ? gridDim.y : gridDim.x;
  const int tab = m_pad * 256;   // entries per QW-query group
#ifdef GULON_FILTER_STAMPS
  unsigned long long stamp0 = 0, stamp1 = 0;
  if constexpr (MAIN == 1) stamp0 = __builtin_amdgcn_s_memrealtime();
#endif
  {   // every query of this tile already goes to the exact scan: nothing to do here
    const int q_lo = tile * NQG * QW, q_hi = min(B, q_lo + NQG * QW);
    bool any_live = false;
    if (q_hi > q_lo)
      for (int t = q_lo / qt; t <= (q_hi - 1) / qt; t++) any_live = any_live || fb_tile[t] == 0;
    if (!any_live) return;
  }
  // MAIN == 2 (the tie replay's long level, one flag per query, every row block of the range): behind the flags sits
  // one number per query tile, the last row block any of its queries can still take a row from (replay.hip)
  int e_limit = e_count;
  if constexpr (MAIN == 2) {
    const int nflags = (ntile * NQG * QW + 15) & ~15;
    e_limit = min(e_count, (fb_tile[nflags + tile] | pwin) - rb_begin + 1);   // (a row sits somewhere in its window)
    if (chunk * e_per_chunk >= e_limit) return;
  }
  // spreads the queue-tail atomics of one query over NSLOT counters
  const int slot = (chunk * NW + wave) & (NSLOT - 1);
  {
    const int n16 = NQG * tab * QW / 16;   // 16-byte units (tab * QW is a multiple of 16)
    const uint4 *src = reinterpret_cast<const uint4 *>(qtab) + (size_t)tile * n16;
    for (int e = tid; e < n16; e += FILTER_THREADS) qlds_raw[e] = src[e];
    if (tid == 0) *reinterpret_cast<int *>(qlds_raw + n16) = NW;   // run counter: the first NW runs are the waves' own
  }
  __syncthreads();
#ifdef GULON_FILTER_STAMPS
  if constexpr (MAIN == 1) stamp1 = __builtin_amdgcn_s_memrealtime();
#endif

  const int e0 = chunk * e_per_chunk;
  const int e1 = min(e_limit, e0 + e_per_chunk);
  const Word *cw = reinterpret_cast<const Word *>(codes);
  auto block_of = [&](int p, int r) { return rb_begin + p * mp.period + mp.lo + r; };
  // The chunk's row blocks are handed to the waves in RUNS through a counter in LDS, not dealt out in a fixed
  // stride.  The LDS arbiter serves older waves first, so with a fixed share each the sixteen waves of a workgroup
  // finished in four groups ~12 us apart (1.25 M-row shard, per-wave stamps: first wave 77 us, last 113 us), the
  // workgroup's slot -- 16 wave slots and 64 KiB of LDS, which the next workgroup needs all at once -- stayed
  // occupied by a thinning set of waves for a third of its life, and with both workgroups of a CU in that state the
  // gather pipe ran dry (53 % of the chip's wave slots active at the troughs).  Drawing runs, fast waves take more
  // blocks and all sixteen end within one run of each other.  The next run is drawn a whole run ahead (the result of
  // the ds_add_rtn is read four row blocks later), so the code word of the next block -- the next of this run or
  // the first of the next -- is still requested one block ahead.
  constexpr int RUN = 4;
  const int nruns = (e1 - e0 + RUN - 1) / RUN;
  // block list position e -> (period p, offset r) = (e / width, e % width), once per run: multiply-high by
  // floor((2^32 - 1) / width) + 1, exact for e < 2^32 / width
  const uint32_t magic = mp.width > 1 ? 0xFFFFFFFFu / (uint32_t)mp.width + 1u : 0u;
  auto locate = [&](int e, int &p, int &r) {
    p = mp.width > 1 ? (int)__umulhi((uint32_t)e, magic) : e;
    r = e - p * mp.width;
  };
  int *run_ctr = reinterpret_cast<int *>(qlds_raw + NQG * tab * QW / 16);   // = NW after the staging barrier
  int run = wave, e = e0 + run * RUN, e_run_end = min(e1, e + RUN);
  int mp_p = 0, mp_r = 0, grab_v = 0;
  Word w_first{};
  if (run < nruns) {
    locate(e, mp_p, mp_r);
    if (lane == 0) grab_v = atomicAdd(run_ctr, 1);
    w_first = cw[((size_t)block_of(mp_p, mp_r) * ng) * 64 + lane];
  }
  while (run < nruns) {
    const int rb = block_of(mp_p, mp_r);
    // the block after this one
    int e_n = e + 1, p_n = mp_p, r_n = mp_r + 1, run_n = run, end_n = e_run_end;
    if (r_n == mp.width) { r_n = 0; p_n++; }
    if (e_n == e_run_end) {
      run_n = __builtin_amdgcn_readfirstlane(grab_v);
      if (run_n < nruns) {
        e_n = e0 + run_n * RUN;
        end_n = min(e1, e_n + RUN);
        locate(e_n, p_n, r_n);
        if (lane == 0) grab_v = atomicAdd(run_ctr, 1);
      }
    }
    const Word *p = cw + ((size_t)rb * ng) * 64 + lane;
    Word w = w_first;
    // (unconditional: the last iteration re-reads its own block -- a conditional load costs a copy of the word back)
    w_first = cw[((size_t)(run_n < nruns ? block_of(p_n, r_n) : rb) * ng) * 64 + lane];
    run = run_n; e = e_n; e_run_end = end_n; mp_p = p_n; mp_r = r_n;

    // acc[s][2*dd+1] : 16-bit sums of queries 4dd+1 (low half) and 4dd+3 (high half) of group s
    // acc[s][2*dd]   : queries 4dd and 4dd+2 -- after the words.  While the words are walked it is the plain 32-bit sum W
    //   of the byte-sum dwords: with S_i the sum of byte i, W = S0 + 2^8 S1 + 2^16 S2 + 2^24 S3 (mod 2^32) and the odd
    //   accumulator is O = S1 + 2^16 S3 exactly, so W - 2^8 O = S0 + 2^16 S2 (mod 2^32, and that is below 2^32: every
    //   S_i < 2^16) -- one subtraction per dword after the last word instead of a mask per dword and widening
    uint32_t acc[NQG][2 * DW];
#pragma unroll
    for (int s = 0; s < NQG; s++)
#pragma unroll
      for (int x = 0; x < 2 * DW; x++) acc[s][x] = 0;

    // one code word (VEC quantizers) of the row block; G0: the index has a single word per row (m <= 16), so the
    // table offsets are compile-time constants -- an LDS address is then one SDWA shift of the code byte (the entry's
    // quantizer rides in the instruction's offset field) instead of an extraction and a shift-add
    // (the word by value in, the next one out: a captured `w` that the body overwrites ended up in scratch memory for
    // every multi-word form -- 32 bytes per lane, the prefetched word waited for at once: m = 64 ran 13 % slower)
    auto word = [&](const int g, auto G0, const Word w) __attribute__((always_inline)) -> Word {
      Word wn = w;
      if (!decltype(G0)::value && g + 1 < ng) wn = p[(size_t)(g + 1) * 64];
      const QE *tj_lds = decltype(G0)::value ? qlds : qlds + g * VEC * 256;
      // the L1-served entries are the LAST GLB of the word, requested first: their latency passes under the
      // LDS gathers of the other entries (byte sums commute).  Uniform base (SGPR pair) + 32-bit lane offset: one
      // instruction per address instead of a 64-bit add chain
      const uint8_t *tj_glb = qtab + ((size_t)tile * NQG * tab + (decltype(G0)::value ? 0 : g * VEC * 256)) * sizeof(QE);
      QE gl[GLB > 0 ? GLB : 1][NQG];
#pragma unroll
      for (int a = 0; a < GLB; a++)
#pragma unroll
        for (int s = 0; s < NQG; s++) {
          const uint8_t *base = tj_glb + ((size_t)(VEC - GLB + a) * 256 + (size_t)s * tab) * sizeof(QE);
          const uint32_t off = code_byte<VEC>(w, VEC - GLB + a) * (uint32_t)sizeof(QE);
          gl[a][s] = *reinterpret_cast<const QE *>(base + off);
        }
#pragma unroll
      for (int b = 0; b < VEC; b += NADD) {
        uint32_t c[NADD];
#pragma unroll
        for (int a = 0; a < NADD; a++) c[a] = code_byte<VEC>(w, b + a);
#pragma unroll
        for (int s = 0; s < NQG; s++) {
          uint32_t xs[DW];
#pragma unroll
          for (int dd = 0; dd < DW; dd++) xs[dd] = 0;
#pragma unroll
          for (int a = 0; a < NADD; a++) {   // bytes cannot carry: NADD * QMAX <= 255
            const int e = b + a;
            if (e >= VEC - GLB) {
#pragma unroll
              for (int dd = 0; dd < DW; dd++) xs[dd] += reinterpret_cast<const uint32_t *>(&gl[e - (VEC - GLB)][s])[dd];
            } else if constexpr (decltype(G0)::value && QW == 16) {
              // the tables sit at LDS address 0 (the kernel has no static LDS; checked by the host at launch): with
              // a literal base the whole address is  (code byte << 4) + constant  -- one SDWA shift
              typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
              const u32x4 y = reinterpret_cast<const __attribute__((address_space(3))) u32x4 *>(0)[e * 256 + c[a] + s * tab];
#pragma unroll
              for (int dd = 0; dd < DW; dd++) xs[dd] += y[dd];
            } else {
              const QE y = tj_lds[e * 256 + c[a] + s * tab];
#pragma unroll
              for (int dd = 0; dd < DW; dd++) xs[dd] += reinterpret_cast<const uint32_t *>(&y)[dd];
            }
          }
#pragma unroll
          for (int dd = 0; dd < DW; dd++) {
#ifdef GULON_FILTER_MASKED_EVEN
            acc[s][2 * dd] += xs[dd] & 0x00FF00FFu;
#else
            acc[s][2 * dd] += xs[dd];                                                // (unmasked: see `acc`)
#endif
            acc[s][2 * dd + 1] += __builtin_amdgcn_perm(0u, xs[dd], 0x0C030C01u);   // bytes 1 and 3
          }
        }
      }
      return wn;
    };
    // the same with the word's number a compile-time constant: table addresses are  (code byte << log2 entry size) +
    // literal, as in the one-word form (the tables sit at LDS address 0); taken where the index has exactly NGC words
    auto word_c = [&](auto gc, const Word w) __attribute__((always_inline)) -> Word {
      constexpr int g = decltype(gc)::value;
      Word wn = w;
      if (g + 1 < ng) wn = p[(size_t)(g + 1) * 64];
      typedef unsigned lvec __attribute__((ext_vector_type(DW)));
#pragma unroll
      for (int b = 0; b < VEC; b += NADD) {
        uint32_t c[NADD];
#pragma unroll
        for (int a = 0; a < NADD; a++) c[a] = code_byte<VEC>(w, b + a);
#pragma unroll
        for (int s = 0; s < NQG; s++) {
          uint32_t xs[DW];
#pragma unroll
          for (int dd = 0; dd < DW; dd++) xs[dd] = 0;
#pragma unroll
          for (int a = 0; a < NADD; a++) {   // bytes cannot carry: NADD * QMAX <= 255
            const int e = g * VEC + b + a;
            const lvec y = reinterpret_cast<const __attribute__((address_space(3))) lvec *>(0)[e * 256 + c[a] + s * tab];
#pragma unroll
            for (int dd = 0; dd < DW; dd++) xs[dd] += y[dd];
          }
#pragma unroll
          for (int dd = 0; dd < DW; dd++) {
#ifdef GULON_FILTER_MASKED_EVEN
            acc[s][2 * dd] += xs[dd] & 0x00FF00FFu;
#else
            acc[s][2 * dd] += xs[dd];                                                // (unmasked: see `acc`)
#endif
            acc[s][2 * dd + 1] += __builtin_amdgcn_perm(0u, xs[dd], 0x0C030C01u);   // bytes 1 and 3
          }
        }
      }
      return wn;
    };
    if constexpr (NG1) (void)word(0, std::true_type{}, w);
#ifndef GULON_FILTER_WORDS_LOOP
#define WC(G) w = word_c(std::integral_constant<int, G>{}, w)
    // (the two forms measured: BASELINE config 5's m = 64 -- 15.16 -> 14.5 ms per batch -- and the reference CLI's default
    // m = 25 -- 0.759 -> 0.716 ms.  Unrolled for every word count the kernels grew to the 128-register limit, and the
    // m = 32 form (two 16-byte words, 16-byte entries) aborted; the others keep the loop.)
    else if (QW == 8 && VEC == 16 && ng == 4) { WC(0); WC(1); WC(2); WC(3); }
    else if (QW == 16 && VEC == 4 && ng == 7) { WC(0); WC(1); WC(2); WC(3); WC(4); WC(5); WC(6); }
#undef WC
#endif
    else
      for (int g = 0; g < ng; g++) w = word(g, std::false_type{}, w);

#ifndef GULON_FILTER_MASKED_EVEN
#pragma unroll
    for (int s = 0; s < NQG; s++)
#pragma unroll
      for (int dd = 0; dd < DW; dd++) acc[s][2 * dd] -= acc[s][2 * dd + 1] << 8;
#endif
    // conflict-ordered copy: which row a lane holds is only looked up (one byte) when a lane has something to report
    constexpr bool PERM = NG1 == 2;
    int row = rb * 64 + lane;
    bool valid = PERM || (row >= row_from && row < row_until);
    uint32_t any = 0;
    uint32_t left[NQG][2 * DW];
#pragma unroll
    for (int s = 0; s < NQG; s++)
#pragma unroll
      for (int x = 0; x < 2 * DW; x++) {
        left[s][x] = pk_sub_sat_u16(QMAXP, acc[s][x]);   // non-zero half <=> that query keeps this row
        any |= left[s][x];
      }
    if (__ballot(valid && any != 0) != 0ull) {
      // rare path.  Its queue addresses are uniform and loop-invariant, and hoisted out of the scan loop
      // they would hold ~60 SGPRs for the whole kernel -- past the 102 that still allow two workgroups per
      // CU.  Opaque copies keep that arithmetic in here.
      int *cnt_l = cnt, *queue_l = queue, *fb_l = const_cast<int *>(fb_tile);
      int slot_l = slot, tile_l = tile;
      asm volatile("" : "+s"(cnt_l), "+s"(queue_l), "+s"(slot_l), "+s"(tile_l), "+s"(fb_l));
      if (PERM) {
        const uint8_t *perm_l = perm;
        int pwin_l = pwin;
        asm volatile("" : "+s"(perm_l), "+s"(pwin_l));
        row = (rb & ~pwin_l) * 64 + perm_l[(size_t)rb * 64 + lane];
        valid = row >= row_from && row < row_until;
      }
#pragma unroll
      for (int s = 0; s < NQG; s++)
#pragma unroll
        for (int x = 0; x < 2 * DW; x++) {
          const uint32_t l = valid ? left[s][x] : 0u;
          if (__ballot(l != 0) == 0ull) continue;
          const int q0 = (tile_l * NQG + s) * QW + 4 * (x >> 1) + (x & 1);
          // a full sub-queue means the bound separates nothing for this query: flag its tile at once, so that
          // the workgroups of the tile that have not started yet return immediately (the exact scan redoes it)
          if (l & 0xFFFFu) {
            const int sq = q0 * NSLOT + slot_l;
            const int pos = atomicAdd(&cnt_l[sq], 1);
            if (pos < cap) queue_l[(size_t)sq * cap + pos] = row;
            else fb_l[q0 / qt] = 1;
          }
          if (l >> 16) {
            const int sq = (q0 + 2) * NSLOT + slot_l;
            const int pos = atomicAdd(&cnt_l[sq], 1);
            if (pos < cap) queue_l[(size_t)sq * cap + pos] = row;
            else fb_l[(q0 + 2) / qt] = 1;
          }
        }
    }
  }
#ifdef GULON_FILTER_STAMPS
  if constexpr (MAIN == 1) {   // per workgroup: [wave][start, staged, end] + [48] = (XCC_ID << 32) | HW_ID of wave 0
    if (lane == 0 && g_filter_stamps) {
      unsigned long long *o = g_filter_stamps + 52 * ((size_t)blockIdx.y * gridDim.x + blockIdx.x);
      o[3 * wave] = stamp0; o[3 * wave + 1] = stamp1; o[3 * wave + 2] = __builtin_amdgcn_s_memrealtime();
      if (wave == 0)
        o[48] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4);
    }
  }
#endif
}

// ---- exact re-evaluation of the survivors; one workgroup (4 waves) per query -------------------
// Wave 0 continues the query's running (K+1)-list; waves 1-3 collect into empty lists but only
// what beats the running list's last entry; the four sorted lists are merged through LDS at the end.
constexpr int SV_WAVES = 4;
// One workgroup = the W queries that share a W-interleaved fp32 table (scan.hip build_tables), SV_WAVES waves each:
// the table (m_pad x 256 x W floats, 64 KiB at m = 16) is copied to LDS once, coalesced, and the sixteen look-ups of
// every survivor are LDS gathers.  (One workgroup per query gathering from the table in global memory sent 17
// scattered 64-byte requests per survivor through the texture path -- 4.4 M per batch -- which is also the path the
// main-stage kernel of the next batch serves three of its sixteen look-ups from.)
template <int VEC>
__global__ __launch_bounds__(64 * SV_WAVES * 4) void survivors_kernel(
    const uint8_t *__restrict__ codes, int ng, int m_pad, const float *__restrict__ tables, int W, int row_base,
    int *__restrict__ cnt, const int *__restrict__ queue, int cap, int B, int keff, float *__restrict__ fin_v,
    int *__restrict__ fin_i, int *__restrict__ fb_tile, int qt, int give_up /* survivors beyond which the filter is abandoned; 0: never */) {
  using Word = typename CodeWord<VEC>::type;
  extern __shared__ uint4 sv_lds_raw[];
  float *tlds = reinterpret_cast<float *>(sv_lds_raw);
  __shared__ unsigned long long mk[4][(SV_WAVES - 1) * 64];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave_wg = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int qslot = wave_wg / SV_WAVES, wave = wave_wg - qslot * SV_WAVES;   // query of the group, wave of the query
  const int q = blockIdx.x * W + qslot;
  {
    const int n16 = m_pad * 256 * W / 4;      // 16-byte units (W * 256 floats per quantizer: a multiple of 4)
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 *src = reinterpret_cast<const u32x4 *>(tables) + (size_t)blockIdx.x * n16;
    u32x4 *dst = reinterpret_cast<u32x4 *>(sv_lds_raw);
    if (blockIdx.x * W < B)      // (read once: past the vector L1 the main-stage kernels gather from)
      for (int e = tid; e < n16; e += 64 * SV_WAVES * W) dst[e] = __builtin_nontemporal_load(src + e);
  }
  // sub-queue fill levels -> exclusive offsets of a flat numbering of this query's survivors
  int mine = lane < NSLOT ? cnt[q * NSLOT + lane] : 0;
  __syncthreads();                                  // every wave has read the counters; the table is in LDS
  if (wave == 0 && lane < NSLOT) cnt[q * NSLOT + lane] = 0;
  bool active = q < B;                              // (uniform per query; no early return: barriers below)
  if (active && __ballot(mine > cap) != 0ull) {
    if (wave == 0 && lane == 0) fb_tile[q / qt] = 1;   // a sub-queue overflowed: the exact scan redoes this query tile
    mine = min(mine, cap);
  }
  int incl = mine;
#pragma unroll
  for (int o = 1; o < NSLOT; o <<= 1) {
    const int up = __shfl_up(incl, o);
    if (lane >= o) incl += up;
  }
  int n = readlane_i(incl, NSLOT - 1);
  if (active && give_up > 0 && n > give_up) {
    // the bound lets too many rows through for this query (data without a tail of near rows):
    // filtering the rest would cost more than it saves -- its tile goes to the exact scan
    if (wave == 0 && lane == 0) fb_tile[q / qt] = 1;
    active = false;
  }
  if (!active) n = 0;
  int start[NSLOT];
#pragma unroll
  for (int sl = 0; sl < NSLOT; sl++) start[sl] = readlane_i(incl - mine, sl);
  auto entry = [&](int e) {   // e-th survivor of the query, e < n
    int sl = 0;
#pragma unroll
    for (int x = 1; x < NSLOT; x++) sl += e >= start[x];
    int off = start[0];
#pragma unroll
    for (int x = 1; x < NSLOT; x++) off = sl == x ? start[x] : off;
    return queue[((size_t)q * NSLOT + sl) * cap + (e - off)];
  };

  // The running list and every candidate as ONE 64-bit key, (distance bits << 32) | row id: distances are sums of
  // squares (>= +0, or NaN, which sorts behind +inf and never enters), so the unsigned order of the key is the
  // (distance, row id) order of the lists.  A wave sorts the 64 candidates of a step with a bitonic network and merges
  // them into its running 64-list (the keff smallest are what counts): ~300 instructions per step.  (Feeding the
  // candidates one by one into a sorted register list -- ballot, readlane, shift, ~40 instructions each, every
  // candidate while a list fills up -- made this small kernel cost the batch 55 us next to another batch's main
  // stage, which leaves no vector-ALU slots to spare.)
  constexpr unsigned long long KEY_PAD = 0x7F8000007FFFFFFFull;      // (+inf, INT_MAX)
  auto make_key = [](float v, int r) { return ((unsigned long long)__float_as_uint(v) << 32) | (unsigned)r; };
  unsigned long long run = KEY_PAD;
  if (active && lane < keff) run = make_key(fin_v[(size_t)q * keff + lane], fin_i[(size_t)q * keff + lane]);
  const unsigned long long bound_key = shfl_u64(run, keff - 1);     // what a candidate has to beat
  if (wave != 0) run = KEY_PAD;                                      // wave 0 continues the list, the others start empty
  // W-interleaved fp32 tables (the group's copy in LDS): entry (j, c) of query q at (j * 256 + c) * W + q%W
  const float *tq = tlds + qslot;
  const Word *cw = reinterpret_cast<const Word *>(codes);
  // survivors e = (SV_WAVES * it + wave) * 64 + lane; two-deep software pipeline: row ids two
  // batches ahead, first code word one batch ahead
  constexpr int STEP = 64 * SV_WAVES;
  const int e0 = wave * 64 + lane;
  int row_n = e0 < n ? entry(e0) : 0;
  Word w_n = cw[((size_t)(row_n >> 6) * ng) * 64 + (row_n & 63)];
  int row_nn = e0 + STEP < n ? entry(e0 + STEP) : 0;
  unsigned long long admit = bound_key;                               // tightens to this wave's keff-th key
  for (int e = e0; e - lane < n; e += STEP) {
    const bool have = e < n;
    const int row = row_n;
    const Word w0 = w_n;
    row_n = row_nn;
    w_n = cw[((size_t)(row_n >> 6) * ng) * 64 + (row_n & 63)];
    row_nn = e + 2 * STEP < n ? entry(e + 2 * STEP) : 0;
    float d = 0.f;                      // the reference's order: j ascending, unfused fp32
    for (int g = 0; g < ng; g++) {
      const Word w = g == 0 ? w0 : cw[((size_t)(row >> 6) * ng + g) * 64 + (row & 63)];
      float t[VEC];
#pragma unroll
      for (int b = 0; b < VEC; b++) t[b] = tq[((size_t)(g * VEC + b) * 256 + code_byte<VEC>(w, b)) * W];
#pragma unroll
      for (int b = 0; b < VEC; b++) d += t[b];
    }
    unsigned long long key = make_key(d, row + row_base);
    if (!(have && key < admit)) key = KEY_PAD;
    if (__ballot(key != KEY_PAD) == 0ull) continue;
    run = merge64_u64(run, sort64_u64(key, lane), lane);
    admit = shfl_u64(run, keff - 1);
    if (admit > bound_key) admit = bound_key;
  }
  if (wave > 0) mk[qslot][(wave - 1) * 64 + lane] = run;
  __syncthreads();
  if (wave == 0 && active) {
#pragma unroll 1
    for (int w2 = 0; w2 < SV_WAVES - 1; w2++) run = merge64_u64(run, mk[qslot][w2 * 64 + lane], lane);
    if (lane < keff) {
      fin_v[(size_t)q * keff + lane] = __uint_as_float((unsigned)(run >> 32));
      fin_i[(size_t)q * keff + lane] = (int)(unsigned)run;
    }
  }
}

template <int QW, int NQG, int VEC, int NADD>
void launch_filter_t(gulon_index *ix, int ftiles, int nchunks, int rb_begin, int e_count, int e_per_chunk, RbMap mp,
                     int from, int until, int cap, int stage, int B, hipStream_t st, int *fb, int qt) {
  const int W_fp32 = ix->w;
  // (GULON_FILTER_LDS_PAD: experiment knob -- bytes of LDS a main-stage workgroup claims on top of its tables, so
  // that fewer of them fit a CU)
  static const size_t lds_pad = getenv("GULON_FILTER_LDS_PAD") ? (size_t)atoi(getenv("GULON_FILTER_LDS_PAD")) : 0;
  const size_t lds_bytes = (size_t)NQG * ix->m_pad * 256 * QW + 16 + (stage == 1 ? lds_pad : 0);   // tables + the run counter
  // (the single-word form only for the instantiation the headline index runs on: m = 16, two workgroups per CU)
  constexpr bool one_word_form = QW == 16 && NQG == 1 && VEC == 16 && (NADD == 4 || NADD == 8);
  // stage: 0 a short stage, 1 the main stage (a tag of its own for the profilers), 2 the tie replay's long level
  // (per-tile row limits behind the flags)
#define GULON_FILTER_PICK(NG1_) \
  (stage == 1 ? filter_kernel<QW, NQG, VEC, NADD, 1, NG1_> : stage == 2 ? filter_kernel<QW, NQG, VEC, NADD, 2, NG1_> \
                                                                         : filter_kernel<QW, NQG, VEC, NADD, 0, NG1_>)
  auto kern = GULON_FILTER_PICK(0);
  const uint8_t *codes = ix->codes.p, *perm = nullptr;
  if (one_word_form && ix->ng == 1) {
    kern = GULON_FILTER_PICK(one_word_form);
    if (ix->fcodes.p && tuning_of(ix).filter_order > 0) {   // the conflict-ordered copy of the codes (conflict_order.hip)
      kern = GULON_FILTER_PICK(2 * one_word_form);
      codes = ix->fcodes.p;
      perm = ix->fperm.p;
    }
    hipFuncAttributes fa;
    HIP_CHECK(hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(kern)));
    GULON_REQUIRE(fa.sharedSizeBytes == 0, "internal: the single-word filter kernel addresses its tables from LDS offset 0");
  }
  HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds_bytes));
#ifdef GULON_FILTER_STAMPS
  static unsigned long long *stamps_d = nullptr;
  static size_t stamps_n = 0;
  if (stage == 1 && getenv("GULON_FILTER_STAMPS")) {
    const size_t need = (size_t)52 * ftiles * nchunks;
    if (need > stamps_n) {
      if (stamps_d) (void)hipFree(stamps_d);
      HIP_CHECK(hipMalloc((void **)&stamps_d, need * 8));
      stamps_n = need;
      HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_filter_stamps), &stamps_d, sizeof(stamps_d)));
    }
  }
#endif
  hipLaunchKernelGGL(kern, stage == 2 ? dim3(nchunks, ftiles) : dim3(ftiles, nchunks), dim3(FILTER_THREADS), lds_bytes, st, codes, perm, ix->ng, ix->m_pad,
                     ix->qtab.p, from, until, rb_begin, e_count, e_per_chunk, mp, ix->sv_cnt.p, ix->sv_queue.p, cap,
                     fb ? fb : ix->fb_tile.p, fb ? qt : W_fp32 * ix->nsub, B, perm ? ix->fwindow - 1 : 0);
  HIP_CHECK(hipGetLastError());
#ifdef GULON_FILTER_STAMPS
  if (stage == 1 && stamps_d && getenv("GULON_FILTER_STAMPS")) {
    HIP_CHECK(hipStreamSynchronize(st));
    std::vector<unsigned long long> h((size_t)52 * ftiles * nchunks);
    HIP_CHECK(hipMemcpy(h.data(), stamps_d, h.size() * 8, hipMemcpyDeviceToHost));
    if (FILE *f = fopen(getenv("GULON_FILTER_STAMPS"), "wb")) {
      const int hdr[4] = {ftiles, nchunks, e_count, e_per_chunk};
      fwrite(hdr, sizeof(int), 4, f);
      fwrite(h.data(), 8, h.size(), f);
      fclose(f);
    }
  }
#endif
}

// The filter kernel owns the LDS of every CU, so two of them (query batches in flight on
// different streams) cannot usefully overlap -- they would only time-slice and blur each other's
// duration.  Launches are therefore chained through one event per device: a batch's small,
// latency-bound kernels overlap with another batch's filter, the filters themselves run in turn.
struct FilterLane {   // per device: completion of the most recent main-stage launch (created on first use, never destroyed)
  static constexpr int MAX_DEVICES = 64;
  std::atomic<hipEvent_t> ev[MAX_DEVICES];
  FilterLane() { for (auto &e : ev) e.store(nullptr, std::memory_order_relaxed); }
  // the device's event; `fresh` = this call created it (nothing to wait for yet).  Lock-free after the first launch
  // on a device: the launch path of every batch of every index goes through here.
  hipEvent_t of(int dev, bool &fresh) {
    fresh = false;
    GULON_REQUIRE(dev >= 0 && dev < MAX_DEVICES, "device ordinal %d out of range", dev);
    hipEvent_t e = ev[dev].load(std::memory_order_acquire);
    if (e) return e;
    hipEvent_t mine = nullptr;
    HIP_CHECK(hipEventCreateWithFlags(&mine, hipEventDisableTiming));
    if (ev[dev].compare_exchange_strong(e, mine, std::memory_order_acq_rel)) { fresh = true; return mine; }
    (void)hipEventDestroy(mine);            // another thread was first
    return e;
  }
};
FilterLane &filter_lane() { static FilterLane l; return l; }

int device_cus() {   // compute units of the current device (cached per ordinal, lock-free)
  static std::atomic<int> cus[FilterLane::MAX_DEVICES];
  int dev = 0;
  HIP_CHECK(hipGetDevice(&dev));
  GULON_REQUIRE(dev >= 0 && dev < FilterLane::MAX_DEVICES, "device ordinal %d out of range", dev);
  int n = cus[dev].load(std::memory_order_relaxed);
  if (n == 0) {
    HIP_CHECK(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev));
    n = std::max(1, n);
    cus[dev].store(n, std::memory_order_relaxed);
  }
  return n;
}

// queries per quantized table entry: as many as LDS holds for one group (16 up to m_pad = 36,
// 8 up to 72, 4 up to 144) ...
int filter_qw(const gulon_index *ix) {
  int qw = 16;
  while (qw > 4 && (size_t)ix->m_pad * 256 * qw > FILTER_LDS_BUDGET) qw /= 2;
  return qw;
}
// ... and how many such groups a workgroup takes: two only if two workgroups of that size still
// fit a CU's 160 KiB -- two resident workgroups hide each other's table staging and barriers
// (measured at m = 16: 3.18 -> 3.01 ms for the main stage against one workgroup with two groups)
int filter_nqg(const gulon_index *ix) {
  if (const char *e = getenv("GULON_FILTER_NQG")) { int v = atoi(e); if (v == 1 || v == 2) return v; }   // experiment knob
  return (size_t)ix->m_pad * 256 * filter_qw(ix) * 4 <= 160 * 1024 ? 2 : 1;
}

void launch_filter(gulon_index *ix, int qw, int nqg, int nadd, int ftiles, int nchunks, int rb_begin, int e_count,
                   int e_per_chunk, RbMap mp, int from, int until, int cap, int stage, int B, hipStream_t st,
                   int *fb = nullptr /* tile flags other than the index's own, one per `qt` queries */, int qt = 1) {
#define GO(W_, Q, V, A) \
  launch_filter_t<W_, Q, V, A>(ix, ftiles, nchunks, rb_begin, e_count, e_per_chunk, mp, from, until, cap, stage, B, st, fb, qt)
#ifdef GULON_FILTER_NADD8   // experiment builds: 5-bit levels, eight entries per widening (one-word form only)
#define GO_QV(W_, Q, V) do { if (nadd == 8 && W_ == 16 && Q == 1 && V == 16) GO(16, 1, 16, 8); else if (nadd == 4) GO(W_, Q, V, 4); else GO(W_, Q, V, 2); } while (0)
#else
#define GO_QV(W_, Q, V) do { if (nadd == 4) GO(W_, Q, V, 4); else GO(W_, Q, V, 2); } while (0)
#endif
#define GO_W(W_) do {                                                  \
    if (ix->vec == 16) { if (nqg == 2) GO_QV(W_, 2, 16); else GO_QV(W_, 1, 16); } \
    else               { if (nqg == 2) GO_QV(W_, 2, 4); else GO_QV(W_, 1, 4); }   \
  } while (0)
  if (qw == 16) GO_W(16); else if (qw == 8) GO_W(8); else GO_W(4);
#undef GO_W
#undef GO_QV
#undef GO
}

// ---- level 2 of the tie replay through the filter (replay_level2_filtered below) ---------------------------------
// One workgroup: the flagged queries in the order the long level serves them.  Compact position c -> slot order[c]:
// the live ones (finite K-th distance over the earlier rows, not served by rp_shortcut, enough flagged queries for a
// launch) first, by the last row that can still insert (`rlast`, replay.hip) -- a query tile then walks the rows up to
// ITS largest such row only, and tiles of queries that need no rows at all are skipped -- then the ones left to the
// segment scan (fb = 1) and the served ones (fb = 2).  tau / fb / fin in compact order; lim[tile] = last row block.
__global__ __launch_bounds__(1024) void rp_filter_order(const int *__restrict__ count, int F, int min_flagged, int K,
                                                        const float *__restrict__ prefix_v, const int *__restrict__ prefix_c,
                                                        const int *__restrict__ done, const int *__restrict__ rlast,
                                                        int row_base, int n_pad, int tile_q, int ntiles, int *__restrict__ order,
                                                        float *__restrict__ tau, int *__restrict__ fb,
                                                        float *__restrict__ fin_v, int *__restrict__ fin_i,
                                                        int *__restrict__ lim /* [n_pad / tile_q] */) {
  __shared__ unsigned long long key[1024];
  const int t = threadIdx.x;
  const int nf = min(*count, F);
  {
    // class 0: live, by last row block; 1: left to the scan; 2: served; 3: not a flagged query
    unsigned long long cls = 3, sub = 0;
    if (t < nf) {
      const bool served = done && done[t] != 0;
      const bool live = !served && nf >= min_flagged && prefix_c[t] >= K && prefix_v[(size_t)t * K + K - 1] < INFINITY;
      cls = live ? 0 : served ? 2 : 1;
      if (live) {
        const int last = rlast ? rlast[t] : INT_MAX;
        sub = last == INT_MAX ? 0x3FFFFFFFull : (unsigned long long)max(0, (last - row_base) >> 6);   // (no result at hand: every row)
      }
    }
    key[t] = (cls << 60) | (sub << 12) | (unsigned long long)t;
  }
  __syncthreads();
  for (int k2 = 2; k2 <= 1024; k2 <<= 1)
    for (int j2 = k2 >> 1; j2 > 0; j2 >>= 1) {
      const int o = t ^ j2;
      if (o > t) {
        const unsigned long long a = key[t], b = key[o];
        const bool up = (t & k2) == 0;
        if ((a > b) == up) { key[t] = b; key[o] = a; }
      }
      __syncthreads();
    }
  if (t < n_pad) {
    const unsigned long long kk = key[t];
    const int cls = (int)(kk >> 60), f = (int)(kk & 0xFFF);
    const bool live = cls == 0;
    order[t] = cls == 3 ? 0 : f;
    tau[t] = live ? prefix_v[(size_t)f * K + K - 1] : INFINITY;
    fb[t] = live ? 0 : cls == 2 ? 2 : 1;       // (padding slots: 1, but no query behind them)
    fin_v[t] = INFINITY;                       // a one-entry "running list" that never tightens the bound
    fin_i[t] = INT_MAX;
  }
  if (t < ntiles) {                            // the tile's largest limit = that of its last live entry (sorted)
    int l = -1;
    for (int c = t * tile_q; c < min(n_pad, (t + 1) * tile_q); c++) {
      const unsigned long long kk = key[c];
      if ((kk >> 60) == 0) l = (int)min((unsigned long long)INT_MAX, (kk >> 12) & 0xFFFFFFFFFFFFull);
    }
    lim[t] = l;
  }
}

// the survivors of flagged query f, exactly (the reference's j-ordered unfused sum from the query's fp32 table):
// those below the bound go to the candidate pool of the literal heap.  A query whose queue overflowed emits nothing
// and is left to the segment scan.
template <int VEC>
__global__ __launch_bounds__(256) void rp_filter_emit(const uint8_t *__restrict__ codes, int ng, int m_pad,
                                                      const float *__restrict__ tables /*[f][m_pad][256]*/, int row_base,
                                                      int *__restrict__ cnt, const int *__restrict__ queue, int cap,
                                                      const int *__restrict__ count, int F, const float *__restrict__ tau,
                                                      int *__restrict__ fb, float *__restrict__ evv, int *__restrict__ evi,
                                                      int *__restrict__ evcnt, int pool, const int *__restrict__ order,
                                                      int *__restrict__ scanme /* [slot]: 1 = left to the segment scan */) {
  using Word = typename CodeWord<VEC>::type;
  // compact position fc (queues, bounds, flags) -> slot f (tables, candidate pool)
  const int fc = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  int mine = lane < NSLOT ? cnt[fc * NSLOT + lane] : 0;
  __syncthreads();                                  // every wave has read the counters
  if (tid < NSLOT) cnt[fc * NSLOT + tid] = 0;
  if (fc >= min(*count, F)) return;
  const int f = order[fc];
  if (fb[fc] != 0) {
    if (tid == 0) scanme[f] = fb[fc] == 1;
    return;
  }
  if (__ballot(mine > cap) != 0ull) {
    if (tid == 0) { fb[fc] = 1; scanme[f] = 1; }    // a sub-queue overflowed
    return;
  }
  if (tid == 0) scanme[f] = 0;
  int incl = mine;
#pragma unroll
  for (int o = 1; o < NSLOT; o <<= 1) {
    const int up = __shfl_up(incl, o);
    if (lane >= o) incl += up;
  }
  const int n = readlane_i(incl, NSLOT - 1);
  int start[NSLOT];
#pragma unroll
  for (int sl = 0; sl < NSLOT; sl++) start[sl] = readlane_i(incl - mine, sl);
  auto entry = [&](int e) {
    int sl = 0;
#pragma unroll
    for (int x = 1; x < NSLOT; x++) sl += e >= start[x];
    int off = start[0];
#pragma unroll
    for (int x = 1; x < NSLOT; x++) off = sl == x ? start[x] : off;
    return queue[((size_t)fc * NSLOT + sl) * cap + (e - off)];
  };
  const float bound = tau[fc];
  const float *tq = tables + (size_t)f * m_pad * 256;
  const Word *cw = reinterpret_cast<const Word *>(codes);
  for (int e = tid; e - lane < n; e += 256) {
    const bool have = e < n;
    const int row = have ? entry(e) : 0;
    float d = 0.f;
    for (int g = 0; g < ng; g++) {
      const Word w = cw[((size_t)(row >> 6) * ng + g) * 64 + (row & 63)];
#pragma unroll
      for (int b = 0; b < VEC; b++) d += tq[(size_t)(g * VEC + b) * 256 + code_byte<VEC>(w, b)];
    }
    const bool emit = have && d < bound;            // TopKHeap.update inserts only below its root (strict)
    const unsigned long long mk = __ballot(emit);
    if (mk) {
      int base = 0;
      if (lane == 0) base = atomicAdd(&evcnt[f], (int)__popcll(mk));
      base = readlane_i(base, 0);
      const int pos = base + (int)__popcll(mk & ((1ull << lane) - 1ull));
      if (emit && pos < pool) { evv[(size_t)f * pool + pos] = d; evi[(size_t)f * pool + pos] = row + row_base; }
    }
  }
}

}  // namespace

// The filter launches of an index whose ordered code copy spans windows of several blocks (conflict_order.hip) cover
// whole windows: a row of [from, until) may sit in any block of its window.  Rows outside the range that come along are
// rejected one by one where a lane reports a survivor.  Returns the first block and the number of blocks to scan.
static void filter_block_range(const gulon_index *ix, int qw, int nqg, int nadd, int from, int until, int &rb_begin,
                               int &rb_total) {
  rb_begin = from / 64;
  int rb_end = ceil_div(until, 64);
  const bool ordered = ix->fcodes.p && ix->ng == 1 && ix->vec == 16 && qw == 16 && nqg == 1 && (nadd == 4 || nadd == 8) &&
                       tuning_of(ix).filter_order > 0;
  if (ordered && ix->fwindow > 1) {
    const int w = ix->fwindow;
    rb_begin -= rb_begin % w;
    rb_end = std::min(ceil_div(ix->n, 64), ceil_div(rb_end, w) * w);
  }
  rb_total = rb_end - rb_begin;
}

bool filter_eligible(const gulon_index *ix, int K, int rb_total) {
  const ScanTuning &t = tuning_of(ix);
  return t.filter && !ix->wide && K >= 1 && K <= GULON_MAX_K && (size_t)ix->m_pad * 256 * 4 <= FILTER_LDS_BUDGET &&
         rb_total >= t.filter_min_rb && rb_total >= t.filter_period;
}

void run_filter_query(gulon_index *ix, const float *dQ, int B, int K, int from, int until, bool final_out, int *d_oi,
                      float *d_od, int *d_oc, int *d_of, float *d_pv, int *d_pi, hipStream_t st,
                      const SharedBounds *sb) {
  const ScanTuning &t = tuning_of(ix);
  const int phase = sb ? sb->phase : 0;
#ifdef GULON_FILTER_STAMPS
  if (phase != 2) skip_calls++;
#endif
  const int keff = K + 1;
  const int W = ix->w;               // fp32 table interleave of the exact kernels
  const int QT = W * ix->nsub;
  const int ntiles = ceil_div(B, QT);
  const int Bp = ntiles * QT;
  const int qw = filter_qw(ix);
  const int nqg = filter_nqg(ix);
  // entries summed in 8 bits before widening: 4 (6-bit levels) is cheapest, but the bound's slack
  // grows with m / levels, so wide indexes take 2 (7-bit levels) unless told otherwise
  const int nadd = t.filter_nadd ? t.filter_nadd : (ix->m_pad <= 16 ? 4 : 2);
  const int qmax = 255 / nadd;
  const int ftiles = ceil_div(B, qw * nqg);
  const int Bq = ceil_div(ftiles * nqg * qw, 16) * 16;   // whole 16-query quantisation groups
  const int cap = std::max(64, t.filter_cap / NSLOT);   // entries per sub-queue
  const int rb_begin = from / 64;
  const int rb_total = ceil_div(until, 64) - rb_begin;
  int frb_begin = rb_begin, frb_total = rb_total;        // what the filter launches scan (whole ordering windows)
  filter_block_range(ix, qw, nqg, nadd, from, until, frb_begin, frb_total);
  const int P = t.filter_period;
  const int s0 = std::min(std::max(t.filter_stage0, 0), P - 2);
  const int s1 = std::min(std::max(t.filter_stage1, 1), P - 1 - s0);
  const RbMap all{1, 0, 1};
  RbMap stages[3] = {{P, 0, s0}, {P, s0, s1}, {P, s0 + s1, P - s0 - s1}};
  const int NW = t.threads / 64;
  // sample size ~ sqrt(rows): the sample scan costs ~1.4 us per 1000 rows, the first filter
  // stage's slow path ~ rows / sample -- 54 K rows at 10 M, 19 K at 1.25 M; filter_sample caps it
  const int srows = std::max(4096, std::min(t.filter_sample, (int)(17.0 * std::sqrt((double)rb_total * 64.0))));
  if (phase == 2) {
    // bounds shared by `lists` shards: their union is a sample `lists` times this shard's; once that
    // is about half the rows of the short stages, these stages cost more launches than they save survivors
    // (one rank of 2 / 4 / 8 on the bench index: 1.950 / 1.136 / 0.526 ms with them, 1.944 / 1.074 / 0.496 without)
    const long long early = ((long long)rbmap_count(frb_total, stages[0]) + rbmap_count(frb_total, stages[1])) * 64;
    const bool keep = t.filter_shared_stage1 < 0 ? 2LL * sb->lists * srows < early : t.filter_shared_stage1 != 0;
    if (!keep) {
      stages[0].width = 0;
      stages[1].width = 0;
      stages[2] = RbMap{P, 0, P};
    }
  }

  // chunks per query tile: enough workgroups to fill the chip, but every workgroup stages its
  // 64-128 KiB of tables once (268 MB through L2 for 4096 workgroups), so it should get >= 768 row blocks
  // (48 per wave) where the range allows; the launch is a whole number of rounds of what the chip holds
  // at once (CUs x resident workgroups) where that is possible
  static const size_t lds_pad = getenv("GULON_FILTER_LDS_PAD") ? (size_t)atoi(getenv("GULON_FILTER_LDS_PAD")) : 0;
  const size_t filter_lds = (size_t)nqg * ix->m_pad * 256 * qw + lds_pad;
  const int resident = std::max(1, std::min(2048 / FILTER_THREADS, (int)(160 * 1024 / filter_lds)));
  const int slots = device_cus() * resident;
  auto chunking = [&](int e_count, int tiles, int target, int &nchunks, int &per) {
    const int most = std::max(1, ceil_div(target, tiles));        // launch-size cap
    const int fill = std::max(1, ceil_div(slots, tiles));         // every slot of the chip taken once
    static const int wg_blocks = getenv("GULON_FILTER_WG_BLOCKS") ? std::max(16, atoi(getenv("GULON_FILTER_WG_BLOCKS"))) : 768;   // experiment knob
    int nc = std::max(fill, std::min(most, e_count / wg_blocks));
    if (nc > fill) nc -= nc % fill;                               // whole rounds
    nc = std::min(nc, std::max(1, e_count / NW));                 // at least one block per wave
    per = ceil_div(e_count, nc);
    nchunks = ceil_div(e_count, per);
  };

  // fallback launch: W-query tiles with ONE sub-table (<= 64 KiB of LDS, like a filter workgroup).  The index's
  // preferred two-sub-table workgroup needs both LDS halves of a CU at once, which another batch's filter
  // kernel never leaves free: the (normally empty) launch then waited for that whole kernel -- up to 3 ms
  // of its batch's critical path (rocprofv3: 1.31 ms on average for a launch that takes 4.7 us)
  const int fb_tiles = ntiles * ix->nsub;
  const int cfb = std::max(1, ceil_div(256, fb_tiles));   // one workgroup per CU when everything is redone
  const int pfb = ceil_div(rb_total, cfb);
  const int cfb_n = ceil_div(rb_total, pfb);
  ix->tables.ensure((size_t)Bp * ix->m_pad * 256);
  ix->part_v.ensure((size_t)Bp * cfb_n * keff);
  ix->part_i.ensure((size_t)Bp * cfb_n * keff);
  ix->gtau.ensure((size_t)Bp);
  ix->fin_v.ensure((size_t)Bq * keff);
  ix->fin_i.ensure((size_t)Bq * keff);
  ix->qmins.ensure((size_t)Bq * ix->m_pad);
  ix->qtab.ensure((size_t)Bq * ix->m_pad * 256);
  ix->sv_cnt.ensure((size_t)Bq * NSLOT);
  ix->sv_queue.ensure((size_t)Bq * NSLOT * cap);
  ix->fb_tile.ensure((size_t)ntiles);
  ix->tau0.ensure((size_t)Bp);
  ix->last_filter_tiles = ntiles;
  if (phase != 2) {
    // ONE launch: the batch's counters, Index.prepareQuery for all queries (with the table minima of the B real
    // queries; the padding queries of the last 16-query quantisation group are dead and never read theirs) and the
    // bounds from a strided sample of about filter_sample rows, which also resets the running lists
    int sblocks = std::max(NW, std::min(rb_total, ceil_div(srows, 64)));
    const RbMap smap{std::max(1, rb_total / sblocks), 0, 1};
    const int se = rbmap_count(rb_total, smap);
    const size_t lds_bytes = std::max((size_t)ix->m_pad * 256 * 4 * W, (size_t)W * NW * 64 * 4);
    // the sample scan is a pure gather of 16-byte fp32 entries: it reads the conflict-ordered copy where there is one
    const bool ordered = ix->fcodes.p && ix->ng == 1 && ix->vec == 16 && t.filter_order > 0;
    const uint8_t *scodes = ordered ? ix->fcodes.p : ix->codes.p, *sperm = ordered ? ix->fperm.p : nullptr;
#define BS(V, W_)                                                                                                   \
    {                                                                                                               \
      auto kern = bound_tables<V, W_>;                                                                              \
      HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),                                           \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));                   \
      hipLaunchKernelGGL(kern, dim3(Bp / W_), dim3(BOUND_THREADS), lds_bytes, st, ix->cents.p, ix->from.p,          \
                         ix->sdim.p, ix->d, ix->m, ix->k, dQ, ix->tables.p, ix->qmins.p, scodes, sperm,             \
                         ordered ? ix->fwindow - 1 : 0, ix->ng, ix->m_pad, from, until, rb_begin, se, smap, B, keff, ix->tau0.p, ix->fin_v.p, ix->fin_i.p, \
                         phase == 1 ? sb->bounds_out : nullptr, ix->gtau.p, Bp, ix->fb_tile.p, ntiles,              \
                         ix->sv_cnt.p, Bq * NSLOT);                                                                 \
    }
#ifdef GULON_FILTER_STAMPS
    static unsigned long long *bt_d = nullptr;
    if (getenv("GULON_FILTER_STAMPS") && !bt_d) {
      HIP_CHECK(hipMalloc((void **)&bt_d, 8 * 8 * 65536));
      HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_bt_stamps), &bt_d, sizeof(bt_d)));
    }
#endif
    if (GULON_SKIPPED(1)) {}
    else if (ix->vec == 16) { if (W == 4) BS(16, 4) else if (W == 2) BS(16, 2) else BS(16, 1) }
    else               { if (W == 4) BS(4, 4) else if (W == 2) BS(4, 2) else BS(4, 1) }
#undef BS
    HIP_CHECK(hipGetLastError());
#ifdef GULON_FILTER_STAMPS
    if (bt_d && Bp / W <= 65536) {
      HIP_CHECK(hipStreamSynchronize(st));
      std::vector<unsigned long long> h((size_t)8 * (Bp / W));
      HIP_CHECK(hipMemcpy(h.data(), bt_d, h.size() * 8, hipMemcpyDeviceToHost));
      if (FILE *f = fopen((std::string(getenv("GULON_FILTER_STAMPS")) + ".bt").c_str(), "wb")) {
        fwrite(h.data(), 8, h.size(), f);
        fclose(f);
      }
    }
#endif
  }
  if (phase == 1) return;     // the caller exchanges the bounds and comes back with phase 2
  if (phase == 2) {
    hipLaunchKernelGGL(shared_tau, dim3(ceil_div(B, 4)), dim3(256), 0, st, sb->all_bounds, sb->lists, B, keff, ix->tau0.p);
    HIP_CHECK(hipGetLastError());
  }
  const bool stats = getenv("GULON_FILTER_STATS") != nullptr;
  for (int sidx = 0; sidx < 3; sidx++) {
    const RbMap mp = stages[sidx];
    const int en = rbmap_count(frb_total, mp);
    if (en <= 0) continue;
    int nc = 1, per = 1;
    chunking(en, ftiles, t.filter_blocks, nc, per);
#ifdef GULON_FILTER_STAMPS
    static unsigned long long *qt_d = nullptr;
    if (getenv("GULON_FILTER_STAMPS") && !qt_d) {
      HIP_CHECK(hipMalloc((void **)&qt_d, 8 * 4 * 65536));
      HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_qt_stamps), &qt_d, sizeof(qt_d)));
    }
#endif
    if (!GULON_SKIPPED(2))
    hipLaunchKernelGGL(qt_quantize, dim3(Bq / 4, ix->m_pad), dim3(256), 0, st, ix->tables.p, W, Bp, ix->m_pad, ix->k,
                       B, ix->qmins.p, ix->fin_v.p, ix->fin_i.p, ix->tau0.p, keff, qmax, qw, ix->qtab.p,
                       ix->fb_tile.p, QT);
    HIP_CHECK(hipGetLastError());
#ifdef GULON_FILTER_STAMPS
    if (qt_d && (Bq / 4) * ix->m_pad <= 65536) {
      HIP_CHECK(hipStreamSynchronize(st));
      std::vector<unsigned long long> h((size_t)4 * (Bq / 4) * ix->m_pad);
      HIP_CHECK(hipMemcpy(h.data(), qt_d, h.size() * 8, hipMemcpyDeviceToHost));
      if (FILE *f = fopen((std::string(getenv("GULON_FILTER_STAMPS")) + ".qt").c_str(), "wb")) {
        fwrite(h.data(), 8, h.size(), f);
        fclose(f);
      }
    }
#endif
    hipEvent_t ev0 = nullptr, ev1 = nullptr, lane_pending = nullptr;
    const bool main_stage = sidx == 2;
    // a first stage that keeps more than ~3 % of its (query, row) pairs means the bounds do not
    // separate anything for this query: give up on it before the main stage
    const int give_up = main_stage ? 0 : std::max(256, (int)std::min<long long>((long long)en * 64 * 3 / 100, 1 << 30));
    const bool timed = ix->profile && main_stage;   // the roofline line is about the main-stage kernel
    if (timed) {
      ev0 = ix->take_event();
      ev1 = ix->take_event();
    }
    {
      int dev = 0;
      HIP_CHECK(hipGetDevice(&dev));
      bool fresh = false;
      const hipEvent_t lane_ev = filter_lane().of(dev, fresh);
      // only the main stage takes turns.  (Two host threads enqueueing at the same instant may both wait for the SAME
      // earlier launch and then share the chip once: the event orders launches for throughput, never for results.)
      static const bool take_turns = !(getenv("GULON_FILTER_LANE") && atoi(getenv("GULON_FILTER_LANE")) == 0);   // experiment knob
      // (ranges below ~2 M rows -- a shard of an 8-GPU index -- do better without the turn-taking: 0.413 against 0.430 ms
      // per batch at 1.25 M rows, three batches in flight; 10 M rows: 2.36 against 2.38, within the noise, and the
      // kernel durations the bench reports stay those of one kernel at a time)
      const bool turns = take_turns && (long long)rb_total * 64 >= (2ll << 20);
      if (main_stage && !fresh && turns) HIP_CHECK(hipStreamWaitEvent(st, lane_ev, 0));
      if (timed) HIP_CHECK(hipEventRecord(ev0, st));         // after the wait: the kernel's own duration
      if (!(main_stage && GULON_SKIPPED(16)) && !(!main_stage && GULON_SKIPPED(32)))
      launch_filter(ix, qw, nqg, nadd, ftiles, nc, frb_begin, en, per, mp, from, until, cap, main_stage ? 1 : 0, B, st);
      static const int lane_after = getenv("GULON_LANE_AFTER_SURVIVORS") ? atoi(getenv("GULON_LANE_AFTER_SURVIVORS")) : 0;   // experiment knob
      if (main_stage && !lane_after) HIP_CHECK(hipEventRecord(lane_ev, st));
      lane_pending = main_stage && lane_after ? lane_ev : nullptr;
    }
    if (stats) {   // debugging aid (GULON_FILTER_STATS=1): synchronous survivor statistics
      HIP_CHECK(hipStreamSynchronize(st));
      std::vector<int> h((size_t)Bq * NSLOT);
      HIP_CHECK(hipMemcpy(h.data(), ix->sv_cnt.p, sizeof(int) * h.size(), hipMemcpyDeviceToHost));
      long long tot = 0; int mx = 0;
      for (int v : h) { tot += v; mx = std::max(mx, v); }
      fprintf(stderr, "[filter] stage %d: %d row blocks, %d x %d workgroups, survivors total %lld (%.3g of pairs), "
              "max per sub-queue %d (cap %d)\n", sidx + 1, en, ftiles, nc, tot, (double)tot / ((double)en * 64 * B), mx, cap);
    }
    if (timed) {
      HIP_CHECK(hipEventRecord(ev1, st));
      ix->events.emplace_back(ev0, ev1);
      ix->prof_rows += std::min<long long>((long long)en * 64, (long long)until - from);
    }
    {
      const size_t sv_lds = (size_t)ix->m_pad * 256 * W * sizeof(float);
      auto kern = ix->vec == 16 ? survivors_kernel<16> : survivors_kernel<4>;
      HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sv_lds));
      if (!GULON_SKIPPED(4))
        hipLaunchKernelGGL(kern, dim3(Bq / W), dim3(64 * SV_WAVES * W), sv_lds, st, ix->codes.p, ix->ng, ix->m_pad, ix->tables.p,
                           W, ix->row_base, ix->sv_cnt.p, ix->sv_queue.p, cap, B, keff, ix->fin_v.p, ix->fin_i.p,
                           ix->fb_tile.p, QT, give_up);
    }
    HIP_CHECK(hipGetLastError());
    if (lane_pending) HIP_CHECK(hipEventRecord(lane_pending, st));   // the next batch's main stage starts behind this batch's survivor pass
  }

  // fallback (device-side decision): flagged query tiles are rescanned exactly over all rows
  {
    if (!ix->fb_hint_h) {
      HIP_CHECK(hipHostMalloc(reinterpret_cast<void **>(&ix->fb_hint_h), sizeof(int), hipHostMallocMapped));
      *ix->fb_hint_h = 0;
      HIP_CHECK(hipHostGetDevicePointer(reinterpret_cast<void **>(&ix->fb_hint_d), ix->fb_hint_h, 0));
    }
    // a hint, read without synchronisation: did a recent fallback launch of this index find work?
    // (launches are enqueued ahead of their execution, so the word lags by the batches in flight: once seen,
    // it keeps the next 8 launches wide, and every launch that finds work arms it again)
    volatile int *hint = ix->fb_hint_h;
    if (*hint != 0) { *hint = 0; ix->fb_wide_left = 8; }
    const bool wide = ix->fb_wide_left > 0;
    if (wide) ix->fb_wide_left--;
    if (!GULON_SKIPPED(8))
    launch_scan(ix, fb_tiles, cfb_n, rb_begin, rb_total, pfb, all, from, until, keff, st, nullptr, nullptr,
                ix->fb_tile.p, true, wide ? fb_tiles : 8, ix->fb_hint_d);
  }
  if (!GULON_SKIPPED(8))
  launch_merge_enabled(ix->part_v.p, ix->part_i.p, cfb_n, (long long)keff, (long long)cfb_n * keff, B, K, ix->fin_v.p,
                       ix->fin_i.p, ix->fb_tile.p, QT, st);

#ifdef GULON_FILTER_STAMPS
  {   // experiment: what does one more (empty) launch per batch cost while other batches are in flight?
    static const int dummies = getenv("GULON_DUMMY_LAUNCHES") ? atoi(getenv("GULON_DUMMY_LAUNCHES")) : 0;
    for (int i = 0; i < dummies; i++) hipLaunchKernelGGL(dummy_kernel, dim3(64), dim3(64), 0, st, ix->fb_tile.p);
  }
#endif
  if (stats) {
    HIP_CHECK(hipStreamSynchronize(st));
    std::vector<int> h((size_t)ntiles);
    HIP_CHECK(hipMemcpy(h.data(), ix->fb_tile.p, sizeof(int) * h.size(), hipMemcpyDeviceToHost));
    int nfb = 0;
    for (int v : h) nfb += v != 0;
    fprintf(stderr, "[filter] %d of %d query tiles redone by the exact scan\n", nfb, ntiles);
  }
  int *flags = d_of;
  if (final_out && replay_enabled() && flags == nullptr) {
    ix->flags_scratch.ensure((size_t)B);
    flags = ix->flags_scratch.p;
  }
  launch_merge(final_out, ix->fin_v.p, ix->fin_i.p, 1, 0LL, (long long)keff, B, K, d_oi, d_od, d_oc, flags, d_pv, d_pi,
               st);
  if (final_out && replay_enabled()) run_tie_replay(ix, dQ, B, K, from, until, d_oi, d_od, d_oc, flags, st);
  if (final_out && replay_enabled()) run_nonfinite_literal(ix, dQ, B, K, from, until, d_oi, d_od, d_oc, d_of, st);
}

bool replay_level2_filtered(gulon_index *ix, int F, int K, int rb_lo, int rb_hi, int from, int until,
                            const float *tables, const float *mins, const float *prefix_v, const int *prefix_c,
                            const int *count, float *evv, int *evi, int *evcnt, int pool, int **only, hipStream_t st,
                            const int *done, const int *rlast, int *scanme /* [F]: written for every flagged slot */) {
  const ScanTuning &t = tuning_of(ix);
  const int level_from = std::max(from, rb_lo * 64);     // the level's rows: nothing of the earlier levels is emitted again
  int e_count = rb_hi - rb_lo;
  // (the same conditions as filter_eligible; the byte-code kernels only)
  if (!t.filter || ix->wide || (size_t)ix->m_pad * 256 * 4 > FILTER_LDS_BUDGET || e_count < t.filter_min_rb || K < 1)
    return false;
  const int qw = filter_qw(ix), nqg = filter_nqg(ix);
  const int nadd = t.filter_nadd ? t.filter_nadd : (ix->m_pad <= 16 ? 4 : 2);
  const int qmax = 255 / nadd;
  filter_block_range(ix, qw, nqg, nadd, rb_lo * 64, std::min(until, rb_hi * 64), rb_lo, e_count);   // whole ordering windows
  const int ftiles = ceil_div(F, qw * nqg);
  const int Fq = ceil_div(ftiles * nqg * qw, 16) * 16;
  const int cap = std::max(64, t.filter_cap / NSLOT);
  // a launch is worth its fixed costs (quantisation, table staging of every workgroup) from about two query tiles on
  const int min_flagged = 2 * qw;
  GULON_REQUIRE(F <= 1024, "internal: %d flagged queries per replay round", F);
  const int tile_q = qw * nqg;
  ix->rp_fb.ensure((size_t)Fq + ftiles); ix->rp_fini.ensure((size_t)Fq);
  ix->rp_tau.ensure((size_t)Fq); ix->rp_finv.ensure((size_t)Fq);
  ix->rp_order.ensure((size_t)Fq);
  ix->qtab.ensure((size_t)Fq * ix->m_pad * 256);
  ix->sv_cnt.ensure((size_t)Fq * NSLOT);
  ix->sv_queue.ensure((size_t)Fq * NSLOT * cap);
  hipLaunchKernelGGL(rp_filter_order, dim3(1), dim3(1024), 0, st, count, F, min_flagged, K, prefix_v, prefix_c, done, rlast,
                     ix->row_base, Fq, tile_q, ftiles, ix->rp_order.p, ix->rp_tau.p, ix->rp_fb.p, ix->rp_finv.p, ix->rp_fini.p,
                     ix->rp_fb.p + Fq);
  HIP_CHECK(hipMemsetAsync(ix->sv_cnt.p, 0, sizeof(int) * (size_t)Fq * NSLOT, st));
  // tables of the flagged queries are [slot][m_pad][256]: "one query per entry" (W = 1) in qt_quantize's terms, read
  // through the order; queries beyond F (the padding of the last 16-query group) read no table
  hipLaunchKernelGGL(qt_quantize, dim3(Fq / 4, ix->m_pad), dim3(256), 0, st, tables, 1, F, ix->m_pad, ix->k, F, mins,
                     ix->rp_finv.p, ix->rp_fini.p, ix->rp_tau.p, 1, qmax, qw, ix->qtab.p, ix->rp_fb.p, 1, ix->rp_order.p);
  HIP_CHECK(hipGetLastError());
  const size_t filter_lds = (size_t)nqg * ix->m_pad * 256 * qw;
  const int resident = std::max(1, std::min(2048 / FILTER_THREADS, (int)(160 * 1024 / filter_lds)));
  const int slots = device_cus() * resident;
  const int NW = FILTER_THREADS / 64;
  int nc = std::max(1, std::min(std::max(ceil_div(slots, ftiles), e_count / 768), std::max(1, e_count / NW)));
  const int per = ceil_div(e_count, nc);
  nc = ceil_div(e_count, per);
  const RbMap all{1, 0, 1};
  launch_filter(ix, qw, nqg, nadd, ftiles, nc, rb_lo, e_count, per, all, level_from, until, cap, 2, F, st, ix->rp_fb.p, 1);
  if (ix->vec == 16)
    hipLaunchKernelGGL(rp_filter_emit<16>, dim3(F), dim3(256), 0, st, ix->codes.p, ix->ng, ix->m_pad, tables, ix->row_base,
                       ix->sv_cnt.p, ix->sv_queue.p, cap, count, F, ix->rp_tau.p, ix->rp_fb.p, evv, evi, evcnt, pool,
                       ix->rp_order.p, scanme);
  else
    hipLaunchKernelGGL(rp_filter_emit<4>, dim3(F), dim3(256), 0, st, ix->codes.p, ix->ng, ix->m_pad, tables, ix->row_base,
                       ix->sv_cnt.p, ix->sv_queue.p, cap, count, F, ix->rp_tau.p, ix->rp_fb.p, evv, evi, evcnt, pool,
                       ix->rp_order.p, scanme);
  HIP_CHECK(hipGetLastError());
  *only = scanme;
  return true;
}

}  // namespace gulon
