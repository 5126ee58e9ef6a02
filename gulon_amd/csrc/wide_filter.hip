// The quantized lower-bound filter (filter.hip) for indexes with more than 256 centroids per quantizer:
// 10-, 12- and 16-bit codes (Coder.scala:99-127,142-168), as far as a query group's 8-bit tables fit LDS --
// m * k * QW bytes <= 144 KiB with QW = 8 or 4 queries per entry (k = 1024 at m = 16: 128 KiB with 8 queries).
// Results are the exact wide scan's (wide.hip), bit for bit; the same argument as for byte codes:
//
//   1. the exact scan (scan_wide) of a PREFIX of the row range gives every query a running (K+1)-list; its last
//      entry tau is a valid upper bound on the final (K+1)-th distance;
//   2. the fp32 tables T[q][j][c] are quantized downwards against tau exactly as qt_quantize does
//      (q_j[c] = min(QMAX, floor((T - min_j) / delta)), delta = (tau' - sum_j min_j) / (QMAX - 1), fp64, guarded), so
//      a row whose levels sum above QMAX - 1 has a distance above tau and cannot be among the K+1 smallest;
//   3. wf_filter walks the REST of the rows: lane = row, one ds_read_b64 / b32 per (row, quantizer) serves 8 / 4
//      queries (the exact wide scan: one ds_read_b32 per query), four entries summed as bytes before the sums are
//      widened to 16 bits; the few surviving (query, row) pairs are queued;
//   4. wf_survivors re-evaluates them in the reference's arithmetic (j ascending, unfused fp32) and merges them
//      into the running lists under the (distance, row id) order of the exact scan;
//   5. a query with an unusable bound (fewer than K+1 sample rows, NaN / inf) or an overflowing queue is redone
//      by scan_wide over the whole range (decided on the device, per query).
// Larger code books (k = 4096 ... 32 768 at m = 16, four queries per entry): the group's 8-bit table no longer fits
// LDS as a whole, so wf_filter walks it in SLICES of as many quantizers as fit (8 at k = 4096), one launch per slice;
// between the launches the byte sums of every (query, row) are parked in HBM -- one byte each, saturated at 255 (a sum
// above QMAX - 1 is dead and stays dead), one dword per row and four-query group -- and only the last slice tests.
#include <climits>
#include <type_traits>

#include "scan.hpp"

namespace gulon {

namespace {

constexpr int WF_THREADS = 1024;
constexpr int WF_NW = WF_THREADS / 64;
constexpr int WF_NSLOT = 16;                 // survivor sub-queues per query
constexpr size_t WF_LDS_BUDGET = 144 * 1024;
constexpr int WF_NADD = 4, WF_QMAX = 255 / WF_NADD;   // 6-bit levels: four entries summed per byte

__device__ inline uint32_t wf_pk_sub_sat_u16(uint32_t a, uint32_t b) {   // per 16-bit half: max(a - b, 0)
  uint32_t d;
  asm("v_pk_sub_u16 %0, %1, %2 clamp" : "=v"(d) : "v"(a), "v"(b));
  return d;
}

__global__ void wf_reset(int *__restrict__ fb, int n_fb, int *__restrict__ cnt, int n_cnt) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n_fb) fb[t] = 0;
  if (t < n_cnt) cnt[t] = 0;
}

// mins[q][j] = min_c T[q][j][c] (NaN entries ignored, as fminf does)
__global__ __launch_bounds__(256) void wf_table_mins(const float *__restrict__ T, int m, int k, float *__restrict__ mins) {
  __shared__ float red[4];
  const int j = blockIdx.x, q = blockIdx.y, tid = threadIdx.x;
  const float *t = T + ((size_t)q * m + j) * k;
  float mn = INFINITY;
  for (int c = tid; c < k; c += 256) mn = fminf(mn, t[c]);
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) mn = fminf(mn, __shfl_xor(mn, o));
  if ((tid & 63) == 0) red[tid >> 6] = mn;
  __syncthreads();
  if (tid == 0) mins[(size_t)q * m + j] = fminf(fminf(red[0], red[1]), fminf(red[2], red[3]));
}

// qtab[q / QW][j][c][q % QW] <- level of T[q][j][c] against the query's bound (the arithmetic of qt_quantize,
// filter.hip); a query without a usable bound gets QMAX everywhere (nothing survives) and its fb flag
template <int QW>
__global__ __launch_bounds__(256) void wf_quantize(const float *__restrict__ T, int m, int k, int B,
                                                   const float *__restrict__ mins, const float *__restrict__ fin_v,
                                                   const int *__restrict__ fin_i, int keff,
                                                   uint8_t *__restrict__ qtab, size_t tstride /* bytes per query group */,
                                                   int *__restrict__ fb) {
  __shared__ double s_delta[QW], s_inv[QW];
  __shared__ float s_min[QW];
  __shared__ int s_dead[QW];
  const int c = blockIdx.x * 256 + threadIdx.x, j = blockIdx.y, grp = blockIdx.z;
  if (threadIdx.x < QW) {
    const int s = threadIdx.x, q = grp * QW + s;
    int dead = 1;
    double delta = 1.0;
    float mn = 0.f;
    if (q < B) {
      mn = mins[(size_t)q * m + j];
      const bool full = fin_i[(size_t)q * keff + keff - 1] != INT_MAX;
      const float tau = full ? fin_v[(size_t)q * keff + keff - 1] : INFINITY;
      if (!(tau < INFINITY)) {
        if (j == 0 && blockIdx.x == 0) fb[q] = 1;           // no usable bound: redone by the exact scan
      } else {
        double sum_min = 0.0;
        for (int jj = 0; jj < m; jj++) sum_min += (double)mins[(size_t)q * m + jj];
        const double taup = (double)tau * (1.0 + 2.0 * m * 5.97e-8) * (1.0 + 1e-9);
        double budget = taup - sum_min;
        if (!(budget > 0.0)) budget = 0.0;
        delta = budget / (double)(WF_QMAX - 1);
        if (delta < 1e-290) delta = 1e-290;
        dead = 0;
      }
    }
    s_dead[s] = dead; s_delta[s] = delta; s_inv[s] = 1.0 / delta; s_min[s] = mn;
  }
  __syncthreads();
  if (c >= k) return;
  uint32_t out[QW / 4];
#pragma unroll
  for (int u4 = 0; u4 < QW / 4; u4++) {
    uint32_t word = 0;
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int s = u4 * 4 + u, q = grp * QW + s;
      int qv = WF_QMAX;
      if (q < B && !s_dead[s]) {
        const float v = T[((size_t)q * m + j) * k + c];
        if (v == v) {
          double x = ((double)v - (double)s_min[s]) * (1.0 - 8.9e-16);
          if (x < 0.0) x = 0.0;
          const double r = x * s_inv[s];
          if (r < (double)WF_QMAX) {
            qv = (int)r;
            while (qv > 0 && (double)qv * s_delta[s] > x) qv--;   // guard the rounding of the reciprocal
          }
        }
      }
      word |= (uint32_t)qv << (8 * u);
    }
    out[u4] = word;
  }
  uint32_t *dst = reinterpret_cast<uint32_t *>(qtab + (size_t)grp * tstride) + ((size_t)j * k + c) * (QW / 4);
#pragma unroll
  for (int u4 = 0; u4 < QW / 4; u4++) dst[u4] = out[u4];
}

template <int QW> struct WfEntry;
template <> struct WfEntry<8> { using type = uint2; };
template <> struct WfEntry<4> { using type = uint32_t; };

// lane = row; workgroup = QW queries x a chunk of row blocks of [rb_begin, rb_begin + rb_count).
// MQ = ceil(m / 4) groups of four quantizers, a compile-time bound (0: any m, no look-ahead): the codes of the NEXT
// row block are requested before the current one's look-ups, so that their latency passes under the LDS gathers.
// SLICED (QW = 4, MQ = 0): quantizers [j0, j1) of the table only; park[(tile * rb_count + e) * 64 + lane] holds the
// byte sums of the earlier slices (j0 > 0) and receives this one's (j1 < m).
template <int QW, int MQ, bool SLICED = false>
__global__ __launch_bounds__(WF_THREADS) void wf_filter(const uint16_t *__restrict__ codes, int m, int k,
                                                        const uint8_t *__restrict__ qtab, size_t tstride, int row_from, int row_until,
                                                        int rb_begin, int rb_count, int rb_per_chunk,
                                                        int *__restrict__ cnt, int *__restrict__ queue, int cap,
                                                        int *__restrict__ fb, int B, int j0 = 0, int j1 = 0,
                                                        uint32_t *__restrict__ park = nullptr) {
  static_assert(!SLICED || (QW == 4 && MQ > 0), "the sliced form: four queries per entry, at most 4 MQ quantizers per slice");
  constexpr int DW = QW / 4;
  constexpr uint32_t QMAXP = (uint32_t)WF_QMAX * 0x00010001u;   // survive <=> sum <= QMAX - 1
  using QE = typename WfEntry<QW>::type;
  extern __shared__ uint4 wf_lds_raw[];
  QE *lds = reinterpret_cast<QE *>(wf_lds_raw);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tile = blockIdx.x, chunk = blockIdx.y;
  {   // every query of the tile already goes to the exact scan
    bool live = false;
    for (int s = 0; s < QW; s++) live = live || (tile * QW + s < B && fb[tile * QW + s] == 0);
    if (!live) return;
  }
  if (SLICED) {   // quantizers [j0, j1) of the group's tables: (j1 - j0) * k entries of four bytes
    const int n4 = (j1 - j0) * k;
    const uint32_t *src = reinterpret_cast<const uint32_t *>(qtab + (size_t)tile * tstride) + (size_t)j0 * k;
    uint32_t *dst = reinterpret_cast<uint32_t *>(wf_lds_raw);
    for (int e = tid; e < n4; e += WF_THREADS) dst[e] = src[e];
  } else {   // the group's tables: tstride bytes (m * k * QW rounded up to 16)
    const int n16 = (int)(tstride / 16);
    const uint4 *src = reinterpret_cast<const uint4 *>(qtab + (size_t)tile * tstride);
    for (int e = tid; e < n16; e += WF_THREADS) wf_lds_raw[e] = src[e];
  }
  __syncthreads();
  const int jlo = SLICED ? j0 : 0, jhi = SLICED ? j1 : m;
  const int slot = (chunk * WF_NW + wave) & (WF_NSLOT - 1);
  const int e0 = chunk * rb_per_chunk, e1 = min(rb_count, e0 + rb_per_chunk);
  constexpr int MC = MQ > 0 ? 4 * MQ : 1;
  unsigned short cn[MC];                 // MQ > 0: codes of the row block about to be processed (SLICED: of its slice)
  uint32_t pn = 0;                       // SLICED, j0 > 0: its parked byte sums
  auto fetch = [&](int e) {
    if (MQ == 0) return;
    const int ec = min(e, rb_count - 1);                                                     // (clamped: always valid)
    const uint16_t *p = codes + (size_t)(rb_begin + ec) * m * 64 + lane;
#pragma unroll
    for (int j = 0; j < MC; j++) cn[j] = jlo + j < jhi ? p[(size_t)(jlo + j) * 64] : (unsigned short)0;
    if (SLICED && jlo > 0) pn = park[((size_t)tile * rb_count + ec) * 64 + lane];
  };
  fetch(e0 + wave);
  for (int e = e0 + wave; e < e1; e += WF_NW) {
    const int rb = rb_begin + e;
    const uint16_t *p = codes + (size_t)rb * m * 64 + lane;
    unsigned short cc[MC];
#pragma unroll
    for (int j = 0; j < MC; j++) cc[j] = cn[j];
    const uint32_t pw = pn;
    fetch(e + WF_NW);
    uint32_t acc[2 * DW];
#pragma unroll
    for (int x = 0; x < 2 * DW; x++) acc[x] = 0;
    if (SLICED && jlo > 0) {   // the earlier slices' sums: bytes q0 | q1 << 8 | q2 << 16 | q3 << 24
      acc[0] = pw;                                  // (unmasked, like every term of acc[2 dd]: corrected after the groups)
      acc[1] = (pw >> 8) & 0x00FF00FFu;
    }
    auto group = [&](const int j, const int cnt4, auto from_regs) {   // quantizers j .. j + cnt4 - 1 (cnt4 <= 4)
      uint32_t xs[DW];
#pragma unroll
      for (int dd = 0; dd < DW; dd++) xs[dd] = 0;
#pragma unroll
      for (int a = 0; a < WF_NADD; a++) {            // bytes cannot carry: NADD * QMAX <= 255
        if (a < cnt4) {
          const int c = decltype(from_regs)::value ? (int)cc[(j + a - (SLICED ? jlo : 0)) % MC] : (int)p[(size_t)(j + a) * 64];
          const QE y = lds[(size_t)(j + a - jlo) * k + c];
#pragma unroll
          for (int dd = 0; dd < DW; dd++) xs[dd] += reinterpret_cast<const uint32_t *>(&y)[dd];
        }
      }
#pragma unroll
      for (int dd = 0; dd < DW; dd++) {
        acc[2 * dd] += xs[dd];                                               // filter.hip's running sum: W = S0 + 2^8 S1 + 2^16 S2 + 2^24 S3
        acc[2 * dd + 1] += __builtin_amdgcn_perm(0u, xs[dd], 0x0C030C01u);   // bytes 1 and 3: O = S1 + 2^16 S3
      }
    };
    if (MQ > 0) {
#pragma unroll
      for (int g4 = 0; g4 < MQ; g4++) {
        const int left = jhi - jlo - 4 * g4;           // (uniform)
        if (left >= 4) group(jlo + 4 * g4, 4, std::true_type{});
        else if (left > 0) group(jlo + 4 * g4, left, std::true_type{});
      }
    } else {
      int j = jlo;
      for (; j + 4 <= jhi; j += 4) group(j, 4, std::false_type{});
      if (j < jhi) group(j, jhi - j, std::false_type{});
    }
#pragma unroll
    for (int dd = 0; dd < DW; dd++) acc[2 * dd] -= acc[2 * dd + 1] << 8;     // W - 2^8 O = S0 + 2^16 S2
    if (SLICED && jhi < m) {   // not the last slice: park the sums, one byte per query, saturated
      uint32_t a, b;
      asm("v_pk_min_u16 %0, %1, %2" : "=v"(a) : "v"(acc[0]), "v"(0x00FF00FFu));
      asm("v_pk_min_u16 %0, %1, %2" : "=v"(b) : "v"(acc[1]), "v"(0x00FF00FFu));
      park[((size_t)tile * rb_count + e) * 64 + lane] = a | (b << 8);
      continue;
    }
    const int row = rb * 64 + lane;
    const bool valid = row >= row_from && row < row_until;
    uint32_t any = 0, left[2 * DW];
#pragma unroll
    for (int x = 0; x < 2 * DW; x++) {
      left[x] = wf_pk_sub_sat_u16(QMAXP, acc[x]);     // non-zero half <=> that query keeps this row
      any |= left[x];
    }
    if (__ballot(valid && any != 0) != 0ull) {        // rare path
#pragma unroll
      for (int x = 0; x < 2 * DW; x++) {
        const uint32_t l = valid ? left[x] : 0u;
        if (__ballot(l != 0) == 0ull) continue;
        const int q0 = tile * QW + 4 * (x >> 1) + (x & 1);   // low half: query q0, high half: q0 + 2
        if (l & 0xFFFFu) {
          const int sq = q0 * WF_NSLOT + slot;
          const int pos = atomicAdd(&cnt[sq], 1);
          if (pos < cap) queue[(size_t)sq * cap + pos] = row;
          else fb[q0] = 1;                              // (padding queries are all-QMAX: they never get here)
        }
        if (l >> 16) {
          const int sq = (q0 + 2) * WF_NSLOT + slot;
          const int pos = atomicAdd(&cnt[sq], 1);
          if (pos < cap) queue[(size_t)sq * cap + pos] = row;
          else fb[q0 + 2] = 1;
        }
      }
    }
  }
}

// exact re-evaluation of the survivors; one workgroup (4 waves) per query (the structure of survivors_kernel,
// filter.hip: wave 0 continues the running list, the others collect what beats its last entry, merged at the end)
constexpr int WF_SV_WAVES = 4;
__global__ __launch_bounds__(64 * WF_SV_WAVES) void wf_survivors(const uint16_t *__restrict__ codes, int m, int k,
                                                                 const float *__restrict__ T, int row_base,
                                                                 int *__restrict__ cnt, const int *__restrict__ queue,
                                                                 int cap, int B, int keff, float *__restrict__ fin_v,
                                                                 int *__restrict__ fin_i, int *__restrict__ fb) {
  __shared__ float mv[(WF_SV_WAVES - 1) * 64];
  __shared__ int mi[(WF_SV_WAVES - 1) * 64];
  const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int mine = lane < WF_NSLOT ? cnt[q * WF_NSLOT + lane] : 0;
  __syncthreads();                                  // every wave has read the counters
  if (wave == 0 && lane < WF_NSLOT) cnt[q * WF_NSLOT + lane] = 0;
  if (q >= B) return;
  if (__ballot(mine > cap) != 0ull) {
    if (tid == 0) fb[q] = 1;             // a sub-queue overflowed: the exact scan redoes this query
    mine = min(mine, cap);
  }
  int incl = mine;
#pragma unroll
  for (int o = 1; o < WF_NSLOT; o <<= 1) {
    const int up = __shfl_up(incl, o);
    if (lane >= o) incl += up;
  }
  const int n = readlane_i(incl, WF_NSLOT - 1);
  int start[WF_NSLOT];
#pragma unroll
  for (int sl = 0; sl < WF_NSLOT; sl++) start[sl] = readlane_i(incl - mine, sl);
  auto entry = [&](int e) {   // e-th survivor of the query, e < n
    int sl = 0;
#pragma unroll
    for (int x = 1; x < WF_NSLOT; x++) sl += e >= start[x];
    int off = start[0];
#pragma unroll
    for (int x = 1; x < WF_NSLOT; x++) off = sl == x ? start[x] : off;
    return queue[((size_t)q * WF_NSLOT + sl) * cap + (e - off)];
  };
  const float fv = lane < keff ? fin_v[(size_t)q * keff + lane] : INFINITY;
  const int fi = lane < keff ? fin_i[(size_t)q * keff + lane] : INT_MAX;
  const float bound_v = readlane_f(fv, keff - 1);
  const int bound_i = readlane_i(fi, keff - 1);
  WaveList wl;
  wl.init();
  if (wave == 0) { wl.v = fv; wl.i = fi; wl.tau = bound_v; wl.tau_i = bound_i; }
  const float *tq = T + (size_t)q * m * k;
  constexpr int STEP = 64 * WF_SV_WAVES;
  for (int e = wave * 64 + lane; e - lane < n; e += STEP) {
    const bool have = e < n;
    const int row = have ? entry(e) : 0;
    const uint16_t *p = codes + (size_t)(row >> 6) * m * 64 + (row & 63);
    float d = 0.f;                      // the reference's order: j ascending, unfused fp32
    for (int j = 0; j < m; j++) d += tq[(size_t)j * k + p[(size_t)j * 64]];
    const int cr = row + row_base;
    const bool in_bound = d < bound_v || (d == bound_v && cr < bound_i);
    unsigned long long mk = __ballot(have && in_bound && wl.accepts(d, cr));
    while (mk) {
      const int l = __ffsll((long long)mk) - 1;
      mk &= mk - 1;
      const float v = readlane_f(d, l);
      const int r = readlane_i(cr, l);
      if (wl.accepts(v, r)) wl.insert(v, r, keff, lane);
    }
  }
  if (wave > 0) { mv[(wave - 1) * 64 + lane] = wl.v; mi[(wave - 1) * 64 + lane] = wl.i; }
  __syncthreads();
  if (wave == 0) {
    for (int w2 = 0; w2 < WF_SV_WAVES - 1; w2++)
      for (int e = 0; e < keff; e++) {
        const float v = mv[w2 * 64 + e];
        const int r = mi[w2 * 64 + e];
        if (r == INT_MAX) break;      // sorted: the rest is padding
        if (wl.accepts(v, r)) wl.insert(v, r, keff, lane);
      }
    if (lane < keff) {
      fin_v[(size_t)q * keff + lane] = wl.v;
      fin_i[(size_t)q * keff + lane] = wl.i;
    }
  }
}

int wf_qw(const gulon_index *ix) {    // queries per 8-bit table entry; 0: not even one quantizer's entries fit
  const size_t ent = (size_t)ix->m * ix->k;
  if (ent * 8 <= WF_LDS_BUDGET) return 8;
  if ((size_t)ix->k * 4 <= WF_LDS_BUDGET) return 4;
  return 0;
}
int wf_slice(const gulon_index *ix) {   // quantizers per launch of wf_filter (m: the whole table fits)
  const int qw = wf_qw(ix);
  if (qw == 0) return 0;
  const size_t fit = WF_LDS_BUDGET / ((size_t)ix->k * qw);
  if (fit >= (size_t)ix->m) return ix->m;
  return (int)std::min<size_t>(fit, 8);   // (a slice's codes are held in registers while the previous block is looked up)
}

}  // namespace

// (K <= 63, one sub-batch of fp32 tables, the query's fp32 table in LDS for the sample scan, enough rows to pay)
bool wide_filter_eligible(const gulon_index *ix, int B, int K, int rb_total) {
  const ScanTuning &t = tuning_of(ix);
  const size_t table_bytes = (size_t)ix->m * ix->k * sizeof(float);
  const bool sliced = wf_slice(ix) < ix->m;
  return t.filter && ix->wide && K >= 1 && K <= GULON_MAX_K && wf_qw(ix) != 0 &&
         (size_t)B * table_bytes <= (1ull << 30) && rb_total >= 8 * t.filter_min_rb /* 256 K rows by default */ &&
         (!sliced || (size_t)B * rb_total * 64 <= (2ull << 30) /* the parked byte sums */);
}

void run_wide_filter_query(gulon_index *ix, const float *dQ, int B, int K, int from, int until, bool final_out,
                           int *d_oi, float *d_od, int *d_oc, int *d_of, float *d_pv, int *d_pi, hipStream_t st) {
  const ScanTuning &t = tuning_of(ix);
  const int keff = K + 1, m = ix->m, k = ix->k;
  const int QW = wf_qw(ix);
  const int rb_begin = from / 64, rb_total = ceil_div(until, 64) - rb_begin;
  const int ngrp = ceil_div(B, QW), Bq = ngrp * QW;
  const int cap = std::max(64, t.filter_cap / WF_NSLOT);
  // the sample: a prefix of ~ 17 sqrt(rows) rows (as the byte-code filter sizes its sample), at least 16 K rows
  const long long rows = (long long)rb_total * 64;
  int sblocks = (int)std::min<long long>(rb_total / 2, std::max<long long>(256, (long long)(17.0 * std::sqrt((double)rows)) / 64));
  const int s_end = std::min(until, (rb_begin + sblocks) * 64);
  const int f_begin = rb_begin + sblocks, f_count = rb_total - sblocks;      // the filter's row blocks
  // scratch
  ix->tables.ensure((size_t)B * m * k);
  ix->fin_v.ensure((size_t)Bq * keff);
  ix->fin_i.ensure((size_t)Bq * keff);
  ix->qmins.ensure((size_t)B * m);
  const size_t tstride = ((size_t)m * k * QW + 15) & ~(size_t)15;   // 8-bit tables of one query group
  ix->qtab.ensure((size_t)ngrp * tstride);
  ix->sv_cnt.ensure((size_t)Bq * WF_NSLOT);
  ix->sv_queue.ensure((size_t)Bq * WF_NSLOT * cap);
  ix->fb_tile.ensure((size_t)Bq);
  ix->last_filter_tiles = B;
  // scan_wide's launch shape (wide.hip): ~2048 workgroups, at least 2 row blocks per wave
  auto chunks_of = [&](int blocks, int &nchunks, int &per) {
    nchunks = std::max(1, std::min(ceil_div(2048, B), blocks / (2 * 8)));
    per = ceil_div(blocks, nchunks);
    nchunks = ceil_div(blocks, per);
  };
  int nc_s, per_s, nc_all, per_all;
  chunks_of(sblocks, nc_s, per_s);
  chunks_of(rb_total, nc_all, per_all);
  const int lists_max = std::max(nc_s, nc_all) * 8;
  ix->part_v.ensure((size_t)B * lists_max * keff);
  ix->part_i.ensure((size_t)B * lists_max * keff);

  hipLaunchKernelGGL(wf_reset, dim3(ceil_div(Bq * WF_NSLOT, 256)), dim3(256), 0, st, ix->fb_tile.p, Bq, ix->sv_cnt.p,
                     Bq * WF_NSLOT);
  launch_build_tables_wide(ix->cents.p, ix->from.p, ix->sdim.p, ix->d, m, k, dQ, 0, B, ix->tables.p, st);
  // 1. running lists from the prefix
  launch_scan_wide_range(ix, B, K, from, s_end, rb_begin, sblocks, per_s, nc_s, nullptr, st);
  launch_merge(false, ix->part_v.p, ix->part_i.p, nc_s * 8, (long long)keff, (long long)nc_s * 8 * keff, B, K, nullptr,
               nullptr, nullptr, nullptr, ix->fin_v.p, ix->fin_i.p, st);
  // 2.-4. two filter stages over disjoint row ranges: a short one (~5 % of the rows) against the sample's bound, whose
  // survivors tighten the running lists, then the rest against the lists' new last entries (the byte-code filter's
  // stages, filter.hip: the first stage's survivor rate is ~8 times the second's)
  hipLaunchKernelGGL(wf_table_mins, dim3(m, B), dim3(256), 0, st, ix->tables.p, m, k, ix->qmins.p);
  HIP_CHECK(hipGetLastError());
  const int jp = wf_slice(ix);                                       // quantizers per filter launch
  const size_t lds_bytes = jp < m ? (size_t)jp * k * QW : tstride;
  int cus = 256;
  { int dev = 0; HIP_CHECK(hipGetDevice(&dev)); HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev)); }
  const int resident = std::max(1, (int)(160 * 1024 / lds_bytes));
  const int f_first = f_count >= 64 * WF_NW ? std::max(WF_NW, f_count / 20) : 0;      // blocks of the short stage
  const int stage_begin[2] = {f_begin, f_begin + f_first}, stage_count[2] = {f_first, f_count - f_first};
  for (int sidx = 0; sidx < 2; sidx++) {
    const int sb = stage_begin[sidx], sc = stage_count[sidx];
    if (sc <= 0) continue;
    if (QW == 8)
      hipLaunchKernelGGL(wf_quantize<8>, dim3(ceil_div(k, 256), m, ngrp), dim3(256), 0, st, ix->tables.p, m, k, B, ix->qmins.p,
                         ix->fin_v.p, ix->fin_i.p, keff, ix->qtab.p, tstride, ix->fb_tile.p);
    else
      hipLaunchKernelGGL(wf_quantize<4>, dim3(ceil_div(k, 256), m, ngrp), dim3(256), 0, st, ix->tables.p, m, k, B, ix->qmins.p,
                         ix->fin_v.p, ix->fin_i.p, keff, ix->qtab.p, tstride, ix->fb_tile.p);
    HIP_CHECK(hipGetLastError());
    // every workgroup stages its tables once, so few, long chunks
    int nchunks = std::max(1, std::min(ceil_div(cus * resident * 2, ngrp), sc / (4 * WF_NW)));
    const int per = ceil_div(sc, nchunks);
    nchunks = ceil_div(sc, per);
    auto go = [&](auto kern) {
      HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)lds_bytes));
      hipLaunchKernelGGL(kern, dim3(ngrp, nchunks), dim3(WF_THREADS), lds_bytes, st, ix->wcodes.p, m, k, ix->qtab.p, tstride,
                         from, until, sb, sc, per, ix->sv_cnt.p, ix->sv_queue.p, cap, ix->fb_tile.p, B, 0, 0, (uint32_t *)nullptr);
    };
    const int mq = ceil_div(m, 4);
    if (jp < m) {   // the table in slices of jp quantizers, the byte sums parked in between
      ix->wpark.ensure((size_t)ngrp * sc * 64);
      auto kern = jp <= 4 ? wf_filter<4, 1, true> : wf_filter<4, 2, true>;
      HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)lds_bytes));
      for (int j0 = 0; j0 < m; j0 += jp)
        hipLaunchKernelGGL(kern, dim3(ngrp, nchunks), dim3(WF_THREADS), lds_bytes, st, ix->wcodes.p, m, k, ix->qtab.p, tstride,
                           from, until, sb, sc, per, ix->sv_cnt.p, ix->sv_queue.p, cap, ix->fb_tile.p, B, j0,
                           std::min(m, j0 + jp), ix->wpark.p);
    }
    else if (QW == 8) { if (mq <= 2) go(wf_filter<8, 2>); else if (mq <= 4) go(wf_filter<8, 4>); else go(wf_filter<8, 0>); }
    else              { if (mq <= 2) go(wf_filter<4, 2>); else if (mq <= 4) go(wf_filter<4, 4>); else go(wf_filter<4, 0>); }
    HIP_CHECK(hipGetLastError());
    // survivors, exactly (the kernel empties the queues it has read)
    hipLaunchKernelGGL(wf_survivors, dim3(B), dim3(64 * WF_SV_WAVES), 0, st, ix->wcodes.p, m, k, ix->tables.p, ix->row_base,
                       ix->sv_cnt.p, ix->sv_queue.p, cap, B, keff, ix->fin_v.p, ix->fin_i.p, ix->fb_tile.p);
    HIP_CHECK(hipGetLastError());
  }
  // 5. flagged queries: the exact scan over the whole range replaces their lists
  launch_scan_wide_range(ix, B, K, from, until, rb_begin, rb_total, per_all, nc_all, ix->fb_tile.p, st);
  launch_merge_enabled(ix->part_v.p, ix->part_i.p, nc_all * 8, (long long)keff, (long long)nc_all * 8 * keff, B, K,
                       ix->fin_v.p, ix->fin_i.p, ix->fb_tile.p, 1, st);
  int *flags = d_of;
  if (final_out && replay_enabled() && flags == nullptr) {
    ix->flags_scratch.ensure((size_t)B);
    flags = ix->flags_scratch.p;
  }
  launch_merge(final_out, ix->fin_v.p, ix->fin_i.p, 1, 0LL, (long long)keff, B, K, d_oi, d_od, d_oc, flags, d_pv, d_pi, st);
  if (final_out && replay_enabled()) run_tie_replay(ix, dQ, B, K, from, until, d_oi, d_od, d_oc, flags, st);
}

}  // namespace gulon
