// ADC scan path: Index.prepareQuery -> PQIndex.distances -> TopKHeap -> Result.fromHeap
// (Index.scala:352-440, TopKHeap.scala, Index.scala:83-94), re-designed for gfx950.
//
// Kernels
//   relayout_codes   [m][n] u8 (EncodedMatrix SoA) -> [n/64][ng][64][VEC] row-blocked
//   build_tables     per-query m x 256 distance tables, 4 queries interleaved (float4)
//   scan_kernel      tables in LDS (ds_read_b128 = 4 queries per lookup), codes streamed
//                    coalesced, j-ordered fp32 sums, wavefront-register top-k
//   merge_lists      per-query merge of partial lists (chunks, or GPUs)
#include "scan.hpp"

namespace gulon {

// ---------------------------------------------------------------------------
// code re-layout
// ---------------------------------------------------------------------------
template <int VEC>
__global__ void relayout_codes(const uint8_t *__restrict__ src /*[m][n]*/, int n, int m, int ng,
                               uint8_t *__restrict__ dst, long long total /* nblk*ng*64 */) {
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  int lane = (int)(t & 63);
  long long bg = t >> 6;
  int g = (int)(bg % ng);
  long long rb = bg / ng;
  long long row = rb * 64 + lane;
  uint8_t out[VEC];
#pragma unroll
  for (int b = 0; b < VEC; b++) {
    int j = g * VEC + b;
    out[b] = (row < n && j < m) ? src[(size_t)j * n + row] : (uint8_t)0;
  }
  uint8_t *o = dst + (size_t)t * VEC;
#pragma unroll
  for (int b = 0; b < VEC; b++) o[b] = out[b];
}

// unpack 2/4-bit and 10/12/16-bit Coder layouts to one index per row
// (Coder.scala:110-111,125-126,138-139,163-167)
__global__ void unpack_codes(const uint8_t *__restrict__ packed, int width, int n, int bytes_per_code,
                             int m, uint16_t *__restrict__ out /*[m][n]*/) {
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)m * n) return;
  int j = (int)(t / n);
  int i = (int)(t % n);
  const uint8_t *code = packed + (size_t)j * bytes_per_code;
  int lw = width > 8 ? width - 8 : width;
  int off = width > 8 ? n : 0;
  int lo;
  if (lw == 2) lo = (code[off + (i >> 2)] >> ((i & 3) * 2)) & 0x3;
  else if (lw == 4) lo = (code[off + (i >> 1)] >> ((i & 1) * 4)) & 0xF;
  else lo = code[off + i];
  int v = width > 8 ? ((code[i] << lw) | lo) : lo;
  out[t] = (uint16_t)v;
}

__global__ void narrow_codes(const uint16_t *__restrict__ in, long long total, uint8_t *__restrict__ out) {
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < total) out[t] = (uint8_t)in[t];
}

// ---------------------------------------------------------------------------
// Index.prepareQuery (Index.scala:352-383): T[q][j][c] = sum_t (q[from_j+t]-c[t])^2,
// sequential, unfused.  INTERLEAVED: written as [q/4][j_pad][256][q%4] for the scan;
// otherwise as the reference's [B][m][k].
// ---------------------------------------------------------------------------
template <bool INTERLEAVED, int W>
__global__ void build_tables(const float *__restrict__ cents, const int *__restrict__ from,
                             const int *__restrict__ sdim, int d, int m, int k, int m_pad,
                             const float *__restrict__ Q, int B, float *__restrict__ T,
                             const int *__restrict__ live_queries, float *__restrict__ mins) {
  // one thread per (query group of W, quantizer, centroid); a block of 256 threads = the 256
  // centroids of ONE (query group, quantizer), so the per-(query, quantizer) table minimum the
  // quantized filter needs (NaN entries ignored) is a block reduction here (mins != nullptr)
  // (launched with 256 threads: quantizer and query group are block-uniform, so the query values
  // come through scalar loads; the centroid coordinates are fetched eight at a time)
  if (live_queries) B = min(B, *live_queries);   // device-side query count (tie replay: usually 0)
  const int nqg = (B + W - 1) / W;
  if ((long long)blockIdx.x >= (long long)nqg * m_pad) return;
  const int c = threadIdx.x;
  const int j = blockIdx.x % m_pad;
  const int qg = blockIdx.x / m_pad;
  const long long t = (long long)blockIdx.x * 256 + c;
  float acc[W];
#pragma unroll
  for (int u = 0; u < W; u++) acc[u] = 0.f;
  if (j < m && c < k) {
    const int fr = from[j], s = sdim[j];
    const float *cc = cents + (size_t)k * fr + (size_t)c * s;
    for (int t0 = 0; t0 < s; t0 += 8) {
      float cv[8];
#pragma unroll
      for (int e = 0; e < 8; e++) cv[e] = t0 + e < s ? cc[t0 + e] : 0.f;
#pragma unroll
      for (int e = 0; e < 8; e++) {
        if (t0 + e < s) {
#pragma unroll
          for (int u = 0; u < W; u++) {
            const int q = qg * W + u;
            if (q < B) {
              const float dd = Q[(size_t)q * d + fr + t0 + e] - cv[e];
              acc[u] += dd * dd;
            }
          }
        }
      }
    }
  }
  if (INTERLEAVED && mins != nullptr) {
    __shared__ unsigned smin[W];
    if (threadIdx.x < W) smin[threadIdx.x] = 0x7F800000u;
    __syncthreads();
#pragma unroll
    for (int u = 0; u < W; u++) {
      float x = (c < k && acc[u] == acc[u]) ? acc[u] : INFINITY;   // entries beyond k are not real table entries
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) x = fminf(x, __shfl_xor(x, o));
      if ((threadIdx.x & 63) == 0) atomicMin(&smin[u], __float_as_uint(x));   // entries are >= +0: uint order
    }
    __syncthreads();
    if (threadIdx.x < W) mins[(size_t)(qg * W + threadIdx.x) * m_pad + j] = __uint_as_float(smin[threadIdx.x]);
  }
  if (INTERLEAVED) {
#pragma unroll
    for (int u = 0; u < W; u++) T[(size_t)t * W + u] = acc[u];
  } else if (j < m && c < k) {
#pragma unroll
    for (int u = 0; u < W; u++) {
      int q = qg * W + u;
      if (q < B) T[((size_t)q * m + j) * k + c] = acc[u];
    }
  }
}

// ---------------------------------------------------------------------------
// The scan.  One workgroup = QT = 4*NSUB queries x one chunk of row blocks.
// LDS holds NSUB interleaved tables ([j][c] -> float4 of 4 queries); lane = row;
// every (row, quantizer) costs one ds_read_b128 per sub-table and 4 adds.
// ---------------------------------------------------------------------------
// W queries are interleaved per table entry: W = 4 -> ds_read_b128, 2 -> b64, 1 -> b32.
template <int W> struct TabVec;
template <> struct TabVec<4> { using type = float4; };
template <> struct TabVec<2> { using type = float2; };
template <> struct TabVec<1> { using type = float; };

template <int W, int NSUB, int VEC, int THREADS, bool PRUNE>
__device__ __forceinline__ void scan_body(
    const uint8_t *__restrict__ codes, int ng, int m_pad, const float4 *__restrict__ tables,
    int row_from, int row_until, int row_base, int rb_begin, int e_count, int e_per_chunk, RbMap mp, int nchunks,
    int keff, float *__restrict__ part_v, int *__restrict__ part_i, unsigned *__restrict__ gtau, int tau_off4,
    int prune_from, const float *__restrict__ lbv, const int *__restrict__ lbi,
    const uint8_t *__restrict__ perm /* non-null: `codes` is the conflict-ordered copy (conflict_order.hip), perm its row order */,
    const int tile /* query tile of this pass */, unsigned long long *__restrict__ dbg) {
  constexpr int QT = W * NSUB;
  // optional timeline (GULON_SCAN_TIMELINE=1): 4 stamps per workgroup, 100 MHz wall clock
  if (dbg && threadIdx.x == 0) dbg[(blockIdx.y * gridDim.x + blockIdx.x) * 4 + 0] = wall_clock64();
  constexpr int NW = THREADS / 64;
  using Word = typename CodeWord<VEC>::type;
  using TV = typename TabVec<W>::type;
  extern __shared__ float4 lds_raw[];
  const TV *lds = reinterpret_cast<const TV *>(lds_raw);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int chunk = blockIdx.y;
  const int tab_entries = m_pad * 256;  // entries (of W floats) per sub-table

  // Shared pruning thresholds, as the bits of non-negative floats (so unsigned min == float
  // min): tau_sh[q] >= the (K+1)-th smallest distance of query q over everything scanned so
  // far by ANY wave of ANY workgroup (gtau is the cross-workgroup copy).  A row whose
  // distance exceeds it can never be in the final top-(K+1), so it is dropped before the
  // (expensive, wave-serial) insertion.  Only speed depends on how fresh these values are.
  unsigned *tau_sh = reinterpret_cast<unsigned *>(lds_raw + tau_off4);
  if (tid < QT) tau_sh[tid] = __hip_atomic_load(&gtau[tile * QT + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

  // stage the NSUB sub-tables of this query tile
  {
    const int n16 = NSUB * tab_entries * W / 4;   // float4 units
    const float4 *src = tables + (size_t)tile * n16;
    for (int e = tid; e < n16; e += THREADS) lds_raw[e] = src[e];
  }
  __syncthreads();

  if (dbg && threadIdx.x == 0) dbg[(blockIdx.y * gridDim.x + blockIdx.x) * 4 + 1] = wall_clock64();
  WaveList wl[QT];
  int cnt[QT];
#pragma unroll
  for (int q = 0; q < QT; q++) { wl[q].init(); cnt[q] = 0; }

  // large-K "peeling" rounds: only entries strictly after (lbq, lbiq) in (distance, row id)
  // order are eligible (the earlier ones were returned by previous rounds)
  const bool peel = lbv != nullptr;
  float lbq[QT];
  int lbiq[QT];
#pragma unroll
  for (int q = 0; q < QT; q++) {
    lbq[q] = peel ? lbv[tile * QT + q] : -1.f;
    lbiq[q] = peel ? lbi[tile * QT + q] : -1;
  }

  // this chunk's share of the eligible row blocks, in e-space (RbMap); (mp_p, mp_r) tracks
  // e = mp_p * width + mp_r without a division per row block
  const int e0 = chunk * e_per_chunk;
  const int e1 = min(e_count, e0 + e_per_chunk);
  const Word *cw = reinterpret_cast<const Word *>(codes);
  int mp_p = (e0 + wave) / mp.width, mp_r = (e0 + wave) - mp_p * mp.width;
  auto block_of = [&](int p, int r) { return rb_begin + p * mp.period + mp.lo + r; };
  auto advance = [&](int &p, int &r) { r += NW; while (r >= mp.width) { r -= mp.width; p++; } };

  // software pipeline: the first code word of the NEXT row block is in flight while this
  // one is being looked up (global latency would otherwise idle the LDS pipe)
  Word w_first{};
  if (e0 + wave < e1) w_first = cw[((size_t)block_of(mp_p, mp_r) * ng) * 64 + lane];
  for (int e = e0 + wave; e < e1; e += NW) {
    const int rb = block_of(mp_p, mp_r);
    advance(mp_p, mp_r);
    float acc[QT];
#pragma unroll
    for (int q = 0; q < QT; q++) acc[q] = 0.f;

    const Word *p = cw + ((size_t)rb * ng) * 64 + lane;
    Word w = w_first;
    if (e + NW < e1) w_first = cw[((size_t)block_of(mp_p, mp_r) * ng) * 64 + lane];

    // workgroup-shared thresholds (wave-uniform LDS reads); <= keeps exact-tie candidates
    float tsh[QT];
#pragma unroll
    for (int q = 0; q < QT; q++)
      tsh[q] = __uint_as_float(__hip_atomic_load(&tau_sh[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));

    // Exact early termination.  Every table entry is a sum of squares (>= 0) and binary32
    // addition of a non-negative term is monotone, so a j-ordered PARTIAL sum is a lower
    // bound of the full distance.  Once all 64 rows x 4 queries of a sub-table are already
    // above their thresholds, none of them can enter a top-(K+1) list and the remaining
    // look-ups of that sub-table are skipped; rows that survive still get bit-exact sums.
    bool live[NSUB];
#pragma unroll
    for (int s = 0; s < NSUB; s++) live[s] = true;

    for (int g = 0; g < ng; g++) {
      Word wn = w;
      if (g + 1 < ng) wn = p[(size_t)(g + 1) * 64];
      const TV *tj = lds + g * VEC * 256;
#pragma unroll
      for (int seg = 0; seg < VEC / 4; seg++) {
        if (PRUNE && g * VEC + seg * 4 >= prune_from) {
#pragma unroll
          for (int s = 0; s < NSUB; s++)
            if (live[s]) {
              bool in = false;
#pragma unroll
              for (int u = 0; u < W; u++) in = in || acc[W * s + u] <= tsh[W * s + u];
              live[s] = __ballot(in) != 0ull;
            }
        }
#pragma unroll
        for (int s = 0; s < NSUB; s++) {
          if (!PRUNE || live[s]) {
#pragma unroll
            for (int bb = 0; bb < 4; bb++) {
              const int b = seg * 4 + bb;
              uint32_t c = code_byte<VEC>(w, b);
              TV t = tj[b * 256 + c + s * tab_entries];
              const float *tf = reinterpret_cast<const float *>(&t);
#pragma unroll
              for (int u = 0; u < W; u++) acc[W * s + u] += tf[u];
            }
          }
        }
      }
      w = wn;
    }

    const int row = rb * 64 + lane;
    // (ordered copy: which row a lane holds is looked up only when some lane has a candidate)
    const bool valid = perm != nullptr || (row >= row_from && row < row_until);
    unsigned long long masks[QT];
    unsigned long long any = 0;
#pragma unroll
    for (int q = 0; q < QT; q++) {
      unsigned long long mk = (!PRUNE || live[q / W]) ? __ballot(valid && acc[q] <= tsh[q]) : 0ull;
      masks[q] = mk;
      any |= mk;
    }
    if (any) {
      int place = lane;
      unsigned long long in_range = ~0ull;
      if (perm) {
        place = perm[(size_t)rb * 64 + lane];
        const int prow = rb * 64 + place;
        in_range = __ballot(prow >= row_from && prow < row_until);
      }
#pragma unroll
      for (int q = 0; q < QT; q++) {
        unsigned long long mk = masks[q] & in_range;
        while (mk) {
          int l = __ffsll((long long)mk) - 1;
          mk &= mk - 1;
          float cv = readlane_f(acc[q], l);
          int cr = rb * 64 + readlane_i(place, l) + row_base;
          if (peel && !(cv > lbq[q] || (cv == lbq[q] && cr > lbiq[q]))) continue;
          if (cnt[q] < keff || wl[q].accepts(cv, cr)) {
            wl[q].insert(cv, cr, keff, lane);
            if (cnt[q] < keff) cnt[q]++;
            if (wl[q].tau < tsh[q]) {   // this wave's list is full and tighter: publish
              tsh[q] = wl[q].tau;
              if (lane == 0)
                __hip_atomic_fetch_min(&tau_sh[q], __float_as_uint(wl[q].tau), __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_WORKGROUP);
            }
          }
        }
      }
    }
    // every 32 row blocks one wave trades thresholds with the other workgroups of this tile
    if (wave == 0 && (((e - e0) / NW) & 31) == 31 && lane < QT) {
      unsigned mine = __hip_atomic_load(&tau_sh[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      unsigned old = __hip_atomic_fetch_min(&gtau[tile * QT + lane], mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (old < mine)
        __hip_atomic_fetch_min(&tau_sh[lane], old, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }

  // merge the NW per-wave lists of every query through LDS (tables are dead now)
  __syncthreads();
  if (dbg && threadIdx.x == 0) dbg[(blockIdx.y * gridDim.x + blockIdx.x) * 4 + 2] = wall_clock64();
  float *sv = reinterpret_cast<float *>(lds_raw);
  int *si = reinterpret_cast<int *>(sv + QT * NW * 64);
#pragma unroll
  for (int q = 0; q < QT; q++) {
    sv[(q * NW + wave) * 64 + lane] = wl[q].v;
    si[(q * NW + wave) * 64 + lane] = wl[q].i;
  }
  __syncthreads();
  for (int q = wave; q < QT; q += NW) {
    WaveList out;
    out.init();
    for (int w2 = 0; w2 < NW; w2++) {
      for (int e = 0; e < keff; e++) {
        float cv = sv[(q * NW + w2) * 64 + e];
        int cr = si[(q * NW + w2) * 64 + e];
        if (cr == INT_MAX) break;  // sorted: the rest of this list is padding
        if (out.accepts(cv, cr)) out.insert(cv, cr, keff, lane);
      }
    }
    if (lane < keff) {
      size_t o = ((size_t)(tile * QT + q) * nchunks + chunk) * keff + lane;
      part_v[o] = out.v;
      part_i[o] = out.i;
    }
    if (lane == 0 && out.tau < INFINITY)
      __hip_atomic_fetch_min(&gtau[tile * QT + q], __float_as_uint(out.tau), __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
  }
  if (dbg && threadIdx.x == 0) dbg[(blockIdx.y * gridDim.x + blockIdx.x) * 4 + 3] = wall_clock64();
}

#define GULON_SCAN_PARAMS                                                                                          \
  const uint8_t *__restrict__ codes, int ng, int m_pad, const float4 *__restrict__ tables, int row_from,           \
      int row_until, int row_base, int rb_begin, int e_count, int e_per_chunk, RbMap mp, int nchunks, int keff,     \
      float *__restrict__ part_v, int *__restrict__ part_i, unsigned *__restrict__ gtau, int tau_off4,              \
      int prune_from, const float *__restrict__ lbv, const int *__restrict__ lbi, const uint8_t *__restrict__ perm
#define GULON_SCAN_ARGS                                                                                            \
  codes, ng, m_pad, tables, row_from, row_until, row_base, rb_begin, e_count, e_per_chunk, mp, nchunks, keff,      \
      part_v, part_i, gtau, tau_off4, prune_from, lbv, lbi, perm

// workgroup (x, y) = query tile x (fastest: the tiles of one chunk run together) x chunk y of the row blocks
template <int W, int NSUB, int VEC, int THREADS, bool PRUNE>
__global__ __launch_bounds__(THREADS) void scan_kernel(GULON_SCAN_PARAMS, unsigned long long *__restrict__ dbg) {
  scan_body<W, NSUB, VEC, THREADS, PRUNE>(GULON_SCAN_ARGS, (int)blockIdx.x, dbg);
}

// The filter's fallback launch: only the query tiles flagged in tile_enable (one flag per tile_div tiles) are
// scanned, normally none.  It sits on the critical path of its batch while ANOTHER batch's filter kernel
// fills the chip, and a workgroup is dispatched only where a leaving filter workgroup makes room: 64 KiB of
// LDS and, per SIMD, 4 x 56 + 64 = 288 registers.  Four waves of the regular one-sub-table instantiation
// (84 VGPRs) need 352 and waited for a completely empty CU -- for the other batch's whole 3 ms kernel --, so
// this twin is compiled for 7 waves per SIMD (<= 72 VGPRs); and since even then every workgroup waits for
// its turn (256 empty workgroups: 0.9 ms on average), the launch is `gridDim.x` workgroups wide, each
// looping over the tiles x, x + gridDim.x, ...: narrow while nothing falls back, wide once something did
// (*hint, host-mapped, read by the host when it sizes the next launch).
template <int W, int NSUB, int VEC, int THREADS, bool PRUNE>
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(7))) void scan_kernel_lean(
    GULON_SCAN_PARAMS, const int *__restrict__ tile_enable, int tile_div, int ntiles, int *__restrict__ hint) {
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    if (tile_enable[tile / tile_div] == 0) continue;
    if (hint && threadIdx.x == 0 && blockIdx.y == 0) *hint = 1;
    scan_body<W, NSUB, VEC, THREADS, PRUNE>(GULON_SCAN_ARGS, tile, nullptr);
    __syncthreads();   // every wave is done with this tile's tables and thresholds in LDS
  }
}
#undef GULON_SCAN_PARAMS
#undef GULON_SCAN_ARGS

// ---------------------------------------------------------------------------
// merge `lists` sorted partial lists per query; one wave per query.
// in: element (l, q, e) at  l*stride_l + q*stride_q + e.
// FINAL: write idx/dist/count/flags [B][K]; else write a (K+1)-list [B][keff].
// ---------------------------------------------------------------------------
template <bool FINAL>
__global__ __launch_bounds__(64) void merge_lists(const float *__restrict__ in_v, const int *__restrict__ in_i,
                                                  int lists, long long stride_l, long long stride_q,
                                                  int B, int K, int keff, int *__restrict__ out_idx,
                                                  float *__restrict__ out_dist, int *__restrict__ out_count,
                                                  int *__restrict__ out_flags, float *__restrict__ out_pv,
                                                  int *__restrict__ out_pi, const int *__restrict__ tile_enable,
                                                  int qt) {
  const int q = blockIdx.x;
  const int lane = threadIdx.x;
  if (q >= B) return;
  if (tile_enable && tile_enable[q / qt] == 0) return;
  WaveList wl;
  wl.init();
  const int total = lists * keff;
  for (int base = 0; base < total; base += 64) {
    int e = base + lane;
    float cv = INFINITY;
    int cr = INT_MAX;
    if (e < total) {
      int l = e / keff, x = e - l * keff;
      size_t o = (size_t)l * stride_l + (size_t)q * stride_q + x;
      cv = in_v[o];
      cr = in_i[o];
    }
    unsigned long long mk = __ballot(cr != INT_MAX && wl.accepts(cv, cr));
    while (mk) {
      int l = __ffsll((long long)mk) - 1;
      mk &= mk - 1;
      float v = readlane_f(cv, l);
      int r = readlane_i(cr, l);
      if (wl.accepts(v, r)) wl.insert(v, r, keff, lane);
    }
  }
  if (FINAL) {
    int live = __popcll(__ballot(lane < K && wl.i != INT_MAX));
    if (lane < K) {
      bool ok = wl.i != INT_MAX;
      out_idx[(size_t)q * K + lane] = ok ? wl.i : -1;
      out_dist[(size_t)q * K + lane] = ok ? wl.v : INFINITY;
    }
    float nv = __shfl_down(wl.v, 1);
    int ni = __shfl_down(wl.i, 1);
    bool tie = wl.i != INT_MAX && ni != INT_MAX && wl.v == nv && lane + 1 < keff;
    unsigned long long tm = __ballot(tie);
    if (lane == 0) {
      if (out_count) out_count[q] = live;
      if (out_flags) {
        int f = 0;
        if (K >= 1 && ((tm >> (K - 1)) & 1ull)) f |= GULON_FLAG_BOUNDARY_TIE;
        if (K >= 2 && (tm & ((1ull << (K - 1)) - 1ull))) f |= GULON_FLAG_INTERIOR_TIE;
        out_flags[q] = f;
      }
    }
  } else {
    if (lane < keff) {
      out_pv[(size_t)q * keff + lane] = wl.v;
      out_pi[(size_t)q * keff + lane] = wl.i;
    }
  }
}


void launch_merge(bool final_out, const float *in_v, const int *in_i, int lists, long long stride_l,
                  long long stride_q, int B, int K, int *out_idx, float *out_dist, int *out_count,
                  int *out_flags, float *out_pv, int *out_pi, hipStream_t st) {
  if (B <= 0) return;
  int keff = K + 1;
  if (final_out)
    hipLaunchKernelGGL(merge_lists<true>, dim3(B), dim3(64), 0, st, in_v, in_i, lists, stride_l, stride_q, B, K,
                       keff, out_idx, out_dist, out_count, out_flags, out_pv, out_pi, (const int *)nullptr, 1);
  else
    hipLaunchKernelGGL(merge_lists<false>, dim3(B), dim3(64), 0, st, in_v, in_i, lists, stride_l, stride_q, B, K,
                       keff, out_idx, out_dist, out_count, out_flags, out_pv, out_pi, (const int *)nullptr, 1);
  HIP_CHECK(hipGetLastError());
}

void launch_merge_enabled(const float *in_v, const int *in_i, int lists, long long stride_l, long long stride_q, int B,
                          int K, float *out_pv, int *out_pi, const int *tile_enable, int qt, hipStream_t st) {
  if (B <= 0) return;
  hipLaunchKernelGGL(merge_lists<false>, dim3(B), dim3(64), 0, st, in_v, in_i, lists, stride_l, stride_q, B, K, K + 1,
                     (int *)nullptr, (float *)nullptr, (int *)nullptr, (int *)nullptr, out_pv, out_pi, tile_enable, qt);
  HIP_CHECK(hipGetLastError());
}

// ---- large K: peel the result 64 entries at a time (each round = one scan restricted to the
// ---- entries after the previous round's last one) -------------------------------------
__global__ __launch_bounds__(64) void peel_update(const float *__restrict__ tv, const int *__restrict__ ti, int round,
                                                  int cap, float *__restrict__ pv, int *__restrict__ pi,
                                                  float *__restrict__ lbv, int *__restrict__ lbi) {
  const int q = blockIdx.x, lane = threadIdx.x;
  const float v = tv[(size_t)q * 64 + lane];
  const int i = ti[(size_t)q * 64 + lane];
  pv[(size_t)q * cap + round * 64 + lane] = v;
  pi[(size_t)q * cap + round * 64 + lane] = i;
  if (lane == 63) {
    if (i != INT_MAX) { lbv[q] = v; lbi[q] = i; }
    else { lbv[q] = INFINITY; lbi[q] = INT_MAX; }   // list exhausted: nothing is eligible any more
  }
}

__global__ void peel_finalize(const float *__restrict__ pv, const int *__restrict__ pi, int B, int cap, int K,
                              int *__restrict__ out_idx, float *__restrict__ out_dist, int *__restrict__ out_count,
                              int *__restrict__ out_flags) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= B) return;
  const float *v = pv + (size_t)q * cap;
  const int *id = pi + (size_t)q * cap;
  int live = 0, flags = 0;
  for (int e = 0; e < K; e++) {
    const bool ok = id[e] != INT_MAX;
    out_idx[(size_t)q * K + e] = ok ? id[e] : -1;
    out_dist[(size_t)q * K + e] = ok ? v[e] : INFINITY;
    live += ok ? 1 : 0;
    if (ok && e + 1 < cap && id[e + 1] != INT_MAX && v[e] == v[e + 1])
      flags |= (e == K - 1) ? GULON_FLAG_BOUNDARY_TIE : GULON_FLAG_INTERIOR_TIE;
  }
  if (out_count) out_count[q] = live;
  if (out_flags) out_flags[q] = flags;
}

// the peeled list as a partial (K+1)-list for the multi-GPU merge: [B][cap] -> [B][K+1]
__global__ void peel_export(const float *__restrict__ pv, const int *__restrict__ pi, int B, int cap, int keff,
                            float *__restrict__ out_v, int *__restrict__ out_i) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)B * keff) return;
  const int q = (int)(t / keff), e = (int)(t - (long long)q * keff);
  out_v[t] = pv[(size_t)q * cap + e];
  out_i[t] = pi[(size_t)q * cap + e];
}

// Merge of two ascending (distance, row id) lists of `len` entries per query (padded with (+inf, INT_MAX)) into
// the `len` smallest, ascending: every entry's position = its own index + its rank in the other list (row ids
// are distinct across lists, so the order is strict).  b == nullptr: a copies through.
__global__ __launch_bounds__(256) void merge2_long(const float *__restrict__ av, const int *__restrict__ ai,
                                                   const float *__restrict__ bv, const int *__restrict__ bi, int len,
                                                   float *__restrict__ ov, int *__restrict__ oi) {
  const int q = blockIdx.x;
  const float *A = av + (size_t)q * len, *Bv = bv ? bv + (size_t)q * len : nullptr;
  const int *Ai = ai + (size_t)q * len, *Bi = bi ? bi + (size_t)q * len : nullptr;
  float *O = ov + (size_t)q * len;
  int *Oi = oi + (size_t)q * len;
  auto less = [](float v0, int i0, float v1, int i1) { return v0 < v1 || (v0 == v1 && i0 < i1); };
  for (int e = threadIdx.x; e < len; e += blockDim.x) {
    if (!Bv) { O[e] = A[e]; Oi[e] = Ai[e]; continue; }
    {   // A[e]: entries of B before it
      const float v = A[e]; const int id = Ai[e];
      int lo = 0, hi = len;
      while (lo < hi) { const int mid = (lo + hi) >> 1; if (less(Bv[mid], Bi[mid], v, id)) lo = mid + 1; else hi = mid; }
      const int pos = e + lo;
      if (pos < len && id != INT_MAX) { O[pos] = v; Oi[pos] = id; }
    }
    {   // B[e]: entries of A before it
      const float v = Bv[e]; const int id = Bi[e];
      int lo = 0, hi = len;
      while (lo < hi) { const int mid = (lo + hi) >> 1; if (less(A[mid], Ai[mid], v, id)) lo = mid + 1; else hi = mid; }
      const int pos = e + lo;
      if (pos < len && id != INT_MAX) { O[pos] = v; Oi[pos] = id; }
    }
  }
}
__global__ void fill_list(float *__restrict__ v, int *__restrict__ i, long long n) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) { v[t] = INFINITY; i[t] = INT_MAX; }
}

// `lists` partial (K+1)-lists per query (element (l, q, e) at l*stride_l + q*keff + e) -> final [B][K] + count + flags,
// for K beyond the 63 a wavefront list holds: a tree of pairwise merges through device scratch
void launch_merge_long(const float *in_v, const int *in_i, int lists, long long stride_l, int B, int K, int *out_idx,
                       float *out_dist, int *out_count, int *out_flags, hipStream_t st) {
  const int keff = K + 1;
  const size_t per = (size_t)B * keff;
  int width = lists;                       // lists alive in the current round
  DevBuf<float> va, vb;
  DevBuf<int> ia, ib;
  const size_t half = (size_t)((lists + 1) / 2);
  va.alloc(std::max<size_t>(half * per, 1)); ia.alloc(std::max<size_t>(half * per, 1));
  vb.alloc(std::max<size_t>(((half + 1) / 2) * per, 1)); ib.alloc(std::max<size_t>(((half + 1) / 2) * per, 1));
  const float *cv = in_v;
  const int *ci = in_i;
  long long cs = stride_l;
  bool to_a = true;
  if (width == 1) {   // a single list: straight to the finaliser
    launch_peel_finalize(cv, ci, B, keff, K, out_idx, out_dist, out_count, out_flags, st);
    HIP_CHECK(hipStreamSynchronize(st));
    return;
  }
  while (width > 1) {
    const int next = (width + 1) / 2;
    float *ov = to_a ? va.p : vb.p;
    int *oi = to_a ? ia.p : ib.p;
    hipLaunchKernelGGL(fill_list, dim3((unsigned)ceil_div((long long)next * per, 256LL)), dim3(256), 0, st, ov, oi,
                       (long long)next * per);
    for (int p = 0; p < next; p++) {
      const bool pair = 2 * p + 1 < width;
      hipLaunchKernelGGL(merge2_long, dim3(B), dim3(256), 0, st, cv + (size_t)(2 * p) * cs, ci + (size_t)(2 * p) * cs,
                         pair ? cv + (size_t)(2 * p + 1) * cs : nullptr, pair ? ci + (size_t)(2 * p + 1) * cs : nullptr, keff,
                         ov + (size_t)p * per, oi + (size_t)p * per);
    }
    HIP_CHECK(hipGetLastError());
    cv = ov; ci = oi; cs = (long long)per;
    width = next;
    to_a = !to_a;
  }
  launch_peel_finalize(cv, ci, B, keff, K, out_idx, out_dist, out_count, out_flags, st);
  HIP_CHECK(hipStreamSynchronize(st));   // the scratch lists above go out of scope
}

void launch_peel_update(const float *tv, const int *ti, int B, int round, int cap, float *pv, int *pi, float *lbv,
                        int *lbi, hipStream_t st) {
  hipLaunchKernelGGL(peel_update, dim3(B), dim3(64), 0, st, tv, ti, round, cap, pv, pi, lbv, lbi);
  HIP_CHECK(hipGetLastError());
}
void launch_peel_finalize(const float *pv, const int *pi, int B, int cap, int K, int *out_idx, float *out_dist,
                          int *out_count, int *out_flags, hipStream_t st) {
  hipLaunchKernelGGL(peel_finalize, dim3(ceil_div(B, 64)), dim3(64), 0, st, pv, pi, B, cap, K, out_idx, out_dist,
                     out_count, out_flags);
  HIP_CHECK(hipGetLastError());
}

void launch_build_tables(int W, gulon_index *ix, const float *dQ, int B, int Bpad, float *tables, hipStream_t st,
                         const int *live_queries, float *mins) {
  long long total = (long long)(Bpad / W) * ix->m_pad * 256;
  if (total <= 0) return;
#define BT(WW)                                                                                             \
  hipLaunchKernelGGL((build_tables<true, WW>), dim3(ceil_div(total, 256)), dim3(256), 0, st, ix->cents.p,   \
                     ix->from.p, ix->sdim.p, ix->d, ix->m, ix->k, ix->m_pad, dQ, B, tables, live_queries, mins)
  if (W == 4) BT(4); else if (W == 2) BT(2); else BT(1);
#undef BT
  HIP_CHECK(hipGetLastError());
}

ScanTuning::ScanTuning() {
  static const char *keys[] = {"GULON_SCAN_BLOCKS", "GULON_SCAN_PRUNE", "GULON_SCAN_PRUNE_FROM", "GULON_SCAN_FILTER",
                               "GULON_FILTER_MIN_RB", "GULON_FILTER_PERIOD", "GULON_FILTER_STAGE1", "GULON_FILTER_CAP",
                               "GULON_FILTER_NADD", "GULON_FILTER_SAMPLE", "GULON_FILTER_STAGE0", "GULON_FILTER_BLOCKS",
                               "GULON_FILTER_SHARED_STAGE1", "GULON_FILTER_ORDER"};
  for (const char *k : keys)
    if (const char *e = getenv(k)) set(k, atoi(e));
}

bool ScanTuning::set(const char *key, int v) {
  std::string k(key);
  if (k == "GULON_SCAN_BLOCKS") { if (v >= 1) target_blocks = v; }
  else if (k == "GULON_SCAN_PRUNE") prune = v != 0;
  else if (k == "GULON_SCAN_PRUNE_FROM") prune_from = v;
  else if (k == "GULON_SCAN_FILTER") filter = v != 0;
  else if (k == "GULON_FILTER_MIN_RB") filter_min_rb = v < 4 ? 4 : v;
  else if (k == "GULON_FILTER_PERIOD") { if (v >= 3) filter_period = v; }
  else if (k == "GULON_FILTER_STAGE1") { if (v >= 1) filter_stage1 = v; }
  else if (k == "GULON_FILTER_CAP") { if (v >= 64) filter_cap = v; }
  else if (k == "GULON_FILTER_NADD") { if (v == 0 || v == 2 || v == 4 || v == 8) filter_nadd = v; }   // (8: experiment builds only)
  else if (k == "GULON_FILTER_SAMPLE") { if (v >= 1) filter_sample = v; }
  else if (k == "GULON_FILTER_STAGE0") { if (v >= 0) filter_stage0 = v; }
  else if (k == "GULON_FILTER_BLOCKS") { if (v >= 1) filter_blocks = v; }
  else if (k == "GULON_FILTER_SHARED_STAGE1") { if (v >= -1 && v <= 1) filter_shared_stage1 = v; }
  else if (k == "GULON_FILTER_ORDER") { if (v >= 0 && v <= 8) filter_order = v; }
  else return false;
  return true;
}

// The knobs have no process-wide mutable state: a handle takes its settings from the ENVIRONMENT when it is created
// (ScanTuning's constructor) and keeps them; gulon_index_tuning changes one handle.  (A null handle: the compiled-in defaults.)
const ScanTuning &tuning_defaults() { static const ScanTuning t; return t; }

bool replay_enabled() {
  static const bool on = [] { const char *e = getenv("GULON_TIE_REPLAY"); return !(e && atoi(e) == 0); }();
  return on;
}

}  // namespace gulon

using namespace gulon;

// ---------------------------------------------------------------------------
// index handle
// ---------------------------------------------------------------------------
namespace {

constexpr size_t LDS_BUDGET = 144 * 1024;

template <int W, int NSUB, int VEC, int SCAN_THREADS, bool PRUNE>
void launch_scan_p(gulon_index *ix, int ntiles, int nchunks, int rb_begin, int e_count, int e_per_chunk, RbMap mp,
                   int from, int until, int keff, hipStream_t st, const float *lbv, const int *lbi,
                   const int *tile_enable, int tile_div, int grid_x, int *hint) {
  size_t lds_bytes = (size_t)NSUB * ix->m_pad * 256 * W * sizeof(float);
  size_t merge_bytes = (size_t)W * NSUB * (SCAN_THREADS / 64) * 64 * 8;
  if (merge_bytes > lds_bytes) lds_bytes = merge_bytes;
  const int tau_off4 = (int)(lds_bytes / sizeof(float4));   // shared thresholds live after tables/merge area
  lds_bytes += 64;
  int prune_from = tuning_of(ix).prune_from >= 0 ? tuning_of(ix).prune_from : ix->m_pad / 2;
  if (prune_from < 4) prune_from = 4;
  unsigned long long *dbg = nullptr;
  if (getenv("GULON_SCAN_TIMELINE") && !tile_enable) {
    ix->dbg.ensure((size_t)ntiles * nchunks * 4);
    dbg = ix->dbg.p;
  }
  if (tile_enable) {   // the filter's fallback (launch_scan with one_sub): grid_x workgroups loop over the tiles
    if constexpr (NSUB == 1) {
      auto kern = scan_kernel_lean<W, NSUB, VEC, SCAN_THREADS, PRUNE>;
      HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)lds_bytes));
      hipLaunchKernelGGL(kern, dim3(std::max(1, std::min(grid_x, ntiles)), nchunks), dim3(SCAN_THREADS), lds_bytes, st,
                         ix->codes.p, ix->ng, ix->m_pad, reinterpret_cast<const float4 *>(ix->tables.p), from, until,
                         ix->row_base, rb_begin, e_count, e_per_chunk, mp, nchunks, keff, ix->part_v.p, ix->part_i.p,
                         ix->gtau.p, tau_off4, prune_from, lbv, lbi, (const uint8_t *)nullptr, tile_enable, tile_div, ntiles, hint);
      HIP_CHECK(hipGetLastError());
      return;
    } else {
      GULON_REQUIRE(false, "internal: tile-enabled scans use one sub-table");
    }
  }
  auto kern = scan_kernel<W, NSUB, VEC, SCAN_THREADS, PRUNE>;
  HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds_bytes));
  // one-word codes: the exact scan's table gathers are the same ds_read_b128 of 256 x 16-byte entries per quantizer as
  // the filter's, so the conflict-ordered copy serves it too (13 of its 16 quantizers were ordered for)
  // (the exact scan walks row blocks one at a time: it reads the ordered copy only where a block holds its own rows)
  const bool ordered = VEC == 16 && ix->ng == 1 && ix->fcodes.p && ix->fwindow == 1 && tuning_of(ix).filter_order > 0;
  hipLaunchKernelGGL(kern, dim3(ntiles, nchunks), dim3(SCAN_THREADS), lds_bytes, st, ordered ? ix->fcodes.p : ix->codes.p,
                     ix->ng, ix->m_pad,
                     reinterpret_cast<const float4 *>(ix->tables.p), from, until, ix->row_base, rb_begin, e_count,
                     e_per_chunk, mp, nchunks, keff, ix->part_v.p, ix->part_i.p, ix->gtau.p, tau_off4, prune_from, lbv,
                     lbi, ordered ? ix->fperm.p : (const uint8_t *)nullptr, dbg);
  HIP_CHECK(hipGetLastError());
  if (dbg) {   // debugging aid: synchronous dump of the per-workgroup timeline
    HIP_CHECK(hipStreamSynchronize(st));
    const size_t nb = (size_t)ntiles * nchunks;
    std::vector<unsigned long long> h(nb * 4);
    HIP_CHECK(hipMemcpy(h.data(), dbg, sizeof(unsigned long long) * nb * 4, hipMemcpyDeviceToHost));
    unsigned long long t0 = ~0ull, t1 = 0;
    double stage = 0, loop = 0, merge = 0;
    for (size_t b = 0; b < nb; b++) {
      t0 = std::min(t0, h[b * 4]); t1 = std::max(t1, h[b * 4 + 3]);
      stage += (double)(h[b * 4 + 1] - h[b * 4]); loop += (double)(h[b * 4 + 2] - h[b * 4 + 1]);
      merge += (double)(h[b * 4 + 3] - h[b * 4 + 2]);
    }
    std::vector<double> ends;
    for (size_t b = 0; b < nb; b++) ends.push_back((double)(h[b * 4 + 3] - t0) / 100.0);
    std::sort(ends.begin(), ends.end());
    fprintf(stderr, "[scan timeline] %zu workgroups, kernel span %.1f us; per workgroup: stage %.1f us, loop %.1f us, "
            "merge %.1f us; last-finish quantiles 50%%=%.0f 90%%=%.0f 99%%=%.0f 100%%=%.0f us\n", nb,
            (double)(t1 - t0) / 100.0, stage / nb / 100.0, loop / nb / 100.0, merge / nb / 100.0,
            ends[nb / 2], ends[nb * 9 / 10], ends[nb * 99 / 100], ends[nb - 1]);
  }
}

template <int W, int NSUB, int VEC>
void launch_scan_t(gulon_index *ix, int ntiles, int nchunks, int rb_begin, int e_count, int e_per_chunk, RbMap mp,
                   int from, int until, int keff, hipStream_t st, const float *lbv, const int *lbi,
                   const int *tile_enable, int tile_div, int grid_x, int *hint) {
  constexpr int TH = 1024;
  if (tuning_of(ix).prune)
    launch_scan_p<W, NSUB, VEC, TH, true>(ix, ntiles, nchunks, rb_begin, e_count, e_per_chunk, mp, from, until, keff,
                                          st, lbv, lbi, tile_enable, tile_div, grid_x, hint);
  else
    launch_scan_p<W, NSUB, VEC, TH, false>(ix, ntiles, nchunks, rb_begin, e_count, e_per_chunk, mp, from, until, keff,
                                           st, lbv, lbi, tile_enable, tile_div, grid_x, hint);
}

}  // namespace

namespace gulon {
void launch_scan(gulon_index *ix, int ntiles, int nchunks, int rb_begin, int e_count, int e_per_chunk, RbMap mp,
                 int from, int until, int keff, hipStream_t st, const float *lbv, const int *lbi,
                 const int *tile_enable, bool one_sub, int grid_x, int *hint) {
  // tile_enable (the filter's fallback) comes with one_sub: W-query tiles (one sub-table, <= 64 KiB of LDS)
  // whatever the index prefers; ntiles counts THOSE tiles, tile_enable keeps one flag per ix->nsub of them,
  // and grid_x workgroups per chunk loop over them
  GULON_REQUIRE((tile_enable != nullptr) == one_sub, "internal: tile_enable <=> one_sub");
  const int nsub = one_sub ? 1 : ix->nsub;
  const int tile_div = one_sub ? ix->nsub : 1;
#define GO(WW, NS, V)                                                                                          \
  launch_scan_t<WW, NS, V>(ix, ntiles, nchunks, rb_begin, e_count, e_per_chunk, mp, from, until, keff, st, lbv, lbi, \
                           tile_enable, tile_div, grid_x, hint)
  if (ix->w == 4) {
    if (ix->vec == 16) {
      if (nsub == 4) GO(4, 4, 16); else if (nsub == 2) GO(4, 2, 16); else GO(4, 1, 16);
    } else {
      if (nsub == 4) GO(4, 4, 4); else if (nsub == 2) GO(4, 2, 4); else GO(4, 1, 4);
    }
  } else if (ix->w == 2) {
    if (ix->vec == 16) GO(2, 1, 16); else GO(2, 1, 4);
  } else {
    if (ix->vec == 16) GO(1, 1, 16); else GO(1, 1, 4);
  }
#undef GO
}
}  // namespace gulon

namespace {

// Enqueue table build + scan + merge.  Exactly one of (final outputs) / (partial outputs) is used.
__global__ void fill_f32(float *__restrict__ p, long long n, float v) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

void run_query(gulon_index *ix, const float *dQ, int B, int K, int from, int until, bool final_out, int *d_oi,
               float *d_od, int *d_oc, int *d_of, float *d_pv, int *d_pi, hipStream_t st,
               const SharedBounds *sb = nullptr) {
  GULON_REQUIRE(from <= until, "expected: from <= until");                               // Index.scala:418
  GULON_REQUIRE(from >= 0 && until <= ix->n, "expected: from >= 0 && until <= length");  // Index.scala:419
  GULON_REQUIRE(K >= 0 && B >= 0, "k and batch size must be non-negative");
  GULON_UNSUPPORTED(K > GULON_MAX_K_PEELED, "k_nn = %d > %d is not supported", K, GULON_MAX_K_PEELED);
  GULON_UNSUPPORTED(K + 1 > GULON_MAX_K_PEELED && !final_out, "k_nn = %d: a shard returns k_nn + 1 <= %d entries", K,
                    GULON_MAX_K_PEELED);
  if (B == 0) return;
  if (!sb || sb->phase != 2) ix->last_filter_tiles = 0;
  if (sb && sb->phase == 1) {
    // first half of a query with shared bounds: ranges the filter does not take have no bounds to offer
    const int rbt = ceil_div(until, 64) - from / 64;
    if (!(K >= 1 && rbt > 0 && filter_eligible(ix, K, rbt))) {
      const long long nb = (long long)B * (K + 1);
      hipLaunchKernelGGL(fill_f32, dim3((unsigned)ceil_div(nb, 256LL)), dim3(256), 0, st, sb->bounds_out, nb, INFINITY);
      HIP_CHECK(hipGetLastError());
      return;
    }
  }
  const bool peeled = K > GULON_MAX_K;            // results come 64 at a time
  const int keff = peeled ? 64 : K + 1;
  const int W = ix->w;
  const int QT = W * ix->nsub;
  const int ntiles = ceil_div(B, QT);
  const int rb_begin = from / 64;
  const int rb_end = ceil_div(until, 64);
  const int rb_total = rb_end - rb_begin;
  if (K == 0 || rb_total <= 0) {
    // empty heaps: nothing to scan
    if (final_out) {
      if (K > 0) {
        HIP_CHECK(hipMemsetAsync(d_oi, 0xFF, sizeof(int) * (size_t)B * K, st));
        std::vector<float> inf((size_t)B * K, INFINITY);
        HIP_CHECK(hipMemcpyAsync(d_od, inf.data(), sizeof(float) * inf.size(), hipMemcpyHostToDevice, st));
        HIP_CHECK(hipStreamSynchronize(st));
      }
      if (d_oc) HIP_CHECK(hipMemsetAsync(d_oc, 0, sizeof(int) * (size_t)B, st));
      if (d_of) HIP_CHECK(hipMemsetAsync(d_of, 0, sizeof(int) * (size_t)B, st));
    } else {
      const int kout = K + 1;   // (peeled queries: keff is the 64-entry round, the partial list K+1 long)
      std::vector<float> inf((size_t)B * kout, INFINITY);
      std::vector<int> mx((size_t)B * kout, INT_MAX);
      HIP_CHECK(hipMemcpyAsync(d_pv, inf.data(), sizeof(float) * inf.size(), hipMemcpyHostToDevice, st));
      HIP_CHECK(hipMemcpyAsync(d_pi, mx.data(), sizeof(int) * mx.size(), hipMemcpyHostToDevice, st));
      HIP_CHECK(hipStreamSynchronize(st));
    }
    return;
  }
  // chunking: ~4096 workgroups, at least 2 row blocks per wave
  const int NW = tuning_of(ix).threads / 64;
  int want = ceil_div(tuning_of(ix).target_blocks, ntiles);
  int max_chunks = rb_total / (2 * NW);
  if (max_chunks < 1) max_chunks = 1;
  int nchunks = want < max_chunks ? want : max_chunks;
  if (nchunks < 1) nchunks = 1;
  int rb_per_chunk = ceil_div(rb_total, nchunks);
  nchunks = ceil_div(rb_total, rb_per_chunk);

  if (ix->wide) {
    run_wide_query(ix, dQ, B, K, from, until, final_out, d_oi, d_od, d_oc, d_of, d_pv, d_pi, st);
    return;
  }
  // the empty-range partial list of a large-K query is [B][K+1], not [B][64]
  const int Bp = ntiles * QT;
  const RbMap all{1, 0, 1};
  if (!peeled && filter_eligible(ix, K, rb_total)) {
    run_filter_query(ix, dQ, B, K, from, until, final_out, d_oi, d_od, d_oc, d_of, d_pv, d_pi, st, sb);
    return;
  }
  if (peeled) {
    launch_build_tables(W, ix, dQ, B, Bp, (ix->tables.ensure((size_t)Bp * ix->m_pad * 256), ix->tables.p), st);
    const int rounds = ceil_div(K + 1, 64), cap = rounds * 64;
    ix->part_v.ensure((size_t)Bp * nchunks * 64);
    ix->part_i.ensure((size_t)Bp * nchunks * 64);
    ix->gtau.ensure((size_t)Bp);
    ix->peel_v.ensure((size_t)B * cap); ix->peel_i.ensure((size_t)B * cap);
    ix->peel_tv.ensure((size_t)B * 64); ix->peel_ti.ensure((size_t)B * 64);
    ix->peel_lbv.ensure((size_t)Bp); ix->peel_lbi.ensure((size_t)Bp);
    HIP_CHECK(hipMemsetD32Async((hipDeviceptr_t)ix->peel_lbv.p, 0xBF800000 /* -1.0f */, (size_t)Bp, st));
    HIP_CHECK(hipMemsetD32Async((hipDeviceptr_t)ix->peel_lbi.p, 0xFFFFFFFF /* -1 */, (size_t)Bp, st));
    for (int r = 0; r < rounds; r++) {
      HIP_CHECK(hipMemsetD32Async((hipDeviceptr_t)ix->gtau.p, 0x7F800000 /* +inf */, (size_t)Bp, st));
      launch_scan(ix, ntiles, nchunks, rb_begin, rb_total, rb_per_chunk, all, from, until, 64, st, ix->peel_lbv.p,
                  ix->peel_lbi.p);
      launch_merge(false, ix->part_v.p, ix->part_i.p, nchunks, 64LL, (long long)nchunks * 64, B, 63, nullptr, nullptr,
                   nullptr, nullptr, ix->peel_tv.p, ix->peel_ti.p, st);
      hipLaunchKernelGGL(peel_update, dim3(B), dim3(64), 0, st, ix->peel_tv.p, ix->peel_ti.p, r, cap, ix->peel_v.p,
                         ix->peel_i.p, ix->peel_lbv.p, ix->peel_lbi.p);
    }
    if (final_out)
      hipLaunchKernelGGL(peel_finalize, dim3(ceil_div(B, 64)), dim3(64), 0, st, ix->peel_v.p, ix->peel_i.p, B, cap, K,
                         d_oi, d_od, d_oc, d_of);
    else   // a shard of a large-K query: its K+1 smallest as a partial list (merged by gulon_topk_merge_dev)
      hipLaunchKernelGGL(peel_export, dim3((unsigned)ceil_div((long long)B * (K + 1), 256LL)), dim3(256), 0, st,
                         ix->peel_v.p, ix->peel_i.p, B, cap, K + 1, d_pv, d_pi);
    HIP_CHECK(hipGetLastError());
    return;
  }
  ix->tables.ensure((size_t)Bp * ix->m_pad * 256);
  ix->part_v.ensure((size_t)Bp * nchunks * keff);
  ix->part_i.ensure((size_t)Bp * nchunks * keff);
  ix->gtau.ensure((size_t)Bp);
  HIP_CHECK(hipMemsetD32Async((hipDeviceptr_t)ix->gtau.p, 0x7F800000 /* +inf */, (size_t)Bp, st));

  launch_build_tables(W, ix, dQ, B, Bp, ix->tables.p, st);
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (ix->profile) {
    e0 = ix->take_event();
    e1 = ix->take_event();
    HIP_CHECK(hipEventRecord(e0, st));
  }
  launch_scan(ix, ntiles, nchunks, rb_begin, rb_total, rb_per_chunk, all, from, until, keff, st);
  if (ix->profile) {
    HIP_CHECK(hipEventRecord(e1, st));
    ix->events.emplace_back(e0, e1);
    ix->prof_rows += until - from;
  }
  int *flags = d_of;
  if (final_out && replay_enabled() && flags == nullptr) {   // the replay needs the tie flags even if the caller does not
    ix->flags_scratch.ensure((size_t)B);
    flags = ix->flags_scratch.p;
  }
  launch_merge(final_out, ix->part_v.p, ix->part_i.p, nchunks, (long long)keff, (long long)nchunks * keff, B, K, d_oi,
               d_od, d_oc, flags, d_pv, d_pi, st);
  // queries with exact distance ties: replay the reference heap's insertion history
  if (final_out && replay_enabled()) run_tie_replay(ix, dQ, B, K, from, until, d_oi, d_od, d_oc, flags, st);
  // queries whose distances can be NaN / +inf: the literal heap over all rows (TopKHeap.scala:69-79)
  if (final_out && replay_enabled()) run_nonfinite_literal(ix, dQ, B, K, from, until, d_oi, d_od, d_oc, d_of, st);
}

}  // namespace

GULON_API int32_t gulon_index_create(const uint8_t *codes, int32_t n, int32_t d, int32_t m, int32_t k,
                                     const float *cents, int32_t row_base, gulon_index **out) {
  return guarded([&] {
    GULON_REQUIRE(out != nullptr, "out is null");
    *out = nullptr;
    GULON_REQUIRE(n >= 0 && d >= 1 && m >= 1 && m <= d && k >= 1, "bad index shape n=%d d=%d m=%d k=%d", n, d, m, k);
    GULON_REQUIRE(cents != nullptr && (codes != nullptr || n == 0), "null input");
    int width = -1;
    GULON_REQUIRE(gulon_coder_width(k, &width) == GULON_OK && width >= 0, "too many clusters: %d", k);  // PQ.scala:12-15
    std::unique_ptr<gulon_index> ix(new gulon_index());
    ix->tune = std::make_shared<ScanTuning>();   // the environment as it is now
    ix->n = n; ix->d = d; ix->m = m; ix->k = k; ix->row_base = row_base;
    if (k > 256) {
      // Coder.BytePlus widths 10/12/16 (Coder.scala:99-127): 16-bit codes, tables in HBM (wide.hip)
      ix->wide = true;
      std::vector<int> from, until, sdim(m);
      subvectors(d, m, from, until);
      for (int j = 0; j < m; j++) sdim[j] = until[j] - from[j];
      ix->from.upload(from.data(), m);
      ix->sdim.upload(sdim.data(), m);
      ix->cents.upload(cents, (size_t)k * d);
      int bytes_per_code = 0;
      gulon_coder_bytes(width, n, &bytes_per_code);
      DevBuf<uint16_t> wide16(std::max<size_t>((size_t)m * n, 1));
      if (n > 0) {
        DevBuf<uint8_t> packed;
        packed.upload(codes, (size_t)m * bytes_per_code);
        const long long tot = (long long)m * n;
        hipLaunchKernelGGL(unpack_codes, dim3(ceil_div(tot, 256)), dim3(256), 0, 0, packed.p, width, n, bytes_per_code, m,
                           wide16.p);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipDeviceSynchronize());
      }
      wide_store_codes(ix.get(), wide16.p);
      *out = ix.release();
      return;
    }
    ix->vec = (m % 16 == 0) ? 16 : 4;
    ix->ng = ceil_div(m, ix->vec);
    ix->m_pad = ix->ng * ix->vec;
    // queries interleaved per table entry: as many as LDS holds (m = 64 -> 2, m > 72 -> 1)
    ix->w = 4;
    while (ix->w > 1 && (size_t)ix->m_pad * 256 * ix->w * 4 > LDS_BUDGET) ix->w /= 2;
    size_t per_sub = (size_t)ix->m_pad * 256 * ix->w * 4;
    GULON_UNSUPPORTED(per_sub > LDS_BUDGET, "m = %d quantizers need %zu B of LDS for one query's table (> %zu)", m,
                      per_sub, LDS_BUDGET);
    ix->nsub = ix->w < 4 ? 1 : (4 * per_sub <= LDS_BUDGET) ? 4 : (2 * per_sub <= LDS_BUDGET) ? 2 : 1;

    std::vector<int> from, until, sdim(m);
    subvectors(d, m, from, until);
    for (int j = 0; j < m; j++) sdim[j] = until[j] - from[j];
    ix->from.upload(from.data(), m);
    ix->sdim.upload(sdim.data(), m);
    ix->cents.upload(cents, (size_t)k * d);
    ix->cents_absmax = centroid_absmax(ix->cents.p, (long long)k * d);

    int bytes_per_code = 0;
    gulon_coder_bytes(width, n, &bytes_per_code);
    size_t nblk = (size_t)ceil_div(n, 64);
    ix->codes.alloc(std::max<size_t>(nblk * ix->ng * 64 * ix->vec, 16));
    if (n > 0) {
      DevBuf<uint8_t> raw;  // [m][n] one byte per (quantizer,row)
      if (width == 8) {
        raw.upload(codes, (size_t)m * n);
      } else {
        raw.alloc((size_t)m * n);
        if (width == 0) {
          HIP_CHECK(hipMemset(raw.p, 0, (size_t)m * n));
        } else {
          DevBuf<uint8_t> packed;
          packed.upload(codes, (size_t)m * bytes_per_code);
          DevBuf<uint16_t> wide((size_t)m * n);
          long long tot = (long long)m * n;
          hipLaunchKernelGGL(unpack_codes, dim3(ceil_div(tot, 256)), dim3(256), 0, 0, packed.p, width, n,
                             bytes_per_code, m, wide.p);
          hipLaunchKernelGGL(narrow_codes, dim3(ceil_div(tot, 256)), dim3(256), 0, 0, wide.p, tot, raw.p);
          HIP_CHECK(hipGetLastError());
          HIP_CHECK(hipDeviceSynchronize());
        }
      }
      long long total = (long long)nblk * ix->ng * 64;
      if (ix->vec == 16)
        hipLaunchKernelGGL(relayout_codes<16>, dim3(ceil_div(total, 256)), dim3(256), 0, 0, raw.p, n, m, ix->ng,
                           ix->codes.p, total);
      else
        hipLaunchKernelGGL(relayout_codes<4>, dim3(ceil_div(total, 256)), dim3(256), 0, 0, raw.p, n, m, ix->ng,
                           ix->codes.p, total);
      HIP_CHECK(hipGetLastError());
      HIP_CHECK(hipDeviceSynchronize());
      // the filter's conflict-ordered copy (one 16-byte code word per row; ranges the filter is never used for
      // do not need one).  GULON_FILTER_ORDER = rounds of the ordering (0: no copy)
      const int rounds = ix->tune->filter_order;
      if (rounds > 0 && ix->vec == 16 && ix->ng == 1 && (long long)nblk >= ix->tune->filter_min_rb) {
        ix->fcodes.alloc(nblk * 1024);
        ix->fperm.alloc(nblk * 64);
        launch_conflict_order(ix->codes.p, ix->fcodes.p, ix->fperm.p, (long long)nblk, FILTER_LDS_QUANTIZERS, rounds, 0);
        ix->fwindow = conflict_order_windowed() ? 4 : 1;
        HIP_CHECK(hipDeviceSynchronize());
      }
    }
    HIP_CHECK(hipDeviceSynchronize());
    *out = ix.release();
  });
}

namespace gulon {
gulon_index *make_context(gulon_index *parent) {
  std::unique_ptr<gulon_index> c(new gulon_index());
  c->n = parent->n; c->d = parent->d; c->m = parent->m; c->k = parent->k; c->row_base = parent->row_base;
  c->vec = parent->vec; c->ng = parent->ng; c->m_pad = parent->m_pad; c->nsub = parent->nsub; c->w = parent->w;
  c->wide = parent->wide;
  c->cents_absmax = parent->cents_absmax;
  c->tune = parent->tune;
  c->codes.borrow(parent->codes);
  c->fcodes.borrow(parent->fcodes);
  c->fperm.borrow(parent->fperm);
  c->fwindow = parent->fwindow;
  c->wcodes.borrow(parent->wcodes);
  c->cents.borrow(parent->cents);
  c->from.borrow(parent->from);
  c->sdim.borrow(parent->sdim);
  return c.release();
}
}  // namespace gulon

namespace {
void release_index(gulon_index *ix) {
  if (!ix) return;
  if (ix->refs.fetch_sub(1) != 1) return;       // contexts of this index are still alive
  gulon_index *parent = ix->parent;
  delete ix;
  release_index(parent);
}

// host-pointer calls: take a free workspace of this handle (the handle itself first), creating up to
// GULON_HOST_CONTEXTS (default 4) internal contexts; blocks while all are busy
int host_context_limit() {
  static const int lim = [] { const char *e = getenv("GULON_HOST_CONTEXTS"); int v = e ? atoi(e) : 4; return v < 1 ? 1 : v > 64 ? 64 : v; }();
  return lim;
}
gulon_index *acquire_host_context(gulon_index *idx) {
  std::unique_lock<std::mutex> lk(idx->host_mu);
  for (;;) {
    if (!idx->host_self_busy) { idx->host_self_busy = true; return idx; }
    if (!idx->host_free.empty()) { gulon_index *c = idx->host_free.back(); idx->host_free.pop_back(); return c; }
    if ((int)idx->host_all.size() + 1 < host_context_limit()) {
      gulon_index *c = make_context(idx);
      idx->host_all.push_back(c);
      return c;
    }
    idx->host_cv.wait(lk);
  }
}
void release_host_context(gulon_index *idx, gulon_index *c) {
  {
    std::lock_guard<std::mutex> lk(idx->host_mu);
    if (c == idx) idx->host_self_busy = false; else idx->host_free.push_back(c);
  }
  idx->host_cv.notify_one();
}
}  // namespace

GULON_API int32_t gulon_index_destroy(gulon_index *idx) {
  return guarded([&] { release_index(idx); });
}

GULON_API int32_t gulon_index_context_create(gulon_index *parent, gulon_index **out) {
  return guarded([&] {
    GULON_REQUIRE(parent != nullptr && out != nullptr, "null argument");
    *out = nullptr;
    gulon_index *root = parent->parent ? parent->parent : parent;   // contexts of contexts hang off the owner
    gulon_index *c = make_context(root);
    c->parent = root;
    root->refs.fetch_add(1);
    *out = c;
  });
}

GULON_API int32_t gulon_index_batch_query_dev(gulon_index *idx, const float *d_queries, int32_t b, int32_t k_nn,
                                              int32_t from, int32_t until, int32_t *d_out_idx, float *d_out_dist,
                                              int32_t *d_out_count, int32_t *d_out_flags, void *stream) {
  return guarded([&] {
    GULON_REQUIRE(idx != nullptr, "index is null");
    std::lock_guard<std::mutex> lock(idx->mu);
    idx->pend_b = -1;
    StreamOrder so(idx, (hipStream_t)stream);
    run_query(idx, d_queries, b, k_nn, from, until, true, d_out_idx, d_out_dist, d_out_count, d_out_flags, nullptr,
              nullptr, (hipStream_t)stream);
    so.done();
  });
}

GULON_API int32_t gulon_index_scan_partial_dev(gulon_index *idx, const float *d_queries, int32_t b, int32_t k_nn,
                                               int32_t from, int32_t until, float *d_part_dist, int32_t *d_part_idx,
                                               void *stream) {
  return guarded([&] {
    GULON_REQUIRE(idx != nullptr, "index is null");
    std::lock_guard<std::mutex> lock(idx->mu);
    idx->pend_b = -1;
    StreamOrder so(idx, (hipStream_t)stream);
    run_query(idx, d_queries, b, k_nn, from, until, false, nullptr, nullptr, nullptr, nullptr, d_part_dist,
              d_part_idx, (hipStream_t)stream);
    so.done();
  });
}

GULON_API int32_t gulon_index_scan_bounds_dev(gulon_index *idx, const float *d_queries, int32_t b, int32_t k_nn,
                                              int32_t from, int32_t until, float *d_bounds, void *stream) {
  return guarded([&] {
    GULON_REQUIRE(idx != nullptr && d_bounds != nullptr, "null argument");
    GULON_UNSUPPORTED(k_nn > GULON_MAX_K, "k_nn = %d > GULON_MAX_K = %d is only supported for unsharded queries", k_nn,
                      GULON_MAX_K);
    std::lock_guard<std::mutex> lock(idx->mu);
    const SharedBounds sb{1, d_bounds, nullptr, 0};
    idx->pend_b = -1;
    StreamOrder so(idx, (hipStream_t)stream);
    run_query(idx, d_queries, b, k_nn, from, until, false, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
              (hipStream_t)stream, &sb);
    so.done();
    idx->pend_b = b; idx->pend_k = k_nn; idx->pend_from = from; idx->pend_until = until;
  });
}

GULON_API int32_t gulon_index_scan_partial_bounded_dev(gulon_index *idx, const float *d_queries, int32_t b,
                                                       int32_t k_nn, int32_t from, int32_t until,
                                                       const float *d_all_bounds, int32_t lists, float *d_part_dist,
                                                       int32_t *d_part_idx, void *stream) {
  return guarded([&] {
    GULON_REQUIRE(idx != nullptr && d_all_bounds != nullptr && lists >= 1, "bad arguments");
    std::lock_guard<std::mutex> lock(idx->mu);
    GULON_REQUIRE(idx->pend_b == b && idx->pend_k == k_nn && idx->pend_from == from && idx->pend_until == until,
                  "scan_partial_bounded must follow scan_bounds on the same index with the same arguments");
    idx->pend_b = -1;
    const SharedBounds sb{2, nullptr, d_all_bounds, lists};
    StreamOrder so(idx, (hipStream_t)stream);
    run_query(idx, d_queries, b, k_nn, from, until, false, nullptr, nullptr, nullptr, nullptr, d_part_dist,
              d_part_idx, (hipStream_t)stream, &sb);
    so.done();
  });
}

GULON_API int32_t gulon_index_batch_query(gulon_index *idx, const float *queries, int32_t b, int32_t k_nn,
                                          int32_t from, int32_t until, int32_t *out_idx, float *out_dist,
                                          int32_t *out_count, int32_t *out_flags) {
  return guarded([&] {
    GULON_REQUIRE(idx != nullptr, "index is null");
    GULON_REQUIRE(b >= 0 && k_nn >= 0, "k and batch size must be non-negative");
    // concurrent callers (Tests.scala:109-122 queries from a thread pool) each get a workspace of their own
    gulon_index *c = acquire_host_context(idx);
    struct Release { gulon_index *i, *c; ~Release() { release_host_context(i, c); } } rel{idx, c};
    std::lock_guard<std::mutex> lock(c->mu);
    c->pend_b = -1;
    size_t bk = (size_t)b * (size_t)k_nn;
    c->stage_q.ensure((size_t)b * c->d + 1);
    c->stage_oi.ensure(bk + 1);
    c->stage_od.ensure(bk + 1);
    c->stage_oc.ensure((size_t)b + 1);
    c->stage_of.ensure((size_t)b + 1);
    if (!c->host_stream) HIP_CHECK(hipStreamCreateWithFlags(&c->host_stream, hipStreamNonBlocking));
    hipStream_t st = c->host_stream;
    StreamOrder so(c, st);
    if (b > 0) HIP_CHECK(hipMemcpyAsync(c->stage_q.p, queries, sizeof(float) * (size_t)b * c->d, hipMemcpyHostToDevice, st));
    run_query(c, c->stage_q.p, b, k_nn, from, until, true, c->stage_oi.p, c->stage_od.p, c->stage_oc.p,
              c->stage_of.p, nullptr, nullptr, st);
    if (bk) {
      c->stage_oi.download(out_idx, bk, st);
      c->stage_od.download(out_dist, bk, st);
    }
    if (b > 0 && out_count) c->stage_oc.download(out_count, b, st);
    if (b > 0 && out_flags) c->stage_of.download(out_flags, b, st);
    so.done();
    HIP_CHECK(hipStreamSynchronize(st));
  });
}

GULON_API int32_t gulon_index_profile(gulon_index *idx, int32_t enable) {
  return guarded([&] {
    GULON_REQUIRE(idx != nullptr, "index is null");
    std::lock_guard<std::mutex> lock(idx->mu);
    idx->events.clear();
    idx->ev_next = 0;
    idx->prof_rows = 0;
    idx->profile = enable != 0;
    if (idx->profile)
      while (idx->ev_pool.size() < 512) {
        hipEvent_t e = nullptr;
        HIP_CHECK(hipEventCreate(&e));
        idx->ev_pool.push_back(e);
      }
  });
}

GULON_API int32_t gulon_index_profile_read(gulon_index *idx, double *scan_ms_total, int32_t *launches) {
  return guarded([&] {
    GULON_REQUIRE(idx != nullptr, "index is null");
    std::lock_guard<std::mutex> lock(idx->mu);
    double tot = 0;
    for (auto &e : idx->events) {
      HIP_CHECK(hipEventSynchronize(e.second));
      float ms = 0;
      HIP_CHECK(hipEventElapsedTime(&ms, e.first, e.second));
      tot += ms;
    }
    if (scan_ms_total) *scan_ms_total = tot;
    if (launches) *launches = (int32_t)idx->events.size();
  });
}

GULON_API int32_t gulon_index_profile_read_ex(gulon_index *idx, double *ms_total, int32_t *launches,
                                              int64_t *rows_total) {
  int32_t rc = gulon_index_profile_read(idx, ms_total, launches);
  if (rc == GULON_OK && rows_total) *rows_total = idx->prof_rows;
  return rc;
}

GULON_API int32_t gulon_index_filter_stats(gulon_index *idx, int32_t *query_tiles, int32_t *tiles_redone) {
  return guarded([&] {
    GULON_REQUIRE(idx != nullptr, "index is null");
    std::lock_guard<std::mutex> lock(idx->mu);
    HIP_CHECK(hipDeviceSynchronize());
    const int nt = idx->last_filter_tiles;
    int redone = 0;
    if (nt > 0) {
      std::vector<int> h((size_t)nt);
      HIP_CHECK(hipMemcpy(h.data(), idx->fb_tile.p, sizeof(int) * h.size(), hipMemcpyDeviceToHost));
      for (int v : h) redone += v != 0;
    }
    if (query_tiles) *query_tiles = nt;
    if (tiles_redone) *tiles_redone = redone;
  });
}

GULON_API int32_t gulon_index_tuning(gulon_index *idx, const char *key, int32_t value) {
  return guarded([&] {
    GULON_REQUIRE(idx != nullptr && key != nullptr, "null argument");
    std::lock_guard<std::mutex> lock(idx->mu);
    // copy-on-write: contexts created earlier keep the settings they were created with
    auto t = std::make_shared<ScanTuning>(tuning_of(idx));
    GULON_REQUIRE(t->set(key, value), "unknown tuning key");
    idx->tune = t;
  });
}

GULON_API int32_t gulon_topk_merge_dev(const float *d_part_dist, const int32_t *d_part_idx, int32_t lists,
                                       int64_t list_stride, int32_t b, int32_t k_nn, int32_t *d_out_idx,
                                       float *d_out_dist, int32_t *d_out_count, int32_t *d_out_flags, void *stream) {
  return guarded([&] {
    GULON_REQUIRE(lists >= 1 && b >= 0 && k_nn >= 1, "bad merge shape");
    GULON_UNSUPPORTED(k_nn + 1 > GULON_MAX_K_PEELED, "k_nn = %d > %d", k_nn, GULON_MAX_K_PEELED - 1);
    int keff = k_nn + 1;
    GULON_REQUIRE(list_stride == 0 || list_stride >= (long long)b * keff, "list_stride too small");
    if (b == 0) return;
    if (k_nn > GULON_MAX_K) {   // beyond a wavefront list: pairwise merges through scratch (synchronises the stream)
      launch_merge_long(d_part_dist, d_part_idx, lists, list_stride ? (long long)list_stride : (long long)b * keff, b, k_nn,
                        d_out_idx, d_out_dist, d_out_count, d_out_flags, (hipStream_t)stream);
      return;
    }
    launch_merge(true, d_part_dist, d_part_idx, lists, list_stride ? (long long)list_stride : (long long)b * keff,
                 (long long)keff, b, k_nn, d_out_idx,
                 d_out_dist, d_out_count, d_out_flags, nullptr, nullptr, (hipStream_t)stream);
  });
}

GULON_API int32_t gulon_topk_merge(const float *part_dist, const int32_t *part_idx, int32_t lists, int32_t b,
                                   int32_t k_nn, int32_t *out_idx, float *out_dist, int32_t *out_count,
                                   int32_t *out_flags) {
  return guarded([&] {
    GULON_REQUIRE(lists >= 1 && b >= 0 && k_nn >= 1, "bad merge shape");
    GULON_UNSUPPORTED(k_nn > GULON_MAX_K, "k_nn = %d > GULON_MAX_K = %d", k_nn, GULON_MAX_K);
    if (b == 0) return;
    size_t np = (size_t)lists * b * (k_nn + 1), bk = (size_t)b * k_nn;
    DevBuf<float> pv; DevBuf<int> pi; DevBuf<int> oi(bk), oc(b), of(b); DevBuf<float> od(bk);
    pv.upload(part_dist, np); pi.upload(part_idx, np);
    int keff = k_nn + 1;
    launch_merge(true, pv.p, pi.p, lists, (long long)b * keff, (long long)keff, b, k_nn, oi.p, od.p, oc.p, of.p,
                 nullptr, nullptr, nullptr);
    oi.download(out_idx, bk); od.download(out_dist, bk);
    if (out_count) oc.download(out_count, b);
    if (out_flags) of.download(out_flags, b);
    HIP_CHECK(hipDeviceSynchronize());
  });
}

GULON_API int32_t gulon_prepare_query(const float *cents, int32_t d, int32_t m, int32_t k, const float *queries,
                                      int32_t b, float *t_out) {
  return guarded([&] {
    GULON_REQUIRE(d >= 1 && m >= 1 && m <= d && k >= 1 && b >= 0, "bad shape");
    if (b == 0) return;
    std::vector<int> from, until, sdim(m);
    subvectors(d, m, from, until);
    for (int j = 0; j < m; j++) sdim[j] = until[j] - from[j];
    DevBuf<int> dfrom, dsd; DevBuf<float> dc, dq, dt((size_t)b * m * k);
    dfrom.upload(from.data(), m); dsd.upload(sdim.data(), m);
    dc.upload(cents, (size_t)k * d); dq.upload(queries, (size_t)b * d);
    if (k > 256) {
      launch_build_tables_wide(dc.p, dfrom.p, dsd.p, d, m, k, dq.p, 0, b, dt.p, nullptr);
      dt.download(t_out, (size_t)b * m * k);
      HIP_CHECK(hipDeviceSynchronize());
      return;
    }
    long long total = (long long)((b + 3) / 4) * m * 256;
    hipLaunchKernelGGL((build_tables<false, 4>), dim3(ceil_div(total, 256)), dim3(256), 0, 0, dc.p, dfrom.p, dsd.p, d, m, k,
                       m, dq.p, b, dt.p, (const int *)nullptr, (float *)nullptr);
    HIP_CHECK(hipGetLastError());
    dt.download(t_out, (size_t)b * m * k);
    HIP_CHECK(hipDeviceSynchronize());
  });
}
