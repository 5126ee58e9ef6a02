// KMeans / ProductQuantizer on gfx950 (KMeans.scala, ProductQuantizer.scala).
//
//   assign   exact argmin of  offsets[c] - 2*(x . c)  in the JVM's unfused, sequential
//            binary32 order, including the  d == min && rng.nextBoolean()  tie-break:
//            pass 1 computes the draw-free argmin and the number of RNG draws each row
//            makes; a segmented prefix sum places every row in its java.util.Random
//            stream (one stream per rng_batch rows); pass 2 replays only the rows that
//            draw, after an O(log) LCG jump-ahead.
//   update   KMeans.fromAssignment: order-dependent running mean  c += (x - c)/n.
//            Rows are bucketed by cluster with a STABLE counting sort, then every
//            (cluster, dim) chain runs sequentially in row order.
//   train    KMeans.computeClusters loop for a batch of independent problems
//            (ProductQuantizer.apply = m problems, seed = quantizer index).
#include "kmeans.hpp"

#include <chrono>
#include <map>

namespace gulon {

// ---------------------------------------------------------------------------
// centroid prep: KMeans.apply offsets (KMeans.scala:170-186) + zero-padded copy.
// Zero padding is exact: d + 0*0 == d for every binary32 d (a -0 partial sum may
// become +0, which no comparison or later operation here can observe).
// ---------------------------------------------------------------------------
__global__ void prep_centroids(const float *__restrict__ C, int k, int s, int smax, float *__restrict__ Cpad,
                               float *__restrict__ off) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= k) return;
  float acc = 0.f;
  for (int j = 0; j < s; j++) {
    float x = C[(size_t)c * s + j];
    acc += x * x;
    Cpad[(size_t)c * smax + j] = x;
  }
  for (int j = s; j < smax; j++) Cpad[(size_t)c * smax + j] = 0.f;
  off[c] = acc;
}

// ---------------------------------------------------------------------------
// pass 1: draw-free argmin + number of draws (KMeans.scala:33-54 / :76-97 with
// rng.nextBoolean() read as false).  assign[] is written only when some centroid
// won, exactly like the reference (NaN distances never win).
// ---------------------------------------------------------------------------
// STAGE: the padded centroids and their offsets are copied to LDS first (k * (SMAX + 1) floats <= 64 KiB), pairs of
// centroids interleaved per dimension.  Every
// lane reads the same centroid, so from global memory these are scalar loads that the loop waits for centroid by
// centroid (the re-check of 120 K flagged rows: 125 us, a few hundred cycles of load latency per centroid); from LDS
// they are broadcast reads the compiler can request several centroids ahead.
template <int SMAX, bool STAGE>
__global__ __launch_bounds__(256) void assign_exact(const float *__restrict__ X, int n, int ld, int from, int s,
                                                    const float *__restrict__ Cpad, const float *__restrict__ off,
                                                    int k, const int *__restrict__ rows, int nrows,
                                                    int *__restrict__ assign, unsigned *__restrict__ ties,
                                                    unsigned long long *__restrict__ tie_total) {
  extern __shared__ float exact_lds[];   // STAGE: [ceil(k/2)][SMAX][2] centroid PAIRS (c, c+1 interleaved per dim), then [k] offsets
  const int kp = (k + 1) / 2;
  if (STAGE) {
    for (int e = threadIdx.x; e < kp * SMAX * 2; e += 256) {
      const int cp = e / (2 * SMAX), r = e - cp * 2 * SMAX, j = r >> 1, c = 2 * cp + (r & 1);
      exact_lds[e] = c < k ? Cpad[(size_t)c * SMAX + j] : 0.f;
    }
    for (int e = threadIdx.x; e < 2 * kp; e += 256) exact_lds[kp * SMAX * 2 + e] = e < k ? off[e] : 0.f;
    __syncthreads();
  }
  int t = blockIdx.x * 256 + threadIdx.x;
  int i = -1;
  if (t < nrows) i = rows ? rows[t] : t;
  float x[SMAX];
#pragma unroll
  for (int j = 0; j < SMAX; j++) x[j] = 0.f;
  if (i >= 0) {
    const float *row = X + (size_t)i * ld + from;
#pragma unroll
    for (int j = 0; j < SMAX; j++)
      if (j < s) x[j] = row[j];
  }
  float mn = FLT_MAX;
  int best = -1;
  unsigned nt = 0;
  if (STAGE) {
    // two centroids per step on the packed fp32 pipe: each component is the reference's own sequential, unfused chain
    // (v_pk_mul_f32 / v_pk_add_f32 round every component like their scalar forms); the comparisons stay in centroid
    // order.  The kernel is bound by its arithmetic (k * s products per row), so this halves it.
    const f32x2 *pairs = reinterpret_cast<const f32x2 *>(exact_lds);
    const f32x2 *offs = reinterpret_cast<const f32x2 *>(exact_lds + kp * SMAX * 2);
#pragma unroll 2
    for (int cp = 0; cp < kp; cp++) {
      const f32x2 *cc = pairs + (size_t)cp * SMAX;
      f32x2 d = {0.f, 0.f};
#pragma unroll
      for (int j = 0; j < SMAX; j++) {
        const f32x2 xx = {x[j], x[j]};
        d = d + xx * cc[j];
      }
      const f32x2 two = {2.f, 2.f};
      d = offs[cp] - two * d;
      if (d.x < mn) { mn = d.x; best = 2 * cp; }
      else if (d.x == mn) nt++;
      if (2 * cp + 1 < k) {
        if (d.y < mn) { mn = d.y; best = 2 * cp + 1; }
        else if (d.y == mn) nt++;
      }
    }
  } else {
    for (int c = 0; c < k; c++) {
      const float *cc = Cpad + (size_t)c * SMAX;
      float d = 0.f;
#pragma unroll
      for (int j = 0; j < SMAX; j++) d += x[j] * cc[j];
      d = off[c] - 2 * d;
      if (d < mn) { mn = d; best = c; }
      else if (d == mn) nt++;
    }
  }
  if (i >= 0) {
    if (best >= 0) assign[i] = best;
    ties[i] = nt;
  } else {
    nt = 0;
  }
  // any draws in this launch?
  unsigned long long wsum = nt;
  for (int o = 32; o > 0; o >>= 1) wsum += __shfl_down(wsum, o);
  if ((threadIdx.x & 63) == 0 && wsum) atomicAdd(tie_total, wsum);
}

// generic dimension (s > 128): x re-read from memory for every centroid
__global__ __launch_bounds__(256) void assign_exact_generic(const float *__restrict__ X, int n, int ld, int from,
                                                            int s, const float *__restrict__ Cpad, int smax,
                                                            const float *__restrict__ off, int k,
                                                            const int *__restrict__ rows, int nrows,
                                                            int *__restrict__ assign, unsigned *__restrict__ ties,
                                                            unsigned long long *__restrict__ tie_total) {
  int t = blockIdx.x * 256 + threadIdx.x;
  int i = -1;
  if (t < nrows) i = rows ? rows[t] : t;
  unsigned nt = 0;
  if (i >= 0) {
    const float *row = X + (size_t)i * ld + from;
    float mn = FLT_MAX;
    int best = -1;
    for (int c = 0; c < k; c++) {
      const float *cc = Cpad + (size_t)c * smax;
      float d = 0.f;
      for (int j = 0; j < s; j++) d += row[j] * cc[j];
      d = off[c] - 2 * d;
      if (d < mn) { mn = d; best = c; }
      else if (d == mn) nt++;
    }
    if (best >= 0) assign[i] = best;
    ties[i] = nt;
  }
  unsigned long long wsum = nt;
  for (int o = 32; o > 0; o >>= 1) wsum += __shfl_down(wsum, o);
  if ((threadIdx.x & 63) == 0 && wsum) atomicAdd(tie_total, wsum);
}

// ---------------------------------------------------------------------------
// segmented exclusive prefix sum of ties[] (segments = rng_batch rows): where each
// row's draws sit in its java.util.Random stream.
// ---------------------------------------------------------------------------
__device__ inline unsigned block_exclusive_scan_1024(unsigned v, unsigned *total, unsigned *lds /*>=16*/) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned inc = v;
  for (int o = 1; o < 64; o <<= 1) {
    unsigned u = __shfl_up(inc, o);
    if (lane >= o) inc += u;
  }
  if (lane == 63) lds[wave] = inc;
  __syncthreads();
  if (wave == 0) {
    unsigned w = lane < 16 ? lds[lane] : 0;
    unsigned winc = w;
    for (int o = 1; o < 16; o <<= 1) {
      unsigned u = __shfl_up(winc, o);
      if (lane >= o) winc += u;
    }
    if (lane < 16) lds[lane] = winc - w;   // exclusive wave offsets
    if (lane == 15) lds[16] = winc;
  }
  __syncthreads();
  unsigned res = inc - v + lds[wave];
  *total = lds[16];
  __syncthreads();
  return res;
}

// grid (blocks_per_seg, nseg), 1024 threads
__global__ __launch_bounds__(1024) void tie_block_sums(const unsigned *__restrict__ ties, int n, int seg_len,
                                                       int bps, unsigned *__restrict__ local,
                                                       unsigned *__restrict__ block_tot) {
  __shared__ unsigned lds[32];
  int seg = blockIdx.y, blk = blockIdx.x;
  long long seg_begin = (long long)seg * seg_len;
  long long seg_end = seg_begin + seg_len < n ? seg_begin + seg_len : n;
  long long idx = seg_begin + (long long)blk * 1024 + threadIdx.x;
  unsigned v = idx < seg_end ? ties[idx] : 0u;
  unsigned tot;
  unsigned ex = block_exclusive_scan_1024(v, &tot, lds);
  if (idx < seg_end) local[idx] = ex;
  if (threadIdx.x == 0) block_tot[(size_t)seg * bps + blk] = tot;
}

// one block per segment: exclusive scan of that segment's block totals
__global__ __launch_bounds__(1024) void tie_block_scan(const unsigned *__restrict__ block_tot, int bps,
                                                       unsigned long long *__restrict__ block_off) {
  __shared__ unsigned lds[32];
  __shared__ unsigned long long carry;
  int seg = blockIdx.x;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < bps; base += 1024) {
    int b = base + threadIdx.x;
    unsigned v = b < bps ? block_tot[(size_t)seg * bps + b] : 0u;
    unsigned tot;
    unsigned ex = block_exclusive_scan_1024(v, &tot, lds);
    unsigned long long c0 = carry;
    if (b < bps) block_off[(size_t)seg * bps + b] = c0 + ex;
    __syncthreads();
    if (threadIdx.x == 0) carry = c0 + tot;
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------
// pass 2: replay the rows that draw, with the RNG jumped to their stream position.
// ---------------------------------------------------------------------------
template <int SMAX>
__global__ __launch_bounds__(256) void assign_resolve(const float *__restrict__ X, int n, int ld, int from, int s,
                                                      const float *__restrict__ Cpad, int smax,
                                                      const float *__restrict__ off, int k,
                                                      const unsigned *__restrict__ ties,
                                                      const unsigned *__restrict__ local,
                                                      const unsigned long long *__restrict__ block_off, int seg_len,
                                                      int bps, const int *__restrict__ rows, int nrows,
                                                      int *__restrict__ assign) {
  int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= nrows) return;
  int i = rows ? rows[t] : t;          // only the flagged rows can draw when the MFMA filter ran
  if (ties[i] == 0) return;
  int seg = i / seg_len;
  int blk = (i - seg * seg_len) / 1024;
  unsigned long long pos = block_off[(size_t)seg * bps + blk] + local[i];
  JRandom rng(0);
  rng.skip(pos);
  const float *row = X + (size_t)i * ld + from;
  float mn = FLT_MAX;
  int best = -1;
  for (int c = 0; c < k; c++) {
    const float *cc = Cpad + (size_t)c * smax;
    float d = 0.f;
    for (int j = 0; j < s; j++) d += row[j] * cc[j];
    d = off[c] - 2 * d;
    if (d < mn || (d == mn && rng.next_boolean())) { best = c; mn = d; }
  }
  if (best >= 0) assign[i] = best;
}

// The same replay with one WAVE per drawing row (for many centroids / long sub-vectors, where one
// thread walking k * s products is a 100-ms latency chain): rows with draws are compacted first;
// the lanes compute 64 centroids' distances at a time (each lane its centroid's sequential,
// unfused chain -- the reference's arithmetic), then the scan over those 64 runs in centroid order
// on the candidates that can change the minimum or draw (d <= running minimum).
__global__ void collect_tie_rows(const unsigned *__restrict__ ties, const int *__restrict__ rows, int nrows,
                                 int *__restrict__ list, unsigned *__restrict__ count) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nrows) return;
  int i = rows ? rows[t] : t;
  if (ties[i] != 0) list[atomicAdd(count, 1u)] = i;
}

// Stream positions of a FEW drawing rows without the dense per-row arrays: row i's draws start after those of the
// drawing rows before it in its segment.  (The dense form -- zero a 40 MB array, scatter the flagged rows' counts,
// copy, block sums, block scan -- moves 120 MB per problem and iteration for, typically, 70 drawing rows.)
__global__ void tie_positions_sparse(const unsigned *__restrict__ ties, const int *__restrict__ list,
                                     const unsigned *__restrict__ count, int seg_len,
                                     unsigned long long *__restrict__ pos) {
  const unsigned total = *count;
  const unsigned e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const int i = list[e], seg = i / seg_len;
  unsigned long long p = 0;
  for (unsigned o = 0; o < total; o++) {
    const int r = list[o];
    if (r < i && r / seg_len == seg) p += ties[r];
  }
  pos[e] = p;
}

// pos != null: the stream position of list[e] is pos[e] (tie_positions_sparse); else block_off / local (dense form)
__global__ __launch_bounds__(64) void assign_resolve_wave(const float *__restrict__ X, int ld, int from, int s,
                                                          const float *__restrict__ Cpad, int smax,
                                                          const float *__restrict__ off, int k,
                                                          const unsigned *__restrict__ local,
                                                          const unsigned long long *__restrict__ block_off,
                                                          int seg_len, int bps, const int *__restrict__ list,
                                                          const unsigned *__restrict__ count,
                                                          int *__restrict__ assign,
                                                          const unsigned long long *__restrict__ pos) {
  extern __shared__ float rowv[];   // s
  const int lane = threadIdx.x;
  const unsigned total = *count;
  for (unsigned e = blockIdx.x; e < total; e += gridDim.x) {
    const int i = list[e];
    __syncthreads();
    for (int j = lane; j < s; j += 64) rowv[j] = X[(size_t)i * ld + from + j];
    __syncthreads();
    const int seg = i / seg_len;
    const int blk = (i - seg * seg_len) / 1024;
    JRandom rng(0);
    rng.skip(pos ? pos[e] : block_off[(size_t)seg * bps + blk] + local[i]);
    float mn = FLT_MAX;
    int best = -1;
    for (int c0 = 0; c0 < k; c0 += 64) {
      const int c = c0 + lane;
      float d = 0.f;
      if (c < k) {
        const float *cc = Cpad + (size_t)c * smax;
        for (int j = 0; j < s; j++) d += rowv[j] * cc[j];
        d = off[c] - 2 * d;
      }
      // in centroid order: only d < mn or d == mn can act (NaN never does)
      unsigned long long mk = __ballot(c < k && d <= mn);
      while (mk) {
        const int l = __ffsll((long long)mk) - 1;
        mk &= mk - 1;
        const float dv = readlane_f(d, l);
        if (dv < mn || (dv == mn && rng.next_boolean())) { best = c0 + l; mn = dv; }
      }
    }
    if (lane == 0 && best >= 0) assign[i] = best;
  }
}

// ---------------------------------------------------------------------------
// misc small kernels
// ---------------------------------------------------------------------------
__global__ void gather_centroids(const float *__restrict__ X, int ld, int from, int s, const int *__restrict__ rows,
                                 int k, float *__restrict__ C) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= k * s) return;
  int c = t / s, j = t - c * s;
  C[t] = X[(size_t)rows[c] * ld + from + j];
}

__global__ void count_mismatch(const int *__restrict__ a, const int *__restrict__ b, int n,
                               unsigned *__restrict__ out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  bool diff = i < n && a[i] != b[i];
  // plain store of a constant: no atomic needed, and no contention when every wave differs
  if (__ballot(diff) && (threadIdx.x & 63) == 0) *out = 1u;
}

__global__ void scatter_ties(const int *__restrict__ rows, int nrows, const unsigned *__restrict__ ties,
                             unsigned *__restrict__ dense) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < nrows) dense[rows[t]] = ties[rows[t]];
}

__global__ void copy_slice(const float *__restrict__ X, int ld, int from, int s, long long total,
                           float *__restrict__ out) {
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  long long r = t / s;
  int j = (int)(t - r * s);
  out[t] = X[(size_t)r * ld + from + j];
}

__global__ void narrow_assign_u8(const int *__restrict__ a, long long n, uint8_t *__restrict__ out) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (uint8_t)a[i];
}

// ---------------------------------------------------------------------------
// update: stable counting sort by cluster, then sequential chains
// ---------------------------------------------------------------------------
constexpr int CHUNK_ROWS = 1024;   // rows per workgroup-chunk of the counting sort (256 per wave)

// All update kernels take an array of per-problem descriptors and pick theirs with
// blockIdx.y (z for the group scan), so the m sub-quantizers of a ProductQuantizer are
// updated by ONE launch each: the sequential chains are latency-bound (k*s threads per
// problem), and only running all problems' chains side by side fills the GPU.

// per chunk histogram: hist[chunk][k]
__global__ __launch_bounds__(256) void sort_hist(const UpdDesc *__restrict__ descs, int n, int k) {
  const UpdDescG D = load_desc(descs, blockIdx.y);
  extern __shared__ unsigned sh[];  // k
  for (int c = threadIdx.x; c < k; c += 256) sh[c] = 0;
  __syncthreads();
  const long long chunk = blockIdx.x;
  const long long r0 = chunk * CHUNK_ROWS;
  const long long r1 = r0 + CHUNK_ROWS < n ? r0 + CHUNK_ROWS : n;
  for (long long r = r0 + threadIdx.x; r < r1; r += 256) atomicAdd(&sh[D.assign[r]], 1u);
  __syncthreads();
  for (int c = threadIdx.x; c < k; c += 256) D.hist[(size_t)chunk * k + c] = sh[c];
}

// two-level exclusive scan of the per-chunk histograms (per cluster, over chunks):
// level 1: one thread per (group of SCAN_GROUP chunks, cluster) scans its chunks in place
constexpr int SCAN_GROUP = 64;
__global__ void sort_scan_groups(const UpdDesc *__restrict__ descs, long long nchunks, int k) {
  const UpdDescG D = load_desc(descs, blockIdx.z);
  int c = blockIdx.y * blockDim.x + threadIdx.x;
  long long g = blockIdx.x;
  if (c >= k) return;
  long long c0 = g * SCAN_GROUP, c1 = c0 + SCAN_GROUP < nchunks ? c0 + SCAN_GROUP : nchunks;
  unsigned run = 0;
  for (long long ch = c0; ch < c1; ch++) {
    unsigned v = D.hist[(size_t)ch * k + c];
    D.hist[(size_t)ch * k + c] = run;
    run += v;
  }
  D.gtot[(size_t)g * k + c] = run;
}
// level 2: one thread per cluster scans the group totals; then cluster starts
__global__ void sort_scan_top(const UpdDesc *__restrict__ descs, long long ngroups, int k) {
  const UpdDescG D = load_desc(descs, blockIdx.y);
  extern __shared__ unsigned cnt[];  // k
  for (int c = threadIdx.x; c < k; c += blockDim.x) {
    unsigned run = 0;
    for (long long g = 0; g < ngroups; g++) {
      unsigned v = D.gtot[(size_t)g * k + c];
      D.gtot[(size_t)g * k + c] = run;
      run += v;
    }
    cnt[c] = run;
    D.count[c] = run;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned run = 0;
    for (int c = 0; c < k; c++) { D.start[c] = run; run += cnt[c]; }
  }
  // clusters by descending size: update_chains is as slow as its longest chain, and the workgroups holding the
  // longest ones should be the first to get a SIMD (rank by counting; k is small where this matters)
  if (D.corder) {
    for (int c = threadIdx.x; c < k; c += blockDim.x) {
      const unsigned mine = cnt[c];
      int rank = 0;
      for (int o = 0; o < k; o++) rank += cnt[o] > mine || (cnt[o] == mine && o < c);
      D.corder[rank] = c;
    }
  }
}

// Stable placement of a chunk's rows: bucket position start[c] + rank <- the row's slice, rank = rows of the
// same cluster before it (in row order).  The slices themselves move here -- xb[position] = slice of the row,
// sp = s rounded up to even floats so that every bucket row starts 8-byte aligned -- because the chains that
// follow are sequential per (cluster, dim): over row ids they were a random gather of 40-byte pieces (one
// 64-byte sector each, 320 M sectors per full-PQ iteration at BASELINE config 3: the bound of the round-1
// kernel); over the regrouped copy every chain streams contiguous memory.
//
// One workgroup per 1024-row chunk, wave w = rows [256 w, 256 w + 256) in four steps of 64:
//   A  per-wave cluster counts.  The lanes holding the same cluster are found without a loop: one ballot per
//      key bit, the AND of (bit set ? ballot : ~ballot) is the mask of lanes with an equal key;
//   B  chunk-local exclusive scan over (cluster, wave) -> every wave's first position per cluster;
//   C  placement: position = wave's running position + rank among the step's equal keys (popcount of the
//      mask below the lane); same-wave LDS accesses execute in program order, so every lane of a cluster
//      reads the running position before the cluster's first lane advances it;
//   D  (STAGED) the chunk's slices were written to LDS in sorted order; they leave as contiguous runs -- one
//      per cluster, chunk rows / k slices long -- instead of one scattered 40-byte store per row (12.8 GB in
//      21 ms against ~5 ms at BASELINE config 3: partial-line stores do not combine).
// STAGED needs k <= 1024 and sp <= 16 (LDS); otherwise the slices go straight to their positions.
// Measured and dropped (round 2, commit 0fb1c3a has the kernel): the same placement as a persistent stream -- as many
// workgroups as the chip holds, each requesting its NEXT chunk before sorting the current one.  Loads and sort alone
// then take 2.5 ms, but with the stores 8.6 ms against 5.9 ms for this kernel: the stores are what binds -- unaligned
// 160-byte runs leave at 3.4 TB/s at best (scripts/micro/store_runs.hip; whole aligned 128-byte lines: 5.7 TB/s) --
// and a workgroup that lives on has to wait for its own stores before it can trust its look-ahead loads (vmcnt counts
// both, in order), while a workgroup that ends leaves its stores to drain under the next one's loads.
// Also measured and dropped: stage D in 16-byte stores (a unit on an even address opens a pair with its successor, run
// heads and tails leave alone in 8-byte stores of their own): 7.2 ms against 5.9 -- a run then leaves in three
// instructions instead of one, and what the memory side is short of is requests per line, not bytes per request
// (WRITE_SIZE = the algorithmic 12.8 GB either way).
template <int SMAX /* 0: slices go straight to their positions; else staged, s <= SMAX */>
__global__ __launch_bounds__(256) void sort_place(const UpdDesc *__restrict__ descs, int n, int k, int key_bits,
                                                  int cpx /* > 0: chunks per XCD, 1-D grid */) {
  constexpr bool STAGED = SMAX > 0;
  constexpr int NV = STAGED ? SMAX : 1;
  // Which chunk: workgroups are dealt round-robin over the 8 XCDs (observed, for speed only), and a chunk's run of a
  // cluster continues where the previous chunk's ended -- two partial lines per 160-byte run at BASELINE config 3.
  // With consecutive chunks on the SAME XCD, at about the same time, the halves of a line meet in that XCD's L2 and
  // leave as one full line; dealt over eight L2s every run is written back as partial lines.
  int prob = blockIdx.y;
  long long chunk = blockIdx.x;
  if (cpx > 0) {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    prob = slot / cpx;
    chunk = (long long)xcd * cpx + (slot - prob * cpx);
    if (chunk * CHUNK_ROWS >= n) return;
  }
  const UpdDescG D = load_desc(descs, prob);
  extern __shared__ unsigned sh[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  unsigned *wh = sh;                                  // [4][k] per-wave counts, then running positions
  unsigned *gpos = sh + 4 * k;                        // STAGED: [k] bucket position of chunk-local position 0
  int *dstrow = reinterpret_cast<int *>(sh + 5 * k);  // STAGED: [CHUNK_ROWS] bucket position of a local position
  unsigned short *lpos = reinterpret_cast<unsigned short *>(sh + 5 * k + CHUNK_ROWS);   // STAGED: [CHUNK_ROWS] local position of a chunk row
  float *sorted = reinterpret_cast<float *>(sh + ((5 * k + CHUNK_ROWS + CHUNK_ROWS / 2 + 1) & ~1));   // STAGED: [CHUNK_ROWS][sp], 8-byte aligned
  __shared__ unsigned wave_tot[4];
  const long long r0 = chunk * CHUNK_ROWS;
  const long long r1 = r0 + CHUNK_ROWS < n ? r0 + CHUNK_ROWS : n;
  const long long grp = chunk / SCAN_GROUP;
  const int s = D.s, sp = (s + 1) & ~1;
  // STAGED: the wave's 256 slices are 256 * s consecutive floats of the compact source = 64 * s float4: all of them
  // requested now (s loads of 16 bytes per lane), consumed after the positions are known
  f32x4 v[NV];
  const long long wr0 = r0 + wave * 256;                          // first row of this wave
  const int wrows = (int)max(0ll, min(256ll, r1 - wr0));          // its rows
  if (STAGED) {
    const auto src4 = reinterpret_cast<gptr<const f32x4>>(D.x + (size_t)wr0 * s);
    const int nf4 = wrows * s / 4, tail0 = nf4 * 4, total = wrows * s;   // whole float4s, then up to 3 floats
#pragma unroll
    for (int u = 0; u < NV; u++) {
      const int f = lane + 64 * u;
      v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (u < s) {
        if (f < nf4) v[u] = __builtin_nontemporal_load(&src4[f]);   // read once: keeps L2 for the partial lines of the stores (6.1 -> 5.9 ms; nt STORES: 9.2 ms)
        else if (f == nf4 && tail0 < total) {
          const auto tp = D.x + (size_t)wr0 * s + tail0;
          v[u].x = tp[0];
          if (tail0 + 1 < total) v[u].y = tp[1];
          if (tail0 + 2 < total) v[u].z = tp[2];
        }
      }
    }
  }
  for (int e = tid; e < 4 * k; e += 256) wh[e] = 0;
  __syncthreads();
  const unsigned long long lt = (1ull << lane) - 1ull;
  // A: keys of this lane's four rows, per-wave counts
  int keys[4];
  unsigned long long same[4];
#pragma unroll
  for (int t = 0; t < 4; t++) {
    const long long r = wr0 + t * 64 + lane;
    const bool valid = r < r1;
    keys[t] = valid ? D.assign[r] : 0;
  }
#pragma unroll
  for (int t = 0; t < 4; t++) {
    const bool valid = wr0 + t * 64 + lane < r1;
    unsigned long long sm = __ballot(valid);
    for (int bit = 0; bit < key_bits; bit++) {
      const unsigned long long bm = __ballot((keys[t] >> bit) & 1);
      sm &= ((keys[t] >> bit) & 1) ? bm : ~bm;
    }
    same[t] = valid ? sm : 0ull;
    if (valid && (sm & lt) == 0ull) wh[wave * k + keys[t]] += (unsigned)__popcll(sm);   // only this wave touches wh[wave]
  }
  __syncthreads();
  // B: first position of every (cluster, wave)
  if (STAGED) {
    // exclusive scan of the chunk's cluster totals: thread t owns clusters [t*per, t*per + per)
    const int per = (k + 255) / 256;
    unsigned mine = 0;
    for (int c = tid * per; c < min(k, tid * per + per); c++)
      mine += wh[c] + wh[k + c] + wh[2 * k + c] + wh[3 * k + c];
    unsigned incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const unsigned up = __shfl_up(incl, o);
      if (lane >= o) incl += up;
    }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    unsigned base = incl - mine;
    for (int w = 0; w < wave; w++) base += wave_tot[w];
    for (int c = tid * per; c < min(k, tid * per + per); c++) {
      const unsigned t0 = wh[c], t1 = wh[k + c], t2 = wh[2 * k + c], t3 = wh[3 * k + c];
      gpos[c] = D.start[c] + D.gtot[(size_t)grp * k + c] + D.hist[(size_t)chunk * k + c] - base;
      wh[c] = base; wh[k + c] = base + t0; wh[2 * k + c] = base + t0 + t1; wh[3 * k + c] = base + t0 + t1 + t2;
      base += t0 + t1 + t2 + t3;
    }
  } else {
    for (int c = tid; c < k; c += 256) {
      const unsigned t0 = wh[c], t1 = wh[k + c], t2 = wh[2 * k + c];
      const unsigned base = D.start[c] + D.gtot[(size_t)grp * k + c] + D.hist[(size_t)chunk * k + c];
      wh[c] = base; wh[k + c] = base + t0; wh[2 * k + c] = base + t0 + t1; wh[3 * k + c] = base + t0 + t1 + t2;
    }
  }
  __syncthreads();
  // C: placement
#pragma unroll
  for (int t = 0; t < 4; t++) {
    const long long r = wr0 + t * 64 + lane;
    const bool valid = r < r1;
    const int key = keys[t];
    const unsigned b = valid ? wh[wave * k + key] : 0u;
    const unsigned pos = b + (unsigned)__popcll(same[t] & lt);
    if (valid && (same[t] & lt) == 0ull) wh[wave * k + key] = b + (unsigned)__popcll(same[t]);
    if (STAGED) {
      if (valid) { dstrow[pos] = (int)(gpos[key] + pos); lpos[wave * 256 + t * 64 + lane] = (unsigned short)pos; }
    } else if (valid) {
      const auto dst = D.xb + (size_t)pos * sp;
      const auto src = D.x + (size_t)r * D.ld + D.from;
      int j = 0;
      for (; j + 2 <= s; j += 2) *reinterpret_cast<gptr<f32x2>>(dst + j) = f32x2{src[j], src[j + 1]};
      if (j < s) *reinterpret_cast<gptr<f32x2>>(dst + j) = f32x2{src[j], src[j]};   // odd s: the padding column mirrors the last one
    }
  }
  if (!STAGED) return;
  // every prefetched element to the chunk-local position of its row (same-wave LDS order: lpos is written above)
  {
    const unsigned rdiv = (65536u + (unsigned)s - 1u) / (unsigned)s;   // e / s == (e * rdiv) >> 16 for e < 4096, s <= 16
    const int total = wrows * s;
#pragma unroll
    for (int u = 0; u < NV; u++) {
      if (u < s) {
        const float vv[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
        for (int c4 = 0; c4 < 4; c4++) {
          const int e = 4 * (lane + 64 * u) + c4;
          if (e < total) {
            const int row = (int)(((unsigned)e * rdiv) >> 16);
            const int j = e - row * s;
            float *dstp = sorted + (size_t)lpos[wave * 256 + row] * sp + j;
            dstp[0] = vv[c4];
            if (j + 1 == s && sp != s) dstp[1] = vv[c4];   // odd s: the padding column mirrors the last one (update_chains_pk)
          }
        }
      }
    }
  }
  __syncthreads();
  // D: sorted slices leave as float2 units; consecutive units of a cluster's run are consecutive in memory
  const int h = sp >> 1;
  const int units = (int)(r1 - r0) * h;
  const unsigned hdiv = ((1u << 20) + (unsigned)h - 1u) / (unsigned)h;   // u / h == (u * hdiv) >> 20 for u < 8192, h <= 8
  const f32x2 *src2 = reinterpret_cast<const f32x2 *>(sorted);
  const auto dst2 = reinterpret_cast<gptr<f32x2>>(D.xb);
  for (int u = tid; u < units; u += 256) {
    const int lp = (int)(((unsigned long long)(unsigned)u * hdiv) >> 20);
    const int part = u - lp * h;
    dst2[(size_t)dstrow[lp] * h + part] = src2[u];
  }
}

// ---- the running mean's division, off the critical path ---------------------------------------------------
// c <- c + (x - c)/n needs RN((x - c)/n), the correctly rounded quotient (what the JVM's float division gives).
// __fdiv_rn is ~11 dependent instructions; the chain has nothing else to overlap them with (k*s chains per
// problem: ~1.25 waves per SIMD at BASELINE config 3), so its latency IS the kernel's time.  With y = RN(1/n)
// (a table: n is the step number, wave-uniform) Markstein's correction gives the same quotient in three
// dependent operations:   q0 = RN(a y);  r = RN(a - n q0) (exact, one fma);  q = RN(q0 + r y)  ==  RN(a / n)
// whenever y is the correctly rounded reciprocal and nothing under- or overflows (a faithful q0 plus one
// fma-corrected step; Markstein 1990, Cornea-Harrison-Tang 2002).  mean_step_fast is only trusted for
// 2^-60 < |a| < 2^60 (n <= 2^24: every intermediate stays normal); a batch in which any step falls outside is
// recomputed with __fdiv_rn.  gulon_selftest_mean_division checks the identity against __fdiv_rn for EVERY
// n in [1, 2^24) over random and near-halfway numerators (tests/test_gpu_kmeans.py).
__device__ __forceinline__ float mean_quotient_fast(float a, float nf, float y) {
  const float q0 = a * y;
  const float r = __builtin_fmaf(-nf, q0, a);
  return __builtin_fmaf(r, y, q0);
}
// The proofs of the corrected quotient single out divisors whose significand is all ones (for integers: n = 2^j - 1)
// as the ones for which a numerator can exist that the correction rounds the other way; the sampled self-test
// cannot rule such a numerator out, so these ~24 divisors never take the fast path: a batch of 32 steps, divisors
// 32 b + 1 .. 32 b + 32, holds one iff b + 1 is a power of two (b = 0: 1, 3, 7, 15, 31) -- two dozen batches of a
// chain's thousands go through __fdiv_rn.
__device__ __forceinline__ bool batch_has_all_ones_divisor(unsigned b) { return ((b + 1u) & b) == 0u; }
__device__ __forceinline__ bool mean_fast_ok(float a) {
  const float m = fabsf(a);
  return m > 8.673617379884035e-19f /* 2^-60 */ && m < 1.152921504606847e18f /* 2^60 */;
}

__global__ void rcp_table_kernel(float *__restrict__ out, long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = __fdiv_rn(1.0f, (float)(int)(i + 1));
}

#ifdef GULON_TEST_HOOKS
// mismatches of mean_quotient_fast against __fdiv_rn: every divisor n in [1, n_max], `per` numerators each --
// random ones and ones built to sit next to a rounding boundary (n times a random quotient, one ulp up and down)
__global__ void mean_division_selftest(int n_max, int per, unsigned long long seed, unsigned long long *__restrict__ bad) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long total = (long long)n_max * per;
  unsigned long long mism = 0;
  for (long long w = t; w < total; w += (long long)gridDim.x * blockDim.x) {
    const int n = (int)(w / per) + 1;
    const int e = (int)(w % per);
    unsigned long long z = seed + (unsigned long long)w * 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL; z ^= z >> 31;
    const float nf = (float)n;
    // the reciprocal the streamed update computes in its loop (kmeans_stream.hip): every divisor it is used for
    if (e == 0 && __float_as_uint(rcp_rn_int(nf)) != __float_as_uint(__fdiv_rn(1.0f, nf))) mism++;
    // a random float with exponent in [-40, 40]
    const unsigned mant = (unsigned)(z & 0x7FFFFF), sgn = (unsigned)((z >> 23) & 1);
    const int ex = (int)((z >> 24) % 81) - 40;
    float a = __uint_as_float((sgn << 31) | ((unsigned)(ex + 127) << 23) | mant);
    if (e & 1) {   // a = RN(n * q) nudged by -1, 0, +1 ulp: quotients next to q and to its rounding boundaries
      const float q = a;
      a = nf * q;
      const int nudge = (int)((z >> 40) % 3) - 1;
      a = __uint_as_float(__float_as_uint(a) + (unsigned)nudge);
    }
    if (!mean_fast_ok(a)) continue;
    const float y = __fdiv_rn(1.0f, nf);
    const float fast = mean_quotient_fast(a, nf, y);
    const float ref = __fdiv_rn(a, nf);
    if (__float_as_uint(fast) != __float_as_uint(ref)) mism++;
  }
  // the divisors 2^j - 1 (the integers nearest to the all-ones significands the proofs single out): EVERY numerator
  // significand and sign of one binade -- the arithmetic is scale invariant between the under- and overflow ranges
  // the callers keep out of -- with the reciprocal of the streamed update (tests/test_markstein_exhaustive.py is the
  // same sweep on the CPU)
  for (long long w = t; w < (24ll << 24); w += (long long)gridDim.x * blockDim.x) {
    const int j = (int)(w >> 24) + 1;
    if ((1 << j) - 1 > n_max) break;
    const float nf = (float)((1 << j) - 1);
    const float a = __uint_as_float((((unsigned)w >> 23) & 1u) << 31 | 127u << 23 | ((unsigned)w & 0x7FFFFFu));
    const float fast = mean_quotient_fast(a, nf, rcp_rn_int(nf));
    if (__float_as_uint(fast) != __float_as_uint(__fdiv_rn(a, nf))) mism++;
  }
  if (mism) atomicAdd(bad, mism);
}
#endif

// one thread per (cluster, dim): c_j <- c_j + (x_j - c_j)/n over the cluster's rows
// in row order (KMeans.scala:211-224); int->float RNE, correctly rounded quotient (above).
// The cluster's slices are contiguous in xb (sort_place).  A three-deep software pipeline of 32-step batches
// keeps 64 steps of loads in flight per lane (no load depends on the chain).  The kernel is one wave per 64
// chains and there are barely more waves than SIMDs (1280 at BASELINE config 3, two on some SIMDs), so it is
// as fast as its instruction stream is short: SP (the bucket row stride) is a template parameter so that every
// load of a batch is one instruction with an immediate offset from a pointer that advances once per batch, the
// main loop runs clear of the cluster's end (no index clamps), and the range test of the fast quotient is one
// min and one max per step, looked at once per batch.
template <int SP /* bucket row stride in floats; 0: read it from the descriptor */>
__global__ __launch_bounds__(64) void update_chains(const UpdDesc *__restrict__ descs, int k,
                                                    const float *__restrict__ rcp /* scalar loads: see update_chains_pk */) {
  const UpdDescG D = load_desc(descs, blockIdx.x);           // x = problem, y = block of 64 chains: blocks are dispatched x-fastest,
  const int s = D.s, sp = SP ? SP : ((s + 1) & ~1);   // so every problem's longest chains (block 0) come first
  int t = blockIdx.y * blockDim.x + threadIdx.x;
  if (t >= k * s) return;
  const int slot = t / s, j = t - slot * s;
  const int c = D.corder ? D.corder[slot] : slot;
  const unsigned len = D.count[c];
  float p = 0.f;
  if (len == 0) { D.cout[c * s + j] = 0.f; return; }
  const auto col = D.xb + (size_t)D.start[c] * sp + j;
  constexpr int U = 32;
  const unsigned last = len - 1;
  unsigned i = 0;
  if (len >= 3 * U) {
    float xa[U], xb[U], xc[U];
#pragma unroll
    for (int u = 0; u < U; u++) xa[u] = col[(size_t)u * sp];
#pragma unroll
    for (int u = 0; u < U; u++) xb[u] = col[(size_t)(U + u) * sp];
    auto nxt = col + (size_t)2 * U * sp;
    // batches whose look-ahead (two batches) stays inside the cluster
    for (; i + 3 * U <= len; i += U, nxt += (size_t)U * sp) {
#ifdef GULON_CHAINS_NOLOAD   // timing experiment (wrong results): the recurrence alone, no loads in the loop --
#pragma unroll              // 5.5 ms against 8.4 ms for the first update at BASELINE config 3 (longest chain: 118 K steps)
      for (int u = 0; u < U; u++) xc[u] = xa[u] * 1.0001f;
#else
#pragma unroll
      for (int u = 0; u < U; u++) xc[u] = nxt[(size_t)u * sp];
#endif
      const float p0 = p;
      float lo = INFINITY, hi = 0.f;
      float nf = (float)(int)(i + 1);
#pragma unroll
      for (int u = 0; u < U; u++) {
        const float a = xa[u] - p;
        lo = fminf(lo, fabsf(a));
        hi = fmaxf(hi, fabsf(a));
        p = p + mean_quotient_fast(a, nf, rcp[i + u]);
        nf += 1.0f;                                   // exact below 2^24
      }
      // a zero, tiny, huge or NaN numerator somewhere in the batch, or a divisor whose significand is all ones
      // (2^j - 1: see mean_quotient_fast) among its 32: the plain division, from the batch's start
      if (!__all(lo > 8.673617379884035e-19f /* 2^-60 */ && hi < 1.152921504606847e18f /* 2^60 */) ||
          batch_has_all_ones_divisor(i / U)) {
        p = p0;
#pragma unroll
        for (int u = 0; u < U; u++) p = p + __fdiv_rn(xa[u] - p, (float)(int)(i + u + 1));
      }
#pragma unroll
      for (int u = 0; u < U; u++) { xa[u] = xb[u]; xb[u] = xc[u]; }
    }
    // the two batches already in registers
#pragma unroll
    for (int u = 0; u < U; u++) p = p + __fdiv_rn(xa[u] - p, (float)(int)(i + u + 1));
    i += U;
#pragma unroll
    for (int u = 0; u < U; u++) p = p + __fdiv_rn(xb[u] - p, (float)(int)(i + u + 1));
    i += U;
  }
  for (; i <= last; i++) p = p + __fdiv_rn(col[(size_t)i * sp] - p, (float)(int)(i + 1));
  D.cout[c * s + j] = p;
}

// Two chains per lane on the packed fp32 pipe.  The chain kernel is latency bound -- five dependent operations per
// step, nothing to overlap them with -- and with one (cluster, dim) per lane BASELINE config 3 needs 1280 waves for
// 1024 SIMDs: the SIMDs holding two pay every step twice.  v_pk_add/mul/fma_f32 carry two IEEE binary32 lanes per
// register pair at the cost of one instruction, so lane = (cluster, PAIR of dims): 640 waves, one per SIMD, and the
// per-step instruction stream (5 packed + min3 + max3 + one 8-byte load) is no longer than the scalar one was.
// The bucket row stride is even (sort_place), so the pair is one aligned 8-byte load; the second half of the last
// pair of an odd s walks the padding column, which sort_place fills with a copy of the last real column: a chain
// that behaves exactly like its neighbour (same range checks, same path) and is never stored.  (Mirroring it here,
// one v_cndmask per loaded value, made every load wait for its data at once.)
// The wave walks its batches in LOCKSTEP: the trip count is the maximum over its lanes and every lane stays active
// to the end -- a lane whose cluster is exhausted keeps its result aside, re-reads its last batch (in bounds, cached)
// and is left out of the range test.  That makes the step number wave-uniform with all 64 lanes alive, so the 32
// reciprocals of a batch can travel like a 33rd column: lane u of `ycur` holds RN(1 / (i + u + 1)), requested two
// batches ahead by one coalesced load, and every step reads its own with v_readlane; the divisors are converted once
// per batch the same way.  (As scalar loads they were requested at the top of the batch that needs them and waited
// for three times per batch -- a third of the time of a wave that has its SIMD to itself; as per-lane vector loads,
// round 1, they sat behind the look-ahead loads in vmcnt order: 3 of the first update's 8.4 ms at BASELINE config 3.)
// What the batches leave of a cluster (fewer than three batches) is walked with the plain division at the end.
template <int SP /* bucket row stride in floats (even) */>
__global__ __launch_bounds__(64) void update_chains_pk(const UpdDesc *__restrict__ descs, int k,
                                                       const float *__restrict__ rcp) {
  static_assert(SP % 2 == 0 && SP >= 2, "even stride");
  constexpr int HP = SP / 2;
  constexpr int U = 32;
  const UpdDescG D = load_desc(descs, blockIdx.x);           // x = problem, y = block of 64 lanes: every problem's longest chains first
  const int s = D.s;
  const int t = blockIdx.y * blockDim.x + threadIdx.x;
  const int slot = t / HP, jp = t - slot * HP, j = 2 * jp;
  const bool valid = t < k * HP && j < s;          // (j >= s: a problem narrower than the launch's stride)
  const bool pad = j + 1 >= s;
  const int c = valid ? (D.corder ? D.corder[slot] : slot) : 0;
  const unsigned len = valid ? D.count[c] : 0u;
  const auto col = reinterpret_cast<gptr<const f32x2>>(D.xb + (valid ? (size_t)D.start[c] * SP + j : (size_t)0));
  auto slow = [](f32x2 p, f32x2 x, unsigned n1) {
    const float nf = (float)(int)n1;
    p.x = p.x + __fdiv_rn(x.x - p.x, nf);
    p.y = p.y + __fdiv_rn(x.y - p.y, nf);
    return p;
  };
  // batches of this lane (the look-ahead of two batches stays inside the cluster), and of the wave
  const unsigned nb = len >= 3 * U ? (len - 3 * U) / U + 1 : 0u;
  unsigned nb_wave = nb;
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) nb_wave = max(nb_wave, (unsigned)__shfl_xor((int)nb_wave, o));
  nb_wave = (unsigned)__builtin_amdgcn_readfirstlane((int)nb_wave);
  f32x2 p = {0.f, 0.f}, p_keep = {0.f, 0.f};
  if (nb_wave > 0) {       // some cluster of the wave has >= 96 rows, so 96 rows from the bucket's base are in bounds
    const int l32 = threadIdx.x & 31;
    auto nxt = nb > 0 ? col : reinterpret_cast<gptr<const f32x2>>(D.xb);   // lanes without batches read the base rows
    unsigned i = 0, b = 0;
    // one batch: the look-ahead loads of the batch after next into `ld`, then the 32 steps of `cur`
    auto batch = [&](const f32x2 (&cur)[U], float ycur, f32x2 (&ld)[U], float &yld) {
      const bool live = b < nb;                    // this lane's cluster still has this batch
      const auto src = nxt + (size_t)(live ? 2 * U * HP : 0);   // (a finished lane re-reads where it stands: inside its cluster)
#ifdef GULON_CHAINS_NOLOAD   // timing experiment (wrong results): the recurrence alone, no loads in the loop
#pragma unroll
      for (int u = 0; u < U; u++) ld[u] = cur[u] * 1.0001f;
      yld = ycur;
#else
#pragma unroll
      for (int u = 0; u < U; u++) ld[u] = src[(size_t)u * HP];
      yld = rcp[i + 2 * U + l32];
#endif
      const f32x2 p0 = p;
      float lo = INFINITY, hi = 0.f;
      const float nfv = (float)(int)(i + 1 + l32);
#pragma unroll
      for (int u = 0; u < U; u++) {
        const f32x2 a = cur[u] - p;
        lo = fminf(fminf(lo, fabsf(a.x)), fabsf(a.y));
        hi = fmaxf(fmaxf(hi, fabsf(a.x)), fabsf(a.y));
        const float y = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ycur), u));
        const float nf = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(nfv), u));
        const f32x2 y2 = {y, y}, nn = {-nf, -nf};
        const f32x2 q0 = a * y2;
        const f32x2 r = __builtin_elementwise_fma(nn, q0, a);
        p = p + __builtin_elementwise_fma(r, y2, q0);
      }
      // a zero, tiny, huge or NaN numerator somewhere in a live lane's batch, or a divisor whose significand is all ones
      // (2^j - 1: see mean_quotient_fast) among its 32: the plain division, from the batch's start
      if (!__all(!live || (lo > 8.673617379884035e-19f /* 2^-60 */ && hi < 1.152921504606847e18f /* 2^60 */)) ||
          batch_has_all_ones_divisor(b)) {
        p = p0;
#pragma unroll
        for (int u = 0; u < U; u++) p = slow(p, cur[u], i + u + 1);
      }
      if (live) { p_keep = p; nxt += (size_t)U * HP; }
      i += U;
      b++;
    };
    f32x2 xa[U], xb[U], xc[U];
    float ya, yb, yc;
#pragma unroll
    for (int u = 0; u < U; u++) xa[u] = nxt[(size_t)u * HP];
    ya = rcp[l32];
#pragma unroll
    for (int u = 0; u < U; u++) xb[u] = nxt[(size_t)(U + u) * HP];
    yb = rcp[U + l32];
    // three batches per trip, the buffers taking turns (no register copies)
    while (b + 3 <= nb_wave) { batch(xa, ya, xc, yc); batch(xb, yb, xa, ya); batch(xc, yc, xb, yb); }
    if (b < nb_wave) batch(xa, ya, xc, yc);
    if (b < nb_wave) batch(xb, yb, xa, ya);
  }
  // what the batches left of the cluster: fewer than three batches' worth, from memory, the plain division
  p = p_keep;
  for (unsigned i = nb * U; i < len; i++) p = slow(p, col[(size_t)i * HP], i + 1);
  if (valid) {
    D.cout[c * s + j] = p.x;
    if (!pad) D.cout[c * s + j + 1] = p.y;
  }
}

// ---------------------------------------------------------------------------
// host drivers
// ---------------------------------------------------------------------------
static int pick_smax(int s) {
  if (s <= 4) return 4;
  if (s <= 8) return 8;
  if (s <= 10) return 10;   // (BASELINE config 3's sub-vectors are 9 and 10 wide: padded to 16 the exact re-check did 60-78 %
  if (s <= 12) return 12;   //  more products than it had to)
  if (s <= 16) return 16;
  if (s <= 32) return 32;
  if (s <= 64) return 64;
  if (s <= 128) return 128;
  return s;
}

KmeansWorkspace::~KmeansWorkspace() { if (host) (void)hipHostFree(host); }

void KmeansWorkspace::ensure(int n, int k, int s) {
  if (!host) HIP_CHECK(hipHostMalloc((void **)&host, sizeof(HostWords)));
  int smax = pick_smax(s);
  cpad.ensure((size_t)k * smax);
  off.ensure(k);
  ties.ensure((size_t)std::max(n, 1));
  local.ensure((size_t)std::max(n, 1));
  tie_total.ensure(1);
  long long nchunks = ceil_div(std::max(n, 1), CHUNK_ROWS);
  const bool bigk = sizeof(unsigned) * 4 * (size_t)k > 160 * 1024;   // radix path: its scratch lives in the call
  if (!bigk) {
    hist.ensure((size_t)nchunks * k);
    gtot.ensure((size_t)ceil_div(nchunks, SCAN_GROUP) * k);
  }
  count.ensure(k);
  start.ensure(k);
  if (!bigk) xb.ensure((size_t)std::max(n, 1) * (size_t)((s + 1) & ~1));
  corder.ensure(k);
  mismatch.ensure(1);
}

// ---------------------------------------------------------------------------
// KMeans.assign / parAssign on device arrays, as three enqueue stages separated by stream
// synchronisations (the host needs two counters: #flagged rows, #RNG draws).  The stages
// let the trainer run many independent problems on their own streams and pay the
// synchronisations once per iteration instead of once per problem.
// d_assign is written only where a centroid won (caller initialises it).
// rng_batch <= 0: one java.util.Random stream over all rows.
// ---------------------------------------------------------------------------
static void launch_exact(AssignJob &j) {
  KmeansWorkspace &ws = *j.ws;
  const int smax = pick_smax(j.s);
  const int grid = ceil_div(j.nrows, 256);
  // centroids + offsets in LDS when they fit 64 KiB (k = 256: 17 KiB at s <= 16)
#define AE(S)                                                                                                      \
  do {                                                                                                             \
    const size_t lds_ = sizeof(float) * (size_t)((j.k + 1) / 2) * 2 * (S + 1);                                     \
    if (lds_ <= 64 * 1024) {                                                                                       \
      HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(assign_exact<S, true>),                         \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_));                        \
      hipLaunchKernelGGL((assign_exact<S, true>), dim3(grid), dim3(256), lds_, j.st, j.dX, j.n, j.ld, j.from, j.s, \
                         ws.cpad.p, ws.off.p, j.k, j.rows, j.nrows, j.d_assign, ws.ties.p, ws.tie_total.p);        \
    } else {                                                                                                       \
      hipLaunchKernelGGL((assign_exact<S, false>), dim3(grid), dim3(256), 0, j.st, j.dX, j.n, j.ld, j.from, j.s,   \
                         ws.cpad.p, ws.off.p, j.k, j.rows, j.nrows, j.d_assign, ws.ties.p, ws.tie_total.p);        \
    }                                                                                                              \
  } while (0)
  switch (smax) {
    case 4: AE(4); break;
    case 8: AE(8); break;
    case 10: AE(10); break;
    case 12: AE(12); break;
    case 16: AE(16); break;
    case 32: AE(32); break;
    case 64: AE(64); break;
    case 128: AE(128); break;
    default:
      hipLaunchKernelGGL(assign_exact_generic, dim3(grid), dim3(256), 0, j.st, j.dX, j.n, j.ld, j.from, j.s, ws.cpad.p,
                         smax, ws.off.p, j.k, j.rows, j.nrows, j.d_assign, ws.ties.p, ws.tie_total.p);
  }
#undef AE
  HIP_CHECK(hipGetLastError());
  HIP_CHECK(hipMemcpyAsync(&ws.host->total, ws.tie_total.p, sizeof(unsigned long long), hipMemcpyDeviceToHost, j.st));
}

// stage 1: centroid prep, then either the MFMA filter (+ flagged-row count) or the exact
// scan of every row (+ draw count)
void assign_stage1(AssignJob &j) {
  j.done = j.n <= 0;
  if (j.done) return;
  KmeansWorkspace &ws = *j.ws;
  ws.ensure(j.n, j.k, j.s);
  ws.last_draws = 0;
  ws.last_flagged = 0;
  const int smax = pick_smax(j.s);
  hipLaunchKernelGGL(prep_centroids, dim3(ceil_div(j.k, 64)), dim3(64), 0, j.st, j.dC, j.k, j.s, smax, ws.cpad.p,
                     ws.off.p);
  HIP_CHECK(hipMemsetAsync(ws.tie_total.p, 0, sizeof(unsigned long long), j.st));
  j.filtered = j.ps != nullptr && mfma_assign_supported(j.s, j.k);
  if (j.filtered) {
    assign_mfma_filter(ws, *j.ps, j.dC, j.k, j.d_assign, j.st);
    HIP_CHECK(hipMemcpyAsync(&ws.host->flagged, ws.flag_count.p, sizeof(unsigned), hipMemcpyDeviceToHost, j.st));
  } else {
    j.rows = nullptr;
    j.nrows = j.n;
    launch_exact(j);
  }
}

// stage 2 (after a stream sync): exact scan of the flagged rows, or -- unfiltered -- the
// tie replay
constexpr unsigned long long SPARSE_TIE_DRAWS = 8192;   // up to this many draws the positions come from the drawing rows alone

static void launch_tie_replay(AssignJob &j, bool sparse = false) {
  KmeansWorkspace &ws = *j.ws;
  const int n = j.n, smax = pick_smax(j.s);
  const int seg_len = j.rng_batch > 0 ? j.rng_batch : n;
  if (sparse) {   // (filtered jobs, wave-per-row replay: ws.ties holds the flagged rows' counts, indexed by row)
    ws.tie_rows.ensure((size_t)std::max(j.nrows, 1));
    ws.tie_count.ensure(1);
    ws.tie_pos.ensure((size_t)SPARSE_TIE_DRAWS);
    HIP_CHECK(hipMemsetAsync(ws.tie_count.p, 0, sizeof(unsigned), j.st));
    hipLaunchKernelGGL(collect_tie_rows, dim3(ceil_div(j.nrows, 256)), dim3(256), 0, j.st, ws.ties.p, j.rows, j.nrows,
                       ws.tie_rows.p, ws.tie_count.p);
    hipLaunchKernelGGL(tie_positions_sparse, dim3((unsigned)ceil_div((long long)SPARSE_TIE_DRAWS, 256LL)), dim3(256), 0, j.st,
                       ws.ties.p, ws.tie_rows.p, ws.tie_count.p, seg_len, ws.tie_pos.p);
    hipLaunchKernelGGL(assign_resolve_wave, dim3(256), dim3(64), sizeof(float) * (size_t)j.s, j.st, j.dX, j.ld, j.from,
                       j.s, ws.cpad.p, smax, ws.off.p, j.k, (const unsigned *)nullptr, (const unsigned long long *)nullptr,
                       seg_len, 0, ws.tie_rows.p, ws.tie_count.p, j.d_assign, ws.tie_pos.p);
    HIP_CHECK(hipGetLastError());
    return;
  }
  const int nseg = ceil_div(n, seg_len);
  const int bps = ceil_div(seg_len < n ? seg_len : n, 1024);
  ws.block_tot.ensure((size_t)nseg * bps);
  ws.block_off.ensure((size_t)nseg * bps);
  hipLaunchKernelGGL(tie_block_sums, dim3(bps, nseg), dim3(1024), 0, j.st, ws.ties.p, n, seg_len, bps, ws.local.p,
                     ws.block_tot.p);
  hipLaunchKernelGGL(tie_block_scan, dim3(nseg), dim3(1024), 0, j.st, ws.block_tot.p, bps, ws.block_off.p);
  const int *rows = j.filtered ? j.rows : nullptr;
  const int nrows = j.filtered ? j.nrows : n;
  if ((long long)j.k * j.s >= 1024) {   // (all but tiny problems) one wave per drawing row: a thread walking k * s products is a long latency chain
    ws.tie_rows.ensure((size_t)std::max(nrows, 1));
    ws.tie_count.ensure(1);
    HIP_CHECK(hipMemsetAsync(ws.tie_count.p, 0, sizeof(unsigned), j.st));
    hipLaunchKernelGGL(collect_tie_rows, dim3(ceil_div(nrows, 256)), dim3(256), 0, j.st, ws.ties.p, rows, nrows,
                       ws.tie_rows.p, ws.tie_count.p);
    hipLaunchKernelGGL(assign_resolve_wave, dim3(1024), dim3(64), sizeof(float) * (size_t)j.s, j.st, j.dX, j.ld, j.from,
                       j.s, ws.cpad.p, smax, ws.off.p, j.k, ws.local.p, ws.block_off.p, seg_len, bps, ws.tie_rows.p,
                       ws.tie_count.p, j.d_assign, (const unsigned long long *)nullptr);
    HIP_CHECK(hipGetLastError());
    return;
  }
  hipLaunchKernelGGL(assign_resolve<0>, dim3(ceil_div(nrows, 256)), dim3(256), 0, j.st, j.dX, n, j.ld, j.from, j.s,
                     ws.cpad.p, smax, ws.off.p, j.k, ws.ties.p, ws.local.p, ws.block_off.p, seg_len, bps, rows, nrows,
                     j.d_assign);
  HIP_CHECK(hipGetLastError());
}

void assign_stage2(AssignJob &j) {
  if (j.done) return;
  KmeansWorkspace &ws = *j.ws;
  if (j.filtered) {
    ws.last_flagged = ws.host->flagged;
    if (ws.host->flagged == 0) { j.done = true; return; }
    j.rows = ws.flag_rows.p;
    j.nrows = (int)ws.host->flagged;
    launch_exact(j);
  } else {
    ws.last_draws = ws.host->total;
    if (ws.host->total) launch_tie_replay(j);
    j.done = true;
  }
}

// stage 3 (after a stream sync; filtered jobs only): tie replay over the flagged rows
void assign_stage3(AssignJob &j) {
  if (j.done) return;
  KmeansWorkspace &ws = *j.ws;
  ws.last_draws = ws.host->total;
  if (ws.host->total && ws.host->total <= SPARSE_TIE_DRAWS && (long long)j.k * j.s >= 1024) {
    launch_tie_replay(j, true);     // few draws: positions from the drawing rows themselves
  } else if (ws.host->total) {
    // draw counts exist only for the flagged rows: build the dense per-row array (0 elsewhere)
    HIP_CHECK(hipMemsetAsync(ws.local.p, 0, sizeof(unsigned) * (size_t)j.n, j.st));
    hipLaunchKernelGGL(scatter_ties, dim3(ceil_div(j.nrows, 256)), dim3(256), 0, j.st, j.rows, j.nrows, ws.ties.p,
                       ws.local.p);
    HIP_CHECK(hipMemcpyAsync(ws.ties.p, ws.local.p, sizeof(unsigned) * (size_t)j.n, hipMemcpyDeviceToDevice, j.st));
    launch_tie_replay(j);
  }
  j.done = true;
}

void kmeans_assign_dev(KmeansWorkspace &ws, const float *dX, int n, int ld, int from, int s, const float *dC, int k,
                       int rng_batch, int *d_assign, hipStream_t st, const PackedSlice *ps) {
  AssignJob j;
  j.ws = &ws; j.dX = dX; j.n = n; j.ld = ld; j.from = from; j.s = s; j.dC = dC; j.k = k; j.rng_batch = rng_batch;
  j.d_assign = d_assign; j.st = st; j.ps = ps;
  assign_stage1(j);
  if (!j.done) { HIP_CHECK(hipStreamSynchronize(st)); assign_stage2(j); }
  if (!j.done) { HIP_CHECK(hipStreamSynchronize(st)); assign_stage3(j); }
}

// ---- more clusters than the counting sort's LDS counters hold (k > 10240, up to the 65536 of
// ProductQuantizer.coderFactory, ProductQuantizer.scala:11-16): the stable order comes from a two-pass LSD radix
// sort of (row, cluster) pairs -- each pass is the counting sort above with 256 buckets (one byte of the cluster
// id as the key, the 8-byte pair as the "slice") -- and the chains read their rows through the sorted row ids.
__global__ void bigk_pairs(const int *__restrict__ assign, int n, int2 *__restrict__ pairs, int *__restrict__ digit) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const int a = assign[r];
  pairs[r] = make_int2(r, a);
  digit[r] = a & 255;
}
__global__ void bigk_digit2(const int2 *__restrict__ pairs, int n, int *__restrict__ digit) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) digit[i] = (pairs[i].y >> 8) & 255;
}
__global__ void bigk_count(const int *__restrict__ assign, int n, unsigned *__restrict__ count) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r < n) atomicAdd(&count[assign[r]], 1u);
}
__global__ __launch_bounds__(1024) void bigk_starts(const unsigned *__restrict__ count, int k, unsigned *__restrict__ start) {
  __shared__ unsigned part[1024];
  const int per = (k + 1023) / 1024, t = threadIdx.x;
  unsigned sum = 0;
  for (int c = t * per; c < min(k, t * per + per); c++) sum += count[c];
  part[t] = sum;
  __syncthreads();
  if (t == 0) { unsigned run = 0; for (int i = 0; i < 1024; i++) { const unsigned v = part[i]; part[i] = run; run += v; } }
  __syncthreads();
  unsigned run = part[t];
  for (int c = t * per; c < min(k, t * per + per); c++) { start[c] = run; run += count[c]; }
}
// one thread per (cluster, dim), rows through the sorted pairs (KMeans.scala:211-224; IEEE division)
__global__ void update_chains_indirect(const int2 *__restrict__ pairs, const unsigned *__restrict__ count,
                                       const unsigned *__restrict__ start, const float *__restrict__ X, int ld, int from,
                                       int s, int k, float *__restrict__ cout) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)k * s) return;
  const int c = (int)(t / s), j = (int)(t - (long long)c * s);
  const unsigned len = count[c];
  const int2 *ord = pairs + start[c];
  float p = 0.f;
  constexpr int U = 8;
  unsigned i = 0;
  for (; i + U <= len; i += U) {
    float x[U];
#pragma unroll
    for (int u = 0; u < U; u++) x[u] = X[(size_t)ord[i + u].x * ld + from + j];
#pragma unroll
    for (int u = 0; u < U; u++) p = p + __fdiv_rn(x[u] - p, (float)(int)(i + u + 1));
  }
  for (; i < len; i++) p = p + __fdiv_rn(X[(size_t)ord[i].x * ld + from + j] - p, (float)(int)(i + 1));
  cout[t] = p;
}

static void launch_counting_sort(UpdDesc *d_descs, int np, int n, int k, int smax, bool compact, hipStream_t st) {
  long long nchunks = ceil_div(n, CHUNK_ROWS);
  long long ngroups = ceil_div(nchunks, SCAN_GROUP);
  const int sp_max = (smax + 1) & ~1;
  const bool staged = compact && k <= 1024 && sp_max <= 16;
  const size_t shm_hist = sizeof(unsigned) * (size_t)k;
  const size_t shm_place = staged ? sizeof(unsigned) * (5 * (size_t)k + CHUNK_ROWS + CHUNK_ROWS / 2 + 8) + sizeof(float) * CHUNK_ROWS * (size_t)sp_max
                                  : sizeof(unsigned) * 4 * (size_t)k;
  GULON_UNSUPPORTED(shm_place > 160 * 1024, "internal: counting sort with k = %d needs %zu B of LDS", k, shm_place);
  HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(sort_hist), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)shm_hist));
  hipLaunchKernelGGL(sort_hist, dim3((unsigned)nchunks, np), dim3(256), shm_hist, st, d_descs, n, k);
  hipLaunchKernelGGL(sort_scan_groups, dim3((unsigned)ngroups, ceil_div(k, 256), np), dim3(256), 0, st, d_descs,
                     nchunks, k);
  hipLaunchKernelGGL(sort_scan_top, dim3(1, np), dim3(256), sizeof(unsigned) * (size_t)k, st, d_descs, ngroups, k);
  int key_bits = 0;
  while ((1 << key_bits) < k) key_bits++;
  auto place = !staged ? sort_place<0> : smax <= 4 ? sort_place<4> : smax <= 8 ? sort_place<8> : smax <= 12 ? sort_place<12>
                                                                                                       : sort_place<16>;
  HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(place), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)shm_place));
  static const bool xcd_env = [] { const char *e = getenv("GULON_PLACE_XCD"); return !e || atoi(e) != 0; }();
  const int cpx = xcd_env && staged ? (int)ceil_div(nchunks, 8LL) : 0;
  if (cpx > 0)
    hipLaunchKernelGGL(place, dim3((unsigned)(8LL * cpx * np)), dim3(256), shm_place, st, d_descs, n, k, key_bits, cpx);
  else
    hipLaunchKernelGGL(place, dim3((unsigned)nchunks, np), dim3(256), shm_place, st, d_descs, n, k, key_bits, 0);
  HIP_CHECK(hipGetLastError());
}

static void kmeans_update_bigk(const UpdDesc &D, UpdDesc *d_desc, int n, int k, hipStream_t st) {
  // scratch of this rare path lives for the call (synchronised before it goes)
  DevBuf<int2> p0((size_t)n), p1((size_t)n), p2((size_t)n);
  DevBuf<int> digit((size_t)n);
  const long long nchunks = ceil_div(n, CHUNK_ROWS);
  DevBuf<unsigned> hist((size_t)nchunks * 256), gtot((size_t)ceil_div(nchunks, SCAN_GROUP) * 256), c256(256), s256(256);
  hipLaunchKernelGGL(bigk_pairs, dim3(ceil_div(n, 256)), dim3(256), 0, st, D.assign, n, p0.p, digit.p);
  for (int pass = 0; pass < 2; pass++) {
    UpdDesc S{};
    S.assign = digit.p; S.hist = hist.p; S.gtot = gtot.p; S.count = c256.p; S.start = s256.p;
    S.x = reinterpret_cast<const float *>(pass == 0 ? p0.p : p1.p); S.ld = 2; S.from = 0; S.s = 2;
    S.xb = reinterpret_cast<float *>(pass == 0 ? p1.p : p2.p);
    S.cout = nullptr; S.rcp = nullptr; S.corder = nullptr;
    HIP_CHECK(hipMemcpyAsync(d_desc, &S, sizeof(UpdDesc), hipMemcpyHostToDevice, st));
    HIP_CHECK(hipStreamSynchronize(st));   // S is a stack object
    launch_counting_sort(d_desc, 1, n, 256, 2, true, st);
    if (pass == 0) hipLaunchKernelGGL(bigk_digit2, dim3(ceil_div(n, 256)), dim3(256), 0, st, p1.p, n, digit.p);
  }
  HIP_CHECK(hipMemsetAsync(D.count, 0, sizeof(unsigned) * (size_t)k, st));
  hipLaunchKernelGGL(bigk_count, dim3(ceil_div(n, 256)), dim3(256), 0, st, D.assign, n, D.count);
  hipLaunchKernelGGL(bigk_starts, dim3(1), dim3(1024), 0, st, D.count, k, D.start);
  hipLaunchKernelGGL(update_chains_indirect, dim3((unsigned)ceil_div((long long)k * D.s, 256LL)), dim3(256), 0, st, p2.p,
                     D.count, D.start, D.x, D.ld, D.from, D.s, k, D.cout);
  HIP_CHECK(hipGetLastError());
  HIP_CHECK(hipStreamSynchronize(st));
}

// KMeans.fromAssignment for a batch of problems over the same data -> each D.cout (k x s).
// `d_descs` must hold descs.size() entries of device memory.
void kmeans_update_batch(const std::vector<UpdDesc> &descs, UpdDesc *d_descs, int n, int k, hipStream_t st) {
  const int np = (int)descs.size();
  if (np == 0) return;
  if (n <= 0) {
    for (const UpdDesc &D : descs) HIP_CHECK(hipMemsetAsync(D.cout, 0, sizeof(float) * (size_t)k * D.s, st));
    return;
  }
  if (sizeof(unsigned) * 4 * (size_t)k > 160 * 1024) {   // k > 10240: radix-sorted row ids, one problem at a time
    GULON_UNSUPPORTED(k > 65536, "too many clusters: %d", k);
    for (const UpdDesc &D : descs) kmeans_update_bigk(D, d_descs, n, k, st);
    return;
  }
#ifdef GULON_TEST_HOOKS
  if (kmeans_update_fused(descs, d_descs, n, k, st)) return;   // the measured-and-dropped fused update (kmeans_fused.hip; GULON_UPDATE_FUSED=1)
#endif
  HIP_CHECK(hipMemcpyAsync(d_descs, descs.data(), sizeof(UpdDesc) * np, hipMemcpyHostToDevice, st));
  int smax = 1;
  for (const UpdDesc &D : descs) smax = std::max(smax, D.s);
  bool compact = true;   // every problem reads a compact copy of its slice (ld == s): the staged placement's coalesced loads
  for (const UpdDesc &D : descs) compact = compact && D.ld == D.s && D.from == 0;
  const int sp_max = (smax + 1) & ~1;
  launch_counting_sort(d_descs, np, n, k, smax, compact, st);
  bool one_stride = true;   // every problem with the same bucket row stride: the chains take it as a constant
  for (const UpdDesc &D : descs) one_stride = one_stride && ((D.s + 1) & ~1) == sp_max;
  auto chains = !one_stride ? update_chains<0> : sp_max == 2 ? update_chains<2> : sp_max == 4 ? update_chains<4>
              : sp_max == 6 ? update_chains<6> : sp_max == 8 ? update_chains<8> : sp_max == 10 ? update_chains<10>
              : sp_max == 12 ? update_chains<12> : sp_max == 14 ? update_chains<14> : sp_max == 16 ? update_chains<16>
              : update_chains<0>;
  // one wave per workgroup; with more workgroups than SIMDs, 40 KiB of (unused) LDS each keeps four per CU -- one per
  // SIMD at full issue rate -- and the rest, the SHORTEST chains (size order), start as the first ones finish
  // two chains per lane on the packed pipe wherever the stride is a compile-time constant (GULON_CHAINS_PK=0: one per lane)
  static const bool pk_env = [] { const char *e = getenv("GULON_CHAINS_PK"); return !e || atoi(e) != 0; }();
  const bool pk = pk_env && one_stride && sp_max <= 16;
  if (pk) {
    chains = sp_max == 2 ? update_chains_pk<2> : sp_max == 4 ? update_chains_pk<4> : sp_max == 6 ? update_chains_pk<6>
           : sp_max == 8 ? update_chains_pk<8> : sp_max == 10 ? update_chains_pk<10> : sp_max == 12 ? update_chains_pk<12>
           : sp_max == 14 ? update_chains_pk<14> : update_chains_pk<16>;
  }
  const int chain_blocks = pk ? ceil_div((long long)k * (sp_max / 2), 64) : ceil_div((long long)k * smax, 64);
  int cus = 256;
  { int dev = 0; HIP_CHECK(hipGetDevice(&dev)); HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev)); }
  static const int cap_env = [] { const char *e = getenv("GULON_CHAINS_PER_CU"); return e ? atoi(e) : 4; }();
  size_t chain_lds = 0;
  if (cap_env > 0 && (long long)chain_blocks * np > (long long)cap_env * cus) chain_lds = (size_t)(160 * 1024) / cap_env;
  HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(chains), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)std::max<size_t>(chain_lds, 1)));
  hipLaunchKernelGGL(chains, dim3(np, chain_blocks), dim3(64), chain_lds, st, d_descs, k, descs[0].rcp);
  HIP_CHECK(hipGetLastError());
}

// RN(1/i) for i = 1 .. n: shared by every problem on this device (grown on demand, never shrunk)
static const float *rcp_table(int n) {
  struct Tab { DevBuf<float> buf; };
  static std::mutex mu;
  static std::map<int, Tab> tabs;
  std::lock_guard<std::mutex> lock(mu);
  int dev = 0;
  HIP_CHECK(hipGetDevice(&dev));
  Tab &t = tabs[dev];
  if ((size_t)n > t.buf.n) {
    const long long want = std::max<long long>(n, 1024);
    HIP_CHECK(hipDeviceSynchronize());   // nothing in flight may still read the table about to be replaced
    t.buf.alloc((size_t)want);
    hipLaunchKernelGGL(rcp_table_kernel, dim3((unsigned)ceil_div(want, 256)), dim3(256), 0, 0, t.buf.p, want);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipDeviceSynchronize());
  }
  return t.buf.p;
}

// x/ld/from describe where row r's slice starts: x + r*ld + from
UpdDesc make_upd_desc(KmeansWorkspace &ws, const float *x, int ld, int n, int k, int from, int s, const int *d_assign,
                      float *dC) {
  ws.ensure(n, k, s);
  UpdDesc D;
  D.x = x; D.ld = ld;
  D.assign = d_assign; D.hist = ws.hist.p; D.gtot = ws.gtot.p; D.count = ws.count.p; D.start = ws.start.p;
  D.xb = ws.xb.p; D.cout = dC; D.from = from; D.s = s;
  D.rcp = rcp_table(n);
  D.corder = k <= 2048 ? ws.corder.p : nullptr;
  return D;
}

// single-problem form
void kmeans_update_dev(KmeansWorkspace &ws, const float *dX, int n, int ld, int from, int s, int k,
                       const int *d_assign, float *dC, hipStream_t st) {
  std::vector<UpdDesc> descs{make_upd_desc(ws, dX, ld, n, k, from, s, d_assign, dC)};
  ws.descs.ensure(1);
  kmeans_update_batch(descs, ws.descs.p, n, k, st);
  // descs is read by hipMemcpyAsync from pageable memory: staged before the call returns
}

static void validate_assignments(const int32_t *a, int n, int k) {
  for (int i = 0; i < n; i++)
    GULON_REQUIRE(a[i] >= 0 && a[i] < k, "assignment %d at row %d is outside [0,%d)", a[i], i, k);
}

// SummaryStats builder over MathUtils.distance(prev_c, next_c) (KMeans.scala:160-168,
// MathUtils.scala:43-57,85-98): k values, host side.
static void step_stats(const float *prev, const float *next, int k, int s, gulon_kmeans_report *r) {
  int n = 0;
  float m = 0.f, ss = 0.f;
  for (int c = 0; c < k; c++) {
    float sum = 0.f;
    for (int j = 0; j < s; j++) {
      float dx = next[(size_t)c * s + j] - prev[(size_t)c * s + j];
      sum += dx * dx;
    }
    float x = (float)std::sqrt((double)sum);
    n += 1;
    float m0 = m;
    m = m0 + (x - m0) / (float)n;
    ss = ss + (x - m0) * (x - m);
  }
  r->step_count = n;
  r->step_mean = m;
  r->step_s = ss;
}

TrainTrace &train_trace() { static TrainTrace t; return t; }

// KMeans.computeClusters (KMeans.scala:134-157) for `np` independent problems
// (from[p], s[p], seed[p]) over the same n x ld data, run iteration-synchronously: every
// problem owns a stream and a workspace, so the small latency-bound kernels (sequential
// update chains, sorts, tie replays) of different sub-quantizers overlap on the GPU.
// c_out[p] receives k x s[p] floats.
void kmeans_train_batch(const float *dX, int n, int ld, int np, const int *from, const int *sdim, const int *seeds,
                        int k, int max_iterations, float *const *c_out, gulon_kmeans_report *reports,
                        int max_reports, int32_t *n_reports) {
  GULON_REQUIRE(n >= 1, "KMeans.init needs at least one row (n = %d)", n);   // rng.nextInt(0) throws on the JVM
  const bool print = getenv("GULON_TRACE") != nullptr;
  TrainTrace &tt = train_trace();
  const bool trace = print || tt.on;
  auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double t_mark = now();
  // stage timing (GULON_TRACE=1 prints it, gulon_kmeans_trace collects it): a device synchronisation closes
  // every stage, so the stages of the m concurrent problems are timed as a whole, one after the other
  auto lap = [&](const char *what, double *acc = nullptr) {
    if (!trace) return;
    (void)hipDeviceSynchronize();
    double t = now();
    if (print) fprintf(stderr, "[gulon trace] %-28s %8.2f ms\n", what, t - t_mark);
    if (acc && tt.on) *acc += t - t_mark;
    t_mark = t;
  };
  struct Prob {
    KmeansWorkspace ws;
    hipStream_t st = nullptr;
    hipEvent_t assigned = nullptr;   // this problem's assignment of the iteration is final
    DevBuf<float> c_prev, c_next;
    DevBuf<int> a_prev, a_next, d_rows;
    DevBuf<unsigned> mism;
    // host copies of the centroids and of the mismatch counter: PINNED, so that the per-iteration downloads of all
    // problems are truly asynchronous (into pageable memory every hipMemcpyAsync stages and blocks: 64 of them were
    // most of the "convergence test" stage)
    float *h_prev = nullptr, *h_next = nullptr;
    unsigned *h_mism = nullptr;
    PackedSlice packed;      // MFMA-ready copy of this problem's column slice
    DevBuf<float> xs;        // row-major n x s copy of the slice: compact target of the update's gathers
    // the streamed update (kmeans_stream.hip): the slice pair-major, and per-update scratch (chunk-local order, offsets)
    DevBuf<float> xp;
    DevBuf<unsigned short> s_ord, s_coff;
    DevBuf<int> s_wild;
    AssignJob job;
    bool use_mfma = false;
    bool done = false;
    bool shared_stream = false;
    int nrep = 0;
    ~Prob() {
      if (assigned) (void)hipEventDestroy(assigned);
      if (st && !shared_stream) (void)hipStreamDestroy(st);
    }
  };
  std::vector<Prob> P(np);
  // ONE pinned allocation for all problems' host copies (a hipHostMalloc costs milliseconds: three per problem were
  // 1.5 s of a 32-quantizer training)
  struct Pinned {
    void *p = nullptr;
    ~Pinned() { if (p) (void)hipHostFree(p); }
  } pinned;
  {
    size_t words = 0;
    for (int p = 0; p < np; p++) words += 2 * (size_t)k * sdim[p] + 1;
    HIP_CHECK(hipHostMalloc(&pinned.p, sizeof(float) * std::max<size_t>(words, 1)));
    float *w = static_cast<float *>(pinned.p);
    for (int p = 0; p < np; p++) {
      P[p].h_prev = w; w += (size_t)k * sdim[p];
      P[p].h_next = w; w += (size_t)k * sdim[p];
      P[p].h_mism = reinterpret_cast<unsigned *>(w); w += 1;
    }
  }
  auto push_report = [&](int p, const gulon_kmeans_report &r) {
    if (reports && P[p].nrep < max_reports) reports[(size_t)p * max_reports + P[p].nrep] = r;
    P[p].nrep++;
  };
  auto make_job = [&](int p, const float *dC, int *d_assign) {
    Prob &pr = P[p];
    AssignJob &j = pr.job;
    j = AssignJob();
    j.ws = &pr.ws; j.dX = dX; j.n = n; j.ld = ld; j.from = from[p]; j.s = sdim[p]; j.dC = dC; j.k = k;
    j.rng_batch = 25000; j.d_assign = d_assign; j.st = pr.st; j.ps = pr.use_mfma ? &pr.packed : nullptr;
  };
  // run stages 2 and 3 of every listed problem, one synchronisation round per stage.  (Measured and dropped: a few
  // host threads issuing the ~300 short launches of these stages side by side -- 4.3 ms against 4.1 ms with one: the
  // stage is not bound by the host's launch rate.)
  auto finish_assigns = [&](const std::vector<int> &act) {
    for (int stage = 2; stage <= 3; stage++) {
      bool pending = false;
      for (int p : act) pending |= !P[p].job.done;
      if (!pending) break;
      for (int p : act) if (!P[p].job.done) HIP_CHECK(hipStreamSynchronize(P[p].st));
      for (int p : act) { if (stage == 2) assign_stage2(P[p].job); else assign_stage3(P[p].job); }
    }
  };

  DevBuf<UpdDesc> d_descs(np);
  DevBuf<StreamDesc> d_sdescs(np), d_odescs(np);
  bool order_ready = false;           // stream_order has already run on the current a_prev of every active problem
  bool stream_update = true;     // every problem through kmeans_stream.hip (all or none: one batched launch pair)
  for (int p = 0; p < np; p++) stream_update = stream_update && stream_update_supported(n, k, sdim[p]);
  if (stream_update) {
    // the streamed update keeps a second, pair-major copy of every slice (+ 4 bytes of order per row): it is taken only
    // where that fits beside the slices and the assign's packed operands with a quarter of the device's memory to spare
    size_t need = 0, free_b = 0, total_b = 0;
    for (int p = 0; p < np; p++) need += (size_t)stream_padded_rows(n) * (8 * (size_t)((sdim[p] + 1) / 2) + 4);
    for (int p = 0; p < np; p++) need += (size_t)n * (4 * (size_t)sdim[p] + 160 + 16);   // (the buffers allocated below either way)
    HIP_CHECK(hipMemGetInfo(&free_b, &total_b));
    if (need + total_b / 4 > free_b) stream_update = false;
  }
  hipStream_t bst = nullptr;
  hipEvent_t upd_done = nullptr;
  HIP_CHECK(hipStreamCreateWithFlags(&bst, hipStreamNonBlocking));
  HIP_CHECK(hipEventCreateWithFlags(&upd_done, hipEventDisableTiming));
  struct Cleanup {
    hipStream_t &s; hipEvent_t &e;
    ~Cleanup() { if (e) (void)hipEventDestroy(e); if (s) (void)hipStreamDestroy(s); }
  } cleanup{bst, upd_done};
  std::vector<int> all(np);
  for (int p = 0; p < np; p++) {
    all[p] = p;
    const int s = sdim[p];
    Prob &pr = P[p];
    // GULON_KMEANS_SERIAL=1 (profiling): every problem on the batch stream, so that kernels run one at a time and
    // rocprofv3's per-kernel durations are not inflated by the other sub-quantizers' concurrent launches
    static const bool serial = getenv("GULON_KMEANS_SERIAL") != nullptr;
    if (serial) { pr.st = bst; pr.shared_stream = true; }
    else HIP_CHECK(hipStreamCreateWithFlags(&pr.st, hipStreamNonBlocking));
    HIP_CHECK(hipEventCreateWithFlags(&pr.assigned, hipEventDisableTiming));
    pr.c_prev.alloc((size_t)k * s); pr.c_next.alloc((size_t)k * s);
    pr.a_prev.alloc(n); pr.a_next.alloc(n);
    pr.mism.alloc(1);
    // KMeans.init (KMeans.scala:188-196)
    std::vector<int> rows(k);
    JRandom rng((int64_t)seeds[p]);
    for (int c = 0; c < k; c++) rows[c] = rng.next_int(n);
    pr.d_rows.alloc(k);
    HIP_CHECK(hipMemcpy(pr.d_rows.p, rows.data(), sizeof(int) * k, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(gather_centroids, dim3(ceil_div((long long)k * s, 256)), dim3(256), 0, pr.st, dX, ld, from[p], s,
                       pr.d_rows.p, k, pr.c_prev.p);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipMemsetAsync(pr.a_prev.p, 0, sizeof(int) * (size_t)n, pr.st));
    pr.use_mfma = mfma_assign_supported(s, k);
    pr.xs.alloc((size_t)n * s + 2);   // (+2: update_fused reads rows as pairs of floats)
    GULON_UNSUPPORTED((long long)n * s >= (1ll << 32), "slice of %lld elements: a dispatch carries fewer than 2^32 work-items",
                      (long long)n * s);
    hipLaunchKernelGGL(copy_slice, dim3(ceil_div((long long)n * s, 256)), dim3(256), 0, pr.st, dX, ld, from[p], s,
                       (long long)n * s, pr.xs.p);
    // the matrix-core operands from the COMPACT copy: the strided slices of the row-major data cost a partial line per
    // row and pass (pack_slice_split 1.0 ms per sub-quantizer at BASELINE config 3 from the rows, 0.3 from the copy)
    if (pr.use_mfma) pack_slice(pr.xs.p, n, s, 0, s, k, pr.packed, pr.st);
    if (stream_update) {
      const size_t ns = (size_t)stream_padded_rows(n);
      pr.xp.alloc(2 * ns * (size_t)((s + 1) / 2));
      HIP_CHECK(hipMemsetAsync(pr.xp.p, 0, sizeof(float) * 2 * ns * (size_t)((s + 1) / 2), pr.st));   // the padding rows
      pr.s_wild.alloc(1);
      HIP_CHECK(hipMemsetAsync(pr.s_wild.p, 0, sizeof(int), pr.st));
      stream_pack_pairs(pr.xs.p, n, s, pr.xp.p, pr.s_wild.p, pr.st);
      size_t coff_words = 0;
      pr.s_ord.alloc(stream_order_words(n, k, &coff_words));
      pr.s_coff.alloc(coff_words);
    }
    make_job(p, pr.c_prev.p, pr.a_prev.p);
    assign_stage1(pr.job);
    push_report(p, gulon_kmeans_report{0, 0, 0, 0.f, 0.f});
  }
  lap("alloc+pack+stage1");
  finish_assigns(all);
  for (int p = 0; p < np; p++) {
    P[p].c_prev.download(P[p].h_prev, (size_t)k * sdim[p], P[p].st);
    HIP_CHECK(hipStreamSynchronize(P[p].st));
  }
  lap("first assign done");
  if (print) {
    unsigned long long dr = 0;
    for (int p = 0; p < np; p++) dr += P[p].ws.last_draws;
    fprintf(stderr, "[gulon trace]   first assign: tie draws %llu\n", dr);
  }

  for (int i = 0; i <= max_iterations;) {
    std::vector<int> act;
    for (int p = 0; p < np; p++) if (!P[p].done) act.push_back(p);
    if (act.empty()) break;
    // one batched update for all active problems, then every problem's assign on its own stream.  (Measured and
    // dropped: the update in staggered groups of problems -- a group's chains under the next group's regrouping, its
    // assign under the next chains, so that the first assign starts after a quarter of the update: 12 iterations at
    // BASELINE config 3 take 0.66 s with one group, 0.71 / 0.79 / 0.93 s with 2 / 4 / 8 -- the memory-bound regrouping
    // and the assign slow each other down by more than the overlap saves.)
    if (stream_update) {
      std::vector<StreamDesc> sd;
      for (int p : act) {
        StreamDesc D{};
        D.assign = P[p].a_prev.p; D.xp = P[p].xp.p; D.ord = P[p].s_ord.p; D.coff = P[p].s_coff.p; D.cout = P[p].c_next.p; D.wild = P[p].s_wild.p;
        D.s = sdim[p]; D.ns = stream_padded_rows(n);
        sd.push_back(D);
      }
      // (sd is read by hipMemcpyAsync from pageable memory: staged before the calls return)
      if (!order_ready) kmeans_stream_order(sd, d_odescs.p, n, k, bst);
      kmeans_stream_chains(sd, d_sdescs.p, n, k, bst);
      order_ready = false;
    } else {
      std::vector<UpdDesc> descs;
      for (int p : act)
        descs.push_back(make_upd_desc(P[p].ws, P[p].xs.p, sdim[p], n, k, 0, sdim[p], P[p].a_prev.p, P[p].c_next.p));
      kmeans_update_batch(descs, d_descs.p, n, k, bst);
    }
    HIP_CHECK(hipEventRecord(upd_done, bst));
    lap("  update batch", &tt.update_ms);
    for (int p : act) {
      Prob &pr = P[p];
      HIP_CHECK(hipStreamWaitEvent(pr.st, upd_done, 0));
      HIP_CHECK(hipMemsetAsync(pr.a_next.p, 0, sizeof(int) * (size_t)n, pr.st));   // fresh Array[Int] per parAssign
      make_job(p, pr.c_next.p, pr.a_next.p);
      assign_stage1(pr.job);
    }
    lap("  assign stage1", &tt.assign_ms);
    finish_assigns(act);
    lap("  assign stages 2-3", &tt.recheck_ms);
    if (trace) {
      unsigned long long fl = 0, dr = 0;
      double fm = 0, ub = 0;
      for (int p : act) {
        fl += P[p].ws.last_flagged; dr += P[p].ws.last_draws;
        if (P[p].use_mfma) fm += 2.0 * n * k * sdim[p];
        ub += 4.0 * n * sdim[p];
      }
      if (print)
        fprintf(stderr, "[gulon trace]   rows re-checked exactly %llu of %llu (%.3g), tie draws %llu\n", fl,
                (unsigned long long)n * act.size(), (double)fl / ((double)n * act.size()), dr);
      if (tt.on) {
        tt.iterations++; tt.rows_rechecked += fl; tt.rows_total += (unsigned long long)n * act.size();
        tt.mfma_flops += fm; tt.update_bytes += ub;
      }
    }
    static const bool speculate = [] { const char *e = getenv("GULON_UPDATE_SPECULATE"); return !(e && atoi(e) == 0); }();
    if (stream_update && speculate && i < max_iterations) {
      // The next update's chunk order depends on nothing but this assignment: it runs now, on the batch stream, under
      // the convergence test's downloads and host round trip (~1 ms in which the GPU was idle).  Wasted only in the
      // iteration that finds every problem converged.
      std::vector<StreamDesc> sd;
      for (int p : act) {
        Prob &pr = P[p];
        HIP_CHECK(hipEventRecord(pr.assigned, pr.st));
        HIP_CHECK(hipStreamWaitEvent(bst, pr.assigned, 0));
        StreamDesc D{};
        D.assign = pr.a_next.p; D.xp = pr.xp.p; D.ord = pr.s_ord.p; D.coff = pr.s_coff.p; D.cout = pr.c_next.p; D.wild = pr.s_wild.p;
        D.s = sdim[p]; D.ns = stream_padded_rows(n);
        sd.push_back(D);
      }
      kmeans_stream_order(sd, d_odescs.p, n, k, bst);
      order_ready = true;
    }
    for (int p : act) {
      Prob &pr = P[p];
      HIP_CHECK(hipMemsetAsync(pr.mism.p, 0, sizeof(unsigned), pr.st));
      hipLaunchKernelGGL(count_mismatch, dim3(ceil_div(n, 256)), dim3(256), 0, pr.st, pr.a_prev.p, pr.a_next.p, n,
                         pr.mism.p);
      pr.c_next.download(pr.h_next, (size_t)k * sdim[p], pr.st);
      pr.mism.download(pr.h_mism, 1, pr.st);
    }
    for (int p : act) HIP_CHECK(hipStreamSynchronize(P[p].st));
    bool all_conv = true;
    for (int p : act) {
      Prob &pr = P[p];
      bool converged = *pr.h_mism == 0;                      // Arrays.equals(prev, next)
      gulon_kmeans_report r{i, converged ? 1 : 0, 0, 0.f, 0.f};
      step_stats(pr.h_prev, pr.h_next, k, sdim[p], &r);
      push_report(p, r);
      std::swap(pr.c_prev, pr.c_next);
      std::swap(pr.a_prev, pr.a_next);
      std::swap(pr.h_prev, pr.h_next);
      if (converged) pr.done = true; else all_conv = false;
    }
    lap("  mismatch+reports", &tt.converge_ms);
    if (all_conv) break;
    i++;
  }
  for (int p = 0; p < np; p++) {
    memcpy(c_out[p], P[p].h_prev, sizeof(float) * (size_t)k * sdim[p]);
    if (n_reports) n_reports[p] = P[p].nrep;
  }
}

}  // namespace gulon

using namespace gulon;

#ifdef GULON_TEST_HOOKS
GULON_API int32_t gulon_selftest_mean_division(int32_t n_max, int32_t numerators_per_divisor, uint64_t seed,
                                               int64_t *mismatches) {
  return guarded([&] {
    GULON_REQUIRE(mismatches != nullptr && n_max >= 1 && n_max < (1 << 24) && numerators_per_divisor >= 1, "bad arguments");
    DevBuf<unsigned long long> bad(1);
    HIP_CHECK(hipMemset(bad.p, 0, sizeof(unsigned long long)));
    hipLaunchKernelGGL(mean_division_selftest, dim3(8192), dim3(256), 0, 0, n_max, numerators_per_divisor,
                       (unsigned long long)seed, bad.p);
    HIP_CHECK(hipGetLastError());
    unsigned long long h = 0;
    HIP_CHECK(hipMemcpy(&h, bad.p, sizeof(h), hipMemcpyDeviceToHost));
    *mismatches = (int64_t)h;
  });
}
#endif

GULON_API int32_t gulon_kmeans_trace(int32_t enable) {
  return guarded([&] {
    TrainTrace &t = train_trace();
    t = TrainTrace();
    t.on = enable != 0;
  });
}

GULON_API int32_t gulon_kmeans_trace_read(gulon_kmeans_trace_totals *out) {
  return guarded([&] {
    GULON_REQUIRE(out != nullptr, "out is null");
    const TrainTrace &t = train_trace();
    out->iterations = t.iterations;
    out->update_ms = t.update_ms; out->assign_ms = t.assign_ms; out->recheck_ms = t.recheck_ms;
    out->converge_ms = t.converge_ms;
    out->mfma_flops = t.mfma_flops; out->update_bytes = t.update_bytes;
    out->rows_rechecked = (double)t.rows_rechecked; out->rows_total = (double)t.rows_total;
  });
}

static void check_slice(const gulon_dataset *ds, int from, int s, int k) {
  GULON_REQUIRE(ds != nullptr, "dataset is null");
  GULON_REQUIRE(from >= 0 && s >= 0 && from + s <= ds->d, "column slice [%d,%d) outside [0,%d)", from, from + s,
                ds ? ds->d : 0);
  GULON_REQUIRE(k >= 1, "numClusters must be >= 1 (got %d)", k);
}

GULON_API int32_t gulon_kmeans_init(const gulon_dataset *ds, int32_t from, int32_t s, int32_t k, int32_t seed,
                                    float *c_out, int32_t *rows_out) {
  return guarded([&] {
    check_slice(ds, from, s, k);
    GULON_REQUIRE(ds->n >= 1, "KMeans.init needs at least one row");
    std::vector<int> rows(k);
    JRandom rng((int64_t)seed);
    for (int c = 0; c < k; c++) rows[c] = rng.next_int(ds->n);
    if (rows_out) memcpy(rows_out, rows.data(), sizeof(int) * k);
    if (c_out && s > 0) {
      DevBuf<int> dr; dr.upload(rows.data(), k);
      DevBuf<float> dc((size_t)k * s);
      hipLaunchKernelGGL(gather_centroids, dim3(ceil_div((long long)k * s, 256)), dim3(256), 0, 0, ds->x.p, ds->d,
                         from, s, dr.p, k, dc.p);
      HIP_CHECK(hipGetLastError());
      dc.download(c_out, (size_t)k * s);
      HIP_CHECK(hipDeviceSynchronize());
    }
  });
}

GULON_API int32_t gulon_kmeans_assign(const gulon_dataset *ds, int32_t from, int32_t s, const float *centroids,
                                      int32_t k, int32_t rng_batch, int32_t *assignments) {
  return guarded([&] {
    check_slice(ds, from, s, k);
    if (ds->n == 0) return;
    KmeansWorkspace ws;
    DevBuf<float> dc; dc.upload(centroids, std::max<size_t>((size_t)k * s, 1));
    DevBuf<int> da; da.upload(assignments, ds->n);
    PackedSlice packed;
    const bool mf = mfma_assign_supported(s, k);
    if (mf) pack_slice(ds->x.p, ds->n, ds->d, from, s, k, packed, nullptr);
    kmeans_assign_dev(ws, ds->x.p, ds->n, ds->d, from, s, dc.p, k, rng_batch, da.p, nullptr, mf ? &packed : nullptr);
    da.download(assignments, ds->n);
    HIP_CHECK(hipDeviceSynchronize());
  });
}

GULON_API int32_t gulon_kmeans_update(const gulon_dataset *ds, int32_t from, int32_t s, int32_t k,
                                      const int32_t *assignments, float *c_out) {
  return guarded([&] {
    check_slice(ds, from, s, k);
    if (s == 0) return;
    validate_assignments(assignments, ds->n, k);
    KmeansWorkspace ws;
    DevBuf<int> da; da.upload(assignments, std::max(ds->n, 1));
    DevBuf<float> dc((size_t)k * s);
    kmeans_update_dev(ws, ds->x.p, ds->n, ds->d, from, s, k, da.p, dc.p, nullptr);
    dc.download(c_out, (size_t)k * s);
    HIP_CHECK(hipDeviceSynchronize());
  });
}

GULON_API int32_t gulon_kmeans_iterate(const gulon_dataset *ds, int32_t from, int32_t s, const float *c_in, int32_t k,
                                       int32_t iters, float *c_out) {
  return guarded([&] {
    check_slice(ds, from, s, k);
    GULON_REQUIRE(iters >= 0, "iters must be >= 0");
    if (s == 0) return;
    KmeansWorkspace ws;
    DevBuf<float> dc; dc.upload(c_in, (size_t)k * s);
    DevBuf<int> da(std::max(ds->n, 1));
    HIP_CHECK(hipMemset(da.p, 0, sizeof(int) * (size_t)std::max(ds->n, 1)));   // one array reused (KMeans.scala:101)
    PackedSlice packed;
    const bool mf = iters > 0 && mfma_assign_supported(s, k);
    if (mf) pack_slice(ds->x.p, ds->n, ds->d, from, s, k, packed, nullptr);
    for (int it = 0; it < iters; it++) {
      kmeans_assign_dev(ws, ds->x.p, ds->n, ds->d, from, s, dc.p, k, 0, da.p, nullptr, mf ? &packed : nullptr);
      kmeans_update_dev(ws, ds->x.p, ds->n, ds->d, from, s, k, da.p, dc.p, nullptr);
    }
    dc.download(c_out, (size_t)k * s);
    HIP_CHECK(hipDeviceSynchronize());
  });
}

GULON_API int32_t gulon_kmeans_train(const gulon_dataset *ds, int32_t from, int32_t s, int32_t k,
                                     int32_t max_iterations, int32_t seed, float *c_out,
                                     gulon_kmeans_report *reports, int32_t max_reports, int32_t *n_reports) {
  return guarded([&] {
    check_slice(ds, from, s, k);
    GULON_REQUIRE(s >= 1, "dimension must be >= 1");
    float *outs[1] = {c_out};
    kmeans_train_batch(ds->x.p, ds->n, ds->d, 1, &from, &s, &seed, k, max_iterations, outs, reports, max_reports,
                       n_reports);
  });
}

// quantizers [j_begin, j_end) of an m-quantizer ProductQuantizer (all of them: 0, m)
static void pq_train_range(const gulon_dataset *ds, int m, int k, int max_iterations, int j_begin, int j_end,
                           float *cents_out, gulon_kmeans_report *reports, int max_reports, int32_t *n_reports) {
  GULON_REQUIRE(ds != nullptr, "dataset is null");
  GULON_REQUIRE(m >= 1 && m <= ds->d && k >= 1, "bad quantizer shape m=%d k=%d d=%d", m, k, ds->d);
  GULON_REQUIRE(0 <= j_begin && j_begin <= j_end && j_end <= m, "bad quantizer range [%d,%d) of %d", j_begin, j_end, m);
  const int np = j_end - j_begin;
  if (np == 0) return;
  std::vector<int> from, until, f(np), sdim(np), seeds(np);
  subvectors(ds->d, m, from, until);
  std::vector<float *> outs(np);
  for (int p = 0; p < np; p++) {
    const int j = j_begin + p;
    f[p] = from[j];
    sdim[p] = until[j] - from[j];
    seeds[p] = j;                                     // ProductQuantizer.scala:139
    outs[p] = cents_out + (size_t)k * from[j];
  }
  kmeans_train_batch(ds->x.p, ds->n, ds->d, np, f.data(), sdim.data(), seeds.data(), k, max_iterations, outs.data(),
                     reports, max_reports, n_reports);
}

GULON_API int32_t gulon_pq_train(const gulon_dataset *ds, int32_t m, int32_t k, int32_t max_iterations,
                                 float *cents_out, gulon_kmeans_report *reports, int32_t max_reports,
                                 int32_t *n_reports) {
  return guarded([&] { pq_train_range(ds, m, k, max_iterations, 0, m, cents_out, reports, max_reports, n_reports); });
}

GULON_API int32_t gulon_pq_train_range(const gulon_dataset *ds, int32_t m, int32_t k, int32_t max_iterations,
                                       int32_t j_begin, int32_t j_end, float *cents_out,
                                       gulon_kmeans_report *reports, int32_t max_reports, int32_t *n_reports) {
  return guarded(
      [&] { pq_train_range(ds, m, k, max_iterations, j_begin, j_end, cents_out, reports, max_reports, n_reports); });
}

static void pq_encode_range(const gulon_dataset *ds, int m, int k, const float *cents, int j_begin, int j_end,
                            uint8_t *codes_out);

GULON_API int32_t gulon_pq_encode(const gulon_dataset *ds, int32_t m, int32_t k, const float *cents,
                                  uint8_t *codes_out) {
  return guarded([&] { pq_encode_range(ds, m, k, cents, 0, m, codes_out); });
}

// codes_out: (j_end - j_begin) packed code arrays back to back (quantizer j_begin first)
GULON_API int32_t gulon_pq_encode_range(const gulon_dataset *ds, int32_t m, int32_t k, const float *cents,
                                        int32_t j_begin, int32_t j_end, uint8_t *codes_out) {
  return guarded([&] { pq_encode_range(ds, m, k, cents, j_begin, j_end, codes_out); });
}

static void pq_encode_range(const gulon_dataset *ds, int m, int k, const float *cents, int j_begin, int j_end,
                            uint8_t *codes_out) {
  {
    GULON_REQUIRE(ds != nullptr, "dataset is null");
    GULON_REQUIRE(0 <= j_begin && j_begin <= j_end && j_end <= m, "bad quantizer range [%d,%d) of %d", j_begin, j_end, m);
    GULON_REQUIRE(m >= 1 && m <= ds->d && k >= 1, "bad quantizer shape m=%d k=%d d=%d", m, k, ds->d);
    int width = -1;
    GULON_REQUIRE(gulon_coder_width(k, &width) == GULON_OK, "too many clusters: %d", k);
    const int n = ds->n;
    int bytes = 0;
    gulon_coder_bytes(width, n, &bytes);
    if (n == 0 || width == 0) return;   // Coder0: empty codes
    std::vector<int> from, until;
    subvectors(ds->d, m, from, until);
    KmeansWorkspace ws;
    DevBuf<float> dc;
    DevBuf<int> da(n);
    DevBuf<uint8_t> d8(n);
    std::vector<int> h_idx;
    PackedSlice packed;
    for (int j = j_begin; j < j_end; j++) {
      const int s = until[j] - from[j];
      dc.upload(cents + (size_t)k * from[j], (size_t)k * s);
      HIP_CHECK(hipMemset(da.p, 0, sizeof(int) * (size_t)n));
      const bool mf = mfma_assign_supported(s, k);
      if (mf) pack_slice(ds->x.p, n, ds->d, from[j], s, k, packed, nullptr);
      kmeans_assign_dev(ws, ds->x.p, n, ds->d, from[j], s, dc.p, k, 0, da.p, nullptr, mf ? &packed : nullptr);   // serial assign
      uint8_t *out = codes_out + (size_t)(j - j_begin) * bytes;
      if (width == 8) {                                                                   // Coder8: idx.toByte
        hipLaunchKernelGGL(narrow_assign_u8, dim3(ceil_div(n, 256)), dim3(256), 0, 0, da.p, (long long)n, d8.p);
        HIP_CHECK(hipGetLastError());
        d8.download(out, n);
        HIP_CHECK(hipDeviceSynchronize());
      } else {
        h_idx.resize(n);
        da.download(h_idx.data(), n);
        HIP_CHECK(hipDeviceSynchronize());
        GULON_REQUIRE(gulon_coder_build(width, h_idx.data(), n, out) == GULON_OK, "coder failed");
      }
    }
  }
}
