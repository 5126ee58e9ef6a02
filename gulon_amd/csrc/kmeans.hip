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
template <int SMAX>
__global__ __launch_bounds__(256) void assign_exact(const float *__restrict__ X, int n, int ld, int from, int s,
                                                    const float *__restrict__ Cpad, const float *__restrict__ off,
                                                    int k, const int *__restrict__ rows, int nrows,
                                                    int *__restrict__ assign, unsigned *__restrict__ ties,
                                                    unsigned long long *__restrict__ tie_total) {
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
  for (int c = 0; c < k; c++) {
    const float *cc = Cpad + (size_t)c * SMAX;
    float d = 0.f;
#pragma unroll
    for (int j = 0; j < SMAX; j++) d += x[j] * cc[j];
    d = off[c] - 2 * d;
    if (d < mn) { mn = d; best = c; }
    else if (d == mn) nt++;
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

__global__ __launch_bounds__(64) void assign_resolve_wave(const float *__restrict__ X, int ld, int from, int s,
                                                          const float *__restrict__ Cpad, int smax,
                                                          const float *__restrict__ off, int k,
                                                          const unsigned *__restrict__ local,
                                                          const unsigned long long *__restrict__ block_off,
                                                          int seg_len, int bps, const int *__restrict__ list,
                                                          const unsigned *__restrict__ count,
                                                          int *__restrict__ assign) {
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
    rng.skip(block_off[(size_t)seg * bps + blk] + local[i]);
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
constexpr int SORT_ROWS_PER_WAVE = 2048;

// All update kernels take an array of per-problem descriptors and pick theirs with
// blockIdx.y (z for the group scan), so the m sub-quantizers of a ProductQuantizer are
// updated by ONE launch each: the sequential chains are latency-bound (k*s threads per
// problem), and only running all problems' chains side by side fills the GPU.

// per wave-chunk histogram: hist[chunk][k]
__global__ __launch_bounds__(256) void sort_hist(const UpdDesc *__restrict__ descs, int n, int k) {
  const UpdDesc D = descs[blockIdx.y];
  extern __shared__ unsigned sh[];  // 4 * k
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned *h = sh + wave * k;
  for (int c = lane; c < k; c += 64) h[c] = 0;
  long long chunk = (long long)blockIdx.x * 4 + wave;
  long long r0 = chunk * SORT_ROWS_PER_WAVE;
  long long r1 = r0 + SORT_ROWS_PER_WAVE < n ? r0 + SORT_ROWS_PER_WAVE : n;
  for (long long r = r0 + lane; r < r1; r += 64) atomicAdd(&h[D.assign[r]], 1u);
  if (r0 < n)
    for (int c = lane; c < k; c += 64) D.hist[(size_t)chunk * k + c] = h[c];
}

// two-level exclusive scan of the per-chunk histograms (per cluster, over chunks):
// level 1: one thread per (group of SCAN_GROUP chunks, cluster) scans its chunks in place
constexpr int SCAN_GROUP = 64;
__global__ void sort_scan_groups(const UpdDesc *__restrict__ descs, long long nchunks, int k) {
  const UpdDesc D = descs[blockIdx.z];
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
  const UpdDesc D = descs[blockIdx.y];
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
}

// each wave places its chunk's rows in order: order[start[c] + rank] = row (stable).
// The lanes holding the same cluster are found without a loop: one ballot per key bit, and the
// AND of (bit set ? ballot : ~ballot) over the bits is the mask of lanes with an equal key; the
// rank inside the cluster is the population count of that mask below the lane.
__global__ __launch_bounds__(256) void sort_place(const UpdDesc *__restrict__ descs, int n, int k, int key_bits) {
  const UpdDesc D = descs[blockIdx.y];
  extern __shared__ unsigned sh[];  // 4 * k running positions
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned *run = sh + wave * k;
  long long chunk = (long long)blockIdx.x * 4 + wave;
  long long r0 = chunk * SORT_ROWS_PER_WAVE;
  if (r0 >= n) return;
  long long r1 = r0 + SORT_ROWS_PER_WAVE < n ? r0 + SORT_ROWS_PER_WAVE : n;
  const long long grp = chunk / SCAN_GROUP;
  for (int c = lane; c < k; c += 64)
    run[c] = D.start[c] + D.gtot[(size_t)grp * k + c] + D.hist[(size_t)chunk * k + c];
  const unsigned long long lt = (1ull << lane) - 1ull;
  int key_next = r0 + lane < r1 ? D.assign[r0 + lane] : 0;
  for (long long base = r0; base < r1; base += 64) {
    const long long r = base + lane;
    const bool valid = r < r1;
    const int key = key_next;
    if (base + 64 + lane < r1) key_next = D.assign[base + 64 + lane];
    unsigned long long same = __ballot(valid);
    for (int bit = 0; bit < key_bits; bit++) {
      const unsigned long long bm = __ballot((key >> bit) & 1);
      same &= ((key >> bit) & 1) ? bm : ~bm;
    }
    // same-wave LDS accesses execute in program order: every lane of a cluster reads the
    // running position before the cluster's first lane advances it
    const unsigned b = valid ? run[key] : 0u;
    if (valid) D.order[b + __popcll(same & lt)] = (int)r;
    if (valid && (same & lt) == 0ull) run[key] = b + __popcll(same);
  }
}

// one thread per (cluster, dim): c_j <- c_j + (x_j - c_j)/n over the cluster's rows
// in row order (KMeans.scala:211-224).  IEEE division (__fdiv_rn), int->float RNE.
// Two-level software pipeline: row ids two batches ahead, row values one batch ahead of the
// sequential divide-add recurrence (neither load depends on the chain).
__global__ __launch_bounds__(64) void update_chains(const UpdDesc *__restrict__ descs, int k) {
  const UpdDesc D = descs[blockIdx.y];
  const int s = D.s;
  const float *__restrict__ X = D.x;
  const int ld = D.ld;
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= k * s) return;
  int c = t / s, j = t - c * s;
  const unsigned len = D.count[c];
  const int *ord = D.order + D.start[c];
  const float *col = X + D.from + j;
  float p = 0.f;
  if (len == 0) { D.cout[t] = 0.f; return; }
  constexpr int U = 16;
  const unsigned nb = len / U;          // full batches
  const unsigned last = len - 1;
  // every prefetch index is clamped to the last row of the cluster, so the loads are
  // unconditional: a select-or-load per element makes hipcc branch around each load and
  // wait for it individually (one exposed memory latency per step)
  int o1[U];
  float xa[U], xb[U];
#pragma unroll
  for (int u = 0; u < U; u++) xa[u] = col[(size_t)ord[min((unsigned)u, last)] * ld];
#pragma unroll
  for (int u = 0; u < U; u++) o1[u] = ord[min((unsigned)(U + u), last)];
  unsigned i = 0;
  for (unsigned b = 0; b < nb; b++) {
#pragma unroll
    for (int u = 0; u < U; u++) xb[u] = col[(size_t)o1[u] * ld];
#pragma unroll
    for (int u = 0; u < U; u++) o1[u] = ord[min(i + 2 * U + u, last)];
#pragma unroll
    for (int u = 0; u < U; u++) p = p + __fdiv_rn(xa[u] - p, (float)(int)(i + u + 1));
#pragma unroll
    for (int u = 0; u < U; u++) xa[u] = xb[u];
    i += U;
  }
  for (; i < len; i++) {
    float xv = col[(size_t)ord[i] * ld];
    p = p + __fdiv_rn(xv - p, (float)(int)(i + 1));
  }
  D.cout[t] = p;
}

// ---------------------------------------------------------------------------
// host drivers
// ---------------------------------------------------------------------------
static int pick_smax(int s) {
  if (s <= 4) return 4;
  if (s <= 8) return 8;
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
  long long nchunks = ceil_div(std::max(n, 1), SORT_ROWS_PER_WAVE);
  long long nchunks_pad = ((nchunks + 3) / 4) * 4;
  hist.ensure((size_t)nchunks_pad * k);
  gtot.ensure((size_t)ceil_div(nchunks_pad, SCAN_GROUP) * k);
  count.ensure(k);
  start.ensure(k);
  order.ensure((size_t)std::max(n, 1));
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
#define AE(S)                                                                                                      \
  hipLaunchKernelGGL(assign_exact<S>, dim3(grid), dim3(256), 0, j.st, j.dX, j.n, j.ld, j.from, j.s, ws.cpad.p,     \
                     ws.off.p, j.k, j.rows, j.nrows, j.d_assign, ws.ties.p, ws.tie_total.p)
  switch (smax) {
    case 4: AE(4); break;
    case 8: AE(8); break;
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
static void launch_tie_replay(AssignJob &j) {
  KmeansWorkspace &ws = *j.ws;
  const int n = j.n, smax = pick_smax(j.s);
  const int seg_len = j.rng_batch > 0 ? j.rng_batch : n;
  const int nseg = ceil_div(n, seg_len);
  const int bps = ceil_div(seg_len < n ? seg_len : n, 1024);
  ws.block_tot.ensure((size_t)nseg * bps);
  ws.block_off.ensure((size_t)nseg * bps);
  hipLaunchKernelGGL(tie_block_sums, dim3(bps, nseg), dim3(1024), 0, j.st, ws.ties.p, n, seg_len, bps, ws.local.p,
                     ws.block_tot.p);
  hipLaunchKernelGGL(tie_block_scan, dim3(nseg), dim3(1024), 0, j.st, ws.block_tot.p, bps, ws.block_off.p);
  const int *rows = j.filtered ? j.rows : nullptr;
  const int nrows = j.filtered ? j.nrows : n;
  if ((long long)j.k * j.s >= 16384) {   // long per-row replays: one wave per drawing row
    ws.tie_rows.ensure((size_t)std::max(nrows, 1));
    ws.tie_count.ensure(1);
    HIP_CHECK(hipMemsetAsync(ws.tie_count.p, 0, sizeof(unsigned), j.st));
    hipLaunchKernelGGL(collect_tie_rows, dim3(ceil_div(nrows, 256)), dim3(256), 0, j.st, ws.ties.p, rows, nrows,
                       ws.tie_rows.p, ws.tie_count.p);
    hipLaunchKernelGGL(assign_resolve_wave, dim3(1024), dim3(64), sizeof(float) * (size_t)j.s, j.st, j.dX, j.ld, j.from,
                       j.s, ws.cpad.p, smax, ws.off.p, j.k, ws.local.p, ws.block_off.p, seg_len, bps, ws.tie_rows.p,
                       ws.tie_count.p, j.d_assign);
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
  if (ws.host->total) {
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

// KMeans.fromAssignment for a batch of problems over the same data -> each D.cout (k x s).
// `d_descs` must hold descs.size() entries of device memory.
void kmeans_update_batch(const std::vector<UpdDesc> &descs, UpdDesc *d_descs, int n, int k, hipStream_t st) {
  const int np = (int)descs.size();
  if (np == 0) return;
  if (n <= 0) {
    for (const UpdDesc &D : descs) HIP_CHECK(hipMemsetAsync(D.cout, 0, sizeof(float) * (size_t)k * D.s, st));
    return;
  }
  HIP_CHECK(hipMemcpyAsync(d_descs, descs.data(), sizeof(UpdDesc) * np, hipMemcpyHostToDevice, st));
  int smax = 1;
  for (const UpdDesc &D : descs) smax = std::max(smax, D.s);
  long long nchunks = ceil_div(n, SORT_ROWS_PER_WAVE);
  int blocks = ceil_div(nchunks, 4);
  long long ngroups = ceil_div(nchunks, SCAN_GROUP);
  size_t shm = sizeof(unsigned) * 4 * (size_t)k;
  GULON_UNSUPPORTED(shm > 160 * 1024, "k-means update with k = %d clusters needs %zu B of LDS for its per-cluster counters "
                    "(> 160 KiB): train at most 10240 clusters per quantizer on the GPU", k, shm);
  hipLaunchKernelGGL(sort_hist, dim3(blocks, np), dim3(256), shm, st, d_descs, n, k);
  hipLaunchKernelGGL(sort_scan_groups, dim3((unsigned)ngroups, ceil_div(k, 256), np), dim3(256), 0, st, d_descs,
                     nchunks, k);
  hipLaunchKernelGGL(sort_scan_top, dim3(1, np), dim3(256), sizeof(unsigned) * (size_t)k, st, d_descs, ngroups, k);
  int key_bits = 0;
  while ((1 << key_bits) < k) key_bits++;
  hipLaunchKernelGGL(sort_place, dim3(blocks, np), dim3(256), shm, st, d_descs, n, k, key_bits);
  hipLaunchKernelGGL(update_chains, dim3(ceil_div((long long)k * smax, 64), np), dim3(64), 0, st, d_descs, k);
  HIP_CHECK(hipGetLastError());
}

// x/ld/from describe where row r's slice starts: x + r*ld + from
UpdDesc make_upd_desc(KmeansWorkspace &ws, const float *x, int ld, int n, int k, int from, int s, const int *d_assign,
                      float *dC) {
  ws.ensure(n, k, s);
  UpdDesc D;
  D.x = x; D.ld = ld;
  D.assign = d_assign; D.hist = ws.hist.p; D.gtot = ws.gtot.p; D.count = ws.count.p; D.start = ws.start.p;
  D.order = ws.order.p; D.cout = dC; D.from = from; D.s = s;
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
    DevBuf<float> c_prev, c_next;
    DevBuf<int> a_prev, a_next, d_rows;
    DevBuf<unsigned> mism;
    std::vector<float> h_prev, h_next;
    PackedSlice packed;      // MFMA-ready copy of this problem's column slice
    DevBuf<float> xs;        // row-major n x s copy of the slice: compact target of the update's gathers
    AssignJob job;
    bool use_mfma = false;
    bool done = false;
    int nrep = 0;
    ~Prob() { if (st) (void)hipStreamDestroy(st); }
  };
  std::vector<Prob> P(np);
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
  // run stages 2 and 3 of every listed problem, one synchronisation round per stage
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
    HIP_CHECK(hipStreamCreateWithFlags(&pr.st, hipStreamNonBlocking));
    pr.c_prev.alloc((size_t)k * s); pr.c_next.alloc((size_t)k * s);
    pr.a_prev.alloc(n); pr.a_next.alloc(n);
    pr.mism.alloc(1);
    pr.h_prev.resize((size_t)k * s); pr.h_next.resize((size_t)k * s);
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
    if (pr.use_mfma) pack_slice(dX, n, ld, from[p], s, k, pr.packed, pr.st);
    pr.xs.alloc((size_t)n * s);
    GULON_UNSUPPORTED((long long)n * s >= (1ll << 32), "slice of %lld elements: a dispatch carries fewer than 2^32 work-items",
                      (long long)n * s);
    hipLaunchKernelGGL(copy_slice, dim3(ceil_div((long long)n * s, 256)), dim3(256), 0, pr.st, dX, ld, from[p], s,
                       (long long)n * s, pr.xs.p);
    make_job(p, pr.c_prev.p, pr.a_prev.p);
    assign_stage1(pr.job);
    push_report(p, gulon_kmeans_report{0, 0, 0, 0.f, 0.f});
  }
  lap("alloc+pack+stage1");
  finish_assigns(all);
  for (int p = 0; p < np; p++) {
    P[p].c_prev.download(P[p].h_prev.data(), (size_t)k * sdim[p], P[p].st);
    HIP_CHECK(hipStreamSynchronize(P[p].st));
  }
  lap("first assign done");

  for (int i = 0; i <= max_iterations;) {
    std::vector<int> act;
    for (int p = 0; p < np; p++) if (!P[p].done) act.push_back(p);
    if (act.empty()) break;
    // one batched update for all active problems, then every problem's assign on its own stream
    std::vector<UpdDesc> descs;
    for (int p : act)
      descs.push_back(make_upd_desc(P[p].ws, P[p].xs.p, sdim[p], n, k, 0, sdim[p], P[p].a_prev.p, P[p].c_next.p));
    kmeans_update_batch(descs, d_descs.p, n, k, bst);
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
    std::vector<unsigned> h_mism(np, 0);
    for (int p : act) {
      Prob &pr = P[p];
      HIP_CHECK(hipMemsetAsync(pr.mism.p, 0, sizeof(unsigned), pr.st));
      hipLaunchKernelGGL(count_mismatch, dim3(ceil_div(n, 256)), dim3(256), 0, pr.st, pr.a_prev.p, pr.a_next.p, n,
                         pr.mism.p);
      pr.c_next.download(pr.h_next.data(), (size_t)k * sdim[p], pr.st);
      pr.mism.download(&h_mism[p], 1, pr.st);
    }
    for (int p : act) HIP_CHECK(hipStreamSynchronize(P[p].st));
    bool all_conv = true;
    for (int p : act) {
      Prob &pr = P[p];
      bool converged = h_mism[p] == 0;                       // Arrays.equals(prev, next)
      gulon_kmeans_report r{i, converged ? 1 : 0, 0, 0.f, 0.f};
      step_stats(pr.h_prev.data(), pr.h_next.data(), k, sdim[p], &r);
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
    memcpy(c_out[p], P[p].h_prev.data(), sizeof(float) * (size_t)k * sdim[p]);
    if (n_reports) n_reports[p] = P[p].nrep;
  }
}

}  // namespace gulon

using namespace gulon;

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
