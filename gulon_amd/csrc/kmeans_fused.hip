// KMeans.fromAssignment (KMeans.scala:198-226) without the regrouped copy of the data -- an EXPERIMENT, off by
// default (GULON_UPDATE_FUSED=1 turns it on): bit-identical results (tests/test_gpu_kmeans.py runs the k-means
// parity cases through it in a child process), but 20.7 ms per update at BASELINE config 3 against 9.7 ms for the
// bucketed path of kmeans.hip.  Kept with its measurements because it is the design that could reach ~4 ms.
//
// The running mean c += (x - c) / n is order dependent, so every (cluster, dim) chain has to see its rows in row
// order.  kmeans.hip (sort_place + update_chains_pk) moves the row slices into cluster buckets first: 12 GB read,
// 12.8 GB written, 12.8 GB read again at BASELINE config 3 -- 9.7 ms, of which the chains' own arithmetic is
// ~1.5 ms.  Here the slices stay where they are:
//
//   sort_order    per 1024-row chunk, the LOCAL order of its rows by (cluster, row): order[chunk][1024] (uint16) and
//                 the chunk-local start of every cluster coff[chunk][k + 1] (int32) -- the counting sort of
//                 sort_place on row indices instead of slices (6 bytes per row instead of 80): 1.05 ms;
//   update_fused  a workgroup owns G consecutive clusters of one problem (lane = (cluster, pair of dims), packed
//                 fp32, as update_chains_pk) and walks the chunks: the rows of ITS clusters are one contiguous span
//                 of the chunk's order, gathered row by row straight into LDS (double buffered), every record
//                 followed by the step's divisor n and RN(1 / n) (a fourth wave writes them during the chains of the
//                 previous chunk: the table look-up of update_chains_pk becomes part of the ds_read).  The chains
//                 then run while any lane of the wave has rows left in the chunk, each lane only while its own
//                 cluster has (exec mask).
//
// All workgroups of a problem are placed on one XCD (blocks b and b + 8 share one; for speed only): they gather from
// the same lines at about the same time, so HBM sees the data once.
//
// What was measured (in-kernel s_memtime stamps, cycles per chunk of wave 0; 9766 chunks):
//   requests 1500-1650, chains 1700 (7.5 steps per chunk: ~230 cycles per step where the chain itself needs ~60),
//   first barrier 500-900, LDS stores 550, second barrier + ring rotation 250: ~4700 cycles = 2.1 us per chunk.
//   * vmcnt is in order and 6 bits wide: a load consumed soon after its request forces everything requested before
//     it to have arrived (any arithmetic on a loaded word counts -- an int conversion, a subtraction), and more than
//     63 loads in flight cannot be told apart.  Hence: every vector load of an iteration is consumed L - 1 iterations
//     later (register rings, slot = chunk mod L, loop unrolled by L), rows are gathered one per lane with 8-byte
//     loads (16 loads per iteration, not 28), and the span bounds travel the same way (as scalar loads they sat in
//     lgkmcnt, which every ds_read wait of the chain phase then had to drain).  None of it moved the total by more
//     than 10 %, and L = 2, 3, 4 measure the same: latency is not what binds.
//   * what binds is the texture-address unit: 128 scattered 40-byte rows per chunk and workgroup are 64 different
//     cache lines per wave-instruction (row per lane: 5 instructions per row) or ~7 (element per lane: 24
//     instructions), ~1000-2500 TA cycles per chunk and CU either way, in front of chains that wait behind the same
//     waves' requests and LDS stores.  The floor of this design is ~3 look-ups per row (16-byte loads) = 3.3 ms +
//     sort_order, and it needs the phases decoupled: chains in waves of their own that never issue a gather.
#include <climits>

#include "kmeans.hpp"

namespace gulon {

namespace {

constexpr int FCHUNK = 1024;          // rows per chunk (uint16 local indices)
constexpr int FUSED_THREADS = 256;    // waves 0-2: chains (<= 192 lanes), wave 3: divisors; all four gather
constexpr int FUSED_NR = 2;           // span rows per thread held in registers (more: a slow direct path)
#ifndef GULON_FUSED_DEPTH
#define GULON_FUSED_DEPTH 4
#endif
constexpr int FUSED_DEPTH = GULON_FUSED_DEPTH;   // register slots of slices in flight

// order / coff of one chunk (stages A-C of sort_place on row indices)
__global__ __launch_bounds__(256) void sort_order(const UpdDesc *__restrict__ descs, int n, int k, int key_bits,
                                                  int nchunks) {
  const UpdDescG D = load_desc(descs, blockIdx.y);
  extern __shared__ unsigned sh[];
  unsigned *wh = sh;                                  // [4][k] per-wave counts, then running positions
  __shared__ unsigned wave_tot[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long long chunk = blockIdx.x;
  const long long r0 = chunk * FCHUNK;
  const long long r1 = r0 + FCHUNK < n ? r0 + FCHUNK : n;
  const auto order = reinterpret_cast<gptr<unsigned short>>(D.xb) + (size_t)chunk * FCHUNK;
  const auto coff = reinterpret_cast<gptr<int>>(reinterpret_cast<gptr<unsigned short>>(D.xb) + (size_t)nchunks * FCHUNK) +
                    (size_t)chunk * (k + 1);   // (nchunks * FCHUNK uint16: a multiple of 4 bytes)
  const long long wr0 = r0 + wave * 256;
  for (int e = tid; e < 4 * k; e += 256) wh[e] = 0;
  __syncthreads();
  const unsigned long long lt = (1ull << lane) - 1ull;
  int keys[4];
  unsigned long long same[4];
#pragma unroll
  for (int t = 0; t < 4; t++) {
    const long long r = wr0 + t * 64 + lane;
    keys[t] = r < r1 ? D.assign[r] : 0;
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
  {   // chunk-local exclusive scan of the cluster totals: thread t owns clusters [t*per, t*per + per)
    const int per = (k + 255) / 256;
    unsigned mine = 0;
    for (int c = tid * per; c < min(k, tid * per + per); c++) mine += wh[c] + wh[k + c] + wh[2 * k + c] + wh[3 * k + c];
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
      coff[c] = (int)base;
      wh[c] = base; wh[k + c] = base + t0; wh[2 * k + c] = base + t0 + t1; wh[3 * k + c] = base + t0 + t1 + t2;
      base += t0 + t1 + t2 + t3;
    }
    if (tid == 0) coff[k] = (int)(r1 - r0);
  }
  __syncthreads();
#pragma unroll
  for (int t = 0; t < 4; t++) {
    const bool valid = wr0 + t * 64 + lane < r1;
    const int key = keys[t];
    const unsigned b = valid ? wh[wave * k + key] : 0u;
    const unsigned pos = b + (unsigned)__popcll(same[t] & lt);
    if (valid && (same[t] & lt) == 0ull) wh[wave * k + key] = b + (unsigned)__popcll(same[t]);
    if (valid) order[pos] = (unsigned short)(wave * 256 + t * 64 + lane);
  }
}

template <int SP /* record stride of the data part in floats (s rounded up to even) */>
__global__ __launch_bounds__(FUSED_THREADS) void update_fused(const UpdDesc *__restrict__ descs, int np, int n, int k,
                                                             int G, int groups, int nchunks) {
  static_assert(SP % 2 == 0 && SP >= 2 && SP <= 16, "even stride");
  constexpr int HP = SP / 2, REC = SP + 2;
  // all groups of a problem on one XCD: blocks b and b + 8 share one
  const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
  const int prob = (q / groups) * 8 + xcd, grp = q % groups;
  if (prob >= np) return;
  const UpdDescG D = load_desc(descs, prob);
  const int s = D.s;
  const int c_first = grp * G, Gm = min(G, k - c_first);
  if (Gm <= 0) return;
  const auto order = reinterpret_cast<gptr<const unsigned short>>(D.xb);
  const auto coff = reinterpret_cast<gptr<const int>>(order + (size_t)nchunks * FCHUNK);   // [nchunks][k + 1]
  extern __shared__ float lds[];
  float *const bufs = lds;                                         // [2][FCHUNK][REC]
  int *const meta = reinterpret_cast<int *>(lds + 2 * FCHUNK * REC);   // [2][G][2]: span-relative start, count
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // chain identity (waves 0-2)
  const int cl = tid / HP, pair = tid - cl * HP;
  const bool chain_lane = tid < Gm * HP && 2 * pair < s;
  // divisor writer identity (wave 3): lane = cluster of the group
  const int l3 = tid - 192;
  const bool writer = wave == 3 && l3 < Gm;
  unsigned before = 0;                                             // rows of the writer's cluster in earlier chunks

  // span of this group's clusters in chunk c's order.  Per-lane vector loads of one address on purpose: as scalar
  // loads they sit in lgkmcnt, which every ds_read wait of the chain phase then has to drain (SMEM returns out of
  // order: lgkmcnt(0)) -- a scalar-cache miss per iteration on the chains' critical path
  int vzero = 0;
  asm volatile("" : "+v"(vzero));
  auto span = [&](int c, int &base, int &end) {
    const int cc = min(max(c, 0), nchunks - 1);
    const auto p = coff + (size_t)cc * (k + 1) + c_first + vzero;   // (an opaque per-lane zero: not provably uniform)
    base = p[0];
    end = p[Gm];
  };
  auto rows_in = [&](int c, int base, int end) { return (c >= 0 && c < nchunks) ? end - base : 0; };
  auto cluster_span = [&](int c, int &lo, int &hi) {               // (writer lanes) its cluster in chunk c: raw bounds
    lo = 0; hi = 0;
    if (writer) {
      const auto p = coff + (size_t)min(max(c, 0), nchunks - 1) * (k + 1) + c_first + l3;
      lo = p[0];
      hi = p[1];
    }
  };
  // Gathers, one ROW per lane: thread t takes rows t and t + 256 of the span (NR = 2; more rows: the slow path of
  // store_rows), each as HP 8-byte loads (4-byte aligned: rows are s floats apart) -- a per-element mapping needs a
  // dozen loads per thread and chunk, and vmcnt (6 bits) cannot keep more than 63 loads of a wave apart: with
  // 28 loads per iteration the look-ahead collapsed to one iteration whatever the ring depth.  For an odd s the last
  // pair reads one float past the row (the caller's buffer has that slack); the padding column is rewritten below.
  typedef f32x2 __attribute__((aligned(4))) f32x2_u;
  constexpr int NR = FUSED_NR;
  auto load_order = [&](int c, int base, int nrows, int (&ord)[NR]) {
    const int cc = min(max(c, 0), nchunks - 1);
    const auto ob = order + (size_t)cc * FCHUNK;
    const unsigned last = (unsigned)(base + max(nrows - 1, 0));
#pragma unroll
    for (int r = 0; r < NR; r++) ord[r] = ob[min((unsigned)(base + tid + FUSED_THREADS * r), last)];   // clamped: always valid
  };
  auto gather = [&](int c, int nrows, const int (&ord)[NR], f32x2 (&g)[NR][HP]) {
    const int cc = min(max(c, 0), nchunks - 1);
    const auto xb = D.x + (size_t)cc * FCHUNK * s;
#pragma unroll
    for (int r = 0; r < NR; r++) {
      const auto row = xb + (unsigned)(ord[r] * s);
#pragma unroll
      for (int h = 0; h < HP; h++) {
        g[r][h] = f32x2{0.f, 0.f};
        if (tid + FUSED_THREADS * r < nrows && 2 * h < s) g[r][h] = *reinterpret_cast<gptr<const f32x2_u>>(row + 2 * h);
      }
    }
  };
  auto store_rows = [&](int c, int base, int nrows, const f32x2 (&g)[NR][HP], float *B) {
#pragma unroll
    for (int r = 0; r < NR; r++) {
      const int ri = tid + FUSED_THREADS * r;
      if (ri < nrows) {
#pragma unroll
        for (int h = 0; h < HP; h++) {
          f32x2 v = g[r][h];
          if (2 * h + 1 == s) v.y = v.x;                     // odd s: the padding column mirrors the last one
          if (2 * h < s) *reinterpret_cast<f32x2 *>(B + ri * REC + 2 * h) = v;
        }
      }
    }
    // (rare) more rows than the registers hold: straight from memory
    for (int ri = tid + FUSED_THREADS * NR; ri < nrows; ri += FUSED_THREADS) {
      const int lrow = order[(size_t)c * FCHUNK + base + ri];
      for (int j = 0; j < s; j++) {
        const float v = D.x[((size_t)c * FCHUNK + lrow) * s + j];
        B[ri * REC + j] = v;
        if (j + 1 == s && SP != s) B[ri * REC + j + 1] = v;
      }
    }
  };

  f32x2 p = {0.f, 0.f};
  // Pipeline, iteration i = 0 .. nchunks:  chains on chunk i - 1 (buffer (i - 1) & 1)  ||  divisors of chunk i;
  // barrier; the slices of chunk i go to buffer i & 1; barrier.
  // vmcnt counts vector loads IN ORDER: whatever is consumed soonest after its request sets the look-ahead of
  // everything requested before it.  So every vector load of iteration i is consumed in iteration i + L - 1 or later
  // (register rings of L slots, slot = chunk mod L, the loop unrolled by L so that slots are compile-time):
  //   slices  of chunk i + L - 1   (needs its order entries, requested in iteration i - (L - 1))
  //   order   of chunk i + 2L - 2
  //   cluster bounds (writer) of chunk i + L - 1
  //   span bounds of chunk i + 3L - 1 (they enter the ring of known spans, chunks i .. i + 2L - 1, L - 1 iterations later)
  // One chain phase (~0.3 us) does not cover a gather's latency (~2 us under load), L - 1 of them do.
  constexpr int L = FUSED_DEPTH, S = 2 * L;
  int sb[S], sn[S];                         // span bounds of chunks i .. i + 2L - 1 (arrived)
  int lb[L], ln[L];                         // span bounds in flight, slot = chunk mod L
  int ord[L][NR];                           // order entries, slot = chunk mod L
  f32x2 gx[L][NR][HP];                      // slices, slot = chunk mod L
  int clo[L], chi[L];                       // writer: its cluster's bounds, slot = chunk mod L
#pragma unroll
  for (int d = 0; d < S; d++) span(d, sb[d], sn[d]);
  // prologue: order of chunks 0 .. 2L - 3, slices and cluster bounds of chunks 0 .. L - 2
#pragma unroll
  for (int d = 0; d + 1 < L; d++) {
    load_order(d, sb[d], rows_in(d, sb[d], sn[d]), ord[d]);
    gather(d, rows_in(d, sb[d], sn[d]), ord[d], gx[d]);
    cluster_span(d, clo[d], chi[d]);
  }
#pragma unroll
  for (int d = L - 1; d <= 2 * L - 3; d++) load_order(d, sb[d], rows_in(d, sb[d], sn[d]), ord[d % L]);
#pragma unroll
  for (int d = 0; d + 1 < L; d++) span(S + d, lb[d], ln[d]);       // chunks 2L .. 3L - 2
  for (int i0 = 0; i0 <= nchunks; i0 += L) {
#pragma unroll
    for (int d = 0; d < L; d++) {
      const int i = i0 + d;
      if (i > nchunks) break;
      float *const Bc = bufs + (size_t)((i - 1) & 1) * FCHUNK * REC;   // chains read (i >= 1)
      float *const Bn = bufs + (size_t)(i & 1) * FCHUNK * REC;         // filled for chunk i
      int *const Mc = meta + ((i - 1) & 1) * 2 * G, *const Mn = meta + (i & 1) * 2 * G;
      const int b_new = lb[d], n_new = ln[d];                        // chunk i + 2L, requested L - 1 iterations ago
      span(i + 3 * L - 1, lb[(d + L - 1) % L], ln[(d + L - 1) % L]);
      // slot (d + L - 1) % L: its slices went to LDS and its cluster bounds to the writer in the previous iteration;
      // slot (d + L - 2) % L of `ord`: its entries fed the previous iteration's gather
      gather(i + L - 1, rows_in(i + L - 1, sb[L - 1], sn[L - 1]), ord[(d + L - 1) % L], gx[(d + L - 1) % L]);
      cluster_span(i + L - 1, clo[(d + L - 1) % L], chi[(d + L - 1) % L]);
      load_order(i + S - 2, sb[S - 2], rows_in(i + S - 2, sb[S - 2], sn[S - 2]), ord[(d + S - 2) % L]);
      if (wave == 3) {
        // divisors of chunk i: record (st + r) gets n = rows before + r + 1 and RN(1 / n)
        if (writer) {
          const int st = clo[d] - sb[0], ct = i < nchunks ? chi[d] - clo[d] : 0;   // span-relative start, rows of the cluster
          Mn[2 * l3] = st; Mn[2 * l3 + 1] = ct;
          for (int r = 0; r < ct; r++) {
            const float nf = (float)(int)(before + (unsigned)r + 1u);
            Bn[(st + r) * REC + SP] = __fdiv_rn(1.0f, nf);
            Bn[(st + r) * REC + SP + 1] = nf;
          }
          before += (unsigned)ct;
        }
      } else if (i >= 1) {
        // the chains of chunk i - 1
        const int st = chain_lane ? Mc[2 * cl] : 0, cnt = chain_lane ? Mc[2 * cl + 1] : 0;
        // (no wave-wide maximum of cnt up front: six dependent cross-lane steps, ~700 cycles per chunk -- a third of
        // the chain phase; the loops below ask "does any lane have rows left" once per batch instead)
        const float *rp = Bc + st * REC + 2 * pair;
        const float *yp = Bc + st * REC + SP;
        const f32x2 p0 = p;
        float lo = INFINITY, hi = 0.f;
        // the records of the NEXT four steps are requested before the current four are walked (their addresses do
        // not depend on the chain); rows past a lane's cluster are read (inside the buffer) and not used
        constexpr int RB = 4;
        auto fetch4 = [&](int r0, f32x2 (&xs)[RB], f32x2 (&yns)[RB]) {
#pragma unroll
          for (int u = 0; u < RB; u++) {
            const float *q = Bc + min(st + r0 + u, FCHUNK - 1) * REC;   // (the buffer holds FCHUNK records)
            xs[u] = *reinterpret_cast<const f32x2 *>(q + 2 * pair);
            yns[u] = *reinterpret_cast<const f32x2 *>(q + SP);
          }
        };
        auto walk4 = [&](int r0, const f32x2 (&xs)[RB], const f32x2 (&yns)[RB]) {
#pragma unroll
          for (int u = 0; u < RB; u++) {
            if (r0 + u < cnt) {
              const f32x2 a = xs[u] - p;
              lo = fminf(fminf(lo, fabsf(a.x)), fabsf(a.y));
              hi = fmaxf(fmaxf(hi, fabsf(a.x)), fabsf(a.y));
              const f32x2 y2 = {yns[u].x, yns[u].x}, nn = {-yns[u].y, -yns[u].y};
              const f32x2 q0 = a * y2;
              const f32x2 rr = __builtin_elementwise_fma(nn, q0, a);
              p = p + __builtin_elementwise_fma(rr, y2, q0);
            }
          }
        };
        f32x2 xa[RB], ya[RB], xb2[RB], yb2[RB];
        fetch4(0, xa, ya);
        for (int r0 = 0; __any(r0 < cnt); r0 += 2 * RB) {
          fetch4(r0 + RB, xb2, yb2);
          walk4(r0, xa, ya);
          fetch4(r0 + 2 * RB, xa, ya);
          walk4(r0 + RB, xb2, yb2);
        }
        // a zero, tiny, huge or NaN numerator somewhere in the wave's chunk: the plain division, from the chunk's start
        if (__any(!(lo > 8.673617379884035e-19f /* 2^-60 */ && hi < 1.152921504606847e18f /* 2^60 */))) {
          p = p0;
          for (int r = 0; __any(r < cnt); r++) {
            if (r < cnt) {
              const f32x2 x = *reinterpret_cast<const f32x2 *>(rp + r * REC);
              const float nf = yp[r * REC + 1];
              p.x = p.x + __fdiv_rn(x.x - p.x, nf);
              p.y = p.y + __fdiv_rn(x.y - p.y, nf);
            }
          }
        }
      }
      __syncthreads();          // the chains are done with Bc; the divisors of chunk i are in Bn
      store_rows(min(i, nchunks - 1), sb[0], rows_in(i, sb[0], sn[0]), gx[d], Bn);
      __syncthreads();
#pragma unroll
      for (int e = 0; e + 1 < S; e++) { sb[e] = sb[e + 1]; sn[e] = sn[e + 1]; }
      sb[S - 1] = b_new; sn[S - 1] = n_new;
    }
  }
  if (chain_lane) {
    const int cg = c_first + cl, j = 2 * pair;
    D.cout[cg * s + j] = p.x;
    if (j + 1 < s) D.cout[cg * s + j + 1] = p.y;
  }
}

}  // namespace

// Eligibility and launch.  `descs` as for kmeans_update_batch; every problem's xb scratch (n * sp floats) holds
// order and coff.  Returns false when the shape keeps the bucketed path.
bool kmeans_update_fused(const std::vector<UpdDesc> &descs, UpdDesc *d_descs, int n, int k, hipStream_t st) {
  static const bool on = [] { const char *e = getenv("GULON_UPDATE_FUSED"); return e && atoi(e) != 0; }();   // experiment: off
  const int np = (int)descs.size();
  if (!on || np == 0 || n <= 0 || k > 1024 || k < 1) return false;
  int smax = 1;
  bool compact = true, one_stride = true;
  for (const UpdDesc &D : descs) smax = std::max(smax, D.s);
  const int sp = (smax + 1) & ~1;
  for (const UpdDesc &D : descs) {
    compact = compact && D.ld == D.s && D.from == 0;
    one_stride = one_stride && ((D.s + 1) & ~1) == sp;
    // order + coff inside the bucket scratch: 2 n' + 2 chunks (k + 1) bytes <= 4 n sp
    if (D.xb == nullptr) return false;
  }
  if (!compact || !one_stride || sp > 16) return false;
  const int nchunks = ceil_div(n, FCHUNK);
  if ((size_t)nchunks * FCHUNK * 2 + (size_t)nchunks * (k + 1) * 2 > (size_t)n * sp * 4) return false;   // (tiny n with large k)
  const int HP = sp / 2;
  const int G = std::min(std::min(32, 192 / HP), k);
  const int groups = ceil_div(k, G);
  HIP_CHECK(hipMemcpyAsync(d_descs, descs.data(), sizeof(UpdDesc) * np, hipMemcpyHostToDevice, st));
  int key_bits = 0;
  while ((1 << key_bits) < k) key_bits++;
  const size_t shm_order = sizeof(unsigned) * 4 * (size_t)k;
  HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(sort_order), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)shm_order));
  hipLaunchKernelGGL(sort_order, dim3((unsigned)nchunks, np), dim3(256), shm_order, st, d_descs, n, k, key_bits, nchunks);
  HIP_CHECK(hipGetLastError());
  const size_t shm = sizeof(float) * 2 * FCHUNK * (size_t)(sp + 2) + sizeof(int) * 4 * (size_t)G;
  auto go = [&](auto kern) {
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    hipLaunchKernelGGL(kern, dim3(8u * (unsigned)ceil_div(np, 8) * (unsigned)groups), dim3(FUSED_THREADS), shm, st, d_descs,
                       np, n, k, G, groups, nchunks);
  };
  switch (sp) {
    case 2: go(update_fused<2>); break;
    case 4: go(update_fused<4>); break;
    case 6: go(update_fused<6>); break;
    case 8: go(update_fused<8>); break;
    case 10: go(update_fused<10>); break;
    case 12: go(update_fused<12>); break;
    case 14: go(update_fused<14>); break;
    default: go(update_fused<16>); break;
  }
  HIP_CHECK(hipGetLastError());
  return true;
}

}  // namespace gulon
