// KMeans.fromAssignment (KMeans.scala:198-226) without the regrouped copy of the data.
//
// The running mean  c <- c + (x - c) / n  is one sequential chain per (cluster, dimension), over the cluster's rows in
// row order.  kmeans.hip feeds the chains by regrouping every sub-quantizer's slices by cluster (sort_place: 12 GB read,
// 12.8 GB written in 160-byte runs at BASELINE config 3) and streaming the copy back (update_chains_pk: 12.8 GB): 37.6 GB
// moved for 12 GB of data, 9.3-10.2 ms per full-PQ iteration.  Here the data is read ONCE, in row order:
//
//   * once per training, every sub-quantizer's slice is stored PAIR-MAJOR: xp[pair][row] = (x[row][2 pair],
//     x[row][2 pair + 1]) -- the two dimensions a lane of the packed fp32 pipe carries (pack_pairs; an odd last dimension
//     is paired with itself and its second result dropped);
//   * per update, stream_order computes for every chunk of CH rows the stable order of its rows by cluster (ord[row] =
//     the row's POSITION in that order, 2 bytes per row) and where each cluster's rows start in it (coff) -- the
//     counting sort of sort_place without the data;
//   * stream_chains: ONE workgroup per (sub-quantizer, dimension pair): chain threads (thread = cluster) and loader
//     threads.  It walks the chunks in order: the loader waves write the chunk's 8-byte values to LDS AT THEIR POSITIONS
//     in the chunk's stable order, one chunk ahead of the chain waves (and request the chunk after that meanwhile), and
//     every chain thread applies the rows of ITS cluster -- consecutive LDS entries -- to the pair of running means it
//     keeps in registers.  Row order inside a cluster is the chunks' order followed by the stable order inside a chunk:
//     the reference's.
//
// Per step the arithmetic is update_chains_pk's: a = x - c, the correctly rounded quotient a / n as the corrected
// product with y = RN(1 / n) (mean_quotient_fast, kmeans.hip) -- y is computed a group of four steps ahead (n is known
// long before c is): v_rcp_f32 and one fma-corrected Newton step, which IS the correctly rounded reciprocal for every
// integer n < 2^24 (checked exhaustively against 1.0f / n by gulon_selftest_mean_division) -- and the plain division
// where the corrected form is not trusted: numerators below 2^-100 (found per chunk, which is then redone), values of
// 2^99 and beyond / infinities / NaNs anywhere in the slice (found when it is packed), counts from 2^24.
// Traffic per update: the pair-major data once (12.8 GB) + order and offsets (2 bytes per row and pair workgroup,
// mostly served by L2: the pair workgroups of a sub-quantizer walk the same chunks at the same pace).
// BASELINE config 3 (10 M x 300, 32 sub-quantizers, k = 256), per update: stream_order 1.0 ms + stream_chains 3.6 ms
// against 9.3-10.2 ms; the chain kernel's time is its longest chain's: rows of the largest cluster of a wave x 15
// instructions per step x the 7-8 cycles a lone wave needs per instruction.
#include "kmeans.hpp"

namespace gulon {

namespace {

#ifndef GULON_STREAM_CH
#define GULON_STREAM_CH 8192
#endif
constexpr int STREAM_CH = GULON_STREAM_CH;     // rows per chunk (positions are 16-bit; two chunks of values fill 128 of the CU's 160 KiB LDS)

__global__ void pack_pairs(const float *__restrict__ xs /* [n][s] */, long long n, long long ns, int s, int pairs,
                           f32x2 *__restrict__ xp /* [pairs][ns] */, int *__restrict__ wild) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  bool w = false;
  if (t < n * pairs) {
    const int pr = (int)(t / n);
    const long long r = t - (long long)pr * n;
    const int j = 2 * pr;
    const float a = xs[(size_t)r * s + j];
    const float b = j + 1 < s ? xs[(size_t)r * s + j + 1] : a;
    xp[(size_t)pr * ns + r] = f32x2{a, b};
    w = !(fabsf(a) < 6.338253e29f /* 2^99 */) || !(fabsf(b) < 6.338253e29f);   // huge, infinite or NaN
  }
  if (__ballot(w) != 0ull && (threadIdx.x & 63) == 0) atomicOr(wild, 1);
}

// the stable order of a chunk's rows by cluster: ord[chunk * CH + row] = the row's position, coff[chunk * (k + 1) + c] = first
// position of cluster c (coff[..][k] = rows of the chunk).  256 threads, wave w = rows [w CH/4, (w + 1) CH/4) in steps
// of 64; the lanes holding the same cluster are found with one ballot per key bit (as sort_place does).
__global__ __launch_bounds__(256) void stream_order(const StreamDesc *__restrict__ descs, int n, int k, int key_bits) {
  const StreamDesc D = descs[blockIdx.y];
  const auto assign = as_global(D.assign);
  extern __shared__ unsigned sh[];          // wh[4][k], then tot[k]
  unsigned *wh = sh, *tot = sh + 4 * k;
  __shared__ unsigned wave_tot[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long long chunk = blockIdx.x;
  const long long r0 = chunk * STREAM_CH;
  const int rows = (int)min((long long)STREAM_CH, (long long)n - r0);
  constexpr int STEPS = STREAM_CH / 256;
  for (int e = tid; e < 4 * k; e += 256) wh[e] = 0;
  __syncthreads();
  const unsigned long long lt = (1ull << lane) - 1ull;
  int keys[STEPS];
  unsigned long long same[STEPS];
#pragma unroll
  for (int t = 0; t < STEPS; t++) {
    const int lr = wave * (STREAM_CH / 4) + t * 64 + lane;
    keys[t] = lr < rows ? assign[r0 + lr] : 0;
  }
#pragma unroll
  for (int t = 0; t < STEPS; t++) {
    const int lr = wave * (STREAM_CH / 4) + t * 64 + lane;
    const bool valid = lr < rows;
    unsigned long long sm = __ballot(valid);
    for (int bit = 0; bit < key_bits; bit++) {
      const unsigned long long bm = __ballot((keys[t] >> bit) & 1);
      sm &= ((keys[t] >> bit) & 1) ? bm : ~bm;
    }
    same[t] = valid ? sm : 0ull;
    if (valid && (sm & lt) == 0ull) wh[wave * k + keys[t]] += (unsigned)__popcll(sm);   // only this wave touches wh[wave]
  }
  __syncthreads();
  {   // exclusive scan of the chunk's cluster totals: thread t owns clusters [t per, t per + per)
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
      tot[c] = base;
      wh[c] = base; wh[k + c] = base + t0; wh[2 * k + c] = base + t0 + t1; wh[3 * k + c] = base + t0 + t1 + t2;
      base += t0 + t1 + t2 + t3;
    }
  }
  __syncthreads();
  const auto coff = as_global(D.coff) + (size_t)chunk * (k + 1);
  for (int c = tid; c < k; c += 256) coff[c] = (unsigned short)tot[c];
  if (tid == 0) coff[k] = (unsigned short)rows;
  const auto ord = as_global(D.ord) + (size_t)chunk * STREAM_CH;
#pragma unroll
  for (int t = 0; t < STEPS; t++) {
    const int lr = wave * (STREAM_CH / 4) + t * 64 + lane;
    const bool valid = lr < rows;
    const int key = keys[t];
    const unsigned b = valid ? wh[wave * k + key] : 0u;          // same-wave LDS accesses execute in order
    const unsigned pos = b + (unsigned)__popcll(same[t] & lt);
    if (valid && (same[t] & lt) == 0ull) wh[wave * k + key] = b + (unsigned)__popcll(same[t]);
    ord[lr] = valid ? (unsigned short)pos : (unsigned short)STREAM_CH;   // (a padding row of the last chunk: not staged)
  }
}

// T chain threads = clusters (k <= T) and L loader threads; one workgroup per (dimension pair, sub-quantizer).
// The workgroup is alone on its CU (one per (sub-quantizer, pair): 160 at BASELINE config 3), so nothing hides a memory
// round trip but its own look-ahead, and a wave that is alone on its SIMD issues one instruction every 7-8 cycles
// whatever the instruction is (scripts/ubench_latency.hip): the chain waves are as fast as their instruction stream is
// short.  So the chunk's traffic is not theirs: the LOADER waves (one per SIMD, beside the chain wave) request the values
// and positions of the chunk after next, and write the next chunk's values to LDS at their positions, while the chain
// waves step through the current chunk; one barrier per chunk hands the buffers over.  (All in the chain waves, the 34
// loads of a chunk -- issued in a burst that the CU's address unit takes 1300 cycles to absorb -- and the 32 LDS writes
// were 30 % of their time.)  L = 0 (more than 512 clusters: no room for loader waves in a workgroup): the chain
// threads do both.
template <int T, int L>
__global__ __launch_bounds__(T + L) void stream_chains(const StreamDesc *__restrict__ descs, int n, int k, int nchunks) {
  const StreamDesc D = descs[blockIdx.y];
  const int pair = blockIdx.x;
  const int s = D.s;
  if (2 * pair >= s) return;
  constexpr int CH = STREAM_CH;
  constexpr int LT = L ? L : T;                           // threads that move the chunk
  constexpr int ND = CH / 2 / LT;                         // units (two rows: 16 bytes of data, 4 bytes of positions) per such thread
  static_assert(ND >= 1 && CH / 2 % LT == 0, "chunk data divides over the threads");
  constexpr int RING_CAP = L ? 1 : T >= 1024 ? 3 : 8;     // (1024 threads: 128 registers per lane)
  constexpr int RING = 60 / (2 * ND + 2) < RING_CAP ? 60 / (2 * ND + 2) : RING_CAP;   // chunks in flight (vmcnt counts 63 loads at most)
  extern __shared__ f32x2 dat_dyn[];                      // [buffer][position] the chunk's values in its stable order (+ slack: groups of 4 read past the last)
  f32x2 (*dat_s)[CH + 8] = reinterpret_cast<f32x2 (*)[CH + 8]>(dat_dyn);
  const bool loader = L != 0 && (int)threadIdx.x >= T;
  const int c = L != 0 && loader ? (int)threadIdx.x - T : (int)threadIdx.x;   // cluster of a chain thread / index of a loader thread
  const bool active = !loader && c < k;
  // (global address space stated: a flat load counts in lgkmcnt as well, and every LDS wait of the chain loop would
  // then wait for the chunks in flight -- common.hpp, as_global)
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  const auto src_d = reinterpret_cast<gptr<const u32x4>>(as_global(D.xp)) + (size_t)pair * (D.ns / 2);   // ns = chunks x CH rows
  const auto coff = as_global(D.coff);
  const auto src_o = reinterpret_cast<gptr<const unsigned>>(as_global(D.ord));                            // two positions per dword
  const auto cout = as_global(D.cout);
  const unsigned cc = active ? c : 0;
  // every chunk is whole in memory (data and positions are padded to a multiple of CH rows; a padding row's position is
  // CH: the slack entry behind the buffer), so the loads need no guards -- and no branches for the load counters to be
  // merged over
  auto load_data = [&](int t, u32x4 (&d)[ND], unsigned (&o)[ND]) {
    const int tc = min(t, nchunks - 1);
    const auto cd = src_d + (size_t)tc * (CH / 2);
    const auto co = src_o + (size_t)tc * (CH / 2);
#pragma unroll
    for (int u = 0; u < ND; u++) { d[u] = cd[(unsigned)c + (unsigned)(u * LT)]; o[u] = co[(unsigned)c + (unsigned)(u * LT)]; }
  };
  auto load_range = [&](int t, int &cs, int &ce) {       // this cluster's first / one-past-last position in the chunk
    const auto cf = coff + (size_t)min(t, nchunks - 1) * (k + 1);
    cs = cf[cc];
    ce = cf[cc + 1u];
  };
  auto stash = [&](int buf, const u32x4 (&d)[ND], const unsigned (&o)[ND]) {
#pragma unroll
    for (int u = 0; u < ND; u++) {
      dat_s[buf][o[u] & 0xFFFFu] = f32x2{__uint_as_float(d[u].x), __uint_as_float(d[u].y)};
      dat_s[buf][o[u] >> 16] = f32x2{__uint_as_float(d[u].z), __uint_as_float(d[u].w)};
    }
  };
  if (L != 0 && loader) {
    // period t (the chain waves read dat_s[t & 1]): chunk t + 1 goes from the registers to dat_s[(t + 1) & 1], chunk
    // t + 2 is requested; the barriers are the chain waves' (one before the first chunk, one after every chunk)
    u32x4 rd[ND];
    unsigned ro[ND];
    load_data(0, rd, ro);
    stash(0, rd, ro);
    load_data(1, rd, ro);
    __syncthreads();
    for (int t = 0; t < nchunks; t++) {
      if (t + 1 < nchunks) stash((t + 1) & 1, rd, ro);
      load_data(t + 2, rd, ro);
      __syncthreads();
    }
    return;
  }
  u32x4 rd[L ? 1 : RING][L ? 1 : ND];
  unsigned ro[L ? 1 : RING][L ? 1 : ND];
  int rs[RING], re[RING];
  if constexpr (L == 0) {
#pragma unroll
    for (int r = 0; r < RING; r++) { load_data(r, rd[r], ro[r]); load_range(r, rs[r], re[r]); }
    stash(0, rd[0], ro[0]);
  } else {
    load_range(0, rs[0], re[0]);
  }
  __syncthreads();
  // values that the corrected product cannot be trusted with (see below) somewhere in this sub-quantizer's data: every
  // chunk takes the plain division
  const bool wild = *as_global(D.wild) != 0;
  f32x2 p = {0.f, 0.f};
  int cnt = 0;                                        // rows of the cluster so far
  // With one wave per SIMD the wave issues one instruction -- of ANY kind -- every 7-8 cycles (scripts/ubench_latency.hip:
  // dependent or not), so the kernel is as fast as the instruction stream of the wave with the largest clusters is
  // short.  A chain step is the five packed operations of the corrected product (a = x - p; q0 = a y; r = a - n q0;
  // q = q0 + r y; p += q) plus:
  //   * its divisor and reciprocal, computed a GROUP (four steps) ahead, packed: {n, n+1}, {n+2, n+3} -> four v_rcp_f32
  //     and two packed fma-corrected Newton steps each pair (RN(1 / n): rcp_rn_int).  Nothing here depends on whether
  //     the steps happen: a lane's valid steps of a chunk are a prefix, and the next chunk starts from its row count;
  //   * the smallest exponent of a nonzero numerator (v_frexp_exp gives 0 for a zero: a zero numerator's quotient is +0
  //     either way), accumulated and looked at once per chunk: below 2^-100 the quotient can be subnormal, where the
  //     correction can round a tie the other way; a chunk that fails is redone from its saved start with the plain
  //     division.  The other end needs no test in the loop: a running mean stays inside the hull of its rows (RN is
  //     monotone), so |a| < 2^100 follows from |x| < 2^99 for all finite x -- checked once, when the data is packed;
  //   * the lane mask of the step (compare, save, restore).
  f32x2 nA = {1.f, 2.f}, nB = {3.f, 4.f}, yA = {1.f, 1.f}, yB = {1.f, 1.f};   // this group's divisors and reciprocals
  int emin = 0;
  auto recip2 = [](const f32x2 nn) {
    f32x2 y = {__builtin_amdgcn_rcpf(nn.x), __builtin_amdgcn_rcpf(nn.y)};
    const f32x2 e = __builtin_elementwise_fma(-nn, y, f32x2{1.f, 1.f});
    return __builtin_elementwise_fma(e, y, y);
  };
  auto step = [&](const f32x2 x, const f32x2 nn /* {-n, -n} */, const f32x2 y2) {
    const f32x2 a = x - p;
    const f32x2 q0 = a * y2;
    const f32x2 r = __builtin_elementwise_fma(nn, q0, a);
    p = p + __builtin_elementwise_fma(r, y2, q0);
    emin = min(emin, min(__builtin_amdgcn_frexp_expf(a.x), __builtin_amdgcn_frexp_expf(a.y)));
  };
  // four steps on x[0..3] for the rows i .. i + 3 of the lane's range; meanwhile the next group's values are requested
  // into xn and its divisors / reciprocals computed
  auto group = [&](const f32x2 *dat, int i, int ce, const f32x2 (&x)[4], f32x2 (&xn)[4]) {
    const int ib = min(i + 4, CH);
#pragma unroll
    for (int u = 0; u < 4; u++) xn[u] = dat[ib + u];
    const f32x2 nA2 = nA + 4.0f, nB2 = nB + 4.0f;
    const f32x2 yA2 = recip2(nA2), yB2 = recip2(nB2);
    const int left = ce - i;                           // (nested: the lanes of a step are a subset of the previous step's)
    if (left > 0) {
      step(x[0], -__builtin_shufflevector(nA, nA, 0, 0), __builtin_shufflevector(yA, yA, 0, 0));
      if (left > 1) {
        step(x[1], -__builtin_shufflevector(nA, nA, 1, 1), __builtin_shufflevector(yA, yA, 1, 1));
        if (left > 2) {
          step(x[2], -__builtin_shufflevector(nB, nB, 0, 0), __builtin_shufflevector(yB, yB, 0, 0));
          if (left > 3) step(x[3], -__builtin_shufflevector(nB, nB, 1, 1), __builtin_shufflevector(yB, yB, 1, 1));
        }
      }
    }
    nA = nA2; nB = nB2; yA = yA2; yB = yB2;
  };
  // chunk t: its values are in dat_s[t & 1]; register slot t % RING is free (stashed one chunk ago) and takes chunk
  // t + RING; chunk t + 1 leaves slot (t + 1) % RING for dat_s[(t + 1) & 1] after the chain steps
#ifdef STREAM_STAMPS
  unsigned long long st_head = 0, st_loop = 0, st_stash = 0, st_bar = 0, st_steps = 0, st_vm = 0;
#define STAMP(v) const unsigned long long v = __builtin_readcyclecounter()
#else
#define STAMP(v)
#endif
  auto chunk = [&](int t, int slot_cur, int slot_next) {
    STAMP(t0);
    const int cs = active ? rs[slot_cur] : 0, ce = active ? re[slot_cur] : 0;
    if constexpr (L == 0) load_data(t + RING, rd[slot_cur], ro[slot_cur]);
    load_range(t + RING, rs[slot_cur], re[slot_cur]);
    const f32x2 *dat = dat_s[t & 1];
    const f32x2 p_start = p;
    const int cnt_end = cnt + (ce - cs);
    const float n1 = (float)(cnt + 1);
    nA = f32x2{n1, n1 + 1.f}; nB = nA + 2.0f;
    yA = recip2(nA); yB = recip2(nB);
    emin = 0;
    // this cluster's rows of the chunk: positions cs .. ce - 1, four at a time (lanes whose cluster is exhausted sit
    // the steps out; the rows of dat_s past CH exist)
    f32x2 xa[4], xb[4];
#pragma unroll
    for (int u = 0; u < 4; u++) xa[u] = dat[min(cs, CH) + u];
    STAMP(t1);
    for (int i = cs; __ballot(i < ce) != 0ull; i += 8) {
#ifdef STREAM_STAMPS
      st_steps += 4;
#endif
      group(dat, i, ce, xa, xb);
      if (__ballot(i + 4 < ce) == 0ull) break;
#ifdef STREAM_STAMPS
      st_steps += 4;
#endif
      group(dat, i + 4, ce, xb, xa);
    }
    // trusted: every nonzero numerator at least 2^-100 and every divisor below 2^24 (the float count is exact; the
    // corrected product holds for EVERY such divisor with y = RN(1 / n): tests/test_markstein_exhaustive.py for the
    // divisors 2^j - 1 the proofs single out, gulon_selftest_mean_division)
    STAMP(t2);
    const bool redo = wild || emin < -99 || cnt_end >= (1 << 24);
    if (redo) {
      p = p_start;
      int c2 = cnt;
      for (int i = cs; i < ce; i++) {
        c2++;
        const float n2 = (float)c2;                 // RN, as the JVM converts the Int count (KMeans.scala:218)
        const f32x2 a = dat[i] - p;
        p = p + f32x2{__fdiv_rn(a.x, n2), __fdiv_rn(a.y, n2)};
      }
    }
    cnt = cnt_end;
#ifdef STREAM_STAMPS
    const unsigned long long t2b = __builtin_readcyclecounter();
    st_vm += t2b - t2;
#endif
    if constexpr (L == 0)
      if (t + 1 < nchunks) stash((t + 1) & 1, rd[slot_next], ro[slot_next]);
    STAMP(t3);
    __syncthreads();
#ifdef STREAM_STAMPS
    const unsigned long long t4 = __builtin_readcyclecounter();
    st_head += t1 - t0; st_loop += t2 - t1; st_stash += t3 - t2; st_bar += t4 - t3;
#endif
  };
  int t = 0;
  for (; t + RING <= nchunks; t += RING) {
#pragma unroll
    for (int r = 0; r < RING; r++) chunk(t + r, r, (r + 1) % RING);
  }
#pragma unroll
  for (int r = 0; r < RING; r++)
    if (t + r < nchunks) chunk(t + r, r, (r + 1) % RING);
#ifdef STREAM_STAMPS
  if ((c & 63) == 0 && blockIdx.x < 2 && blockIdx.y < 2)
    printf("stream_chains wg (%d,%d) wave %d: head %llu loop %llu stash %llu (check/redo %llu) barrier %llu cycles, %llu step slots, %d chunks\n",
           (int)blockIdx.x, (int)blockIdx.y, c >> 6, st_head, st_loop, st_stash, st_vm, st_bar, st_steps, nchunks);
#endif
  if (active) {
    const int j = 2 * pair;
    cout[(size_t)c * s + j] = p.x;
    if (j + 1 < s) cout[(size_t)c * s + j + 1] = p.y;
  }
}

}  // namespace

bool stream_update_supported(int n, int k, int s) {
  static const bool off = [] { const char *e = getenv("GULON_UPDATE_STREAM"); return e && atoi(e) == 0; }();
  return !off && k >= 1 && k <= 1024 && s >= 1 && n >= 1;
}

// rows per pair of the pair-major copy: whole chunks (the chain kernel reads chunks without guards)
long long stream_padded_rows(int n) { return (((long long)n + STREAM_CH - 1) / STREAM_CH) * STREAM_CH; }

size_t stream_order_words(int n, int k, size_t *coff_words) {
  const size_t nchunks = ((size_t)n + STREAM_CH - 1) / STREAM_CH;
  if (coff_words) *coff_words = nchunks * (size_t)(k + 1);
  return nchunks * STREAM_CH;
}

void stream_pack_pairs(const float *xs, int n, int s, float *xp, int *wild, hipStream_t st) {
  const int pairs = (s + 1) / 2;
  const long long total = (long long)n * pairs;
  GULON_UNSUPPORTED(total >= (1ll << 32) * 256, "slice too large");
  hipLaunchKernelGGL(pack_pairs, dim3((unsigned)ceil_div(total, 256LL)), dim3(256), 0, st, xs, (long long)n,
                     stream_padded_rows(n), s, pairs, reinterpret_cast<f32x2 *>(xp), wild);
  HIP_CHECK(hipGetLastError());
}

// the chunks' stable order of every problem's current assignment (descs[..].assign -> ord, coff)
void kmeans_stream_order(const std::vector<StreamDesc> &descs, StreamDesc *d_descs, int n, int k, hipStream_t st) {
  const int np = (int)descs.size();
  if (np == 0) return;
  HIP_CHECK(hipMemcpyAsync(d_descs, descs.data(), sizeof(StreamDesc) * np, hipMemcpyHostToDevice, st));
  const int nchunks = ceil_div(n, STREAM_CH);
  int key_bits = 1;
  while ((1 << key_bits) < k) key_bits++;
  const size_t lds = sizeof(unsigned) * 5 * (size_t)k;
  hipLaunchKernelGGL(stream_order, dim3(nchunks, np), dim3(256), lds, st, d_descs, n, k, key_bits);
  HIP_CHECK(hipGetLastError());
}

// the running means along that order (xp, ord, coff -> cout)
void kmeans_stream_chains(const std::vector<StreamDesc> &descs, StreamDesc *d_descs, int n, int k, hipStream_t st) {
  const int np = (int)descs.size();
  if (np == 0) return;
  HIP_CHECK(hipMemcpyAsync(d_descs, descs.data(), sizeof(StreamDesc) * np, hipMemcpyHostToDevice, st));
  const int nchunks = ceil_div(n, STREAM_CH);
  int pairs_max = 1;
  for (const StreamDesc &D : descs) pairs_max = std::max(pairs_max, (D.s + 1) / 2);
  const size_t chain_lds = sizeof(float) * 2 * 2 * (STREAM_CH + 8);
  static const bool lds_set = [&] {
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(stream_chains<256, 256>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)chain_lds));
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(stream_chains<512, 256>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)chain_lds));
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(stream_chains<1024, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)chain_lds));
    return true;
  }();
  (void)lds_set;
  if (k <= 256) hipLaunchKernelGGL((stream_chains<256, 256>), dim3(pairs_max, np), dim3(512), chain_lds, st, d_descs, n, k, nchunks);
  else if (k <= 512) hipLaunchKernelGGL((stream_chains<512, 256>), dim3(pairs_max, np), dim3(768), chain_lds, st, d_descs, n, k, nchunks);
  else hipLaunchKernelGGL((stream_chains<1024, 0>), dim3(pairs_max, np), dim3(1024), chain_lds, st, d_descs, n, k, nchunks);
  HIP_CHECK(hipGetLastError());
}

}  // namespace gulon

#ifdef GULON_TEST_HOOKS
using namespace gulon;
// KMeans.fromAssignment (KMeans.scala:198-226) of ONE slice through the streamed update, whatever the caller's shape
// (tests/test_gpu_kmeans.py: the library proper takes this path inside PQ training only)
GULON_API int32_t gulon_selftest_stream_update(const float *x, int32_t n, int32_t ld, int32_t from, int32_t s, int32_t k,
                                               const int32_t *assign, float *centroids_out) {
  return guarded([&] {
    GULON_REQUIRE(x && assign && centroids_out && n >= 1 && s >= 1 && from >= 0 && from + s <= ld, "bad arguments");
    GULON_REQUIRE(stream_update_supported(n, k, s), "shape not supported by the streamed update");
    std::vector<float> slice((size_t)n * s);
    for (int r = 0; r < n; r++) memcpy(&slice[(size_t)r * s], x + (size_t)r * ld + from, sizeof(float) * s);
    DevBuf<float> xs((size_t)n * s), cout((size_t)k * s);
    DevBuf<int> a(n), wild(1);
    const size_t ns = (size_t)stream_padded_rows(n);
    DevBuf<float> xp(2 * ns * (size_t)((s + 1) / 2));
    size_t coff_words = 0;
    DevBuf<unsigned short> ord(stream_order_words(n, k, &coff_words)), coff(coff_words);
    DevBuf<StreamDesc> d_desc(1);
    HIP_CHECK(hipMemcpy(xs.p, slice.data(), sizeof(float) * slice.size(), hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(a.p, assign, sizeof(int) * (size_t)n, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemset(wild.p, 0, sizeof(int)));
    HIP_CHECK(hipMemset(xp.p, 0, sizeof(float) * 2 * ns * (size_t)((s + 1) / 2)));
    HIP_CHECK(hipMemset(cout.p, 0, sizeof(float) * (size_t)k * s));
    stream_pack_pairs(xs.p, n, s, xp.p, wild.p, 0);
    StreamDesc D;
    D.assign = a.p; D.xp = xp.p; D.ord = ord.p; D.coff = coff.p; D.cout = cout.p; D.wild = wild.p;
    D.s = s; D.pad = 0; D.ns = (long long)ns;
    kmeans_stream_order({D}, d_desc.p, n, k, 0);
    kmeans_stream_chains({D}, d_desc.p, n, k, 0);
    HIP_CHECK(hipDeviceSynchronize());
    HIP_CHECK(hipMemcpy(centroids_out, cout.p, sizeof(float) * (size_t)k * s, hipMemcpyDeviceToHost));
  });
}
#endif
