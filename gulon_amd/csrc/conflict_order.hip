// Row order inside the 64-row blocks of the filter's code copy, chosen for the LDS bank conflicts of its table gathers.
//
// The main stage of the filtered scan (filter.hip) is bound by one instruction: a `ds_read_b128` per (row block,
// quantizer) in which every lane fetches the 16-byte table entry of ITS row's code -- a random gather.  What such a
// gather costs on gfx950 was measured one address pattern at a time (scripts/micro/lds_pattern.hip, 262 patterns,
// model error 0.27 cycles rms):
//   * a bank column is 16 bytes wide, 16 of them (address bits [7:4]);
//   * the 64 lanes are served in four groups of 16, one after the other:
//       {0-3, 12-15, 20-23, 24-27}, {4-7, 8-11, 16-19, 28-31} and the same sets + 32 (found by exchanging the
//       addresses of two lanes of a conflict-free pattern: nothing changes iff both are in one group);
//   * a group takes as many cycles as its fullest bank column holds DISTINCT addresses (equal addresses merge);
//   * cycles = sum over the four groups (+ 0.5), with a floor of 5.4 for the instruction itself.
// Random codes: 11.7 cycles on average (measured 12.1); conflict-free: 4 (measured 5.4).
//
// Which rows share a group is free to choose: the block's 64 rows are dealt to the four groups so that, summed over
// the quantizers whose look-ups go through LDS, the fullest columns are as empty as possible.  One wave per block,
// lane = row: for every row in turn all 63 exchanges with another row are evaluated at once (exact change of the sum
// of maxima from per-(group, quantizer, column) counters in LDS; ties broken by the sum of squared counts) and the
// best one is applied if it lowers the potential -- 64 steps per round, the potential falls strictly, so it ends.
// Simulated on uniform codes (13 quantizers): 12.3 -> 9.6 cycles per gather after one round, 9.5 after three.
//
// The permuted copy serves the filter kernel only; `perm[block * 64 + lane]` = the row's place in the block (what
// a surviving lane adds to the block's first row id, and what the from/until masks of boundary blocks test).
#include "scan.hpp"

namespace gulon {
namespace {

__device__ __forceinline__ int lds_group_of_lane(int lane) { return 2 * (lane >> 5) + ((0x96 >> ((lane & 31) >> 2)) & 1); }
// the r-th lane (r < 16) of group g
__device__ __forceinline__ int lds_lane_of_group(int g, int r) {
  const int quads = (g & 1) ? 0x7421 : 0x6530;   // quads of the half-wave in the group, one nibble each
  return 32 * (g >> 1) + 4 * ((quads >> (4 * (r >> 2))) & 15) + (r & 3);
}

template <int NQ /* quantizers counted: the first NQ bytes of a row's code word */>
__global__ __launch_bounds__(256) void conflict_order(const uint4 *__restrict__ src, uint4 *__restrict__ dst,
                                                      uint8_t *__restrict__ perm, long long nblk, int rounds) {
  __shared__ uint32_t cnt_s[4][4 * 16 * 4];   // per wave: [group][quantizer (16)][column] bytes, four columns per word
  __shared__ uint16_t st_s[4][4 * 16];        // per wave: [group][quantizer] fullest column | number of such columns << 8
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long blk = (long long)blockIdx.x * 4 + wave;
  if (blk >= nblk) return;   // waves are independent: no workgroup barrier below
  uint32_t *cw = cnt_s[wave];
  uint8_t *cb = reinterpret_cast<uint8_t *>(cw);
  uint16_t *st = st_s[wave];
  const uint4 w = src[blk * 64 + lane];
  const uint32_t ws[4] = {w.x, w.y, w.z, w.w};
  unsigned long long cols = 0;   // nibble q = bank column of the row's entry of quantizer q
#pragma unroll
  for (int q = 0; q < 16; q++) cols |= (unsigned long long)((ws[q >> 2] >> (8 * (q & 3))) & 15u) << (4 * q);
  int grp = lds_group_of_lane(lane);
  for (int e = lane; e < 256; e += 64) cw[e] = 0;
  // same-wave LDS operations execute in order
#pragma unroll
  for (int q = 0; q < NQ; q++) {
    const int c = (int)(cols >> (4 * q)) & 15;
    atomicAdd(&cw[(grp * 16 + q) * 4 + (c >> 2)], 1u << (8 * (c & 3)));
  }
  auto restat = [&](int g, int q) {
    int M = 0, nM = 0;
#pragma unroll
    for (int c4 = 0; c4 < 4; c4++) {
      const uint32_t v = cw[(g * 16 + q) * 4 + c4];
#pragma unroll
      for (int b = 0; b < 4; b++) {
        const int x = (v >> (8 * b)) & 255;
        if (x > M) { M = x; nM = 1; } else if (x == M) nM++;
      }
    }
    st[g * 16 + q] = (uint16_t)(M | (nM << 8));
  };
  if ((lane & 15) < NQ) restat(lane >> 4, lane & 15);
  const int cols_lo = (int)(cols & 0xffffffffu), cols_hi = (int)(cols >> 32);
  for (int step = 0; step < rounds * 64; step++) {
    const int r = step & 63;
    const int a = __builtin_amdgcn_readlane(grp, r);
    const unsigned long long ci64 = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane(cols_hi, r) << 32) |
                                    (unsigned)__builtin_amdgcn_readlane(cols_lo, r);
    const int b = grp;
    int key = 1 << 30;
    if (b != a) {
      int d = 0, sec = 0;
#pragma unroll
      for (int q = 0; q < NQ; q++) {
        const int ci = (int)(ci64 >> (4 * q)) & 15, cj = (int)(cols >> (4 * q)) & 15;
        if (ci != cj) {
          const int sa = st[a * 16 + q], sb = st[b * 16 + q];
          const int Ma = sa & 255, na = sa >> 8, Mb = sb & 255, nb = sb >> 8;
          const int ca_i = cb[(a * 16 + q) * 16 + ci], ca_j = cb[(a * 16 + q) * 16 + cj];
          const int cb_i = cb[(b * 16 + q) * 16 + ci], cb_j = cb[(b * 16 + q) * 16 + cj];
          d += max((ca_i == Ma && na == 1) ? Ma - 1 : Ma, ca_j + 1) - Ma;
          d += max((cb_j == Mb && nb == 1) ? Mb - 1 : Mb, cb_i + 1) - Mb;
          sec += (ca_j - ca_i + 1) + (cb_i - cb_j + 1);
        }
      }
      key = d * 2048 + sec;
    }
    // the best exchange of row r: minimum of (key, lane) over the wave
    int best = ((key + (1 << 20)) << 6) | lane;
    if (key == (1 << 30)) best = 0x7fffffff;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) best = min(best, __shfl_xor(best, o));
    const int bkey = (best >> 6) - (1 << 20);
    if (best == 0x7fffffff || bkey >= 0) continue;
    const int j = best & 63;
    const int gb = __builtin_amdgcn_readlane(grp, j);
    const unsigned long long cj64 = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane(cols_hi, j) << 32) |
                                    (unsigned)__builtin_amdgcn_readlane(cols_lo, j);
    if (lane < NQ) {
      const int q = lane;
      const int ci = (int)(ci64 >> (4 * q)) & 15, cj = (int)(cj64 >> (4 * q)) & 15;
      if (ci != cj) {
        cb[(a * 16 + q) * 16 + ci] -= 1; cb[(a * 16 + q) * 16 + cj] += 1;
        cb[(gb * 16 + q) * 16 + cj] -= 1; cb[(gb * 16 + q) * 16 + ci] += 1;
        restat(a, q);
        restat(gb, q);
      }
    }
    if (lane == r) grp = gb;
    if (lane == j) grp = a;
  }
  // places: the rows of group g take its lanes in row order
  int target = lane;
  const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
  for (int g = 0; g < 4; g++) {
    const unsigned long long mk = __ballot(grp == g);
    if (grp == g) target = lds_lane_of_group(g, __popcll(mk & lt));
  }
  dst[blk * 64 + target] = w;
  perm[blk * 64 + target] = (uint8_t)lane;
}

// ---- the same ordering over a WINDOW of four blocks (256 rows) ------------------------------------------------------
// A row may land in any of the window's 4 x 4 lane groups: sixteen rows per group, 16 groups to balance instead of 4 --
// simulated on uniform codes the model cost of a gather falls to 8.7 cycles against 9.6 inside one block (12.3 plain).
// One workgroup per window, thread = row: for every row in turn its exchange with each of the other 255 rows is
// evaluated by that row's thread (the exact change of the sum of the fullest columns, as above), the best one of the
// workgroup is applied if it lowers the potential.  perm[block * 64 + lane] = the row's place in the WINDOW
// (0 .. 255): a byte still.  A last window of fewer than four blocks orders its rows among the blocks it has.
// What this asks of the readers of the copy: a block holds rows of its whole window, so a row range is scanned in
// whole windows (filter.hip rounds its block range outwards and tests every reported row against the range).
template <int NQ>
__global__ __launch_bounds__(256) void conflict_order_window(const uint4 *__restrict__ src, uint4 *__restrict__ dst,
                                                             uint8_t *__restrict__ perm, long long nblk, int rounds) {
  __shared__ uint32_t cnt_s[16 * 16 * 4];        // [group (16)][quantizer (16)][column] bytes, four columns per word
  __shared__ uint16_t st_s[16 * 16];             // [group][quantizer] fullest column | number of such columns << 8
  __shared__ uint8_t grp_s[256];                 // group of every row
  __shared__ unsigned long long cols_s[256];     // bank columns of every row's 16 entries (one nibble each)
  __shared__ int best_s[4];
  __shared__ int rank_s[4][16];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const long long wb = (long long)blockIdx.x * 4;                       // first block of the window
  const int nrows = (int)min(256ll, (nblk - wb) * 64);                  // rows of the window (whole blocks)
  const bool live = t < nrows;
  uint8_t *cb = reinterpret_cast<uint8_t *>(cnt_s);
  uint4 w = make_uint4(0, 0, 0, 0);
  if (live) w = src[wb * 64 + t];
  const uint32_t ws[4] = {w.x, w.y, w.z, w.w};
  unsigned long long cols = 0;
#pragma unroll
  for (int q = 0; q < 16; q++) cols |= (unsigned long long)((ws[q >> 2] >> (8 * (q & 3))) & 15u) << (4 * q);
  int grp = 4 * wave + lds_group_of_lane(lane);
  for (int e = t; e < 16 * 16 * 4; e += 256) cnt_s[e] = 0;
  grp_s[t] = (uint8_t)grp;
  cols_s[t] = cols;
  __syncthreads();
  if (live) {
#pragma unroll
    for (int q = 0; q < NQ; q++) {
      const int c = (int)(cols >> (4 * q)) & 15;
      atomicAdd(&cnt_s[(grp * 16 + q) * 4 + (c >> 2)], 1u << (8 * (c & 3)));
    }
  }
  __syncthreads();
  auto restat = [&](int g, int q) {
    int M = 0, nM = 0;
#pragma unroll
    for (int c4 = 0; c4 < 4; c4++) {
      const uint32_t v = cnt_s[(g * 16 + q) * 4 + c4];
#pragma unroll
      for (int b = 0; b < 4; b++) {
        const int x = (v >> (8 * b)) & 255;
        if (x > M) { M = x; nM = 1; } else if (x == M) nM++;
      }
    }
    st_s[g * 16 + q] = (uint16_t)(M | (nM << 8));
  };
  if ((t & 15) < NQ) restat(t >> 4, t & 15);
  __syncthreads();
  for (int step = 0; step < rounds * 256; step++) {
    const int r = step & 255;
    if (r >= nrows) continue;                       // (uniform)
    const int a = grp_s[r];
    const unsigned long long ci64 = cols_s[r];
    const int b = grp;
    int key = 1 << 30;
    if (live && b != a) {
      int d = 0, sec = 0;
#pragma unroll
      for (int q = 0; q < NQ; q++) {
        const int ci = (int)(ci64 >> (4 * q)) & 15, cj = (int)(cols >> (4 * q)) & 15;
        if (ci != cj) {
          const int sa = st_s[a * 16 + q], sb = st_s[b * 16 + q];
          const int Ma = sa & 255, na = sa >> 8, Mb = sb & 255, nb = sb >> 8;
          const int ca_i = cb[(a * 16 + q) * 16 + ci], ca_j = cb[(a * 16 + q) * 16 + cj];
          const int cb_i = cb[(b * 16 + q) * 16 + ci], cb_j = cb[(b * 16 + q) * 16 + cj];
          d += max((ca_i == Ma && na == 1) ? Ma - 1 : Ma, ca_j + 1) - Ma;
          d += max((cb_j == Mb && nb == 1) ? Mb - 1 : Mb, cb_i + 1) - Mb;
          sec += (ca_j - ca_i + 1) + (cb_i - cb_j + 1);
        }
      }
      key = d * 2048 + sec;
    }
    // the best exchange of row r: minimum of (key, row) over the workgroup
    int best = ((key + (1 << 20)) << 8) | t;
    if (key == (1 << 30)) best = 0x7fffffff;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) best = min(best, __shfl_xor(best, o));
    if (lane == 0) best_s[wave] = best;
    __syncthreads();
    best = min(min(best_s[0], best_s[1]), min(best_s[2], best_s[3]));
    const int bkey = (best >> 8) - (1 << 20);
    if (!(best == 0x7fffffff || bkey >= 0)) {       // (uniform)
      const int j = best & 255;
      const int gb = grp_s[j];
      const unsigned long long cj64 = cols_s[j];
      if (t < NQ) {
        const int q = t;
        const int ci = (int)(ci64 >> (4 * q)) & 15, cj = (int)(cj64 >> (4 * q)) & 15;
        if (ci != cj) {
          cb[(a * 16 + q) * 16 + ci] -= 1; cb[(a * 16 + q) * 16 + cj] += 1;
          cb[(gb * 16 + q) * 16 + cj] -= 1; cb[(gb * 16 + q) * 16 + ci] += 1;
          restat(a, q);
          restat(gb, q);
        }
      }
      __syncthreads();                              // every thread has read grp_s[r], grp_s[j]
      if (t == r) { grp = gb; grp_s[t] = (uint8_t)gb; }
      if (t == j) { grp = a; grp_s[t] = (uint8_t)a; }
    }
    __syncthreads();
  }
  // places: the rows of group g take the lanes of lane group g % 4 in block g / 4, in row order
  int below = 0;                                    // rows of my group in the waves before mine
  {
    for (int g = 0; g < 16; g++) {
      const unsigned long long mk = __ballot(live && grp == g);
      if (lane == 0) rank_s[wave][g] = __popcll(mk);
    }
    __syncthreads();
    for (int w2 = 0; w2 < wave; w2++) below += rank_s[w2][grp];
  }
  int within = 0;                                   // ... and in my wave, before me
  for (int g = 0; g < 16; g++) {
    const unsigned long long mk = __ballot(live && grp == g);
    if (grp == g) within = __popcll(mk & ((1ull << lane) - 1ull));
  }
  if (live) {
    const long long slot = (wb + (grp >> 2)) * 64 + lds_lane_of_group(grp & 3, below + within);
    dst[slot] = w;
    perm[slot] = (uint8_t)t;
  }
}

}  // namespace

// codes `src` [nblk][64] 16-byte code words -> `dst` (may not alias src), perm [nblk * 64]
void launch_conflict_order(const uint8_t *src, uint8_t *dst, uint8_t *perm, long long nblk, int nq, int rounds,
                           hipStream_t st) {
  if (nblk <= 0) return;
  const dim3 grid((unsigned)ceil_div(nblk, 4LL));
  const auto s4 = reinterpret_cast<const uint4 *>(src);
  const auto d4 = reinterpret_cast<uint4 *>(dst);
  GULON_REQUIRE(nq == FILTER_LDS_QUANTIZERS, "internal: conflict ordering over %d quantizers", nq);
  // (GULON_FILTER_ORDER_WINDOW=0: the ordering inside single blocks, for A/B measurements; the copy's readers are told
  // which one they have through gulon_index::fwindow)
  if (conflict_order_windowed())
    hipLaunchKernelGGL(conflict_order_window<FILTER_LDS_QUANTIZERS>, grid, dim3(256), 0, st, s4, d4, perm, nblk, rounds);
  else
    hipLaunchKernelGGL(conflict_order<FILTER_LDS_QUANTIZERS>, grid, dim3(256), 0, st, s4, d4, perm, nblk, rounds);
  HIP_CHECK(hipGetLastError());
}

bool conflict_order_windowed() {
  static const bool on = [] { const char *e = getenv("GULON_FILTER_ORDER_WINDOW"); return !(e && atoi(e) == 0); }();
  return on;
}

}  // namespace gulon

using namespace gulon;

#ifdef GULON_TEST_HOOKS
GULON_API int32_t gulon_selftest_conflict_order(const uint8_t *codes, int64_t n_blocks, int32_t rounds, uint8_t *codes_out,
                                                uint8_t *place_out) {
  return guarded([&] {
    GULON_REQUIRE(codes != nullptr && codes_out != nullptr && place_out != nullptr && n_blocks >= 0 && rounds >= 0 && rounds <= 8,
                  "bad arguments");
    if (n_blocks == 0) return;
    DevBuf<uint8_t> src, dst((size_t)n_blocks * 1024), place((size_t)n_blocks * 64);
    src.upload(codes, (size_t)n_blocks * 1024);
    launch_conflict_order(src.p, dst.p, place.p, n_blocks, FILTER_LDS_QUANTIZERS, rounds, 0);
    HIP_CHECK(hipDeviceSynchronize());
    HIP_CHECK(hipMemcpy(codes_out, dst.p, (size_t)n_blocks * 1024, hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(place_out, place.p, (size_t)n_blocks * 64, hipMemcpyDeviceToHost));
  });
}
#endif
