// Internal declarations for the KMeans kernels' host drivers.
#pragma once
#include <cfloat>
#include <vector>

#include "common.hpp"

namespace gulon {

// MFMA-ready copy of the column slice X[:, from:from+s]: [ceil(n/64)*2][T][64] floats,
// element (tile, t, l) = X[tile*32 + (l&31)][from + 2t + (l>>5)]  (kmeans_mfma.hip)
// split = true (s <= 16, k <= 576): every element as THREE bf16 pieces x = x1 + x2 + x3 (exact), in the operand order of
// v_mfma_f32_32x32x16_bf16: xq as uint4 [ceil(n/64)*2][3][64], lane l = row tile*32 + (l & 31), elements 8 (l >> 5) .. +7
// of the slice; xn[row] = |x|^2 (sequential fp32), the scale of the error band.
struct PackedSlice {
  DevBuf<float> xq;
  DevBuf<float> xn;
  int n = 0, from = 0, s = 0, T = 0;
  bool split = false;
  int words = 0;   // split: 0 = three pieces per lane and tile, else that many compact operand words (s <= 13: kmeans_mfma.hip)
};

// one problem of a batched KMeans.fromAssignment (kmeans.hip)
struct UpdDesc {
  const int *assign;
  unsigned *hist, *gtot, *count, *start;
  float *xb;        // the slices regrouped by cluster, stable: bucket position q holds a row at xb + q*sp (sp = s rounded up to even)
  float *cout;
  const float *x;   // row r's slice = x + r*ld + from
  const float *rcp; // rcp[i] = RN(1 / (i + 1)): the running mean's divisors (update_chains)
  int *corder;      // clusters by descending size (the longest chains are dispatched first); null: cluster order
  int ld, from, s;
};

// the same descriptor as the device uses it: every pointer typed as global memory (see as_global, common.hpp)
struct UpdDescG {
  gptr<const int> assign;
  gptr<unsigned> hist, gtot, count, start;
  gptr<float> xb, cout;
  gptr<const float> x;
  gptr<int> corder;
  int ld, from, s;
};
__device__ __forceinline__ UpdDescG load_desc(const UpdDesc *__restrict__ descs, int i) {
  const UpdDesc D = descs[i];
  return UpdDescG{as_global(D.assign), as_global(D.hist), as_global(D.gtot), as_global(D.count), as_global(D.start),
                  as_global(D.xb), as_global(D.cout), as_global(D.x), as_global(D.corder), D.ld, D.from, D.s};
}

struct KmeansWorkspace {
  struct HostWords { unsigned flagged; unsigned pad; unsigned long long total; };
  HostWords *host = nullptr;           // pinned: counters copied back asynchronously
  KmeansWorkspace() = default;
  KmeansWorkspace(const KmeansWorkspace &) = delete;
  KmeansWorkspace &operator=(const KmeansWorkspace &) = delete;
  ~KmeansWorkspace();
  DevBuf<float> cpad, off;
  DevBuf<unsigned> ties, local, block_tot;
  DevBuf<unsigned long long> tie_total, block_off;
  DevBuf<unsigned> hist, gtot, count, start, mismatch;
  DevBuf<float> xb;                    // row slices grouped by cluster, stable (row order inside a cluster)
  DevBuf<int> corder;                  // clusters by descending size
  DevBuf<UpdDesc> descs;
  // MFMA filter
  DevBuf<float> apack, offp;
  DevBuf<unsigned> cmax2, flag_count, flag_ties, tie_count;
  DevBuf<int> flag_rows, tie_rows;
  DevBuf<unsigned long long> tie_pos;  // stream positions of the drawing rows (sparse tie replay)
  unsigned long long last_draws = 0;   // RNG draws made by the last assign (0 = no exact ties)
  unsigned last_flagged = 0;           // rows the MFMA filter sent to the exact kernel
  void ensure(int n, int k, int s);
};

// stage times of kmeans_train_batch, accumulated over its iterations while enabled (gulon_kmeans_trace)
struct TrainTrace {
  bool on = false;
  int iterations = 0;
  double update_ms = 0, assign_ms = 0, recheck_ms = 0, converge_ms = 0;
  double mfma_flops = 0, update_bytes = 0;
  unsigned long long rows_rechecked = 0, rows_total = 0;
};
TrainTrace &train_trace();
bool mfma_assign_supported(int s, int k);
void pack_slice(const float *dX, int n, int ld, int from, int s, int k, PackedSlice &ps, hipStream_t st);
void assign_mfma_filter(KmeansWorkspace &ws, const PackedSlice &ps, const float *dC, int k, int *d_assign,
                        hipStream_t st);

// One assign call, split into enqueue stages (kmeans.hip): stage1 -> sync -> stage2 -> sync -> stage3.
struct AssignJob {
  KmeansWorkspace *ws = nullptr;
  const float *dX = nullptr;
  int n = 0, ld = 0, from = 0, s = 0;
  const float *dC = nullptr;
  int k = 0, rng_batch = 0;
  int *d_assign = nullptr;
  hipStream_t st = nullptr;
  const PackedSlice *ps = nullptr;
  bool filtered = false, done = false;
  const int *rows = nullptr;
  int nrows = 0;
};
void assign_stage1(AssignJob &j);
void assign_stage2(AssignJob &j);
void assign_stage3(AssignJob &j);

// KMeans.assign / parAssign.  `ps` (nullable): packed copy of the slice => MFMA filter +
// exact re-check of the flagged rows; without it every row takes the exact VALU kernel.
void kmeans_assign_dev(KmeansWorkspace &ws, const float *dX, int n, int ld, int from, int s, const float *dC, int k,
                       int rng_batch, int *d_assign, hipStream_t st, const PackedSlice *ps = nullptr);
#ifdef __HIPCC__
// RN(1 / n) for an integer n < 2^24 whose significand is not all ones: the hardware reciprocal (1 ulp) and one
// fma-corrected Newton step (Markstein).  Checked against 1.0f / n for EVERY such n by gulon_selftest_mean_division.
__device__ __forceinline__ float rcp_rn_int(float nf) {
  const float y0 = __builtin_amdgcn_rcpf(nf);
  const float e = __builtin_fmaf(-nf, y0, 1.0f);
  return __builtin_fmaf(e, y0, y0);
}
#endif

// kmeans_stream.hip: KMeans.fromAssignment from a pair-major copy of the slices, read once in row order
struct StreamDesc {
  const int *assign;        // [n]
  const float *xp;          // pair-major slice: [pairs][ns] float2, pair j = dimensions (2j, 2j + 1)
  unsigned short *ord;      // scratch: [chunks][STREAM_CH] local row indices, stable by cluster
  unsigned short *coff;     // scratch: [chunks][k + 1] first position of every cluster in its chunk
  float *cout;              // k x s centroids out
  const int *wild;          // nonzero: the slice holds values the corrected quotient is not trusted with (stream_pack_pairs)
  int s, pad;
  long long ns;             // rows per pair in xp = stream_padded_rows(n): whole chunks, the padding zeroed
};
bool stream_update_supported(int n, int k, int s);
size_t stream_order_words(int n, int k, size_t *coff_words);
long long stream_padded_rows(int n);   // rows per pair of the pair-major copy (whole chunks, zero padded)
void stream_pack_pairs(const float *xs, int n, int s, float *xp /* [pairs][ns] float2 */, int *wild /* zeroed; set if any |x| >= 2^99, infinite or NaN */,
                       hipStream_t st);
void kmeans_stream_order(const std::vector<StreamDesc> &descs, StreamDesc *d_descs, int n, int k, hipStream_t st);
void kmeans_stream_chains(const std::vector<StreamDesc> &descs, StreamDesc *d_descs, int n, int k, hipStream_t st);

// kmeans_fused.hip: KMeans.fromAssignment without the regrouped copy (false: the shape keeps the bucketed path)
bool kmeans_update_fused(const std::vector<UpdDesc> &descs, UpdDesc *d_descs, int n, int k, hipStream_t st);
void kmeans_update_dev(KmeansWorkspace &ws, const float *dX, int n, int ld, int from, int s, int k,
                       const int *d_assign, float *dC, hipStream_t st);

}  // namespace gulon
