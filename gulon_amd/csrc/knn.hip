// Index.exactNearestNeighbours (Index.scala:209-229) with MathUtils.distanceSq
// (MathUtils.scala:85-95): brute-force squared L2, sequential unfused fp32, then the
// same wavefront top-k as the ADC scan.  Used for BASELINE config 1 and for the
// recall ground truth (Tests.scala:89-97).
#include "scan.hpp"

namespace gulon {

constexpr int KNN_QT = 16;       // queries per workgroup
constexpr int KNN_DT = 32;       // dims staged per step
constexpr int KNN_THREADS = 256; // 4 waves, lane = row

// Qt[tile][t][KNN_QT]: queries transposed so one uniform (scalar) load serves a wave
__global__ void transpose_queries(const float *__restrict__ Q, int B, int d, int ntiles, float *__restrict__ Qt) {
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long total = (long long)ntiles * d * KNN_QT;
  if (t >= total) return;
  int u = (int)(t % KNN_QT);
  int dim = (int)((t / KNN_QT) % d);
  int tile = (int)(t / ((long long)KNN_QT * d));
  int q = tile * KNN_QT + u;
  Qt[t] = q < B ? Q[(size_t)q * d + dim] : 0.f;
}

__global__ __launch_bounds__(KNN_THREADS) void knn_kernel(const float *__restrict__ X, int n, int d,
                                                          const float *__restrict__ Qt, int row_from, int row_until,
                                                          int rows_per_chunk, int nchunks, int keff,
                                                          float *__restrict__ part_v, int *__restrict__ part_i,
                                                          const float *__restrict__ lbv, const int *__restrict__ lbi) {
  constexpr int NW = KNN_THREADS / 64;
  __shared__ float xs[KNN_THREADS * (KNN_DT + 1)];
  __shared__ float sv[KNN_QT * NW * 64];
  __shared__ int si[KNN_QT * NW * 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tile = blockIdx.x, chunk = blockIdx.y;
  const float *qt = Qt + (size_t)tile * d * KNN_QT;

  WaveList wl[KNN_QT];
  int cnt[KNN_QT];
#pragma unroll
  for (int q = 0; q < KNN_QT; q++) { wl[q].init(); cnt[q] = 0; }

  // large-K peeling: only entries after (lbq, lbiq) in (distance, row id) order are eligible
  const bool peel = lbv != nullptr;
  float lbq[KNN_QT];
  int lbiq[KNN_QT];
#pragma unroll
  for (int q = 0; q < KNN_QT; q++) {
    lbq[q] = peel ? lbv[tile * KNN_QT + q] : -1.f;
    lbiq[q] = peel ? lbi[tile * KNN_QT + q] : -1;
  }

  const int r0 = (row_from / KNN_THREADS) * KNN_THREADS + chunk * rows_per_chunk;
  const int r1 = min(row_until, r0 + rows_per_chunk);
  for (int base = r0; base < r1; base += KNN_THREADS) {
    float acc[KNN_QT];
#pragma unroll
    for (int q = 0; q < KNN_QT; q++) acc[q] = 0.f;
    for (int d0 = 0; d0 < d; d0 += KNN_DT) {
      __syncthreads();
      // stage rows [base, base+256) x dims [d0, d0+DT)
      for (int e = tid; e < KNN_THREADS * KNN_DT; e += KNN_THREADS) {
        int r = e / KNN_DT, c = e % KNN_DT;
        int row = base + r, dim = d0 + c;
        xs[r * (KNN_DT + 1) + c] = (row < n && dim < d) ? X[(size_t)row * d + dim] : 0.f;
      }
      __syncthreads();
      const int dl = min(KNN_DT, d - d0);
      for (int c = 0; c < dl; c++) {
        float x = xs[tid * (KNN_DT + 1) + c];
        const float *qv = qt + (size_t)(d0 + c) * KNN_QT;
#pragma unroll
        for (int q = 0; q < KNN_QT; q++) {
          float dx = qv[q] - x;  // dx = y(i) - x(i), y = query (MathUtils.scala:90)
          acc[q] += dx * dx;
        }
      }
    }
    const int row = base + tid;
    const bool valid = row >= row_from && row < row_until;
    unsigned long long vmask = __ballot(valid);
#pragma unroll
    for (int q = 0; q < KNN_QT; q++) {
      unsigned long long mk = __ballot(valid && acc[q] < wl[q].tau);
      if (cnt[q] < keff) mk = vmask;
      while (mk) {
        int l = __ffsll((long long)mk) - 1;
        mk &= mk - 1;
        float cv = readlane_f(acc[q], l);
        int cr = base + wave * 64 + l;
        if (peel && !(cv > lbq[q] || (cv == lbq[q] && cr > lbiq[q]))) continue;
        if (cnt[q] < keff || wl[q].accepts(cv, cr)) {
          wl[q].insert(cv, cr, keff, lane);
          if (cnt[q] < keff) cnt[q]++;
        }
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < KNN_QT; q++) {
    sv[(q * NW + wave) * 64 + lane] = wl[q].v;
    si[(q * NW + wave) * 64 + lane] = wl[q].i;
  }
  __syncthreads();
  for (int q = wave; q < KNN_QT; q += NW) {
    WaveList out;
    out.init();
    for (int w2 = 0; w2 < NW; w2++) {
      for (int e = 0; e < keff; e++) {
        float cv = sv[(q * NW + w2) * 64 + e];
        int cr = si[(q * NW + w2) * 64 + e];
        if (cr == INT_MAX) break;
        if (out.accepts(cv, cr)) out.insert(cv, cr, keff, lane);
      }
    }
    if (lane < keff) {
      size_t o = ((size_t)(tile * KNN_QT + q) * nchunks + chunk) * keff + lane;
      part_v[o] = out.v;
      part_i[o] = out.i;
    }
  }
}

}  // namespace gulon

using namespace gulon;

GULON_API int32_t gulon_exact_knn(const gulon_dataset *ds, int32_t from, int32_t until, const float *queries,
                                  int32_t b, int32_t k_nn, int32_t *out_idx, float *out_dist, int32_t *out_count,
                                  int32_t *out_flags) {
  return guarded([&] {
    GULON_REQUIRE(ds != nullptr, "dataset is null");
    GULON_REQUIRE(from <= until, "invalid range: expected from=%d <= until=%d", from, until);          // Index.scala:213
    GULON_REQUIRE(until <= ds->n, "invalid range: expected until=%d <= vectors.length=%d", until, ds->n);  // :214
    GULON_REQUIRE(from >= 0 && b >= 0 && k_nn >= 0, "bad arguments");
    GULON_UNSUPPORTED(k_nn > GULON_MAX_K_PEELED, "k_nn = %d > %d", k_nn, GULON_MAX_K_PEELED);
    if (b == 0) return;
    const bool peeled = k_nn > GULON_MAX_K;
    const int K = k_nn, keff = peeled ? 64 : K + 1, d = ds->d;
    size_t bk = (size_t)b * K;
    if (K == 0 || from == until) {
      for (size_t i = 0; i < bk; i++) { out_idx[i] = -1; out_dist[i] = INFINITY; }
      for (int q = 0; q < b; q++) { if (out_count) out_count[q] = 0; if (out_flags) out_flags[q] = 0; }
      return;
    }
    const int ntiles = ceil_div(b, KNN_QT);
    const int rbeg = (from / KNN_THREADS) * KNN_THREADS;
    const int groups = ceil_div(until - rbeg, KNN_THREADS);
    int want = ceil_div(2048, ntiles);
    int nchunks = want < groups ? want : groups;
    int groups_per_chunk = ceil_div(groups, nchunks);
    nchunks = ceil_div(groups, groups_per_chunk);
    const int rows_per_chunk = groups_per_chunk * KNN_THREADS;
    const int Bp = ntiles * KNN_QT;

    DevBuf<float> dq, dqt((size_t)ntiles * d * KNN_QT), pv((size_t)Bp * nchunks * keff), od(bk);
    DevBuf<int> pi((size_t)Bp * nchunks * keff), oi(bk), oc(b), of(b);
    dq.upload(queries, (size_t)b * d);
    long long tq = (long long)ntiles * d * KNN_QT;
    hipLaunchKernelGGL(transpose_queries, dim3(ceil_div(tq, 256)), dim3(256), 0, 0, dq.p, b, d, ntiles, dqt.p);
    if (peeled) {
      const int rounds = ceil_div(K + 1, 64), cap = rounds * 64;
      DevBuf<float> acc_v((size_t)b * cap), tv((size_t)b * 64), lbv(Bp);
      DevBuf<int> acc_i((size_t)b * cap), ti((size_t)b * 64), lbi(Bp);
      HIP_CHECK(hipMemsetD32Async((hipDeviceptr_t)lbv.p, 0xBF800000 /* -1.0f */, (size_t)Bp, 0));
      HIP_CHECK(hipMemsetD32Async((hipDeviceptr_t)lbi.p, 0xFFFFFFFF, (size_t)Bp, 0));
      for (int r = 0; r < rounds; r++) {
        hipLaunchKernelGGL(knn_kernel, dim3(ntiles, nchunks), dim3(KNN_THREADS), 0, 0, ds->x.p, ds->n, d, dqt.p, from,
                           until, rows_per_chunk, nchunks, 64, pv.p, pi.p, lbv.p, lbi.p);
        HIP_CHECK(hipGetLastError());
        launch_merge(false, pv.p, pi.p, nchunks, 64LL, (long long)nchunks * 64, b, 63, nullptr, nullptr, nullptr,
                     nullptr, tv.p, ti.p, nullptr);
        launch_peel_update(tv.p, ti.p, b, r, cap, acc_v.p, acc_i.p, lbv.p, lbi.p, nullptr);
      }
      launch_peel_finalize(acc_v.p, acc_i.p, b, cap, K, oi.p, od.p, oc.p, of.p, nullptr);
      HIP_CHECK(hipDeviceSynchronize());
    } else {
      hipLaunchKernelGGL(knn_kernel, dim3(ntiles, nchunks), dim3(KNN_THREADS), 0, 0, ds->x.p, ds->n, d, dqt.p, from,
                         until, rows_per_chunk, nchunks, keff, pv.p, pi.p, (const float *)nullptr,
                         (const int *)nullptr);
      HIP_CHECK(hipGetLastError());
      launch_merge(true, pv.p, pi.p, nchunks, (long long)keff, (long long)nchunks * keff, b, K, oi.p, od.p, oc.p,
                   of.p, nullptr, nullptr, nullptr);
    }
    oi.download(out_idx, bk); od.download(out_dist, bk);
    if (out_count) oc.download(out_count, b);
    if (out_flags) of.download(out_flags, b);
    HIP_CHECK(hipDeviceSynchronize());
  });
}
