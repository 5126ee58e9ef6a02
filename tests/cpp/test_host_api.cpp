// C++ host-API parity test: gulon.hpp (over the C ABI) vs the CPU oracle, bit for bit.
// Reads like the reference's specs: build a ProductQuantizer, Index.sorted, batchQuery, and
// check the `require` behaviour.  Needs a GPU; run by tests/test_cpp_host.py (-m gpu).
#include <cstdio>
#include <cstring>
#include <vector>

#include "gulon/gulon.hpp"

extern "C" {
void go_synth_fill(float *X, int64_t row0, int64_t nrows, int32_t d, int32_t kind, uint64_t seed, int32_t ncentres);
void go_pq_train(const float *X, int32_t n, int32_t d, int32_t m, int32_t k, int32_t max_iterations, float *cents,
                 int32_t *iters_out, int32_t *converged_out);
void go_pq_encode(const float *X, int32_t n, int32_t d, int32_t m, int32_t k, const float *cents, int32_t *idx_out);
int32_t go_pq_batch_query(const int32_t *idx, int32_t n, int32_t d, int32_t m, int32_t k, const float *cents,
                          const float *Q, int32_t B, int32_t K, int32_t from_row, int32_t until_row,
                          int32_t *out_idx, float *out_dist, int32_t *out_count);
void go_kmeans_assign(const float *X, int32_t n, int32_t ld, int32_t from, int32_t s, const float *C, int32_t k,
                      int32_t rng_batch, int32_t *assignments);
int32_t go_grouped_query(const int32_t *idx, int32_t n, int32_t d, int32_t m, int32_t k, const float *pq_cents,
                         const float *gcent, const int32_t *offsets, int32_t g, const float *Q, int32_t B, int32_t K,
                         int32_t strategy, int32_t limit, int32_t *out_idx, float *out_dist, int32_t *out_count);
}

static int fails = 0;
#define EXPECT(cond)                                                         \
  do {                                                                       \
    if (!(cond)) { fprintf(stderr, "FAIL %s:%d %s\n", __FILE__, __LINE__, #cond); fails++; } \
  } while (0)

int main() {
  const int n = 20000, d = 32, m = 4, k = 256, K = 10, B = 8, iters = 3;
  std::vector<float> X((size_t)n * d);
  go_synth_fill(X.data(), 0, n, d, 1, 42, 20);

  gulon::Matrix data = gulon::Matrix::synthetic(n, d, 1, 42, 20);
  std::vector<int32_t> probe = {0, 1, n - 1};
  std::vector<float> got = data.getRows(probe);
  EXPECT(memcmp(got.data(), X.data(), sizeof(float) * d * 2) == 0);   // device generator == oracle generator

  // ProductQuantizer.apply + encode
  gulon::ProductQuantizer pq = gulon::ProductQuantizer::apply(data, {k, m, iters});
  std::vector<float> cents((size_t)k * d);
  go_pq_train(X.data(), n, d, m, k, iters, cents.data(), nullptr, nullptr);
  EXPECT(memcmp(pq.centroids.data(), cents.data(), sizeof(float) * cents.size()) == 0);
  gulon::EncodedMatrix em = pq.encode(data);
  std::vector<int32_t> idx((size_t)m * n);
  go_pq_encode(X.data(), n, d, m, k, cents.data(), idx.data());
  bool codes_ok = em.width == 8 && em.bytesPerCode == n;
  for (size_t i = 0; codes_ok && i < idx.size(); i++) codes_ok = em.packed[i] == (uint8_t)idx[i];
  EXPECT(codes_ok);

  // KMeans.assign on one sub-quantizer
  auto subs = gulon::Vectors::subvectors(data, m);
  gulon::KMeans km(subs[2].dimension(), std::vector<float>(pq.centroids.begin() + (size_t)k * subs[2].from,
                                                          pq.centroids.begin() + (size_t)k * subs[2].until));
  std::vector<int32_t> a = km.parAssign(subs[2]), ea(n, 0);
  go_kmeans_assign(X.data(), n, d, subs[2].from, subs[2].dimension(), km.centroids.data(), k, 25000, ea.data());
  EXPECT(a == ea);

  // Index.sorted + batchQuery
  auto index = gulon::Index::sorted(data, pq);
  std::vector<float> Q(X.begin(), X.begin() + (size_t)B * d);
  auto res = index->batchQuery(K, Q);
  std::vector<int32_t> oi((size_t)B * K), oc(B);
  std::vector<float> od((size_t)B * K);
  go_pq_batch_query(idx.data(), n, d, m, k, cents.data(), Q.data(), B, K, 0, n, oi.data(), od.data(), oc.data());
  for (int q = 0; q < B; q++) {
    EXPECT((int)res[q].rows.size() == oc[q]);
    EXPECT(memcmp(res[q].distances.data(), od.data() + (size_t)q * K, sizeof(float) * oc[q]) == 0);
    EXPECT(memcmp(res[q].rows.data(), oi.data() + (size_t)q * K, sizeof(int32_t) * oc[q]) == 0);
  }

  // require(from <= until) / require(until <= length)  (Index.scala:418-419)
  bool threw = false;
  try { index->batchQuery(K, Q, 10, 5); } catch (const std::invalid_argument &) { threw = true; }
  EXPECT(threw);
  threw = false;
  try { index->batchQuery(K, Q, 0, n + 1); } catch (const std::invalid_argument &) { threw = true; }
  EXPECT(threw);

  // GroupedIndex: coarse KMeans -> group -> residual PQ -> Index.grouped -> batchQuery (Index.scala:231-308)
  {
    gulon::KMeans::Config cc;
    cc.numClusters = 9; cc.maxIterations = iters;
    gulon::KMeans coarse = gulon::KMeans::computeClusters(gulon::Vectors{&data, 0, d}, cc);
    gulon::GroupedVectors gv = gulon::group(data, coarse);
    EXPECT((int)gv.perm.size() == n && gv.groups() >= 1);
    gulon::ProductQuantizer rpq = gulon::ProductQuantizer::apply(*gv.residuals, {k, m, iters});
    auto gindex = gulon::Index::grouped(gv, rpq, gulon::GroupedIndex::Strategy::LimitGroups, 3);
    auto gres = gindex->batchQuery(K, Q);
    gulon::EncodedMatrix rem = rpq.encode(*gv.residuals);
    std::vector<int32_t> ridx((size_t)m * n);
    for (size_t i = 0; i < ridx.size(); i++) ridx[i] = rem.packed[i];
    std::vector<int32_t> gi((size_t)B * K), gc(B);
    std::vector<float> gd((size_t)B * K);
    std::vector<int32_t> off = gv.offsets;
    if (off.empty()) off.push_back(0);
    EXPECT(go_grouped_query(ridx.data(), n, d, m, k, rpq.centroids.data(), gv.centroids.data(), off.data(),
                            gv.groups(), Q.data(), B, K, 0, 3, gi.data(), gd.data(), gc.data()) == 0);
    for (int q = 0; q < B; q++) {
      EXPECT((int)gres[q].rows.size() == gc[q]);
      EXPECT(memcmp(gres[q].distances.data(), gd.data() + (size_t)q * K, sizeof(float) * gc[q]) == 0);
      EXPECT(memcmp(gres[q].rows.data(), gi.data() + (size_t)q * K, sizeof(int32_t) * gc[q]) == 0);
    }
  }

  if (fails == 0) printf("cpp host api OK\n");
  return fails ? 1 : 0;
}
