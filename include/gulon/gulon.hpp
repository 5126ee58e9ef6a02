// gulon.hpp -- C++ host-side mirror of Gulon's Scala API for the accelerated path, written
// over the C ABI (gulon_hip.h) only.  The reference is JVM code and no JVM toolchain exists
// in the build image, so this header (plus the Python mirror used by the tests) stands where
// the Scala bodies would delegate through JNI.  Same names, argument
// meaning and error behaviour as the reference:
//   require(...) failures        -> std::invalid_argument   (IllegalArgumentException)
//   IllegalStateException        -> std::logic_error
//   everything else              -> std::runtime_error
// Paths cited are relative to /root/reference/core/src/main/scala/net/tixxit/gulon/.
#pragma once

#include <algorithm>
#include <cstdint>
#include <functional>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../gulon_hip.h"

namespace gulon {

inline void check(int32_t rc) {
  if (rc == GULON_OK) return;
  std::string msg = gulon_last_error();
  if (rc == GULON_ERR_INVALID_ARGUMENT) throw std::invalid_argument("requirement failed: " + msg);
  if (rc == GULON_ERR_ILLEGAL_STATE) throw std::logic_error(msg);
  throw std::runtime_error(msg);
}

// Matrix (Matrix.scala:3), flat row-major, resident in HBM.
class Matrix {
 public:
  Matrix(const float *data, int rows, int cols) : rows_(rows), cols_(cols) {
    check(gulon_dataset_create(data, rows, cols, &h_));
  }
  // WordVectors.Grouped.residuals (WordVectors.scala:118-138): out[i] = m[perm[i]] - centroids[groupOf[i]]
  static Matrix groupResiduals(const Matrix &m, const std::vector<int32_t> &perm, const std::vector<int32_t> &groupOf,
                               const std::vector<float> &centroids, int groups) {
    Matrix r;
    r.rows_ = m.rows_; r.cols_ = m.cols_;
    check(gulon_dataset_group_residuals(m.h_, perm.data(), groupOf.data(), centroids.data(), groups, &r.h_));
    return r;
  }
  static Matrix synthetic(int rows, int cols, int kind, uint64_t seed, int ncentres) {
    Matrix m;
    m.rows_ = rows; m.cols_ = cols;
    check(gulon_dataset_create_synth(rows, cols, kind, seed, ncentres, &m.h_));
    return m;
  }
  Matrix(Matrix &&o) noexcept : h_(o.h_), rows_(o.rows_), cols_(o.cols_) { o.h_ = nullptr; }
  Matrix(const Matrix &) = delete;
  Matrix &operator=(const Matrix &) = delete;
  ~Matrix() { if (h_) gulon_dataset_destroy(h_); }
  int rows() const { return rows_; }
  int cols() const { return cols_; }
  const gulon_dataset *handle() const { return h_; }
  std::vector<float> getRows(const std::vector<int32_t> &rows) const {
    std::vector<float> out(rows.size() * (size_t)cols_);
    check(gulon_dataset_get_rows(h_, rows.data(), (int32_t)rows.size(), out.data()));
    return out;
  }

 private:
  Matrix() = default;
  gulon_dataset *h_ = nullptr;
  int rows_ = 0, cols_ = 0;
};

// Vectors (Vectors.scala): (matrix, from, until) column-slice view.
struct Vectors {
  const Matrix *matrix;
  int from, until;
  int dimension() const { return until - from; }
  int size() const { return matrix->rows(); }
  // Vectors.subvectors (Vectors.scala:84-104)
  static std::vector<Vectors> subvectors(const Matrix &m, int numSubvectors) {
    std::vector<int32_t> fr(numSubvectors), un(numSubvectors);
    check(gulon_subvectors(m.cols(), numSubvectors, fr.data(), un.data()));
    std::vector<Vectors> out;
    for (int i = 0; i < numSubvectors; i++) out.push_back(Vectors{&m, fr[i], un[i]});
    return out;
  }
};

// KMeans (KMeans.scala)
class KMeans {
 public:
  struct ProgressReport { int numIterations, maxIterations, stepCount; float stepMean, stepS; bool converged; };
  struct Config {
    int numClusters, maxIterations, seed = 0;
    std::function<void(const ProgressReport &)> report;
  };
  KMeans(int dimension, std::vector<float> centroids) : dimension(dimension), centroids(std::move(centroids)) {}
  int dimension;
  std::vector<float> centroids;   // k x dimension
  int k() const { return dimension ? (int)(centroids.size() / dimension) : 0; }

  std::vector<int32_t> assign(const Vectors &v) const { return assignImpl(v, 0); }            // :18-22,70-98
  std::vector<int32_t> parAssign(const Vectors &v) const { return assignImpl(v, 25000); }     // :57-68
  KMeans iterate(const Vectors &v, int iters) const {                                           // :100-106
    std::vector<float> out(centroids.size());
    check(gulon_kmeans_iterate(v.matrix->handle(), v.from, v.dimension(), centroids.data(), k(), iters, out.data()));
    return KMeans(dimension, out);
  }
  static KMeans init(int k, const Vectors &v, int seed = 0) {                                   // :188-196
    std::vector<float> c((size_t)k * v.dimension());
    check(gulon_kmeans_init(v.matrix->handle(), v.from, v.dimension(), k, seed, c.data(), nullptr));
    return KMeans(v.dimension(), c);
  }
  static KMeans fromAssignment(int k, int dimension, const Vectors &v, const std::vector<int32_t> &a) {   // :198-226
    std::vector<float> c((size_t)k * v.dimension());
    check(gulon_kmeans_update(v.matrix->handle(), v.from, v.dimension(), k, a.data(), c.data()));
    return KMeans(dimension, c);
  }
  static KMeans computeClusters(const Vectors &v, const Config &cfg) {                           // :134-157
    std::vector<float> c((size_t)cfg.numClusters * v.dimension());
    std::vector<gulon_kmeans_report> reps(cfg.maxIterations + 3);
    int32_t nrep = 0;
    check(gulon_kmeans_train(v.matrix->handle(), v.from, v.dimension(), cfg.numClusters, cfg.maxIterations, cfg.seed,
                             c.data(), reps.data(), (int32_t)reps.size(), &nrep));
    if (cfg.report)
      for (int i = 0; i < nrep; i++)
        cfg.report(ProgressReport{reps[i].num_iterations, cfg.maxIterations, reps[i].step_count, reps[i].step_mean,
                                  reps[i].step_s, reps[i].converged != 0});
    return KMeans(v.dimension(), c);
  }

 private:
  std::vector<int32_t> assignImpl(const Vectors &v, int rngBatch) const {
    std::vector<int32_t> a(v.size(), 0);
    check(gulon_kmeans_assign(v.matrix->handle(), v.from, v.dimension(), centroids.data(), k(), rngBatch, a.data()));
    return a;
  }
};

// EncodedMatrix (EncodedMatrix.scala): m packed code arrays, each covering all rows.
struct EncodedMatrix {
  int width = 8, length = 0, bytesPerCode = 0;
  std::vector<uint8_t> packed;   // [m][bytesPerCode]
};

// ProductQuantizer (ProductQuantizer.scala)
class ProductQuantizer {
 public:
  struct Config { int numClusters, numQuantizers, maxIterations; };
  int numClusters = 0, numQuantizers = 0, dimension = 0;
  std::vector<float> centroids;   // k*d, quantizer j at k*from_j

  static ProductQuantizer apply(const Matrix &vectors, const Config &cfg) {                       // :150-153
    ProductQuantizer pq;
    pq.numClusters = cfg.numClusters; pq.numQuantizers = cfg.numQuantizers; pq.dimension = vectors.cols();
    pq.centroids.resize((size_t)cfg.numClusters * vectors.cols());
    check(gulon_pq_train(vectors.handle(), cfg.numQuantizers, cfg.numClusters, cfg.maxIterations, pq.centroids.data(),
                         nullptr, 0, nullptr));
    return pq;
  }
  int coderWidth() const {                                                                        // :11-16
    int32_t w = 0;
    check(gulon_coder_width(numClusters, &w));
    return w;
  }
  EncodedMatrix encode(const Matrix &vectors) const {                                             // :25-35
    EncodedMatrix em;
    em.width = coderWidth(); em.length = vectors.rows();
    int32_t b = 0;
    check(gulon_coder_bytes(em.width, em.length, &b));
    em.bytesPerCode = b;
    em.packed.resize((size_t)numQuantizers * b + 1);
    check(gulon_pq_encode(vectors.handle(), numQuantizers, numClusters, centroids.data(), em.packed.data()));
    return em;
  }
};

// Index (Index.scala)
struct Result {                       // Index.Result with int row ids, ascending squared-L2 distances
  std::vector<int32_t> rows;
  std::vector<float> distances;
  int flags = 0;
};

class PQIndex {                       // Index.scala:385-441
 public:
  PQIndex(const ProductQuantizer &pq, const EncodedMatrix &data, int rowBase = 0)
      : dimension_(pq.dimension), length_(data.length) {
    check(gulon_index_create(data.packed.data(), data.length, pq.dimension, pq.numQuantizers, pq.numClusters,
                             pq.centroids.data(), rowBase, &h_));
  }
  PQIndex(const PQIndex &) = delete;
  PQIndex &operator=(const PQIndex &) = delete;
  ~PQIndex() { if (h_) gulon_index_destroy(h_); }
  int dimension() const { return dimension_; }
  int length() const { return length_; }
  // batchQuery(k, vectors, from, until) + Result.fromHeap (:417-440, :83-94)
  std::vector<Result> batchQuery(int k, const std::vector<float> &queries, int from, int until) const {
    const int b = dimension_ ? (int)(queries.size() / dimension_) : 0;
    std::vector<int32_t> idx((size_t)b * k + 1), cnt(b + 1), flg(b + 1);
    std::vector<float> dist((size_t)b * k + 1);
    check(gulon_index_batch_query(h_, queries.data(), b, k, from, until, idx.data(), dist.data(), cnt.data(),
                                  flg.data()));
    std::vector<Result> out(b);
    for (int q = 0; q < b; q++) {
      out[q].rows.assign(idx.begin() + (size_t)q * k, idx.begin() + (size_t)q * k + cnt[q]);
      out[q].distances.assign(dist.begin() + (size_t)q * k, dist.begin() + (size_t)q * k + cnt[q]);
      out[q].flags = flg[q];
    }
    return out;
  }
  std::vector<Result> batchQuery(int k, const std::vector<float> &queries) const {
    return batchQuery(k, queries, 0, length_);
  }

 private:
  gulon_index *h_ = nullptr;
  int dimension_, length_;
};

// WordVectors.Grouped without the keys (WordVectors.scala:24-58,118-138): rows regrouped by coarse
// cluster, the centroids of the non-empty clusters, the group offsets and the residual matrix.
struct GroupedVectors {
  std::vector<int32_t> perm;          // grouped position -> original row
  std::vector<float> centroids;       // [g][d]
  std::vector<int32_t> offsets;       // [g-1]
  std::unique_ptr<Matrix> residuals;  // grouped row - its group's centroid, resident in HBM
  int groups() const { return (int)offsets.size() + 1; }
};

// WordVectors.grouped: parAssign, stable ordering by cluster (rows keep their order inside a group).
inline GroupedVectors group(const Matrix &vectors, const KMeans &clustering) {
  if (clustering.k() <= 0) throw std::invalid_argument("requirement failed: must have at least 1 cluster");
  const int n = vectors.rows(), d = vectors.cols();
  const std::vector<int32_t> a = clustering.parAssign(Vectors{&vectors, 0, d});
  GroupedVectors gv;
  gv.perm.resize(n);
  for (int i = 0; i < n; i++) gv.perm[i] = i;
  std::stable_sort(gv.perm.begin(), gv.perm.end(), [&](int32_t x, int32_t y) { return a[x] < a[y]; });
  std::vector<int32_t> groupOf(n);
  int g = 0;
  for (int i = 0; i < n; i++) {
    const int c = a[gv.perm[i]];
    if (i == 0 || c != a[gv.perm[i - 1]]) {
      if (i > 0) gv.offsets.push_back(i);
      gv.centroids.insert(gv.centroids.end(), clustering.centroids.begin() + (size_t)c * d,
                          clustering.centroids.begin() + (size_t)(c + 1) * d);
      g++;
    }
    groupOf[i] = g - 1;
  }
  if (n > 0) gv.residuals.reset(new Matrix(Matrix::groupResiduals(vectors, gv.perm, groupOf, gv.centroids, g)));
  return gv;
}

class GroupedIndex {                  // Index.scala:231-308 over row ids (grouped positions)
 public:
  enum class Strategy { LimitGroups = 0, LimitVectors = 1 };
  GroupedIndex(const ProductQuantizer &residualsQuantizer, const EncodedMatrix &encodedResiduals,
               const std::vector<float> &centroids, const std::vector<int32_t> &offsets, Strategy strategy, int limit)
      : dimension_(residualsQuantizer.dimension), strategy_(strategy), limit_(limit) {
    const int g = (int)offsets.size() + 1;
    if ((size_t)g * dimension_ != centroids.size()) throw std::logic_error("centroids.length != offsets.length + 1");
    check(gulon_grouped_index_create(encodedResiduals.packed.data(), encodedResiduals.length, dimension_,
                                     residualsQuantizer.numQuantizers, residualsQuantizer.numClusters,
                                     residualsQuantizer.centroids.data(), centroids.data(),
                                     offsets.empty() ? nullptr : offsets.data(), g, &h_));
  }
  GroupedIndex(const GroupedIndex &) = delete;
  GroupedIndex &operator=(const GroupedIndex &) = delete;
  ~GroupedIndex() { if (h_) gulon_grouped_index_destroy(h_); }
  // GroupedIndex.batchQuery (:254-282): searchSpace, per-group PQIndex.query on the residual, heap.merge
  std::vector<Result> batchQuery(int k, const std::vector<float> &queries) const {
    const int b = dimension_ ? (int)(queries.size() / dimension_) : 0;
    std::vector<int32_t> idx((size_t)b * k + 1), cnt(b + 1);
    std::vector<float> dist((size_t)b * k + 1);
    check(gulon_grouped_index_batch_query(h_, queries.data(), b, k, (int32_t)strategy_, limit_, idx.data(),
                                          dist.data(), cnt.data()));
    std::vector<Result> out(b);
    for (int q = 0; q < b; q++) {
      out[q].rows.assign(idx.begin() + (size_t)q * k, idx.begin() + (size_t)q * k + cnt[q]);
      out[q].distances.assign(dist.begin() + (size_t)q * k, dist.begin() + (size_t)q * k + cnt[q]);
    }
    return out;
  }

 private:
  gulon_grouped_index *h_ = nullptr;
  int dimension_;
  Strategy strategy_;
  int limit_;
};

namespace Index {
// Index.grouped (Index.scala:133-147): the quantizer is one on the residuals.
inline std::unique_ptr<GroupedIndex> grouped(const GroupedVectors &gv, const ProductQuantizer &residualsQuantizer,
                                             GroupedIndex::Strategy strategy, int limit) {
  return std::unique_ptr<GroupedIndex>(new GroupedIndex(residualsQuantizer, residualsQuantizer.encode(*gv.residuals),
                                                        gv.centroids, gv.offsets, strategy, limit));
}
// Index.sorted (Index.scala:107-114): encode, then wrap.
inline std::unique_ptr<PQIndex> sorted(const Matrix &vectors, const ProductQuantizer &pq) {
  return std::unique_ptr<PQIndex>(new PQIndex(pq, pq.encode(vectors)));
}
// Index.exactNearestNeighbours (Index.scala:209-229) for a batch of queries.
inline std::vector<Result> exactNearestNeighbours(const Matrix &vectors, int from, int until,
                                                  const std::vector<float> &queries, int k) {
  const int b = (int)(queries.size() / vectors.cols());
  std::vector<int32_t> idx((size_t)b * k + 1), cnt(b + 1), flg(b + 1);
  std::vector<float> dist((size_t)b * k + 1);
  check(gulon_exact_knn(vectors.handle(), from, until, queries.data(), b, k, idx.data(), dist.data(), cnt.data(),
                        flg.data()));
  std::vector<Result> out(b);
  for (int q = 0; q < b; q++) {
    out[q].rows.assign(idx.begin() + (size_t)q * k, idx.begin() + (size_t)q * k + cnt[q]);
    out[q].distances.assign(dist.begin() + (size_t)q * k, dist.begin() + (size_t)q * k + cnt[q]);
    out[q].flags = flg[q];
  }
  return out;
}
}  // namespace Index

}  // namespace gulon
