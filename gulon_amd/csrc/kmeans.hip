// KMeans / ProductQuantizer entry points (KMeans.scala, ProductQuantizer.scala).
#include "common.hpp"

using namespace gulon;

#define NOT_YET(name) \
  return guarded([&] { GULON_UNSUPPORTED(true, name " is not implemented yet"); })

GULON_API int32_t gulon_kmeans_init(const gulon_dataset *, int32_t, int32_t, int32_t, int32_t, float *, int32_t *) { NOT_YET("gulon_kmeans_init"); }
GULON_API int32_t gulon_kmeans_assign(const gulon_dataset *, int32_t, int32_t, const float *, int32_t, int32_t, int32_t *) { NOT_YET("gulon_kmeans_assign"); }
GULON_API int32_t gulon_kmeans_update(const gulon_dataset *, int32_t, int32_t, int32_t, const int32_t *, float *) { NOT_YET("gulon_kmeans_update"); }
GULON_API int32_t gulon_kmeans_iterate(const gulon_dataset *, int32_t, int32_t, const float *, int32_t, int32_t, float *) { NOT_YET("gulon_kmeans_iterate"); }
GULON_API int32_t gulon_kmeans_train(const gulon_dataset *, int32_t, int32_t, int32_t, int32_t, int32_t, float *, gulon_kmeans_report *, int32_t, int32_t *) { NOT_YET("gulon_kmeans_train"); }
GULON_API int32_t gulon_pq_train(const gulon_dataset *, int32_t, int32_t, int32_t, float *, gulon_kmeans_report *, int32_t, int32_t *) { NOT_YET("gulon_pq_train"); }
GULON_API int32_t gulon_pq_encode(const gulon_dataset *, int32_t, int32_t, const float *, uint8_t *) { NOT_YET("gulon_pq_encode"); }
