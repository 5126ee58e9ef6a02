// Internal declarations for the KMeans kernels' host drivers.
#pragma once
#include <cfloat>

#include "common.hpp"

namespace gulon {

struct KmeansWorkspace {
  DevBuf<float> cpad, off;
  DevBuf<unsigned> ties, local, block_tot;
  DevBuf<unsigned long long> tie_total, block_off;
  DevBuf<unsigned> hist, count, start, mismatch;
  DevBuf<int> order;
  unsigned long long last_draws = 0;   // RNG draws made by the last assign (0 = no exact ties)
  void ensure(int n, int k, int s);
};

void kmeans_assign_dev(KmeansWorkspace &ws, const float *dX, int n, int ld, int from, int s, const float *dC, int k,
                       int rng_batch, int *d_assign, hipStream_t st);
void kmeans_update_dev(KmeansWorkspace &ws, const float *dX, int n, int ld, int from, int s, int k,
                       const int *d_assign, float *dC, hipStream_t st);

}  // namespace gulon
