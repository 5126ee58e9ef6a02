// Internal declarations shared by the top-k users (scan, exact kNN).
#pragma once
#include <algorithm>
#include <memory>

#include "common.hpp"

namespace gulon {
// Merge `lists` sorted partial lists per query (element (l,q,e) at l*stride_l + q*stride_q + e,
// each list K+1 long, padded with (+inf, INT_MAX)).  final_out: write [B][K] idx/dist/count/flags,
// else write one (K+1)-list per query to out_pv/out_pi.
void launch_merge(bool final_out, const float *in_v, const int *in_i, int lists, long long stride_l,
                  long long stride_q, int B, int K, int *out_idx, float *out_dist, int *out_count,
                  int *out_flags, float *out_pv, int *out_pi, hipStream_t st);
}  // namespace gulon
