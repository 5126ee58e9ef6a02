// GroupedIndex.query's pre-selection (Index.scala:265-299) batched BY GROUP with 8-bit bound tables: grouped_filter.hip
#pragma once

#include "common.hpp"

namespace gulon {

constexpr int GF_QT = 16;       // queries per tile: one 16-byte table entry holds their bytes
constexpr int GF_CAP = 16384;   // survivors kept per query (more: the query goes to the literal kernels)
constexpr int GF_WAVES = 16;    // per-query lists written by gf_survivors (= gq_approx_scan's)
constexpr int GF_LIST = 64;     // entries per list (= gq_approx_scan's)
constexpr int GF_TG = 64;       // groups per workgroup of gf_tiles (four threads each)
constexpr int GF_PLACED = 512;  // entries at or below the GF_LIST-th smallest value that gf_survivors orders (more: the literal kernels)
constexpr int GF_SAMPLE_GROUPS = 16;   // nearest groups whose rows give a query its threshold, at most
constexpr int GF_SAMPLE_ROWS = 2048;   // ... and rows of them, at most

struct GfTile {                 // one group x up to GF_QT of the queries that search it
  int c, nq, r0, r1;            // group, queries in the tile, the group's rows [r0, r1)
  float xl;                     // the group's smallest row norm
  int pad[3];
  int qid[GF_QT];               // (padded with the last query)
};

struct GroupFilter {
  // per index: an 8-bit level of every row's |g + decode(codes)|^2 above the smallest one of its group
  DevBuf<uint8_t> xcode;
  DevBuf<float> xnlo;           // [g] the smallest row norm of every group
  float xn_step = 0.f;
  DevBuf<float> gnorm;          // [g] |centroid|^2
  float gnmax = 0.f;
  bool built = false;
  // scratch
  DevBuf<int> gcnt, pairs /* [g][B]: every group's queries */, meta, qcnt;
  DevBuf<GfTile> tiles;
  DevBuf<uint8_t> qb;           // [B][17][256] levels: 16 quantizers and the row-norm level
  DevBuf<float> qs;             // [B][4]: budget at base 0, 1 / step, spare, spare
  DevBuf<uint2> queue;          // [B][GF_CAP] (row, bits of the pair's base)
};

// xnorm: n floats (every one finite)
void group_filter_build(GroupFilter &gf, const float *xnorm, int n, const float *gcent, const int *bounds, int g, int d);

// The shapes the filter takes: 8-bit codes, at most 16 quantizers (one 16-byte word or up to four 4-byte words per
// row), at most 256 centroids per quantizer.
bool group_filter_applies(int m, int m_pad, int ng, int vec, int k, int d);

// Writes, for every query, the GF_LIST smallest (D~, row) of its searched rows, ascending by (D~, row) and padded with
// (+inf, INT_MAX) -- what gq_approx_scan + merge_lists produce (amv / ami: [B][GF_LIST]) -- and anan ([B][GF_WAVES]: a
// non-zero entry sends the query to the literal kernels).
void group_filter_run(GroupFilter &gf, const uint8_t *codes, int ng, int vec, int m, int m_pad, int k, int d,
                      float *P /* out: [B][m_pad][256], the queries' -2 q.c tables (gq_ptables') */, const float *pq_cents,
                      const int *from, const int *sdim,
                      const float *xnorm, float xnmax, const float *gcent, const int *bounds, int g, const float *Q,
                      const float *cdist /* [B][g] squared query-centroid distances */, const int *nn, int nn_stride,
                      const int *nn_cnt, int B, float *amv, int *ami, int *anan, hipStream_t st);

}  // namespace gulon
