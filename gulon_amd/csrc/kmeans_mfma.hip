// MFMA filter for KMeans.assign (KMeans.scala:24-98) on gfx950.
//
// The reference computes, per row and centroid, d = offsets[c] - 2*(x . c) with an
// unfused sequential fp32 chain and takes the argmin with strict '<' (ties draw from
// java.util.Random).  v_mfma_f32_32x32x2_f32 computes the same quantity as an fma chain
// (one rounding per product-add), so its value d' differs from d by at most
//     E = (2s+1) * 2^-24 * (|c|^2 + 2 |x||c|) * (1 + o(1))
// (the epilogue additionally perturbs d' by <= 31 ulp to carry the centroid index; that is
// added to the band)
// (Higham's gamma bounds for both chains).  The filter replays the reference's scan over
// c = 0..k-1 on d':  if every comparison against the running minimum is decided by more
// than 2E, the reference's scan takes exactly the same branches, draws no random bit, and
// the MFMA argmin IS the reference's answer.  Rows with any comparison inside the 2E band
// (including exact ties) are appended to a list and re-evaluated by the exact VALU kernel
// (assign_exact + assign_resolve in kmeans.hip).  Output is therefore bit-identical to the
// reference algorithm; the MFMA only decides which rows need the expensive path.
//
// Layout: centroid tiles are the MFMA M dimension, data rows the N dimension, the
// sub-vector the K dimension (2 per instruction).  The data slice is pre-packed once per
// (dataset, slice) as [n/32][T][64] so every B operand is one coalesced 256-byte load; the
// -2*c A operands and the offsets (C-init) sit in LDS.  After the K loop two 32x32 tiles
// are exchanged with v_permlane32_swap so that each lane owns ONE data row and sees its
// 32 distances of the centroid block in ascending centroid order.
#include <type_traits>

#include "kmeans.hpp"

namespace gulon {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// pack X[:, from:from+s] into MFMA B-operand order: out[(tile*T + t)*64 + l] =
// X[tile*32 + (l&31)][from + 2t + (l>>5)]  (0 outside the matrix / the slice)
__global__ void pack_slice_kernel(const float *__restrict__ X, int n, int ld, int from, int s, int T,
                                  long long total, float *__restrict__ out) {
  long long t0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t0 >= total) return;
  int l = (int)(t0 & 63);
  long long tt = t0 >> 6;
  int t = (int)(tt % T);
  long long tile = tt / T;
  long long row = tile * 32 + (l & 31);
  int e = 2 * t + (l >> 5);
  out[t0] = (row < n && e < s) ? X[(size_t)row * ld + from + e] : 0.f;
}

// A operands (-2*c), C-init offsets (1e38 beyond k) and max |c|^2
__global__ void pack_centroids_kernel(const float *__restrict__ C, const float *__restrict__ off, int k, int s, int T,
                                      int nkb, float *__restrict__ apack, float *__restrict__ offp,
                                      unsigned *__restrict__ cmax2_bits) {
  int t0 = blockIdx.x * blockDim.x + threadIdx.x;
  if (t0 < nkb * T * 64) {
    int l = t0 & 63, t = (t0 >> 6) % T, kb = (t0 >> 6) / T;
    int c = kb * 32 + (l & 31), e = 2 * t + (l >> 5);
    apack[t0] = (c < k && e < s) ? -2.0f * C[(size_t)c * s + e] : 0.f;
  }
  if (t0 < nkb * 32) {
    // padding centroids (last block): a huge FINITE offset -- never the minimum, and its key stays a
    // number (+inf with an index in the mantissa would be a NaN and flag every row through the band)
    float o = t0 < k ? off[t0] : 1.0e38f;
    offp[t0] = o;
    if (t0 < k && o == o && o < INFINITY) atomicMax(cmax2_bits, __float_as_uint(o));
  }
}

template <int T>
__global__ __launch_bounds__(256) void assign_mfma(const float *__restrict__ xq, int n, long long npairs,
                                                   const float *__restrict__ apack, const float *__restrict__ offp,
                                                   int nkb, const unsigned *__restrict__ cmax2_bits, float errk,
                                                   int *__restrict__ assign, int *__restrict__ flag_rows,
                                                   unsigned *__restrict__ flag_count) {
  extern __shared__ float sm[];
  float *sA = sm;                       // nkb*T*64
  float *sOff = sm + (size_t)nkb * T * 64;  // 2 copies of nkb*32
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int e = tid; e < nkb * T * 64; e += 256) sA[e] = apack[e];
  for (int e = tid; e < nkb * 32; e += 256) { sOff[e] = offp[e]; sOff[nkb * 32 + e] = offp[e]; }
  __syncthreads();
  const float cmax2 = __uint_as_float(*cmax2_bits);
  const int half = lane >> 5;

  // B operands of the NEXT tile pair are loaded while the current one is in the matrix pipe
  const long long pstride = (long long)gridDim.x * 4;
  float nbx[T], nby[T];
  {
    const long long pp0 = (long long)blockIdx.x * 4 + wave;
    const float *px = xq + (size_t)(2 * min(pp0, npairs - 1)) * T * 64 + lane;
#pragma unroll
    for (int t = 0; t < T; t++) {
      nbx[t] = px[(size_t)t * 64];
      nby[t] = px[(size_t)(T + t) * 64];
    }
  }
  for (long long pp = (long long)blockIdx.x * 4 + wave; pp < npairs; pp += pstride) {
    float bx[T], by[T];
#pragma unroll
    for (int t = 0; t < T; t++) { bx[t] = nbx[t]; by[t] = nby[t]; }
    {
      const float *px = xq + (size_t)(2 * min(pp + pstride, npairs - 1)) * T * 64 + lane;   // clamped: always valid
#pragma unroll
      for (int t = 0; t < T; t++) {
        nbx[t] = px[(size_t)t * 64];
        nby[t] = px[(size_t)(T + t) * 64];
      }
    }
    // |x|^2 of the row this lane will own after the swap
    float nx = 0.f, ny = 0.f;
#pragma unroll
    for (int t = 0; t < T; t++) { nx += bx[t] * bx[t]; ny += by[t] * by[t]; }
    {
      auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(nx), __float_as_uint(ny), false, false);
      nx = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    // 2E band (see file header); non-finite inputs make it NaN => row flagged below
    const float e2 = errk * (cmax2 + 2.0f * __fsqrt_rn(nx * cmax2)) + 1e-30f;

    float pmin = FLT_MAX;      // running minimum key
    float mband = INFINITY;    // smallest |key - running minimum| seen by this row's scan
    int best = -1;

    // One centroid block = 2*T MFMAs (tiles X and Y, C-init = offsets) and one scan epilogue of ~150 VALU
    // instructions.  A wave issues in order, so ten MFMAs followed by the scan leave the matrix pipe idle for the
    // whole scan (and two waves of a SIMD fall into lockstep: both in their MFMAs, then both in their scans --
    // 42 % matrix utilisation in round 1).  The MFMAs of block kb+1 are therefore issued one at a time BETWEEN the
    // eight pieces of the scan of block kb, pinned there with scheduling barriers: every MFMA's 64 cycles are
    // covered by ~16 VALU instructions of the same wave.
    //
    // C-init: the two accumulator tiles are initialised by two separate LDS reads (sOff is stored twice) instead
    // of 32 register copies of one read.
    auto init_block = [&](int kb, f32x16 &ax, f32x16 &ay, float (&a)[T]) {
      const float4 *so = reinterpret_cast<const float4 *>(sOff + kb * 32 + 4 * half);
      const float4 *so2 = reinterpret_cast<const float4 *>(sOff + nkb * 32 + kb * 32 + 4 * half);
#pragma unroll
      for (int g = 0; g < 4; g++) {
        float4 o = so[2 * g];   // centroids 8g + 4*half + (0..3)
        ax[4 * g + 0] = o.x; ax[4 * g + 1] = o.y; ax[4 * g + 2] = o.z; ax[4 * g + 3] = o.w;
        float4 o2 = so2[2 * g];
        ay[4 * g + 0] = o2.x; ay[4 * g + 1] = o2.y; ay[4 * g + 2] = o2.z; ay[4 * g + 3] = o2.w;
      }
      const float *pa = sA + (size_t)kb * T * 64 + lane;
#pragma unroll
      for (int t = 0; t < T; t++) a[t] = pa[t * 64];
    };
    // MFMA number m of a block: t = m / 2, tile X (even m) or Y (odd m)
    auto one_mfma = [&](int m, f32x16 &ax, f32x16 &ay, const float (&a)[T]) {
      if (m & 1) ay = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m >> 1], by[m >> 1], ay, 0, 0, 0);
      else ax = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m >> 1], bx[m >> 1], ax, 0, 0, 0);
    };
    // Piece h (0..7) of the scan epilogue of the block held in (ax, ay): centroids 4h .. 4h+3 of the block.
    // After the swaps lanes 0-31 hold tile X's row (lane), lanes 32-63 tile Y's row (lane-32):
    // ax[r] = centroids 8(r>>2) + (r&3), ay[r] = centroids 8(r>>2) + 4 + (r&3) of this block.
    // The reference's scan over c on d' (ascending centroid index), in 3.5 VALU ops per distance: the position j
    // inside the block replaces the 5 low mantissa bits of d' ("key"), so ONE running float minimum carries both
    // the value and the winner; the band test accumulates  min |key_j - running_min_before_j|  with a 3-input min
    // and is compared once per row.  The <= 31 ulp key perturbation is part of the band (errk).
    auto scan_piece = [&](int h, f32x16 &ax, f32x16 &ay) {
      if ((h & 1) == 0) {
#pragma unroll
        for (int r = 4 * (h >> 1); r < 4 * (h >> 1) + 4; r++) {
          auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(ax[r]), __float_as_uint(ay[r]), false, false);
          ax[r] = __uint_as_float(sw[0]);
          ay[r] = __uint_as_float(sw[1]);
        }
      }
      unsigned key[4];
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const int j = 4 * h + e;
        const float v = (h & 1) ? ay[4 * (h >> 1) + e] : ax[4 * (h >> 1) + e];
        key[e] = (__float_as_uint(v) & ~31u) | (unsigned)j;
      }
#pragma unroll
      for (int e = 0; e < 4; e += 2) {
        const float k0 = __uint_as_float(key[e]), k1 = __uint_as_float(key[e + 1]);
        float q0, q1;
        // plain v_min_f32 / v_min3_f32: the intrinsic forms add a canonicalising v_max per value
        asm("v_min_f32 %0, %1, %2" : "=v"(q0) : "v"(pmin), "v"(k0));
        asm("v_min_f32 %0, %1, %2" : "=v"(q1) : "v"(q0), "v"(k1));
        const float d0 = k0 - pmin, d1 = k1 - q0;
        asm("v_min3_f32 %0, %1, |%2|, |%3|" : "=v"(mband) : "v"(mband), "v"(d0), "v"(d1));
        pmin = q1;
      }
    };
    // scan block kb (in sx, sy) while the MFMAs of block kb + 1 go into (nx_, ny_)
    auto step = [&](auto next_tag, int kb, f32x16 &sx, f32x16 &sy, f32x16 &nx_, f32x16 &ny_) {
      constexpr bool NEXT = decltype(next_tag)::value;
      float a[T];
      if (NEXT) init_block(kb + 1, nx_, ny_, a);
      const float qbefore = pmin;
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int h = 0; h < 8; h++) {
        if (NEXT) {
#pragma unroll
          for (int m = 0; m < 2 * T; m++)
            if (m * 8 / (2 * T) == h) one_mfma(m, nx_, ny_, a);
          __builtin_amdgcn_sched_barrier(0);
        }
        scan_piece(h, sx, sy);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (__float_as_uint(pmin) != __float_as_uint(qbefore)) best = kb * 32 + (int)(__float_as_uint(pmin) & 31u);
    };
    using Yes = std::integral_constant<bool, true>;
    using No = std::integral_constant<bool, false>;

    f32x16 ax, ay, bx2, by2;
    {
      float a0[T];
      init_block(0, ax, ay, a0);
#pragma unroll
      for (int m = 0; m < 2 * T; m++) one_mfma(m, ax, ay, a0);
    }
    int kb = 0;
    for (; kb + 2 < nkb; kb += 2) {
      step(Yes{}, kb, ax, ay, bx2, by2);
      step(Yes{}, kb + 1, bx2, by2, ax, ay);
    }
    if (kb + 1 < nkb) {
      step(Yes{}, kb, ax, ay, bx2, by2);
      step(No{}, kb + 1, bx2, by2, ax, ay);
    } else {
      step(No{}, kb, ax, ay, bx2, by2);
    }

    const long long row = pp * 64 + lane;
    const bool in_range = row < n;
    const bool my_amb = !(mband > e2);   // also true when mband is NaN
    const bool flagged = in_range && (my_amb || best < 0 || !(e2 < INFINITY) || !(fabsf(pmin) < INFINITY));
    if (in_range && !flagged) assign[row] = best;
    const unsigned long long fm = __ballot(flagged);
    if (fm) {
      unsigned base = 0;
      if (lane == 0) base = atomicAdd(flag_count, (unsigned)__popcll(fm));
      base = __builtin_amdgcn_readfirstlane(base);
      if (flagged) flag_rows[base + __popcll(fm & ((1ull << lane) - 1ull))] = (int)row;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Streaming variant for long sub-vectors and/or many centroids (coarse clustering of whole
// vectors: s = d = 128, k = n / 1000): the A operands no longer fit LDS, so every workgroup walks
// the centroid blocks with a double-buffered 32-centroid A block (T*64 floats) shared by its four
// waves, while each wave keeps the B operands of its tile pair (2*T floats per lane) in registers.
// Per centroid block a wave issues 2*T MFMAs against one scan epilogue, so the matrix pipe
// dominates (T = 64: 128 MFMAs = 8192 cycles vs ~400 cycles of VALU).  Same error band, same
// flagged-row protocol, same output as assign_mfma<T>.
// ---------------------------------------------------------------------------------------------
template <int T>
__global__ __launch_bounds__(256) void assign_mfma_stream(const float *__restrict__ xq, int n, long long npairs,
                                                          const float *__restrict__ apack,
                                                          const float *__restrict__ offp, int nkb,
                                                          const unsigned *__restrict__ cmax2_bits, float errk,
                                                          int *__restrict__ assign, int *__restrict__ flag_rows,
                                                          unsigned *__restrict__ flag_count) {
  extern __shared__ float sm[];
  constexpr int ABLK = T * 64;           // A operands of one centroid block
  constexpr int BUF = ABLK + 32;         // + its 32 offsets
  constexpr int PER = (BUF + 255) / 256; // floats staged per thread
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5;
  const float cmax2 = __uint_as_float(*cmax2_bits);
  float stage[PER];
  auto fetch = [&](int kb) {             // global -> registers (block kb's A operands and offsets)
#pragma unroll
    for (int u = 0; u < PER; u++) {
      const int e = tid + u * 256;
      stage[u] = e < ABLK ? apack[(size_t)kb * ABLK + e] : e < BUF ? offp[(size_t)kb * 32 + (e - ABLK)] : 0.f;
    }
  };
  auto commit = [&](int buf) {           // registers -> LDS buffer
#pragma unroll
    for (int u = 0; u < PER; u++) {
      const int e = tid + u * 256;
      if (e < BUF) sm[buf * BUF + e] = stage[u];
    }
  };
  const long long ngroups = (npairs + 3) / 4;
  for (long long pg = blockIdx.x; pg < ngroups; pg += gridDim.x) {
    const long long pp = pg * 4 + wave;
    const bool have_pair = pp < npairs;
    float bx[T], by[T];
    {
      const float *px = xq + (size_t)(2 * (have_pair ? pp : npairs - 1)) * T * 64 + lane;   // clamped: always valid
#pragma unroll
      for (int t = 0; t < T; t++) {
        bx[t] = px[(size_t)t * 64];
        by[t] = px[(size_t)(T + t) * 64];
      }
    }
    float nx = 0.f, ny = 0.f;
#pragma unroll
    for (int t = 0; t < T; t++) { nx += bx[t] * bx[t]; ny += by[t] * by[t]; }
    {
      auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(nx), __float_as_uint(ny), false, false);
      nx = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    const float e2 = errk * (cmax2 + 2.0f * __fsqrt_rn(nx * cmax2)) + 1e-30f;
    float pmin = FLT_MAX, mband = INFINITY;
    int best = -1;

    __syncthreads();                     // the previous group's last block is no longer read
    fetch(0);
    commit(0);
    __syncthreads();
    for (int kb = 0; kb < nkb; kb++) {
      const float *sA = sm + (kb & 1) * BUF;
      if (kb + 1 < nkb) fetch(kb + 1);   // in flight during this block's MFMAs
      f32x16 ax, ay;
      {
        const float4 *so = reinterpret_cast<const float4 *>(sA + ABLK + 4 * half);
#pragma unroll
        for (int g = 0; g < 4; g++) {
          const float4 o = so[2 * g];    // centroids 8g + 4*half + (0..3)
          ax[4 * g + 0] = o.x; ax[4 * g + 1] = o.y; ax[4 * g + 2] = o.z; ax[4 * g + 3] = o.w;
          ay[4 * g + 0] = o.x; ay[4 * g + 1] = o.y; ay[4 * g + 2] = o.z; ay[4 * g + 3] = o.w;
        }
      }
#pragma unroll
      for (int t = 0; t < T; t++) {
        const float a = sA[t * 64 + lane];
        ax = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bx[t], ax, 0, 0, 0);
        ay = __builtin_amdgcn_mfma_f32_32x32x2f32(a, by[t], ay, 0, 0, 0);
      }
      // scan epilogue: identical to assign_mfma<T>::scan_block
#pragma unroll
      for (int r = 0; r < 16; r++) {
        auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(ax[r]), __float_as_uint(ay[r]), false, false);
        ax[r] = __uint_as_float(sw[0]);
        ay[r] = __uint_as_float(sw[1]);
      }
      unsigned key[32];
#pragma unroll
      for (int j = 0; j < 32; j++) {
        const float v = ((j >> 2) & 1) ? ay[4 * (j >> 3) + (j & 3)] : ax[4 * (j >> 3) + (j & 3)];
        key[j] = (__float_as_uint(v) & ~31u) | (unsigned)j;
      }
      const float qbefore = pmin;
#pragma unroll
      for (int j = 0; j < 32; j += 2) {
        const float k0 = __uint_as_float(key[j]), k1 = __uint_as_float(key[j + 1]);
        float q0, q1;
        asm("v_min_f32 %0, %1, %2" : "=v"(q0) : "v"(pmin), "v"(k0));
        asm("v_min_f32 %0, %1, %2" : "=v"(q1) : "v"(q0), "v"(k1));
        const float d0 = k0 - pmin, d1 = k1 - q0;
        asm("v_min3_f32 %0, %1, |%2|, |%3|" : "=v"(mband) : "v"(mband), "v"(d0), "v"(d1));
        pmin = q1;
      }
      if (__float_as_uint(pmin) != __float_as_uint(qbefore)) best = kb * 32 + (int)(__float_as_uint(pmin) & 31u);
      if (kb + 1 < nkb) commit((kb + 1) & 1);   // the other buffer was last read during block kb-1
      __syncthreads();
    }

    const long long row = pp * 64 + lane;
    const bool in_range = have_pair && row < n;
    const bool my_amb = !(mband > e2);   // also true when mband is NaN
    const bool flagged = in_range && (my_amb || best < 0 || !(e2 < INFINITY) || !(fabsf(pmin) < INFINITY));
    if (in_range && !flagged) assign[row] = best;
    const unsigned long long fm = __ballot(flagged);
    if (fm) {
      unsigned base = 0;
      if (lane == 0) base = atomicAdd(flag_count, (unsigned)__popcll(fm));
      base = __builtin_amdgcn_readfirstlane(base);
      if (flagged) flag_rows[base + __popcll(fm & ((1ull << lane) - 1ull))] = (int)row;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// The same filter on the bf16 matrix cores (s <= 16, k <= 576).
//
// v_mfma_f32_32x32x2_f32 runs on the vector ALUs' fp32 rate and does not overlap with VALU work (matrix busy 46 %
// + vector busy 54 % of the launch in round 1, unchanged by interleaving the instruction streams): five of them per
// 32 x 32 tile plus ~130 VALU instructions of scan epilogue capped the kernel at ~45 % of the fp32 matrix peak.
// The bf16 cores are a separate pipe (scripts/micro/bf16_mfma.hip: 12 MFMAs + 144 VALU per iteration take 5.45 ms
// interleaved against 3.48 + 4.06 alone).  An fp32 value is the exact sum of three bf16 pieces (8 + 8 + 8
// significand bits, round-to-nearest splitting), so
//     (-2 c) . x  =  sum over pieces (a, b) of  c_a . x_b ,
// and the six products with a + b <= 4 carry everything down to 2^-23 of |c||x| -- one v_mfma_f32_32x32x16_bf16
// each (K = 16 covers the whole sub-vector), fp32 accumulation, C-init = offsets.  The value d' differs from the
// reference's unfused fp32 chain by at most
//     E = [ (s + 1) 2^-24 (reference chain)  +  6 (s + 2) 2^-24 (six accumulations of <= s + 1 terms each)
//           +  2^-22.9 (dropped pieces) ] * (|c|^2 + 2 |x||c|),
// and the scan epilogue, the band test (2E plus the 31-ulp key perturbation) and the flagged-row protocol are
// those of assign_mfma<T>: rows with any comparison inside the band go to the exact kernel, so the output is
// bit-identical to the reference whatever the matrix cores round like (gulon_selftest_assign_band measures the
// band's margin on the hardware).
// ---------------------------------------------------------------------------------------------
#ifndef GULON_BF16_WAVES
#define GULON_BF16_WAVES
#endif
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned bf16_rn_bits(float f) {   // round to nearest even; finite inputs
  const unsigned u = __float_as_uint(f);
  return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ float bf16_bits_to_f32(unsigned h) { return __uint_as_float(h << 16); }
// x = p1 + p2 + p3 exactly (each a bf16): p1 = RN(x), p2 = RN(x - p1), p3 = x - p1 - p2
__device__ __forceinline__ void split3(float x, unsigned &p1, unsigned &p2, unsigned &p3) {
  p1 = bf16_rn_bits(x);
  const float r1 = x - bf16_bits_to_f32(p1);
  p2 = bf16_rn_bits(r1);
  const float r2 = r1 - bf16_bits_to_f32(p2);
  p3 = bf16_rn_bits(r2);
}
__device__ __forceinline__ void pack8(const float (&v)[8], uint4 &o1, uint4 &o2, uint4 &o3) {
  unsigned a[8], b[8], c[8];
#pragma unroll
  for (int e = 0; e < 8; e++) split3(v[e], a[e], b[e], c[e]);
  o1 = make_uint4(a[0] | (a[1] << 16), a[2] | (a[3] << 16), a[4] | (a[5] << 16), a[6] | (a[7] << 16));
  o2 = make_uint4(b[0] | (b[1] << 16), b[2] | (b[3] << 16), b[4] | (b[5] << 16), b[6] | (b[7] << 16));
  o3 = make_uint4(c[0] | (c[1] << 16), c[2] | (c[3] << 16), c[4] | (c[5] << 16), c[6] | (c[7] << 16));
}

// The compact operand layout (s <= 13).  The six piece products (a3,b1) (a1,b3) (a2,b2) (a2,b1) (a1,b2) (a1,b1) need
// 6 s of a matrix instruction's K slots, not 6 x 16: product t takes the slots [t s, (t + 1) s) of a virtual K of 16 NA
// (NA = ceil(6 s / 16) operand words per lane: 3 for s <= 8, 4 for s <= 10, 5 for s <= 13), piece pa[t] of -2 c on the A
// side, piece pb[t] of x on the B side, zeros behind the last product -- NA matrix instructions per tile and centroid
// block instead of six (config 3, s = 9 / 10: four).  Slots are summed smallest products first, as before.
__host__ __device__ constexpr int split_words(int s) { return s <= 8 ? 3 : s <= 10 ? 4 : s <= 13 ? 5 : 6; }
// the operand word (8 slots) of lane half `half` of instruction u: get(e) = element e of the row / centroid
template <typename Get>
__device__ __forceinline__ uint4 compact_word(Get get, int s, int u, int half, bool a_side) {
  unsigned h[8];
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const int g = 16 * u + 8 * half + j;
    const int t = g / s, e = g - t * s;
    unsigned p1 = 0, p2 = 0, p3 = 0;
    if (t < 6) split3(get(e), p1, p2, p3);
    // pa = {2, 0, 1, 1, 0, 0}, pb = {0, 2, 1, 0, 1, 0}
    const int piece = a_side ? (t == 0 ? 2 : (t == 2 || t == 3) ? 1 : 0) : (t == 1 ? 2 : (t == 2 || t == 4) ? 1 : 0);
    h[j] = piece == 0 ? p1 : piece == 1 ? p2 : p3;
  }
  return make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16));
}

// rows -> NA compact operand words per (tile, lane) + |x|^2 per row.  A workgroup packs four tiles: their 128 row slices go
// through LDS (every value is read from memory once, where a thread per lane re-read it for each of the up to three
// products it appears in: 1.45 ms per sub-quantizer at config 3 against 0.64 for the three-piece layout), and the
// slot -> (product, element) map is a table instead of a division per slot.
__global__ __launch_bounds__(256) void pack_slice_compact(const float *__restrict__ X, int n, int ld, int from, int s, int na,
                                                          long long nlanes, uint4 *__restrict__ out, float *__restrict__ xn) {
  __shared__ float xs[128 * 13];
  __shared__ unsigned char tab[16 * 5];
  const int tid = threadIdx.x;
  const long long tile0 = (long long)blockIdx.x * 4, row0 = tile0 * 32;
  for (int i = tid; i < 128 * s; i += 256) {
    const int r = i / s, e = i - r * s;
    xs[i] = row0 + r < n ? X[(size_t)(row0 + r) * ld + from + e] : 0.f;
  }
  if (tid < 16 * na) { const int t = tid / s; tab[tid] = (unsigned char)(t < 6 ? (t << 4) | (tid - t * s) : 0xF0); }
  __syncthreads();
  const int l = tid & 63, rl = (tid >> 6) * 32 + (l & 31);
  const long long tile = tile0 + (tid >> 6);
  if (tile * 64 + l >= nlanes) return;
  const float *xr = xs + rl * s;
  for (int u = 0; u < na; u++) {
    unsigned h[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int te = tab[16 * u + 8 * (l >> 5) + j], t = te >> 4;
      unsigned p1 = 0, p2 = 0, p3 = 0;
      if (t < 6) split3(xr[te & 15], p1, p2, p3);
      const int piece = t == 1 ? 2 : (t == 2 || t == 4) ? 1 : 0;      // pb = {0, 2, 1, 0, 1, 0}
      h[j] = piece == 0 ? p1 : piece == 1 ? p2 : p3;
    }
    out[(tile * na + u) * 64 + l] = make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16));
  }
  const long long row = row0 + rl;
  if (l < 32 && row < n) {
    float acc = 0.f;
    for (int e = 0; e < s; e++) { const float x = xr[e]; acc += x * x; }
    xn[row] = acc;
  }
}

// rows -> three bf16 operand pieces per (tile, lane) + |x|^2 per row
__global__ void pack_slice_split(const float *__restrict__ X, int n, int ld, int from, int s, long long nlanes,
                                 uint4 *__restrict__ out, float *__restrict__ xn) {
  const long long t0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t0 >= nlanes) return;
  const int l = (int)(t0 & 63);
  const long long tile = t0 >> 6;
  const long long row = tile * 32 + (l & 31);
  const int e0 = 8 * (l >> 5);
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; e++) v[e] = (row < n && e0 + e < s) ? X[(size_t)row * ld + from + e0 + e] : 0.f;
  uint4 o1, o2, o3;
  pack8(v, o1, o2, o3);
  out[(tile * 3 + 0) * 64 + l] = o1;
  out[(tile * 3 + 1) * 64 + l] = o2;
  out[(tile * 3 + 2) * 64 + l] = o3;
  if (l < 32 && row < n) {
    float acc = 0.f;
    for (int e = 0; e < s; e++) { const float x = X[(size_t)row * ld + from + e]; acc += x * x; }
    xn[row] = acc;
  }
}

// A operands: -2 c in three bf16 pieces [kb][piece][64 lanes]; offsets (1e38 beyond k) and max |c|^2 as before
__global__ void pack_centroids_split(const float *__restrict__ C, const float *__restrict__ off, int k, int s, int nkb,
                                     uint4 *__restrict__ apack, float *__restrict__ offp,
                                     unsigned *__restrict__ cmax2_bits, int na /* 0: three pieces; else compact words */) {
  const int t0 = blockIdx.x * blockDim.x + threadIdx.x;
  if (t0 < nkb * 64 && na) {           // the compact layout (same centroid order over the rows of the A operand)
    const int l = t0 & 63, kb = t0 >> 6, i = l & 31;
    const int c = kb * 32 + 16 * ((i >> 2) & 1) + 4 * (i >> 3) + (i & 3);
    auto get = [&](int e) { return c < k ? -2.0f * C[(size_t)c * s + e] : 0.f; };
    for (int u = 0; u < na; u++) apack[(kb * na + u) * 64 + l] = compact_word(get, s, u, l >> 5, true);
  } else if (t0 < nkb * 64) {
    const int l = t0 & 63, kb = t0 >> 6;
    // Which centroid sits in which row of the A operand is ours to choose.  The 32x32 result leaves a lane with rows
    // 8g + 4h + t (g, t = 0..3; h = lane / 32): row i = 8g + 4h + t carries centroid 16h + 4g + t of the block, so that
    // the lower half-wave ends up with the block's centroids 0..15 and the upper one with 16..31, each in register order
    // (assign_bf16 scans its 16 in order and the two halves meet once per block).
    const int i = l & 31;
    const int c = kb * 32 + 16 * ((i >> 2) & 1) + 4 * (i >> 3) + (i & 3), e0 = 8 * (l >> 5);
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; e++) v[e] = (c < k && e0 + e < s) ? -2.0f * C[(size_t)c * s + e0 + e] : 0.f;
    uint4 o1, o2, o3;
    pack8(v, o1, o2, o3);
    apack[(kb * 3 + 0) * 64 + l] = o1;
    apack[(kb * 3 + 1) * 64 + l] = o2;
    apack[(kb * 3 + 2) * 64 + l] = o3;
  }
  if (t0 < nkb * 32) {
    float o = t0 < k ? off[t0] : 1.0e38f;
    offp[t0] = o;
    if (t0 < k && o == o && o < INFINITY) atomicMax(cmax2_bits, __float_as_uint(o));
  }
}

// probe != nullptr (one workgroup, selftest): the raw d' of the first tile pair against centroid block 0 go to
// probe[64 rows][32 centroids] and nothing else is written
// NA operand words per lane, tile and centroid block: three pieces and six products of pairs of them (COMPACT = false),
// or NA compact words, one matrix instruction each (see compact_word)
template <int NA, bool COMPACT>
__global__ __launch_bounds__(256) GULON_BF16_WAVES void assign_bf16(const uint4 *__restrict__ xq, const float *__restrict__ xn, int n,
                                                   long long npairs, const uint4 *__restrict__ apack,
                                                   const float *__restrict__ offp, int nkb,
                                                   const unsigned *__restrict__ cmax2_bits, float errk,
                                                   int *__restrict__ assign, int *__restrict__ flag_rows,
                                                   unsigned *__restrict__ flag_count, float *__restrict__ probe) {
  extern __shared__ float sm[];
#ifdef GULON_BF16_CLOCKS   // experiment builds: shader cycles per 100 MHz tick over the life of workgroup 0
  const unsigned long long clk_c0 = clock64(), clk_w0 = wall_clock64();
#endif
  constexpr int NMF = COMPACT ? 2 * NA : 12;                       // matrix instructions per block and tile pair
  uint4 *sA = reinterpret_cast<uint4 *>(sm);                       // [nkb][NA][64]
  float *sOff = sm + (size_t)nkb * NA * 64 * 4;                     // 2 copies of nkb*32
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int e = tid; e < nkb * NA * 64; e += 256) sA[e] = apack[e];
  for (int e = tid; e < nkb * 32; e += 256) { sOff[e] = offp[e]; sOff[nkb * 32 + e] = offp[e]; }
  __syncthreads();
  const float cmax2 = __uint_as_float(*cmax2_bits);
  const int half = lane >> 5;
  const bool upper = half != 0;

  // B operands of the NEXT tile pair are loaded while the current one is in the matrix pipe
  const long long pstride = (long long)gridDim.x * 4;
  uint4 nb[2][NA];
  float nxn;
  auto load_pair = [&](long long pp) {
    const long long pc = min(pp, npairs - 1);                      // clamped: always valid
    const uint4 *px = xq + (size_t)(2 * pc) * NA * 64 + lane;
#pragma unroll
    for (int y = 0; y < 2; y++)
#pragma unroll
      for (int p = 0; p < NA; p++) nb[y][p] = px[(size_t)(y * NA + p) * 64];
    const long long row = pc * 64 + lane;
    nxn = row < n ? xn[row] : 0.f;
  };
  load_pair((long long)blockIdx.x * 4 + wave);
  f32x16 ax, ay, bx2, by2;
  bool have_acc = false;             // (ax, ay) already hold this pair's first block
  for (long long pp = (long long)blockIdx.x * 4 + wave; pp < npairs; pp += pstride) {
    bf16x8 bx[NA], by[NA];
#pragma unroll
    for (int p = 0; p < NA; p++) { bx[p] = __builtin_bit_cast(bf16x8, nb[0][p]); by[p] = __builtin_bit_cast(bf16x8, nb[1][p]); }
    const float nx = nxn;            // |x|^2 of the row this lane answers for (row pp*64 + lane)
    load_pair(pp + pstride);
    // 2E band (see file header); non-finite inputs make it NaN => row flagged below
    const float e2 = errk * (cmax2 + 2.0f * __fsqrt_rn(nx * cmax2)) + 1e-30f;

    // The scan of a row's 32 distances to a centroid block is the reference's: in centroid order, every key against
    // the running minimum of the keys before it.  A row's 32 distances sit in TWO lanes (l and l + 32, 16 each: the
    // matrix core's result layout).  Round 2 brought them into one lane with 16 v_permlane32_swap per block --
    // 19 cycles each, 300 of the ~1200 cycles a wave spent per block.  Here the row order of the A operand puts the
    // block's centroids 0..15 into the lower half-wave and 16..31 into the upper one, and the scan takes two passes
    // over the accumulators, both halves and both row tiles working side by side:
    //   pass 1  every lane: the minimum of its 16 keys (a tree of v_min3: no order inside a half is needed for it);
    //           ONE exchange per tile and block tells the upper half where the lower half left the running minimum;
    //   pass 2  every lane scans its 16 keys in order from the state it now knows: the minimum before the block for
    //           the lower half, min(that, the lower half's 16) for the upper half.
    // (Not replaceable by "smallest and second smallest within the band": the reference draws a random bit at EVERY exact
    // tie of its scan, also at one that a later, smaller distance makes irrelevant for this row -- and the draw moves the
    // stream the later rows of the 25 000-row batch break their ties with.  A running (min, second) per lane -- two
    // instructions per key, no exchanges -- was built and fails test_init_assign_update_bit_exact for exactly that.)
    // Running minimum and the centroid it sits at are the same in both lanes of a row afterwards; the band distance
    // accumulates per lane and the two are combined once per tile pair.  A key carries its register number (4 bits:
    // 15 ulp of perturbation instead of 31); which half a block's minimum came from is read off the two minima.
    struct Scan {
      float pin;       // running minimum key before the current block (both lanes of a row agree)
      float mband;     // smallest |key - running minimum| this LANE saw
      int best;        // where the running minimum sits: 2 * centroid block + (second half of the block?); -1: none yet
                       // (the register inside the half is the key's low four bits: read off `pin` once, at the end)
    };
    Scan sx{FLT_MAX, INFINITY, -1}, sy{FLT_MAX, INFINITY, -1};

    auto init_block = [&](int kb, f32x16 &ax, f32x16 &ay, bf16x8 (&a)[NA]) {
      const float4 *so = reinterpret_cast<const float4 *>(sOff + kb * 32 + 16 * half);
      const float4 *so2 = reinterpret_cast<const float4 *>(sOff + nkb * 32 + kb * 32 + 16 * half);
#pragma unroll
      for (int g = 0; g < 4; g++) {
        float4 o = so[g];       // centroids 16 * half + 4 g + (0..3) = registers 4 g + (0..3)
        ax[4 * g + 0] = o.x; ax[4 * g + 1] = o.y; ax[4 * g + 2] = o.z; ax[4 * g + 3] = o.w;
        float4 o2 = so2[g];
        ay[4 * g + 0] = o2.x; ay[4 * g + 1] = o2.y; ay[4 * g + 2] = o2.z; ay[4 * g + 3] = o2.w;
      }
#pragma unroll
      for (int p = 0; p < NA; p++) a[p] = __builtin_bit_cast(bf16x8, sA[(kb * NA + p) * 64 + lane]);
    };
    // the six piece products, smallest first: (a3,b1) (a1,b3) (a2,b2) (a2,b1) (a1,b2) (a1,b1); MFMA m: product m / 2,
    // tile X (even m) or Y (odd m)
    auto one_mfma = [&](int m, f32x16 &ax, f32x16 &ay, const bf16x8 (&a)[NA]) {
      constexpr int pa[6] = {2, 0, 1, 1, 0, 0}, pb[6] = {0, 2, 1, 0, 1, 0};
      const int t = m >> 1;
      const int ia = COMPACT ? t : pa[t], ib = COMPACT ? t : pb[t];
      if (m & 1) ay = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ia], by[ib], ay, 0, 0, 0);
      else ax = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ia], bx[ib], ax, 0, 0, 0);
    };
    auto fmin3 = [](float a, float b, float c) {
      float r;
      asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
      return r;
    };
    auto fmin2 = [](float a, float b) {
      float r;
      asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
      return r;
    };
    // the minimum of a lane's 16 keys (a tree: no order needed)
    auto min16 = [&](const f32x16 &acc) {
      const float m0 = fmin3(acc[0], acc[1], acc[2]), m1 = fmin3(acc[3], acc[4], acc[5]);
      const float m2 = fmin3(acc[6], acc[7], acc[8]), m3 = fmin3(acc[9], acc[10], acc[11]);
      const float m4 = fmin3(acc[12], acc[13], acc[14]);
      return fmin2(fmin3(m0, m1, m2), fmin3(m3, m4, acc[15]));
    };
    // pass 2 over keys r, r + 1 of BOTH tiles, from running minima (px, py); the two tiles' chains side by side.
    // (Both differences of a tile as one v_pk_add_f32 -- the keys are neighbouring accumulator registers, the minima kept
    // as a register pair -- was measured with the same compiler flags on one box: 18.5-18.8 ms per stage against 17.9-18.4
    // with the two v_sub_f32.  The packed instruction saves an issue slot and costs more than it saves.)
    auto scan2 = [&](const f32x16 &cx, const f32x16 &cy, int r, float &px, float &py, float &mbx, float &mby) {
      const float kx0 = cx[r], kx1 = cx[r + 1], ky0 = cy[r], ky1 = cy[r + 1];
      const float qx0 = fmin2(px, kx0), qy0 = fmin2(py, ky0), qx1 = fmin2(qx0, kx1), qy1 = fmin2(qy0, ky1);
      const float dx0 = kx0 - px, dx1 = kx1 - qx0, dy0 = ky0 - py, dy1 = ky1 - qy0;
      asm("v_min3_f32 %0, %1, |%2|, |%3|" : "=v"(mbx) : "v"(mbx), "v"(dx0), "v"(dx1));
      asm("v_min3_f32 %0, %1, |%2|, |%3|" : "=v"(mby) : "v"(mby), "v"(dy0), "v"(dy1));
      px = qx1; py = qy1;
    };
    // after a block: the row's running minimum and where it sits.  lo / hi = the minima of the block's first and
    // second 16 centroids (both lanes of a row hold both after the exchange); branch-free
    auto close_block = [&](Scan &st, int kb, float lo, float hi) {
      const float blk = fmin2(lo, hi);
      // equal keys of the two halves (same distance bits, same register): the lower half's centroid comes first
      const int where = hi < lo ? 2 * kb + 1 : 2 * kb;
      st.best = blk < st.pin ? where : st.best;         // (a NaN never wins: such a row is flagged by its band)
      st.pin = fmin2(st.pin, blk);
    };
    // One block: the scan of the accumulators (cx, cy) while the matrix instructions of the NEXT block -- the same row
    // tiles against centroid block kb + 1, or, at a pair's last block, the next pair's tiles against block 0 -- fill
    // (nx_, ny_).  Twelve matrix instructions, twelve slices of vector work between them: a matrix instruction
    // that depends on the one two slices back (same accumulator) never waits, and neither does the wave's issue.
    auto step = [&](auto next_tag, int kb, int kb_next, f32x16 &cx, f32x16 &cy, f32x16 &nx_, f32x16 &ny_,
                    const bf16x8 (&nbx)[NA], const bf16x8 (&nby)[NA]) {
      constexpr bool NEXT = decltype(next_tag)::value;
      bf16x8 a[NA];
      // twelve slices of vector work; slice `site` is followed by matrix instruction ceil(site NMF / 12) if that differs
      // from the next slice's (NMF = 12: every slice; 8: slices 0 1 3 4 6 7 9 10)
      auto mf = [&](int site) {
        if (!NEXT) return;
        const int m = (site * NMF + 11) / 12;
        if ((((site + 1) * NMF + 11) / 12) == m) return;
        constexpr int pa[6] = {2, 0, 1, 1, 0, 0}, pb[6] = {0, 2, 1, 0, 1, 0};
        const int t = m >> 1;
        const int ia = COMPACT ? t : pa[t], ib = COMPACT ? t : pb[t];
        if (m & 1) ny_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ia], nby[ib], ny_, 0, 0, 0);
        else nx_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ia], nbx[ib], nx_, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      };
      if (NEXT) init_block(kb_next, nx_, ny_, a);
      __builtin_amdgcn_sched_barrier(0);
      // pass 1: keys, minima of the 16, exchange
#pragma unroll
      for (int r = 0; r < 16; r++) cx[r] = __uint_as_float((__float_as_uint(cx[r]) & ~15u) | (unsigned)r);
      mf(0);
#pragma unroll
      for (int r = 0; r < 16; r++) cy[r] = __uint_as_float((__float_as_uint(cy[r]) & ~15u) | (unsigned)r);
      mf(1);
      const float mx = min16(cx), my = min16(cy);
      // permlane32_swap(a, b): a's upper half <-> b's lower half.  With a = b = m: the first result is the LOWER lane's
      // value in both lanes of a row, the second the UPPER lane's
      auto s1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
      auto s2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(my), __float_as_uint(my), false, false);
      const float lox = __uint_as_float(s1[0]), hix = __uint_as_float(s1[1]);
      const float loy = __uint_as_float(s2[0]), hiy = __uint_as_float(s2[1]);
      // the lower half starts from the minimum before the block, the upper half from min(that, the lower half's 16)
      float px = fmin2(sx.pin, upper ? lox : FLT_MAX);
      float py = fmin2(sy.pin, upper ? loy : FLT_MAX);
      mf(2);
      // pass 2
#pragma unroll
      for (int h = 0; h < 4; h++) {
        scan2(cx, cy, 4 * h, px, py, sx.mband, sy.mband);
        mf(3 + 2 * h);
        scan2(cx, cy, 4 * h + 2, px, py, sx.mband, sy.mband);
        mf(4 + 2 * h);
      }
      close_block(sx, kb, lox, hix);
      close_block(sy, kb, loy, hiy);
      mf(11);
    };
    using Yes = std::integral_constant<bool, true>;
    using No = std::integral_constant<bool, false>;

    if (!have_acc) {               // the pair's first block (a pair's last step leaves it ready: see below)
      bf16x8 a0[NA];
      init_block(0, ax, ay, a0);
#pragma unroll
      for (int m = 0; m < NMF; m++) one_mfma(m, ax, ay, a0);
    }
    if (probe) {   // selftest: raw distances of (first 64 rows) x (centroid block 0)
      if (blockIdx.x == 0 && wave == 0) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const int c = 16 * half + r;
          probe[(size_t)(lane & 31) * 32 + c] = ax[r];
          probe[(size_t)(32 + (lane & 31)) * 32 + c] = ay[r];
        }
      }
      return;
    }
    // the NEXT pair's B operands (requested at the top of this iteration, one whole pair ahead)
    bf16x8 fx[NA], fy[NA];
    const bool more = pp + pstride < npairs;
    int kb = 0;
    for (; kb + 2 < nkb; kb += 2) {
      step(Yes{}, kb, kb + 1, ax, ay, bx2, by2, bx, by);
      step(Yes{}, kb + 1, kb + 2, bx2, by2, ax, ay, bx, by);
    }
    if (kb + 1 < nkb) {
      step(Yes{}, kb, kb + 1, ax, ay, bx2, by2, bx, by);
      // an even number of blocks: the last step's idle accumulators are (ax, ay) again -- the next pair's first block
      // goes there, its twelve matrix instructions between this block's vector work instead of in a row of their own
#pragma unroll
      for (int p = 0; p < NA; p++) { fx[p] = __builtin_bit_cast(bf16x8, nb[0][p]); fy[p] = __builtin_bit_cast(bf16x8, nb[1][p]); }
      if (more) { step(Yes{}, kb + 1, 0, bx2, by2, ax, ay, fx, fy); have_acc = true; }
      else { step(No{}, kb + 1, 0, bx2, by2, ax, ay, fx, fy); have_acc = false; }
    } else {
      step(No{}, kb, 0, ax, ay, bx2, by2, bx, by);
      have_acc = false;
    }

    // a row's band distance = the smaller of its two lanes' (NaN if either is); lane l < 32 answers for row l of
    // tile X, lane l >= 32 for row l - 32 of tile Y
    float mband;
    {
      // after the swap: r[0] = {lower lanes: their X band, upper lanes: the lower lanes' Y band},
      //                 r[1] = {lower lanes: the upper lanes' X band, upper lanes: their Y band}
      auto sb = __builtin_amdgcn_permlane32_swap(__float_as_uint(sx.mband), __float_as_uint(sy.mband), false, false);
      const float mine = upper ? sy.mband : sx.mband;
      const float theirs = __uint_as_float(upper ? sb[0] : sb[1]);
      mband = (mine == mine && theirs == theirs) ? fminf(mine, theirs) : NAN;
    }
    const float pmin = upper ? sy.pin : sx.pin;
    const int where = upper ? sy.best : sx.best;
    const int best = where < 0 ? -1 : 16 * where + (int)(__float_as_uint(pmin) & 15u);
    const long long row = pp * 64 + lane;
    const bool in_range = row < n;
    const bool my_amb = !(mband > e2);   // also true when mband is NaN
    const bool flagged = in_range && (my_amb || best < 0 || !(e2 < INFINITY) || !(fabsf(pmin) < INFINITY));
    if (in_range && !flagged) assign[row] = best;
    const unsigned long long fm = __ballot(flagged);
    if (fm) {
      unsigned base = 0;
      if (lane == 0) base = atomicAdd(flag_count, (unsigned)__popcll(fm));
      base = __builtin_amdgcn_readfirstlane(base);
      if (flagged) flag_rows[base + __popcll(fm & ((1ull << lane) - 1ull))] = (int)row;
    }
  }
#ifdef GULON_BF16_CLOCKS
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    const unsigned long long dc = clock64() - clk_c0, dw = wall_clock64() - clk_w0;
    printf("[clocks] %llu shader cycles in %llu ticks of 10 ns: %.3f GHz\n", dc, dw, (double)dc / (double)dw / 10.0);
  }
#endif
}

// operand words per lane, tile and centroid block: 0 = three pieces (six matrix instructions), else the compact layout
static int split_compact_words(int s) {
  static const bool off = [] { const char *e = getenv("GULON_KMEANS_COMPACT"); return e && atoi(e) == 0; }();   // A/B knob
  return !off && split_words(s) < 6 ? split_words(s) : 0;
}
static bool mfma_split(int s, int k) {
  static const bool off = getenv("GULON_KMEANS_F32_MFMA") != nullptr;   // A/B knob: the fp32 matrix instruction instead
  const int nkb = (k + 31) / 32;
  const int na = split_compact_words(s) ? split_compact_words(s) : 3;
  return !off && s >= 1 && s <= 16 && ((size_t)nkb * (na * 64 * 16 + 64 * 4)) <= 60 * 1024;
}
static float split_errk(int s) {
  // band = 2.1 * E + key perturbation (see the kernel's header): E / S = [(s+1) + 6 (s+2)] 2^-24 + 2^-22.9
  return 2.1f * ((float)(7 * s + 13) * 5.9604645e-8f + 1.28e-7f) + 2.1f * 15.0f * 1.1920929e-7f;   // (4 index bits in a key)
}

// K-steps of the kernel that handles (s, k): the resident-A kernel when all A operands fit LDS and
// s <= 16, else the streaming kernel with T rounded up to 16 / 32 / 64 (zero padding is exact);
// 0 = not supported (s > 128)
int mfma_kernel_T(int s, int k) {
  if (s < 1) return 0;
  const int T = (s + 1) / 2;
  const int nkb = (k + 31) / 32;
  const size_t lds = ((size_t)nkb * T * 64 + (size_t)nkb * 64) * sizeof(float);
  if (T <= 8 && lds <= 60 * 1024) return T;
  if (T <= 16) return 16;
  if (T <= 32) return 32;
  if (T <= 64) return 64;
  return 0;
}
static bool mfma_streams(int s, int k) {
  const int T = (s + 1) / 2;
  const int nkb = (k + 31) / 32;
  const size_t lds = ((size_t)nkb * T * 64 + (size_t)nkb * 64) * sizeof(float);
  return !(T <= 8 && lds <= 60 * 1024);
}

bool mfma_assign_supported(int s, int k) {
  if (getenv("GULON_KMEANS_NO_MFMA")) return false;
  return mfma_kernel_T(s, k) != 0;
}

void pack_slice(const float *dX, int n, int ld, int from, int s, int k, PackedSlice &ps, hipStream_t st) {
  ps.n = n; ps.from = from; ps.s = s; ps.T = mfma_kernel_T(s, k);
  ps.split = mfma_split(s, k);
  long long ntile = ((long long)n + 31) / 32;
  ntile = (ntile + 1) / 2 * 2;   // whole pairs
  if (ps.split) {
    const long long nlanes = ntile * 64;
    ps.words = split_compact_words(s);
    const int na = ps.words ? ps.words : 3;
    ps.xq.ensure((size_t)std::max<long long>(nlanes * na * 4, 4));    // uint4 per (tile, piece or word, lane)
    ps.xn.ensure((size_t)std::max<long long>(ntile * 32, 1));
    GULON_UNSUPPORTED(nlanes >= (1ll << 32), "slice of %lld rows: a dispatch carries fewer than 2^32 work-items", (long long)n);
    if (n > 0 && ps.words)
      hipLaunchKernelGGL(pack_slice_compact, dim3((unsigned)ceil_div(nlanes, 256LL)), dim3(256), 0, st, dX, n, ld, from, s, na,
                         nlanes, reinterpret_cast<uint4 *>(ps.xq.p), ps.xn.p);
    else if (n > 0)
      hipLaunchKernelGGL(pack_slice_split, dim3((unsigned)ceil_div(nlanes, 256LL)), dim3(256), 0, st, dX, n, ld, from, s,
                         nlanes, reinterpret_cast<uint4 *>(ps.xq.p), ps.xn.p);
    HIP_CHECK(hipGetLastError());
    return;
  }
  long long total = ntile * ps.T * 64;
  ps.xq.ensure((size_t)std::max<long long>(total, 1));
  GULON_UNSUPPORTED(total >= (1ll << 32), "slice of %lld elements: a dispatch carries fewer than 2^32 work-items", total);
  if (total)
    hipLaunchKernelGGL(pack_slice_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, st, dX, n, ld, from, s, ps.T,
                       total, ps.xq.p);
  HIP_CHECK(hipGetLastError());
}

// Runs the filter.  d_assign receives the answer of every unflagged row; flagged rows are
// appended to ws.flag_rows and *ws.flag_count (device) holds their number.
void assign_mfma_filter(KmeansWorkspace &ws, const PackedSlice &ps, const float *dC, int k, int *d_assign,
                        hipStream_t st) {
  const int s = ps.s, T = ps.T, n = ps.n;
  const int nkb = (k + 31) / 32;
  GULON_REQUIRE(T == mfma_kernel_T(s, k) && ps.split == mfma_split(s, k) && (!ps.split || ps.words == split_compact_words(s)),
                "packed slice was laid out for another kernel (T = %d)", T);
  if (ps.split) {
    const int na = ps.words ? ps.words : 3;
    ws.apack.ensure((size_t)nkb * na * 64 * 4);
    ws.offp.ensure((size_t)nkb * 32);
    ws.cmax2.ensure(1);
    ws.flag_rows.ensure((size_t)std::max(n, 1));
    ws.flag_count.ensure(1);
    HIP_CHECK(hipMemsetAsync(ws.cmax2.p, 0, sizeof(unsigned), st));
    HIP_CHECK(hipMemsetAsync(ws.flag_count.p, 0, sizeof(unsigned), st));
    hipLaunchKernelGGL(pack_centroids_split, dim3(ceil_div(nkb * 64, 256)), dim3(256), 0, st, dC, ws.off.p, k, s, nkb,
                       reinterpret_cast<uint4 *>(ws.apack.p), ws.offp.p, ws.cmax2.p, ps.words);
    const long long npairs = ((long long)n + 63) / 64;
    const size_t lds = (size_t)nkb * (na * 64 * 16 + 64 * 4);
    // (launch size, measured at config 3 with the 32 sub-quantizers' launches side by side: 512 workgroups 19.4 ms per
    // stage, 1024 18.4, 2048 18.0-18.4, 3072 / 4096 the same, 8192 18.7, 16384 19.2)
    int grid = (int)std::min<long long>((npairs + 3) / 4, 256 * 8);
    if (grid < 1) grid = 1;
    auto kern = ps.words == 3 ? assign_bf16<3, true> : ps.words == 4 ? assign_bf16<4, true> : ps.words == 5 ? assign_bf16<5, true>
                                                                                                            : assign_bf16<3, false>;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, reinterpret_cast<const uint4 *>(ps.xq.p), ps.xn.p, n,
                       npairs, reinterpret_cast<const uint4 *>(ws.apack.p), ws.offp.p, nkb, ws.cmax2.p, split_errk(s), d_assign,
                       ws.flag_rows.p, ws.flag_count.p, (float *)nullptr);
    HIP_CHECK(hipGetLastError());
    return;
  }
  ws.apack.ensure((size_t)nkb * T * 64);
  ws.offp.ensure((size_t)nkb * 32);
  ws.cmax2.ensure(1);
  ws.flag_rows.ensure((size_t)std::max(n, 1));
  ws.flag_count.ensure(1);
  HIP_CHECK(hipMemsetAsync(ws.cmax2.p, 0, sizeof(unsigned), st));
  HIP_CHECK(hipMemsetAsync(ws.flag_count.p, 0, sizeof(unsigned), st));
  int nthreads = std::max(nkb * T * 64, nkb * 32);
  hipLaunchKernelGGL(pack_centroids_kernel, dim3(ceil_div(nthreads, 256)), dim3(256), 0, st, dC, ws.off.p, k, s, T,
                     nkb, ws.apack.p, ws.offp.p, ws.cmax2.p);
  const long long npairs = ((long long)n + 63) / 64;
  // band = 2E + key perturbation: 2.1 * (2s+4) * 2^-24 * S  +  2 * 31 ulp * S,  S = |c|max^2 + 2|x||c|max
  // (margins cover sqrt/norm rounding; the second term is the index embedded in 5 mantissa bits)
  const float errk = 2.1f * (float)(2 * s + 4) * 5.9604645e-8f + 2.1f * 31.0f * 1.1920929e-7f;
  size_t lds = ((size_t)nkb * T * 64 + (size_t)nkb * 64) * sizeof(float);
  int grid = (int)std::min<long long>((npairs + 3) / 4, 256 * 8);
  if (grid < 1) grid = 1;
  if (mfma_streams(s, k)) {
    const size_t slds = 2 * ((size_t)T * 64 + 32) * sizeof(float);
    const int sgrid = (int)std::max<long long>(1, std::min<long long>((npairs + 3) / 4, 256 * 2));
#define AS(TT)                                                                                                     \
    hipLaunchKernelGGL(assign_mfma_stream<TT>, dim3(sgrid), dim3(256), slds, st, ps.xq.p, n, npairs, ws.apack.p,    \
                       ws.offp.p, nkb, ws.cmax2.p, errk, d_assign, ws.flag_rows.p, ws.flag_count.p)
    if (T == 16) AS(16); else if (T == 32) AS(32); else AS(64);
#undef AS
    HIP_CHECK(hipGetLastError());
    return;
  }
#define AM(TT)                                                                                                   \
  hipLaunchKernelGGL(assign_mfma<TT>, dim3(grid), dim3(256), lds, st, ps.xq.p, n, npairs, ws.apack.p, ws.offp.p, \
                     nkb, ws.cmax2.p, errk, d_assign, ws.flag_rows.p, ws.flag_count.p)
  switch (T) {
    case 1: AM(1); break;
    case 2: AM(2); break;
    case 3: AM(3); break;
    case 4: AM(4); break;
    case 5: AM(5); break;
    case 6: AM(6); break;
    case 7: AM(7); break;
    case 8: AM(8); break;
    default: GULON_UNSUPPORTED(true, "MFMA assign: sub-dimension %d too large", s);
  }
#undef AM
  HIP_CHECK(hipGetLastError());
}

// Margin of the bf16-split filter's error band on this GPU: 64 random rows x 32 random centroids of sub-dimension s,
// coordinates ~ scale * U(-1, 1) with a few large outliers; d' from the matrix cores (probe mode of assign_bf16) against
// the reference's own arithmetic (offsets[c] - 2 * dot, sequential unfused fp32, KMeans.scala:42-47) computed on the
// host.  Returns max |d' - d| / (band of that row); the filter is sound while this stays below 1/2 (band = 2E).
double selftest_assign_band(int s, unsigned long long seed, float scale) {
  GULON_REQUIRE(mfma_split(s, 32), "the bf16-split filter does not take s = %d", s);
  const int n = 64, k = 32;
  std::vector<float> X((size_t)n * s), C((size_t)k * s), off(k);
  unsigned long long z = seed * 0x9E3779B97F4A7C15ULL + 12345;
  auto rnd = [&]() {
    z ^= z << 13; z ^= z >> 7; z ^= z << 17;
    return (float)((double)(z >> 11) / 9007199254740992.0 * 2.0 - 1.0);
  };
  for (auto &v : X) { v = scale * rnd(); if ((z & 63) == 0) v *= 37.0f; }
  for (auto &v : C) { v = scale * rnd(); if ((z & 127) == 0) v *= 11.0f; }
  for (int c = 0; c < k; c++) {          // KMeans.apply offsets (KMeans.scala:170-186): sequential sum of squares
    float o = 0.f;
    for (int e = 0; e < s; e++) o += C[(size_t)c * s + e] * C[(size_t)c * s + e];
    off[c] = o;
  }
  DevBuf<float> dX, dC, dOff, probe((size_t)n * k);
  dX.upload(X.data(), X.size()); dC.upload(C.data(), C.size()); dOff.upload(off.data(), off.size());
  PackedSlice ps;
  pack_slice(dX.p, n, s, 0, s, k, ps, nullptr);
  const int na = ps.words ? ps.words : 3;
  DevBuf<uint4> ap((size_t)na * 64);
  DevBuf<float> offp(32);
  DevBuf<unsigned> cmax(1), fc(1);
  DevBuf<int> as(n), fr(n);
  HIP_CHECK(hipMemset(cmax.p, 0, 4)); HIP_CHECK(hipMemset(fc.p, 0, 4));
  hipLaunchKernelGGL(pack_centroids_split, dim3(1), dim3(256), 0, 0, dC.p, dOff.p, k, s, 1, ap.p, offp.p, cmax.p, ps.words);
  const size_t lds = (size_t)(na * 64 * 16 + 64 * 4);
  auto kern = ps.words == 3 ? assign_bf16<3, true> : ps.words == 4 ? assign_bf16<4, true> : ps.words == 5 ? assign_bf16<5, true>
                                                                                                          : assign_bf16<3, false>;
  hipLaunchKernelGGL(kern, dim3(1), dim3(256), lds, 0, reinterpret_cast<const uint4 *>(ps.xq.p), ps.xn.p, n, 1ll, ap.p,
                     offp.p, 1, cmax.p, split_errk(s), as.p, fr.p, fc.p, probe.p);
  HIP_CHECK(hipGetLastError());
  std::vector<float> got((size_t)n * k);
  probe.download(got.data(), got.size());
  HIP_CHECK(hipDeviceSynchronize());
  float cmax2 = 0.f;
  for (int c = 0; c < k; c++) cmax2 = std::max(cmax2, off[c]);
  double worst = 0.0;
  for (int r = 0; r < n; r++) {
    float nx = 0.f;
    for (int e = 0; e < s; e++) nx += X[(size_t)r * s + e] * X[(size_t)r * s + e];
    const float band = split_errk(s) * (cmax2 + 2.0f * std::sqrt(nx * cmax2)) + 1e-30f;
    for (int c = 0; c < k; c++) {
      volatile float dot = 0.f;          // the reference's chain, unfused
      for (int e = 0; e < s; e++) { volatile float pr = X[(size_t)r * s + e] * C[(size_t)c * s + e]; dot = dot + pr; }
      volatile float two = 2 * dot;
      const float d = off[c] - two;
      worst = std::max(worst, std::fabs((double)got[(size_t)r * k + c] - (double)d) / (double)band);
    }
  }
  return worst;
}

}  // namespace gulon

#ifdef GULON_TEST_HOOKS
GULON_API int32_t gulon_selftest_assign_band(int32_t s, uint64_t seed, float scale, double *max_error_over_band) {
  return gulon::guarded([&] {
    GULON_REQUIRE(max_error_over_band != nullptr && s >= 1 && s <= 16, "bad arguments");
    *max_error_over_band = gulon::selftest_assign_band(s, seed, scale);
  });
}
#endif
