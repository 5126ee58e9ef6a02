// Microbenchmarks behind the bf16-split assign filter (kmeans_mfma.hip):
//  1. operand / result layout of v_mfma_f32_32x32x16_bf16 and exactness of its products;
//  2. does a bf16 MFMA stream overlap with a VALU stream on gfx950 (the f32 MFMA does not)?
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/bf16_mfma.hip -o /tmp/bf16_mfma && /tmp/bf16_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

static unsigned short f2bf(float f) { unsigned u; memcpy(&u, &f, 4); return (unsigned short)(u >> 16); }   // exact for short values
static float bf2f(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }

__global__ void layout_kernel(const unsigned short *A /*[32][16]*/, const unsigned short *B /*[32][16]*/, float *C /*[32][32]*/) {
  const int l = threadIdx.x;
  uint4 ua, ub;
  unsigned short ta[8], tb[8];
  for (int e = 0; e < 8; e++) { ta[e] = A[(l % 32) * 16 + 8 * (l / 32) + e]; tb[e] = B[(l % 32) * 16 + 8 * (l / 32) + e]; }
  memcpy(&ua, ta, 16); memcpy(&ub, tb, 16);
  f32x16 c;
  for (int i = 0; i < 16; i++) c[i] = 0.f;
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ua), __builtin_bit_cast(bf16x8, ub), c, 0, 0, 0);
  // assumed: c[r] = C[row = 8 (r / 4) + 4 (l / 32) + r % 4][col = l % 32]
  for (int r = 0; r < 16; r++) C[(8 * (r / 4) + 4 * (l / 32) + r % 4) * 32 + l % 32] = c[r];
}

template <int MODE>   // 1: MFMA only, 2: VALU only, 3: both interleaved
__global__ __launch_bounds__(256) void overlap_kernel(const uint4 *in, float *out, int iters) {
  const int l = threadIdx.x;
  uint4 a = in[l], b = in[l + 256];
  f32x16 c0, c1;
  for (int i = 0; i < 16; i++) { c0[i] = 0.f; c1[i] = 0.f; }
  float pmin = 3.0e38f, band = 3.0e38f, key = __uint_as_float(a.x);
  unsigned salt = b.y;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int g = 0; g < 12; g++) {
      if (MODE & 1) {
        if (g & 1) c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c1, 0, 0, 0);
        else c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c0, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (MODE & 2) {
#pragma unroll
        for (int v = 0; v < 3; v++) {   // 3 x 4 = 12 VALU per MFMA slot: the scan epilogue's mix
          const float k0 = __uint_as_float((__float_as_uint(key) & ~31u) | (salt & 31u));
          float q0;
          asm volatile("v_min_f32 %0, %1, %2" : "=v"(q0) : "v"(pmin), "v"(k0));
          const float d0 = k0 - pmin;
          asm volatile("v_min3_f32 %0, %1, |%2|, |%3|" : "=v"(band) : "v"(band), "v"(d0), "v"(d0));
          pmin = q0;
          key = key * 1.0001f;
          salt += 7u;
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  float s = pmin + band + key;
  for (int i = 0; i < 16; i++) s += c0[i] + c1[i];
  out[blockIdx.x * 256 + l] = s;
}

int main() {
  // ---- 1. layout
  std::vector<unsigned short> A(32 * 16), B(32 * 16);
  for (int i = 0; i < 32 * 16; i++) { A[i] = f2bf((float)((i * 7) % 13 - 6)); B[i] = f2bf((float)((i * 5) % 11 - 5) * 0.5f); }
  unsigned short *dA, *dB; float *dC;
  CK(hipMalloc(&dA, 1024)); CK(hipMalloc(&dB, 1024)); CK(hipMalloc(&dC, 4096));
  CK(hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 1024, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(layout_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dC);
  std::vector<float> C(1024);
  CK(hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost));
  int bad = 0;
  for (int i = 0; i < 32; i++) for (int j = 0; j < 32; j++) {
    float ref = 0.f;
    for (int k = 0; k < 16; k++) ref += bf2f(A[i * 16 + k]) * bf2f(B[j * 16 + k]);
    if (ref != C[i * 32 + j]) bad++;
  }
  printf("layout: C[i][j] = sum_k A[i][k] B[j][k] with A row = lane %% 32, k = 8 (lane / 32) + e: %d of 1024 entries differ\n", bad);
  // ---- 2. overlap
  uint4 *din; float *dout;
  CK(hipMalloc(&din, 512 * 16)); CK(hipMemset(din, 0x3c, 512 * 16)); CK(hipMalloc(&dout, 2048 * 256 * 4));
  const int iters = 2000;
  auto run = [&](int mode) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; rep++) {
      CK(hipEventRecord(e0));
      if (mode == 1) hipLaunchKernelGGL(overlap_kernel<1>, dim3(2048), dim3(256), 0, 0, din, dout, iters);
      if (mode == 2) hipLaunchKernelGGL(overlap_kernel<2>, dim3(2048), dim3(256), 0, 0, din, dout, iters);
      if (mode == 3) hipLaunchKernelGGL(overlap_kernel<3>, dim3(2048), dim3(256), 0, 0, din, dout, iters);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    }
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms;
  };
  const float m1 = run(1), m2 = run(2), m3 = run(3);
  printf("12 bf16 MFMAs per iteration: %.2f ms; 144 VALU per iteration: %.2f ms; interleaved: %.2f ms (sum %.2f, max %.2f)\n",
         m1, m2, m3, m1 + m2, m1 > m2 ? m1 : m2);
  return 0;
}
