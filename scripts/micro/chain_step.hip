// Cycles per step of the running-mean recurrence p += RN((x - p) / n) in its three-operation quotient form
// (kmeans.hip, update_chains*), ONE wave per SIMD, no memory traffic in the loop -- what a chain costs when nothing
// but its own dependent operations is in the way:
//   mode 0: one chain per lane, scalar fp32 (5 dependent operations per step)
//   mode 1: two chains per lane on the packed pipe (v_pk_add/mul/fma_f32)
//   mode 2: two chains per lane, scalar fp32, interleaved (10 operations, two independent chains)
//   mode 3: four chains per lane, scalar fp32, interleaved
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off scripts/micro/chain_step.hip -o /tmp/chain_step && /tmp/chain_step
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float q3(float a, float nf, float y) {
  const float q0 = a * y;
  const float r = __builtin_fmaf(-nf, q0, a);
  return __builtin_fmaf(r, y, q0);
}

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, const float *__restrict__ rcp, int steps, unsigned long long *cyc) {
  const float x0 = threadIdx.x * 0.37f + 1.0f, x1 = x0 * 1.7f, x2 = x0 * 0.3f, x3 = x0 * 2.9f;
  float p0 = 0, p1 = 0, p2 = 0, p3 = 0;
  f32x2 pp = {0.f, 0.f};
  const f32x2 xx = {x0, x1};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  float nf = 1.0f;
  for (int i = 0; i < steps; i += 8) {
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const float y = rcp[i + u];
      if (MODE == 0) {
        p0 = p0 + q3(x0 - p0, nf, y);
      } else if (MODE == 1) {
        const f32x2 a = xx - pp;
        const f32x2 y2 = {y, y}, nn = {-nf, -nf};
        const f32x2 q0 = a * y2;
        const f32x2 r = __builtin_elementwise_fma(nn, q0, a);
        pp = pp + __builtin_elementwise_fma(r, y2, q0);
      } else if (MODE == 2) {
        p0 = p0 + q3(x0 - p0, nf, y);
        p1 = p1 + q3(x1 - p1, nf, y);
      } else {
        p0 = p0 + q3(x0 - p0, nf, y);
        p1 = p1 + q3(x1 - p1, nf, y);
        p2 = p2 + q3(x2 - p2, nf, y);
        p3 = p3 + q3(x3 - p3, nf, y);
      }
      nf += 1.0f;
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = p0 + p1 + p2 + p3 + pp.x + pp.y;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
  const int steps = 1 << 16, blocks = 1024;
  float *out, *rcp; unsigned long long *cyc;
  CK(hipMalloc(&out, blocks * 256 * 4)); CK(hipMalloc(&rcp, steps * 4)); CK(hipMalloc(&cyc, blocks * 8));
  float *h = (float *)malloc(steps * 4);
  for (int i = 0; i < steps; i++) h[i] = 1.0f / (float)(i + 1);
  CK(hipMemcpy(rcp, h, steps * 4, hipMemcpyHostToDevice));
  unsigned long long *hc = (unsigned long long *)malloc(blocks * 8);
  for (int mode = 0; mode < 4; mode++) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms = 0;
    for (int rep = 0; rep < 2; rep++) {
      CK(hipEventRecord(e0));
      // 40 KiB of LDS per one-wave workgroup: four per CU, one per SIMD
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(64), 40 * 1024, 0, out, rcp, steps, cyc);
      if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(64), 40 * 1024, 0, out, rcp, steps, cyc);
      if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(64), 40 * 1024, 0, out, rcp, steps, cyc);
      if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(64), 40 * 1024, 0, out, rcp, steps, cyc);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms, e0, e1));
    }
    CK(hipMemcpy(hc, cyc, blocks * 8, hipMemcpyDeviceToHost));
    double avg = 0;
    for (int b = 0; b < blocks; b++) avg += (double)hc[b];
    avg /= blocks;
    printf("mode %d: %.3f ms, %.1f s_memtime ticks per step (%.1f ns per step)\n", mode, ms, avg / steps, ms * 1e6 / steps);
  }
  // Where do the waves of ONE workgroup go?  256 blocks of 256 threads (four waves, 150 KiB of LDS: one block per CU)
  // against the 1024 one-wave blocks above: the same per-step time means one wave per SIMD in both.
  {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms = 0;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    for (int rep = 0; rep < 2; rep++) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 150 * 1024, 0, out, rcp, steps, cyc);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms, e0, e1));
    }
    CK(hipMemcpy(hc, cyc, 256 * 8, hipMemcpyDeviceToHost));
    double avg = 0;
    for (int b = 0; b < 256; b++) avg += (double)hc[b];
    printf("mode 1, four-wave workgroups (one per CU): %.3f ms, %.1f ticks per step\n", ms, avg / 256 / steps);
  }
  return 0;
}
