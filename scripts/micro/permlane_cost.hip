// Cost of v_permlane32_swap against plain VALU and against ds_bpermute (the scan epilogue of the k-means assign filter
// swaps 16 accumulator pairs per 32-centroid block):
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/permlane_cost.hip -o /tmp/permlane_cost && /tmp/permlane_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
template <int MODE>   // 0: 32 v_add per iteration; 1: 32 v_permlane32_swap pairs (16 instr); 2: 32 ds_bpermute
__global__ __launch_bounds__(256) void k(float *out, int iters) {
  float v[32];
  for (int i = 0; i < 32; i++) v[i] = threadIdx.x * 0.5f + i;
  for (int it = 0; it < iters; it++) {
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < 32; i++) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[i]) : "v"(1.0f));
    } else if (MODE == 1) {
#pragma unroll
      for (int i = 0; i < 32; i += 2) {
        auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[i]), __float_as_uint(v[i + 1]), false, false);
        v[i] = __uint_as_float(r[0]); v[i + 1] = __uint_as_float(r[1]);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 32; i++) v[i] = __shfl_xor(v[i], 32);
    }
  }
  float s = 0;
  for (int i = 0; i < 32; i++) s += v[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
  float *out; CK(hipMalloc(&out, 2048 * 256 * 4));
  const int iters = 4000;
  for (int mode = 0; mode < 3; mode++) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms = 0;
    for (int rep = 0; rep < 2; rep++) {
      CK(hipEventRecord(e0));
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(2048), dim3(256), 0, 0, out, iters);
      if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(2048), dim3(256), 0, 0, out, iters);
      if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(2048), dim3(256), 0, 0, out, iters);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms, e0, e1));
    }
    // wave-instructions per SIMD: 2048 blocks * 4 waves / 1024 SIMDs * iters * n
    const double n = mode == 1 ? 16 : 32;
    const double per = ms * 1e-3 * 2.4e9 / (8.0 * iters * n);
    printf("mode %d: %.2f ms -> %.1f cycles per wave-instruction per SIMD (at 2.4 GHz)\n", mode, ms, per);
  }
  return 0;
}
