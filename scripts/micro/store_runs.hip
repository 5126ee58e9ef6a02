// What a scattered-run store stream costs (the k-means placement, kmeans.hip sort_place*): every workgroup writes
// 40 KiB as 256 runs of RUN bytes, run c of workgroup b at  base[c] + b * RUN  (neighbouring workgroups continue
// each other's runs, as consecutive chunks do), from registers, nothing else in the kernel.
//   mode 0: 8-byte stores (dwordx2), lanes walk the runs in order            (what sort_place does)
//   mode 1: 16-byte stores (dwordx4), runs 16-byte aligned (RUN = 160 or 192)
//   mode 2: 8-byte stores, ONE contiguous 40 KiB block per workgroup
//   mode 3: 16-byte stores, one contiguous block
//   mode 4: 4-byte stores, runs
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/store_runs.hip -o /tmp/store_runs && /tmp/store_runs
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int RUN>
__global__ __launch_bounds__(256) void k(char *out, long long cluster_stride, int nwg_per_xcd) {
  // XCD-contiguous block order, as the placement uses it
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const long long b = (long long)xcd * nwg_per_xcd + slot;
  const int tid = threadIdx.x;
  constexpr int BYTES = 256 * RUN;
  if (MODE == 0 || MODE == 4 || MODE == 1) {
    constexpr int W = MODE == 0 ? 8 : MODE == 4 ? 4 : 16;
    constexpr int UPR = RUN / W;                 // units per run
#pragma unroll
    for (int it = 0; it < BYTES / W / 256; it++) {
      const int u = tid + 256 * it;
      const int c = u / UPR, part = u - c * UPR;
      char *p = out + (long long)c * cluster_stride + b * RUN + (long long)part * W;
      if (W == 8) *reinterpret_cast<f32x2 *>(p) = f32x2{(float)u, 1.f};
      else if (W == 4) *reinterpret_cast<float *>(p) = (float)u;
      else *reinterpret_cast<f32x4 *>(p) = f32x4{(float)u, 1.f, 2.f, 3.f};
    }
  } else {
    constexpr int W = MODE == 2 ? 8 : 16;
#pragma unroll
    for (int it = 0; it < BYTES / W / 256; it++) {
      const int u = tid + 256 * it;
      char *p = out + b * BYTES + (long long)u * W;
      if (W == 8) *reinterpret_cast<f32x2 *>(p) = f32x2{(float)u, 1.f};
      else *reinterpret_cast<f32x4 *>(p) = f32x4{(float)u, 1.f, 2.f, 3.f};
    }
  }
}

template <int MODE, int RUN>
void run(char *out, int nwg) {
  const long long cluster_stride = (long long)nwg * RUN;   // every cluster's bucket holds one run per workgroup
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float ms = 0;
  for (int rep = 0; rep < 3; rep++) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<MODE, RUN>), dim3(nwg), dim3(256), 0, 0, out, cluster_stride, nwg / 8);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
  }
  const double bytes = (double)nwg * 256 * RUN;
  printf("mode %d run %3d: %.3f ms  %.2f TB/s  (%.0f cycles per wave-store at 2.4 GHz, 256 CUs)\n", MODE, RUN, ms,
         bytes / ms * 1e-9, ms * 1e-3 * 2.4e9 * 256 / ((double)nwg * 4 * (256 * RUN / (MODE == 0 || MODE == 2 ? 8 : MODE == 4 ? 4 : 16) / 256)));
}

int main() {
  const int nwg = 8 * 4096;                       // 32768 workgroups x 40 KiB = 1.3 GB
  char *out; CK(hipMalloc(&out, (size_t)nwg * 256 * 320 + (1 << 20)));   // the largest RUN below
  run<0, 160>(out, nwg);
  run<4, 160>(out, nwg);
  run<1, 160>(out, nwg);
  run<1, 192>(out, nwg);
  run<2, 160>(out, nwg);
  run<3, 160>(out, nwg);
  run<0, 320>(out, nwg);
  run<1, 320>(out, nwg);
  run<1, 128>(out, nwg);      // whole aligned lines: what a placement that carries partial lines over would write
  run<0, 128>(out, nwg);
  run<1, 256>(out, nwg);
  run<1, 64>(out, nwg);
  return 0;
}
