// LDS random-gather microbenchmark (gfx950): cycles per ds_read_b128 wave-instruction when
//   mode 1: every lane reads its own random 16-B entry               (the filter's pattern)
//   mode 2: lane pairs read the two halves of one random 32-B entry
//   mode 4: lane quads read the four quarters of one random 64-B entry
//   mode 16: 16 lanes read one random 256-B row (conflict-free reference)
//   mode 0: random row per lane, column = lane % 16 (distinct bank groups inside each 16-lane group)
//   modes 102 / 103 / 104: random row per lane, column = lane % 8 / % 6 / % 4: at most 2 / 3 / 4 lanes of a 16-lane
//     group in one bank group (different rows) -- the cost model "a pass of 16 lanes takes max(2, largest load) cycles"
// build: hipcc --offload-arch=gfx950 -O3 -o lds_gather lds_gather.hip ; run: ./lds_gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int SHARE>
__global__ __launch_bounds__(1024) void gather(const uint4 *__restrict__ codes, int iters, uint4 *__restrict__ out,
                                               unsigned long long *__restrict__ cycles) {
  extern __shared__ uint4 lds[];   // 8192 entries of 16 B = 128 KiB
  const int tid = threadIdx.x, lane = tid & 63;
  for (int e = tid; e < 8192; e += 1024) lds[e] = make_uint4(e, e * 3, e * 5, e * 7);
  __syncthreads();
  uint4 acc = make_uint4(0, 0, 0, 0);
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
    uint4 w = codes[((size_t)(blockIdx.x & 7) * 64 + (it & 63)) * 1024 + tid];   // small, cache-resident code stream
    const uint32_t ws[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
    for (int b = 0; b < 16; b++) {
      uint32_t c = (ws[b >> 2] >> (8 * (b & 3))) & 0xFF;
      if (SHARE > 1) {
        // the code of the first lane of each SHARE-group (SHARE = 2, 4: DPP quad_perm; 16: row broadcast via readlane-free shuffle)
        if (SHARE == 2) c = __builtin_amdgcn_mov_dpp((int)c, 0xA0, 0xf, 0xf, true);        // [0,0,2,2]
        else if (SHARE == 4) c = __builtin_amdgcn_mov_dpp((int)c, 0x00, 0xf, 0xf, true);   // [0,0,0,0]
        else c = __builtin_amdgcn_readlane((int)c, 0) + (lane >> 4);   // one row per 16-lane group, no LDS traffic
      }
      // 256 "codes" per quantizer slot, entry = SHARE * 16 B; 8192 / 16 = 512 entries of 16 B per b
      const int per_b = 512 / (SHARE == 0 ? 1 : SHARE);   // distinct entries available per b (LDS capacity)
      // SHARE == 0: every lane its own random 256-B row, but the 16 lanes of a group in 16 different
      // 16-B columns (conflict-free if the bank is address bits [7:4] only)
      const int idx = SHARE > 100 ? b * 512 + (c % 32) * 16 + (lane % (SHARE == 102 ? 8 : SHARE == 103 ? 6 : 4)) : SHARE == 0 ? b * 512 + (c % 32) * 16 + (lane & 15) : b * 512 + (c % per_b) * SHARE + (lane % SHARE);
      const uint4 x = lds[idx];
      acc.x += x.x; acc.y ^= x.y; acc.z += x.z; acc.w ^= x.w;
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  out[(size_t)blockIdx.x * 1024 + tid] = acc;
  if (tid == 0) cycles[blockIdx.x] = t1 - t0;
}

int main() {
  const int blocks = 256, iters = 2000;
  std::vector<uint4> h((size_t)blocks * iters * 1024);
  srand(1);
  for (auto &v : h) v = make_uint4(rand() * 65537u ^ rand(), rand() * 65537u ^ rand(), rand() * 65537u ^ rand(), rand() * 65537u ^ rand());
  uint4 *d, *o; unsigned long long *cy;
  hipMalloc(&d, h.size() * sizeof(uint4)); hipMalloc(&o, (size_t)blocks * 1024 * sizeof(uint4)); hipMalloc(&cy, blocks * 8);
  hipMemcpy(d, h.data(), h.size() * sizeof(uint4), hipMemcpyHostToDevice);
  auto run = [&](auto kern, const char *name) {
    hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(1024), 131072, 0, d, iters, o, cy);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(1024), 131072, 0, d, iters, o, cy);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // wave-instructions per CU: 16 waves * iters * 16
    double winst = 16.0 * iters * 16;
    printf("%-8s %.3f ms  -> %.1f ns per wave-instruction per CU = %.1f cycles at 2.4 GHz\n", name, ms, ms * 1e6 / winst,
           ms * 1e6 / winst * 2.4);
  };
  run(gather<1>, "own16B");
  run(gather<0>, "owncol");
  run(gather<102>, "load2");
  run(gather<103>, "load3");
  run(gather<104>, "load4");
  run(gather<2>, "pair32B");
  run(gather<4>, "quad64B");
  run(gather<16>, "row256B");
  return 0;
}
