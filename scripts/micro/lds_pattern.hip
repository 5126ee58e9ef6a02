// What does ONE ds_read_b128 wave-instruction cost as a function of its 64 addresses?  (gfx950; the filter's table
// look-ups.)  Every lane reads the 16-byte entry idx[lane] of a 128 KiB LDS array, the same pattern again and again,
// 16 waves per CU so that the LDS pipe is the bound; one launch per pattern, patterns printed with their time so that a
// cost model can be fitted offline (scripts/micro/lds_pattern_fit.py).
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/lds_pattern.hip -o /tmp/lds_pattern && /tmp/lds_pattern > patterns.txt
//   lds_pattern pat|swap [16|8|4]: the address patterns / the lane exchanges, for ds_read_b128 / b64 / b32
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <typename E>   // uint4: ds_read_b128, uint2: ds_read_b64, uint32_t: ds_read_b32; idx in entries of sizeof(E)
__global__ __launch_bounds__(1024) void probe(const int *__restrict__ idx, int iters, uint4 *__restrict__ out) {
  extern __shared__ uint4 lds_raw[];   // 128 KiB
  E *lds = reinterpret_cast<E *>(lds_raw);
  const int tid = threadIdx.x, lane = tid & 63;
  for (int e = tid; e < 8192; e += 1024) lds_raw[e] = make_uint4(e, e * 3, e * 5, e * 7);
  __syncthreads();
  int a = idx[lane];
  uint32_t acc = 0;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 16; u++) {
      asm volatile("" : "+v"(a));
      const E x = lds[a];
      const uint32_t *w = reinterpret_cast<const uint32_t *>(&x);
#pragma unroll
      for (int c = 0; c < (int)(sizeof(E) / 4); c++) acc += w[c];
    }
  }
  out[(size_t)blockIdx.x * 1024 + tid] = make_uint4(acc, 0, 0, 0);
}

int main(int argc, char **argv) {
  const bool swaps = argc > 1 && !strcmp(argv[1], "swap");
  const int width = argc > 2 ? atoi(argv[2]) : 16;   // bytes per entry: 16 (default), 8 or 4
  const int blocks = 256, iters = swaps ? 150 : 400;
  int *d_idx; uint4 *o;
  CK(hipMalloc(&d_idx, 64 * 4)); CK(hipMalloc(&o, (size_t)blocks * 1024 * sizeof(uint4)));
  auto kern = width == 16 ? probe<uint4> : width == 8 ? probe<uint2> : probe<uint32_t>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
  std::vector<std::vector<int>> pats;
  std::vector<const char *> names;
  auto add = [&](const char *nm, std::vector<int> p) { pats.push_back(p); names.push_back(nm); };
  std::mt19937 rng(7);
  std::vector<int> p(64);
  if (swaps) {
    // which lanes are looked at together?  Linear addresses (conflict-free) with the entries of lanes i and j exchanged:
    // nothing changes if i and j are served in the same group, both groups get a two-way conflict otherwise
    static char nm[64 * 64][16];
    for (int i = 0; i < 64; i++)
      for (int j = i + 1; j < 64; j++) {
        if ((i & 15) == (j & 15)) continue;
        for (int l = 0; l < 64; l++) p[l] = l;
        std::swap(p[i], p[j]);
        snprintf(nm[i * 64 + j], 16, "swap_%d_%d", i, j);
        add(nm[i * 64 + j], p);
      }
  } else {
  for (int s : {1, 2, 3, 4, 5, 8, 16, 17, 32, 64, 128}) { for (int l = 0; l < 64; l++) p[l] = (l * s) % 8192; add("stride", p); }
  // lane groups of G consecutive lanes share a bank column (entry % 16 == group id % 16), rows differ
  for (int G : {2, 4, 8, 16, 32}) { for (int l = 0; l < 64; l++) p[l] = ((l % G) * 16 + (l / G)) % 8192 + 1024; add("groupcol", p); }
  // lanes l and l + D collide on the column, all else linear
  for (int D : {1, 2, 4, 8, 16, 32}) { for (int l = 0; l < 64; l++) p[l] = l; for (int l = 0; l < 64; l++) if ((l / D) & 1) p[l] = p[l - D] + 16 * (l + 1); add("pairD", p); }
  // random entries of a 256-entry table (the filter's look-up), of a 16-entry table, of the whole array
  for (int r = 0; r < 150; r++) { for (int l = 0; l < 64; l++) p[l] = rng() % 256; add("rand256", p); }
  for (int r = 0; r < 20; r++) { for (int l = 0; l < 64; l++) p[l] = rng() % 8192; add("rand8192", p); }
  // random rows, structured columns
  for (int M : {16, 8, 6, 4, 2, 1}) for (int r = 0; r < 5; r++) { for (int l = 0; l < 64; l++) p[l] = (rng() % 16) * 16 + (l % M); add("randrow_colmod", p); }
  // random rows, column = permutation of lane within groups of 16 / 8 / 32 lanes
  for (int r = 0; r < 10; r++) { for (int l = 0; l < 64; l++) p[l] = (rng() % 16) * 16 + ((l * 5 + 3) % 16); add("randrow_perm16", p); }
  for (int r = 0; r < 10; r++) { for (int l = 0; l < 64; l++) p[l] = (rng() % 16) * 16 + ((l >> 2) % 16); add("randrow_colq", p); }   // quads share a column
  for (int r = 0; r < 10; r++) { for (int l = 0; l < 64; l++) p[l] = (rng() % 16) * 16 + (l & 7) * 2; add("randrow_evencol8", p); }    // 8 lanes -> 8 even columns
  for (int r = 0; r < 10; r++) { for (int l = 0; l < 64; l++) p[l] = (rng() % 16) * 16 + (l & 7) + 8 * ((l >> 5) & 1); add("randrow_col8_hi", p); }
  }
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (size_t i = 0; i < pats.size(); i++) {
    CK(hipMemcpy(d_idx, pats[i].data(), 64 * 4, hipMemcpyHostToDevice));
    float best = 1e9f;
    for (int rep = 0; rep < 3; rep++) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(kern, dim3(blocks), dim3(1024), 131072, 0, d_idx, iters, o);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep > 0 && ms < best) best = ms;
    }
    const double winst = 16.0 * iters * 16;
    printf("%s %.2f", names[i], best * 1e6 / winst * 2.4);
    for (int l = 0; l < 64; l++) printf(" %d", pats[i][l]);
    printf("\n");
  }
  return 0;
}
