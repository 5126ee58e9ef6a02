// single-wave instruction latency / issue probes for gfx950 (timing experiments; not part of the library)
//   hipcc --offload-arch=gfx950 -O3 -o build/ubench/latency scripts/ubench_latency.hip && build/ubench/latency
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define N 4096
#define REP16(x) x x x x x x x x x x x x x x x x
template <int MODE>
__global__ void probe(float *out, unsigned long long *cyc, float seed) {
  float a = seed + threadIdx.x, b = seed * 0.5f, c = 1.0f + seed;
  f32x2 pa = {a, a + 1}, pb = {b, b + 1}, pc = {c, c};
  f32x2 qa = pa + 2.f, qb = pb + 3.f;
  float d = a + 3.f, e = b + 5.f;
  int m = (int)seed;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < N / 16; i++) {
    if (MODE == 0) { REP16(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));) }          // dependent fma
    if (MODE == 1) { REP16(asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(pa) : "v"(pb), "v"(pc));) }     // dependent pk fma
    if (MODE == 2) { REP16(asm volatile("v_pk_fma_f32 %0, %0, %2, %3\n v_pk_fma_f32 %1, %1, %2, %3" : "+v"(pa), "+v"(qa) : "v"(pb), "v"(pc));) }  // two chains
    if (MODE == 3) { REP16(asm volatile("v_rcp_f32 %0, %0" : "+v"(a));) }                                     // dependent rcp
    if (MODE == 4) { REP16(asm volatile("v_rcp_f32 %0, %1\n v_fma_f32 %2, %2, %1, %1" : "=v"(d), "+v"(b), "+v"(a));) }  // rcp + independent fma
    if (MODE == 5) { REP16(asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(pa) : "v"(pb));) }                  // dependent pk add
    if (MODE == 6) { REP16(asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(pa) : "v"(pc));) }                  // dependent pk mul
    if (MODE == 7) { REP16(asm volatile("v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3" : "+v"(a), "+v"(d) : "v"(b), "v"(c));) }  // two fp32 chains
    if (MODE == 8) { REP16(asm volatile("v_cmp_lt_f32 vcc, %0, %1\n s_and_saveexec_b64 s[10:11], vcc\n v_fma_f32 %0, %0, %1, %2\n s_or_b64 exec, exec, s[10:11]" : "+v"(a) : "v"(b), "v"(c) : "vcc", "s10", "s11");) }  // masked step skeleton
    if (MODE == 9) { REP16(asm volatile("v_fma_f32 %0, %0, %3, %4\n v_fma_f32 %1, %1, %3, %4\n v_fma_f32 %2, %2, %3, %4" : "+v"(a), "+v"(d), "+v"(e) : "v"(b), "v"(c));) }  // three chains
    if (MODE == 10) { REP16(asm volatile("v_frexp_exp_i32_f32 %0, %1" : "=v"(m) : "v"(a)); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));) }
    if (MODE == 11) { REP16(asm volatile("v_pk_fma_f32 %0, %0, %1, %2\n s_nop 0" : "+v"(pa) : "v"(pb), "v"(pc));) }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = a + b + c + d + e + pa.x + pa.y + qa.x + qb.y + (float)m;
  if (threadIdx.x == 0) *cyc = t1 - t0;
}
int main() {
  float *out; unsigned long long *cyc, h;
  hipMalloc(&out, 4096); hipMalloc(&cyc, 8);
  const char *names[] = {"dependent v_fma_f32", "dependent v_pk_fma_f32", "two v_pk_fma_f32 chains (per pair)", "dependent v_rcp_f32",
                         "v_rcp_f32 + independent fma (per pair)", "dependent v_pk_add_f32", "dependent v_pk_mul_f32",
                         "two v_fma_f32 chains (per pair)", "cmp+saveexec+fma+restore (per group of 4)", "three v_fma_f32 chains (per triple)",
                         "frexp + dependent fma (per pair)", "dependent pk_fma + s_nop (per pair)"};
#define RUN(M) hipLaunchKernelGGL(probe<M>, dim3(1), dim3(64), 0, 0, out, cyc, 1.0f); hipLaunchKernelGGL(probe<M>, dim3(1), dim3(64), 0, 0, out, cyc, 1.0f); \
  hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); printf("%-46s %7.2f cycles per unit\n", names[M], (double)h / N);
  RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) RUN(10) RUN(11)
  return 0;
}
