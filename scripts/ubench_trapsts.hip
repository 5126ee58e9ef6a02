// do the sticky exception bits of TRAPSTS see a packed fp32 result that underflows?  (timing/behaviour probe; not part of the library)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
__global__ void probe(const float *in, unsigned *out) {
  // in[0..1] = a pair, in[2] = y
  f32x2 a = {in[0], in[1]};
  const float y = in[2];
  unsigned before, after_clear, after;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_TRAPSTS, 0, 9)" : "=s"(before));
  asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_TRAPSTS, 0, 9), 0");
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_TRAPSTS, 0, 9)" : "=s"(after_clear));
  f32x2 q;
  asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]\n s_nop 7\n s_nop 7" : "=v"(q) : "v"(a), "v"(f32x2{y, y}));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_TRAPSTS, 0, 9)" : "=s"(after));
  if (threadIdx.x == 0) { out[0] = before; out[1] = after_clear; out[2] = after; out[3] = __float_as_uint(q.x); out[4] = __float_as_uint(q.y); }
}
int main() {
  float *in; unsigned *out;
  hipMalloc(&in, 16); hipMalloc(&out, 64);
  const float cases[][3] = {{1.5f, 2.5f, 0.3f}, {1e-30f, 1.0f, 1e-15f}, {1.0f, 1e-30f, 1e-15f}, {3e38f, 1.0f, 16.f}, {1e-42f, 1.f, 1.f}, {0.f, 0.f, 0.3f}};
  for (auto &cs : cases) {
    hipMemcpy(in, cs, 12, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, in, out);
    unsigned h[5]; hipMemcpy(h, out, 20, hipMemcpyDeviceToHost);
    printf("a=(%g,%g) y=%g: EXCP before %03x cleared %03x after %03x  q bits %08x %08x\n", cs[0], cs[1], cs[2], h[0], h[1], h[2], h[3], h[4]);
  }
  return 0;
}
