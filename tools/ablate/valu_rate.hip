// Diagnostic: issue cost (cycles per wave-instruction on one SIMD) of the VALU instructions the operand split can be built from.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
template <int OP>
__global__ void k(unsigned long long* out, float a, float b) {
  float x = a + threadIdx.x, y = b; unsigned h = threadIdx.x; float2 p = make_float2(x, y);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < 64; ++i) {
    if (OP == 0) { REP64(asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0" : "+v"(h) : "v"(x), "v"(y));) }
    if (OP == 1) { REP64(asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x) : "v"(y), "v"(y));) }
    if (OP == 2) { REP64(asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(p) : "v"(p));) }
    if (OP == 3) { REP64(asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h) : "v"(x), "v"(y));) }
    if (OP == 4) { REP64(asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(x) : "v"(h));) }
    if (OP == 5) { REP64(asm volatile("v_pk_mul_f32 %0, %1, %1" : "=v"(p) : "v"(p));) }
    if (OP == 6) { REP64(asm volatile("v_fma_mixhi_f16 %0, %1, %2, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(h) : "v"(x), "v"(y));) }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  if (x == 12345.f && h == 77 && p.x == 3.f) out[1] = 0;
}
int main() {
  unsigned long long* d; CK(hipMalloc(&d, 64 * 8));
  const char* names[] = {"v_fma_mixlo_f16", "v_fma_f32 (dependent)", "v_pk_fma_f32 (dependent)", "v_cvt_pk_f16_f32", "v_cvt_f32_f16", "v_pk_mul_f32", "v_fma_mixhi_f16 (f16 src2)"};
  for (int op = 0; op < 7; ++op) {
    for (int waves = 1; waves <= 4; waves *= 4) {   // one wave, then four waves (one per SIMD)
      switch (op) { case 0: hipLaunchKernelGGL(k<0>, 1, 64 * waves, 0, 0, d, 1.f, 2.f); break; case 1: hipLaunchKernelGGL(k<1>, 1, 64 * waves, 0, 0, d, 1.f, 2.f); break;
        case 2: hipLaunchKernelGGL(k<2>, 1, 64 * waves, 0, 0, d, 1.f, 2.f); break; case 3: hipLaunchKernelGGL(k<3>, 1, 64 * waves, 0, 0, d, 1.f, 2.f); break;
        case 4: hipLaunchKernelGGL(k<4>, 1, 64 * waves, 0, 0, d, 1.f, 2.f); break; case 5: hipLaunchKernelGGL(k<5>, 1, 64 * waves, 0, 0, d, 1.f, 2.f); break;
        default: hipLaunchKernelGGL(k<6>, 1, 64 * waves, 0, 0, d, 1.f, 2.f); }
      CK(hipDeviceSynchronize());
      unsigned long long c; CK(hipMemcpy(&c, d, 8, hipMemcpyDeviceToHost));
      printf("%-28s waves %d: %.2f cycles (s_memtime ticks) per instruction\n", names[op], waves, (double)c / (64.0 * 64.0));
    }
  }
  return 0;
}
