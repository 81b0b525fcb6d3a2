// v_sin_f32 on UNREDUCED arguments: is the hardware's own range reduction exact inside its documented domain (|x| <= 256 revolutions)?
//   hipcc --offload-arch=gfx950 -O3 vsin_range.hip -o vsin_range && ./vsin_range
// Input x in revolutions (an exact float); truth = sin(2 pi (x - rint(x))) in double (x - rint(x) is exact in double).
// Prints per range: max |v_sin(x) - truth|, max |v_sin(fract(x)) - truth|, how many results differ between the two forms, and what
// the instruction returns beyond the domain.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>

__global__ void eval(const float* x, float* direct, float* reduced, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float a, b, fr;
  asm volatile("v_sin_f32 %0, %1" : "=v"(a) : "v"(x[i]));
  asm volatile("v_fract_f32 %0, %1" : "=v"(fr) : "v"(x[i]));
  asm volatile("v_sin_f32 %0, %1" : "=v"(b) : "v"(fr));
  direct[i] = a; reduced[i] = b;
}

int main() {
  const int n = 1 << 22;
  const float ranges[] = {0.5f, 2.f, 8.f, 32.f, 128.f, 255.9f, 300.f, 1000.f, 1e6f};
  float *dx, *da, *db;
  (void)hipMalloc(&dx, n * 4); (void)hipMalloc(&da, n * 4); (void)hipMalloc(&db, n * 4);
  std::vector<float> x(n), a(n), b(n);
  const double two_pi = 6.283185307179586476925;
  for (float R : ranges) {
    unsigned s = 777u;
    for (int i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; x[i] = ((s >> 8) * (1.f / 8388608.f) - 1.f) * R; }
    (void)hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
    eval<<<n / 256, 256>>>(dx, da, db, n);
    (void)hipMemcpy(a.data(), da, n * 4, hipMemcpyDeviceToHost);
    (void)hipMemcpy(b.data(), db, n * 4, hipMemcpyDeviceToHost);
    double ea = 0, eb = 0; int differ = 0, zeros = 0; float amax = 0;
    for (int i = 0; i < n; ++i) {
      const double xd = x[i], t = std::sin(two_pi * (xd - std::rint(xd)));
      ea = std::fmax(ea, std::fabs(a[i] - t)); eb = std::fmax(eb, std::fabs(b[i] - t));
      differ += a[i] != b[i]; zeros += a[i] == 0.f; amax = std::fmax(amax, std::fabs(a[i]));
    }
    printf("|x| <= %9.1f rev: v_sin(x) max err %.3e | v_sin(fract x) max err %.3e | results differ %d of %d | direct == 0: %d, max |direct| %.9g\n",
           R, ea, eb, differ, n, zeros, amax);
  }
  return 0;
}
