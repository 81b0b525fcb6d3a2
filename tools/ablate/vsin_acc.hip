// Accuracy of the hardware sine (v_sin_f32, argument in revolutions) against fp64, for the SIREN epilogue:
//   hipcc --offload-arch=gfx950 -O3 vsin_acc.hip -o vsin_acc && ./vsin_acc
// Prints, per argument range, the maximum absolute error of
//   (a) v_sin_f32(fma(x, 1/2pi, 0))                       -- the hardware path
//   (b) the exact-revolution polynomial (k = rint(x/pi), f = x/pi - k, f * P(f^2))
//   (c) correctly rounded float sin (the reference's own arithmetic)
// all against sin((double)x).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>

__device__ __forceinline__ float hw_sin_rev(float r) {
  float s;
  asm volatile("v_sin_f32 %0, %1" : "=v"(s) : "v"(r));
  return s;
}
__device__ __forceinline__ float hw_cos_rev(float r) {
  float s;
  asm volatile("v_cos_f32 %0, %1" : "=v"(s) : "v"(r));
  return s;
}

__device__ __forceinline__ float poly_sin_pi(float u) {   // sin(pi u)
  const float t = u + 12582912.f;
  const float kf = t - 12582912.f;
  const float f = u - kf;                                  // exact, |f| <= 0.5
  const float f2 = f * f;
  float q = fmaf(f2, -0.00737043094571435f, 0.08214588661112823f);
  q = fmaf(q, f2, -0.5992645293207921f);
  q = fmaf(q, f2, 2.550164039877345f);
  q = fmaf(q, f2, -5.16771278004997f);
  q = fmaf(q, f2, 3.141592653589793f);
  const float s = f * q;
  return __uint_as_float((__float_as_uint(t) << 31) ^ __float_as_uint(s));
}

__global__ void eval(const float* x, float* hw, float* hwc, float* pl, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float r = x[i] * 0.15915494309189535f;
  hw[i] = hw_sin_rev(r);
  hwc[i] = hw_cos_rev(r);
  pl[i] = poly_sin_pi(x[i] * 0.3183098861837907f);
}

int main() {
  const int n = 1 << 22;
  const float ranges[] = {1.f, 4.f, 8.f, 30.f, 100.f, 600.f};
  float *dx, *dh, *dc, *dp;
  hipMalloc(&dx, n * 4); hipMalloc(&dh, n * 4); hipMalloc(&dc, n * 4); hipMalloc(&dp, n * 4);
  std::vector<float> x(n), h(n), c(n), p(n);
  for (float R : ranges) {
    unsigned s = 12345u;
    for (int i = 0; i < n; ++i) {
      s = s * 1664525u + 1013904223u;
      x[i] = ((s >> 8) * (1.f / 8388608.f) - 1.f) * R;
    }
    hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
    eval<<<n / 256, 256>>>(dx, dh, dc, dp, n);
    hipMemcpy(h.data(), dh, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(c.data(), dc, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(p.data(), dp, n * 4, hipMemcpyDeviceToHost);
    double eh = 0, ec = 0, ep = 0, ef = 0, rms_h = 0, rms_p = 0, rms_f = 0;
    int signbad = 0;
    for (int i = 0; i < n; ++i) {
      const double t = std::sin((double)x[i]), tc = std::cos((double)x[i]);
      eh = std::fmax(eh, std::fabs(h[i] - t));
      ec = std::fmax(ec, std::fabs(c[i] - tc));
      ep = std::fmax(ep, std::fabs(p[i] - t));
      ef = std::fmax(ef, std::fabs((double)sinf(x[i]) - t));
      rms_h += (h[i] - t) * (h[i] - t); rms_p += (p[i] - t) * (p[i] - t);
      const double dfl = (double)sinf(x[i]) - t; rms_f += dfl * dfl;
      if (std::fabs(tc) > 1e-5 && ((c[i] < 0) != (tc < 0))) ++signbad;
    }
    printf("|x| <= %6.1f : v_sin max %.3e rms %.3e | v_cos max %.3e (sign mismatches beyond 1e-5: %d) | poly max %.3e rms %.3e | float sinf max %.3e rms %.3e\n",
           R, eh, std::sqrt(rms_h / n), ec, signbad, ep, std::sqrt(rms_p / n), ef, std::sqrt(rms_f / n));
  }
  return 0;
}
