// Shared device/host helpers for libsnerf_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

namespace snerf {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__host__ __device__ inline int round_up(int x, int m) { return (x + m - 1) / m * m; }
__host__ __device__ inline size_t round_up_sz(size_t x, size_t m) { return (x + m - 1) / m * m; }

// ---- elementwise math with the reference's semantics (torch defaults) ----------------------------
// torch.nn.Softplus(beta=1, threshold=20): x if x > 20 else log1p(exp(x))
__device__ __forceinline__ float softplus_f(float x) { return x > 20.f ? x : log1pf(expf(x)); }
// d softplus / dx = sigmoid(x) (1 above the threshold)
__device__ __forceinline__ float softplus_grad_f(float x) { return x > 20.f ? 1.f : 1.f / (1.f + expf(-x)); }
__device__ __forceinline__ float sigmoid_f(float x) { return 1.f / (1.f + expf(-x)); }

// Accurate sincos for SIREN / positional encoding arguments.
// Cody-Waite reduction by pi/2 split in four parts (exact products for |k| < 2^15), then Chebyshev-fit
// polynomials on [-pi/4, pi/4]: <= 1.55 ulp / 9.3e-8 abs for |x| <= 3e4 (measured against fp64).
// Larger arguments take a double-precision three-part reduction (valid to |x| ~ 1.6e6; beyond that the
// spacing of fp32 itself exceeds 0.1 rad).  No library call: ocml's sincosf keeps Payne-Hanek tables in
// scratch, which would push the GEMM accumulators out of registers.  (The reference evaluates torch.sin
// on ATen kernels, <= 1-2 ulp; hardware v_sin_f32 is NOT accurate enough once amplified by w0 = 30.)
__device__ __forceinline__ void sincos_acc(float x, float* s, float* c) {
  float r;
  int k;
  if (__builtin_expect(fabsf(x) <= 30000.f, 1)) {
    const float kf = rintf(x * 0.63661977236758134308f);  // x * 2/pi
    k = (int)kf;
    // pi/2 = A + B + C + D; A and B carry few mantissa bits so kf*A, kf*B are exact for |kf| < 2^15
    r = fmaf(kf, -1.5703125f, x);
    r = fmaf(kf, -4.837512969970703125e-4f, r);
    r = fmaf(kf, -7.549790126404332e-08f, r);
    r = fmaf(kf, 1.7151245100058819e-15f, r);
  } else {
    const double xd = (double)x;
    const double kd = rint(xd * 0.63661977236758134308);
    double rd = fma(kd, -1.57079632673412561417e+00, xd);  // 33-bit pieces of pi/2 (fdlibm pio2_1..3)
    rd = fma(kd, -6.07710050630396597660e-11, rd);
    rd = fma(kd, -2.02226624871116645580e-21, rd);
    r = (float)rd;
    k = (int)((long long)kd & 3);
  }
  const float r2 = r * r;
  // sin(r) = r + r^3 P(r^2), cos(r) = 1 - r^2/2 + r^4 Q(r^2)
  float ps = fmaf(r2, 2.7237618203173253e-06f, -0.00019839989971755576f);
  ps = fmaf(ps, r2, 0.00833333169215251f);
  ps = fmaf(ps, r2, -0.16666666663377128f);
  const float sr = fmaf(r * r2, ps, r);
  float pc = fmaf(r2, -2.7290681713851145e-07f, 2.48005195681466e-05f);
  pc = fmaf(pc, r2, -0.0013888887519541702f);
  pc = fmaf(pc, r2, 0.04166666666392173f);
  const float cr = fmaf(r2 * r2, pc, fmaf(r2, -0.5f, 1.0f));
  const bool swap = k & 1;
  float ss = swap ? cr : sr;
  float cc = swap ? sr : cr;
  if (k & 2) ss = -ss;
  if ((k + 1) & 2) cc = -cc;
  *s = ss;
  *c = cc;
}

// sin(x) and the SIGN of cos(x) only -- what a SIREN layer needs (the backward pass rebuilds |cos| = sqrt(1 - sin^2)).
// Reduction by pi (four-part Cody-Waite, exact for |k| < 2^15) to r in [-pi/2, pi/2], ONE odd polynomial up to r^11
// (fit error 2e-11; fp32 evaluation <= 1.9 ulp / 1.1e-7 abs against fp64 for |x| <= 3e4), sin x = (-1)^k sin r and
// cos x < 0 iff k is odd, corrected when rounding of k left |r| a hair above pi/2.  ~17 VALU against ~28 for
// sincos_acc.  Larger arguments reduce in double like sincos_acc.
__device__ __forceinline__ float sin_signcos(float x, bool* cos_neg) {
  float r;
  int k;
  if (__builtin_expect(fabsf(x) <= 30000.f, 1)) {
    const float kf = rintf(x * 0.31830988618379067154f);  // x / pi
    k = (int)kf;
    r = fmaf(kf, -3.140625f, x);
    r = fmaf(kf, -9.67502593994140625e-4f, r);
    r = fmaf(kf, -1.509958025280866e-07f, r);
    r = fmaf(kf, 3.4302490200117638e-15f, r);
  } else {
    const double xd = (double)x;
    const double kd = rint(xd * 0.31830988618379067154);
    double rd = fma(kd, -3.14159265346825122833e+00, xd);  // 2 x the 33-bit pieces of pi/2 (fdlibm pio2_1..3)
    rd = fma(kd, -1.21542010126079319532e-10, rd);
    rd = fma(kd, -4.04453249742233291160e-21, rd);
    r = (float)rd;
    k = (int)((long long)kd & 1);
  }
  const float r2 = r * r;
  float p = fmaf(r2, -2.5028294103890403e-08f, 2.755689592959243e-06f);
  p = fmaf(p, r2, -0.00019841265748254955f);
  p = fmaf(p, r2, 0.008333333767950535f);
  p = fmaf(p, r2, -0.1666666716337204f);
  const float s = fmaf(r * r2, p, r);
  const bool odd = k & 1;
  *cos_neg = odd != (fabsf(r) > 1.57079637f);
  return odd ? -s : s;
}

// Four at once: the large-argument test is ONE wave-uniform branch (ballot), so the common path is straight-line code
// whose four dependency chains the scheduler interleaves; per-element tests cost an exec-mask dance and a branch each
// and serialise the chains.  Returns the four sines; bit c of *neg = cos(x_c) < 0.
__device__ __forceinline__ float4 sin4_signcos(float4 x, unsigned* neg) {
  const float m = fmaxf(fmaxf(fabsf(x.x), fabsf(x.y)), fmaxf(fabsf(x.z), fabsf(x.w)));
  float4 s;
  if (__builtin_expect(__builtin_amdgcn_ballot_w64(!(m <= 30000.f)) == 0ull, 1)) {
    float xs[4] = {x.x, x.y, x.z, x.w}, out[4];
    unsigned bits = 0u;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float kf = rintf(xs[c] * 0.31830988618379067154f);
      float r = fmaf(kf, -3.140625f, xs[c]);
      r = fmaf(kf, -9.67502593994140625e-4f, r);
      r = fmaf(kf, -1.509958025280866e-07f, r);
      r = fmaf(kf, 3.4302490200117638e-15f, r);
      const float r2 = r * r;
      float p = fmaf(r2, -2.5028294103890403e-08f, 2.755689592959243e-06f);
      p = fmaf(p, r2, -0.00019841265748254955f);
      p = fmaf(p, r2, 0.008333333767950535f);
      p = fmaf(p, r2, -0.1666666716337204f);
      const float sv = fmaf(r * r2, p, r);
      const unsigned odd = (unsigned)(int)kf & 1u;
      out[c] = __uint_as_float(__float_as_uint(sv) ^ (odd << 31));
      bits |= (odd ^ (fabsf(r) > 1.57079637f ? 1u : 0u)) << c;
    }
    s = make_float4(out[0], out[1], out[2], out[3]);
    *neg = bits;
  } else {
    bool n0, n1, n2, n3;
    s.x = sin_signcos(x.x, &n0); s.y = sin_signcos(x.y, &n1); s.z = sin_signcos(x.z, &n2); s.w = sin_signcos(x.w, &n3);
    *neg = (n0 ? 1u : 0u) | (n1 ? 2u : 0u) | (n2 ? 4u : 0u) | (n3 ? 8u : 0u);
  }
  return s;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

}  // namespace snerf

// ---- host-side error plumbing ------------------------------------------------------------------
namespace snerf {
void set_error(const char* fmt, ...);
}
#define SNERF_HIP_CHECK(expr)                                                              \
  do {                                                                                     \
    hipError_t _e = (expr);                                                                \
    if (_e != hipSuccess) {                                                                \
      snerf::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return SNERF_ERR_HIP;                                                                \
    }                                                                                      \
  } while (0)
#define SNERF_LAUNCH_CHECK() SNERF_HIP_CHECK(hipGetLastError())
