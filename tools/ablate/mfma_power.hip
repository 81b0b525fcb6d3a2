// Diagnostic (not part of the product): what the matrix pipe SUSTAINS under the package power cap.  Register-resident
// v_mfma_f32_32x32x16_f16 loops (no LDS, no memory traffic inside the loop), two waves per SIMD on every CU, back-to-back launches for
// a few seconds per operand kind: random fp16 in [-1, 1) (every multiplier input toggles), hi / lo planes of random fp32 values as the
// product's k-loop sees them (hi * hi, hi * lo, lo * hi), and zeros.  Prints the rate per half second; run it under
// tools/ablate/run_mfma_power.sh, which samples rocm-smi (clock, power) beside it.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ablate/mfma_power tools/ablate/mfma_power.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// src: per lane 4 fragments of 8 halves (a0, a1, b0, b1); 8 independent accumulators; 16 MFMAs per iteration
__global__ __launch_bounds__(256, 2) void k(const f16x8* __restrict__ src, float* __restrict__ out, int iters) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  f16x8 a0 = src[4 * t], a1 = src[4 * t + 1], b0 = src[4 * t + 2], b1 = src[4 * t + 3];
  f32x16 acc[8];
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16((i & 1) ? a1 : a0, (i & 2) ? b1 : b0, acc[i], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16((i & 2) ? a1 : a0, (i & 1) ? b1 : b0, acc[i], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[t] = s;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
// the same accumulator footprint (128 registers) on v_mfma_f32_16x16x32_f16: 32 independent 16 x 16 accumulators, 32 MFMAs per iteration
// (= the FLOPs of 16 of the 32 x 32 x 16 instructions)
__global__ __launch_bounds__(256, 2) void k16(const f16x8* __restrict__ src, float* __restrict__ out, int iters) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  f16x8 a0 = src[4 * t], a1 = src[4 * t + 1], b0 = src[4 * t + 2], b1 = src[4 * t + 3];
  f32x4 acc[32];
  for (int i = 0; i < 32; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16((i & 1) ? a1 : a0, (i & 2) ? b1 : b0, acc[i], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int i = 0; i < 32; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
  out[t] = s;
}

int main(int argc, char** argv) {
  const double seconds = argc > 1 ? atof(argv[1]) : 4.0;
  int dev = 0, cus = 256;
  CK(hipGetDevice(&dev));
  CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  const int grid = cus * 2, nthr = grid * 256, iters = 20000;     // 2 workgroups of 4 waves per CU = 2 waves per SIMD
  std::vector<_Float16> h((size_t)nthr * 32);
  f16x8* d; float* o;
  CK(hipMalloc(&d, h.size() * 2)); CK(hipMalloc(&o, (size_t)nthr * 4));
  const char* kinds[] = {"zeros", "random fp16 in [-1, 1)", "hi / lo planes of random fp32 (a0, b0 = hi; a1, b1 = lo)", "zeros again"};
  for (int shape = 0; shape < 2; ++shape)
  for (int kind = 0; kind < 4; ++kind) {
    if (shape == 1 && kind == 3) continue;
    srand(1);
    for (size_t i = 0; i < h.size(); ++i) {
      const float x = 2.f * rand() / (float)RAND_MAX - 1.f;
      if (kind == 0 || kind == 3) h[i] = (_Float16)0.f;
      else if (kind == 1) h[i] = (_Float16)x;
      else {   // fragments 0, 2 of a lane: hi = fp16(x * 2^13); fragments 1, 3: lo = fp16(x * 2^13 - hi)  (block-scaled planes, csrc/bsp.h)
        const size_t frag = (i / 8) & 3;
        const float v = x * 8192.f; const _Float16 hi = (_Float16)v;
        h[i] = (frag & 1) ? (_Float16)(v - (float)hi) : hi;
      }
    }
    CK(hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    const double flop = 2.0 * 32 * 32 * 16 * 16.0 * iters * (nthr / 64.0);
    printf("== %s, %s\n", shape == 0 ? "v_mfma_f32_32x32x16_f16" : "v_mfma_f32_16x16x32_f16", kinds[kind]); fflush(stdout);
    auto t0 = std::chrono::steady_clock::now();
    double last = 0; int n = 0, n_last = 0;
    for (;;) {
      if (shape == 0) hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, d, o, iters);
      else hipLaunchKernelGGL(k16, dim3(grid), dim3(256), 0, 0, d, o, iters);
      ++n;
      if ((n & 3) == 0) {
        CK(hipDeviceSynchronize());
        const double t = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (t - last >= 0.5) { printf("   t = %.1f s: %.0f TFLOP/s\n", t, flop * (n - n_last) / (t - last) / 1e12); fflush(stdout); last = t; n_last = n; }
        if (t >= seconds) break;
      }
    }
  }
  return 0;
}
