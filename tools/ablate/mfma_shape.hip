// Diagnostic (not part of the product): LDS-fed bf16 MFMA loops in the split kernel's configuration (256 threads, 64x64 wave
// tile, 3 planes, 48 KB LDS -> 3 workgroups per CU) for the two instruction shapes:
//   shape 0: 12 ds_read_b128 + 24 x v_mfma_f32_32x32x16_bf16 per 16-k tile (six plane products per 32x32 block)
//   shape 1: 20 ds_read_b128 + 48 x v_mfma_f32_16x16x32_bf16 per 16-k tile (K-concatenated plane pairs: 3 per 16x16 block)
// Random operands (the clock the chip holds depends on the data).  Prints ms and the equivalent plane-product TFLOP/s.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <string.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
constexpr int PLANE = 4096, STAGE = 6 * PLANE;

template <int SHAPE>
__global__ __launch_bounds__(256, 3) void k(const unsigned short* __restrict__ src, float* __restrict__ out, int iters) {
  __shared__ __attribute__((aligned(16))) char lds[2 * STAGE];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  for (int i = t; i < 2 * STAGE / 16; i += 256) reinterpret_cast<uint4*>(lds)[i] = reinterpret_cast<const uint4*>(src)[i + (blockIdx.x & 7) * 64];
  __syncthreads();
  const int wi0 = (wave >> 1) * 64, wj0 = (wave & 1) * 64;
  float sum = 0.f;
  if (SHAPE == 0) {
    f32x16 acc[2][2];
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
      const char* sa = lds + (it & 1) * STAGE; const char* sb = sa + 3 * PLANE;
      bf16x8 a[3][2], b[3][2];
      for (int pl = 0; pl < 3; ++pl) for (int m = 0; m < 2; ++m) {
        const int ra = wi0 + 32 * m + (lane & 31), rb = wj0 + 32 * m + (lane & 31), h = lane >> 5;
        a[pl][m] = *reinterpret_cast<const bf16x8*>(sa + pl * PLANE + ra * 32 + (((h ^ (ra >> 3)) & 1) << 4));
        b[pl][m] = *reinterpret_cast<const bf16x8*>(sb + pl * PLANE + rb * 32 + (((h ^ (rb >> 3)) & 1) << 4));
      }
      for (int m = 0; m < 2; ++m) for (int n = 0; n < 2; ++n) {
        f32x16 c = acc[m][n];
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][m], b[1][n], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][m], b[2][n], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2][m], b[0][n], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][m], b[1][n], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][m], b[0][n], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][m], b[0][n], c, 0, 0, 0);
        acc[m][n] = c;
      }
      __syncthreads();
    }
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) sum += acc[a][b][r];
  } else {
    f32x4 acc[4][4];
    for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) for (int r = 0; r < 4; ++r) acc[a][b][r] = 0.f;
    const int g = lane >> 4, r16 = lane & 15, half = g & 1, sel = g >> 1;   // k-group g: plane pair member sel, k-half
    for (int it = 0; it < iters; ++it) {
      const char* sa = lds + (it & 1) * STAGE; const char* sb = sa + 3 * PLANE;
      // A combos: [h|h] [m|m] [h|l]; B combos: [h|m] [l|h]
      const int pa[3][2] = {{0, 0}, {1, 1}, {0, 2}}, pb[2][2] = {{0, 1}, {2, 0}};
      bf16x8 b[2][4];
      for (int c = 0; c < 2; ++c) for (int n = 0; n < 4; ++n) {
        const int rb = wj0 + 16 * n + r16;
        b[c][n] = *reinterpret_cast<const bf16x8*>(sb + pb[c][sel] * PLANE + rb * 32 + (((half ^ (rb >> 3)) & 1) << 4));
      }
      for (int m = 0; m < 4; ++m) {
        bf16x8 a[3];
        const int ra = wi0 + 16 * m + r16;
        for (int c = 0; c < 3; ++c) a[c] = *reinterpret_cast<const bf16x8*>(sa + pa[c][sel] * PLANE + ra * 32 + (((half ^ (ra >> 3)) & 1) << 4));
        for (int n = 0; n < 4; ++n) {
          f32x4 c = acc[m][n];
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[1][n], c, 0, 0, 0);   // hl + lh
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[0][n], c, 0, 0, 0);   // mh + mm
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[0][n], c, 0, 0, 0);   // hh + hm
          acc[m][n] = c;
        }
      }
      __syncthreads();
    }
    for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) for (int r = 0; r < 4; ++r) sum += acc[a][b][r];
  }
  out[blockIdx.x * 256 + t] = sum;
}

int main() {
  setvbuf(stdout, nullptr, _IOLBF, 0);
  const int WGS = 768 * 8, ITERS = 512;
  std::vector<unsigned short> h(2 * STAGE / 2 + 8 * 64 * 8);
  for (size_t i = 0; i < h.size(); ++i) { float f = (float)((i * 2654435761u >> 7) & 0xffff) / 65536.f - 0.5f; unsigned u; memcpy(&u, &f, 4); h[i] = (unsigned short)(u >> 16); }
  unsigned short* src; float* out;
  CK(hipMalloc(&src, h.size() * 2)); CK(hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice));
  CK(hipMalloc(&out, (size_t)WGS * 256 * 4));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const double flop = (double)WGS * 4 * ITERS * 6.0 * 2 * 64 * 64 * 16;   // plane-product flops
  for (int rep = 0; rep < 3; ++rep)
    for (int shape = 0; shape < 2; ++shape) {
      for (int w = 0; w < (rep == 0 ? 40 : 3); ++w) { if (shape == 0) hipLaunchKernelGGL(k<0>, dim3(WGS), dim3(256), 0, 0, src, out, ITERS); else hipLaunchKernelGGL(k<1>, dim3(WGS), dim3(256), 0, 0, src, out, ITERS); }
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(a, 0));
      for (int w = 0; w < 10; ++w) { if (shape == 0) hipLaunchKernelGGL(k<0>, dim3(WGS), dim3(256), 0, 0, src, out, ITERS); else hipLaunchKernelGGL(k<1>, dim3(WGS), dim3(256), 0, 0, src, out, ITERS); }
      CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 10;
      printf("rep %d shape %s: %.3f ms  %.0f TFLOP/s (plane products)\n", rep, shape ? "16x16x32 (K-concat pairs)" : "32x32x16", ms, flop / ms / 1e9);
    }
  return 0;
}
