// Diagnostic: per-launch cost of back-to-back kernels on one stream (empty kernel, LDS-heavy empty kernel,
// and a streaming write kernel that leaves dirty lines behind).
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;} } while (0)
__global__ void k_empty(float* p) { if (p == nullptr && threadIdx.x == 9999) p[0] = 1.f; }
__global__ void k_lds(float* p) { __shared__ float s[8704]; s[threadIdx.x] = 1.f; __syncthreads(); if (s[(threadIdx.x + 1) & 255] == 2.f) p[0] = 1.f; }
__global__ void k_write(float4* p, size_t n) { size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; for (; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = make_float4(1.f, 2.f, 3.f, 4.f); }
template <class F> static float run(F f, int reps) { hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b); for (int i = 0; i < 3; ++i) f(); hipDeviceSynchronize(); hipEventRecord(a, 0); for (int i = 0; i < reps; ++i) f(); hipEventRecord(b, 0); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); return ms / reps * 1000.f; }
int main() {
  float* d; CK(hipMalloc(&d, (size_t)512 << 20));
  printf("empty 256 WG        %.2f us/launch\n", run([&] { hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, 0, d); }, 200));
  printf("empty 8192 WG       %.2f us/launch\n", run([&] { hipLaunchKernelGGL(k_empty, dim3(8192), dim3(256), 0, 0, d); }, 200));
  printf("lds34K 8192 WG      %.2f us/launch\n", run([&] { hipLaunchKernelGGL(k_lds, dim3(8192), dim3(256), 0, 0, d); }, 200));
  for (size_t mb : {16, 64, 256, 512}) {
    size_t n = (mb << 20) / 16;
    float us = run([&] { hipLaunchKernelGGL(k_write, dim3(2048), dim3(256), 0, 0, (float4*)d, n); }, 50);
    printf("write %4zu MB       %.2f us/launch  (%.2f TB/s)\n", mb, us, (double)(mb << 20) / us / 1e6);
  }
  return 0;
}
