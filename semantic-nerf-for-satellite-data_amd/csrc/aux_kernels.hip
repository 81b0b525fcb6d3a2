// Small HBM-bound kernels around the GEMMs: ray sampling, positional encoding, parameter
// (un)packing, slab reductions.  All are launched with shapes checked on the host (api.hip).
#include "aux_kernels.h"

namespace snerf {

// ---- stratified sampling: z = lower + (upper - lower) * u ------------------------------------------
// framework/components/rendering.py:95-110.  Every product/sum is rounded separately (no FMA
// contraction) so z is bit-identical to the reference's chain of elementwise ATen ops.
// NB: HIP's *_rn float intrinsics are plain operators and hipcc contracts a*b+c into an FMA by
// default (-ffp-contract=fast), so contraction is switched off per function here.
__device__ __forceinline__ float mul_nofma(float a, float b) {
#pragma clang fp contract(off)
  return a * b;
}
__device__ __forceinline__ float add_nofma(float a, float b) {
#pragma clang fp contract(off)
  return a + b;
}
__device__ __forceinline__ float z_lin(float near, float far, float s) {
#pragma clang fp contract(off)
  const float a = near * (1.f - s);
  const float b = far * s;
  return a + b;
}

__global__ void sample_z_kernel(const float* __restrict__ rays, const float* __restrict__ zsteps,
                                const float* __restrict__ u, float* __restrict__ z, int N, int S) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= N * S) return;
  const int n = g / S, j = g - n * S;
  const float near = rays[n * 8 + 6], far = rays[n * 8 + 7];
  const float zj = z_lin(near, far, zsteps[j]);
  float out = zj;
  if (u != nullptr) {
    float lower = zj, upper = zj;
    if (j > 0) lower = mul_nofma(0.5f, add_nofma(z_lin(near, far, zsteps[j - 1]), zj));
    if (j < S - 1) upper = mul_nofma(0.5f, add_nofma(zj, z_lin(near, far, zsteps[j + 1])));
    out = add_nofma(lower, mul_nofma(add_nofma(upper, -lower), u[g]));
  }
  z[g] = out;
}

int launch_sample_z(const float* rays, const float* zsteps, const float* u, float* z, int N, int S,
                    hipStream_t st) {
  const int n = N * S;
  hipLaunchKernelGGL(sample_z_kernel, dim3((n + 255) / 256), dim3(256), 0, st, rays, zsteps, u, z, N, S);
  SNERF_LAUNCH_CHECK();
  return 0;
}

// ---- parameter pack / gradient unpack ---------------------------------------------------------------
__global__ void copy_table_kernel(CopyTable tb, float* __restrict__ packed, int mode) {
  const CopyEntry e = tb.e[blockIdx.x];
  const int total = e.rows * e.cols;
  for (int idx = blockIdx.y * blockDim.x + threadIdx.x; idx < total; idx += gridDim.y * blockDim.x) {
    const int r = idx / e.cols, c = idx - r * e.cols;
    float* pk = packed + e.dst_off + (size_t)r * e.dst_ld + c;
    float* us = e.user + (size_t)r * e.user_ld + c;
    if (mode == 0) *pk = *us;          // pack: parameter tensor -> packed
    else if (mode == 1) *us = *pk;     // unpack (overwrite)
    else *us += *pk;                   // unpack (accumulate)
  }
}

int launch_copy_table(const CopyTable& tb, float* packed, int mode, hipStream_t st) {
  if (tb.n <= 0) return 0;
  hipLaunchKernelGGL(copy_table_kernel, dim3(tb.n, 64), dim3(256), 0, st, tb, packed, mode);   // 64 workgroups per tensor: the 512 x 512 ones are 1 MB each
  SNERF_LAUNCH_CHECK();
  return 0;
}

// ---- slab reductions ---------------------------------------------------------------------------------
// out[g][j] (+)= sum_{q in group g} in[q][j]; deterministic (fixed order), no atomics.
__global__ void reduce_rows_kernel(const float* __restrict__ in, int n_in, size_t in_stride, int width,
                                   float* __restrict__ out, size_t out_stride, int group, int accumulate) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= width) return;
  const int g = blockIdx.y;
  const int q0 = g * group, q1 = min(n_in, q0 + group);
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int q = q0;
  for (; q + 3 < q1; q += 4) {
    s0 += in[(size_t)(q + 0) * in_stride + j];
    s1 += in[(size_t)(q + 1) * in_stride + j];
    s2 += in[(size_t)(q + 2) * in_stride + j];
    s3 += in[(size_t)(q + 3) * in_stride + j];
  }
  for (; q < q1; ++q) s0 += in[(size_t)q * in_stride + j];
  const float s = (s0 + s1) + (s2 + s3);
  float* o = out + (size_t)g * out_stride + j;
  *o = accumulate ? (*o + s) : s;
}

int launch_reduce_rows(const float* in, int n_in, size_t in_stride, int width, float* out,
                       size_t out_stride, int group, int accumulate, hipStream_t st) {
  if (n_in <= 0 || width <= 0) return 0;
  const int ng = (n_in + group - 1) / group;
  hipLaunchKernelGGL(reduce_rows_kernel, dim3((width + 255) / 256, ng), dim3(256), 0, st, in, n_in, in_stride,
                     width, out, out_stride, group, accumulate);
  SNERF_LAUNCH_CHECK();
  return 0;
}

// out[j] += sum_q in[q][j] over MANY partial rows in one launch: a workgroup owns 16 columns, its 16 row groups each sum
// every 16th row (four accumulators, fixed order), then the 16 partial sums are added in group order.  Deterministic.
__global__ __launch_bounds__(256) void reduce_cols_kernel(const float* __restrict__ in, int n_in, size_t in_stride, int width,
                                                          float* __restrict__ out) {
  __shared__ float part[16][17];
  const int c = threadIdx.x & 15, rg = threadIdx.x >> 4;
  const int j = blockIdx.x * 16 + c;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (j < width) {
    int q = rg;
    for (; q + 48 < n_in; q += 64) {
      s0 += in[(size_t)q * in_stride + j];
      s1 += in[(size_t)(q + 16) * in_stride + j];
      s2 += in[(size_t)(q + 32) * in_stride + j];
      s3 += in[(size_t)(q + 48) * in_stride + j];
    }
    for (; q < n_in; q += 16) s0 += in[(size_t)q * in_stride + j];
  }
  part[rg][c] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (rg == 0 && j < width) {
    float s = 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) s += part[g][c];
    out[j] += s;
  }
}

// deterministic reduction of [n_in][stride] partials into out[width] (+=), one launch
int reduce_partials(const float* in, int n_in, size_t in_stride, int width, float* tmp, float* out,
                    hipStream_t st) {
  (void)tmp;
  if (n_in <= 0 || width <= 0) return 0;
  if (n_in <= 64) return launch_reduce_rows(in, n_in, in_stride, width, out, 0, n_in, 1, st);
  hipLaunchKernelGGL(reduce_cols_kernel, dim3((width + 15) / 16), dim3(256), 0, st, in, n_in, in_stride, width, out);
  SNERF_LAUNCH_CHECK();
  return 0;
}

// ---- zero fill ---------------------------------------------------------------------------------------------------------
// A kernel instead of hipMemsetAsync wherever a hot call clears device memory: the memset NODES those calls leave in a
// captured HIP graph did not reproduce the eager calls on ROCm 7.2 (first replay right, later replays with stale
// contents; tools/ablate/graph_probe.py).  4-byte granularity.
__global__ void zero_words_kernel(unsigned* p, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = 0u;
}
int launch_zero_bytes(void* p, size_t bytes, hipStream_t st) {
  const size_t n = bytes / 4;
  if (n == 0) return 0;
  hipLaunchKernelGGL(zero_words_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (unsigned*)p, n);
  SNERF_LAUNCH_CHECK();
  return 0;
}

// ---- per-ray embedding rows ---------------------------------------------------------------------------------------
// ---- per-ray sums of a [P][32] fp32 buffer (gradient of the transient codes: bsp_pass.hip, backward step 3) -----------------
// one wave per ray: lane j takes the samples j, j + 64, ...; the 64 partial sums are folded by shuffles in a fixed tree
__global__ __launch_bounds__(256) void ray_sum32_kernel(const float* __restrict__ d32, int col0, int N, int S, int tau, float* __restrict__ out) {
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (n >= N) return;
  float acc[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) acc[c] = 0.f;
  for (int j = lane; j < S; j += 64) {
    const float* q = d32 + ((size_t)n * S + j) * 32 + col0;
#pragma unroll
    for (int c = 0; c < 16; ++c)
      if (c < tau) acc[c] += q[c];
  }
#pragma unroll
  for (int c = 0; c < 16; ++c)
    if (c < tau) {
      float v = acc[c];
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
      if (lane == 0) out[(size_t)n * tau + c] = v;
    }
}
int launch_ray_sum32(const float* d32, int col0, int N, int S, int tau, float* out, hipStream_t st) {
  if (tau > 16 || col0 < 0 || col0 + tau > 32) { set_error("ray_sum32: tau <= 16 columns inside the 32-wide row"); return SNERF_ERR_BAD_DESC; }
  hipLaunchKernelGGL(ray_sum32_kernel, dim3((N + 3) / 4), dim3(256), 0, st, d32, col0, N, S, tau, out);
  SNERF_LAUNCH_CHECK();
  return SNERF_OK;
}

__global__ void embedding_rows_kernel(const float* __restrict__ table, int n_embed, int tau, const long long* __restrict__ idx,
                                      int n, float* __restrict__ rows) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * tau) return;
  const int r = i / tau, c = i - r * tau;
  const long long v = idx[r];
  if (v < 0 || v >= n_embed) { rows[i] = __builtin_nanf(""); return; }   // torch's nn.Embedding raises; a kernel cannot: NaN rows (as the label path does)
  rows[i] = table[(size_t)v * tau + c];
}
// one workgroup per table row v: thread t adds the rays t, t + 256, ... that index v (ascending), then the 256 partial
// sums are added in thread order -- a fixed summation order for any ray permutation of the same batch order
__global__ __launch_bounds__(256) void embedding_backward_kernel(const long long* __restrict__ idx, const float* __restrict__ d_rows,
                                                                 int n, int tau, float* __restrict__ grad) {
  constexpr int TMAX = 16;                 // t_dim <= 16 (api.hip: make_plan; launch_embedding_backward checks)
  __shared__ float part[TMAX][257];
  const long long v = blockIdx.x;
  float s[TMAX];
#pragma unroll
  for (int c = 0; c < TMAX; ++c) s[c] = 0.f;
  for (int r = threadIdx.x; r < n; r += 256)   // the index list is read once for all columns
    if (idx[r] == v) {
#pragma unroll
      for (int c = 0; c < TMAX; ++c)
        if (c < tau) s[c] += d_rows[(size_t)r * tau + c];
    }
#pragma unroll
  for (int c = 0; c < TMAX; ++c)
    if (c < tau) part[c][threadIdx.x] = s[c];
  __syncthreads();
  if ((int)threadIdx.x < tau) {            // one thread per column adds the 256 partials in thread order
    float a = 0.f;
    for (int t = 0; t < 256; ++t) a += part[threadIdx.x][t];
    grad[(size_t)v * tau + threadIdx.x] += a;
  }
}
int launch_embedding_rows(const float* table, int n_embed, int tau, const long long* idx, int n, float* rows, hipStream_t st) {
  hipLaunchKernelGGL(embedding_rows_kernel, dim3((n * tau + 255) / 256), dim3(256), 0, st, table, n_embed, tau, idx, n, rows);
  SNERF_LAUNCH_CHECK();
  return 0;
}
int launch_embedding_backward(const long long* idx, const float* d_rows, int n, int tau, int n_embed, float* grad, hipStream_t st) {
  if (tau < 1 || tau > 16) { set_error("embedding_backward: t_dim must be in 1..16"); return SNERF_ERR_BAD_DESC; }
  hipLaunchKernelGGL(embedding_backward_kernel, dim3(n_embed), dim3(256), 0, st, idx, d_rows, n, tau, grad);
  SNERF_LAUNCH_CHECK();
  return 0;
}

// ---- batched reductions: every slab / column-sum reduction of a backward pass in TWO launches ------------------------
// out[e] += sum_{q < n_in} in[q * stride + e]; job tables travel as kernel arguments.  Fixed order -> deterministic.
// "elem" jobs (dW split slabs: few rows, many elements): a thread owns four consecutive elements (or one, `vec` = 0) and
// walks the slabs; "col" jobs (column-sum partials: thousands of rows, <= 1024 columns): reduce_cols_kernel's scheme.
__global__ __launch_bounds__(256) void reduce_elem_jobs_kernel(const RedChunk tb) {
  int k = 0;
  while (k + 1 < tb.n && (int)blockIdx.x >= tb.j[k + 1].blk0) ++k;
  const RedJob jb = tb.j[k];
  const size_t e0 = ((size_t)(blockIdx.x - jb.blk0) * 256 + threadIdx.x) * (jb.vec ? 4 : 1);
  if (e0 >= (size_t)jb.width) return;
  // 2-D jobs (a column block of a wider matrix): row / column of the element (a vector of four never straddles a row: cols % 4 == 0)
  const size_t r = e0 / (size_t)jb.cols, cc = e0 - r * (size_t)jb.cols;
  const size_t e = jb.cols == jb.width ? e0 : r * (size_t)jb.in_ld + cc;
  const size_t eo = jb.cols == jb.width ? e0 : r * (size_t)jb.out_ld + cc;
  if (jb.vec) {
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
    int q = 0;
    for (; q + 1 < jb.n_in; q += 2) {
      const float4 a = *reinterpret_cast<const float4*>(jb.in + (size_t)q * jb.stride + e);
      const float4 b = *reinterpret_cast<const float4*>(jb.in + (size_t)(q + 1) * jb.stride + e);
      s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w;
      s1.x += b.x; s1.y += b.y; s1.z += b.z; s1.w += b.w;
    }
    if (q < jb.n_in) { const float4 a = *reinterpret_cast<const float4*>(jb.in + (size_t)q * jb.stride + e); s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w; }
    float4* o = reinterpret_cast<float4*>(jb.out + eo);
    float4 v = *o;
    v.x += s0.x + s1.x; v.y += s0.y + s1.y; v.z += s0.z + s1.z; v.w += s0.w + s1.w;
    *o = v;
  } else {
    float s0 = 0.f, s1 = 0.f;
    int q = 0;
    for (; q + 1 < jb.n_in; q += 2) { s0 += jb.in[(size_t)q * jb.stride + e]; s1 += jb.in[(size_t)(q + 1) * jb.stride + e]; }
    if (q < jb.n_in) s0 += jb.in[(size_t)q * jb.stride + e];
    jb.out[eo] += s0 + s1;
  }
}

__global__ __launch_bounds__(256) void reduce_col_jobs_kernel(const RedChunk tb) {
  __shared__ float part[16][17];
  int k = 0;
  while (k + 1 < tb.n && (int)blockIdx.x >= tb.j[k + 1].blk0) ++k;
  const RedJob jb = tb.j[k];
  const int c = threadIdx.x & 15, rg = threadIdx.x >> 4;
  const int j = ((int)blockIdx.x - jb.blk0) * 16 + c;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (j < jb.width) {
    int q = rg;
    for (; q + 48 < jb.n_in; q += 64) {
      s0 += jb.in[(size_t)q * jb.stride + j];
      s1 += jb.in[(size_t)(q + 16) * jb.stride + j];
      s2 += jb.in[(size_t)(q + 32) * jb.stride + j];
      s3 += jb.in[(size_t)(q + 48) * jb.stride + j];
    }
    for (; q < jb.n_in; q += 16) s0 += jb.in[(size_t)q * jb.stride + j];
  }
  part[rg][c] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (rg == 0 && j < jb.width) {
    float s = 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) s += part[g][c];
    jb.out[j] += s;
  }
}

int red_add_elem(RedTable& tb, const float* in, int n_in, size_t stride, size_t width, float* out) {
  if (n_in <= 0 || width == 0) return 0;
  if (tb.n >= RED_MAX) { set_error("reduction queue full"); return SNERF_ERR_WORKSPACE; }
  RedJob& j = tb.j[tb.n++];
  j.in = in; j.out = out; j.stride = stride; j.n_in = n_in; j.width = (long long)width; j.blk0 = tb.blocks;
  j.cols = (int)(width < 0x7fffffff ? width : 0x7fffffff); j.in_ld = j.out_ld = 0;
  if ((long long)j.cols != j.width) { set_error("reduction job too wide"); return SNERF_ERR_WORKSPACE; }
  j.vec = ((width & 3) == 0 && (stride & 3) == 0 && ((uintptr_t)in & 15) == 0 && ((uintptr_t)out & 15) == 0) ? 1 : 0;
  tb.blocks += (int)((width / (j.vec ? 4 : 1) + 255) / 256);
  return 0;
}
int red_add_elem2d(RedTable& tb, const float* in, int n_in, size_t stride, int rows, int cols, int in_ld, float* out, int out_ld) {
  if (n_in <= 0 || rows <= 0 || cols <= 0) return 0;
  if (cols == in_ld && cols == out_ld) return red_add_elem(tb, in, n_in, stride, (size_t)rows * cols, out);
  if (tb.n >= RED_MAX) { set_error("reduction queue full"); return SNERF_ERR_WORKSPACE; }
  RedJob& j = tb.j[tb.n++];
  j.in = in; j.out = out; j.stride = stride; j.n_in = n_in; j.width = (long long)rows * cols; j.blk0 = tb.blocks;
  j.cols = cols; j.in_ld = in_ld; j.out_ld = out_ld;
  j.vec = ((cols & 3) == 0 && (in_ld & 3) == 0 && (out_ld & 3) == 0 && (stride & 3) == 0 && ((uintptr_t)in & 15) == 0 && ((uintptr_t)out & 15) == 0) ? 1 : 0;
  tb.blocks += (int)(((size_t)j.width / (j.vec ? 4 : 1) + 255) / 256);
  return 0;
}
int red_add_col(RedTable& tb, const float* in, int n_in, size_t stride, int width, float* out) {
  if (n_in <= 0 || width <= 0) return 0;
  if (tb.n >= RED_MAX) { set_error("reduction queue full"); return SNERF_ERR_WORKSPACE; }
  RedJob& j = tb.j[tb.n++];
  j.in = in; j.out = out; j.stride = stride; j.n_in = n_in; j.width = width; j.blk0 = tb.blocks; j.vec = 0;
  tb.blocks += (width + 15) / 16;
  return 0;
}
// The job tables travel as kernel arguments in chunks of RED_CHUNK jobs (~1.2 KB): by-value arguments beyond ~2 KB did not
// survive capture in a HIP graph on ROCm 7.2 (replays read a corrupt table -> wild addresses; the eager launch was fine).
template <class K>
static int launch_red_chunks(K kernel, const RedTable& tb, hipStream_t st) {
  for (int first = 0; first < tb.n; first += RED_CHUNK) {
    RedChunk c;
    c.n = tb.n - first < RED_CHUNK ? tb.n - first : RED_CHUNK;
    const int b0 = tb.j[first].blk0;
    for (int i = 0; i < c.n; ++i) { c.j[i] = tb.j[first + i]; c.j[i].blk0 -= b0; }
    const int b1 = first + c.n < tb.n ? tb.j[first + c.n].blk0 : tb.blocks;
    hipLaunchKernelGGL(kernel, dim3(b1 - b0), dim3(256), 0, st, c);
    SNERF_LAUNCH_CHECK();
  }
  return 0;
}
int launch_reductions(const RedTable& elem, const RedTable& col, hipStream_t st) {
  if (elem.n > 0) { int rc = launch_red_chunks(reduce_elem_jobs_kernel, elem, st); if (rc) return rc; }
  if (col.n > 0) { int rc = launch_red_chunks(reduce_col_jobs_kernel, col, st); if (rc) return rc; }
  return 0;
}

// column sums of a [rows][32] buffer -> partial[blocks][32] (256 rows per block, float4 loads: 8 lanes per row)

}  // namespace snerf
