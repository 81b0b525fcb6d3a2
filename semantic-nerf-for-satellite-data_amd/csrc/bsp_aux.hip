// Producers and converters of block-scaled fp16-plane tensors (bsp.h) that are not GEMM epilogues: positional
// encoding + per-sample extras, the weight pack, fp32 <-> planes conversion (32-wide head gradients; tests).  All HBM-bound.
#include "bsp.h"
#include "aux_kernels.h"

namespace snerf {
namespace bsp {

__device__ __forceinline__ float block_max_256(float m, float* sm) {   // |max| over a 256-thread workgroup (sm: 4 floats)
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
  __syncthreads();
  return fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
}

// ---- fp32 -> planes: one workgroup per (128-row, 128-column) block ----------------------------------------------------
__global__ __launch_bounds__(256) void to_planes_kernel(const float* __restrict__ src, int ld_src, int rows, int cols,
                                                        char* __restrict__ dst, int* __restrict__ E, int ld, int col0, int pl) {
  __shared__ float sm[4];
  const int rb = blockIdx.y, cb = blockIdx.x;
  const int t = threadIdx.x, c8 = cb * 128 + (t & 15) * 8;
  float v[8][8];
  float m = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int r = rb * 128 + (t >> 4) + 16 * i;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const float x = (r < rows && c8 + c < cols) ? src[(size_t)r * ld_src + c8 + c] : 0.f;
      v[i][c] = x;
      m = fmaxf(m, fabsf(x));
    }
  }
  m = block_max_256(m, sm);
  const int e = exp_of_maxbits(__float_as_uint(m));
  if (t == 0) E[(size_t)rb * ncb_of(ld) + (col0 >> 7) + cb] = e;
  const float sc = pow2f(e);
  if (col0 + c8 >= ld) return;          // columns [cols, ld) inside the block are written as zeros (finite pads)
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int r = rb * 128 + (t >> 4) + 16 * i;
    if (r >= rows) continue;
    u32x4 hi, lo;
    split8(v[i], sc, hi, lo);
    char* d = dst + (size_t)r * ld * 2 * pl + g16_off(col0 + c8, pl);
    *reinterpret_cast<u32x4*>(d) = hi;
    if (pl == 2) *reinterpret_cast<u32x4*>(d + 32) = lo;
  }
}

int launch_to_planes(const float* src, int ld_src, int rows, int cols, char* dst, int* E, int ld, int col0, int pl, hipStream_t st) {
  if (!src || !dst || !E || rows <= 0 || cols <= 0 || (ld & 15) || (col0 & 127) || col0 + cols > ld || pl < 1 || pl > 2) {
    set_error("to_planes: bad argument (ld % 16, col0 % 128)");
    return SNERF_ERR_BAD_DESC;
  }
  hipLaunchKernelGGL(to_planes_kernel, dim3((cols + 127) / 128, (rows + 127) / 128), dim3(256), 0, st, src, ld_src, rows, cols,
                     dst, E, ld, col0, pl);
  SNERF_LAUNCH_CHECK();
  return SNERF_OK;
}

__global__ void from_planes_kernel(const char* __restrict__ src, const int* __restrict__ E, int ld, int col0, int rows, int cols,
                                   float* __restrict__ dst, int ld_dst, int pl) {
  const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= (size_t)rows * cols) return;
  const int r = (int)(g / cols), c = (int)(g - (size_t)r * cols);
  const char* p = src + (size_t)r * ld * 2 * pl + g16_off(col0 + c, pl);
  const float hi = (float)*reinterpret_cast<const _Float16*>(p), lo = pl == 2 ? (float)*reinterpret_cast<const _Float16*>(p + 32) : 0.f;
  const int e = E[(size_t)(r >> 7) * ncb_of(ld) + ((col0 + c) >> 7)];
  dst[(size_t)r * ld_dst + c] = __builtin_amdgcn_ldexpf(hi + lo, -e);
}

int launch_from_planes(const char* src, const int* E, int ld, int col0, int rows, int cols, float* dst, int ld_dst, int pl, hipStream_t st) {
  const size_t n = (size_t)rows * cols;
  hipLaunchKernelGGL(from_planes_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src, E, ld, col0, rows, cols, dst, ld_dst, pl);
  SNERF_LAUNCH_CHECK();
  return SNERF_OK;
}

// ---- weight pack ------------------------------------------------------------------------------------------------------
// pass 1: |max| of every master matrix (float bits, atomicMax: order-independent); pass 2: every operand (a matrix or its
// transpose) into WF16 units of 32 rows x 16 k = 2 KiB = [plane][lane][8 fp16] with lane = 32 (k / 8) + row: the B operand
// fragment of v_mfma_f32_32x32x16_f16 for 32 output columns, one contiguous KiB per plane.
__global__ __launch_bounds__(256) void wmax_kernel(WPackChunk tb, const float* __restrict__ master, unsigned* __restrict__ maxbits) {
  const WPackJob j = tb.j[blockIdx.y];
  if (j.transposed) return;            // its matrix is covered by the non-transposed job with the same exponent slot
  float m = 0.f;
  // a wave per row (rows strided over the grid's waves), lanes along the row: coalesced, no index division per element (the
  // element-strided loop this replaces spent 43 us on 2.8 M parameters, most of it in 64-bit divisions)
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int r = blockIdx.x * 4 + wave; r < j.m_rows; r += gridDim.x * 4) {
    const float* row = master + j.src_off + (size_t)r * j.src_ld;
    for (int c = lane; c < j.m_cols; c += 64) m = fmaxf(m, fabsf(row[c]));
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(maxbits + j.e_idx, __float_as_uint(m));
}

__global__ __launch_bounds__(256) void wpack_kernel(WPackChunk tb, const float* __restrict__ master, char* __restrict__ planes,
                                                    int* __restrict__ exps, const unsigned* __restrict__ maxbits, int pl) {
  const WPackJob j = tb.j[blockIdx.y];
  const int rb32 = (j.rows + 31) >> 5, nks = j.K >> 4;
  const int e = exp_of_maxbits(maxbits[j.e_idx]);
  if (blockIdx.x == 0 && threadIdx.x == 0) exps[j.e_idx] = e;
  const float sc = pow2f(e);
  for (int unit = blockIdx.x; unit < rb32 * nks; unit += gridDim.x) {
    const int ks = unit / rb32, ub = unit - ks * rb32;
    const int r = threadIdx.x >> 3, kp = (threadIdx.x & 7) * 2;       // row inside the unit, first of two k
    const int ra = ub * 32 + wf16_row(r), ka = ks * 16 + kp;   // slot r of the unit holds matrix row wf16_row(r)
    float x[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int k = ka + q;
      const bool ok = j.transposed ? (k < j.m_rows && ra < j.m_cols) : (ra < j.m_rows && k < j.m_cols);
      x[q] = ok ? master[j.src_off + (j.transposed ? (size_t)k * j.src_ld + ra : (size_t)ra * j.src_ld + k)] : 0.f;
    }
    const _Float16 h0 = (_Float16)(x[0] * sc), h1 = (_Float16)(x[1] * sc);
    const _Float16 l0 = (_Float16)(x[0] * sc - (float)h0), l1 = (_Float16)(x[1] * sc - (float)h1);
    typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
    char* u = planes + j.dst_off + (size_t)unit * 1024 * pl + (32 * (kp >> 3) + r) * 16 + (kp & 7) * 2;   // one plane: 1 KiB units
    *reinterpret_cast<f16x2*>(u) = f16x2{h0, h1};
    if (pl == 2) *reinterpret_cast<f16x2*>(u + 1024) = f16x2{l0, l1};
  }
}

int launch_wpack(const WPackTable& tb, const float* master, char* planes, int* exps, unsigned* maxbits, int pl, hipStream_t st) {
  if (tb.n <= 0) return SNERF_OK;
  { int rc = launch_zero_bytes(maxbits, WPACK_MAX * sizeof(unsigned), st); if (rc) return rc; }
  // the table travels in chunks of WPACK_CHUNK jobs: by-value kernel arguments beyond ~2 KB did not survive capture in a HIP
  // graph on ROCm 7.2 (aux_kernels.hip: launch_red_chunks).  All maxima first: a transposed job shares its matrix's slot.
  for (int pass = 0; pass < 2; ++pass)
    for (int first = 0; first < tb.n; first += WPACK_CHUNK) {
      WPackChunk c;
      c.n = tb.n - first < WPACK_CHUNK ? tb.n - first : WPACK_CHUNK;
      for (int i = 0; i < c.n; ++i) c.j[i] = tb.j[first + i];
      if (pass == 0) hipLaunchKernelGGL(wmax_kernel, dim3(32, c.n), dim3(256), 0, st, c, master, maxbits);
      else hipLaunchKernelGGL(wpack_kernel, dim3(128, c.n), dim3(256), 0, st, c, master, planes, exps, maxbits, pl);
      SNERF_LAUNCH_CHECK();
    }
  return SNERF_OK;
}

// zero `width` bytes (a multiple of 16) of every row of a pitched region: the pad columns between feats and extras of narrow
// networks.  (A kernel, not hipMemset2DAsync: the memset node that call leaves in a captured hipGraph did not reproduce the
// eager call on ROCm 7.2 -- replays zeroed the tensor beside the pad columns.)
__global__ void zero_cols_kernel(char* base, size_t pitch, int width16, int rows) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * width16) return;
  const int r = i / width16, c = i - r * width16;
  *reinterpret_cast<u32x4*>(base + (size_t)r * pitch + 16 * (size_t)c) = u32x4{0u, 0u, 0u, 0u};
}
int launch_zero_cols(char* base, size_t pitch, size_t width_bytes, int rows, hipStream_t st) {
  const int w16 = (int)(width_bytes / 16);
  if (w16 <= 0 || rows <= 0) return SNERF_OK;
  hipLaunchKernelGGL(zero_cols_kernel, dim3((unsigned)(((size_t)rows * w16 + 255) / 256)), dim3(256), 0, st, base, pitch, w16, rows);
  SNERF_LAUNCH_CHECK();
  return SNERF_OK;
}

// ---- sample position + positional encoding + per-sample extras, as planes ------------------------------------------
// One workgroup per 128-point block.  gamma(x) is in [-1, 1]: its blocks take the fixed exponent 13; raw positions
// (baseline SatNeRF: identity encoding) and the extras [sun | t | t_s] get the exponent of their own block maximum.
// (point indices are below 2^30 -- api.hip: make_plan -- so ray = point / S is a 32-bit division: the 64-bit one this replaces cost
//  more instructions per item than four of its sincos evaluations)
__device__ __forceinline__ void point_xyz(const EncodeArgs& a, long long point, float (&x)[3]) {
  const int n = (int)((unsigned)point / (unsigned)a.S);
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    if (a.xyz != nullptr) {
      x[c] = a.xyz[point * 3 + c];
    } else {
#pragma clang fp contract(off)
      const float o = a.rays[n * 8 + c];
      const float d = a.dir_is_sun ? a.sun_d[(size_t)n * a.sun_stride + c] : a.rays[n * 8 + 3 + c];
      const float dz = d * a.z[point];
      x[c] = o + dz;
    }
  }
}
__device__ __forceinline__ float extras_value(const EncodeArgs& a, int n, int c) {
  if (c >= a.x_sun && c < a.x_sun + 3) return a.sun_d[(size_t)n * a.sun_stride + (c - a.x_sun)];
  if (a.t != nullptr && c >= a.x_t && c < a.x_t + a.tau) return a.t[(size_t)n * a.tau + (c - a.x_t)];
  if (a.t_s != nullptr && a.x_ts >= 0 && c >= a.x_ts && c < a.x_ts + a.tau) return a.t_s[(size_t)n * a.tau + (c - a.x_ts)];
  return 0.f;
}

struct EncodeBsp {
  EncodeArgs a;                 // pe / fa fields unused here
  char* pe; int* Epe;           // planes [P][Ep]
  char* fa; int* Efa; int fa_col0;   // extras columns [fa_col0, fa_col0 + 16) of the [P][FA] tensor (fa may be null)
  int pl;                       // planes (bsp.h)
};

__global__ __launch_bounds__(256) void encode_bsp_kernel(EncodeBsp g) {
  __shared__ float sm[4];
  const EncodeArgs& a = g.a;
  const int rb = blockIdx.x, t = threadIdx.x;
  if (rb == 0) for (int i = t; i < a.zero_n; i += 256) a.zero[i] = 0u;   // (nothing before the pass's first GEMM launch reads them)
  const long long P = (long long)a.N * a.S;
  const long long p0 = (long long)rb * 128;
  const int npts = (int)min((long long)128, P - p0);
  // block maxima
  float mx = 0.f, me = 0.f;
  if (a.F == 0)
    for (int i = t; i < npts; i += 256) {
      float x[3];
      point_xyz(a, p0 + i, x);
      mx = fmaxf(mx, fmaxf(fabsf(x[0]), fmaxf(fabsf(x[1]), fabsf(x[2]))));
    }
  if (g.fa != nullptr) {
    const int n0 = (int)((unsigned)p0 / (unsigned)a.S), n1 = (int)((unsigned)(p0 + npts - 1) / (unsigned)a.S);
    for (int i = t; i < (n1 - n0 + 1) * 16; i += 256) me = fmaxf(me, fabsf(extras_value(a, n0 + (i >> 4), i & 15)));
  }
  const int e_pe = a.F > 0 ? 13 : exp_of_maxbits(__float_as_uint(block_max_256(mx, sm)));
  const int e_x = g.fa != nullptr ? exp_of_maxbits(__float_as_uint(block_max_256(me, sm))) : 0;
  if (t == 0) {
    g.Epe[rb] = e_pe;                                              // Ep <= 128: one column block
    if (g.fa != nullptr) g.Efa[(size_t)rb * ncb_of(a.FA) + (g.fa_col0 >> 7)] = e_x;
  }
  const float s_pe = pow2f(e_pe), s_x = pow2f(e_x);
  const int gpe = a.Ep >> 4, ngrp = gpe + (g.fa != nullptr ? 1 : 0);
  for (int item = t; item < npts * ngrp; item += 256) {
    const int pl = item / ngrp, grp = item - pl * ngrp;
    const long long point = p0 + pl;
    float v[16];
    char* d;
    float sc;
    if (grp < gpe) {
      float x[3];
      point_xyz(a, point, x);
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int col = 16 * grp + j;
        float r = 0.f;
        if (a.F > 0) {
          if (col < 6 * a.F) {
            const int k = col / 6, q = col - 6 * k, c = q >= 3 ? q - 3 : q;
            const float xc = c == 0 ? x[0] : (c == 1 ? x[1] : x[2]);
            float s, co;
            sincos_acc((float)(1 << k) * xc, &s, &co);
            r = q >= 3 ? co : s;
          }
        } else if (col < 3) {
          r = col == 0 ? x[0] : (col == 1 ? x[1] : x[2]);
        }
        v[j] = r;
      }
      d = g.pe + (size_t)point * a.Ep * 2 * g.pl + (size_t)grp * 32 * g.pl;
      sc = s_pe;
    } else {
      const int n = (int)((unsigned)point / (unsigned)a.S);
#pragma unroll
      for (int j = 0; j < 16; ++j) v[j] = extras_value(a, n, j);
      d = g.fa + (size_t)point * a.FA * 2 * g.pl + g16_off(g.fa_col0, g.pl);
      sc = s_x;
    }
    u32x4 hi, lo;
    float h8[8];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
      for (int j = 0; j < 8; ++j) h8[j] = v[8 * half + j];
      split8(h8, sc, hi, lo);
      *reinterpret_cast<u32x4*>(d + 16 * half) = hi;
      if (g.pl == 2) *reinterpret_cast<u32x4*>(d + 32 + 16 * half) = lo;
    }
  }
}

int launch_encode_bsp(const EncodeArgs& a, char* pe, int* Epe, char* fa, int* Efa, int fa_col0, int pl, hipStream_t st) {
  if (a.Ep > 128 || (a.Ep & 15) || (fa && ((fa_col0 & 127) || (a.FA & 15) || a.Xp != 16))) {
    set_error("encode: Ep <= 128, Ep % 16 == 0, extras block of 16 columns at a multiple of 128");
    return SNERF_ERR_BAD_DESC;
  }
  EncodeBsp g{a, pe, Epe, fa, Efa, fa_col0, pl};
  const long long P = (long long)a.N * a.S;
  hipLaunchKernelGGL(encode_bsp_kernel, dim3((unsigned)((P + 127) / 128)), dim3(256), 0, st, g);
  SNERF_LAUNCH_CHECK();
  return SNERF_OK;
}

// ---- 32-wide head gradients: column sums (bias gradients) and the planes the dX / dW GEMMs read, in one pass ---------
// [rows][32] fp32 -> partial[blocks][32] (256 rows per block) + planes [rows][32] with one exponent per 128 rows
__global__ __launch_bounds__(256) void colsum32_bsp_kernel(const float* __restrict__ in, int rows, float* __restrict__ partial,
                                                           char* __restrict__ pl, int* __restrict__ E, int npl) {
  __shared__ float4 sm[32][8];
  __shared__ float smx[2][2];
  const int c4 = threadIdx.x & 7, rg = threadIdx.x >> 3;   // 32 row groups x 8 column quads; row = r0 + rg + 32 i
  const int r0 = blockIdx.x * 256;
  float4 v[8];
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  float m[2] = {0.f, 0.f};                                 // rows 0-127 / 128-255 of the block
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int r = r0 + rg + 32 * i;
    v[i] = r < rows ? *reinterpret_cast<const float4*>(in + (size_t)r * 32 + 4 * c4) : make_float4(0.f, 0.f, 0.f, 0.f);
    s.x += v[i].x; s.y += v[i].y; s.z += v[i].z; s.w += v[i].w;
    m[i >> 2] = fmaxf(fmaxf(m[i >> 2], fmaxf(fabsf(v[i].x), fabsf(v[i].y))), fmaxf(fabsf(v[i].z), fabsf(v[i].w)));
  }
  sm[rg][c4] = s;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m[h] = fmaxf(m[h], __shfl_xor(m[h], o, 64));
  }
  __syncthreads();
  if (threadIdx.x < 2) smx[0][threadIdx.x] = 0.f;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) {   // four waves: order-independent max through LDS atomics on the float bits
    atomicMax(reinterpret_cast<unsigned*>(&smx[0][0]), __float_as_uint(m[0]));
    atomicMax(reinterpret_cast<unsigned*>(&smx[0][1]), __float_as_uint(m[1]));
  }
  __syncthreads();
  if (threadIdx.x < 8) {   // fixed summation order over the 32 row groups
    float4 tsum = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < 32; ++i) { const float4 q = sm[i][threadIdx.x]; tsum.x += q.x; tsum.y += q.y; tsum.z += q.z; tsum.w += q.w; }
    *reinterpret_cast<float4*>(partial + (size_t)blockIdx.x * 32 + 4 * threadIdx.x) = tsum;
  }
  if (pl == nullptr) return;
  int e[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    e[h] = exp_of_maxbits(__float_as_uint(smx[0][h]));
    if (threadIdx.x == 0 && r0 + 128 * h < rows) E[(r0 >> 7) + h] = e[h];
  }
  // each lane holds 4 consecutive columns: 8-byte hi and lo pieces
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int r = r0 + rg + 32 * i;
    if (r >= rows) continue;
    const float sc = pow2f(e[i >> 2]);
    const float x[4] = {v[i].x * sc, v[i].y * sc, v[i].z * sc, v[i].w * sc};
    f16x4 h, l;
#pragma unroll
    for (int c = 0; c < 4; ++c) { h[c] = (_Float16)x[c]; l[c] = (_Float16)(x[c] - (float)h[c]); }
    char* d = pl + (size_t)r * 64 * npl + g16_off(4 * c4, npl);
    *reinterpret_cast<f16x4*>(d) = h;
    if (npl == 2) *reinterpret_cast<f16x4*>(d + 32) = l;
  }
}

int launch_colsum32_bsp(const float* in, int rows, float* partial, char* planes, int* E, int pl, hipStream_t st) {
  hipLaunchKernelGGL(colsum32_bsp_kernel, dim3((rows + 255) / 256), dim3(256), 0, st, in, rows, partial, planes, E, pl);
  SNERF_LAUNCH_CHECK();
  return SNERF_OK;
}

}  // namespace bsp
}  // namespace snerf
