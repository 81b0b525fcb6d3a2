// Pieces shared by the fp32-MFMA GEMM (gemm.hip) and the split-bf16 GEMM (gemm_x6.hip):
// kernel arguments, buffer-descriptor helpers, workgroup->tile map and the fused epilogue.
#pragma once
#include "../gemm.h"
#include "../tiles.h"
#include "../../../include/snerf_hip.h"

namespace snerf {

constexpr int NT = 256;
constexpr int BK = 16;
constexpr unsigned OOB = 0xFFFFFFF0u;  // voffset that the buffer bounds check always rejects

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t srd_t;

__device__ __forceinline__ srd_t make_srd(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ float4 buf_load4(srd_t s, unsigned off) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(s, off, 0, 0);
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

struct KArgs {
  const float* A; const float* A2; const float* B;
  float* C; float* C2;
  const float* bias; const float* aux; float* colsum;
  unsigned* C2s; const unsigned* auxs; int sign_col0, sign_groups;
  unsigned epi_mask;   // which (act, aux) kinds take the straight-line epilogue: bit 0 plain, 1 sin, 2 relu, 3 sinrec, 4 relu mask
  unsigned bytesA, bytesA2, bytesB;
  const unsigned short* Bpl; unsigned pl_stride_bytes; unsigned bytesBpl; int bt_rows, bt_row0, bt_k0;
  int lda, lda2, Ka, ldb, I, J, K, ldc, ldaux, ldcs;
  int act, aux_mode;
  float w0;
  int k_split;
  unsigned long long slab_stride;
  int tiles_i, tiles_j;
};

typedef unsigned int u32x4s __attribute__((ext_vector_type(4)));
__device__ __forceinline__ size_t uniform_sz(size_t v) {
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return ((size_t)hi << 32) | lo;
}
// |cos| from the stored sine: hardware square root (1 ulp) of the once-rounded 1 - h^2; sqrtf() would expand into the
// ~15-instruction correctly rounded sequence, four times per lane and pass
__device__ __forceinline__ float cos_from_sin(float h) { return __builtin_amdgcn_sqrtf(fmaxf(fmaf(-h, h, 1.f), 0.f)); }

// thread -> (row, k) of its r-th float4 in a BI x BK k-tile (element e = t + NT*r of BI*BK/4).
// KC: BK/4 lanes cover one row's BK floats; IC: BI/4 lanes cover one k-row.
template <int BI, bool IC, int NTH = NT>
__device__ __forceinline__ bool tile_coord(int t, int r, int& il, int& kl) {
  const int e = t + NTH * r;
  if (IC) {
    constexpr int V = BI / 4;
    il = 4 * (e % V);
    kl = e / V;
  } else {
    constexpr int Q = BK / 4;
    il = e / Q;
    kl = 4 * (e % Q);
  }
  return e < BI * BK / 4;
}

// per-thread loader state of one operand.  Offsets are relative to a per-workgroup descriptor base (the tile's
// first row for KC operands, the K-split's first k-row for IC operands), so 32-bit offsets never span more than one
// tile / one split however large the operand is.
template <int BI, bool IC, int NTH = NT>
struct Loader {
  static constexpr int NV = (BI * BK / 4 + NTH - 1) / NTH;
  unsigned base[NV];  // byte offset of the float4 at the first k of the range, or OOB if its row is outside
  int kl[NV];
  __device__ __forceinline__ void init(int t, int i0, int I, int ld) {
#pragma unroll
    for (int r = 0; r < NV; ++r) {
      int il, k;
      const bool in = tile_coord<BI, IC, NTH>(t, r, il, k);
      kl[r] = k;
      const unsigned o = IC ? ((unsigned)k * (unsigned)ld + (unsigned)(i0 + il)) * 4u : ((unsigned)il * (unsigned)ld + (unsigned)k) * 4u;
      base[r] = (in && i0 + il < I) ? o : OOB;
    }
  }
  // kbytes: byte offset of the tile's first k relative to the descriptor base; krem: valid k's left
  __device__ __forceinline__ void load(float4 (&v)[NV], srd_t srd, unsigned kbytes, int krem) const {
#pragma unroll
    for (int r = 0; r < NV; ++r) {
      const unsigned o = (base[r] != OOB && kl[r] < krem) ? base[r] + kbytes : OOB;
      v[r] = buf_load4(srd, o);
    }
  }
};

// descriptor of a K-contiguous operand (element (i,k) at P[i*ld + k]) based at row i0
__device__ __forceinline__ srd_t srd_rows(const float* P, int ld, int i0, int I, int Kseg) {
  const long long rows = (long long)I - i0;
  const unsigned long long bytes = rows > 0 ? ((unsigned long long)(rows - 1) * ld + Kseg) * 4ull : 0ull;
  return make_srd(P + (size_t)i0 * ld, (unsigned)(bytes < 0xFFFFFFF0ull ? bytes : 0xFFFFFFF0ull));
}
// descriptor of a row-contiguous operand (element (i,k) at P[k*ld + i]) based at k-row kBeg
__device__ __forceinline__ srd_t srd_krows(const float* P, int ld, int kBeg, int kEnd, int I) {
  const long long ks = (long long)kEnd - kBeg;
  const unsigned long long bytes = ks > 0 ? ((unsigned long long)(ks - 1) * ld + I) * 4ull : 0ull;
  return make_srd(P + (size_t)kBeg * ld, (unsigned)(bytes < 0xFFFFFFF0ull ? bytes : 0xFFFFFFF0ull));
}

// Fused epilogue.  C/D layout of the 32x32 MFMA (fp32 and bf16 forms alike): col = lane & 31,
// row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5).  Each wave transposes its 32-row accumulator blocks through
// a private LDS strip [32][WJ+4] (written in MFMA layout: conflict-free b32; read back as rows: b128) and
// then works on whole rows: bias / aux loads and the C (+cos) stores are 16 B per lane, 256 contiguous
// bytes per row; bias-gradient column sums by a butterfly over the row bits.
template <int MI, int NJ, int WJ>
__device__ __forceinline__ void gemm_epilogue(const f32x16 (&acc)[MI][NJ], float* __restrict__ lds, int wave, int lane,
                                              int row0, int col0, const KArgs& p, float* __restrict__ C, float acc_scale = 1.f) {
  constexpr int EP = WJ + 4;
  constexpr int EPI = 32 * EP;
  const int i0 = row0, wi0 = 0, j0 = col0, wj0 = 0;
  float* strip = lds + wave * EPI;
  constexpr int LPR = WJ / 4;        // lanes per row
  constexpr int RPP = 64 / LPR;      // rows per pass
  const int lc = lane & 31, lh = lane >> 5;
  const int rrow = lane / LPR, c4 = 4 * (lane % LPR);
  const int col = j0 + wj0 + c4;
  const bool col_ok = col < p.J;     // J % 4 == 0 (host check): a float4 is entirely inside or outside
  float4 bj = make_float4(0.f, 0.f, 0.f, 0.f);
  if (p.bias != nullptr && col_ok) bj = *reinterpret_cast<const float4*>(p.bias + col);
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int rbase = i0 + wi0 + 32 * mi;
    if (rbase < p.I) {
#pragma unroll
      for (int nj = 0; nj < NJ; ++nj)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          strip[((r & 3) + 8 * (r >> 2) + 4 * lh) * EP + 32 * nj + lc] = acc[mi][nj][r] * acc_scale;
      float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);
      // sign words of the activation derivative (see GemmArgs::C2s): one word per lane and 32-row block
      unsigned sbits = 0u, sword = 0u;
      size_t sidx = 0;
      if (WJ == 64 && (p.C2s != nullptr || p.aux_mode == AUX_SINREC)) {
        const int cabs = col + p.sign_col0;
        sidx = ((size_t)(rbase >> 5) * p.sign_groups + (cabs >> 6)) * 64 + rrow * 16 + ((cabs >> 2) & 15);
        if (p.aux_mode == AUX_SINREC && col_ok) sword = p.auxs[sidx];
      }
#pragma unroll
      for (int ps = 0; ps < 32 / RPP; ++ps) {
        const int rl = rrow + RPP * ps;
        const int row = rbase + rl;
        const bool ok = col_ok && row < p.I;
        float4 v = *reinterpret_cast<const float4*>(&strip[rl * EP + c4]);
        v.x += bj.x; v.y += bj.y; v.z += bj.z; v.w += bj.w;
        const size_t off = (size_t)row * p.ldc + col;
        if (p.act == ACT_SIN && p.C2 == nullptr) {
          // sin and the sign of cos only (the derivative is rebuilt from h in the backward epilogue)
          unsigned nb;
          v = sin4_signcos(make_float4(p.w0 * v.x, p.w0 * v.y, p.w0 * v.z, p.w0 * v.w), &nb);
          if (WJ == 64 && p.C2s != nullptr) sbits |= nb << (4 * ps);
        } else if (p.act == ACT_SIN) {   // diagnostic form: w0*cos stored as floats (SNERF_DERIV=float)
          float4 cn;
          sincos_acc(p.w0 * v.x, &v.x, &cn.x);
          sincos_acc(p.w0 * v.y, &v.y, &cn.y);
          sincos_acc(p.w0 * v.z, &v.z, &cn.z);
          sincos_acc(p.w0 * v.w, &v.w, &cn.w);
          if (p.C2 != nullptr && ok) {
            cn.x *= p.w0; cn.y *= p.w0; cn.z *= p.w0; cn.w *= p.w0;
            *reinterpret_cast<float4*>(p.C2 + off) = cn;
          }
        } else if (p.act == ACT_RELU) {
          v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        }
        if (p.aux_mode != AUX_NONE) {
          float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
          if (ok) x = *reinterpret_cast<const float4*>(p.aux + (size_t)row * p.ldaux + col);
          if (p.aux_mode == AUX_MUL) { v.x *= x.x; v.y *= x.y; v.z *= x.z; v.w *= x.w; }
          else if (p.aux_mode == AUX_SINREC) {
            // x = h = sin(w0 z): |cos| = sqrt(1 - h^2) (one rounding: fma), sign from the stored bit
            const unsigned nib = sword >> (4 * ps);
            const float dx = p.w0 * cos_from_sin(x.x), dy = p.w0 * cos_from_sin(x.y);
            const float dz = p.w0 * cos_from_sin(x.z), dw = p.w0 * cos_from_sin(x.w);
            v.x *= (nib & 1u) ? -dx : dx; v.y *= (nib & 2u) ? -dy : dy;
            v.z *= (nib & 4u) ? -dz : dz; v.w *= (nib & 8u) ? -dw : dw;
          }
          else { v.x = x.x > 0.f ? v.x : 0.f; v.y = x.y > 0.f ? v.y : 0.f; v.z = x.z > 0.f ? v.z : 0.f; v.w = x.w > 0.f ? v.w : 0.f; }
        }
        if (ok) {
#ifdef SNERF_ABL_NOSTORE
          asm volatile("" :: "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));
#elif defined(SNERF_ABL_NTSTORE)
          __builtin_nontemporal_store(v.x, C + off); __builtin_nontemporal_store(v.y, C + off + 1);
          __builtin_nontemporal_store(v.z, C + off + 2); __builtin_nontemporal_store(v.w, C + off + 3);
#else
          *reinterpret_cast<float4*>(C + off) = v;
#endif
          cs.x += v.x; cs.y += v.y; cs.z += v.z; cs.w += v.w;
        }
      }
      if (WJ == 64 && p.C2s != nullptr && col_ok) p.C2s[sidx] = sbits;
#ifndef SNERF_ABL_CLOCK
      if (p.colsum != nullptr) {
        // rows of this 32-row block live on lanes with equal (lane % LPR): butterfly over the row bits
#pragma unroll
        for (int o = LPR; o < 64; o <<= 1) {
          cs.x += __shfl_xor(cs.x, o, 64); cs.y += __shfl_xor(cs.y, o, 64);
          cs.z += __shfl_xor(cs.z, o, 64); cs.w += __shfl_xor(cs.w, o, 64);
        }
        if (lane < LPR && col_ok) *reinterpret_cast<float4*>(p.colsum + (size_t)(rbase >> 5) * p.ldcs + col) = cs;
      }
#endif
    }
  }
}

// ---- straight-line epilogue (64-wide wave tiles) -----------------------------------------------------------------
// Same arithmetic and layouts as gemm_epilogue, specialised at compile time on (ACT, AUX) so that the pass loop has
// NO runtime branches: bounds are enforced by buffer descriptors (an out-of-range element gets the offset the hardware
// rejects), the aux operand of a 32-row block is requested in one batch before the accumulators go through the LDS
// strip, and the eight row stores of a block are issued back to back.  With branches in the loop the compiler has to
// fall back to s_waitcnt vmcnt(0) after every load / before every LDS read: each of the 16 stores and aux loads of a
// wave then cost a full memory round trip (measured: 24-32 k of a workgroup's 85-90 k cycles).
__device__ __forceinline__ void buf_store4(srd_t s, unsigned off, float4 v) {
  u32x4s d = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
  __builtin_amdgcn_raw_buffer_store_b128(d, s, off, 0, 0);
}

template <int MI, int NJ, int ACT, int AUX, bool COLSUM>
__device__ __forceinline__ void gemm_epilogue_fast(const f32x16 (&acc)[MI][NJ], float* __restrict__ lds, int wave, int lane,
                                                   int row0, int col0, const KArgs& p, float* __restrict__ C, float acc_scale) {
  constexpr int WJ = 64, EP = WJ + 4, EPI = 32 * EP, LPR = 16, RPP = 4, NPASS = 8;
  static_assert(NJ == 2, "64-wide wave tile");
  float* strip = lds + wave * EPI;
  const int lc = lane & 31, lh = lane >> 5;
  const int rrow = lane / LPR, c4 = 4 * (lane % LPR);
  const int col = col0 + c4;
  const bool col_ok = col < p.J;
  // descriptors based at the wave tile's first element; offsets are tile-relative (< 2^31 by launch_gemm's span check)
  // row0 / col0 derive from the wave id: wave-uniform, but only readfirstlane tells the compiler so -- a descriptor it
  // believes divergent is wrapped in a readfirstlane "waterfall" loop around every access
  const size_t offC = uniform_sz((size_t)row0 * p.ldc + col0), offA = uniform_sz((size_t)row0 * p.ldaux + col0);
  const srd_t srdC = make_srd(C + offC, 0xFFFFFFE0u);
  const srd_t srdAux = make_srd(AUX != AUX_NONE ? p.aux + offA : nullptr, AUX != AUX_NONE ? 0xFFFFFFE0u : 0u);
  float4 bj = make_float4(0.f, 0.f, 0.f, 0.f);
  if (p.bias != nullptr && col_ok) bj = *reinterpret_cast<const float4*>(p.bias + col);
  const bool signs_out = ACT == ACT_SIN && p.C2s != nullptr;
  const srd_t srdS = make_srd(signs_out ? p.C2s : nullptr, signs_out ? 0xFFFFFFE0u : 0u);
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int rbase = row0 + 32 * mi;
    // aux of the whole 32-row block in flight before the LDS round trip (offsets recomputed per pass: registers)
    auto okp = [&](int ps) { return col_ok && (row0 + 32 * mi + rrow + RPP * ps) < p.I; };
    float4 ax[AUX != AUX_NONE ? NPASS : 1];
    unsigned sword = 0u;
    const int cabs = col + p.sign_col0;
    const size_t sidx = ((size_t)(rbase >> 5) * p.sign_groups + (cabs >> 6)) * 64 + rrow * 16 + ((cabs >> 2) & 15);
    if (AUX != AUX_NONE) {
#pragma unroll
      for (int ps = 0; ps < NPASS; ++ps)
        ax[ps] = buf_load4(srdAux, okp(ps) ? ((unsigned)(32 * mi + rrow + RPP * ps) * (unsigned)p.ldaux + (unsigned)c4) * 4u : OOB);
      if (AUX == AUX_SINREC) sword = p.auxs[(col_ok && rbase < p.I) ? sidx : 0];
    }
#pragma unroll
    for (int nj = 0; nj < NJ; ++nj)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        strip[((r & 3) + 8 * (r >> 2) + 4 * lh) * EP + 32 * nj + lc] = acc[mi][nj][r] * acc_scale;
    float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);
    unsigned sbits = 0u;
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
      const bool ok = okp(ps);
      float4 w = *reinterpret_cast<const float4*>(&strip[(rrow + RPP * ps) * EP + c4]);
      w.x += bj.x; w.y += bj.y; w.z += bj.z; w.w += bj.w;
      if (ACT == ACT_SIN) {
        unsigned nb;
        w = sin4_signcos(make_float4(p.w0 * w.x, p.w0 * w.y, p.w0 * w.z, p.w0 * w.w), &nb);
        sbits |= nb << (4 * ps);
      } else if (ACT == ACT_RELU) {
        w.x = fmaxf(w.x, 0.f); w.y = fmaxf(w.y, 0.f); w.z = fmaxf(w.z, 0.f); w.w = fmaxf(w.w, 0.f);
      }
      if (AUX == AUX_SINREC) {
        const float4 x = ax[ps];
        const unsigned nib = sword >> (4 * ps);
        const float dx = p.w0 * cos_from_sin(x.x), dy = p.w0 * cos_from_sin(x.y);
        const float dz = p.w0 * cos_from_sin(x.z), dw = p.w0 * cos_from_sin(x.w);
        w.x *= (nib & 1u) ? -dx : dx; w.y *= (nib & 2u) ? -dy : dy;
        w.z *= (nib & 4u) ? -dz : dz; w.w *= (nib & 8u) ? -dw : dw;
      } else if (AUX == AUX_RELU_MASK) {
        const float4 x = ax[ps];
        w.x = x.x > 0.f ? w.x : 0.f; w.y = x.y > 0.f ? w.y : 0.f; w.z = x.z > 0.f ? w.z : 0.f; w.w = x.w > 0.f ? w.w : 0.f;
      }
      if (COLSUM) {   // out-of-range rows must not enter the column sums (the store itself is rejected by the descriptor)
        if (!ok) w = make_float4(0.f, 0.f, 0.f, 0.f);
        cs.x += w.x; cs.y += w.y; cs.z += w.z; cs.w += w.w;
      }
      buf_store4(srdC, ok ? ((unsigned)(32 * mi + rrow + RPP * ps) * (unsigned)p.ldc + (unsigned)c4) * 4u : OOB, w);
    }
    if (signs_out) __builtin_amdgcn_raw_buffer_store_b32(sbits, srdS, (col_ok && rbase < p.I) ? (unsigned)(sidx * 4) : OOB, 0, 0);
    if (COLSUM && p.colsum != nullptr) {   // wave-uniform
#pragma unroll
      for (int o = LPR; o < 64; o <<= 1) {
        cs.x += __shfl_xor(cs.x, o, 64); cs.y += __shfl_xor(cs.y, o, 64);
        cs.z += __shfl_xor(cs.z, o, 64); cs.w += __shfl_xor(cs.w, o, 64);
      }
      if (lane < LPR && col_ok && rbase < p.I) *reinterpret_cast<float4*>(p.colsum + (size_t)(rbase >> 5) * p.ldcs + col) = cs;
    }
  }
}

// Picks the straight-line epilogue for the (act, aux) pairs the passes use; anything else (the float-derivative
// diagnostics, AUX_MUL) goes through the general gemm_epilogue.
template <int MI, int NJ, int WJ>
__device__ __forceinline__ void gemm_epilogue_dispatch(const f32x16 (&acc)[MI][NJ], float* __restrict__ lds, int wave, int lane,
                                                       int row0, int col0, const KArgs& p, float* __restrict__ C,
                                                       float acc_scale = 1.f) {
  if constexpr (WJ == 64 && NJ == 2) {
#ifndef SNERF_ABL_OLDEPI
    const bool plain_store = p.C2 == nullptr && ((size_t)(MI * 32) * (size_t)(p.ldc > p.ldaux ? p.ldc : p.ldaux) * 4u < 0x7FFFFFFFu);
    if (plain_store && p.aux_mode == AUX_NONE && p.colsum == nullptr) {
      if (p.act == ACT_NONE && (p.epi_mask & 1u)) return gemm_epilogue_fast<MI, NJ, ACT_NONE, AUX_NONE, false>(acc, lds, wave, lane, row0, col0, p, C, acc_scale);
      if (p.act == ACT_SIN && (p.epi_mask & 2u)) return gemm_epilogue_fast<MI, NJ, ACT_SIN, AUX_NONE, false>(acc, lds, wave, lane, row0, col0, p, C, acc_scale);
      if (p.act == ACT_RELU && (p.epi_mask & 4u)) return gemm_epilogue_fast<MI, NJ, ACT_RELU, AUX_NONE, false>(acc, lds, wave, lane, row0, col0, p, C, acc_scale);
    } else if (plain_store && p.aux_mode == AUX_NONE && p.act == ACT_NONE) {   // plain store + bias-gradient column sums
      if (p.epi_mask & 1u) return gemm_epilogue_fast<MI, NJ, ACT_NONE, AUX_NONE, true>(acc, lds, wave, lane, row0, col0, p, C, acc_scale);
    } else if (plain_store && p.act == ACT_NONE) {
      if (p.aux_mode == AUX_SINREC && (p.epi_mask & 8u)) return gemm_epilogue_fast<MI, NJ, ACT_NONE, AUX_SINREC, true>(acc, lds, wave, lane, row0, col0, p, C, acc_scale);
      if (p.aux_mode == AUX_RELU_MASK && (p.epi_mask & 16u)) return gemm_epilogue_fast<MI, NJ, ACT_NONE, AUX_RELU_MASK, true>(acc, lds, wave, lane, row0, col0, p, C, acc_scale);
    }
#endif
  }
  gemm_epilogue<MI, NJ, WJ>(acc, lds, wave, lane, row0, col0, p, C, acc_scale);
}

constexpr int epilogue_lds_floats(int WJ, int waves = 4) { return waves * 32 * (WJ + 4); }

}  // namespace snerf
