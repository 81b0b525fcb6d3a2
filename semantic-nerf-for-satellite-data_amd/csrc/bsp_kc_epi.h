// Epilogue helpers of the K-contiguous GEMMs (bsp_kc.hip): the half-turn sine, DPP reductions.
#pragma once
#include "bsp_dev.h"

namespace snerf {
namespace bsp {

constexpr float INV_PI = 0.31830988618379067154f;
constexpr unsigned OOBH = 0x80000000u;           // rejected voffset that survives the addition of an instruction offset

constexpr int SIN_POLY = 0, SIN_HW = 1;

// sin(pi u_c) in place for eight values.  SIGNS: bit "cos(pi u_c) < 0" (= parity of round(u_c)) enters `sw` from the top,
// earlier bits move down (after 32 calls' worth the first element sits in bit 0).
template <bool SIGNS, int SINM>
__device__ __forceinline__ void sinpi8(float (&u)[8], unsigned& sw) {
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const float t = u[c] + 12582912.f;                 // low mantissa bits = k = round(u)
    const unsigned tb = __float_as_uint(t);
    float s;
    if (SINM == SIN_HW) {
      float fr;
      asm("v_fract_f32 %0, %1" : "=v"(fr) : "v"(0.5f * u[c]));
      asm("v_sin_f32 %0, %1" : "=v"(s) : "v"(fr));
    } else {
      const float kf = t - 12582912.f;
      const float f = u[c] - kf;                       // exact, |f| <= 1/2
      const float f2 = f * f;
      float q = fmaf(f2, 0.077218386155008978f, -0.59804419391100816f);
      q = fmaf(q, f2, 2.5500311935191413f);
      q = fmaf(q, f2, -5.1677068661679284f);
      q = fmaf(q, f2, 3.1415925798055815f);
      s = __uint_as_float((tb << 31) + __float_as_uint(f * q));   // (-1)^k: one v_lshl_add
    }
    u[c] = s;
    if (SIGNS) sw = __builtin_amdgcn_alignbit(tb, sw, 1);
  }
}

// sum over the 32 lanes that share l >> 5 (DPP adds inside the 16-lane rows, row_bcast15 across the pair of rows); valid in
// lanes 16-31 (l >> 5 == 0) and 48-63 (l >> 5 == 1)
__device__ __forceinline__ float sum32(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));   // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));   // row_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xA, 0xF, false));  // row_bcast15 -> rows 1, 3
  return v;
}

// maximum over the wave of non-negative values (DPP inside the 16-lane rows, then the four rows through scalar registers:
// no lane-index registers to keep alive as ds_bpermute shuffles need)
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true)));
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true)));
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true)));
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true)));
  const int b = __builtin_bit_cast(int, v);
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
  return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}

}  // namespace bsp
}  // namespace snerf
