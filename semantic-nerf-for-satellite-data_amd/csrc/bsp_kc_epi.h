// Epilogue helpers of the K-contiguous GEMMs (bsp_kc.hip): the half-turn sine, DPP reductions.
#pragma once
#include "bsp_dev.h"

namespace snerf {
namespace bsp {

constexpr unsigned OOBH = 0x80000000u;           // rejected voffset that survives the addition of an instruction offset

constexpr float INV_2PI = 0.15915494309189533577f;

// sin(2 pi x_c) in place for eight values x_c in REVOLUTIONS (the FMA that applies scale and bias forms them), by v_sin_f32 on the
// unreduced argument: inside the instruction's domain (|x| <= 256 revolutions) its own range reduction is exact -- measured on
// gfx950 against fp64, bit for bit the results of v_sin_f32(v_fract_f32(x)) (tools/ablate/vsin_range.hip) -- and a SIREN
// pre-activation of |w0 z| <= 1,608 rad lies inside it; KcArgs::sin_wide (host: |w0| > 30) keeps the explicit v_fract_f32 for
// layers that may leave it.
// SIGNS: bit "cos(2 pi x_c) < 0" (= parity of round(2 x_c), the low mantissa bit of 2 x_c + 1.5 * 2^23) enters `sw` from the top,
// earlier bits move down (after 32 calls' worth the first element sits in bit 0).
constexpr int SIN_FRACT = 0, SIN_DIRECT = 1;
template <bool SIGNS, int SINM>
__device__ __forceinline__ void sin2pi8(float (&x)[8], unsigned& sw) {
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    if (SIGNS) {
      const float t = fmaf(x[c], 2.f, 12582912.f);       // low mantissa bit = parity of k = round(2 x)
      sw = __builtin_amdgcn_alignbit(__float_as_uint(t), sw, 1);
    }
    x[c] = __builtin_amdgcn_sinf(SINM == SIN_FRACT ? __builtin_amdgcn_fractf(x[c]) : x[c]);
  }
  // gfx940+: a VALU instruction may not read a transcendental's result in the very next issue slot.  The compiler pads the readers
  // it can see; the plane split that consumes these values is made of asm statements (bsp.h: split8 / cvt8), which it cannot.  One
  // wait state behind the eight sines, tied to all eight values, keeps every reader -- asm or not -- at least one slot away
  // (tools/check_vgpr_hazards.py scan 3 holds the generated code to it).
  asm volatile("s_nop 0" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]));
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
