// Split-bf16 GEMM for gfx950: fp32 operands, fp32 result, fp32-level accuracy, contraction on the bf16
// matrix cores (v_mfma_f32_32x32x16_bf16, 16x the fp32 MFMA rate).
//
// Every fp32 operand element a is split ON THE FLY (in the tile loader, between the global load and the
// LDS store) into three bf16 planes a = hi + mid + lo (each the bf16 rounding of the running residual;
// residuals are exact in fp32, so the planes carry 24 significant bits).  A product a*b is then the sum of
// the six plane products hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid -- each exact in the fp32
// accumulator (8 x 8 bit) -- and the dropped terms are <= 2^-26 |a b|, below fp32's own rounding.  Six bf16
// MFMAs (6 x 32 cycles) replace eight fp32 MFMAs (8 x 64 cycles) per 32x32x16 block: 2.67x less matrix-
// pipe time at unchanged numerics, with HBM traffic and the fused epilogue (gemm_common.h) unchanged.
//
// LDS images (BK = 16, per plane):
//   KC source (k contiguous: activations X[M][K], weights W[N][K]): [row][16 k] bf16 = 32-B rows; the MFMA
//     fragment of lane l (row l & 31, k = 8 (l >> 5) ... + 7) is ONE ds_read_b128; the two 16-B halves of a
//     row are swapped on rows with bit 3 set so that every 16-lane group of the b128 read covers all 64 banks.
//   IC source (row contiguous: dZ / X read along the points for dW = dZ^T X): [16 k][128 rows] bf16, written
//     as is (ds_write_b64); the fragment is gathered by two ds_read_b64_tr_b16 (hardware transpose read:
//     per 16-lane group a 4 (k) x 16 (rows) block, delivered k-major per row).  Row index XOR 32 (k & 3)
//     spreads the four k rows of a block over the 64 banks.
#include "gemm_common.h"

namespace snerf {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

constexpr int XBI = 128, XBJ = 128, XWI = 64, XWJ = 64;
constexpr int PLANE_BYTES = XBI * BK * 2;            // 4096 B: one bf16 plane of a 128 x 16 operand tile
constexpr int OPERAND_BYTES = 3 * PLANE_BYTES;       // hi | mid | lo
constexpr int STAGE_BYTES = 2 * OPERAND_BYTES;       // A | B

// 4 fp32 -> 3 planes x 4 bf16 (8 bytes each)
__device__ __forceinline__ void split3(const float4 v, bf16x4& hi, bf16x4& mid, bf16x4& lo) {
  const float x[4] = {v.x, v.y, v.z, v.w};
#ifdef SNERF_ABL_NOSPLIT  // diagnostic: one conversion, planes not meaningful
  for (int i = 0; i < 4; ++i) { hi[i] = (__bf16)x[i]; mid[i] = hi[i]; lo[i] = hi[i]; }
  return;
#endif
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const __bf16 h = (__bf16)x[i];
    const float r1 = x[i] - (float)h;
    const __bf16 m = (__bf16)r1;
    const float r2 = r1 - (float)m;
    hi[i] = h; mid[i] = m; lo[i] = (__bf16)r2;
  }
}

// byte offset inside one plane
__device__ __forceinline__ int kc_off(int row, int k) {  // k multiple of 4
  return row * 32 + ((((k >> 3) ^ (row >> 3)) & 1) << 4) + ((k >> 2) & 1) * 8;
}
__device__ __forceinline__ int ic_off(int row, int k) {  // row multiple of 4
  return k * 256 + ((row ^ (32 * (k & 3))) << 1);
}

template <bool IC>
__device__ __forceinline__ void store_planes(const float4 (&v)[2], char* __restrict__ op, int t) {
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    int il, kl;
    tile_coord<XBI, IC>(t, r, il, kl);
    bf16x4 hi, mid, lo;
    split3(v[r], hi, mid, lo);
    const int o = IC ? ic_off(il, kl) : kc_off(il, kl);
    *reinterpret_cast<bf16x4*>(op + o) = hi;
    *reinterpret_cast<bf16x4*>(op + PLANE_BYTES + o) = mid;
    *reinterpret_cast<bf16x4*>(op + 2 * PLANE_BYTES + o) = lo;
  }
}

// MFMA operand fragment of one 32-row block (rows r0 .. r0+31 of the tile) from plane `pl`
template <bool IC>
__device__ __forceinline__ bf16x8 load_frag(const char* __restrict__ pl, int r0, int lane) {
  if (!IC) {
    const int row = r0 + (lane & 31), h = lane >> 5;
    return *reinterpret_cast<const bf16x8*>(pl + row * 32 + (((h ^ (row >> 3)) & 1) << 4));
  } else {
    // 16-lane group g: rows r0 + 16 (g & 1) ... + 15, k = 8 (g >> 1) ... + 7 in two 4-k blocks.
    // Lane 4q + p of the group supplies the address of k-row q, rows 4p ... 4p + 3.
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    const int rowb = r0 + 16 * (g & 1) + 4 * pp;
    const int k0 = 8 * (g >> 1) + q;
    typedef bf16x4 __attribute__((address_space(3))) * lds4_t;
    const bf16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4_t)(pl + ic_off(rowb, k0)));
    const bf16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4_t)(pl + ic_off(rowb, k0 + 4)));
    bf16x8 f;
    f[0] = lo4[0]; f[1] = lo4[1]; f[2] = lo4[2]; f[3] = lo4[3];
    f[4] = hi4[0]; f[5] = hi4[1]; f[6] = hi4[2]; f[7] = hi4[3];
    return f;
  }
}

// BPL: the B operand (weights) comes pre-split as bf16 planes (snerf_pack_params): each thread moves one 16-B chunk
// (8 k of one row) per plane from global to LDS with no conversion work.
template <bool IC, bool BPL>
__global__ __launch_bounds__(NT, 3) void gemm_x6_kernel(const KArgs p) {
  constexpr int MI = XWI / 32, NJ = XWJ / 32;
  constexpr int EPI_BYTES = epilogue_lds_floats(XWJ) * 4;
  constexpr int LDS_BYTES = (2 * STAGE_BYTES > EPI_BYTES) ? 2 * STAGE_BYTES : EPI_BYTES;
  __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES];

  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int wi0 = (wave >> 1) * XWI, wj0 = (wave & 1) * XWJ;
  __builtin_amdgcn_s_setprio(2);  // non-MFMA phases at raised priority (see gemm.hip)
  int ti, tj;
  int kBeg = 0, kEnd = p.K;
  float* C = p.C;
  if (p.k_split > 0) {
    int tile, split;
    split_tile_of_block(blockIdx.x, blockIdx.z, gridDim.x, gridDim.z, tile, split);
    ti = tile / p.tiles_j; tj = tile - ti * p.tiles_j;
    kBeg = split * p.k_split;
    kEnd = min(p.K, kBeg + p.k_split);
    C += (size_t)split * p.slab_stride;
  } else {
    tile_of_block(blockIdx.x, p.tiles_i, p.tiles_j, ti, tj);
  }
  const int i0 = ti * XBI, j0 = tj * XBJ;
  const int nkt = (kEnd - kBeg + BK - 1) / BK;

  f32x16 acc[MI][NJ];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int nj = 0; nj < NJ; ++nj)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][nj][r] = 0.f;

  const bool wave_live = (i0 + wi0 < p.I) && (j0 + wj0 < p.J);

  const int Ka1 = min(p.Ka, p.K);  // length of the first A segment
  const srd_t srdA = IC ? srd_krows(p.A, p.lda, kBeg, kEnd, p.I) : srd_rows(p.A, p.lda, i0, p.I, Ka1);
  const srd_t srdA2 = srd_rows(p.A2, p.lda2, i0, p.I, p.K - Ka1);
  const srd_t srdB = IC ? srd_krows(p.B, p.ldb, kBeg, kEnd, p.J) : srd_rows(p.B, p.ldb, j0, p.J, p.K);
  Loader<XBI, IC> la1, la2;
  Loader<XBJ, IC> lb1;
  la1.init(t, i0, p.I, p.lda);
  la2.init(t, i0, p.I, p.lda2);
  lb1.init(t, j0, p.J, p.ldb);
  const unsigned stepA = IC ? (unsigned)p.lda * 4u : 4u;
  const unsigned stepB = IC ? (unsigned)p.ldb * 4u : 4u;

  float4 ra[2], rb[2];
  u32x4 rbp[3];
#ifdef SNERF_ABL_APLANES
  u32x4 rap[3];
#endif
  const srd_t srdBp = make_srd(p.Bpl, p.bytesBpl);
  // tile of k-tile kt = contiguous 4 KB per plane at ((bt_k0/16 + kt) * bt_rows + bt_row0 + j0) * 32 bytes; thread t moves
  // chunk t (16 B).  Rows past the matrix end read the next k-tile's rows (or zeros past the buffer): those columns
  // are >= J and masked in the epilogue.
  const unsigned bp_base = ((unsigned)(p.bt_k0 >> 4) * (unsigned)p.bt_rows + (unsigned)(p.bt_row0 + j0)) * 32u + 16u * t;
  const unsigned bp_step = (unsigned)p.bt_rows * 32u;   // bytes per k-tile
  // branch-free (the loop body must stay ONE basic block for the MFMA / VALU interleave below): the A segment
  // is chosen with scalar selects; beyond kEnd every lane's offset is out of bounds (zeros, no traffic)
  auto fetch = [&](int k0) {
#ifdef SNERF_ABL_APLANES
    {
      const unsigned o = (k0 < kEnd) ? ((unsigned)(k0 >> 4) * (unsigned)p.I + (unsigned)i0) * 32u + 16u * t : OOB;
      for (int pl = 0; pl < 3; ++pl) rap[pl] = __builtin_amdgcn_raw_buffer_load_b128(srdA, o == OOB ? OOB : o + pl * ((unsigned)p.I * (unsigned)p.K * 2u / 3u & ~15u), 0, 0);
    }
#endif
    const bool s2 = k0 >= p.Ka;
    const srd_t sA = s2 ? srdA2 : srdA;
    const unsigned kbA = s2 ? (unsigned)(k0 - p.Ka) * 4u : (unsigned)(IC ? k0 - kBeg : k0) * stepA;
    const int kremA = s2 ? kEnd - k0 : min(kEnd, p.Ka) - k0;
#ifndef SNERF_ABL_APLANES
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const unsigned base = s2 ? la2.base[r] : la1.base[r];
      ra[r] = buf_load4(sA, (base != OOB && la1.kl[r] < kremA) ? base + kbA : OOB);
    }
#endif
    if (BPL) {
      const unsigned o = (k0 < kEnd) ? bp_base + (unsigned)((k0 - kBeg) >> 4) * bp_step + (unsigned)(kBeg >> 4) * bp_step : OOB;
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
        rbp[pl] = __builtin_amdgcn_raw_buffer_load_b128(srdBp, o == OOB ? OOB : o + pl * p.pl_stride_bytes, 0, 0);
    } else {
      lb1.load(rb, srdB, (unsigned)(IC ? k0 - kBeg : k0) * stepB, kEnd - k0);
    }
  };
  auto store_b = [&](char* dst) {
    if (BPL) {
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<u32x4*>(dst + pl * PLANE_BYTES + 16 * t) = rbp[pl];
    } else {
      store_planes<IC>(rb, dst, t);
    }
  };
  // Software pipeline: at the start of iteration kt the registers hold tile kt+1 (requested at the end of
  // iteration kt-1); it is split and stored into the idle LDS stage underneath the second half of this
  // iteration's MFMAs, then the loads of tile kt+2 are issued.
  fetch(kBeg);
  store_planes<IC>(ra, lds, t);
  store_b(lds + OPERAND_BYTES);
  fetch(kBeg + BK);
  __syncthreads();

  __builtin_amdgcn_s_setprio(0);
  for (int kt = 0; kt < nkt; ++kt) {
    const char* sa = lds + (kt & 1) * STAGE_BYTES;
    const char* sb = sa + OPERAND_BYTES;
    char* da = lds + ((kt + 1) & 1) * STAGE_BYTES;
    // (Waves outside the problem compute on zero tiles; for the last tile the stores rewrite stale registers
    // into the idle stage, which nobody reads again.)
    bf16x8 a[3][MI], b[3][NJ];
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) a[pl][mi] = load_frag<IC>(sa + pl * PLANE_BYTES, wi0 + 32 * mi, lane);
#pragma unroll
      for (int nj = 0; nj < NJ; ++nj) b[pl][nj] = load_frag<IC>(sb + pl * PLANE_BYTES, wj0 + 32 * nj, lane);
    }
    // six plane products per 32x32x16 block, smallest terms first, the dominant hi*hi last
    auto block = [&](int mi, int nj) {
      f32x16 c = acc[mi][nj];
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][mi], b[1][nj], c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][mi], b[2][nj], c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2][mi], b[0][nj], c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][mi], b[1][nj], c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][mi], b[0][nj], c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][mi], b[0][nj], c, 0, 0, 0);
      acc[mi][nj] = c;
    };
    block(0, 0);
    block(0, 1);
#if defined(SNERF_ABL_NOLDSSTORE)
    for (int r = 0; r < 2; ++r) { asm volatile("" :: "v"(ra[r].x), "v"(ra[r].y), "v"(ra[r].z), "v"(ra[r].w)); asm volatile("" :: "v"(rb[r].x), "v"(rb[r].y), "v"(rb[r].z), "v"(rb[r].w)); }
#elif !defined(SNERF_ABL_NOGLOAD)
#ifdef SNERF_ABL_APLANES
    for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<u32x4*>(da + pl * PLANE_BYTES + 16 * t) = rap[pl];
#else
    store_planes<IC>(ra, da, t);
#endif
    store_b(da + OPERAND_BYTES);
#endif
    block(1, 0);
    block(1, 1);
#if !defined(SNERF_ABL_NOGLOAD) && !defined(SNERF_ABL_NOVMEM)
    fetch(kBeg + (kt + 2) * BK);
#endif
    // Pin the interleave in the emitted code: fragment reads, 12 MFMAs, then per remaining MFMA a slice of the
    // split (VALU) and LDS-store work of tile kt+1, then the loads of tile kt+2.
    __builtin_amdgcn_sched_group_barrier(0x100, IC ? 24 : 12, 0);  // DS reads
    __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);            // MFMA x 12
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);           // 1 MFMA
      __builtin_amdgcn_sched_group_barrier(0x002, BPL ? 6 : 12, 0); // VALU slice
      __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);           // 1 DS write
    }
    __builtin_amdgcn_sched_group_barrier(0x020, 4, 0);             // VMEM reads
    __syncthreads();
  }
  __builtin_amdgcn_s_setprio(2);
  if (!wave_live) return;
#ifdef SNERF_ABL_NOEPI
  { float sum = 0.f; for (int mi = 0; mi < MI; ++mi) for (int nj = 0; nj < NJ; ++nj) for (int r = 0; r < 16; ++r) sum += acc[mi][nj][r];
    if (sum == 12345.678f) C[0] = sum; return; }
#endif
  gemm_epilogue<MI, NJ, XWJ>(acc, reinterpret_cast<float*>(lds), wave, lane, i0 + wi0, j0 + wj0, p, C);
}

void launch_x6(bool ic, bool b_planes, const KArgs& p, dim3 grid, hipStream_t stream) {
  if (ic) hipLaunchKernelGGL((gemm_x6_kernel<true, false>), grid, dim3(NT), 0, stream, p);
  else if (b_planes) hipLaunchKernelGGL((gemm_x6_kernel<false, true>), grid, dim3(NT), 0, stream, p);
  else hipLaunchKernelGGL((gemm_x6_kernel<false, false>), grid, dim3(NT), 0, stream, p);
}

}  // namespace snerf
