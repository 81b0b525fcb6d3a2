// Split-plane GEMMs for gfx950: fp32 operands, fp32 result, fp32-class accuracy, contraction on the 16-bit matrix cores
// (v_mfma_f32_32x32x16_bf16, 16x the fp32 MFMA rate).  These kernels serve SNERF_FLAG_SPLIT3 and the reduced bf16 modes;
// the default arithmetic (fp16 planes, block exponents) lives in bsp_gemm.hip.
//
//   NP 3 (SNERF_FLAG_SPLIT3): every operand element is split ON THE FLY (in the tile loader, between the global load
//     and the LDS store) into three bf16 planes a = hi + mid + lo (each the bf16 rounding of the running residual;
//     residuals are exact in fp32, so the planes carry 24 significant bits); a product is the six plane products
//     hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid -- each exact in the fp32 accumulator -- and the dropped terms are
//     <= 2^-26 |a b|, below fp32's own rounding.  NP 2 / NP 1 are the REDUCED modes.
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
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

// Two tile configurations share the kernel: BT = 128 (128 x 128 tile, 4 waves of 64 x 64, 3 workgroups per CU)
// and BT = 256 (256 x 256 tile, 8 waves of 128 x 64 in a 2 x 4 grid, one workgroup per CU = 2 waves per SIMD with
// a 256-register budget).  The big tile halves the LDS-read, split (VALU) and L2 traffic per MFMA.
template <int BT> struct TileCfg {
  static constexpr int NTH = BT == 256 ? 512 : 256;
  static constexpr int WAVES = NTH / 64;
  static constexpr int WAVES_J = BT / 64;              // waves along the columns
  static constexpr int WI = BT / 2, WJ = 64;           // wave tile
  static constexpr int MI = WI / 32, NJ = WJ / 32;
  static constexpr int PLANE_BYTES = BT * BK * 2;      // one bf16 plane of a BT x 16 operand tile
  static constexpr int MIN_WG = BT == 256 ? 1 : 3;
};

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
template <int BT>
__device__ __forceinline__ int ic_off(int row, int k) {  // row multiple of 4
  return k * (2 * BT) + ((row ^ (32 * (k & 3))) << 1);
}

template <bool IC, int NP, int BT>
__device__ __forceinline__ void store_planes(const float4 (&v)[2], char* __restrict__ op, int t) {
  using T = TileCfg<BT>;
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    int il, kl;
    tile_coord<BT, IC, T::NTH>(t, r, il, kl);
    bf16x4 hi, mid, lo;
    split3(v[r], hi, mid, lo);
    const int o = IC ? ic_off<BT>(il, kl) : kc_off(il, kl);
    *reinterpret_cast<bf16x4*>(op + o) = hi;
    if (NP > 1) *reinterpret_cast<bf16x4*>(op + T::PLANE_BYTES + o) = mid;
    if (NP > 2) *reinterpret_cast<bf16x4*>(op + 2 * T::PLANE_BYTES + o) = lo;
  }
}

// MFMA operand fragment of one 32-row block (rows r0 .. r0+31 of the tile) from plane `pl`
template <bool IC, int BT>
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
    const bf16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4_t)(pl + ic_off<BT>(rowb, k0)));
    const bf16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4_t)(pl + ic_off<BT>(rowb, k0 + 4)));
    bf16x8 f;
    f[0] = lo4[0]; f[1] = lo4[1]; f[2] = lo4[2]; f[3] = lo4[3];
    f[4] = hi4[0]; f[5] = hi4[1]; f[6] = hi4[2]; f[7] = hi4[3];
    return f;
  }
}

// BPL: the B operand (weights) comes pre-split as bf16 planes (snerf_pack_params): each thread moves one 16-B chunk
// (8 k of one row) per plane from global to LDS with no conversion work.
// NP bf16 planes per operand: 3 = fp32-class (six products, the default); 2 = hi | mid with the three products
// hh, hm, mh (~16 significant bits); 1 = plain bf16 operands, one product.
template <bool IC, bool BPL, int NP, int BT>
__global__ __launch_bounds__(TileCfg<BT>::NTH, TileCfg<BT>::MIN_WG) void gemm_x6_kernel(const KArgs p) {
  using T = TileCfg<BT>;
  constexpr int MI = T::MI, NJ = T::NJ, NTH = T::NTH;
  constexpr int PLANE_BYTES = T::PLANE_BYTES;
  constexpr int OPERAND_BYTES = NP * PLANE_BYTES;      // hi | mid | lo
  constexpr int STAGE_BYTES = 2 * OPERAND_BYTES;       // A | B
  constexpr int EPI_BYTES = epilogue_lds_floats(T::WJ, T::WAVES) * 4;
  constexpr int LDS_BYTES = (2 * STAGE_BYTES > EPI_BYTES) ? 2 * STAGE_BYTES : EPI_BYTES;
  __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES];

  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int wi0 = (wave / T::WAVES_J) * T::WI, wj0 = (wave % T::WAVES_J) * T::WJ;
#ifdef SNERF_ABL_CLOCK
  const unsigned long long clk0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
#endif
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
  const int i0 = ti * BT, j0 = tj * BT;
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
  Loader<BT, IC, NTH> la1, la2, lb1;
  la1.init(t, i0, p.I, p.lda);
  la2.init(t, i0, p.I, p.lda2);
  lb1.init(t, j0, p.J, p.ldb);
  const unsigned stepA = IC ? (unsigned)p.lda * 4u : 4u;
  const unsigned stepB = IC ? (unsigned)p.ldb * 4u : 4u;

  // One tile in flight = the registers of one global -> LDS hop (A as fp32, B as fp32 or as bf16 planes).
  struct Tile { float4 ra[2]; float4 rb[2]; u32x4 rbp[NP]; };
  const srd_t srdBp = make_srd(p.Bpl, p.bytesBpl);
  // tile of k-tile kt = contiguous PLANE_BYTES per plane at ((bt_k0/16 + kt) * bt_rows + bt_row0 + j0) * 32 bytes; thread
  // t moves chunk t (16 B).  Rows past the matrix end read the next k-tile's rows (or zeros past the buffer): those
  // columns are >= J and masked in the epilogue.
  const unsigned bp_base = ((unsigned)(p.bt_k0 >> 4) * (unsigned)p.bt_rows + (unsigned)(p.bt_row0 + j0)) * 32u + 16u * t;
  const unsigned bp_step = (unsigned)p.bt_rows * 32u;   // bytes per k-tile
  // branch-free (the loop body must stay ONE basic block for the MFMA / VALU interleave below): the A segment
  // is chosen with scalar selects; beyond kEnd every lane's offset is out of bounds (zeros, no traffic)
  auto fetch = [&](int k0, Tile& r) {
    const bool s2 = k0 >= p.Ka;
    const srd_t sA = s2 ? srdA2 : srdA;
    const unsigned kbA = s2 ? (unsigned)(k0 - p.Ka) * 4u : (unsigned)(IC ? k0 - kBeg : k0) * stepA;
    const int kremA = s2 ? kEnd - k0 : min(kEnd, p.Ka) - k0;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const unsigned base = s2 ? la2.base[q] : la1.base[q];
      r.ra[q] = buf_load4(sA, (base != OOB && la1.kl[q] < kremA) ? base + kbA : OOB);
    }
    if (BPL) {
      const unsigned o = (k0 < kEnd) ? bp_base + (unsigned)(k0 >> 4) * bp_step : OOB;
#pragma unroll
      for (int pl = 0; pl < NP; ++pl)
        r.rbp[pl] = __builtin_amdgcn_raw_buffer_load_b128(srdBp, o == OOB ? OOB : o + pl * p.pl_stride_bytes, 0, 0);
    } else {
      lb1.load(r.rb, srdB, (unsigned)(IC ? k0 - kBeg : k0) * stepB, kEnd - k0);
    }
  };
  auto store = [&](const Tile& r, char* stage) {
    store_planes<IC, NP, BT>(r.ra, stage, t);
    if (BPL) {
#pragma unroll
      for (int pl = 0; pl < NP; ++pl) *reinterpret_cast<u32x4*>(stage + OPERAND_BYTES + pl * PLANE_BYTES + 16 * t) = r.rbp[pl];
    } else {
      store_planes<IC, NP, BT>(r.rb, stage + OPERAND_BYTES, t);
    }
  };
  // One k-tile: MFMAs on LDS stage kt & 1; `r` (tile kt+1, requested a full iteration earlier) is split and stored
  // into the idle stage underneath the second half of the MFMAs; then `r` is re-used to request tile kt+3.  Two
  // register tiles alternate, so every global load has ~1.5 iterations to land before its registers are read.
  auto ktile = [&](int kt, Tile& r) {
    const char* sa = lds + (kt & 1) * STAGE_BYTES;
    const char* sb = sa + OPERAND_BYTES;
    char* dst = lds + ((kt + 1) & 1) * STAGE_BYTES;
    // A fragments live in two rolling register sets (one 32-row block each): block row mi + 2 is read from LDS
    // as soon as the MFMAs of block row mi have been issued, underneath the MFMAs of block row mi + 1.
    bf16x8 b[NP][NJ], a[2][NP];
    auto load_a = [&](int mi) {
#pragma unroll
      for (int pl = 0; pl < NP; ++pl) a[mi & 1][pl] = load_frag<IC, BT>(sa + pl * PLANE_BYTES, wi0 + 32 * mi, lane);
    };
#pragma unroll
    for (int pl = 0; pl < NP; ++pl)
#pragma unroll
      for (int nj = 0; nj < NJ; ++nj) b[pl][nj] = load_frag<IC, BT>(sb + pl * PLANE_BYTES, wj0 + 32 * nj, lane);
    load_a(0);
    load_a(1);
    // plane products per 32x32x16 block (six for NP = 3), smallest terms first, the dominant hi*hi last
    auto block = [&](int mi, int nj) {
      const bf16x8 (&am)[NP] = a[mi & 1];
      f32x16 c = acc[mi][nj];
      if constexpr (NP > 2) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[1], b[1][nj], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[0], b[2][nj], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[2], b[0][nj], c, 0, 0, 0);
      }
      if constexpr (NP > 1) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[0], b[1][nj], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[1], b[0][nj], c, 0, 0, 0);
      }
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[0], b[0][nj], c, 0, 0, 0);
      acc[mi][nj] = c;
    };
#ifdef SNERF_ABL_PRIO
    __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      if (mi == MI / 2) {
#ifdef SNERF_ABL_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
#if defined(SNERF_ABL_NOLDSSTORE)
        for (int q = 0; q < 2; ++q) { asm volatile("" :: "v"(r.ra[q].x), "v"(r.ra[q].y), "v"(r.ra[q].z), "v"(r.ra[q].w)); }
#elif !defined(SNERF_ABL_NOGLOAD)
        store(r, dst);
#endif
#ifdef SNERF_ABL_PRIO
        __builtin_amdgcn_s_setprio(1);
#endif
      }
#pragma unroll
      for (int nj = 0; nj < NJ; ++nj) block(mi, nj);
      if (mi + 2 < MI) load_a(mi + 2);
    }
#ifdef SNERF_ABL_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
#if !defined(SNERF_ABL_NOGLOAD) && !defined(SNERF_ABL_NOVMEM)
    fetch(kBeg + (kt + 3) * BK, r);
#endif
    // Pin the interleave in the emitted code (BT = 128): fragment reads, the first half of the MFMAs, then per
    // remaining MFMA a slice of the split (VALU) and LDS-store work of tile kt+1, then the loads of tile kt+3.
    if constexpr (NP == 3 && BT == 128) {
      __builtin_amdgcn_sched_group_barrier(0x100, IC ? 24 : 12, 0);  // DS reads
      __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);            // MFMA x 12
#pragma unroll
      for (int i = 0; i < 12; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);           // 1 MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, BPL ? 6 : 12, 0); // VALU slice
        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);           // 1 DS write
      }
      __builtin_amdgcn_sched_group_barrier(0x020, 4, 0);             // VMEM reads
    }
    __syncthreads();
  };

  Tile r0, r1;
  fetch(kBeg, r0);
  store(r0, lds);
  fetch(kBeg + BK, r0);
  fetch(kBeg + 2 * BK, r1);
  __syncthreads();

#ifdef SNERF_ABL_CLOCK
  const unsigned long long clk1 = __builtin_amdgcn_s_memtime();
#endif
  __builtin_amdgcn_s_setprio(0);
  // k-tiles in pairs (an odd count runs one extra tile of zeros: fetches beyond kEnd return zeros)
  for (int kt = 0; kt < nkt; kt += 2) {
    ktile(kt, r0);
    ktile(kt + 1, r1);
  }
#ifdef SNERF_ABL_CLOCK
  const unsigned long long clk2 = __builtin_amdgcn_s_memtime();
#endif
  __builtin_amdgcn_s_setprio(2);
  if (!wave_live) return;
#ifdef SNERF_ABL_NOEPI
  { float sum = 0.f; for (int mi = 0; mi < MI; ++mi) for (int nj = 0; nj < NJ; ++nj) for (int r = 0; r < 16; ++r) sum += acc[mi][nj][r];
    if (sum == 12345.678f) C[0] = sum; return; }
#endif
  gemm_epilogue_dispatch<MI, NJ, T::WJ>(acc, reinterpret_cast<float*>(lds), wave, lane, i0 + wi0, j0 + wj0, p, C, 1.f);
#ifdef SNERF_ABL_CLOCK
  {  // diagnostic build: cycle stamps of this workgroup's phases into the (otherwise unused) colsum buffer
    const unsigned long long clk3 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long clk4 = __builtin_amdgcn_s_memtime();
    if (t == 0 && p.colsum != nullptr) {
      unsigned long long* dbg = reinterpret_cast<unsigned long long*>(p.colsum) + 8 * (size_t)blockIdx.x;
      dbg[0] = clk1 - clk0; dbg[1] = clk2 - clk1; dbg[2] = clk3 - clk2; dbg[3] = clk4 - clk3;
      dbg[4] = rt0; dbg[5] = __builtin_amdgcn_s_memrealtime(); dbg[6] = clk4 - clk0;
    }
  }
#endif
}

template <int NP, int BT>
static void launch_np(bool ic, bool b_planes, const KArgs& p, dim3 grid, hipStream_t stream) {
  const dim3 block(TileCfg<BT>::NTH);
  if (ic) hipLaunchKernelGGL((gemm_x6_kernel<true, false, NP, BT>), grid, block, 0, stream, p);
  else if (b_planes) hipLaunchKernelGGL((gemm_x6_kernel<false, true, NP, BT>), grid, block, 0, stream, p);
  else hipLaunchKernelGGL((gemm_x6_kernel<false, false, NP, BT>), grid, block, 0, stream, p);
}

// tile: 128 or 256 (p.tiles_i / p.tiles_j and the grid must have been computed for it)
void launch_x6(bool ic, bool b_planes, int planes, int tile, const KArgs& p, dim3 grid, hipStream_t stream) {
  if (tile == 256) {
    if (planes == 1) launch_np<1, 256>(ic, b_planes, p, grid, stream);
    else if (planes == 2) launch_np<2, 256>(ic, b_planes, p, grid, stream);
    else launch_np<3, 256>(ic, b_planes, p, grid, stream);
  } else {
    if (planes == 1) launch_np<1, 128>(ic, b_planes, p, grid, stream);
    else if (planes == 2) launch_np<2, 128>(ic, b_planes, p, grid, stream);
    else launch_np<3, 128>(ic, b_planes, p, grid, stream);
  }
}

}  // namespace snerf
