// fp32 MFMA GEMM for gfx950 (v_mfma_f32_32x32x2_f32), LDS-tiled, register-prefetched, with the
// layer-specific work fused into the tile loader (two-segment A = skip concat) and the epilogue
// (bias + SIREN sincos / ReLU, activation-derivative multiply, bias-gradient column sums, split-K slabs).
//
// Replaces, per layer: torch.nn.Linear + Siren/ReLU of RSSemanticNeRF (semantic/models/rs_semantic.py:173-258,
// 260-340; baseline/models/commons.py:27-38) forward, and their autograd backward (dX, dW, db).
//
// Tiling (64-wide wavefronts): 256 threads = 4 waves; block tile BI x BJ, wave tile WI x WJ made of
// 32x32 MFMA tiles; BK = 16; ~35 KB LDS + <= 168 VGPR -> 3 workgroups per CU.  Both operands live in LDS as
// [k][row] (row contiguous) so that the MFMA operand fetch -- lane l wants (row = l & 31, k = l >> 5) --
// is a conflict-free ds_read_b32 for either global storage order:
//   KC source (k contiguous, e.g. activations X[M][K], weights W[N][K]): float4 along k from global
//     (4 lanes cover one row's 64-B piece), transposed on the LDS write (pitch BI+1 -> conflict-free b32 writes);
//   IC source (row contiguous, e.g. W read as B[k=n][j=kin] for dX, dZ/X read along m for dW):
//     float4 along rows, ds_write_b128 as is (pitch BI+4).
// Global loads are buffer loads (SRD in SGPRs, 32-bit per-lane offsets, hardware bounds check returns 0
// beyond the operand) so ragged tiles need no exec-mask branches in the k-loop.
// Epilogue: each wave transposes its 32-row accumulator blocks through a private LDS strip and then works
// on whole rows, so bias/aux loads and the C (+cos) stores are 16-B per lane, 256 contiguous bytes per row.
#include "gemm_common.h"   // legacy/

#include <vector>

namespace snerf {

void launch_x6(bool ic, bool b_planes, int planes, int tile, const KArgs& p, dim3 grid, hipStream_t stream);  // gemm_x6.hip

template <int BI, bool IC>
struct Tile {
  static constexpr int PITCH = IC ? BI + 4 : BI + 1;  // LDS row pitch in floats
  static constexpr int FLOATS = BK * PITCH;
  static constexpr int NV = (BI * BK / 4 + NT - 1) / NT;  // float4 per thread per k-tile
};

template <int BI, bool IC>
__device__ __forceinline__ void r2s(const float4 (&v)[Tile<BI, IC>::NV], float* __restrict__ lds, int t) {
  constexpr int PITCH = Tile<BI, IC>::PITCH;
#pragma unroll
  for (int r = 0; r < Tile<BI, IC>::NV; ++r) {
    int il, kl;
    const bool in = tile_coord<BI, IC>(t, r, il, kl);
    if (in) {
      if (IC) {
        *reinterpret_cast<float4*>(&lds[kl * PITCH + il]) = v[r];
      } else {
        lds[(kl + 0) * PITCH + il] = v[r].x;
        lds[(kl + 1) * PITCH + il] = v[r].y;
        lds[(kl + 2) * PITCH + il] = v[r].z;
        lds[(kl + 3) * PITCH + il] = v[r].w;
      }
    }
  }
}

template <int BI, int BJ, int WI, int WJ, bool A_IC, bool B_IC>
__global__ __launch_bounds__(NT, 3) void gemm_kernel(const KArgs p) {
  using TA = Tile<BI, A_IC>;
  using TB = Tile<BJ, B_IC>;
  constexpr int MI = WI / 32, NJ = WJ / 32;
  constexpr int WAVES_J = BJ / WJ;
  static_assert((BI / WI) * (BJ / WJ) == 4, "4 waves per workgroup");
  constexpr int STAGE = TA::FLOATS + TB::FLOATS;
  constexpr int EP = WJ + 4;             // pitch of the epilogue transpose strip
  constexpr int EPI = 32 * EP;           // floats per wave
  constexpr int LDS_FLOATS = (2 * STAGE > 4 * EPI) ? 2 * STAGE : 4 * EPI;
  __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];

  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int wi0 = (wave / WAVES_J) * WI, wj0 = (wave % WAVES_J) * WJ;
#ifdef SNERF_ABL_CLOCK  // diagnostic build only: shader clock = d(s_memtime) / d(s_memrealtime) * 100 MHz
  const unsigned long long clk_t0 = __builtin_amdgcn_s_memtime(), clk_r0 = __builtin_amdgcn_s_memrealtime();
#endif
  // Issue arbitration on a SIMD is by priority, then age: a freshly started workgroup would otherwise crawl
  // through its prologue behind three older MFMA-bound ones.  Non-MFMA phases (prologue, the per-tile
  // store+fetch block, epilogue) run at raised priority; the MFMA phase at 0.
  __builtin_amdgcn_s_setprio(2);
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
  const int i0 = ti * BI, j0 = tj * BJ;
  const int nkt = (kEnd - kBeg + BK - 1) / BK;

  f32x16 acc[MI][NJ];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int nj = 0; nj < NJ; ++nj)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][nj][r] = 0.f;

  // a wave whose whole WI x WJ tile lies outside the problem skips its MFMAs (wave-uniform);
  // partially covered tiles compute on zero-filled LDS rows and mask the stores instead.
  const bool wave_live = (i0 + wi0 < p.I) && (j0 + wj0 < p.J);

  const int Ka1 = min(p.Ka, p.K);  // length of the first A segment
  const srd_t srdA = A_IC ? srd_krows(p.A, p.lda, kBeg, kEnd, p.I) : srd_rows(p.A, p.lda, i0, p.I, Ka1);
  const srd_t srdA2 = srd_rows(p.A2, p.lda2, i0, p.I, p.K - Ka1);
  const srd_t srdB = B_IC ? srd_krows(p.B, p.ldb, kBeg, kEnd, p.J) : srd_rows(p.B, p.ldb, j0, p.J, p.K);
  Loader<BI, A_IC> la1, la2;
  Loader<BJ, B_IC> lb1;
  la1.init(t, i0, p.I, p.lda);
  la2.init(t, i0, p.I, p.lda2);
  lb1.init(t, j0, p.J, p.ldb);
  const unsigned stepA = A_IC ? (unsigned)p.lda * 4u : 4u;  // bytes per unit of k
  const unsigned stepB = B_IC ? (unsigned)p.ldb * 4u : 4u;

  float4 ra[TA::NV], rb[TB::NV];
  // k-tiles never straddle the two A segments (Ka % BK == 0, checked on the host)
  auto fetch = [&](int k0) {
    if (k0 < p.Ka) la1.load(ra, srdA, (unsigned)(A_IC ? k0 - kBeg : k0) * stepA, min(kEnd, p.Ka) - k0);
    else la2.load(ra, srdA2, (unsigned)(k0 - p.Ka) * 4u, kEnd - k0);
    lb1.load(rb, srdB, (unsigned)(B_IC ? k0 - kBeg : k0) * stepB, kEnd - k0);
  };
  // Software pipeline (one register set, two LDS stages): at the start of iteration kt the registers hold
  // tile kt+1 (loaded during iteration kt-1, so its vmcnt wait is free); they are written to the LDS stage that
  // every wave finished reading before the previous barrier, and the loads of tile kt+2 are issued right away.
  // The LDS stores then complete underneath this iteration's MFMAs instead of in front of the barrier.
  if (nkt > 0) {
    fetch(kBeg);
    r2s<BI, A_IC>(ra, lds, t);
    r2s<BJ, B_IC>(rb, lds + TA::FLOATS, t);
    if (nkt > 1) fetch(kBeg + BK);
  }
  __syncthreads();
#ifdef SNERF_ABL_CLOCK
  const unsigned long long clk_t1 = __builtin_amdgcn_s_memtime();
#endif

  for (int kt = 0; kt < nkt; ++kt) {
    const float* sa = lds + (kt & 1) * STAGE;
    const float* sb = sa + TA::FLOATS;
#ifndef SNERF_ABL_NOGLOAD
    if (kt + 1 < nkt) {
      float* da = lds + ((kt + 1) & 1) * STAGE;
      r2s<BI, A_IC>(ra, da, t);
      r2s<BJ, B_IC>(rb, da + TA::FLOATS, t);
      if (kt + 2 < nkt) fetch(kBeg + (kt + 2) * BK);
    }
#endif
    __builtin_amdgcn_s_setprio(0);
    const float* la = sa + (lane >> 5) * TA::PITCH + wi0 + (lane & 31);
    const float* lb = sb + (lane >> 5) * TB::PITCH + wj0 + (lane & 31);
    if (wave_live) {
      // operand fragments of k-pair kp+1 are fetched from LDS while the MFMAs of k-pair kp issue
      float a[2][MI], b[2][NJ];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) a[0][mi] = la[32 * mi];
#pragma unroll
      for (int nj = 0; nj < NJ; ++nj) b[0][nj] = lb[32 * nj];
#pragma unroll
      for (int kp = 0; kp < BK / 2; ++kp) {
        const int cur = kp & 1, nxt = cur ^ 1;
        if (kp + 1 < BK / 2) {
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) a[nxt][mi] = la[(2 * kp + 2) * TA::PITCH + 32 * mi];
#pragma unroll
          for (int nj = 0; nj < NJ; ++nj) b[nxt][nj] = lb[(2 * kp + 2) * TB::PITCH + 32 * nj];
        }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int nj = 0; nj < NJ; ++nj)
            acc[mi][nj] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][mi], b[cur][nj], acc[mi][nj], 0, 0, 0);
      }
    }
    __builtin_amdgcn_s_setprio(1);
    __syncthreads();
  }
  __builtin_amdgcn_s_setprio(2);

#ifdef SNERF_ABL_CLOCK
  if (t == 0 && p.colsum != nullptr && p.aux_mode == AUX_NONE) {
    unsigned long long* dbg = reinterpret_cast<unsigned long long*>(p.colsum) + 8 * (size_t)(blockIdx.x + gridDim.x * blockIdx.z);
    dbg[0] = __builtin_amdgcn_s_memtime() - clk_t0;
    dbg[1] = __builtin_amdgcn_s_memrealtime() - clk_r0;
    dbg[2] = clk_t1 - clk_t0;
    dbg[4] = clk_t0;
  }
  const unsigned long long clk_t2 = __builtin_amdgcn_s_memtime();
#endif
  // ---- epilogue ----------------------------------------------------------------------------------
  if (!wave_live) return;
#ifdef SNERF_ABL_NOEPI
  {  // keep the accumulators live, store (almost) nothing
    float sum = 0.f;
    for (int mi = 0; mi < MI; ++mi) for (int nj = 0; nj < NJ; ++nj) for (int r = 0; r < 16; ++r) sum += acc[mi][nj][r];
    if (sum == 12345.678f) C[0] = sum;
    return;
  }
#endif
  gemm_epilogue_dispatch<MI, NJ, WJ>(acc, lds, wave, lane, i0 + wi0, j0 + wj0, p, C);
#ifdef SNERF_ABL_CLOCK
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (t == 0 && p.colsum != nullptr && p.aux_mode == AUX_NONE) {
    unsigned long long* dbg = reinterpret_cast<unsigned long long*>(p.colsum) + 8 * (size_t)(blockIdx.x + gridDim.x * blockIdx.z);
    dbg[3] = __builtin_amdgcn_s_memtime() - clk_t2;
    dbg[5] = clk_r0;
    dbg[6] = __builtin_amdgcn_s_memrealtime();
    dbg[7] = __builtin_amdgcn_s_getreg(/*HW_REG_XCC_ID*/ (20) | (0 << 6) | ((4 - 1) << 11)) + 1;
  }
#endif
}

int prof_hook_begin(double flops, int variant, hipStream_t st);   // profile.hip
void prof_hook_end(int token, hipStream_t st);

template <int BI, int BJ, int WI, int WJ, bool A_IC, bool B_IC>
static int launch_cfg(const GemmArgs& g, hipStream_t stream) {
  const int tok = prof_hook_begin(2.0 * (double)g.I * (double)g.J * (double)g.K, (BI == 128 && BJ == 128) ? (A_IC ? 2 : (B_IC ? 1 : 0)) : 3, stream);
  KArgs p;
  p.A = g.A; p.A2 = g.A2 ? g.A2 : g.A; p.B = g.B; p.C = g.C; p.C2 = g.C2;
  p.bias = g.bias; p.aux = g.aux; p.colsum = g.colsum;
  p.C2s = g.C2s; p.auxs = g.aux_sign; p.sign_col0 = g.aux_sign ? g.sign_col0 : 0;
  p.sign_groups = ((g.aux_sign ? g.ldaux : g.ldc) + 63) / 64;
  static const unsigned epi_mask = getenv("SNERF_FAST_EPI") ? (unsigned)atoi(getenv("SNERF_FAST_EPI")) : 31u;  // all kinds; env for A/B of single kinds (bit 0 plain, 1 sin, 2 relu, 3 sinrec, 4 relu mask)
  p.epi_mask = epi_mask;
  p.lda = g.lda; p.lda2 = g.A2 ? g.lda2 : g.lda; p.Ka = g.A2 ? g.Ka : 0x7fffffff;
  p.ldb = g.ldb; p.I = g.I; p.J = g.J; p.K = g.K; p.ldc = g.ldc; p.ldaux = g.ldaux; p.ldcs = g.ldcs;
  // operand extents for the buffer descriptors (bytes; checked < 4 GiB by launch_gemm)
  const int K1 = g.A2 ? g.Ka : g.K;
  p.bytesA = (unsigned)(g.a_ic ? ((size_t)(K1 - 1) * g.lda + g.I) * 4 : ((size_t)(g.I - 1) * g.lda + K1) * 4);
  p.bytesA2 = g.A2 ? (unsigned)(((size_t)(g.I - 1) * g.lda2 + (g.K - g.Ka)) * 4) : p.bytesA;
  p.bytesB = (unsigned)(g.b_ic ? ((size_t)(g.K - 1) * g.ldb + g.J) * 4 : ((size_t)(g.J - 1) * g.ldb + g.K) * 4);
  p.Bpl = g.Bpl; p.pl_stride_bytes = (unsigned)(g.pl_stride * 2); p.bytesBpl = g.Bpl ? (unsigned)((2 * g.pl_stride + g.bt_elems) * 2) : 0;
  p.bt_rows = g.bt_rows; p.bt_row0 = g.bt_row0; p.bt_k0 = g.bt_k0;
  p.act = g.act; p.aux_mode = g.aux ? g.aux_mode : AUX_NONE; p.w0 = g.w0;
  p.k_split = g.k_split; p.slab_stride = g.slab_stride;
  p.tiles_i = (g.I + BI - 1) / BI;
  p.tiles_j = (g.J + BJ - 1) / BJ;
  dim3 grid(p.tiles_i * p.tiles_j, 1, g.k_split > 0 ? g.n_split : 1);
  if (g.x6 && BI == 128 && BJ == 128 && A_IC == B_IC) {
    int tile = g.tile;
    if (tile != 128 && tile != 256) {
      static const int forced = getenv("SNERF_X6_TILE") ? atoi(getenv("SNERF_X6_TILE")) : 0;  // diagnostics: 128 | 256
      // measured in the training step: the 256 x 256 tile wins for the row-contiguous (dW) operands (0.53 vs 0.63 ms),
      // the 128 x 128 tile with 3 workgroups per CU for the K-contiguous forward / dX launches (0.69 vs 0.73 ms)
      tile = (forced == 128 || forced == 256) ? forced : (A_IC ? x6_tile(g.I, g.J) : 128);
    }
    if (tile == 256) {
      p.tiles_i = (g.I + 255) / 256; p.tiles_j = (g.J + 255) / 256;
      grid.x = p.tiles_i * p.tiles_j;
    }
    launch_x6(A_IC, g.Bpl != nullptr && !A_IC, g.planes, tile, p, grid, stream);
  }
  else hipLaunchKernelGGL((gemm_kernel<BI, BJ, WI, WJ, A_IC, B_IC>), grid, dim3(NT), 0, stream, p);
  SNERF_LAUNCH_CHECK();
  prof_hook_end(tok, stream);
  return SNERF_OK;
}

int launch_gemm(const GemmArgs& g, hipStream_t stream) {
  // host-side shape checks: a kernel that faults can reset the whole node, so refuse anything
  // the loaders' float4 accesses do not cover.
  auto bad = [&](const char* why) {
    set_error("launch_gemm: %s (I=%d J=%d K=%d lda=%d ldb=%d ldc=%d a_ic=%d b_ic=%d)", why, g.I, g.J, g.K,
              g.lda, g.ldb, g.ldc, (int)g.a_ic, (int)g.b_ic);
    return SNERF_ERR_BAD_DESC;
  };
  if (!g.A || !g.B || !g.C) return bad("null operand");
  if (g.I <= 0 || g.J <= 0 || g.K <= 0) return bad("empty problem");
  if ((g.lda & 3) || (g.ldb & 3) || (g.ldc & 3) || (g.A2 && (g.lda2 & 3))) return bad("leading dimensions must be multiples of 4");
  if (((uintptr_t)g.A & 15) || ((uintptr_t)g.B & 15) || ((uintptr_t)g.C & 15) || (g.A2 && ((uintptr_t)g.A2 & 15)))
    return bad("operands must be 16-byte aligned");
  if (g.J & 3) return bad("J must be a multiple of 4 (float4 epilogue)");
  if (g.C2 && ((uintptr_t)g.C2 & 15)) return bad("C2 must be 16-byte aligned");
  if (g.bias && ((uintptr_t)g.bias & 15)) return bad("bias must be 16-byte aligned");
  if (g.aux && (((uintptr_t)g.aux & 15) || (g.ldaux & 3))) return bad("aux must be 16-byte aligned with ldaux % 4 == 0");
  if (g.colsum && (((uintptr_t)g.colsum & 15) || (g.ldcs & 3))) return bad("colsum must be 16-byte aligned with ldcs % 4 == 0");
  if ((size_t)128 * g.lda >= 0x3FFFFFFFull || (size_t)128 * g.ldb >= 0x3FFFFFFFull) return bad("leading dimension too large");
  if (g.k_split > 0 && ((size_t)g.k_split * g.lda >= 0x3FFFFFFFull || (size_t)g.k_split * g.ldb >= 0x3FFFFFFFull)) return bad("k_split * ld exceeds the 32-bit tile span");
  if (g.a_ic && g.k_split == 0 && (size_t)g.K * g.lda >= 0x3FFFFFFFull) return bad("row-contiguous operand without split-K exceeds the 32-bit span");
  if (g.a_ic) { if (g.I & 3) return bad("IC A needs I % 4 == 0"); if (g.A2) return bad("two-segment A is KC only"); }
  else { if (g.K & 3) return bad("KC A needs K % 4 == 0"); if (g.A2 && (g.Ka % BK)) return bad("Ka must be a multiple of 16"); }
  if (g.b_ic) { if (g.J & 3) return bad("IC B needs J % 4 == 0"); }
  else { if (g.K & 3) return bad("KC B needs K % 4 == 0"); }
  if (g.k_split > 0 && (g.k_split % 32)) return bad("k_split must be a multiple of 32");
  if (g.Bpl && ((g.bt_k0 & 15) || ((uintptr_t)g.Bpl & 15) || (g.pl_stride & 7) || g.bt_rows <= 0 || g.pl_stride * 6 >= 0xFFFFFFF0ull))
    return bad("pre-split B planes need bt_k0 % 16 == 0 and 16-byte alignment");
  if (g.aux && g.aux_mode != AUX_NONE && g.ldaux <= 0) return bad("aux needs ldaux");
  if (g.C2s && (g.narrow_j || g.act != ACT_SIN || g.aux_sign)) return bad("C2s (sign words) needs the 64-wide epilogue, ACT_SIN and no AUX_SINREC");
  if (g.aux_mode == AUX_SINREC && (!g.aux || !g.aux_sign || g.narrow_j || (g.sign_col0 & 3))) return bad("AUX_SINREC needs aux, aux_sign, sign_col0 % 4 == 0 and the 64-wide epilogue");

  if (!g.a_ic && !g.b_ic) {
    if (g.narrow_j) return launch_cfg<128, 32, 32, 32, false, false>(g, stream);
    return launch_cfg<128, 128, 64, 64, false, false>(g, stream);
  }
  if (!g.a_ic && g.b_ic) {
    if (g.narrow_j) return launch_cfg<128, 32, 32, 32, false, true>(g, stream);
    return launch_cfg<128, 128, 64, 64, false, true>(g, stream);
  }
  if (g.a_ic && g.b_ic) {
    if (g.narrow_i) return launch_cfg<32, 128, 32, 32, true, true>(g, stream);
    return launch_cfg<128, 128, 64, 64, true, true>(g, stream);
  }
  return bad("unsupported operand layout combination (IC A with KC B)");
}

}  // namespace snerf
