// fp32 MFMA GEMM for gfx950 (v_mfma_f32_32x32x2_f32), LDS-tiled, register-prefetched, with the
// layer-specific work fused into the tile loader (two-segment A = skip concat) and the epilogue
// (bias + SIREN sincos / ReLU, activation-derivative multiply, bias-gradient column sums, split-K slabs).
//
// Replaces, per layer: torch.nn.Linear + Siren/ReLU of RSSemanticNeRF (semantic/models/rs_semantic.py:173-258,
// 260-340; baseline/models/commons.py:27-38) forward, and their autograd backward (dX, dW, db).
//
// Tiling (64-wide wavefronts): 256 threads = 4 waves; block tile BI x BJ, wave tile WI x WJ made of
// 32x32 MFMA tiles; BK = 32.  Both operands live in LDS as [k][row] (row contiguous) so that the
// MFMA operand fetch -- lane l wants (row = l & 31, k = l >> 5) -- is a conflict-free ds_read_b32
// for either global storage order:
//   KC source (k contiguous, e.g. activations X[M][K], weights W[N][K]): float4 along k from global
//     (8 lanes cover one 128-B line), transposed on the LDS write (pitch BI+1 -> conflict-free b32 writes);
//   IC source (row contiguous, e.g. W read as B[k=n][j=kin] for dX, dZ/X read along m for dW):
//     float4 along rows, ds_write_b128 as is (pitch BI+4).
// fp32 MFMA issues at 64 cycles per 32x32x2 per SIMD, so one ds_read_b32 per MFMA keeps LDS at
// ~1/8 of its rate; the kernel is MFMA-issue bound by construction (roofline).
#include "gemm.h"
#include "../../include/snerf_hip.h"

#include <vector>

namespace snerf {

constexpr int BK = 32;
constexpr int NT = 256;

template <int BI, bool IC>
struct Tile {
  static constexpr int NV = BI / 32;                  // float4 per thread per k-tile
  static constexpr int PITCH = IC ? BI + 4 : BI + 1;  // LDS row pitch in floats
  static constexpr int FLOATS = BK * PITCH;
};

template <int BI, bool IC>
__device__ __forceinline__ void g2r(float4 (&v)[BI / 32], const float* __restrict__ P, int ld,
                                    const float* __restrict__ P2, int ld2, int Ka, int i0, int I,
                                    int k0, int kEnd, int t) {
#pragma unroll
  for (int r = 0; r < BI / 32; ++r) {
    int i, k;
    if (IC) {
      constexpr int V = BI / 4;
      i = i0 + 4 * (t % V);
      k = k0 + t / V + (NT / V) * r;
    } else {
      i = i0 + (t >> 3) + 32 * r;
      k = k0 + 4 * (t & 7);
    }
    float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < I && k < kEnd) {
      const float* src;
      if (IC) {
        src = P + (size_t)k * ld + i;
      } else {
        src = (k < Ka) ? (P + (size_t)i * ld + k) : (P2 + (size_t)i * ld2 + (k - Ka));
      }
      x = *reinterpret_cast<const float4*>(src);
    }
    v[r] = x;
  }
}

template <int BI, bool IC>
__device__ __forceinline__ void r2s(const float4 (&v)[BI / 32], float* __restrict__ lds, int t) {
  constexpr int PITCH = Tile<BI, IC>::PITCH;
#pragma unroll
  for (int r = 0; r < BI / 32; ++r) {
    if (IC) {
      constexpr int V = BI / 4;
      const int il = 4 * (t % V);
      const int kl = t / V + (NT / V) * r;
      *reinterpret_cast<float4*>(&lds[kl * PITCH + il]) = v[r];
    } else {
      const int il = (t >> 3) + 32 * r;
      const int kl = 4 * (t & 7);
      lds[(kl + 0) * PITCH + il] = v[r].x;
      lds[(kl + 1) * PITCH + il] = v[r].y;
      lds[(kl + 2) * PITCH + il] = v[r].z;
      lds[(kl + 3) * PITCH + il] = v[r].w;
    }
  }
}

struct KArgs {
  const float* A; const float* A2; const float* B;
  float* C; float* C2;
  const float* bias; const float* aux; float* colsum;
  int lda, lda2, Ka, ldb, I, J, K, ldc, ldaux, ldcs;
  int act, aux_mode;
  float w0;
  int k_split;
  unsigned long long slab_stride;
  int tiles_i, tiles_j;
};

// Workgroup -> tile map: blocks b and b+8 share an XCD (round-robin dispatch), and the J-tiles of
// one I-tile re-read the same A rows, so give each XCD group runs of consecutive J-tiles of the
// same I-tile: those re-reads then hit that XCD's L2 instead of HBM. Speed only, never correctness.
__device__ __forceinline__ void tile_of_block(int b, int tiles_i, int tiles_j, int& ti, int& tj) {
  const int n = tiles_i * tiles_j;
  const int xcd = b & 7, q = b >> 3;
  const int per = n >> 3;  // tiles per XCD group (exact part)
  if (b < (per << 3)) {
    const int lin = xcd * per + q;  // contiguous chunk of the (ti-major) tile order per XCD group
    ti = lin / tiles_j;
    tj = lin - ti * tiles_j;
  } else {  // remainder tiles (n % 8): identity order
    ti = b / tiles_j;
    tj = b - ti * tiles_j;
  }
}

template <int BI, int BJ, int WI, int WJ, bool A_IC, bool B_IC>
__global__ __launch_bounds__(NT, 2) void gemm_kernel(const KArgs p) {
  using TA = Tile<BI, A_IC>;
  using TB = Tile<BJ, B_IC>;
  constexpr int MI = WI / 32, NJ = WJ / 32;
  constexpr int WAVES_J = BJ / WJ;
  static_assert((BI / WI) * (BJ / WJ) == 4, "4 waves per workgroup");
  constexpr int STAGE = TA::FLOATS + TB::FLOATS;
  __shared__ __attribute__((aligned(16))) float lds[2 * STAGE];

  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int wi0 = (wave / WAVES_J) * WI, wj0 = (wave % WAVES_J) * WJ;
  int ti, tj;
  tile_of_block(blockIdx.x, p.tiles_i, p.tiles_j, ti, tj);
  const int i0 = ti * BI, j0 = tj * BJ;

  int kBeg = 0, kEnd = p.K;
  float* C = p.C;
  if (p.k_split > 0) {
    kBeg = blockIdx.z * p.k_split;
    kEnd = min(p.K, kBeg + p.k_split);
    C += (size_t)blockIdx.z * p.slab_stride;
  }
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

  float4 ra[TA::NV], rb[TB::NV];
  if (nkt > 0) {
    g2r<BI, A_IC>(ra, p.A, p.lda, p.A2, p.lda2, p.Ka, i0, p.I, kBeg, kEnd, t);
    g2r<BJ, B_IC>(rb, p.B, p.ldb, p.B, p.ldb, 0x7fffffff, j0, p.J, kBeg, kEnd, t);
    r2s<BI, A_IC>(ra, lds, t);
    r2s<BJ, B_IC>(rb, lds + TA::FLOATS, t);
  }
  __syncthreads();

  for (int kt = 0; kt < nkt; ++kt) {
    const float* sa = lds + (kt & 1) * STAGE;
    const float* sb = sa + TA::FLOATS;
    const bool more = (kt + 1) < nkt;
    if (more) {  // global loads of the next k-tile fly under this tile's MFMAs
      const int k0 = kBeg + (kt + 1) * BK;
      g2r<BI, A_IC>(ra, p.A, p.lda, p.A2, p.lda2, p.Ka, i0, p.I, k0, kEnd, t);
      g2r<BJ, B_IC>(rb, p.B, p.ldb, p.B, p.ldb, 0x7fffffff, j0, p.J, k0, kEnd, t);
    }
    const float* la = sa + (lane >> 5) * TA::PITCH + wi0 + (lane & 31);
    const float* lb = sb + (lane >> 5) * TB::PITCH + wj0 + (lane & 31);
    if (wave_live) {
#pragma unroll
      for (int kp = 0; kp < BK / 2; ++kp) {
        float a[MI], b[NJ];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) a[mi] = la[(2 * kp) * TA::PITCH + 32 * mi];
#pragma unroll
        for (int nj = 0; nj < NJ; ++nj) b[nj] = lb[(2 * kp) * TB::PITCH + 32 * nj];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int nj = 0; nj < NJ; ++nj)
            acc[mi][nj] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi], b[nj], acc[mi][nj], 0, 0, 0);
      }
    }
    if (more) {
      float* da = lds + ((kt + 1) & 1) * STAGE;
      r2s<BI, A_IC>(ra, da, t);
      r2s<BJ, B_IC>(rb, da + TA::FLOATS, t);
    }
    __syncthreads();
  }

  // ---- epilogue. C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
  if (!wave_live) return;
  const int lc = lane & 31, lh = lane >> 5;
#pragma unroll
  for (int nj = 0; nj < NJ; ++nj) {
    const int col = j0 + wj0 + 32 * nj + lc;
    const bool col_ok = col < p.J;
    const float bj = (p.bias != nullptr && col_ok) ? p.bias[col] : 0.f;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int row0 = i0 + wi0 + 32 * mi + 4 * lh;
      float cs = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = row0 + (r & 3) + 8 * (r >> 2);
        const bool ok = col_ok && row < p.I;
        const size_t off = (size_t)row * p.ldc + col;
        float v = acc[mi][nj][r] + bj;
        if (p.act == ACT_SIN) {
          float sn, cn;
          sincos_acc(p.w0 * v, &sn, &cn);
          v = sn;
          if (p.C2 != nullptr && ok) p.C2[off] = p.w0 * cn;
        } else if (p.act == ACT_RELU) {
          v = fmaxf(v, 0.f);
        }
        if (p.aux_mode != AUX_NONE) {
          const float x = ok ? p.aux[(size_t)row * p.ldaux + col] : 0.f;
          v = (p.aux_mode == AUX_MUL) ? v * x : (x > 0.f ? v : 0.f);
        }
        if (ok) C[off] = v;
        cs += ok ? v : 0.f;
      }
      if (p.colsum != nullptr) {
        cs += __shfl_xor(cs, 32, 64);
        // one partial per 32-row block: this MFMA tile covers rows [i0 + wi0 + 32 mi, +32)
        const int rb32 = (i0 + wi0 + 32 * mi) >> 5;
        if (lh == 0 && col_ok && (i0 + wi0 + 32 * mi) < p.I) p.colsum[(size_t)rb32 * p.ldcs + col] = cs;
      }
    }
  }
}

// ---- optional per-launch timing (snerf_profile_begin/_end): HIP events on the launch stream ----------
struct ProfRec { hipEvent_t a, b; double flops; int variant; };
static bool g_prof_on = false;
static std::vector<ProfRec> g_prof;
static std::vector<hipEvent_t> g_prof_pool;
static size_t g_prof_used = 0;

static hipEvent_t prof_event() {
  if (g_prof_used == g_prof_pool.size()) {
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    g_prof_pool.push_back(e);
  }
  return g_prof_pool[g_prof_used++];
}

template <int BI, int BJ, int WI, int WJ, bool A_IC, bool B_IC>
static int launch_cfg(const GemmArgs& g, hipStream_t stream) {
  ProfRec rec{nullptr, nullptr, 0.0, 3};
  if (g_prof_on) {
    rec.a = prof_event(); rec.b = prof_event();
    rec.flops = 2.0 * (double)g.I * (double)g.J * (double)g.K;
    if (BI == 128 && BJ == 128) rec.variant = A_IC ? 2 : (B_IC ? 1 : 0);
    if (rec.a && rec.b) (void)hipEventRecord(rec.a, stream);
  }
  KArgs p;
  p.A = g.A; p.A2 = g.A2 ? g.A2 : g.A; p.B = g.B; p.C = g.C; p.C2 = g.C2;
  p.bias = g.bias; p.aux = g.aux; p.colsum = g.colsum;
  p.lda = g.lda; p.lda2 = g.A2 ? g.lda2 : g.lda; p.Ka = g.A2 ? g.Ka : 0x7fffffff;
  p.ldb = g.ldb; p.I = g.I; p.J = g.J; p.K = g.K; p.ldc = g.ldc; p.ldaux = g.ldaux; p.ldcs = g.ldcs;
  p.act = g.act; p.aux_mode = g.aux ? g.aux_mode : AUX_NONE; p.w0 = g.w0;
  p.k_split = g.k_split; p.slab_stride = g.slab_stride;
  p.tiles_i = (g.I + BI - 1) / BI;
  p.tiles_j = (g.J + BJ - 1) / BJ;
  dim3 grid(p.tiles_i * p.tiles_j, 1, g.k_split > 0 ? g.n_split : 1);
  hipLaunchKernelGGL((gemm_kernel<BI, BJ, WI, WJ, A_IC, B_IC>), grid, dim3(NT), 0, stream, p);
  SNERF_LAUNCH_CHECK();
  if (g_prof_on && rec.a && rec.b) {
    (void)hipEventRecord(rec.b, stream);
    g_prof.push_back(rec);
  }
  return SNERF_OK;
}

int profile_begin() {
  g_prof.clear();
  g_prof_used = 0;
  g_prof_on = true;
  return SNERF_OK;
}

int profile_end(SnerfProfile* out) {
  g_prof_on = false;
  if (!out) { set_error("snerf_profile_end: null output"); return SNERF_ERR_NULL; }
  for (int v = 0; v < SNERF_PROFILE_VARIANTS; ++v) { out->ms[v] = 0.0; out->flops[v] = 0.0; out->launches[v] = 0; }
  for (const ProfRec& r : g_prof) {
    SNERF_HIP_CHECK(hipEventSynchronize(r.b));
    float ms = 0.f;
    SNERF_HIP_CHECK(hipEventElapsedTime(&ms, r.a, r.b));
    out->ms[r.variant] += ms;
    out->flops[r.variant] += r.flops;
    out->launches[r.variant] += 1;
  }
  g_prof.clear();
  g_prof_used = 0;
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
  if ((g.lda & 3) || (g.ldb & 3) || (g.A2 && (g.lda2 & 3))) return bad("leading dimensions must be multiples of 4");
  if (((uintptr_t)g.A & 15) || ((uintptr_t)g.B & 15) || (g.A2 && ((uintptr_t)g.A2 & 15))) return bad("operands must be 16-byte aligned");
  if (g.a_ic) { if (g.I & 3) return bad("IC A needs I % 4 == 0"); if (g.A2) return bad("two-segment A is KC only"); }
  else { if (g.K & 3) return bad("KC A needs K % 4 == 0"); if (g.A2 && (g.Ka & 3)) return bad("Ka % 4"); }
  if (g.b_ic) { if (g.J & 3) return bad("IC B needs J % 4 == 0"); }
  else { if (g.K & 3) return bad("KC B needs K % 4 == 0"); }
  if (g.k_split > 0 && (g.k_split % BK)) return bad("k_split must be a multiple of 32");
  if (g.aux && g.aux_mode != AUX_NONE && g.ldaux <= 0) return bad("aux needs ldaux");

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
