// Block-scaled fp16-plane tensors ("BSP") -- the storage format of every activation / activation-gradient tensor and
// of the pre-packed weights in the default arithmetic (SNERF_FLAG_F16X2), and the argument structs of the kernels
// that produce and consume it (bsp_gemm.hip, bsp_aux.hip).
//
// A logical fp32 tensor X[R][K] (K % 16 == 0) is kept as
//   * two fp16 planes of x * 2^e: hi = fp16(x 2^e), lo = fp16(x 2^e - hi) -- 22 significant bits, the SAME 4 bytes per
//     element as fp32 -- interleaved per 16 elements ("G16"): row r is ld * 4 bytes, 16-column group g of it is 64 bytes
//     [hi(16) | lo(16)].  A 16-deep k-step of a K-contiguous GEMM therefore reads one contiguous 64-byte piece per row,
//     a 256-column slab of a point row is 1 KiB contiguous, and either can be copied to LDS by buffer_load ... lds
//     with no conversion work in the consumer's k-loop (round 1 split fp32 operands on the fly: ~30 VALU / 12 MFMA);
//   * one exponent e per (128-row, 128-column) block, chosen by the PRODUCING workgroup from the block's own |max| so
//     that max 2^e lies in [2^13, 2^14): no tensor-wide maximum, no atomics, no scale "one layer ahead", and a quiet
//     row block next to a loud one keeps its own 22 bits.  A consumer contracts hi hi + hi lo + lo hi on
//     v_mfma_f32_32x32x16_f16 (dropped lo lo term: 2^-22 relative) and, where the exponent changes along its
//     contraction axis, rescales its fp32 accumulators by the power of two (exact).
// ONE-PLANE form (SNERF_FLAG_F16X1, the reduced-precision mode of the reference's `precision = 16` runs): the same layout with the
// lo plane left out -- row r is ld * 2 bytes, a 16-column group 32 bytes, the weight units 1 KiB -- the same block exponents, ONE
// product per contraction step.  11 significant bits relative to the block's maximum (fp16 keeps them down to 2^-27 of it).  Every
// size / offset helper below takes the plane count `pl` (1 or 2); the kernels are templated on it.
// Weights are packed once per step in MFMA fragment order ("WF16": 2 KiB units of 32 rows x 16 k = [plane][lane][16 B],
// lane = 32 (k / 8) + slot, slot m holding row wf16_row(m) of the unit: one fragment is one contiguous KiB that a wave
// loads straight into registers) with one exponent per matrix.
#pragma once
#include "common.h"
#include "gemm.h"

namespace snerf {
namespace bsp {

constexpr int RB = 128;          // rows per exponent block
constexpr int CB = 128;          // columns per exponent block
constexpr int E_MAX = 100;       // |exponent| clamp (2^e must be an fp32 normal)
// Quiet side: a block whose |max| is below 2^-47 keeps exponent 60 and loses bits (it flushes to zero below ~2^-84).  Where the
// exponent changes along a contraction the GEMMs rescale their fp32 accumulators by 2^(difference); with the quiet side
// capped, a loud block (|max| up to ~1e8) next to a vanishing one moves them by at most 2^74 -- finite for K <= 2048.
constexpr int E_QUIET = 60;

typedef __amdgpu_buffer_rsrc_t srd_t;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
constexpr unsigned OOB = 0xFFFFFFF0u;   // voffset the buffer bounds check always rejects (loads return 0)

__host__ __device__ inline int ncb_of(int ld) { return (ld + CB - 1) / CB; }
__host__ __device__ inline size_t plane_bytes(size_t rows, int ld, int pl = 2) { return rows * (size_t)ld * 2 * pl; }
__host__ __device__ inline size_t etab_ints(size_t rows, int ld) { return ((rows + RB - 1) / RB) * (size_t)ncb_of(ld); }
// byte offset of column k's hi element inside a row (lo, two planes: + 32)
__host__ __device__ inline unsigned g16_off(int k, int pl = 2) { return (unsigned)(k >> 4) * 32u * (unsigned)pl + (unsigned)(k & 15) * 2u; }

// exponent for a block whose |max| has these float bits: max * 2^e in [2^13, 2^14); 0 for an empty / non-finite block
__host__ __device__ inline int exp_of_maxbits(unsigned b) {
  if (b == 0u || b >= 0x7f800000u) return 0;
  int e = 13 - ((int)(b >> 23) - 127);
  return e < -E_MAX ? -E_MAX : (e > E_QUIET ? E_QUIET : e);
}
__device__ __forceinline__ float pow2f(int e) { return __uint_as_float((unsigned)(127 + e) << 23); }   // |e| <= 126

__device__ __forceinline__ srd_t make_srd(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ unsigned clamp_bytes(unsigned long long b) { return (unsigned)(b < 0xFFFFFFF0ull ? b : 0xFFFFFFF0ull); }
__device__ __forceinline__ size_t uniform_sz(size_t v) {
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return ((size_t)hi << 32) | lo;
}

// 8 fp32 values (times 2^e given as `scale`) -> 8 hi + 8 lo fp16, packed four dwords each.  Two mixed-precision FMAs per
// element: hi = f16(x s) and lo = f16(x s - hi) (one rounding each; x s is exact).
__device__ __forceinline__ void split8(const float (&x)[8], float scale, u32x4& hi, u32x4& lo) {
  unsigned h[4], l[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h[i]) : "v"(x[2 * i]), "v"(scale));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h[i]) : "v"(x[2 * i + 1]), "v"(scale));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(l[i]) : "v"(x[2 * i]), "v"(scale), "v"(h[i]));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l[i]) : "v"(x[2 * i + 1]), "v"(scale), "v"(h[i]));
  }
  hi = u32x4{h[0], h[1], h[2], h[3]};
  lo = u32x4{l[0], l[1], l[2], l[3]};
}
// one plane: hi only
__device__ __forceinline__ void cvt8(const float (&x)[8], float scale, u32x4& hi) {
  unsigned h[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h[i]) : "v"(x[2 * i]), "v"(scale));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h[i]) : "v"(x[2 * i + 1]), "v"(scale));
  }
  hi = u32x4{h[0], h[1], h[2], h[3]};
}
__device__ __forceinline__ void join8_1(const u32x4 hi, float inv_scale, float (&x)[8]) {
  const f16x8 h = __builtin_bit_cast(f16x8, hi);
#pragma unroll
  for (int i = 0; i < 8; ++i) x[i] = (float)h[i] * inv_scale;
}
// the inverse: 8 hi + 8 lo fp16 -> fp32 (hi + lo) * inv_scale
__device__ __forceinline__ void join8(const u32x4 hi, const u32x4 lo, float inv_scale, float (&x)[8]) {
  const f16x8 h = __builtin_bit_cast(f16x8, hi), l = __builtin_bit_cast(f16x8, lo);
#pragma unroll
  for (int i = 0; i < 8; ++i) x[i] = fmaf((float)h[i], inv_scale, (float)l[i] * inv_scale);   // two v_fma_mix_f32; exact (power of two)
}

// hi + lo as they lie (the value times 2^e, no scale applied): ONE mixed-precision FMA per element, exact
__device__ __forceinline__ void sum8(const u32x4 hi, const u32x4 lo, float (&x)[8]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel_hi:[1,0,1]" : "=v"(x[2 * i]) : "v"(hi[i]), "v"(lo[i]));
    asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(x[2 * i + 1]) : "v"(hi[i]), "v"(lo[i]));
  }
}
// one plane as it lies: 8 fp16 -> fp32
__device__ __forceinline__ void cvt8f(const u32x4 hi, float (&x)[8]) {
  const f16x8 h = __builtin_bit_cast(f16x8, hi);
#pragma unroll
  for (int i = 0; i < 8; ++i) x[i] = (float)h[i];
}

// Row order inside a 32-row unit of a WF16 weight pack: MFMA row m of the unit holds matrix row (unit base) + wf16_row(m),
// bits 2 and 3 of m exchanged (an involution).  The K-contiguous kernel uses the weights as the matrix cores' A operand,
// so the accumulator of lane l holds, for ONE point (l & 31), the output columns m = (r & 3) + 8 (r >> 2) + 4 (l >> 5) of
// the unit; with this order those are columns 16 (r >> 3) + 8 (l >> 5) + (r & 7): two runs of eight consecutive
// columns = one 16-byte piece of each plane per 16-column group, stored (and, for the derivative epilogues, loaded)
// without any exchange between lanes.
__host__ __device__ inline int wf16_row(int m) { return (m & 0x13) | ((m & 4) << 1) | ((m & 8) >> 1); }

// ---- K-contiguous GEMM:  C[i][j] = epilogue( sum_k A(i,k) W(j,k) ) -------------------------------------------------
// A: one or two BSP segments along k ([0,Ka) from A, [Ka,K) from A2).  W: a WP16 pack.  C: BSP (planes + exponents,
// + sign words of cos for ACT_SIN) or, for the 32-wide head outputs, plain fp32 [I][32].
struct KcArgs {
  const char* A = nullptr; const int* EA = nullptr; int lda = 0; int a_col0 = 0;      // element (i,k) = column a_col0 + k of the tensor
  const char* A2 = nullptr; const int* EA2 = nullptr; int lda2 = 0; int a2_col0 = 0;
  int Ka = 0;                        // k-length of the first segment (= K when there is no second one); % 16 == 0
  const char* W = nullptr;           // WF16 pack of the weight matrix; unit (ks, rb32) at ((ks * w_rb32 + rb32) * 2048) bytes
  const int* EW = nullptr;           // the matrix's exponent (one int, device)
  int w_rb32 = 0, w_row0 = 0, w_k0 = 0; unsigned w_bytes = 0;   // rows / 32 of the pack; first row (% 32 == 0) and first k (% 16 == 0) of the operand
  int I = 0, J = 0, K = 0;           // J % 16 == 0 (BSP output) / J <= 32 (fp32 output); K % 16 == 0
  // output
  char* C = nullptr; int* EC = nullptr; int ldc = 0; int c_col0 = 0;                  // BSP: column c_col0 + j (c_col0 % 128 == 0)
  float* Cf = nullptr;               // fp32 [I][32] output of the narrow variant
  const float* bias = nullptr;       // [J] or null
  int act = ACT_NONE; float w0 = 1.f;
  unsigned* Csign = nullptr;         // ACT_SIN, training: sign words of cos(w0 z) of the output tensor (layout: sign_index())
  // backward epilogue: multiply by the activation derivative rebuilt from the stored activation h (BSP, same shape as C)
  int aux_mode = AUX_NONE;           // AUX_SINREC: w0 sign sqrt(1 - h^2); AUX_RELU_MASK: h > 0
  const char* H = nullptr; const int* EH = nullptr; int ldh = 0; int h_col0 = 0; const unsigned* Hsign = nullptr;
  float* colsum = nullptr; int ldcs = 0;   // partial column sums of the stored values, one row per 128-row tile (bias gradients)
  int tiles_i = 0, tiles_j = 0;
  int tj_skip = -1;                  // >= 0: column tile tj_skip is NOT computed (a head block nobody asked for: the beta block of the fused first head
                                     // layer in a frame that wants rgb / depth / labels only); tiles_j then counts the computed tiles, tiles_jr all of them
  int tiles_jr = 0;
  int rev = 0;                       // walk the tiles of every XCD group backwards (tiles.h)
  int dbg = 0;                       // diagnostic builds only (bsp_kc.hip: DIAG); ignored by the product kernels
  int* tile_ctr = nullptr;           // 8 zeroed ints (one 64-byte slot per launch): tiles beyond the first are drawn from them; null: fixed shares
  int pl = 2;                        // planes of EVERY plane tensor of the launch (A, A2, W pack, C, H): 2 (default arithmetic) or 1
  // ACT_SIN launches of whole 256-column tiles: besides the output, every wave writes for each of its points the dot product of
  // its 64 output values (fp32, before they are rounded to planes) with nd_w -- the 1-wide projection that follows the layer
  // (sigma after the trunk, sun visibility after its last hidden layer), as tiles_j * 4 partial sums per point:
  // nd_out[(tj * 4 + wave) * nd_stride + i].  The consumer adds them in that order (composite.h).
  // General form (nd_omax = 5: the final head layers): column tile tj takes nd_rows[tj] (<= nd_omax) projections, rows nd_row0[tj] .. of
  // the matrix nd_w (leading dimension nd_ldw), restricted to the tile's 256 columns; nd_out[((tj * 4 + wave) * nd_omax + o) * nd_stride + i].
  // The 1-wide form is nd_omax = 1, nd_rows = 1, nd_row0 = 0 for every tile (launch_kc fills that in when nd_omax == 0).
  const float* nd_w = nullptr; float* nd_out = nullptr; unsigned long long nd_stride = 0;
  int nd_omax = 0, nd_ldw = 0;
  int nd_rows[8] = {0, 0, 0, 0, 0, 0, 0, 0}, nd_row0[8] = {0, 0, 0, 0, 0, 0, 0, 0}, nd_woff[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // nd_woff: filled by launch_kc
};
// sign word of (32-row block, 64-column group, lane) of a tensor with `ld` columns: bit 8 ps + c <-> row (lane >> 3) + 8 ps,
// column 8 (lane & 7) + c of that block (the epilogue's own lane mapping, so producer and consumer touch one word per lane)
__host__ __device__ inline size_t sign_words(size_t rows, int ld) { return ((rows + 31) / 32) * (size_t)((ld + 63) / 64) * 64; }

// ---- weight-gradient GEMM:  C[i][j] (+ split-K slabs) = sum_p A(p,i) B(p,j) ------------------------------------------
struct DwArgs {
  const char* A = nullptr; const int* EA = nullptr; int lda = 0; int a_col0 = 0;   // dZ [P][.]: BSP
  const char* B = nullptr; const int* EB = nullptr; int ldb = 0; int b_col0 = 0;   // X  [P][.]: BSP
  int I = 0, J = 0, P = 0;
  float* C = nullptr; int ldc = 0;           // fp32 slabs [n_split][slab_stride], row i at i * ldc
  int k_split = 0, n_split = 1; unsigned long long slab_stride = 0;   // k_split % 128 == 0
  int tiles_i = 0, tiles_j = 0;
  int pl = 2;                        // planes of A and B
};

// ---- the SIREN trunk as one persistent launch (bsp_trunk.hip; one plane, W = 512, gamma of 64 columns) -----------------------------
constexpr int TR_MAXL = 8;
constexpr int TR_SLOTS = TR_MAXL + 1;   // + the feats layer as entry L of the per-layer arrays
struct TrunkArgs {
  const char* pe = nullptr; const int* Epe = nullptr;       // gamma(x): one-plane tensor [P][64] + its exponent per 128 points
  int P = 0, W = 0, L = 0; unsigned skip_mask = 0;
  int gamma_free_layer = 0;                                  // filled by launch_trunk: the last layer that reads gamma (>= 2)
  const char* Wp[TR_SLOTS] = {}; const int* EW[TR_SLOTS] = {}; unsigned w_bytes[TR_SLOTS] = {}; int K[TR_SLOTS] = {};   // WF16 packs (one plane), K: 64 | 512 | 576
  const float* bias[TR_SLOTS] = {}; float w0[TR_SLOTS] = {};
  // feats = W_f h + b_f (no activation) behind the last SIREN layer, as entry L of the arrays above: planes [P][ldf] at column 0 + one
  // exponent per (128 points, 128 columns); null: the trunk alone (the last layer's planes leave instead)
  char* F = nullptr; int* EF = nullptr; int ldf = 0;
  char* H[TR_MAXL] = {}; int* EH[TR_MAXL] = {}; unsigned* Hsign[TR_MAXL] = {};   // outputs [P][W] one plane: the last layer always, every layer when training
  const float* nd_w = nullptr; float* nd_out = nullptr; unsigned long long nd_stride = 0;   // sigma's projection: 8 partial sums per point
  int* tile_ctr = nullptr;                                   // 8 zeroed ints
  int dbg = 0;                                               // diagnostic builds only (bsp_trunk.hip: TRUNK_DIAG_BUILD); ignored by the product kernels
};
int launch_trunk(const TrunkArgs& a, bool train, hipStream_t st);
void trunk_set_grid_override(int n);                       // test hook: persistent grid of n workgroups (0: one per CU)
void trunk_set_fusion(int on);                             // test hook: 0 = the launch-per-layer path for every pass
bool trunk_fusion_enabled();

int launch_kc(const KcArgs& a, hipStream_t st);          // 128 x 256 tiles, BSP output
void kc_set_grid_override(int n);                          // test hook: persistent grid of n workgroups (0: two per CU)
int launch_kc_narrow(const KcArgs& a, hipStream_t st);   // 128 x 32 tiles, fp32 output (Cf), bias only
int launch_dw(const DwArgs& a, bool narrow_i, hipStream_t st);   // 256 x 256 tiles (narrow_i: 32 x 256)

// ---- producers / converters (bsp_aux.hip) ------------------------------------------------------------------------------
// fp32 [rows][ld_src] (cols valid) -> BSP planes + exponents (block maxima taken over the valid rows / columns)
int launch_to_planes(const float* src, int ld_src, int rows, int cols, char* dst, int* E, int ld, int col0, int pl, hipStream_t st);
int launch_from_planes(const char* src, const int* E, int ld, int col0, int rows, int cols, float* dst, int ld_dst, int pl, hipStream_t st);

struct WPackJob {        // one weight operand: fp32 master matrix (possibly read transposed) -> WF16 pack
  unsigned long long src_off;   // float offset of the matrix in the packed fp32 region
  int src_ld;                   // its leading dimension
  int rows, K;                  // rows x K of the OPERAND (K % 16 == 0); transposed: operand(r, k) = master(k, r)
  int transposed;
  unsigned long long dst_off;   // byte offset of the pack inside the plane region
  int e_idx;                    // exponent slot (shared by a matrix and its transpose)
  int m_rows, m_cols;           // extent of the master matrix (for the |max| pass)
};
constexpr int WPACK_MAX = 48;
struct WPackTable { WPackJob j[WPACK_MAX]; int n; };      // host-side table
constexpr int WPACK_CHUNK = 24;
struct WPackChunk { WPackJob j[WPACK_CHUNK]; int n; };   // what one launch carries as its argument
__host__ __device__ inline size_t wp16_bytes(int rows, int K, int pl = 2) { return (size_t)((rows + 31) / 32) * (size_t)(K / 16) * 1024 * pl; }
int launch_wpack(const WPackTable& tb, const float* master, char* planes, int* exps, unsigned* maxbits, int pl, hipStream_t st);

}  // namespace bsp
struct EncodeArgs;
namespace bsp {
// x = o + d z, gamma(x) (or raw x) as planes [P][Ep]; the [sun | t | t_s] block as columns [fa_col0, +16) of the [P][FA] tensor
int launch_encode_bsp(const EncodeArgs& a, char* pe, int* Epe, char* fa, int* Efa, int fa_col0, int pl, hipStream_t st);
int launch_zero_cols(char* base, size_t pitch, size_t width_bytes, int rows, hipStream_t st);   // width_bytes % 16 == 0
// [rows][32] fp32 -> 256-row partial column sums (+ planes [rows][32] and one exponent per 128 rows when planes != null)
int launch_colsum32_bsp(const float* in, int rows, float* partial, char* planes, int* E, int pl, hipStream_t st);

}  // namespace bsp
}  // namespace snerf
