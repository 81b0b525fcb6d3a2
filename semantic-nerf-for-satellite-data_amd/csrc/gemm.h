// Tiled fp32-MFMA GEMM with fused prologue/epilogue: the one contraction kernel behind every layer
// of the MLP (forward, dX and dW).  See gemm.hip for the kernel; DESIGN.md "Kernels" for the roofline.
#pragma once
#include "common.h"
#include "../../include/snerf_hip.h"

namespace snerf {

enum GemmAct { ACT_NONE = 0, ACT_SIN = 1, ACT_RELU = 2 };
enum GemmAux { AUX_NONE = 0, AUX_MUL = 1, AUX_RELU_MASK = 2, AUX_SINREC = 3 };

// C[i, j] = epilogue( sum_k A(i,k) * B(j,k) )
// Operand storage: "KC" (k-contiguous): element (i,k) at P[i*ld + k]; "IC" (i-contiguous): P[k*ld + i].
struct GemmArgs {
  // A operand. KC mode may be split in two k-segments ([0,Ka) from A, [Ka,K) from A2): skip-concat
  // [gamma, h] of the trunk and [d feats | d sigma] of the backward pass without materialising the cat.
  const float* A = nullptr;  int lda = 0;  int Ka = 0;
  const float* A2 = nullptr; int lda2 = 0;
  bool a_ic = false;
  const float* B = nullptr;  int ldb = 0;  bool b_ic = false;
  int I = 0, J = 0, K = 0;
  float* C = nullptr; int ldc = 0;
  // forward epilogue: + bias[j], activation; C2 (optional) receives w0*cos(w0*z) for the backward pass
  const float* bias = nullptr;
  int act = ACT_NONE;
  float w0 = 1.f;
  float* C2 = nullptr;
  // C2s (optional, instead of C2): ONE SIGN BIT of cos(w0*z) per element -- all the backward pass needs besides the
  // stored activation h = sin(w0*z): w0*cos = w0 * sign * sqrt(1 - h^2).  32x less derivative traffic than C2.
  // Layout: per 32-row block and 64-column group 64 words; word rrow*16 + (col/4)%16 holds, for rows rrow + 4*ps
  // (ps = 0..7) and the 4 columns of one epilogue lane, bit 4*ps + c.  sign_floats() gives the buffer size.
  unsigned* C2s = nullptr;
  // backward epilogue: multiply by aux (saved activation derivative) or mask by aux > 0 (ReLU);
  // colsum (optional): per-32-row partial column sums of the final values, [ceil(I/32)][ldcs] (bias grads)
  const float* aux = nullptr; int ldaux = 0; int aux_mode = AUX_NONE;
  // AUX_SINREC: aux = the stored activation h (same shape as C), aux_sign = the sign words written through C2s by the
  // forward launch, sign_col0 = column of aux inside its buffer (a multiple of 4); multiplies by w0*sign*sqrt(1-h^2)
  const unsigned* aux_sign = nullptr; int sign_col0 = 0;
  float* colsum = nullptr; int ldcs = 0;
  // split-K (dW): blockIdx.z = split s handles k in [s*k_split, (s+1)*k_split), writes C + s*slab_stride
  int k_split = 0; int n_split = 1; size_t slab_stride = 0;
  bool narrow_j = false;  // 128x32 tile (J <= 32-wide heads)
  bool narrow_i = false;  // 32x128 tile (dW of the narrow heads)
  // pre-split B (weights): bf16 planes hi|mid|lo of the weight MATRIX that B points into, stored k-tile-major
  // ([k/16][row][16], LDS swizzle baked in) so that one B tile is a contiguous 4 KB block per plane.
  // Bpl = plane 0 of the matrix, plane p at Bpl + p*pl_stride elements; the operand is rows [bt_row0, +J) and
  // k >= bt_k0 of a matrix with bt_rows rows.
  const unsigned short* Bpl = nullptr; size_t pl_stride = 0; int bt_rows = 0, bt_row0 = 0, bt_k0 = 0; size_t bt_elems = 0;
  int tile = 0;           // split kernel tile: 128, 256, or 0 = choose (256 when it wastes no more area than 128)
  int planes = 3;         // bf16 planes per operand of the split kernel: 3 fp32-class, 2 ~16-bit, 1 plain bf16
  bool x6 = false;        // split-bf16 MFMA (gemm_x6.hip) for the 128x128 tile when both operands share a layout
};

// floats of a sign-word buffer covering `rows` (a multiple of 32) x `ld` elements (GemmArgs::C2s)
inline size_t sign_floats(size_t rows, int ld) { return rows / 32 * (size_t)((ld + 63) / 64) * 64; }

// tile of the split kernel for an I x J problem: 256 x 256 when that covers no more padded area than 128 x 128
inline int x6_tile(int I, int J) {
  const long long a128 = (long long)((I + 127) / 128) * ((J + 127) / 128) * 128 * 128;
  const long long a256 = (long long)((I + 255) / 256) * ((J + 255) / 256) * 256 * 256;
  return a256 <= a128 ? 256 : 128;
}

int launch_gemm(const GemmArgs& g, hipStream_t stream);
int profile_begin();
int profile_end(struct ::SnerfProfile* out);

}  // namespace snerf
