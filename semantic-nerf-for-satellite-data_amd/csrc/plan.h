// Host-side layout plan: packed parameter offsets and workspace regions derived from a SnerfDesc.
// Everything the kernels index is computed here once per call, so host code can check every
// operand shape before a launch (a faulting kernel can reset the node).
#pragma once
#include "common.h"
#include "../../include/snerf_hip.h"

namespace snerf {

constexpr int NARROW = 32;   // padded width of the 1..(5+C)-wide head outputs
constexpr int MAX_CLASSES = 16;
constexpr int ND_FIN = 5;       // projections per column tile of a folded final-layer launch (bsp_kc.hip: NDOT = 5)
constexpr int MAX_SKY_UNITS = 8;  // feat_last <= 512 (units per lane in the composite kernels)

struct Plan {
  // dims
  int N = 0, S = 0, P = 0, Pp = 0;
  int W = 0, H = 0, L = 0, E = 0, Ep = 0, F = 0, tau = 0, C = 0;
  bool siren = false, train = false, sc = false, sem_sigmoid = false;
  unsigned skip_mask = 0;
  // extras columns appended to the feats buffer: [sun(3) | t(tau) | t_s(tau)] padded to 4
  int x_sun = 0, x_t = 3, x_ts = -1, Xp = 0, FA = 0;
  // first head layers fused into one GEMM: blocks of H rows; sun block last
  int nblk = 0, blk_rgb = 0, blk_sem = -1, blk_beta = -1, blk_sbeta = -1, blk_sun = -1;
  int N1 = 0;   // rows of the fused first-head-layer matrix = KF + H
  int KF = 0;   // contraction length of the block-diagonal final-layer matrix: all blocks but sun, (nblk - 1) H, rounded up to
                // 128 so that the sun block starts on an exponent block
  int sun_col = 0;  // first row of the sun block in the fused first-layer matrix / its column in the h1 buffer (= KF)
  int Wf = 0;   // column of the extras block [sun | t | t_s] behind the feats columns: W rounded up to 128
  bool rgb_t = false, sem_t = false, sem_ts = false, sbeta_ts = false;
  // final-layer output columns inside the NARROW-wide buffer
  static constexpr int col_rgb = 0, col_beta = 3, col_sbeta = 4, col_sem = 5;

  // packed parameter layout (float offsets). weights: row-major [rows][ld]
  size_t w_tr[SNERF_MAX_LAYERS] = {0}, b_tr[SNERF_MAX_LAYERS] = {0};
  int k_tr[SNERF_MAX_LAYERS] = {0};
  size_t w_fs = 0, b_fs = 0;        // [W + NARROW][W]: feats rows then sigma row
  size_t w_h1 = 0, b_h1 = 0;        // [N1][FA]
  size_t w_s2 = 0, b_s2 = 0, w_s3 = 0, b_s3 = 0;  // [H][H]
  size_t w_s4 = 0, b_s4 = 0;        // [NARROW][H], row 0 used
  size_t w_fin = 0, b_fin = 0;      // [NARROW][KF]
  size_t sky = 0;                   // [H][4] w0 | [H] b0 | [4][H] w2 | [4] b2
  int sky_floats = 0;
  size_t n_fp32 = 0;         // floats of the fp32 region (the parameters; gradients use the same layout); the weight packs follow it
  size_t packed_floats = 0;  // whole packed buffer in floats: fp32 region + WF16 packs + exponents
  int pl = 2;        // fp16 planes of every activation tensor and weight pack (csrc/bsp.h): 2 = the default arithmetic (SNERF_FLAG_F16X2),
                     // 1 = SNERF_FLAG_F16X1 (reduced precision)

  // workspace layout (byte offsets)
  size_t o_z = 0, o_T = 0, o_rgbraw = 0, o_pe = 0, o_fa = 0, o_h1 = 0, o_c1 = 0;
  size_t o_h[SNERF_MAX_LAYERS] = {0}, o_c[SNERF_MAX_LAYERS] = {0};
  size_t o_s2 = 0, o_s3 = 0, o_cs2 = 0, o_cs3 = 0;
  size_t o_sigo = 0, o_fino = 0, o_suno = 0;
  // sigma / sun-visibility pre-activations as partial dot products of the producing SIREN launches' epilogues (bsp_kc.hip: NDOT):
  // [4 * (W / 256)][Pp] and [4 * (H / 256)][Pp] floats; folded when the producing layer is a SIREN layer of whole 256-column tiles
  size_t o_sigpart = 0, o_sunpart = 0, o_finpart = 0;
  bool nd_sig = false, nd_sun = false;
  // the whole SIREN trunk as ONE persistent launch with the activation tile resident in LDS (bsp_trunk.hip): one plane, W = 512, gamma
  // of 64 columns; training passes leave every layer's planes / sign words for the backward pass (round 5)
  bool fuse_trunk = false;
  bool nd_fin = false;   // the final layers of the rgb / semantic / beta heads ride in the fused first head layer's epilogue (H = 256: one column tile per head)
  // backward scratch
  size_t o_dza = 0, o_dzb = 0, o_dsa = 0, o_dsb = 0, o_dsig = 0, o_dfin = 0, o_dsun = 0;
  size_t o_skyslab = 0;
  size_t o_kcq = 0;                 // tile counters of the K-contiguous launches: KCQ_SLOTS slots of 64 bytes, zeroed at the start of a pass
  size_t o_rq = 0, rq_floats = 0;   // reduction arena of the block-scaled plane backward (bsp_pass.hip: per-launch slabs / column-sum partials)
  // ---- block-scaled plane layout (csrc/bsp.h): every activation buffer above holds G16 planes and has an exponent table; the
  //      32-wide head gradients also exist as planes
  size_t e_pe = 0, e_fa = 0, e_h1 = 0, e_s2 = 0, e_s3 = 0, e_h[SNERF_MAX_LAYERS] = {0};
  size_t e_dza = 0, e_dzb = 0, e_dsa = 0, e_dsb = 0, e_dsig = 0, e_dfin = 0, e_dsun = 0;
  size_t o_pdsig = 0, o_pdfin = 0, o_pdsun = 0;          // planes [Pp][32] of the narrow gradients
  // weight operand packs (WF16) behind the fp32 region of the packed buffer
  enum { WJ_MAX = 48 };
  int n_wjobs = 0;
  int wj_tr[SNERF_MAX_LAYERS] = {0}, wj_tt[SNERF_MAX_LAYERS] = {0};
  int wj_fs = 0, wj_sig = 0, wj_tfs = 0, wj_h1 = 0, wj_th1 = 0, wj_s2 = 0, wj_ts2 = 0, wj_s3 = 0, wj_ts3 = 0, wj_s4 = 0, wj_ts4 = 0,
      wj_fin = 0, wj_tfin = 0;
  unsigned long long wj_off[WJ_MAX] = {0};   // byte offset of each pack inside the plane region
  int wj_rows[WJ_MAX] = {0}, wj_K[WJ_MAX] = {0}, wj_e[WJ_MAX] = {0};
  size_t wp_bytes = 0;                       // plane region size; then WJ_MAX exponents (int) and WJ_MAX |max| words
  int h1w = 0;       // width of the h1 buffer in this pass (N1, or H for the sc pass)
  int maxw = 0;      // widest dz buffer
  int nrb = 0;       // 32-row blocks (colsum partials)
  int comp_blocks = 0;
  size_t ws_bytes = 0;
};

constexpr int KCQ_SLOTS = 64;
struct DwSplit { int ns = 1; int k_split = 32; };
DwSplit dw_choose_bsp(int P, int rows, int cols, bool narrow_rows);

// returns SNERF_OK or an error (message via set_error)
int make_plan(const SnerfDesc* d, Plan* pl);

size_t bsp_rq_floats(const Plan& p);   // bsp_pass.hip

}  // namespace snerf
