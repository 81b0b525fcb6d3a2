#pragma once
#include "common.h"
#include "plan.h"

namespace snerf {

struct CompArgs {
  int N = 0, S = 0, H = 0, C = 0;
  int sc = 0, sem_sigmoid = 0, has_sbeta = 0;
  const float* z = nullptr;                         // [N][S]
  const float* sigo = nullptr; const float* fino = nullptr; const float* suno = nullptr;  // [P][NARROW] pre-activations
  // SIREN passes at full width: the sigma / sun-visibility pre-activations arrive as per-wave partial dot products written by
  // the epilogues of the launches that produce their inputs (bsp_kc.hip: NDOT) -- value(p) = *bias + sum_q part[q * part_stride + p]
  // in this fixed order; null: read sigo / suno
  const float* sig_part = nullptr; const float* sig_bias = nullptr; int n_sig_part = 0;
  const float* sun_part = nullptr; const float* sun_bias = nullptr; int n_sun_part = 0;
  size_t part_stride = 0;
  // the final head layers likewise (NDOT = 5): column c of `fino` = fin_bias[c] + sum_{w < 4} fin_part[((blk * 4 + w) * ND_FIN + o) * part_stride + p],
  // blk = the head's block (fin_blk: rgb, semantic, beta, beta_s), o = the column's index inside its head
  const float* fin_part = nullptr; const float* fin_bias = nullptr; int fin_blk[4] = {0, 0, 0, 0};
  const float* sun_d = nullptr; int sun_stride = 3;
  const float* sky = nullptr;                       // packed sky params
  float* o_rgb = nullptr; float* o_depth = nullptr; float* o_weights = nullptr; float* o_transparency = nullptr;
  float* o_albedo = nullptr; float* o_sun = nullptr; float* o_sky = nullptr; float* o_beta = nullptr;
  float* o_sigmas = nullptr; float* o_beta_s = nullptr; float* o_logits = nullptr; long long* o_label = nullptr;
  float* save_T = nullptr; float* save_rgbraw = nullptr;  // kept for the backward pass (train mode)
};

struct CompBwdArgs {
  CompArgs f;
  const float* T = nullptr; const float* rgbraw = nullptr;
  const float* g_rgb = nullptr; const float* g_depth = nullptr; const float* g_weights = nullptr;
  const float* g_transparency = nullptr; const float* g_albedo = nullptr; const float* g_sun = nullptr;
  const float* g_sky = nullptr; const float* g_beta = nullptr; const float* g_sigmas = nullptr;
  const float* g_beta_s = nullptr; const float* g_logits = nullptr;
  float* d_sigo = nullptr; float* d_fino = nullptr; float* d_suno = nullptr;  // [P][NARROW]
  float* sky_slab = nullptr;  // [waves][9H+4]
  unsigned* zero = nullptr; int zero_n = 0;   // words the first workgroup clears (the backward pass's tile counters)
};

int launch_composite_fwd(const CompArgs& a, hipStream_t st);
int composite_bwd_blocks(int N);
int launch_composite_bwd(const CompBwdArgs& b, hipStream_t st);

}  // namespace snerf
