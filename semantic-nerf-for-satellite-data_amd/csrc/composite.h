#pragma once
#include "common.h"
#include "plan.h"

namespace snerf {

struct CompArgs {
  int N = 0, S = 0, H = 0, C = 0;
  int sc = 0, sem_sigmoid = 0, has_sbeta = 0;
  const float* z = nullptr;                         // [N][S]
  const float* sigo = nullptr; const float* fino = nullptr; const float* suno = nullptr;  // [P][NARROW] pre-activations
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
};

int launch_composite_fwd(const CompArgs& a, hipStream_t st);
int composite_bwd_blocks(int N);
int launch_composite_bwd(const CompBwdArgs& b, hipStream_t st);

}  // namespace snerf
