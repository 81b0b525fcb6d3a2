// Fused loss kernels (one wavefront per ray, lanes over samples): phase 1 = per-ray sums and counts,
// phase 2 = loss_dict values + gradients w.r.t. the rendered tensors.  See include/snerf_hip.h.
#include "aux_kernels.h"
#include "plan.h"

namespace snerf {

enum { T_COLOR = 0, T_LOGB = 1, T_SC2 = 2, T_SC3 = 3, T_CE = 4, T_CECNT = 5, T_INVB = 6, T_LOGBS = 7,
       T_CAR = 8, T_CARCNT = 9, T_DS = 10, T_N = 11 };

struct RayLoss {  // per-ray quantities shared by both phases
  float wb, wbs, sc2, sc3s, d2, ce, lse, mx;
  bool valid_ce, car;
};

__device__ __forceinline__ RayLoss ray_loss(const SnerfLossCfg& c, const SnerfLossIn& in, int ray, int lane) {
  RayLoss r;
  const int S = c.n_samples;
  float wb = 0.f, wbs = 0.f, sc2 = 0.f, sc3 = 0.f;
  for (int j = lane; j < S; j += 64) {
    const size_t p = (size_t)ray * S + j;
    if (c.color_mode == 2 || c.sem_mode == 2 || c.car_reg) {
      const float w = in.weights[p];
      wb += w * in.beta[p];
      if (c.use_sbeta) wbs += w * in.beta_semantic[p];
    }
    if (c.has_sc) {
      const float v = in.sun_sc[p];
      const float d = in.transparency_sc[p] - v;
      sc2 += d * d;
      sc3 += in.weights_sc[p] * v;
    }
  }
  r.wb = wave_sum(wb); r.wbs = wave_sum(wbs); r.sc2 = wave_sum(sc2); r.sc3s = wave_sum(sc3);
  r.d2 = 0.f;
  if (c.color_mode) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float d = in.rgb[(size_t)ray * 3 + k] - in.gt_rgb[(size_t)ray * 3 + k];
      r.d2 += d * d;
    }
  }
  r.valid_ce = false; r.car = false; r.ce = 0.f; r.lse = 0.f; r.mx = 0.f;
  if (c.sem_mode || c.car_reg) {
    const bool m = in.mask ? (in.mask[ray] != 0) : true;
    const long long y = in.labels[ray];
    r.car = c.car_reg && m && (y == (long long)c.car_label);
    if (c.sem_mode) {
      r.valid_ce = m && (y != (long long)c.ignore_index);
      const float* l = in.semantic_logits + (size_t)ray * c.n_classes;
      float mx = -INFINITY;
      for (int k = 0; k < c.n_classes; ++k) mx = fmaxf(mx, l[k]);
      float se = 0.f;
      for (int k = 0; k < c.n_classes; ++k) se += expf(l[k] - mx);
      r.mx = mx;
      r.lse = logf(se);
      // a label outside [0, C) that is not the ignore index makes torch's cross_entropy raise (semantic/components/loss.py:52-54);
      // an asynchronous kernel cannot raise, so the term (and with it the total loss) becomes NaN -- never an out-of-range read
      if (r.valid_ce) r.ce = (y >= 0 && y < (long long)c.n_classes) ? (mx + r.lse) - l[y] : __builtin_nanf("");  // -log_softmax(l)[y]
    }
  }
  return r;
}

__global__ __launch_bounds__(256) void loss_partial_kernel(SnerfLossCfg c, SnerfLossIn in, float* __restrict__ partial) {
  const int lane = threadIdx.x & 63;
  const int wave_g = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int nwaves = gridDim.x * 4;
  float t[SNERF_LOSS_NTOT];
#pragma unroll
  for (int i = 0; i < SNERF_LOSS_NTOT; ++i) t[i] = 0.f;
  for (int ray = wave_g; ray < c.n_rays; ray += nwaves) {
    const RayLoss r = ray_loss(c, in, ray, lane);
    const float bbar = r.wb + 0.05f;
    if (c.color_mode == 1) t[T_COLOR] += r.d2;
    if (c.color_mode == 2) { t[T_COLOR] += r.d2 / (2.f * bbar * bbar); t[T_LOGB] += logf(bbar); }
    if (c.has_sc) { t[T_SC2] += r.sc2; t[T_SC3] += 1.f - r.sc3s; }
    if (c.sem_mode) { t[T_CE] += r.ce; t[T_CECNT] += r.valid_ce ? 1.f : 0.f; }
    if (c.sem_mode == 2) {
      const float bs = (c.use_sbeta ? r.wbs : r.wb) + 0.05f;
      t[T_INVB] += 1.f / (2.f * bs * bs);
      t[T_LOGBS] += logf(bs);
    }
    if (c.car_reg && r.car) { const float e = 1.f - r.wb; t[T_CAR] += e * e; t[T_CARCNT] += 1.f; }
    if (c.has_depth) {
      const float w = in.depth_weights ? in.depth_weights[ray] : 1.f;
      const float e = in.depth[ray] - in.gt_depth[ray];
      t[T_DS] += w * (e * e);
    }
    t[T_N] += 1.f;
  }
  if (lane < SNERF_LOSS_NTOT) {
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < SNERF_LOSS_NTOT; ++i) v = (lane == i) ? t[i] : v;
    partial[(size_t)wave_g * SNERF_LOSS_NTOT + lane] = v;
  }
}

__global__ __launch_bounds__(256) void loss_finish_kernel(SnerfLossCfg c, SnerfLossIn in, const float* __restrict__ tot,
                                                          float Ng_arg, float gs, float* __restrict__ terms, SnerfLossGrads g) {
  const float Ng = Ng_arg > 0.f ? Ng_arg : tot[T_N];   // 0: the ray count summed (and all-reduced) with the other totals
  const int lane = threadIdx.x & 63;
  const int wave_g = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int nwaves = gridDim.x * 4;
  const int S = c.n_samples, C = c.n_classes;
  const float ce_mean = tot[T_CE] / tot[T_CECNT];
  const float invb_mean = tot[T_INVB] / Ng;
  if (wave_g == 0 && lane == 0 && terms) {
    for (int i = 0; i < SNERF_LOSS_NTERMS; ++i) terms[i] = 0.f;
    if (c.color_mode == 1) terms[SNERF_TERM_COLOR] = tot[T_COLOR] / (3.f * Ng);
    if (c.color_mode == 2) { terms[SNERF_TERM_COLOR] = tot[T_COLOR] / (3.f * Ng); terms[SNERF_TERM_LOGBETA] = (3.f + tot[T_LOGB] / Ng) * 0.5f; }
    if (c.has_sc) { terms[SNERF_TERM_SC2] = c.sc_lambda / 3.f * (tot[T_SC2] / Ng); terms[SNERF_TERM_SC3] = c.sc_lambda / 3.f * (tot[T_SC3] / Ng); }
    if (c.sem_mode == 1) terms[SNERF_TERM_SEMANTIC] = c.lambda_s * ce_mean;
    if (c.sem_mode == 2) {
      terms[SNERF_TERM_SEMANTIC] = c.lambda_s * (ce_mean * invb_mean);
      if (c.use_sbeta) terms[SNERF_TERM_SEMANTIC_LOGBETA] = c.lambda_s * (3.f + tot[T_LOGBS] / Ng) * 0.5f;
    }
    if (c.car_reg) terms[SNERF_TERM_CAR_REG] = c.lambda_c * (tot[T_CAR] / tot[T_CARCNT]);  // NaN if no car ray, as in the reference
    if (c.has_depth) terms[SNERF_TERM_DS] = c.ds_lambda / 3.f * (tot[T_DS] / Ng);
  }
  for (int ray = wave_g; ray < c.n_rays; ray += nwaves) {
    const RayLoss r = ray_loss(c, in, ray, lane);
    const float bbar = r.wb + 0.05f;
    float g_bbar = 0.f;  // d loss / d (sum_j w_j beta_j) through beta (colour, L_t, beta-weighted CE without beta_s)
    float g_bs = 0.f;    // d loss / d (sum_j w_j beta_in_j) of the semantic-uncertainty term
    if (c.color_mode == 2) g_bbar += -r.d2 / (bbar * bbar * bbar * 3.f * Ng) + 1.f / (2.f * Ng * bbar);
    if (c.car_reg && r.car) g_bbar += c.lambda_c * (-2.f) * (1.f - r.wb) / tot[T_CARCNT];
    if (c.sem_mode == 2) {
      const float bs = (c.use_sbeta ? r.wbs : r.wb) + 0.05f;
      g_bs = c.lambda_s * ce_mean * (-1.f / (bs * bs * bs * Ng));
      if (c.use_sbeta) g_bs += c.lambda_s / (2.f * Ng * bs);
    }
    const bool need_wb = (c.color_mode == 2) || c.car_reg || (c.sem_mode == 2);
    for (int j = lane; j < S; j += 64) {
      const size_t p = (size_t)ray * S + j;
      if (need_wb) {
        const float w = in.weights[p], b = in.beta[p];
        float gw = g_bbar * b, gb = g_bbar * w, gbs = 0.f;
        if (c.sem_mode == 2) {
          if (c.use_sbeta) {
            gw += g_bs * in.beta_semantic[p];
            gbs = c.detach_beta_for_s ? 0.f : g_bs * w;
          } else {
            gw += g_bs * b;
            gb += c.detach_beta_for_s ? 0.f : g_bs * w;
          }
        }
        if (g.weights) g.weights[p] = gs * gw;
        if (g.beta) g.beta[p] = gs * gb;
        if (g.beta_semantic && c.use_sbeta) g.beta_semantic[p] = gs * gbs;
      } else {
        if (g.weights) g.weights[p] = 0.f;
        if (g.beta) g.beta[p] = 0.f;
        if (g.beta_semantic && c.use_sbeta) g.beta_semantic[p] = 0.f;
      }
      if (c.has_sc && g.sun_sc) {
        const float v = in.sun_sc[p];
        g.sun_sc[p] = gs * (c.sc_lambda / (3.f * Ng)) * (-2.f * (in.transparency_sc[p] - v) - in.weights_sc[p]);
      }
    }
    if (lane == 0) {
      if (g.rgb) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const float d = c.color_mode ? (in.rgb[(size_t)ray * 3 + k] - in.gt_rgb[(size_t)ray * 3 + k]) : 0.f;
          float v = 0.f;
          if (c.color_mode == 1) v = 2.f * d / (3.f * Ng);
          if (c.color_mode == 2) v = d / (bbar * bbar * 3.f * Ng);
          g.rgb[(size_t)ray * 3 + k] = gs * v;
        }
      }
      if (g.depth && c.has_depth) {
        const float w = in.depth_weights ? in.depth_weights[ray] : 1.f;
        g.depth[ray] = gs * (c.ds_lambda / 3.f) / Ng * 2.f * w * (in.depth[ray] - in.gt_depth[ray]);
      }
    }
    if (g.semantic_logits && C > 0) {
      float coef = 0.f;
      if (c.sem_mode == 1) coef = c.lambda_s / tot[T_CECNT];
      if (c.sem_mode == 2) coef = c.lambda_s * invb_mean / tot[T_CECNT];
      if (lane < C) {
        float v = 0.f;
        if (r.valid_ce) {
          const float* l = in.semantic_logits + (size_t)ray * C;
          const float sm = expf(l[lane] - r.mx - r.lse);
          v = coef * (sm - ((long long)lane == in.labels[ray] ? 1.f : 0.f));
        }
        g.semantic_logits[(size_t)ray * C + lane] = gs * v;
      }
    }
  }
}

static int loss_blocks(int N) { return max(1, min((N + 3) / 4, 256)); }

}  // namespace snerf

using namespace snerf;

static int check_loss(const SnerfLossCfg* c, const SnerfLossIn* in) {
  if (!c || !in) { set_error("snerf_loss: null argument"); return SNERF_ERR_NULL; }
  if (c->n_rays <= 0 || c->n_samples <= 0) { set_error("snerf_loss: n_rays and n_samples must be positive"); return SNERF_ERR_BAD_DESC; }
  if (c->n_classes < 0 || c->n_classes > 64) { set_error("snerf_loss: n_classes out of range"); return SNERF_ERR_BAD_DESC; }
  auto need = [&](const void* p, const char* name) { if (!p) { set_error("snerf_loss: '%s' is required by this SnerfLossCfg", name); return true; } return false; };
  if (c->color_mode && (need(in->rgb, "rgb") || need(in->gt_rgb, "gt_rgb"))) return SNERF_ERR_NULL;
  if ((c->color_mode == 2 || c->sem_mode == 2 || c->car_reg) && (need(in->weights, "weights") || need(in->beta, "beta"))) return SNERF_ERR_NULL;
  if (c->use_sbeta && c->sem_mode == 2 && need(in->beta_semantic, "beta_semantic")) return SNERF_ERR_NULL;
  if (c->has_sc && (need(in->sun_sc, "sun_sc") || need(in->transparency_sc, "transparency_sc") || need(in->weights_sc, "weights_sc"))) return SNERF_ERR_NULL;
  if (c->sem_mode && (need(in->semantic_logits, "semantic_logits") || need(in->labels, "labels") || c->n_classes <= 0)) { if (c->n_classes <= 0) set_error("snerf_loss: semantic loss needs n_classes > 0"); return SNERF_ERR_NULL; }
  if (c->car_reg && need(in->labels, "labels")) return SNERF_ERR_NULL;
  if (c->has_depth && (need(in->depth, "depth") || need(in->gt_depth, "gt_depth"))) return SNERF_ERR_NULL;
  return SNERF_OK;
}

extern "C" {

size_t snerf_loss_workspace_bytes(const SnerfLossCfg* cfg) {
  if (!cfg || cfg->n_rays <= 0) return 0;
  return ((size_t)loss_blocks(cfg->n_rays) * 4 + 64) * SNERF_LOSS_NTOT * sizeof(float);
}

int snerf_loss_partial(const SnerfLossCfg* cfg, const SnerfLossIn* in, float* totals, void* workspace,
                       size_t workspace_bytes, void* stream) {
  int rc = check_loss(cfg, in);
  if (rc) return rc;
  if (!totals || !workspace || workspace_bytes < snerf_loss_workspace_bytes(cfg)) { set_error("snerf_loss_partial: totals/workspace missing or too small"); return SNERF_ERR_WORKSPACE; }
  hipStream_t st = (hipStream_t)stream;
  const int blocks = loss_blocks(cfg->n_rays);
  float* partial = (float*)workspace;
  float* tmp = partial + (size_t)blocks * 4 * SNERF_LOSS_NTOT;
  hipLaunchKernelGGL(loss_partial_kernel, dim3(blocks), dim3(256), 0, st, *cfg, *in, partial);
  SNERF_LAUNCH_CHECK();
  { int rc = launch_zero_bytes(totals, SNERF_LOSS_NTOT * sizeof(float), st); if (rc) return rc; }
  return reduce_partials(partial, blocks * 4, SNERF_LOSS_NTOT, SNERF_LOSS_NTOT, tmp, totals, st);
}

int snerf_loss_finish(const SnerfLossCfg* cfg, const SnerfLossIn* in, const float* totals, float n_rays_global,
                      float grad_scale, float* terms, const SnerfLossGrads* grads, void* stream) {
  int rc = check_loss(cfg, in);
  if (rc) return rc;
  if (!totals || !grads) { set_error("snerf_loss_finish: null argument"); return SNERF_ERR_NULL; }
  if (!(n_rays_global >= 0.f)) { set_error("snerf_loss_finish: n_rays_global must be positive, or 0 to take the count from totals"); return SNERF_ERR_BAD_DESC; }
  hipLaunchKernelGGL(loss_finish_kernel, dim3(loss_blocks(cfg->n_rays)), dim3(256), 0, (hipStream_t)stream, *cfg, *in,
                     totals, n_rays_global, grad_scale, terms, *grads);
  SNERF_LAUNCH_CHECK();
  return SNERF_OK;
}

}  // extern "C"
