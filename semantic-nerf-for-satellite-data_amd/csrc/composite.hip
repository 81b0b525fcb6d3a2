// Irradiance-model alpha compositing, forward and backward: one 64-wide wavefront per ray, one lane
// per sample (S > 64 is walked in chunks of 64 with a carried transmittance), prefix products and
// per-ray sums via wavefront shuffles -- no LDS, no atomics.
//
// Replaces convert_sigmas (framework/util/rendering.py:4-34), the head activations and the
// compositing tail of inference() (semantic/models/rs_semantic.py:81-126: column split, irradiance
// = sun + (1-sun)*sky, rgb = clamp(sum w*albedo*irr), logits = sum w*sem, argmax), the sky-colour
// MLP (rs_semantic.py:229-234,296 -- evaluated once per RAY here: it only depends on sun_d), and
// their autograd backward.
#include "composite.h"

namespace snerf {

__device__ __forceinline__ float wave_incl_prod(float v, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const float y = __shfl_up(v, o, 64);
    if (lane >= o) v *= y;
  }
  return v;
}
// inclusive suffix sum: v_l + v_{l+1} + ... + v_63
__device__ __forceinline__ float wave_suffix_sum(float v, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const float y = __shfl_down(v, o, 64);
    if (lane + o < 64) v += y;
  }
  return v;
}

// sky colour for one ray: k = sigmoid(W2 relu(W0 sun + b0) + b2); hidden units spread over lanes
__device__ __forceinline__ void sky_forward(const float* __restrict__ sky, int H, float sx, float sy, float sz,
                                            int lane, float (&hu)[MAX_SKY_UNITS], float (&k)[3]) {
  const float* w0 = sky;
  const float* b0 = sky + 4 * H;
  const float* w2 = sky + 5 * H;
  const float* b2 = sky + 9 * H;
  float part[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < MAX_SKY_UNITS; ++i) {
    const int u = lane + 64 * i;
    hu[i] = 0.f;
    if (u < H) {
      const float pre = w0[u * 4 + 0] * sx + w0[u * 4 + 1] * sy + w0[u * 4 + 2] * sz + b0[u];
      hu[i] = fmaxf(pre, 0.f);
#pragma unroll
      for (int c = 0; c < 3; ++c) part[c] += w2[c * H + u] * hu[i];
    }
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) k[c] = sigmoid_f(wave_sum(part[c]) + b2[c]);
}

// pre-activation of a 1-wide head (sigma, sun visibility): the 32-wide buffer's column 0, or bias + the partial dot products of the
// producing launch's epilogue summed in a fixed order (CompArgs::sig_part / sun_part)
__device__ __forceinline__ float narrow_pre(const float* full, const float* part, const float* bias, int n, size_t stride, size_t p) {
  if (part == nullptr) return full[p * NARROW];
  float s = *bias;
  for (int q = 0; q < n; ++q) s += part[(size_t)q * stride + p];
  return s;
}

// pre-activation `o` of head `h` (0 rgb, 1 semantic, 2 beta, 3 beta_s; `col` = its column in the 32-wide buffer)
__device__ __forceinline__ float fin_pre(const CompArgs& a, size_t p, int col, int h, int o) {
  if (a.fin_part == nullptr) return a.fino[p * NARROW + col];
  const float* q = a.fin_part + (size_t)(a.fin_blk[h] * 4 * ND_FIN + o) * a.part_stride + p;
  float s = a.fin_bias[col];
#pragma unroll
  for (int w = 0; w < 4; ++w) s += q[(size_t)(w * ND_FIN) * a.part_stride];
  return s;
}

__global__ __launch_bounds__(256) void composite_fwd_kernel(CompArgs a) {
  const int lane = threadIdx.x & 63;
  const int wave_g = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int nwaves = gridDim.x * 4;
  const int S = a.S, C = a.C;
  for (int ray = wave_g; ray < a.N; ray += nwaves) {
    float k[3] = {0.f, 0.f, 0.f};
    if (!a.sc) {
      float hu[MAX_SKY_UNITS];
      const float* sd = a.sun_d + (size_t)ray * a.sun_stride;
      sky_forward(a.sky, a.H, sd[0], sd[1], sd[2], lane, hu, k);
    }
    float carryT = 1.f;
    float acc_rgb[3] = {0.f, 0.f, 0.f}, acc_depth = 0.f, acc_log[MAX_CLASSES];
#pragma unroll
    for (int c = 0; c < MAX_CLASSES; ++c) acc_log[c] = 0.f;

    for (int c0 = 0; c0 < S; c0 += 64) {
      const int j = c0 + lane;
      const bool valid = j < S;
      const size_t p = (size_t)ray * S + (valid ? j : 0);
      const float zj = a.z[p];
      const float delta = (j >= S - 1) ? 1e10f : (a.z[p + 1] - zj);
      const float spre = narrow_pre(a.sigo, a.sig_part, a.sig_bias, a.n_sig_part, a.part_stride, p);
      const float sigma = softplus_f(spre);
      const float e = expf(-delta * fmaxf(sigma, 0.f));
      const float alpha = 1.f - e;
      const float tau = valid ? ((1.f - alpha) + 1e-10f) : 1.f;
      const float incl = wave_incl_prod(tau, lane);
      float excl = __shfl_up(incl, 1, 64);
      if (lane == 0) excl = 1.f;
      const float T = carryT * excl;
      carryT = carryT * __shfl(incl, 63, 64);
      const float w = alpha * T;
      const float v = sigmoid_f(narrow_pre(a.suno, a.sun_part, a.sun_bias, a.n_sun_part, a.part_stride, p));
      if (valid) {
        if (a.o_weights) a.o_weights[p] = w;
        if (a.o_transparency) a.o_transparency[p] = T;
        if (a.o_sun) a.o_sun[p] = v;
        if (a.save_T) a.save_T[p] = T;
      }
      if (a.sc) continue;
      float al[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        al[c] = sigmoid_f(fin_pre(a, p, Plan::col_rgb + c, 0, c)) * 1.002f - 0.001f;
        const float irr = v + (1.f - v) * k[c];
        if (valid) acc_rgb[c] += (w * al[c]) * irr;
      }
      const float beta = a.o_beta ? softplus_f(fin_pre(a, p, Plan::col_beta, 2, 0)) : 0.f;   // (not computed when nobody asked: bsp_pass.hip, tj_skip)
      if (valid) {
        acc_depth += w * zj;
        if (a.o_sigmas) a.o_sigmas[p] = sigma;
        if (a.o_beta) a.o_beta[p] = beta;
        if (a.o_albedo) { a.o_albedo[p * 3 + 0] = al[0]; a.o_albedo[p * 3 + 1] = al[1]; a.o_albedo[p * 3 + 2] = al[2]; }
        if (a.o_sky) { a.o_sky[p * 3 + 0] = k[0]; a.o_sky[p * 3 + 1] = k[1]; a.o_sky[p * 3 + 2] = k[2]; }
        if (a.o_beta_s && a.has_sbeta) a.o_beta_s[p] = softplus_f(fin_pre(a, p, Plan::col_sbeta, 3, 0));
      }
#pragma unroll
      for (int c = 0; c < MAX_CLASSES; ++c) {
        if (c < C) {
          const float pre = fin_pre(a, p, Plan::col_sem + c, 1, c);
          const float q = a.sem_sigmoid ? sigmoid_f(pre) : pre;
          if (valid) acc_log[c] += w * q;
        }
      }
    }
    if (a.sc) continue;
    const float depth = wave_sum(acc_depth);
    float rgb[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) rgb[c] = wave_sum(acc_rgb[c]);
    float best = -INFINITY;
    int best_c = 0;
#pragma unroll
    for (int c = 0; c < MAX_CLASSES; ++c) {
      if (c < C) {
        const float l = wave_sum(acc_log[c]);
        if (lane == 0 && a.o_logits) a.o_logits[(size_t)ray * C + c] = l;
        if (l > best) { best = l; best_c = c; }  // first maximum, like torch.argmax on CPU
      }
    }
    if (lane == 0) {
      if (a.save_rgbraw) { a.save_rgbraw[ray * 3 + 0] = rgb[0]; a.save_rgbraw[ray * 3 + 1] = rgb[1]; a.save_rgbraw[ray * 3 + 2] = rgb[2]; }
      if (a.o_rgb)
        for (int c = 0; c < 3; ++c) a.o_rgb[(size_t)ray * 3 + c] = fminf(fmaxf(rgb[c], 0.f), 1.f);
      if (a.o_depth) a.o_depth[ray] = depth;
      if (a.o_label && C > 0) a.o_label[ray] = (long long)best_c;
    }
  }
}

int launch_composite_fwd(const CompArgs& a, hipStream_t st) {
  const int blocks = max(1, min((a.N + 3) / 4, 2048));
  hipLaunchKernelGGL(composite_fwd_kernel, dim3(blocks), dim3(256), 0, st, a);
  SNERF_LAUNCH_CHECK();
  return 0;
}

// ---- backward -----------------------------------------------------------------------------------------
// Per ray, with G_j = dL/dw_j (direct), Hj = dL/dT_j (direct) = g_T[j] + G_j*alpha_j... see DESIGN.md:
//   dL/dalpha_j = G_j T_j - (sum_{j'>j} Hd_j' T_j') / tau_j,  Hd_j = g_T[j] + G_j alpha_j
// (same algebra as torch's cumprod backward for non-zero inputs: tau_j >= 1e-10).
__global__ __launch_bounds__(256) void composite_bwd_kernel(CompBwdArgs b) {
  if (blockIdx.x == 0) for (int i = threadIdx.x; i < b.zero_n; i += 256) b.zero[i] = 0u;
  const CompArgs& a = b.f;
  const int lane = threadIdx.x & 63;
  const int wave_g = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int nwaves = gridDim.x * 4;
  const int S = a.S, C = a.C, H = a.H;
  const int nchunk = (S + 63) / 64;

  // sky parameter-gradient accumulators (per lane: its hidden units), summed over this wave's rays
  float g_w0[MAX_SKY_UNITS][3], g_b0[MAX_SKY_UNITS], g_w2[3][MAX_SKY_UNITS], g_b2[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < MAX_SKY_UNITS; ++i) {
    g_b0[i] = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) { g_w0[i][c] = 0.f; g_w2[c][i] = 0.f; }
  }

  for (int ray = wave_g; ray < a.N; ray += nwaves) {
    float k[3] = {0.f, 0.f, 0.f}, hu[MAX_SKY_UNITS];
    float sx = 0.f, sy = 0.f, sz = 0.f;
    float grgb[3] = {0.f, 0.f, 0.f}, gdepth = 0.f;
    if (!a.sc) {
      const float* sd = a.sun_d + (size_t)ray * a.sun_stride;
      sx = sd[0]; sy = sd[1]; sz = sd[2];
      sky_forward(a.sky, H, sx, sy, sz, lane, hu, k);
      if (b.g_rgb) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float raw = b.rgbraw[ray * 3 + c];
          grgb[c] = (raw >= 0.f && raw <= 1.f) ? b.g_rgb[(size_t)ray * 3 + c] : 0.f;  // clamp backward
        }
      }
      if (b.g_depth) gdepth = b.g_depth[ray];
    }
    float gk[3] = {0.f, 0.f, 0.f};  // d loss / d sky colour (per-lane partial)
    float suffix_carry = 0.f;       // sum over later chunks of Hd_j' T_j'
    for (int ch = nchunk - 1; ch >= 0; --ch) {
      const int j = ch * 64 + lane;
      const bool valid = j < S;
      const size_t p = (size_t)ray * S + (valid ? j : 0);
      const float zj = a.z[p];
      const float delta = (j >= S - 1) ? 1e10f : (a.z[p + 1] - zj);
      const float spre = narrow_pre(a.sigo, a.sig_part, a.sig_bias, a.n_sig_part, a.part_stride, p);
      const float sigma = softplus_f(spre);
      const float e = expf(-delta * fmaxf(sigma, 0.f));
      const float alpha = 1.f - e;
      const float tau = (1.f - alpha) + 1e-10f;
      const float T = b.T[p];
      const float w = alpha * T;
      const float vpre = narrow_pre(a.suno, a.sun_part, a.sun_bias, a.n_sun_part, a.part_stride, p);
      const float v = sigmoid_f(vpre);

      float G = b.g_weights ? b.g_weights[p] : 0.f;
      float d_apre[3] = {0.f, 0.f, 0.f}, d_bpre = 0.f, d_sbpre = 0.f, d_qpre[MAX_CLASSES];
      float g_v = b.g_sun ? b.g_sun[p] : 0.f;
#pragma unroll
      for (int c = 0; c < MAX_CLASSES; ++c) d_qpre[c] = 0.f;
      if (!a.sc) {
        G += gdepth * zj;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float sa = sigmoid_f(fin_pre(a, p, Plan::col_rgb + c, 0, c));
          const float al = sa * 1.002f - 0.001f;
          const float irr = v + (1.f - v) * k[c];
          G += grgb[c] * (al * irr);
          const float g_al = grgb[c] * (w * irr) + (b.g_albedo ? b.g_albedo[p * 3 + c] : 0.f);
          d_apre[c] = g_al * 1.002f * sa * (1.f - sa);
          const float g_irr = grgb[c] * (w * al);
          g_v += g_irr * (1.f - k[c]);
          if (valid) gk[c] += g_irr * (1.f - v) + (b.g_sky ? b.g_sky[p * 3 + c] : 0.f);
        }
        const float bpre = fin_pre(a, p, Plan::col_beta, 2, 0);
        d_bpre = (b.g_beta ? b.g_beta[p] : 0.f) * softplus_grad_f(bpre);
        if (a.has_sbeta) d_sbpre = (b.g_beta_s ? b.g_beta_s[p] : 0.f) * softplus_grad_f(fin_pre(a, p, Plan::col_sbeta, 3, 0));
#pragma unroll
        for (int c = 0; c < MAX_CLASSES; ++c) {
          if (c < C) {
            const float gl = b.g_logits ? b.g_logits[(size_t)ray * C + c] : 0.f;
            const float pre = fin_pre(a, p, Plan::col_sem + c, 1, c);
            const float q = a.sem_sigmoid ? sigmoid_f(pre) : pre;
            G += gl * q;
            d_qpre[c] = a.sem_sigmoid ? (gl * w) * q * (1.f - q) : gl * w;
          }
        }
      }
      const float gT = b.g_transparency ? b.g_transparency[p] : 0.f;
      const float X = valid ? (gT + G * alpha) * T : 0.f;
      const float incl = wave_suffix_sum(X, lane);
      float nxt = __shfl_down(incl, 1, 64);  // sum over lanes l' > l
      if (lane == 63) nxt = 0.f;
      const float suffix_excl = nxt + suffix_carry;
      suffix_carry += __shfl(incl, 0, 64);
      const float d_alpha = G * T - suffix_excl / tau;
      const float d_sigma = (sigma > 0.f ? d_alpha * (delta * e) : 0.f) + (b.g_sigmas ? b.g_sigmas[p] : 0.f);
      const float d_spre = d_sigma * softplus_grad_f(spre);
      const float d_vpre = g_v * v * (1.f - v);
      if (valid) {
        // full NARROW-wide rows: the GEMMs that consume these buffers read all 32 columns
        float4* o = reinterpret_cast<float4*>(b.d_sigo + p * NARROW);
        o[0] = make_float4(d_spre, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 1; i < 8; ++i) o[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        o = reinterpret_cast<float4*>(b.d_suno + p * NARROW);
        o[0] = make_float4(d_vpre, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 1; i < 8; ++i) o[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!a.sc) {
          float row[NARROW];
#pragma unroll
          for (int i = 0; i < NARROW; ++i) row[i] = 0.f;
          row[Plan::col_rgb + 0] = d_apre[0]; row[Plan::col_rgb + 1] = d_apre[1]; row[Plan::col_rgb + 2] = d_apre[2];
          row[Plan::col_beta] = d_bpre;
          row[Plan::col_sbeta] = d_sbpre;
#pragma unroll
          for (int c = 0; c < MAX_CLASSES; ++c) row[Plan::col_sem + c] = d_qpre[c];
          o = reinterpret_cast<float4*>(b.d_fino + p * NARROW);
#pragma unroll
          for (int i = 0; i < 8; ++i) o[i] = make_float4(row[4 * i], row[4 * i + 1], row[4 * i + 2], row[4 * i + 3]);
        }
      }
    }
    if (!a.sc) {
      // sky MLP backward for this ray
      const float* w2 = a.sky + 5 * H;
      float dpre[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        dpre[c] = wave_sum(gk[c]) * k[c] * (1.f - k[c]);
        g_b2[c] += dpre[c];  // identical on every lane; lane 0's copy is stored
      }
#pragma unroll
      for (int i = 0; i < MAX_SKY_UNITS; ++i) {
        const int u = lane + 64 * i;
        if (u < H) {
          float dh = 0.f;
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            g_w2[c][i] += dpre[c] * hu[i];
            dh += w2[c * H + u] * dpre[c];
          }
          if (hu[i] > 0.f) {
            g_b0[i] += dh;
            g_w0[i][0] += dh * sx; g_w0[i][1] += dh * sy; g_w0[i][2] += dh * sz;
          }
        }
      }
    }
  }
  if (!a.sc && b.sky_slab) {
    // one slab row per wave, laid out like the packed sky parameters: [H][4] | [H] | [4][H] | [4]
    float* o = b.sky_slab + (size_t)wave_g * (9 * H + 4);
#pragma unroll
    for (int i = 0; i < MAX_SKY_UNITS; ++i) {
      const int u = lane + 64 * i;
      if (u < H) {
        o[u * 4 + 0] = g_w0[i][0]; o[u * 4 + 1] = g_w0[i][1]; o[u * 4 + 2] = g_w0[i][2]; o[u * 4 + 3] = 0.f;
        o[4 * H + u] = g_b0[i];
        o[5 * H + 0 * H + u] = g_w2[0][i]; o[5 * H + 1 * H + u] = g_w2[1][i]; o[5 * H + 2 * H + u] = g_w2[2][i];
        o[5 * H + 3 * H + u] = 0.f;
      }
    }
    if (lane == 0) { o[9 * H + 0] = g_b2[0]; o[9 * H + 1] = g_b2[1]; o[9 * H + 2] = g_b2[2]; o[9 * H + 3] = 0.f; }
  }
}

int composite_bwd_blocks(int N) { return max(1, min((N + 3) / 4, 512)); }

int launch_composite_bwd(const CompBwdArgs& b, hipStream_t st) {
  hipLaunchKernelGGL(composite_bwd_kernel, dim3(composite_bwd_blocks(b.f.N)), dim3(256), 0, st, b);
  SNERF_LAUNCH_CHECK();
  return 0;
}

}  // namespace snerf
