// Forward / backward launch sequences of one rendering pass: every activation tensor lives as block-scaled fp16 planes
// (bsp.h; two planes in the default arithmetic, one under SNERF_FLAG_F16X1 -- Plan::pl, the kernels are templated on it),
// every dense layer is one launch of bsp_kc.hip / bsp_gemm.hip, weights come pre-packed in fragment order.
// Reference: semantic/models/rs_semantic.py:260-340 (forward), its autograd backward.
#include "aux_kernels.h"
#include "bsp.h"
#include "composite.h"
#include "plan.h"

namespace snerf {

#define RC(expr) do { int _rc = (expr); if (_rc) return _rc; } while (0)

namespace {
// Every other K-contiguous launch of a pass walks its tiles backwards (tiles.h): a consumer starts with the rows its producer
// wrote last.  Measured 372.0 -> 369.8 us per launch, 28.27 -> 28.12 ms per step (two A/B pairs on one box).
struct Ws {
  char* base;
  char* c(size_t off) const { return base + off; }
  float* f(size_t off) const { return reinterpret_cast<float*>(base + off); }
  int* i(size_t off) const { return reinterpret_cast<int*>(base + off); }
  unsigned* u(size_t off) const { return reinterpret_cast<unsigned*>(base + off); }
};
// weight operand `job` of the packed buffer (rows [row0, ..), k >= k0 of it)
void weights(bsp::KcArgs& g, const Plan& p, const float* pk, int job, int row0 = 0, int k0 = 0) {
  const char* planes = reinterpret_cast<const char*>(pk + p.n_fp32);
  g.W = planes + p.wj_off[job];
  g.EW = reinterpret_cast<const int*>(planes + p.wp_bytes) + p.wj_e[job];
  g.w_rb32 = (p.wj_rows[job] + 31) / 32;
  g.w_bytes = (unsigned)bsp::wp16_bytes(p.wj_rows[job], p.wj_K[job], p.pl);
  g.w_row0 = row0; g.w_k0 = k0;
  g.pl = p.pl;
}
// sigma / sun-visibility pre-activations of SIREN passes: partial dot products of the producing launches (Plan::nd_sig / nd_sun)
void narrow_parts(CompArgs& c, const Plan& p, const float* pk, const Ws& ws) {
  c.part_stride = p.Pp;
  if (p.nd_sig) { c.sig_part = ws.f(p.o_sigpart); c.n_sig_part = 4 * (p.W / 256); c.sig_bias = pk + p.b_fs + p.W; }
  if (p.nd_sun) { c.sun_part = ws.f(p.o_sunpart); c.n_sun_part = 4 * (p.H / 256); c.sun_bias = pk + p.b_s4; }
  if (p.nd_fin) {
    c.fin_part = ws.f(p.o_finpart); c.fin_bias = pk + p.b_fin;
    c.fin_blk[0] = p.blk_rgb; c.fin_blk[1] = p.blk_sem < 0 ? 0 : p.blk_sem; c.fin_blk[2] = p.blk_beta; c.fin_blk[3] = p.blk_sbeta < 0 ? 0 : p.blk_sbeta;
  }
}
}  // namespace

int forward_bsp(const Plan& p, const float* pk, const SnerfInputs* in, const SnerfOutputs* out, void* workspace, hipStream_t st) {
  const Ws ws{(char*)workspace};
  // tile counters of the K-contiguous launches: one zeroed 64-byte slot per launch, in launch order.  (Zeroed by a kernel:
  // with hipMemsetAsync the forward's and the backward's memset of this region became two identical memset nodes of a
  // captured HIP graph, and replays on ROCm 7.2 then ran the backward's launches on counters that had not been zeroed.)  The first
  // kernel of the pass does it on the side: the encode kernel here, the composite backward in backward_bsp.
  int kcq = 0;
  auto launch_kc = [&](bsp::KcArgs& g) { g.rev = kcq & 1; g.tile_ctr = kcq < KCQ_SLOTS ? ws.i(p.o_kcq) + 16 * kcq++ : nullptr; return bsp::launch_kc(g, st); };
  const int P = p.P, W = p.W, H = p.H;
  float* z = ws.f(p.o_z);
  // 1. depths
  if (in->z_vals) SNERF_HIP_CHECK(hipMemcpyAsync(z, in->z_vals, sizeof(float) * P, hipMemcpyDeviceToDevice, st));
  else RC(launch_sample_z(in->rays, in->z_steps, in->u, z, p.N, p.S, st));
  if (out->z_vals) SNERF_HIP_CHECK(hipMemcpyAsync(out->z_vals, z, sizeof(float) * P, hipMemcpyDeviceToDevice, st));
  // 2. positions + encoding + extras, written as planes with their block exponents
  EncodeArgs ea;
  ea.rays = in->xyz ? nullptr : in->rays; ea.xyz = in->xyz; ea.z = z;
  ea.sun_d = in->sun_d; ea.sun_stride = in->sun_stride; ea.t = in->t; ea.t_s = in->t_s;
  ea.dir_is_sun = (p.sc && !in->xyz) ? 1 : 0;
  ea.N = p.N; ea.S = p.S; ea.F = p.F; ea.Ep = p.Ep;
  ea.FA = p.FA; ea.W = p.Wf; ea.Xp = p.Xp; ea.x_sun = p.x_sun; ea.x_t = p.x_t; ea.x_ts = p.x_ts; ea.tau = p.tau;
  ea.zero = ws.u(p.o_kcq); ea.zero_n = KCQ_SLOTS * 16;
  RC(bsp::launch_encode_bsp(ea, ws.c(p.o_pe), ws.i(p.e_pe), ws.c(p.o_fa), ws.i(p.e_fa), p.Wf, p.pl, st));
  const size_t EB = 2 * (size_t)p.pl;   // bytes per element of a plane tensor
  if (p.Wf > W)   // pad columns between feats and extras (narrow test networks only): zero planes, read against zero weights
    RC(bsp::launch_zero_cols(ws.c(p.o_fa) + (size_t)W * EB, (size_t)p.FA * EB, (size_t)(p.Wf - W) * EB, P, st));
  // 3. trunk (rs_semantic.py:325-334)
  const int act = p.siren ? ACT_SIN : ACT_RELU;
  const bool fused = p.fuse_trunk && bsp::trunk_fusion_enabled();
  if (fused) {   // one persistent launch, the activation tile resident in LDS (bsp_trunk.hip); inference: only the last layer's planes + sigma's partials leave
    bsp::TrunkArgs g;
    g.pe = ws.c(p.o_pe); g.Epe = ws.i(p.e_pe); g.P = P; g.W = W; g.L = p.L; g.skip_mask = p.skip_mask;
    const char* planes = reinterpret_cast<const char*>(pk + p.n_fp32);
    for (int i = 0; i < p.L; ++i) {
      const int job = p.wj_tr[i];
      g.Wp[i] = planes + p.wj_off[job];
      g.EW[i] = reinterpret_cast<const int*>(planes + p.wp_bytes) + p.wj_e[job];
      g.w_bytes[i] = (unsigned)bsp::wp16_bytes(p.wj_rows[job], p.wj_K[job], 1);
      g.K[i] = p.k_tr[i];
      g.bias[i] = pk + p.b_tr[i];
      g.w0[i] = i == 0 ? 30.f : 1.f;
    }
    const bool fused_feats = !p.train;
    if (fused_feats) {  // feats (rs_semantic.py:338) rides as one more layer: written into the first W columns of the [feats | sun | t | t_s] tensor
      const int job = p.wj_fs, i = p.L;
      g.Wp[i] = planes + p.wj_off[job];
      g.EW[i] = reinterpret_cast<const int*>(planes + p.wp_bytes) + p.wj_e[job];
      g.w_bytes[i] = (unsigned)bsp::wp16_bytes(p.wj_rows[job], p.wj_K[job], 1);
      g.K[i] = W; g.bias[i] = pk + p.b_fs; g.w0[i] = 1.f;
      g.F = ws.c(p.o_fa); g.EF = ws.i(p.e_fa); g.ldf = p.FA;
    }
    for (int i = 0; i < p.L && p.train; ++i) {   // training (feats stays a launch of its own): every layer's planes and sign words leave for the backward pass
      g.H[i] = ws.c(p.o_h[i]); g.EH[i] = ws.i(p.e_h[i]); g.Hsign[i] = ws.u(p.o_c[i]);
    }
    g.nd_w = pk + p.w_fs + (size_t)W * W; g.nd_out = ws.f(p.o_sigpart); g.nd_stride = p.Pp;
    g.tile_ctr = ws.i(p.o_kcq) + 16 * kcq++;
    RC(bsp::launch_trunk(g, p.train, st));
  }
  for (int i = 0; i < p.L && !fused; ++i) {
    bsp::KcArgs g;
    const bool skip = (p.skip_mask >> i) & 1u;
    if (i == 0) { g.A = ws.c(p.o_pe); g.EA = ws.i(p.e_pe); g.lda = p.Ep; g.Ka = p.Ep; }
    else if (skip) { g.A = ws.c(p.o_pe); g.EA = ws.i(p.e_pe); g.lda = p.Ep; g.Ka = p.Ep;
                     g.A2 = ws.c(p.o_h[i - 1]); g.EA2 = ws.i(p.e_h[i - 1]); g.lda2 = W; }
    else { g.A = ws.c(p.o_h[i - 1]); g.EA = ws.i(p.e_h[i - 1]); g.lda = W; g.Ka = W; }
    weights(g, p, pk, p.wj_tr[i]);
    g.I = P; g.J = W; g.K = p.k_tr[i];
    g.C = ws.c(p.o_h[i]); g.EC = ws.i(p.e_h[i]); g.ldc = W;
    g.bias = pk + p.b_tr[i]; g.act = act; g.w0 = (p.siren && i == 0) ? 30.f : 1.f;
    if (p.train && p.siren) g.Csign = ws.u(p.o_c[i]);
    if (i == p.L - 1 && p.nd_sig) {   // sigma's 1-wide projection rides in this launch's epilogue (bsp_kc.hip: NDOT)
      g.nd_w = pk + p.w_fs + (size_t)W * W; g.nd_out = ws.f(p.o_sigpart); g.nd_stride = p.Pp;
    }
    RC(launch_kc(g));
  }
  const char* hl = ws.c(p.o_h[p.L - 1]); const int* ehl = ws.i(p.e_h[p.L - 1]);
  if (!p.nd_sig) {  // sigma pre-activation (rs_semantic.py:337) -> 32-wide fp32 buffer, column 0
    bsp::KcArgs g;
    g.A = hl; g.EA = ehl; g.lda = W; g.Ka = W; weights(g, p, pk, p.wj_sig);
    g.I = P; g.J = NARROW; g.K = W; g.Cf = ws.f(p.o_sigo); g.bias = pk + p.b_fs + W;
    RC(bsp::launch_kc_narrow(g, st));
  }
  if (!fused || p.train) {  // feats (rs_semantic.py:338), written into the first W columns of the [feats | sun | t | t_s] tensor
    bsp::KcArgs g;
    g.A = hl; g.EA = ehl; g.lda = W; g.Ka = W; weights(g, p, pk, p.wj_fs);
    g.I = P; g.J = W; g.K = W; g.C = ws.c(p.o_fa); g.EC = ws.i(p.e_fa); g.ldc = p.FA; g.bias = pk + p.b_fs;
    RC(launch_kc(g));
  }
  const int r0 = p.sc ? p.sun_col : 0;
  {  // first layer of every head in one GEMM (sc pass: sun-visibility block only)
    bsp::KcArgs g;
    g.A = ws.c(p.o_fa); g.EA = ws.i(p.e_fa); g.lda = p.FA; g.Ka = p.FA; weights(g, p, pk, p.wj_h1, r0);
    g.I = P; g.J = p.h1w; g.K = p.FA; g.C = ws.c(p.o_h1); g.EC = ws.i(p.e_h1); g.ldc = p.h1w;
    g.bias = pk + p.b_h1 + r0; g.act = act; g.w0 = 1.f;
    if (p.train && p.siren) g.Csign = ws.u(p.o_c1);
    if (p.nd_fin) {   // the heads' final layers (block-diagonal [32][KF]: block b's rows read only block b's 256 columns) in this launch's epilogue
      g.nd_w = pk + p.w_fin; g.nd_ldw = p.KF; g.nd_omax = ND_FIN; g.nd_out = ws.f(p.o_finpart); g.nd_stride = p.Pp;
      auto blk = [&](int b, int col, int n) { if (b >= 0) { g.nd_rows[b] = n; g.nd_row0[b] = col; } };
      blk(p.blk_rgb, Plan::col_rgb, 3); blk(p.blk_sem, Plan::col_sem, p.C); blk(p.blk_beta, Plan::col_beta, 1); blk(p.blk_sbeta, Plan::col_sbeta, 1);
      // A frame that asks for no beta (full-frame inference: rgb / depth / labels -- eval/extract_pointcloud.py:66-79) does not compute the
      // beta block: a quarter of this launch.  Only with the finals folded (each block's final rows read its own tile only; the 32-wide
      // final launch would contract the unwritten columns) and when beta is this pass's only use of the block.
      if (!p.train && H == 256 && p.blk_beta >= 0 && out->beta == nullptr && !p.rgb_t && p.blk_sbeta < 0) g.tj_skip = p.blk_beta;
    }
    RC(launch_kc(g));
  }
  const int sun_col = p.sc ? 0 : p.sun_col;
  {  // sun visibility layers 2, 3 (rs_semantic.py:217-227)
    bsp::KcArgs g;
    g.A = ws.c(p.o_h1); g.EA = ws.i(p.e_h1); g.lda = p.h1w; g.a_col0 = sun_col; g.Ka = H; weights(g, p, pk, p.wj_s2);
    g.I = P; g.J = H; g.K = H; g.C = ws.c(p.o_s2); g.EC = ws.i(p.e_s2); g.ldc = H; g.bias = pk + p.b_s2; g.act = act;
    if (p.train && p.siren) g.Csign = ws.u(p.o_cs2);
    RC(launch_kc(g));
    g.A = ws.c(p.o_s2); g.EA = ws.i(p.e_s2); g.lda = H; g.a_col0 = 0; weights(g, p, pk, p.wj_s3);
    g.C = ws.c(p.o_s3); g.EC = ws.i(p.e_s3); g.bias = pk + p.b_s3;
    if (p.train && p.siren) g.Csign = ws.u(p.o_cs3);
    if (p.nd_sun) { g.nd_w = pk + p.w_s4; g.nd_out = ws.f(p.o_sunpart); g.nd_stride = p.Pp; }   // the sun-visibility output likewise
    RC(launch_kc(g));
  }
  if (!p.nd_sun) {  // sun visibility output pre-activation
    bsp::KcArgs g;
    g.A = ws.c(p.o_s3); g.EA = ws.i(p.e_s3); g.lda = H; g.Ka = H; weights(g, p, pk, p.wj_s4);
    g.I = P; g.J = NARROW; g.K = H; g.Cf = ws.f(p.o_suno); g.bias = pk + p.b_s4;
    RC(bsp::launch_kc_narrow(g, st));
  }
  if (!p.sc && !p.nd_fin) {  // last layer of rgb / beta / beta_s / semantic heads: block-diagonal [32][KF]
    bsp::KcArgs g;
    g.A = ws.c(p.o_h1); g.EA = ws.i(p.e_h1); g.lda = p.h1w; g.Ka = p.KF; weights(g, p, pk, p.wj_fin);
    g.I = P; g.J = NARROW; g.K = p.KF; g.Cf = ws.f(p.o_fino); g.bias = pk + p.b_fin;
    RC(bsp::launch_kc_narrow(g, st));
  }
  // 4. composite (reads the three 32-wide fp32 buffers)
  CompArgs c;
  c.N = p.N; c.S = p.S; c.H = H; c.C = p.C; c.sc = p.sc; c.sem_sigmoid = p.sem_sigmoid; c.has_sbeta = p.blk_sbeta >= 0;
  c.z = z; c.sigo = ws.f(p.o_sigo); c.fino = ws.f(p.o_fino); c.suno = ws.f(p.o_suno);
  narrow_parts(c, p, pk, ws);
  c.sun_d = in->sun_d; c.sun_stride = in->sun_stride; c.sky = pk + p.sky;
  c.o_rgb = out->rgb; c.o_depth = out->depth; c.o_weights = out->weights; c.o_transparency = out->transparency;
  c.o_albedo = out->albedo; c.o_sun = out->sun; c.o_sky = out->sky; c.o_beta = out->beta; c.o_sigmas = out->sigmas;
  c.o_beta_s = out->beta_semantic; c.o_logits = out->semantic_logits; c.o_label = (long long*)out->semantic_label;
  if (p.train) { c.save_T = ws.f(p.o_T); c.save_rgbraw = ws.f(p.o_rgbraw); }
  RC(launch_composite_fwd(c, st));
  return SNERF_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
namespace {
// Deferred reductions: every dW launch writes its split slabs, every dX launch its column-sum partials, into a region of
// its OWN inside the reduction arena (p.o_rq); the reductions into the packed gradient buffer are queued and run as two
// launches at the end of the pass (aux_kernels.hip: launch_reductions) instead of ~60 small ones in between the GEMMs.
struct RQ {
  float* base; size_t cap, used = 0; bool over = false;
  RedTable elem, col;
  float* take(size_t floats) {
    const size_t n = round_up_sz(floats, 64);
    if (used + n > cap) { over = true; return base; }
    float* r = base + used; used += n; return r;
  }
};
struct DwMat { DwSplit sp; size_t stride; int ldw; float* slab; };
// `split_rows`: the rows the split count is chosen for when they differ from the slab's (the [W + 32][W] matrix of feats + sigma:
// its wide launch covers W rows -- counting the 32 sigma rows as a third row tile gave it 43 splits, 172 workgroups for 256 CUs)
DwMat dw_begin(const Plan& p, RQ& rq, int rows, int ldw, int cols, bool narrow_rows, int split_rows = 0) {
  DwMat m;
  m.sp = dw_choose_bsp(p.P, split_rows > 0 ? split_rows : rows, cols, narrow_rows);
  m.stride = round_up_sz((size_t)rows * ldw, 64);
  m.ldw = ldw;
  m.slab = rq.take(m.stride * m.sp.ns);
  return m;
}
// slab[split][i][slab_off + j] = sum_p dZ[p][dz_col0 + i] X[p][x_col0 + j]
int dw_gemm(const Plan& p, const DwMat& m, const char* dz, const int* edz, int lddz, int dz_col0, int I, bool narrow_i,
            const char* X, const int* ex, int ldx, int x_col0, int J, size_t slab_off, hipStream_t st) {
  bsp::DwArgs g;
  g.A = dz; g.EA = edz; g.lda = lddz; g.a_col0 = dz_col0;
  g.B = X; g.EB = ex; g.ldb = ldx; g.b_col0 = x_col0;
  g.I = I; g.J = J; g.P = p.P;
  g.C = m.slab + slab_off; g.ldc = m.ldw;
  g.k_split = m.sp.k_split; g.n_split = m.sp.ns; g.slab_stride = m.stride; g.pl = p.pl;
  return bsp::launch_dw(g, narrow_i, st);
}
int dw_reduce(RQ& rq, const DwMat& m, size_t count, float* gout) { return red_add_elem(rq.elem, m.slab, m.sp.ns, m.stride, count, gout); }
// column-sum partials of a dX launch: one row per 128-row tile, `width` columns
inline int cs_ld(int width) { return (width + 3) & ~3; }
float* cs_take(const Plan& p, RQ& rq, int width) { return rq.take((size_t)((p.P + 127) / 128) * cs_ld(width)); }
int bias_from_colsum(const Plan& p, RQ& rq, const float* cs, int width, float* gout) {
  return red_add_col(rq.col, cs, (p.P + 127) / 128, (size_t)cs_ld(width), width, gout);
}
// bias gradient of a 32-wide pre-activation gradient; the same pass writes its planes + exponents (the dX / dW operands)
int narrow_grad(const Plan& p, RQ& rq, const float* dnar, char* planes, int* E, float* gout, hipStream_t st) {
  const int nb = (p.P + 255) / 256;
  float* part = rq.take((size_t)nb * NARROW);
  RC(bsp::launch_colsum32_bsp(dnar, p.P, part, planes, E, p.pl, st));
  return red_add_col(rq.col, part, nb, NARROW, NARROW, gout);
}
}  // namespace

// floats of the reduction arena: the same sequence of takes as backward_bsp below
size_t bsp_rq_floats(const Plan& p) {
  size_t n = 0;
  auto take = [&](size_t f) { n += round_up_sz(f, 64); };
  auto slab = [&](int rows, int ldw, int cols, bool narrow, int split_rows = 0) { take(round_up_sz((size_t)rows * ldw, 64) * dw_choose_bsp(p.P, split_rows > 0 ? split_rows : rows, cols, narrow).ns); };
  auto cs = [&](int width) { take((size_t)((p.P + 127) / 128) * ((width + 3) & ~3)); };
  auto nar = [&]() { take((size_t)((p.P + 255) / 256) * NARROW); };
  const int W = p.W, H = p.H;
  if (!p.sc) { nar(); slab(NARROW, p.KF, p.KF, true); cs(p.KF); }
  nar(); slab(NARROW, H, H, true); slab(H, H, H, false); slab(H, H, H, false); cs(H); cs(H); cs(H);
  slab(p.h1w, p.FA, p.FA, false); cs(p.FA);
  nar(); slab(W + NARROW, W, W, false, W); cs(W);
  for (int i = p.L - 1; i >= 0; --i) {
    const bool skip = i > 0 && ((p.skip_mask >> i) & 1u);
    if (skip) { slab(W, p.Ep, p.Ep, false); slab(W, W, W, false); }      // [gamma | h] columns: two slab regions (backward_bsp)
    else slab(W, p.k_tr[i], i == 0 ? p.Ep : W, false);
    if (i > 0) cs(W);
  }
  return n;
}

int backward_bsp(const Plan& p, const float* pk, const SnerfInputs* in, const SnerfOutGrads* go, float* gp, float* d_t, float* d_t_s,
                 void* workspace, hipStream_t st) {
  const Ws ws{(char*)workspace};
  const int P = p.P, W = p.W, H = p.H;
  int kcq = 0;     // (the counters are cleared by the composite backward kernel, the first launch of this pass)
  auto launch_kc = [&](bsp::KcArgs& g) { g.rev = kcq & 1; g.tile_ctr = kcq < KCQ_SLOTS ? ws.i(p.o_kcq) + 16 * kcq++ : nullptr; return bsp::launch_kc(g, st); };
  // activation derivative in a dX epilogue, rebuilt from the stored activation h (planes o_h / exponents e_h, leading
  // dimension ld, column col0): siren w0 * sign(cos) * sqrt(1 - h^2) with the sign words o_c; relu: h > 0
  auto dact = [&](bsp::KcArgs& g, size_t o_c, size_t o_h, size_t e_h, int ld, int col0 = 0, float w0 = 1.f) {
    g.H = ws.c(o_h); g.EH = ws.i(e_h); g.ldh = ld; g.h_col0 = col0;
    if (p.siren) { g.aux_mode = AUX_SINREC; g.Hsign = ws.u(o_c); g.w0 = w0; }
    else g.aux_mode = AUX_RELU_MASK;
  };
  RQ rq{ws.f(p.o_rq), p.rq_floats};
  auto colsum = [&](bsp::KcArgs& g, int width) { g.colsum = cs_take(p, rq, width); g.ldcs = cs_ld(width); return g.colsum; };
  const float* cs_ = nullptr;
  float* dsig = ws.f(p.o_dsig); float* dfin = ws.f(p.o_dfin); float* dsun = ws.f(p.o_dsun);
  // 0. composite backward -> gradients of the 32-wide pre-activations (+ sky MLP grads)
  CompBwdArgs b;
  CompArgs& c = b.f;
  c.N = p.N; c.S = p.S; c.H = H; c.C = p.C; c.sc = p.sc; c.sem_sigmoid = p.sem_sigmoid; c.has_sbeta = p.blk_sbeta >= 0;
  c.z = ws.f(p.o_z); c.sigo = ws.f(p.o_sigo); c.fino = ws.f(p.o_fino); c.suno = ws.f(p.o_suno);
  narrow_parts(c, p, pk, ws);
  c.sun_d = in->sun_d; c.sun_stride = in->sun_stride; c.sky = pk + p.sky;
  b.T = ws.f(p.o_T); b.rgbraw = ws.f(p.o_rgbraw);
  b.g_rgb = go->rgb; b.g_depth = go->depth; b.g_weights = go->weights; b.g_transparency = go->transparency;
  b.g_albedo = go->albedo; b.g_sun = go->sun; b.g_sky = go->sky; b.g_beta = go->beta; b.g_sigmas = go->sigmas;
  b.g_beta_s = go->beta_semantic; b.g_logits = go->semantic_logits;
  b.d_sigo = dsig; b.d_fino = dfin; b.d_suno = dsun; b.sky_slab = p.sc ? nullptr : ws.f(p.o_skyslab);
  b.zero = ws.u(p.o_kcq); b.zero_n = KCQ_SLOTS * 16;
  RC(launch_composite_bwd(b, st));
  if (!p.sc)
    RC(red_add_col(rq.col, ws.f(p.o_skyslab), p.comp_blocks * 4, (size_t)p.sky_floats, p.sky_floats, gp + p.sky));

  char* dz1 = ws.c(p.o_dza); int* edz1 = ws.i(p.e_dza);   // d(pre-activation) of the fused first head layer, [P][h1w]
  const int sun_col = p.sc ? 0 : p.sun_col;
  if (!p.sc) {
    // 1. final head layers: bias gradient + planes of dfin, dW, then dz1[:, :KF] = (dfin . W_fin) * act'
    RC(narrow_grad(p, rq, dfin, ws.c(p.o_pdfin), ws.i(p.e_dfin), gp + p.b_fin, st));
    const DwMat mf = dw_begin(p, rq, NARROW, p.KF, p.KF, true);
    RC(dw_gemm(p, mf, ws.c(p.o_pdfin), ws.i(p.e_dfin), NARROW, 0, NARROW, true, ws.c(p.o_h1), ws.i(p.e_h1), p.h1w, 0, p.KF, 0, st));
    RC(dw_reduce(rq, mf, (size_t)NARROW * p.KF, gp + p.w_fin));
    bsp::KcArgs g;
    g.A = ws.c(p.o_pdfin); g.EA = ws.i(p.e_dfin); g.lda = NARROW; g.Ka = NARROW; weights(g, p, pk, p.wj_tfin);
    g.I = P; g.J = p.KF; g.K = NARROW; g.C = dz1; g.EC = edz1; g.ldc = p.h1w;
    dact(g, p.o_c1, p.o_h1, p.e_h1, p.h1w);
    cs_ = colsum(g, p.KF);
    RC(launch_kc(g));
    RC(bias_from_colsum(p, rq, cs_, p.KF, gp + p.b_h1));
  }
  {  // 2. sun visibility chain: output layer, layer 3, layer 2
    RC(narrow_grad(p, rq, dsun, ws.c(p.o_pdsun), ws.i(p.e_dsun), gp + p.b_s4, st));
    const DwMat m4 = dw_begin(p, rq, NARROW, H, H, true);
    const DwMat mh = dw_begin(p, rq, H, H, H, false), mh2 = dw_begin(p, rq, H, H, H, false);   // one slab region per matrix
    RC(dw_gemm(p, m4, ws.c(p.o_pdsun), ws.i(p.e_dsun), NARROW, 0, NARROW, true, ws.c(p.o_s3), ws.i(p.e_s3), H, 0, H, 0, st));
    RC(dw_reduce(rq, m4, (size_t)NARROW * H, gp + p.w_s4));
    bsp::KcArgs g;
    g.A = ws.c(p.o_pdsun); g.EA = ws.i(p.e_dsun); g.lda = NARROW; g.Ka = NARROW; weights(g, p, pk, p.wj_ts4);
    g.I = P; g.J = H; g.K = NARROW; g.C = ws.c(p.o_dsa); g.EC = ws.i(p.e_dsa); g.ldc = H;
    dact(g, p.o_cs3, p.o_s3, p.e_s3, H);
    cs_ = colsum(g, H);
    RC(launch_kc(g));  // dz_s3
    RC(bias_from_colsum(p, rq, cs_, H, gp + p.b_s3));
    RC(dw_gemm(p, mh, ws.c(p.o_dsa), ws.i(p.e_dsa), H, 0, H, false, ws.c(p.o_s2), ws.i(p.e_s2), H, 0, H, 0, st));
    RC(dw_reduce(rq, mh, (size_t)H * H, gp + p.w_s3));
    g.A = ws.c(p.o_dsa); g.EA = ws.i(p.e_dsa); g.lda = H; g.Ka = H; g.K = H; weights(g, p, pk, p.wj_ts3);
    g.C = ws.c(p.o_dsb); g.EC = ws.i(p.e_dsb);
    dact(g, p.o_cs2, p.o_s2, p.e_s2, H);
    cs_ = colsum(g, H);
    RC(launch_kc(g));  // dz_s2
    RC(bias_from_colsum(p, rq, cs_, H, gp + p.b_s2));
    RC(dw_gemm(p, mh2, ws.c(p.o_dsb), ws.i(p.e_dsb), H, 0, H, false, ws.c(p.o_h1), ws.i(p.e_h1), p.h1w, sun_col, H, 0, st));
    RC(dw_reduce(rq, mh2, (size_t)H * H, gp + p.w_s2));
    g.A = ws.c(p.o_dsb); g.EA = ws.i(p.e_dsb); weights(g, p, pk, p.wj_ts2);
    g.C = dz1; g.EC = edz1; g.ldc = p.h1w; g.c_col0 = sun_col;
    dact(g, p.o_c1, p.o_h1, p.e_h1, p.h1w, sun_col);
    cs_ = colsum(g, H);
    RC(launch_kc(g));  // dz1[:, sun block]
    RC(bias_from_colsum(p, rq, cs_, H, gp + p.b_h1 + (size_t)p.sun_col));
  }
  char* dfa = ws.c(p.o_dzb); int* edfa = ws.i(p.e_dzb);   // [P][FA]
  {  // 3. fused first head layer: dW, then d[feats | extras]
    const int r0 = p.sc ? p.sun_col : 0;
    const DwMat m1 = dw_begin(p, rq, p.h1w, p.FA, p.FA, false);
    RC(dw_gemm(p, m1, dz1, edz1, p.h1w, 0, p.h1w, false, ws.c(p.o_fa), ws.i(p.e_fa), p.FA, 0, p.FA, 0, st));
    RC(dw_reduce(rq, m1, (size_t)p.h1w * p.FA, gp + p.w_h1 + (size_t)r0 * p.FA));
    bsp::KcArgs g;
    g.A = dz1; g.EA = edz1; g.lda = p.h1w; g.Ka = p.h1w; weights(g, p, pk, p.wj_th1, 0, r0);
    // d feats only (J = W): the 16 extras columns would be a third column tile of 256 for 16 useful columns (a third of this
    // launch: 954 -> ~640 us at 4096 x 64), and nothing needs d sun_d.  The gradient of the transient codes comes from a 32-wide
    // launch over the head blocks that read them instead (default: the beta block alone, K = H).
    g.I = P; g.J = W; g.K = p.h1w; g.C = dfa; g.EC = edfa; g.ldc = p.FA;
    cs_ = colsum(g, p.FA);   // columns [0, W) = bias gradient of feats_from_xyz
    RC(launch_kc(g));
    RC(red_add_col(rq.col, cs_, (P + 127) / 128, (size_t)cs_ld(p.FA), W, gp + p.b_fs));
    const bool want_t = d_t != nullptr, want_ts = d_t_s != nullptr && p.x_ts >= 0;
    if (p.sc) {   // the sun-visibility block reads [feats | sun_d] only: no gradient reaches t / t_s through this pass
      if (want_t) RC(launch_zero_bytes(d_t, (size_t)p.N * p.tau * sizeof(float), st));
      if (want_ts) RC(launch_zero_bytes(d_t_s, (size_t)p.N * p.tau * sizeof(float), st));
    } else if (want_t || want_ts) {
      // d extras[p][c] = sum_j dz1[p][j] W_h1[j][Wf + c] over the blocks whose first layer reads t or t_s (api.hip: head1)
      int b_lo = p.blk_beta, b_hi = p.blk_beta;
      auto use = [&](int blk, bool on) { if (on && blk >= 0) { if (blk < b_lo) b_lo = blk; if (blk > b_hi) b_hi = blk; } };
      use(p.blk_rgb, p.rgb_t); use(p.blk_sem, p.sem_t || p.sem_ts); use(p.blk_sbeta, true);
      const int k_lo = b_lo * H, k_n = (b_hi + 1 - b_lo) * H;
      float* dext = ws.f(p.o_dfin);      // [P][32] fp32: the final-layer gradients that lived here were consumed in step 1
      bsp::KcArgs x;
      x.A = dz1; x.EA = edz1; x.lda = p.h1w; x.a_col0 = k_lo; x.Ka = k_n; weights(x, p, pk, p.wj_th1, p.Wf, k_lo);
      x.I = P; x.J = p.Xp; x.K = k_n; x.Cf = dext;
      RC(bsp::launch_kc_narrow(x, st));
      if (want_t) RC(launch_ray_sum32(dext, p.x_t, p.N, p.S, p.tau, d_t, st));
      if (want_ts) RC(launch_ray_sum32(dext, p.x_ts, p.N, p.S, p.tau, d_t_s, st));
    }
  }
  char* dz = ws.c(p.o_dza); int* edz = ws.i(p.e_dza);   // dz1 is dead from here on
  {  // 4. feats + sigma: dW for the [W + 32][W] matrix, then dz of the last trunk layer
    const char* hl = ws.c(p.o_h[p.L - 1]); const int* ehl = ws.i(p.e_h[p.L - 1]);
    // The density branch carries a gradient only if one reaches weights / transparency (/ sigmas, depth, rgb, logits in the main
    // pass).  In the solar-correction pass of a training step none does: the loss detaches T' and w' (baseline/components/loss.py:8-10)
    // and the composite backward then writes d sigma = 0 exactly -- its bias sums, its dW launch and the 32 extra contraction columns
    // of the dX launch are skipped (SURVEY 8(d) counts the sc backward "through sun_v / feats / trunk only").
    const bool sig_live = !p.sc || go->weights != nullptr || go->transparency != nullptr || go->sigmas != nullptr;
    if (sig_live) RC(narrow_grad(p, rq, dsig, ws.c(p.o_pdsig), ws.i(p.e_dsig), gp + p.b_fs + W, st));
    const DwMat ms = dw_begin(p, rq, W + NARROW, W, W, false, W);
    RC(dw_gemm(p, ms, dfa, edfa, p.FA, 0, W, false, hl, ehl, W, 0, W, 0, st));
    if (sig_live) RC(dw_gemm(p, ms, ws.c(p.o_pdsig), ws.i(p.e_dsig), NARROW, 0, NARROW, true, hl, ehl, W, 0, W, (size_t)W * W, st));
    RC(dw_reduce(rq, ms, (size_t)(sig_live ? W + NARROW : W) * W, gp + p.w_fs));
    bsp::KcArgs g;
    g.A = dfa; g.EA = edfa; g.lda = p.FA; g.Ka = W;
    if (sig_live) { g.A2 = ws.c(p.o_pdsig); g.EA2 = ws.i(p.e_dsig); g.lda2 = NARROW; }
    weights(g, p, pk, p.wj_tfs); g.I = P; g.J = W; g.K = sig_live ? W + NARROW : W;
    g.C = dz; g.EC = edz; g.ldc = W;
    dact(g, p.o_c[p.L - 1], p.o_h[p.L - 1], p.e_h[p.L - 1], W, 0, (p.L == 1) ? 30.f : 1.f);
    cs_ = colsum(g, W);
    RC(launch_kc(g));
    RC(bias_from_colsum(p, rq, cs_, W, gp + p.b_tr[p.L - 1]));
  }
  // 5. trunk, last layer to first
  char* dz_cur = dz; int* edz_cur = edz;
  char* dz_nxt = ws.c(p.o_dzb); int* edz_nxt = ws.i(p.e_dzb);
  for (int i = p.L - 1; i >= 0; --i) {
    const bool skip = (p.skip_mask >> i) & 1u;
    if (i > 0 && skip) {
      // skip layer W_i = [W_gamma (Ep columns) | W_h (W columns)]: the two column blocks are two contractions of very different
      // width over the same dz, so each gets a slab region and a split count of its own (the 64-column block: two tiles x 128 splits;
      // sharing the wide block's 64 splits left half the chip idle for it: 275 us against 180) and a 2-D reduction into its columns
      const DwMat mg = dw_begin(p, rq, W, p.Ep, p.Ep, false), mh = dw_begin(p, rq, W, W, W, false);
      RC(dw_gemm(p, mg, dz_cur, edz_cur, W, 0, W, false, ws.c(p.o_pe), ws.i(p.e_pe), p.Ep, 0, p.Ep, 0, st));
      RC(dw_gemm(p, mh, dz_cur, edz_cur, W, 0, W, false, ws.c(p.o_h[i - 1]), ws.i(p.e_h[i - 1]), W, 0, W, 0, st));
      RC(red_add_elem2d(rq.elem, mg.slab, mg.sp.ns, mg.stride, W, p.Ep, p.Ep, gp + p.w_tr[i], p.k_tr[i]));
      RC(red_add_elem2d(rq.elem, mh.slab, mh.sp.ns, mh.stride, W, W, W, gp + p.w_tr[i] + p.Ep, p.k_tr[i]));
    } else {
      const DwMat mt = dw_begin(p, rq, W, p.k_tr[i], i == 0 ? p.Ep : W, false);
      if (i == 0) RC(dw_gemm(p, mt, dz_cur, edz_cur, W, 0, W, false, ws.c(p.o_pe), ws.i(p.e_pe), p.Ep, 0, p.Ep, 0, st));
      else RC(dw_gemm(p, mt, dz_cur, edz_cur, W, 0, W, false, ws.c(p.o_h[i - 1]), ws.i(p.e_h[i - 1]), W, 0, W, 0, st));
      RC(dw_reduce(rq, mt, (size_t)W * p.k_tr[i], gp + p.w_tr[i]));
    }
    if (i == 0) break;
    bsp::KcArgs g;
    g.A = dz_cur; g.EA = edz_cur; g.lda = W; g.Ka = W; weights(g, p, pk, p.wj_tt[i]);
    g.I = P; g.J = W; g.K = W; g.C = dz_nxt; g.EC = edz_nxt; g.ldc = W;
    dact(g, p.o_c[i - 1], p.o_h[i - 1], p.e_h[i - 1], W, 0, (i - 1 == 0) ? 30.f : 1.f);
    cs_ = colsum(g, W);
    RC(launch_kc(g));
    RC(bias_from_colsum(p, rq, cs_, W, gp + p.b_tr[i - 1]));
    char* t = dz_cur; dz_cur = dz_nxt; dz_nxt = t;
    int* te = edz_cur; edz_cur = edz_nxt; edz_nxt = te;
  }
  if (rq.over) { set_error("reduction arena too small (plan / pass mismatch)"); return SNERF_ERR_WORKSPACE; }
  return launch_reductions(rq.elem, rq.col, st);
}

// weight operand packs of the default arithmetic: table of jobs for bsp::launch_wpack (offsets into the fp32 region)
void build_wjobs(const Plan& p, bsp::WPackTable& tb) {
  tb.n = p.n_wjobs;
  auto set = [&](int j, size_t src_off, int src_ld, int transposed, int m_rows, int m_cols) {
    bsp::WPackJob& w = tb.j[j];
    w.src_off = src_off; w.src_ld = src_ld; w.rows = p.wj_rows[j]; w.K = p.wj_K[j]; w.transposed = transposed;
    w.dst_off = p.wj_off[j]; w.e_idx = p.wj_e[j]; w.m_rows = m_rows; w.m_cols = m_cols;
  };
  const int W = p.W, H = p.H;
  for (int i = 0; i < p.L; ++i) {
    set(p.wj_tr[i], p.w_tr[i], p.k_tr[i], 0, W, p.k_tr[i]);
    if (i > 0) set(p.wj_tt[i], p.w_tr[i] + (((p.skip_mask >> i) & 1u) ? p.Ep : 0), p.k_tr[i], 1, W, W);
  }
  set(p.wj_fs, p.w_fs, W, 0, W, W);
  set(p.wj_sig, p.w_fs + (size_t)W * W, W, 0, NARROW, W);
  set(p.wj_tfs, p.w_fs, W, 1, W + NARROW, W);
  set(p.wj_h1, p.w_h1, p.FA, 0, p.N1, p.FA);
  set(p.wj_th1, p.w_h1, p.FA, 1, p.N1, p.FA);
  set(p.wj_s2, p.w_s2, H, 0, H, H); set(p.wj_ts2, p.w_s2, H, 1, H, H);
  set(p.wj_s3, p.w_s3, H, 0, H, H); set(p.wj_ts3, p.w_s3, H, 1, H, H);
  set(p.wj_s4, p.w_s4, H, 0, NARROW, H); set(p.wj_ts4, p.w_s4, H, 1, NARROW, H);
  set(p.wj_fin, p.w_fin, p.KF, 0, NARROW, p.KF); set(p.wj_tfin, p.w_fin, p.KF, 1, NARROW, p.KF);
}

}  // namespace snerf
