// C-ABI of libsnerf_hip.so (include/snerf_hip.h): layout plan, parameter packing and the
// forward / backward launch sequences of one rendering pass.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "aux_kernels.h"
#include "bsp.h"
#include "composite.h"
#include "gemm.h"
#include "plan.h"

namespace snerf {

// bsp_pass.hip: the pass sequences of the default arithmetic (block-scaled fp16-plane activations)
int forward_bsp(const Plan& p, const float* pk, const SnerfInputs* in, const SnerfOutputs* out, void* workspace, hipStream_t st);
int backward_bsp(const Plan& p, const float* pk, const SnerfInputs* in, const SnerfOutGrads* go, float* gp, float* d_t, float* d_t_s,
                 void* workspace, hipStream_t st);
void build_wjobs(const Plan& p, bsp::WPackTable& tb);

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// ---------------------------------------------------------------------------------------------------
// plan
// ---------------------------------------------------------------------------------------------------
// split-K choice for the plane dW kernel: 256 x 256 tiles, one workgroup per CU -> tiles x splits ~ 256 (ONE round of
// the chip: every workgroup writes a 256 KB slab, so the slab traffic of a launch is 64 MB however the matrix is shaped;
// two rounds doubled it and the reduction that follows), splits of whole 128-point exponent blocks, <= 16384 points each
DwSplit dw_choose_bsp(int P, int rows, int cols, bool narrow_rows) {
  const int tiles = (narrow_rows ? 1 : (rows + 255) / 256) * ((cols + 255) / 256);
  int ns = (256 + tiles / 2) / tiles;
  const int ns_max = P / 1024 > 1 ? P / 1024 : 1;
  if (ns > ns_max) ns = ns_max;
  if (ns < 1) ns = 1;
  DwSplit d;
  d.k_split = round_up((P + ns - 1) / ns, 128);
  if (d.k_split > 16384) d.k_split = 16384;
  d.ns = (P + d.k_split - 1) / d.k_split;
  return d;
}

// ---- block-scaled plane layout: weight operand packs + workspace -------------------------------------------
static void plan_bsp(Plan& p) {
  // weight operands: every K-contiguous GEMM's B, as a WF16 pack behind the fp32 region (csrc/bsp.h)
  size_t off = 0;
  int n = 0, ne = 0;
  auto job = [&](int rows, int K, int e) {
    p.wj_off[n] = off; p.wj_rows[n] = rows; p.wj_K[n] = K; p.wj_e[n] = e;
    off += bsp::wp16_bytes(rows, K, p.pl);
    return n++;
  };
  for (int i = 0; i < p.L; ++i) {
    const int e = ne++;
    p.wj_tr[i] = job(p.W, p.k_tr[i], e);
    p.wj_tt[i] = i > 0 ? job(p.W, p.W, e) : -1;
  }
  { const int e = ne++; p.wj_fs = job(p.W, p.W, e); p.wj_sig = job(NARROW, p.W, e); p.wj_tfs = job(p.W, p.W + NARROW, e); }
  { const int e = ne++; p.wj_h1 = job(p.N1, p.FA, e); p.wj_th1 = job(p.FA, p.N1, e); }
  { const int e = ne++; p.wj_s2 = job(p.H, p.H, e); p.wj_ts2 = job(p.H, p.H, e); }
  { const int e = ne++; p.wj_s3 = job(p.H, p.H, e); p.wj_ts3 = job(p.H, p.H, e); }
  { const int e = ne++; p.wj_s4 = job(NARROW, p.H, e); p.wj_ts4 = job(p.H, NARROW, e); }
  { const int e = ne++; p.wj_fin = job(NARROW, p.KF, e); p.wj_tfin = job(p.KF, NARROW, e); }
  p.n_wjobs = n;
  p.wp_bytes = round_up_sz(off, 256);
  p.packed_floats = p.n_fp32 + (p.wp_bytes + 2 * Plan::WJ_MAX * 4 + 256) / 4;

  // workspace: activations as planes (4 bytes per element, like fp32; 2 with one plane) + exponent tables + sign words
  size_t wo = 0;
  auto wtake = [&](size_t bytes) { size_t o = wo; wo += round_up_sz(bytes, 256); return o; };
  const size_t Pp = p.Pp;
  auto planes = [&](int ld) { return wtake(bsp::plane_bytes(Pp, ld, p.pl)); };
  auto etab = [&](int ld) { return wtake(bsp::etab_ints(Pp, ld) * 4); };
  auto signs = [&](int ld) { return wtake(bsp::sign_words(Pp, ld) * 4); };
  const bool keep_c = p.train && p.siren;
  p.h1w = p.sc ? p.H : p.N1;
  p.o_z = wtake(((size_t)p.P + 4) * 4);
  p.o_T = wtake((size_t)p.P * 4);
  p.o_rgbraw = wtake((size_t)p.N * 3 * 4);
  p.o_pe = planes(p.Ep); p.e_pe = etab(p.Ep);
  if (p.train) {
    for (int i = 0; i < p.L; ++i) { p.o_h[i] = planes(p.W); p.e_h[i] = etab(p.W); p.o_c[i] = keep_c ? signs(p.W) : 0; }
  } else {
    const size_t a = planes(p.W), b = planes(p.W), ea = etab(p.W), eb = etab(p.W);
    for (int i = 0; i < p.L; ++i) { p.o_h[i] = (i & 1) ? b : a; p.e_h[i] = (i & 1) ? eb : ea; p.o_c[i] = 0; }
  }
  p.o_fa = planes(p.FA); p.e_fa = etab(p.FA);
  p.o_h1 = planes(p.h1w); p.e_h1 = etab(p.h1w); p.o_c1 = keep_c ? signs(p.h1w) : 0;
  p.o_s2 = planes(p.H); p.e_s2 = etab(p.H); p.o_cs2 = keep_c ? signs(p.H) : 0;
  p.o_s3 = planes(p.H); p.e_s3 = etab(p.H); p.o_cs3 = keep_c ? signs(p.H) : 0;
  // the narrow projections: folded into the producing SIREN launch's epilogue where its width is whole 256-column tiles (partial sums
  // per wave, summed by the composite), otherwise a 32-wide fp32 buffer written by a launch of their own
  p.nd_sig = p.siren && p.W % 256 == 0 && p.W <= 1024;
  p.nd_sun = p.siren && p.H % 256 == 0 && p.H <= 1024;
  p.nd_fin = !p.sc && p.siren && p.H == 256 && p.C <= ND_FIN;   // one 256-column tile per head block, at most ND_FIN outputs per block
  {  // bsp_trunk.hip: the shapes it is written for (launch_trunk checks them again)
    bool ok = p.pl == 1 && p.siren && p.nd_sig && p.W == 512 && p.Ep == 64 && p.L >= 3 && p.L <= bsp::TR_MAXL &&
              !(p.skip_mask & 1u) && (p.skip_mask >> (p.L - 1)) == 0u;
    for (int i = 0; ok && i < p.L; ++i) ok = p.k_tr[i] == (i == 0 ? 64 : (((p.skip_mask >> i) & 1u) ? 576 : 512));
    p.fuse_trunk = ok;
  }
  if (p.nd_sig) p.o_sigpart = wtake((size_t)4 * (p.W / 256) * Pp * 4); else p.o_sigo = wtake(Pp * NARROW * 4);
  if (p.nd_sun) p.o_sunpart = wtake((size_t)4 * (p.H / 256) * Pp * 4); else p.o_suno = wtake(Pp * NARROW * 4);
  if (p.nd_fin) p.o_finpart = wtake((size_t)4 * (p.KF / 256) * ND_FIN * Pp * 4); else if (!p.sc) p.o_fino = wtake(Pp * NARROW * 4);
  p.o_kcq = wtake((size_t)KCQ_SLOTS * 64);
  p.maxw = p.W > p.FA ? p.W : p.FA;
  if (p.h1w > p.maxw) p.maxw = p.h1w;
  p.nrb = (p.P + 31) / 32;
  p.comp_blocks = composite_bwd_blocks(p.N);
  if (p.train) {
    p.o_dza = planes(p.maxw); p.e_dza = etab(p.maxw);
    p.o_dzb = planes(p.maxw); p.e_dzb = etab(p.maxw);
    p.o_dsa = planes(p.H); p.e_dsa = etab(p.H);
    p.o_dsb = planes(p.H); p.e_dsb = etab(p.H);
    p.o_dsig = wtake(Pp * NARROW * 4); p.o_dfin = wtake(Pp * NARROW * 4); p.o_dsun = wtake(Pp * NARROW * 4);
    p.o_pdsig = planes(NARROW); p.e_dsig = etab(NARROW);
    p.o_pdfin = planes(NARROW); p.e_dfin = etab(NARROW);
    p.o_pdsun = planes(NARROW); p.e_dsun = etab(NARROW);
    p.rq_floats = bsp_rq_floats(p);
    p.o_rq = wtake(p.rq_floats * 4);
    p.o_skyslab = wtake((size_t)p.comp_blocks * 4 * p.sky_floats * 4);
  }
  p.ws_bytes = wo;
}

int make_plan(const SnerfDesc* d, Plan* pl) {
  if (!d || !pl) { set_error("null descriptor"); return SNERF_ERR_NULL; }
  Plan& p = *pl;
  auto bad = [&](const char* why) { set_error("bad SnerfDesc: %s", why); return SNERF_ERR_BAD_DESC; };
  if (d->n_rays <= 0 || d->n_samples <= 0) return bad("n_rays and n_samples must be positive");
  if ((long long)d->n_rays * d->n_samples > (1ll << 30)) return bad("n_rays * n_samples too large for one pass (chunk the rays)");
  if (d->fc_layers < 1 || d->fc_layers > SNERF_MAX_LAYERS) return bad("fc_layers out of range");
  if (d->fc_units < 16 || (d->fc_units & 15)) return bad("fc_units must be a positive multiple of 16");
  if (d->feat_last < 8 || (d->feat_last & 7) || d->feat_last > 64 * MAX_SKY_UNITS) return bad("feat_last must be a multiple of 8, <= 512");
  if (d->n_freq < 0 || d->n_freq > 16) return bad("n_freq out of range");
  if (d->t_dim < 1 || d->t_dim > 16) return bad("t_dim out of range");
  if (d->n_classes < 0 || d->n_classes > MAX_CLASSES) return bad("n_classes out of range");
  if (d->skip_mask & 1u) return bad("layer 0 cannot be a skip layer");
  p.N = d->n_rays; p.S = d->n_samples; p.P = p.N * p.S; p.Pp = round_up(p.P, 128);
  p.W = d->fc_units; p.H = d->feat_last; p.L = d->fc_layers; p.F = d->n_freq;
  {  // arithmetic: none of the bits = the default (SNERF_FLAG_F16X2), for C and Python callers alike
    const unsigned sel = d->flags & (SNERF_FLAG_F16X2 | SNERF_FLAG_F16X1);
    if (sel == (SNERF_FLAG_F16X2 | SNERF_FLAG_F16X1)) return bad("more than one arithmetic flag (SNERF_FLAG_F16X2 / SNERF_FLAG_F16X1)");
    p.pl = (sel & SNERF_FLAG_F16X1) ? 1 : 2;
  }
  p.E = p.F > 0 ? 6 * p.F : 3; p.Ep = round_up(p.E, p.pl == 2 ? 32 : 64);  // LDS stages (128 bytes per row: 32 / 64 k) never straddle the [gamma | h] segments
  p.tau = d->t_dim; p.C = d->n_classes;
  p.siren = d->siren != 0; p.sem_sigmoid = d->sem_sigmoid != 0;
  p.train = (d->flags & SNERF_FLAG_TRAIN) != 0; p.sc = (d->flags & SNERF_FLAG_SC_PASS) != 0;
  p.skip_mask = d->skip_mask;
  const bool sem = p.C > 0;
  const bool sbeta = sem && d->use_separate_beta_for_s;
  const bool sep_ts = sem && d->use_separate_tj_for_semantic;
  p.rgb_t = sem && d->use_tj_instead_of_beta;
  p.sem_t = sem && d->use_tj_for_s && !sep_ts;
  p.sem_ts = sem && d->use_tj_for_s && sep_ts;
  p.sbeta_ts = sbeta && sep_ts;
  p.x_sun = 0; p.x_t = 3; p.x_ts = sep_ts ? 3 + p.tau : -1;
  p.Xp = round_up(3 + p.tau + (sep_ts ? p.tau : 0), 16);  // FA % 16 == 0: weight planes are stored in 16-k tiles
  if ((p.H & 15) || (p.W & 31) || p.Xp != 16) return bad("fc_units % 32 == 0, feat_last % 16 == 0 and 3 + t_dim (x2 with separate t_s) <= 16 are required");
  if (p.pl == 1 && ((p.W & 63) || (p.H & 31))) return bad("SNERF_FLAG_F16X1 needs fc_units % 64 == 0 and feat_last % 32 == 0 (LDS stages of 64 columns)");
  p.Wf = round_up(p.W, 128);         // extras block on an exponent-block boundary
  p.FA = p.Wf + p.Xp;
  int nb = 0;
  p.blk_rgb = nb++;
  p.blk_sem = sem ? nb++ : -1;
  p.blk_beta = nb++;
  p.blk_sbeta = sbeta ? nb++ : -1;
  p.blk_sun = nb++;
  p.nblk = nb;
  p.KF = (nb - 1) * p.H;
  p.KF = round_up(p.KF, 128);           // sun block on an exponent-block boundary (pad rows / columns are zero)
  p.sun_col = p.KF;
  p.N1 = p.KF + p.H;

  size_t off = 0;
  auto take = [&](size_t n) { size_t o = off; off += round_up_sz(n, 64); return o; };
  for (int i = 0; i < p.L; ++i) {
    p.k_tr[i] = (i == 0) ? p.Ep : (((p.skip_mask >> i) & 1u) ? p.Ep + p.W : p.W);
    p.w_tr[i] = take((size_t)p.W * p.k_tr[i]);
    p.b_tr[i] = take(p.W);
  }
  p.w_fs = take((size_t)(p.W + NARROW) * p.W); p.b_fs = take(p.W + NARROW);
  p.w_h1 = take((size_t)p.N1 * p.FA); p.b_h1 = take(p.N1);
  p.w_s2 = take((size_t)p.H * p.H); p.b_s2 = take(p.H);
  p.w_s3 = take((size_t)p.H * p.H); p.b_s3 = take(p.H);
  p.w_s4 = take((size_t)NARROW * p.H); p.b_s4 = take(NARROW);
  p.w_fin = take((size_t)NARROW * p.KF); p.b_fin = take(NARROW);
  p.sky_floats = 9 * p.H + 4;
  p.sky = take(p.sky_floats);
  p.n_fp32 = off;
  plan_bsp(p);
  return SNERF_OK;
}

// ---------------------------------------------------------------------------------------------------
// pack / unpack tables
// ---------------------------------------------------------------------------------------------------
struct TableBuilder {
  CopyTable tabs[4];
  int nt = 0;
  bool missing = false;
  TableBuilder() { memset(tabs, 0, sizeof(tabs)); nt = 1; }
  void add(float* user, int user_ld, int rows, int cols, size_t dst_off, int dst_ld) {
    if (!user) { missing = true; return; }
    if (tabs[nt - 1].n == COPY_TABLE_MAX) ++nt;
    CopyEntry& e = tabs[nt - 1].e[tabs[nt - 1].n++];
    e.user = user; e.user_ld = user_ld; e.rows = rows; e.cols = cols; e.dst_off = dst_off; e.dst_ld = dst_ld;
  }
};

static void build_tables(const Plan& p, const SnerfParams* w, TableBuilder& tb) {
  const int W = p.W, H = p.H, E = p.E;
  for (int i = 0; i < p.L; ++i) {
    const bool skip = (p.skip_mask >> i) & 1u;
    if (i == 0) {
      tb.add(w->fc_w[0], E, W, E, p.w_tr[0], p.k_tr[0]);
    } else if (skip) {  // [gamma | h] columns: gamma -> [0,E), h -> [Ep, Ep+W)
      tb.add(w->fc_w[i], E + W, W, E, p.w_tr[i], p.k_tr[i]);
      tb.add(w->fc_w[i] ? w->fc_w[i] + E : nullptr, E + W, W, W, p.w_tr[i] + p.Ep, p.k_tr[i]);
    } else {
      tb.add(w->fc_w[i], W, W, W, p.w_tr[i], p.k_tr[i]);
    }
    tb.add(w->fc_b[i], W, 1, W, p.b_tr[i], W);
  }
  tb.add(w->feats_w, W, W, W, p.w_fs, W);
  tb.add(w->sigma_w, W, 1, W, p.w_fs + (size_t)W * W, W);
  tb.add(w->feats_b, W, 1, W, p.b_fs, W);
  tb.add(w->sigma_b, 1, 1, 1, p.b_fs + W, 1);
  auto head1 = [&](int blk, float* w0, float* b0, int extra, int xcol) {
    const int in = W + extra;
    const size_t r0 = blk == p.blk_sun ? (size_t)p.sun_col : (size_t)blk * H;
    const size_t row0 = p.w_h1 + r0 * p.FA;
    tb.add(w0, in, H, W, row0, p.FA);
    if (extra > 0) tb.add(w0 ? w0 + W : nullptr, in, H, extra, row0 + p.Wf + xcol, p.FA);
    tb.add(b0, H, 1, H, p.b_h1 + r0, H);
  };
  head1(p.blk_rgb, w->rgb_w0, w->rgb_b0, p.rgb_t ? p.tau : 0, p.x_t);
  if (p.blk_sem >= 0) head1(p.blk_sem, w->sem_w0, w->sem_b0, (p.sem_t || p.sem_ts) ? p.tau : 0, p.sem_ts ? p.x_ts : p.x_t);
  head1(p.blk_beta, w->beta_w0, w->beta_b0, p.tau, p.x_t);
  if (p.blk_sbeta >= 0) head1(p.blk_sbeta, w->sbeta_w0, w->sbeta_b0, p.tau, p.sbeta_ts ? p.x_ts : p.x_t);
  head1(p.blk_sun, w->sun_w[0], w->sun_b[0], 3, p.x_sun);
  tb.add(w->sun_w[1], H, H, H, p.w_s2, H); tb.add(w->sun_b[1], H, 1, H, p.b_s2, H);
  tb.add(w->sun_w[2], H, H, H, p.w_s3, H); tb.add(w->sun_b[2], H, 1, H, p.b_s3, H);
  tb.add(w->sun_w[3], H, 1, H, p.w_s4, H); tb.add(w->sun_b[3], 1, 1, 1, p.b_s4, 1);
  auto fin = [&](int blk, int col, int rows, float* w2, float* b2) {
    tb.add(w2, H, rows, H, p.w_fin + (size_t)col * p.KF + (size_t)blk * H, p.KF);
    tb.add(b2, rows, 1, rows, p.b_fin + col, rows);
  };
  fin(p.blk_rgb, Plan::col_rgb, 3, w->rgb_w2, w->rgb_b2);
  fin(p.blk_beta, Plan::col_beta, 1, w->beta_w2, w->beta_b2);
  if (p.blk_sbeta >= 0) fin(p.blk_sbeta, Plan::col_sbeta, 1, w->sbeta_w2, w->sbeta_b2);
  if (p.blk_sem >= 0) fin(p.blk_sem, Plan::col_sem, p.C, w->sem_w2, w->sem_b2);
  tb.add(w->sky_w0, 3, H, 3, p.sky, 4);
  tb.add(w->sky_b0, H, 1, H, p.sky + 4 * (size_t)H, H);
  tb.add(w->sky_w2, H, 3, H, p.sky + 5 * (size_t)H, H);
  tb.add(w->sky_b2, 3, 1, 3, p.sky + 9 * (size_t)H, 3);
}

#define RC(expr) do { int _rc = (expr); if (_rc) return _rc; } while (0)

static int check_inputs(const Plan& p, const SnerfInputs* in) {
  if (!in) { set_error("null inputs"); return SNERF_ERR_NULL; }
  if (!in->sun_d || in->sun_stride < 3) { set_error("sun_d (N,3) with stride >= 3 is required"); return SNERF_ERR_NULL; }
  if (!in->t) { set_error("t (N,tau) is required"); return SNERF_ERR_NULL; }
  if (p.x_ts >= 0 && !in->t_s) { set_error("t_s is required with use_separate_tj_for_semantic"); return SNERF_ERR_NULL; }
  if (in->xyz) {
    if (!in->z_vals) { set_error("explicit xyz needs explicit z_vals"); return SNERF_ERR_NULL; }
  } else {
    if (!in->rays) { set_error("rays or xyz is required"); return SNERF_ERR_NULL; }
    if (!in->z_vals && !in->z_steps) { set_error("z_steps is required when z_vals is not given"); return SNERF_ERR_NULL; }
  }
  return SNERF_OK;
}

}  // namespace snerf

// =====================================================================================================
using namespace snerf;

namespace {
struct DevBuf {
  void* p = nullptr;
  ~DevBuf() { if (p) (void)hipFree(p); }
  int alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 16) == hipSuccess ? 0 : 1; }
  template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};
#define TALLOC(buf, bytes) do { if ((buf).alloc(bytes)) { set_error("test hook: hipMalloc failed"); return SNERF_ERR_HIP; } } while (0)
}  // namespace

extern "C" {

int snerf_version(void) { return SNERF_ABI_VERSION; }
const char* snerf_last_error(void) { return g_err; }

size_t snerf_packed_floats(const SnerfDesc* desc) {
  Plan p;
  if (make_plan(desc, &p)) return 0;
  return p.packed_floats;
}

size_t snerf_grad_floats(const SnerfDesc* desc) {
  Plan p;
  if (make_plan(desc, &p)) return 0;
  return p.n_fp32;
}

size_t snerf_workspace_bytes(const SnerfDesc* desc) {
  Plan p;
  if (make_plan(desc, &p)) return 0;
  return p.ws_bytes;
}

int snerf_pack_params(const SnerfDesc* desc, const SnerfParams* params, float* packed, void* stream) {
  Plan p;
  RC(make_plan(desc, &p));
  if (!params || !packed) { set_error("null params/packed"); return SNERF_ERR_NULL; }
  TableBuilder tb;
  build_tables(p, params, tb);
  if (tb.missing) { set_error("snerf_pack_params: a parameter tensor required by this SnerfDesc is NULL"); return SNERF_ERR_NULL; }
  hipStream_t st = (hipStream_t)stream;
  RC(launch_zero_bytes(packed, p.n_fp32 * sizeof(float), st));
  for (int i = 0; i < tb.nt; ++i) RC(launch_copy_table(tb.tabs[i], packed, 0, st));
  // every weight operand (matrix or its transpose) as a fragment-ordered fp16-plane pack with one exponent per matrix -- two
  // launches over a job table (|max| pass, pack pass)
  bsp::WPackTable wt;
  build_wjobs(p, wt);
  char* planes = reinterpret_cast<char*>(packed + p.n_fp32);
  int* exps = reinterpret_cast<int*>(planes + p.wp_bytes);
  return bsp::launch_wpack(wt, packed, planes, exps, reinterpret_cast<unsigned*>(exps + Plan::WJ_MAX), p.pl, st);
}

int snerf_unpack_grads(const SnerfDesc* desc, const float* packed_grads, const SnerfParams* grads, int accumulate,
                       void* stream) {
  Plan p;
  RC(make_plan(desc, &p));
  if (!grads || !packed_grads) { set_error("null grads"); return SNERF_ERR_NULL; }
  TableBuilder tb;
  build_tables(p, grads, tb);
  if (tb.missing) { set_error("snerf_unpack_grads: a gradient tensor required by this SnerfDesc is NULL"); return SNERF_ERR_NULL; }
  hipStream_t st = (hipStream_t)stream;
  for (int i = 0; i < tb.nt; ++i) RC(launch_copy_table(tb.tabs[i], const_cast<float*>(packed_grads), accumulate ? 2 : 1, st));
  return SNERF_OK;
}

int snerf_forward(const SnerfDesc* desc, const float* packed_params, const SnerfInputs* in, const SnerfOutputs* out,
                  void* workspace, size_t workspace_bytes, void* stream) {
  Plan p;
  RC(make_plan(desc, &p));
  if (!packed_params || !out || !workspace) { set_error("snerf_forward: null argument"); return SNERF_ERR_NULL; }
  if (workspace_bytes < p.ws_bytes) { set_error("snerf_forward: workspace too small (%zu < %zu)", workspace_bytes, p.ws_bytes); return SNERF_ERR_WORKSPACE; }
  if (((uintptr_t)workspace & 255) || ((uintptr_t)packed_params & 255)) { set_error("workspace and packed params must be 256-byte aligned"); return SNERF_ERR_WORKSPACE; }
  RC(check_inputs(p, in));
  return forward_bsp(p, packed_params, in, out, workspace, (hipStream_t)stream);
}

int snerf_backward(const SnerfDesc* desc, const float* packed_params, const SnerfInputs* in, const SnerfOutGrads* gout,
                   float* packed_grads, float* d_t, float* d_t_s, void* workspace, size_t workspace_bytes, void* stream) {
  Plan p;
  RC(make_plan(desc, &p));
  if (!p.train) { set_error("snerf_backward needs the SnerfDesc used for the SNERF_FLAG_TRAIN forward"); return SNERF_ERR_BAD_DESC; }
  if (!packed_params || !gout || !packed_grads || !workspace) { set_error("snerf_backward: null argument"); return SNERF_ERR_NULL; }
  if (workspace_bytes < p.ws_bytes) { set_error("snerf_backward: workspace too small (%zu < %zu)", workspace_bytes, p.ws_bytes); return SNERF_ERR_WORKSPACE; }
  if (((uintptr_t)workspace & 255) || ((uintptr_t)packed_params & 255) || ((uintptr_t)packed_grads & 255)) { set_error("workspace and packed buffers must be 256-byte aligned"); return SNERF_ERR_WORKSPACE; }
  RC(check_inputs(p, in));
  return backward_bsp(p, packed_params, in, gout, packed_grads, d_t, d_t_s, workspace, (hipStream_t)stream);
}

int snerf_sample_z(const float* rays, const float* z_steps, const float* u, float* z, int n_rays, int n_samples, void* stream) {
  if (!rays || !z_steps || !z || n_rays <= 0 || n_samples <= 0) { set_error("snerf_sample_z: bad argument"); return SNERF_ERR_NULL; }
  return launch_sample_z(rays, z_steps, u, z, n_rays, n_samples, (hipStream_t)stream);
}

int snerf_embedding_rows(const float* table, int n_embed, int tau, const long long* idx, int n, float* rows, void* stream) {
  if (!table || !idx || !rows || n_embed <= 0 || tau <= 0 || n <= 0) { set_error("snerf_embedding_rows: bad argument"); return SNERF_ERR_NULL; }
  return launch_embedding_rows(table, n_embed, tau, idx, n, rows, (hipStream_t)stream);
}

int snerf_embedding_backward(const long long* idx, const float* d_rows, int n, int tau, int n_embed, float* grad_table, void* stream) {
  if (!idx || !d_rows || !grad_table || n_embed <= 0 || tau <= 0 || n <= 0) { set_error("snerf_embedding_backward: bad argument"); return SNERF_ERR_NULL; }
  return launch_embedding_backward(idx, d_rows, n, tau, n_embed, grad_table, (hipStream_t)stream);
}

int snerf_profile_begin(void) { return profile_begin(); }
int snerf_profile_end(SnerfProfile* out) { return profile_end(out); }

// ---- test hooks of the block-scaled plane kernels (tests/test_gpu_bsp.py): fp32 in / fp32 out around ONE launch of the
// kernel under test; the conversions run through the library's own to_planes / from_planes / weight pack.  Synchronous,
// allocating -- never on the product path.

int snerf_test_bsp_roundtrip(const float* src, int rows, int cols, int ld, int col0, float* dst, int* exps_out, int planes, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  DevBuf pl, E;
  const size_t rp = round_up_sz(rows, 128);
  TALLOC(pl, bsp::plane_bytes(rp, ld, planes)); TALLOC(E, bsp::etab_ints(rp, ld) * 4);
  SNERF_HIP_CHECK(hipMemsetAsync(E.p, 0, bsp::etab_ints(rp, ld) * 4, st));
  RC(bsp::launch_to_planes(src, cols, rows, cols, pl.as<char>(), E.as<int>(), ld, col0, planes, st));
  RC(bsp::launch_from_planes(pl.as<char>(), E.as<int>(), ld, col0, rows, cols, dst, cols, planes, st));
  if (exps_out) SNERF_HIP_CHECK(hipMemcpyAsync(exps_out, E.p, bsp::etab_ints(rp, ld) * 4, hipMemcpyDeviceToDevice, st));
  SNERF_HIP_CHECK(hipStreamSynchronize(st));
  return SNERF_OK;
}

// persistent grid of the K-contiguous launches: n workgroups instead of two per CU (0: default) -- small test problems then walk
// several tiles per workgroup and draw them from the tile counters
int snerf_test_set_kc_grid(int n) { bsp::kc_set_grid_override(n); bsp::trunk_set_grid_override(n); return SNERF_OK; }
// 0: every pass takes the launch-per-layer path (the fused trunk of bsp_trunk.hip is compared with it bit for bit); 1: default
int snerf_test_set_trunk_fusion(int on) { bsp::trunk_set_fusion(on); return SNERF_OK; }

// C[I][J] = epilogue(A[I][Ka] | A2[I][K-Ka]) . W[J][K]^T).  The A tensors are placed at column a_col0 of wider plane
// tensors and the output at column c_col0 (exercises the column-offset / exponent-block arithmetic).
int snerf_test_bsp_kc(const float* A, const float* A2, int Ka, const float* W, const float* bias, int I, int J, int K, int a_col0,
                      int c_col0, int act, float w0, int aux_mode, const float* Hact, const unsigned* Hsign, float* C,
                      unsigned* Csign, float* colsum, const float* nd_w, float* nd_out, const int* nd_rows, int narrow, int planes, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (K % 16 || Ka % 16 || Ka <= 0 || Ka > K) { set_error("test_bsp_kc: K, Ka % 16"); return SNERF_ERR_BAD_DESC; }
  if (planes != 1 && planes != 2) { set_error("test_bsp_kc: planes"); return SNERF_ERR_BAD_DESC; }
  const int pl = planes;
  const size_t rp = round_up_sz(I, 128);
  const int lda = a_col0 + Ka, lda2 = K > Ka ? K - Ka : 16, ldc = c_col0 + round_up(J, 16);
  DevBuf pa, ea, pa2, ea2, wp, we, wm, pc, ec, ph, eh;
  // (+ 64 pad columns: a one-plane launch whose K is an odd multiple of 16 reads -- against zero weights -- up to 16 columns
  //  beyond K; in the passes those are the next row's columns, here they must not be uninitialised memory)
  TALLOC(pa, bsp::plane_bytes(rp + 1, lda, pl)); TALLOC(ea, bsp::etab_ints(rp, lda) * 4);
  TALLOC(pa2, bsp::plane_bytes(rp + 1, lda2, pl)); TALLOC(ea2, bsp::etab_ints(rp, lda2) * 4);
  SNERF_HIP_CHECK(hipMemsetAsync(pa.p, 0, bsp::plane_bytes(rp + 1, lda, pl), st));
  SNERF_HIP_CHECK(hipMemsetAsync(pa2.p, 0, bsp::plane_bytes(rp + 1, lda2, pl), st));
  RC(bsp::launch_to_planes(A, Ka, I, Ka, pa.as<char>(), ea.as<int>(), lda, a_col0 & ~127, pl, st));
  if (a_col0 & 127) { set_error("test_bsp_kc: a_col0 % 128"); return SNERF_ERR_BAD_DESC; }
  if (K > Ka) RC(bsp::launch_to_planes(A2, K - Ka, I, K - Ka, pa2.as<char>(), ea2.as<int>(), lda2, 0, pl, st));
  bsp::WPackTable tb; tb.n = 1;
  tb.j[0] = bsp::WPackJob{0ull, K, J, K, 0, 0ull, 0, J, K};
  TALLOC(wp, bsp::wp16_bytes(J, K, pl)); TALLOC(we, bsp::WPACK_MAX * 4); TALLOC(wm, bsp::WPACK_MAX * 4);
  RC(bsp::launch_wpack(tb, W, wp.as<char>(), we.as<int>(), wm.as<unsigned>(), pl, st));
  bsp::KcArgs g;
  g.pl = pl;
  g.A = pa.as<char>(); g.EA = ea.as<int>(); g.lda = lda; g.a_col0 = a_col0; g.Ka = Ka;
  if (K > Ka) { g.A2 = pa2.as<char>(); g.EA2 = ea2.as<int>(); g.lda2 = lda2; g.a2_col0 = 0; }
  g.W = wp.as<char>(); g.EW = we.as<int>(); g.w_rb32 = (J + 31) / 32; g.w_bytes = (unsigned)bsp::wp16_bytes(J, K, pl);
  g.I = I; g.J = J; g.K = K; g.bias = bias; g.act = act; g.w0 = w0;
  if (narrow) {
    g.Cf = C;
    RC(bsp::launch_kc_narrow(g, st));
    SNERF_HIP_CHECK(hipStreamSynchronize(st));
    return SNERF_OK;
  }
  TALLOC(pc, bsp::plane_bytes(rp, ldc, pl)); TALLOC(ec, bsp::etab_ints(rp, ldc) * 4);
  g.C = pc.as<char>(); g.EC = ec.as<int>(); g.ldc = ldc; g.c_col0 = c_col0; g.Csign = Csign;
  if (aux_mode != AUX_NONE) {
    TALLOC(ph, bsp::plane_bytes(rp, ldc, pl)); TALLOC(eh, bsp::etab_ints(rp, ldc) * 4);
    RC(bsp::launch_to_planes(Hact, J, I, J, ph.as<char>(), eh.as<int>(), ldc, c_col0, pl, st));
    g.aux_mode = aux_mode; g.H = ph.as<char>(); g.EH = eh.as<int>(); g.ldh = ldc; g.h_col0 = c_col0; g.Hsign = Hsign;
  }
  g.colsum = colsum; g.ldcs = J;
  g.nd_w = nd_w; g.nd_out = nd_out; g.nd_stride = (unsigned long long)I;
  if (nd_w && nd_rows) {   // several projections per column tile: nd_w [sum rows][J], tile tj's rows follow tile tj - 1's
    g.nd_omax = ND_FIN; g.nd_ldw = J;
    for (int tj = 0, r = 0; tj < (J + 255) / 256 && tj < 8; ++tj) { g.nd_rows[tj] = nd_rows[tj]; g.nd_row0[tj] = r; r += nd_rows[tj]; }
  }
  DevBuf ctr; TALLOC(ctr, 64);
  SNERF_HIP_CHECK(hipMemsetAsync(ctr.p, 0, 64, st));
  g.tile_ctr = ctr.as<int>();
  RC(bsp::launch_kc(g, st));
  RC(bsp::launch_from_planes(pc.as<char>(), ec.as<int>(), ldc, c_col0, I, J, C, J, pl, st));
  SNERF_HIP_CHECK(hipStreamSynchronize(st));
  return SNERF_OK;
}

// C[I][J] = sum_p A[p][a_col0 + i] B[p][b_col0 + j] through split-K slabs + the library's deterministic slab reduction
int snerf_test_bsp_dw(const float* A, int lda_src, const float* B, int ldb_src, int P, int I, int J, int a_col0, int b_col0,
                      int k_split, int narrow_i, float* C, int planes, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (planes != 1 && planes != 2) { set_error("test_bsp_dw: planes"); return SNERF_ERR_BAD_DESC; }
  const int pl = planes;
  const size_t rp = round_up_sz(P, 128);
  const int lda = round_up(lda_src, 16), ldb = round_up(ldb_src, 16);
  DevBuf pa, ea, pb, eb, slab, tmp;
  TALLOC(pa, bsp::plane_bytes(rp, lda, pl)); TALLOC(ea, bsp::etab_ints(rp, lda) * 4);
  TALLOC(pb, bsp::plane_bytes(rp, ldb, pl)); TALLOC(eb, bsp::etab_ints(rp, ldb) * 4);
  RC(bsp::launch_to_planes(A, lda_src, P, lda_src, pa.as<char>(), ea.as<int>(), lda, 0, pl, st));
  RC(bsp::launch_to_planes(B, ldb_src, P, ldb_src, pb.as<char>(), eb.as<int>(), ldb, 0, pl, st));
  const int ns = (P + k_split - 1) / k_split;
  const size_t stride = round_up_sz((size_t)I * J, 64);
  TALLOC(slab, stride * ns * 4); TALLOC(tmp, 64 * stride * 4);
  bsp::DwArgs g;
  g.A = pa.as<char>(); g.EA = ea.as<int>(); g.lda = lda; g.a_col0 = a_col0;
  g.B = pb.as<char>(); g.EB = eb.as<int>(); g.ldb = ldb; g.b_col0 = b_col0;
  g.I = I; g.J = J; g.P = P; g.C = slab.as<float>(); g.ldc = J; g.k_split = k_split; g.n_split = ns; g.slab_stride = stride; g.pl = pl;
  RC(bsp::launch_dw(g, narrow_i != 0, st));
  SNERF_HIP_CHECK(hipMemsetAsync(C, 0, (size_t)I * J * 4, st));
  RC(reduce_partials(slab.as<float>(), ns, stride, I * J, tmp.as<float>(), C, st));
  SNERF_HIP_CHECK(hipStreamSynchronize(st));
  return SNERF_OK;
}

}  // extern "C"
