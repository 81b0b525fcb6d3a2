/*
 * snerf_hip.h -- C-ABI of libsnerf_hip.so: the MI355X (gfx950) implementation of the semantic
 * Sat-NeRF ray-marching hot path (encode -> SIREN/ReLU MLP + heads -> irradiance alpha-composite
 * -> losses, forward and backward).
 *
 * The reference (wagnva/semantic-nerf-for-satellite-data) is pure Python/PyTorch and has no FFI of
 * its own; each entry point below names the reference interface it replaces (paths relative to the
 * reference repository).  INTEGRATION.md shows the ctypes binding a maintainer adds on the
 * reference side.
 *
 * Conventions
 *   - plain C: pointers + sizes only, no torch types; every pointer is a DEVICE pointer on the
 *     current HIP device unless stated otherwise; row-major fp32 unless stated otherwise;
 *   - return 0 on success, non-zero error code otherwise; snerf_last_error() gives the message;
 *     no exceptions cross the ABI;
 *   - no allocation and no synchronisation inside the hot calls: the caller provides the workspace
 *     (size from snerf_workspace_bytes) and a stream; calls are asynchronous on that stream
 *     (hipGraph-capturable);
 *   - one caller thread per device/stream (the reference drives everything from one Python thread,
 *     framework/pipelines.py:306-320); re-entrant across devices (one process per GPU).
 */
#ifndef SNERF_HIP_H
#define SNERF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SNERF_ABI_VERSION 5
#define SNERF_MAX_LAYERS 16

/* error codes */
#define SNERF_OK 0
#define SNERF_ERR_BAD_DESC 1
#define SNERF_ERR_WORKSPACE 2
#define SNERF_ERR_NULL 3
#define SNERF_ERR_HIP 4

/* SnerfDesc.flags */
#define SNERF_FLAG_TRAIN 1u   /* keep every activation needed by snerf_backward in the workspace */
#define SNERF_FLAG_SC_PASS 2u /* solar-correction variant: sample points on o + sun_d*z, evaluate only the
                                 trunk + sigma + sun-visibility branch and return weights/transparency/sun
                                 (semantic/components/rendering.py:59-78) */

/* Arithmetic of the dense contractions.  With NONE of the arithmetic bits set a pass runs the default, SNERF_FLAG_F16X2
 * (the same for C and Python callers); the two bits exclude each other. */
#define SNERF_FLAG_F16X2 64u   /* DEFAULT (flags = 0 means this): fp32-class arithmetic on the fp16 matrix cores.  Every activation
                                  tensor of the workspace is two fp16 planes (22 significant bits) with one power-of-two exponent
                                  per 128 x 128 block, written once by the producing kernel; a product is hi*hi + hi*lo + lo*hi on
                                  v_mfma_f32_32x32x16_f16 with fp32 accumulation; the dropped lo*lo term is 2^-22 relative, below
                                  an fp32 GEMM's own rounding (normwise).  Needs fc_units % 32 == 0, feat_last % 16 == 0 and
                                  3 + t_dim (x2 with a separate t_s) <= 16: other shapes return SNERF_ERR_BAD_DESC */
#define SNERF_FLAG_F16X1 8u    /* REDUCED precision (the reference's `precision = 16` runs, baseline/pipelines/nerf.py:65; BASELINE.json
                                  configs[2], [4]): the same block-scaled tensors with ONE fp16 plane -- 11 significant bits relative
                                  to the block's maximum, 2 bytes per element -- weights packed as one plane, one MFMA product per
                                  contraction step (a third of the default's matrix work, half its operand bytes), fp32 accumulate.
                                  Same kernels, templated on the plane count.  Judged on PSNR / mIoU, not on the 1e-4 parity bar.
                                  Needs fc_units % 64 == 0 and feat_last % 32 == 0 on top of the default's shape rules */

/* Model + batch description.  Field names follow the reference config
 * (configs/pipelines/rs_semantic.toml:13-67, semantic/pipelines/rs_semantic.py:125-141). */
typedef struct SnerfDesc {
  int32_t n_rays;      /* N */
  int32_t n_samples;   /* S (n_samples) */
  int32_t fc_units;    /* W */
  int32_t fc_layers;   /* L <= SNERF_MAX_LAYERS */
  int32_t feat_last;   /* H = W/2, or W with fc_use_full_features */
  uint32_t skip_mask;  /* bit i set <=> i in fc_skips */
  int32_t n_freq;      /* mapping_pos_n_freq; 0 = identity encoding (baseline SatNeRF, satnerf.py:140-141) */
  int32_t siren;       /* activation_function == "siren" */
  int32_t t_dim;       /* t_embedding_tau */
  int32_t n_classes;   /* semantic classes C; 0 = baseline SatNeRF without semantic head */
  int32_t sem_sigmoid; /* semantic_activation_function == "sigmoid" */
  int32_t use_tj_instead_of_beta;
  int32_t use_tj_for_s;
  int32_t use_separate_beta_for_s;
  int32_t use_separate_tj_for_semantic;
  uint32_t flags;
} SnerfDesc;

/* Raw device pointers to the parameter tensors, in the reference's state_dict layout
 * (fc_net.{2i}.{weight,bias}, sigma_from_xyz.0.*, feats_from_xyz.*, rgb_from_xyzdir.{0,2}.*,
 * semantic_prediction.{0,2}.*, sun_v_net.{0,2,4,6}.*, sky_color.{0,2}.*, beta_from_xyz.{0,2}.*,
 * semantic_beta_from_xyz.{0,2}.*).  torch Linear layout: weight (out,in) row-major contiguous, bias (out).
 * The library never owns them.  Used both for parameters (read) and for their gradients (written). */
typedef struct SnerfParams {
  float* fc_w[SNERF_MAX_LAYERS];
  float* fc_b[SNERF_MAX_LAYERS];
  float* sigma_w; float* sigma_b;
  float* feats_w; float* feats_b;
  float* rgb_w0; float* rgb_b0; float* rgb_w2; float* rgb_b2;
  float* sem_w0; float* sem_b0; float* sem_w2; float* sem_b2;       /* NULL when n_classes == 0 */
  float* sun_w[4]; float* sun_b[4];
  float* sky_w0; float* sky_b0; float* sky_w2; float* sky_b2;
  float* beta_w0; float* beta_b0; float* beta_w2; float* beta_b2;
  float* sbeta_w0; float* sbeta_b0; float* sbeta_w2; float* sbeta_b2; /* NULL unless use_separate_beta_for_s */
} SnerfParams;

/* Ray batch.  Either (rays [+ z_vals | z_steps [+ u]]) -- the renderer seam,
 * framework/components/rendering.py:84-157 -- or explicit (xyz, z_vals) -- the inference() seam,
 * semantic/models/rs_semantic.py:8-19. */
typedef struct SnerfInputs {
  const float* rays;    /* (N,8): origin 0:3, dir 3:6, near 6, far 7 (framework/components/rays.py:7-38) */
  const float* xyz;     /* (N,S,3) explicit sample positions, or NULL */
  const float* z_vals;  /* (N,S) explicit depths, or NULL => stratified sampling from rays */
  const float* z_steps; /* (S) = linspace(0,1,S) from the host; needed when z_vals == NULL */
  const float* u;       /* (N,S) uniform [0,1) jitter, or NULL => no perturbation */
  const float* sun_d;   /* (N,3) with row stride sun_stride floats (4 when pointing into extras (N,4)) */
  int32_t sun_stride;
  int32_t _pad;
  const float* t;       /* (N,tau) transient embedding rows (models["t"](ts)) */
  const float* t_s;     /* (N,tau) or NULL */
} SnerfInputs;

/* Result tensors = the dict returned by inference() (rs_semantic.py:111-128); any pointer may be NULL.
 * With SNERF_FLAG_SC_PASS only weights / transparency / sun are produced. */
typedef struct SnerfOutputs {
  float* rgb;             /* (N,3)  */
  float* depth;           /* (N)    */
  float* weights;         /* (N,S)  */
  float* transparency;    /* (N,S)  */
  float* albedo;          /* (N,S,3)*/
  float* sun;             /* (N,S,1)*/
  float* sky;             /* (N,S,3)*/
  float* beta;            /* (N,S,1)*/
  float* sigmas;          /* (N,S)  */
  float* beta_semantic;   /* (N,S,1) if use_separate_beta_for_s */
  float* semantic_logits; /* (N,C)  */
  int64_t* semantic_label;/* (N)    */
  float* z_vals;          /* (N,S) depths actually used (sampled or copied) */
} SnerfOutputs;

/* Gradients of the scalar loss w.r.t. the result tensors (same shapes; NULL = zero). */
typedef struct SnerfOutGrads {
  const float* rgb; const float* depth; const float* weights; const float* transparency;
  const float* albedo; const float* sun; const float* sky; const float* beta; const float* sigmas;
  const float* beta_semantic; const float* semantic_logits;
} SnerfOutGrads;

/* ---- library info ------------------------------------------------------------------------------ */
int snerf_version(void);
const char* snerf_last_error(void);

/* ---- sizes (host-side, no GPU work) -------------------------------------------------------------- */
/* number of floats of the packed parameter / packed gradient buffer */
size_t snerf_packed_floats(const SnerfDesc* desc);
/* floats of a packed GRADIENT buffer: the leading fp32 region of the packed layout -- all that snerf_backward accumulates
 * into and snerf_unpack_grads reads (the weight-operand packs behind it exist for parameters only) */
size_t snerf_grad_floats(const SnerfDesc* desc);
/* workspace bytes for one pass (activations + scratch) under desc->flags; the same buffer must be
 * handed to snerf_backward for that pass */
size_t snerf_workspace_bytes(const SnerfDesc* desc);

/* ---- parameter packing ------------------------------------------------------------------------- */
/* Gather the state_dict tensors into the padded, MFMA-friendly packed layout (DESIGN.md "Data layout").
 * Replaces nothing in the reference (torch.nn.Linear owns its layout there); run once per optimiser step.
 * The weight operands inside the buffer follow desc->flags' arithmetic (fragment-ordered fp16 planes + one exponent per matrix:
 * two planes in the default arithmetic, one under SNERF_FLAG_F16X1): pack, forward and backward must use the same arithmetic flag. */
int snerf_pack_params(const SnerfDesc* desc, const SnerfParams* params, float* packed, void* stream);
/* Scatter packed gradients back into tensors shaped like the parameters (overwrite, or add if accumulate). */
int snerf_unpack_grads(const SnerfDesc* desc, const float* packed_grads, const SnerfParams* grads,
                       int accumulate, void* stream);

/* ---- hot path ---------------------------------------------------------------------------------- */
/* One rendering pass: sample -> encode -> MLP -> composite.
 * Replaces BaseRenderer.render_rays/sample_rays (framework/components/rendering.py:84-157),
 * RSSemanticRendering._model_rendering (semantic/components/rendering.py:18-80; one call per pass),
 * inference + RSSemanticNeRF.forward (semantic/models/rs_semantic.py:8-128,260-340;
 * baseline/models/satnerf.py:8-98,203-255) and convert_sigmas (framework/util/rendering.py:4-34). */
int snerf_forward(const SnerfDesc* desc, const float* packed_params, const SnerfInputs* in,
                  const SnerfOutputs* out, void* workspace, size_t workspace_bytes, void* stream);

/* Stratified depths only (sample_rays, framework/components/rendering.py:95-110): z (N,S) from rays (N,8),
 * z_steps (S) and an optional jitter tensor u (N,S).  Lets a caller share one z between passes on different streams. */
int snerf_sample_z(const float* rays, const float* z_steps, const float* u, float* z, int n_rays, int n_samples, void* stream);

/* The per-ray embedding rows (nn.Embedding(50, tau) indexed by the rays' image index `ts`:
 * semantic/components/rendering.py:35-46, baseline/components/rendering.py:29-40).
 * snerf_embedding_rows: rows[n][:] = table[idx[n]][:]  (forward; idx is int64 as torch hands it over).
 * snerf_embedding_backward: grad_table[v][:] += sum over the rays n with idx[n] == v of d_rows[n][:], summed in a fixed
 * order (no atomics: the gradient is bitwise reproducible, like every other one of the path).  Indices outside
 * [0, n_embed) give zero rows in the forward and are skipped in the backward (torch's nn.Embedding asserts on the device). */
int snerf_embedding_rows(const float* table, int n_embed, int tau, const long long* idx, int n, float* rows, void* stream);
int snerf_embedding_backward(const long long* idx, const float* d_rows, int n, int tau, int n_embed, float* grad_table, void* stream);

/* Backward of one pass (what autograd does in the reference for the ops above): consumes the
 * activations that snerf_forward(SNERF_FLAG_TRAIN) left in `workspace`, ACCUMULATES parameter
 * gradients into packed_grads (caller zeroes it once per step) and writes d loss / d t (N,tau)
 * [and d t_s] when those pointers are non-NULL. */
int snerf_backward(const SnerfDesc* desc, const float* packed_params, const SnerfInputs* in,
                   const SnerfOutGrads* gout, float* packed_grads, float* d_t, float* d_t_s,
                   void* workspace, size_t workspace_bytes, void* stream);

/* ---- fused losses ---------------------------------------------------------------------------------
 * Replaces SNerfLoss / SatNerfLoss / uncertainty_aware_loss / solar_correction / DepthLoss
 * (baseline/components/loss.py:4-94), SemanticLoss / SemanticUncertaintyLoss / SemanticCarRegLoss
 * (semantic/components/loss.py:6-157) and their autograd backward.  Two phases because the means have
 * data-dependent denominators (CE over non-ignored rays, L_t over car rays): phase 1 reduces per-ray
 * sums and counts into totals[SNERF_LOSS_NTOT]; under data parallelism the host all-reduces that small
 * vector; phase 2 turns totals into the loss_dict values and d loss / d result tensors. */
#define SNERF_LOSS_NTOT 16
#define SNERF_LOSS_NTERMS 8
/* indices into terms[]: the reference's loss_dict keys */
#define SNERF_TERM_COLOR 0            /* coarse_color */
#define SNERF_TERM_LOGBETA 1          /* coarse_logbeta */
#define SNERF_TERM_SC2 2              /* coarse_sc_term2 */
#define SNERF_TERM_SC3 3              /* coarse_sc_term3 */
#define SNERF_TERM_SEMANTIC 4         /* coarse_semantic */
#define SNERF_TERM_SEMANTIC_LOGBETA 5 /* coarse_semantic_logbeta */
#define SNERF_TERM_CAR_REG 6          /* coarse_car_reg_loss */
#define SNERF_TERM_DS 7               /* coarse_ds */

typedef struct SnerfLossCfg {
  int32_t n_rays, n_samples, n_classes;
  int32_t color_mode;        /* 0 none, 1 SNerfLoss (plain MSE), 2 SatNerfLoss (beta-weighted + log beta) */
  int32_t has_sc;            /* solar-correction terms (needs the *_sc inputs) */
  int32_t sem_mode;          /* 0 none, 1 SemanticLoss, 2 SemanticUncertaintyLoss */
  int32_t ignore_index;      /* CrossEntropyLoss ignore_index: car index, or -100 */
  int32_t use_sbeta;         /* beta_semantic given (use_separate_beta_for_s) */
  int32_t detach_beta_for_s;
  int32_t car_reg;           /* SemanticCarRegLoss */
  int32_t car_label;
  int32_t has_depth;         /* DepthLoss on `depth` */
  float sc_lambda, lambda_s, lambda_c, ds_lambda;
} SnerfLossCfg;

typedef struct SnerfLossIn {
  const float* rgb; const float* weights; const float* beta; const float* beta_semantic;
  const float* semantic_logits;
  const float* sun_sc; const float* transparency_sc; const float* weights_sc;
  const float* depth;
  const float* gt_rgb;            /* (N,3) */
  const int64_t* labels;          /* (N) */
  const uint8_t* mask;            /* (N) bool, NULL = all rays (semantic_sparsity_mask) */
  const float* gt_depth;          /* (N) */
  const float* depth_weights;     /* (N) or NULL = 1 (ds_noweights) */
} SnerfLossIn;

typedef struct SnerfLossGrads { /* any may be NULL */
  float* rgb; float* weights; float* beta; float* beta_semantic; float* semantic_logits; float* sun_sc; float* depth;
} SnerfLossGrads;

size_t snerf_loss_workspace_bytes(const SnerfLossCfg* cfg);
int snerf_loss_partial(const SnerfLossCfg* cfg, const SnerfLossIn* in, float* totals, void* workspace,
                       size_t workspace_bytes, void* stream);
/* n_rays_global = number of rays the means run over (sum over ranks), or 0 = use the ray count that snerf_loss_partial
 * summed into `totals` (all-reduced with the other sums, so unequal shards are handled); grads are scaled by grad_scale.
 * A label outside [0, n_classes) that is not ignore_index makes the CE term NaN (torch raises there). */
int snerf_loss_finish(const SnerfLossCfg* cfg, const SnerfLossIn* in, const float* totals, float n_rays_global,
                      float grad_scale, float* terms, const SnerfLossGrads* grads, void* stream);

/* ---- optimiser --------------------------------------------------------------------------------------
 * One fused Adam step over flat fp32 buffers (parameters, gradients, exp_avg, exp_avg_sq), replacing the reference's
 * torch.optim.Adam(lr=cfgs.pipeline.learnrate, weight_decay=0) over ~60 tensors
 * (baseline/pipelines/base_ray_pipeline.py:246-269; StepLR(gamma=0.9) is the caller's `lr`).  `step` counts from 1
 * (bias corrections 1 - beta^step, computed in double like torch does on the host); gradients are multiplied by
 * `grad_scale` first (1.0 normally).  n must be a multiple of 4, all buffers 16-byte aligned device memory. */
int snerf_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, unsigned long long n,
                    float lr, float beta1, float beta2, float eps, int step, float grad_scale, void* stream);

/* ---- measurement hook ----------------------------------------------------------------------------
 * Between snerf_profile_begin and snerf_profile_end every GEMM launch is bracketed by HIP events on the
 * stream it is launched on; _end synchronises those events and returns, per kernel variant, the summed
 * device time, the algorithmic FLOPs (2*I*J*K of each launch) and the launch count.
 * variant 0: K-contiguous GEMMs (forward X.W^T and dX = dZ.(W^T)^T): gemm_kc_kernel of bsp_kc.hip (128x256 tile),
 *         1: the SIREN trunk as one persistent launch (trunk_kernel of bsp_trunk.hip; one-plane mode),
 *         2: dW = dZ^T.X (both operands read along the points, split over the points): gemm_dw_kernel (256x256 tile),
 *         3: the 32-wide head variants (gemm_kcn_kernel, gemm_dw_kernel<32>). */
#define SNERF_PROFILE_VARIANTS 4
typedef struct SnerfProfile {
  double ms[SNERF_PROFILE_VARIANTS];
  double flops[SNERF_PROFILE_VARIANTS];
  int64_t launches[SNERF_PROFILE_VARIANTS];
} SnerfProfile;
int snerf_profile_begin(void);
int snerf_profile_end(SnerfProfile* out);

/* test hooks of the block-scaled fp16-plane kernels (csrc/bsp.h): fp32 in / fp32 out around one launch; `planes` = 2 (default
 * arithmetic) or 1 (SNERF_FLAG_F16X1); synchronous and allocating -- tests only */
int snerf_test_set_kc_grid(int n_workgroups);   /* persistent grid of the K-contiguous launches (0: two per CU): forces the tile loop at test sizes */
int snerf_test_set_trunk_fusion(int on);        /* 0: launch-per-layer trunk for every pass; 1 (default): one-plane passes of the W = 512 SIREN model
                                                 * run the trunk as ONE persistent launch (csrc/bsp_trunk.hip) */
int snerf_test_bsp_roundtrip(const float* src, int rows, int cols, int ld, int col0, float* dst, int* exps_out, int planes, void* stream);
int snerf_test_bsp_kc(const float* A, const float* A2, int Ka, const float* W, const float* bias, int I, int J, int K, int a_col0,
                      int c_col0, int act, float w0, int aux_mode, const float* Hact, const unsigned* Hsign, float* C,
                      unsigned* Csign, float* colsum, const float* nd_w, float* nd_out, const int* nd_rows, int narrow, int planes, void* stream);
/* nd_w [J] / nd_out [ceil(J/256)*4][I]: the folded 1-wide projection of ACT_SIN launches; with nd_rows (HOST array, one count <= 5 per
 * 256-column tile): nd_w [sum nd_rows][J], nd_out [ceil(J/256)*4*5][I] -- the folded final head layers */
int snerf_test_bsp_dw(const float* A, int lda_src, const float* B, int ldb_src, int P, int I, int J, int a_col0, int b_col0,
                      int k_split, int narrow_i, float* C, int planes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SNERF_HIP_H */
