"""ctypes binding of libsnerf_hip.so -- mirrors include/snerf_hip.h field for field."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SNERF_LIB_PATH: an alternative build of the library (ablation harness of tools/ablate; diagnostics only)
LIB_PATH = os.environ.get("SNERF_LIB_PATH") or os.path.join(_HERE, "libsnerf_hip.so")
MAX_LAYERS = 16
ABI_VERSION = 5   # include/snerf_hip.h SNERF_ABI_VERSION

FLAG_TRAIN = 1
FLAG_SC_PASS = 2
# arithmetic bits of SnerfDesc.flags (include/snerf_hip.h): none set = the default, f16x2
FLAG_F16X2 = 64       # default: fp32-class on the fp16 matrix cores, two fp16 planes of power-of-two-scaled operands, three products
FLAG_F16X1 = 8        # reduced precision: ONE fp16 plane of the same block-scaled tensors, one product (precision = 16 / "medium" / "high")
# ModelSpec.mfma -> SnerfDesc.flags ("bf16" = the name BASELINE.json's configs use for the reduced-precision runs: same mode)
MFMA_FLAGS = {"f16x2": 0, "f16x1": FLAG_F16X1, "bf16": FLAG_F16X1}

_fp = C.POINTER(C.c_float)


class SnerfDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("n_rays", "n_samples", "fc_units", "fc_layers", "feat_last")] + [
        ("skip_mask", C.c_uint32)] + [(n, C.c_int32) for n in (
            "n_freq", "siren", "t_dim", "n_classes", "sem_sigmoid", "use_tj_instead_of_beta", "use_tj_for_s",
            "use_separate_beta_for_s", "use_separate_tj_for_semantic")] + [("flags", C.c_uint32)]


_PARAM_FIELDS = (["sigma_w", "sigma_b", "feats_w", "feats_b", "rgb_w0", "rgb_b0", "rgb_w2", "rgb_b2",
                  "sem_w0", "sem_b0", "sem_w2", "sem_b2"], ["sky_w0", "sky_b0", "sky_w2", "sky_b2",
                                                             "beta_w0", "beta_b0", "beta_w2", "beta_b2",
                                                             "sbeta_w0", "sbeta_b0", "sbeta_w2", "sbeta_b2"])


class SnerfParams(C.Structure):
    _fields_ = ([("fc_w", C.c_void_p * MAX_LAYERS), ("fc_b", C.c_void_p * MAX_LAYERS)]
                + [(n, C.c_void_p) for n in _PARAM_FIELDS[0]]
                + [("sun_w", C.c_void_p * 4), ("sun_b", C.c_void_p * 4)]
                + [(n, C.c_void_p) for n in _PARAM_FIELDS[1]])


class SnerfInputs(C.Structure):
    _fields_ = [("rays", C.c_void_p), ("xyz", C.c_void_p), ("z_vals", C.c_void_p), ("z_steps", C.c_void_p),
                ("u", C.c_void_p), ("sun_d", C.c_void_p), ("sun_stride", C.c_int32), ("_pad", C.c_int32),
                ("t", C.c_void_p), ("t_s", C.c_void_p)]


OUTPUT_FIELDS = ("rgb", "depth", "weights", "transparency", "albedo", "sun", "sky", "beta", "sigmas",
                 "beta_semantic", "semantic_logits", "semantic_label", "z_vals")
GRAD_FIELDS = ("rgb", "depth", "weights", "transparency", "albedo", "sun", "sky", "beta", "sigmas",
               "beta_semantic", "semantic_logits")


class SnerfOutputs(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in OUTPUT_FIELDS]


class SnerfOutGrads(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in GRAD_FIELDS]


class SnerfLossCfg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("n_rays", "n_samples", "n_classes", "color_mode", "has_sc", "sem_mode",
                                          "ignore_index", "use_sbeta", "detach_beta_for_s", "car_reg", "car_label",
                                          "has_depth")] + [(n, C.c_float) for n in (
                                              "sc_lambda", "lambda_s", "lambda_c", "ds_lambda")]


LOSS_IN_FIELDS = ("rgb", "weights", "beta", "beta_semantic", "semantic_logits", "sun_sc", "transparency_sc",
                  "weights_sc", "depth", "gt_rgb", "labels", "mask", "gt_depth", "depth_weights")
LOSS_GRAD_FIELDS = ("rgb", "weights", "beta", "beta_semantic", "semantic_logits", "sun_sc", "depth")
LOSS_NTOT = 16
LOSS_TERMS = ("coarse_color", "coarse_logbeta", "coarse_sc_term2", "coarse_sc_term3", "coarse_semantic",
              "coarse_semantic_logbeta", "coarse_car_reg_loss", "coarse_ds")


class SnerfLossIn(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in LOSS_IN_FIELDS]


class SnerfLossGrads(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in LOSS_GRAD_FIELDS]


class SnerfProfile(C.Structure):
    _fields_ = [("ms", C.c_double * 4), ("flops", C.c_double * 4), ("launches", C.c_int64 * 4)]


PROFILE_VARIANTS = ("K-contiguous dense layers (forward X.W^T and dX = dZ.W): gemm_kc_kernel, 128 x 256 tile", "SIREN trunk as one persistent launch (one-plane mode): trunk_kernel, activation tile resident in LDS",
                    "weight gradients (dW = dZ^T.X, split-K slabs): gemm_dw_kernel, 256 x 256 tile", "32-wide head variants (forward + dW)")

_lib = None


def lib():
    """Load the HIP library; no fallback of any kind."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"libsnerf_hip.so not found at {LIB_PATH}: the HIP extension is the product path and has no "
            "fallback. Build it with `make -C semantic-nerf-for-satellite-data_amd/csrc` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`).")
    # torch bundles its own HIP runtime: import it FIRST so that libsnerf_hip.so binds to the runtime that
    # owns torch's device context and streams (loading ours first brings up a second runtime that then
    # reports "no ROCm-capable device").
    import torch  # noqa: F401
    L = C.CDLL(LIB_PATH)
    L.snerf_version.restype = C.c_int
    L.snerf_last_error.restype = C.c_char_p
    L.snerf_packed_floats.restype = C.c_size_t
    L.snerf_packed_floats.argtypes = [C.POINTER(SnerfDesc)]
    L.snerf_grad_floats.restype = C.c_size_t
    L.snerf_grad_floats.argtypes = [C.POINTER(SnerfDesc)]
    L.snerf_workspace_bytes.restype = C.c_size_t
    L.snerf_workspace_bytes.argtypes = [C.POINTER(SnerfDesc)]
    L.snerf_pack_params.restype = C.c_int
    L.snerf_pack_params.argtypes = [C.POINTER(SnerfDesc), C.POINTER(SnerfParams), C.c_void_p, C.c_void_p]
    L.snerf_unpack_grads.restype = C.c_int
    L.snerf_unpack_grads.argtypes = [C.POINTER(SnerfDesc), C.c_void_p, C.POINTER(SnerfParams), C.c_int, C.c_void_p]
    L.snerf_forward.restype = C.c_int
    L.snerf_forward.argtypes = [C.POINTER(SnerfDesc), C.c_void_p, C.POINTER(SnerfInputs), C.POINTER(SnerfOutputs),
                                C.c_void_p, C.c_size_t, C.c_void_p]
    L.snerf_backward.restype = C.c_int
    L.snerf_backward.argtypes = [C.POINTER(SnerfDesc), C.c_void_p, C.POINTER(SnerfInputs), C.POINTER(SnerfOutGrads),
                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    L.snerf_test_set_kc_grid.restype = C.c_int
    L.snerf_test_set_kc_grid.argtypes = [C.c_int]
    L.snerf_test_set_trunk_fusion.restype = C.c_int
    L.snerf_test_set_trunk_fusion.argtypes = [C.c_int]
    L.snerf_test_bsp_roundtrip.restype = C.c_int
    L.snerf_test_bsp_roundtrip.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    L.snerf_test_bsp_kc.restype = C.c_int
    L.snerf_test_bsp_kc.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                    C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.c_int, C.c_int, C.c_void_p]
    L.snerf_test_bsp_dw.restype = C.c_int
    L.snerf_test_bsp_dw.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                    C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    L.snerf_loss_workspace_bytes.restype = C.c_size_t
    L.snerf_loss_workspace_bytes.argtypes = [C.POINTER(SnerfLossCfg)]
    L.snerf_loss_partial.restype = C.c_int
    L.snerf_loss_partial.argtypes = [C.POINTER(SnerfLossCfg), C.POINTER(SnerfLossIn), C.c_void_p, C.c_void_p,
                                     C.c_size_t, C.c_void_p]
    L.snerf_loss_finish.restype = C.c_int
    L.snerf_loss_finish.argtypes = [C.POINTER(SnerfLossCfg), C.POINTER(SnerfLossIn), C.c_void_p, C.c_float, C.c_float,
                                    C.c_void_p, C.POINTER(SnerfLossGrads), C.c_void_p]
    L.snerf_sample_z.restype = C.c_int
    L.snerf_sample_z.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    L.snerf_embedding_rows.restype = C.c_int
    L.snerf_embedding_rows.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    L.snerf_embedding_backward.restype = C.c_int
    L.snerf_embedding_backward.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.snerf_adam_step.restype = C.c_int
    L.snerf_adam_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_ulonglong, C.c_float, C.c_float,
                                  C.c_float, C.c_float, C.c_int, C.c_float, C.c_void_p]
    L.snerf_profile_begin.restype = C.c_int
    L.snerf_profile_end.restype = C.c_int
    L.snerf_profile_end.argtypes = [C.POINTER(SnerfProfile)]
    if L.snerf_version() != ABI_VERSION:
        raise RuntimeError(f"libsnerf_hip.so ABI version {L.snerf_version()} != {ABI_VERSION}")
    _lib = L
    return L


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed (code {rc}): {lib().snerf_last_error().decode()}")


EXPORTED_SYMBOLS = ("snerf_version", "snerf_last_error", "snerf_packed_floats", "snerf_grad_floats", "snerf_workspace_bytes",
                    "snerf_pack_params", "snerf_unpack_grads", "snerf_forward", "snerf_backward",
                    "snerf_loss_workspace_bytes", "snerf_loss_partial", "snerf_loss_finish", "snerf_profile_begin",
                    "snerf_profile_end", "snerf_sample_z", "snerf_adam_step", "snerf_test_bsp_roundtrip", "snerf_test_bsp_kc",
                    "snerf_test_bsp_dw", "snerf_test_set_kc_grid", "snerf_test_set_trunk_fusion", "snerf_embedding_rows", "snerf_embedding_backward")
