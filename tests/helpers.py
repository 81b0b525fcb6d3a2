"""Shared helpers for the parity tests (fixtures -> oracle config / tensors)."""
import json
import os

import numpy as np
import torch

from oracle import snerf_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_fixture(name):
    z = np.load(os.path.join(GOLDEN, f"{name}.npz"), allow_pickle=False)
    meta = json.loads(str(z["meta_json"]))
    c = dict(meta["cfg"])
    c["fc_skips"] = tuple(c["fc_skips"])
    cfg = O.OracleCfg(**c)
    return z, meta, cfg


def fixture_params(z, meta, cfg):
    """Parameters regenerate from (cfg, seed); small fixtures also store them, which pins the generator."""
    params = O.init_params_numpy(cfg, meta["seed"])
    stored = [k for k in z.files if k.startswith("param_")]
    for k in stored:
        assert np.array_equal(z[k], params[k[len("param_"):]]), k
    return params


def fixture_batch(z, prefix="in_"):
    """Main batch (prefix 'in_') or the depth-ray batch ('in_depth_': rays / extras / u only)."""
    depth_keys = {"in_depth_rays", "in_depth_extras", "in_depth_u"}
    b = {}
    for k in z.files:
        if prefix == "in_" and k.startswith("in_") and k not in depth_keys:
            b[k[3:]] = z[k]
        elif prefix == "in_depth_" and k in depth_keys:
            b[k[len(prefix):]] = z[k]
    return O.batch_to_torch(b)


def max_abs(a, b):
    a = torch.as_tensor(np.asarray(a)).double()
    b = torch.as_tensor(np.asarray(b)).double()
    return float((a - b).abs().max()) if a.numel() else 0.0


def rel_err(a, b):
    a = torch.as_tensor(np.asarray(a)).double().reshape(-1)
    b = torch.as_tensor(np.asarray(b)).double().reshape(-1)
    return float((a - b).norm() / (b.norm() + 1e-30))
