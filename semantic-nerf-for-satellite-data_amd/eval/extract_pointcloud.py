"""Point clouds from a trained model -- the device side of eval/extract_pointcloud.py:66-114: full-frame inference of
one image's rays (lean: rgb + depth [+ label] only, written in place per chunk), then the ray end points
xyz = o + d * depth in double precision (baseline/dataset/satnerf_dataset.py:156-171).  Lat/lon/alt conversion, DSM
rasterisation and the normals variants stay with the dataset / CPU tooling (out of scope)."""
import numpy as np
import torch

from .utils.util import lean_inference, sharded_lean_inference


def get_xyz_from_nerf_prediction(rays: torch.Tensor, depth: torch.Tensor) -> torch.Tensor:
    """baseline/dataset/satnerf_dataset.py:156-171 -- in double, as the reference does, on whatever device holds the rays."""
    rays = rays.double()
    depth = depth.double()
    return rays[:, 0:3] + rays[:, 3:6] * depth.view(-1, 1)


@torch.no_grad()
def extract_pointcloud(cfgs, renderer, models, rays, extras, render_options=None, with_labels=None, sharded=False):
    """One image -> {"xyz_n" (R,3) f64, "colors" (R,3) f32, "depth" (R) f32 [, "labels" (R) i64]}, all on the device.
    `sharded=False` (default) is the reference's single-device call: safe from one rank of a process group (the usual export
    inside a data-parallel run).  `sharded=True` is a COLLECTIVE: every rank of the group must call it with the same rays;
    the image is rendered in rank shards and the per-ray results gathered, so every rank returns the whole cloud
    (eval/utils/util.py: sharded_lean_inference)."""
    sem = models["coarse"].spec.n_classes > 0
    if with_labels is None:
        with_labels = sem
    keys = ["rgb_coarse", "depth_coarse"] + (["semantic_label_coarse"] if with_labels and sem else [])
    infer = sharded_lean_inference if sharded else lean_inference
    res = infer(cfgs, renderer, models, rays, extras, keys=keys, render_options=render_options or {})
    out = {"xyz_n": get_xyz_from_nerf_prediction(rays, res["depth_coarse"]), "colors": res["rgb_coarse"],
           "depth": res["depth_coarse"]}
    if "semantic_label_coarse" in res:
        out["labels"] = res["semantic_label_coarse"]
    return out


def filtered_indices(n_points: int, keep: int = 30000, seed: int = 0) -> torch.Tensor:
    """the reference's reduced cloud: a seeded randperm prefix (eval/extract_pointcloud.py:96-100)"""
    g = torch.Generator().manual_seed(seed)
    return torch.randperm(n_points, generator=g)[:keep]


def save_ply(fp: str, xyz, colors=None, labels=None):
    """binary little-endian PLY: double xyz, uchar rgb (colors in [0,1]), optional uchar label"""
    xyz = np.asarray(xyz.detach().cpu() if torch.is_tensor(xyz) else xyz, dtype="<f8")
    fields = [("x", "<f8"), ("y", "<f8"), ("z", "<f8")]
    props = ["property double x", "property double y", "property double z"]
    if colors is not None:
        colors = np.asarray(colors.detach().cpu() if torch.is_tensor(colors) else colors)
        fields += [("red", "u1"), ("green", "u1"), ("blue", "u1")]
        props += ["property uchar red", "property uchar green", "property uchar blue"]
    if labels is not None:
        labels = np.asarray(labels.detach().cpu() if torch.is_tensor(labels) else labels)
        fields += [("label", "u1")]
        props += ["property uchar label"]
    rec = np.empty(xyz.shape[0], dtype=fields)
    rec["x"], rec["y"], rec["z"] = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    if colors is not None:
        c = np.clip(np.rint(colors * 255.0), 0, 255).astype("u1")
        rec["red"], rec["green"], rec["blue"] = c[:, 0], c[:, 1], c[:, 2]
    if labels is not None:
        rec["label"] = labels.astype("u1")
    with open(fp, "wb") as f:
        f.write(("ply\nformat binary_little_endian 1.0\nelement vertex %d\n%s\nend_header\n" % (xyz.shape[0], "\n".join(props))).encode())
        f.write(rec.tobytes())
    return fp
