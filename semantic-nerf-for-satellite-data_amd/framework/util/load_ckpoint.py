"""Checkpoint interop -- mirror of framework/util/load_ckpoint.py:12-129 plus the writer the reference gets from
pytorch-lightning's ModelCheckpoint (framework/pipelines.py:259-293).

Layout (Lightning's): {"epoch", "global_step", "state_dict": {"model_<key>.<param>": tensor, ...},
"optimizer_states": [Adam state_dict], "lr_schedulers": [StepLR state_dict]} under <log_dp>/ckpoints/epoch=<n>.ckpt
or last.ckpt.  Files are read with torch.load(weights_only=True) only -- nothing in a checkpoint is executed; a file
the safe loader refuses is reported, not unpickled."""
import glob
import os

import torch


def find_ckpoint_fp(log_dp, epoch=-1):
    """framework/util/load_ckpoint.py:12-28: epoch >= 0 -> ckpoints/epoch=<n>.ckpt, else last.ckpt or the highest epoch."""
    if epoch >= 0:
        return os.path.join(log_dp, "ckpoints", f"epoch={epoch}.ckpt"), epoch
    fp = os.path.join(log_dp, "ckpoints", "last.ckpt")
    if not os.path.isfile(fp):
        fps = sorted(glob.glob(os.path.join(log_dp, "ckpoints", "*.ckpt")),
                     key=lambda x: int(x[x.index("=") + 1: x.index(".ckpt")]))
        assert len(fps) > 0, "cannot find a single *.ckpt to load"
        fp = fps[-1]
        x = os.path.basename(fp)
        epoch = int(x[x.index("=") + 1: x.index(".ckpt")])
    return fp, epoch


def _safe_load(fp, device):
    try:
        return torch.load(fp, map_location=device, weights_only=True)
    except Exception as e:  # the safe unpickler met something that is not plain tensors / containers
        raise RuntimeError(f"{fp}: torch.load(weights_only=True) refused this checkpoint ({type(e).__name__}: {e}); "
                           "it is not loaded any other way") from e


def extract_model_state_dict(checkpoint_fp, model_name, cuda_device="cpu", prefixes_to_ignore=(), prefixes_to_load=None):
    """Weights of one model out of a (Lightning or plain) checkpoint -- framework/util/load_ckpoint.py:94-129."""
    ck = _safe_load(checkpoint_fp, cuda_device)
    if "state_dict" in ck:
        ck = ck["state_dict"]
    out = {}
    for k, v in ck.items():
        if not k.startswith(model_name + "."):
            continue
        k = k[len(model_name) + 1:]
        ignore = prefixes_to_load is not None
        if any(k.startswith(p) for p in prefixes_to_ignore):
            ignore = True
        if prefixes_to_load is not None and any(k.startswith(p) for p in prefixes_to_load):
            ignore = False
        if not ignore:
            out[k] = v
    return out


@torch.no_grad()
def load_ckpoint(model, checkpoint_fp, model_name, cuda_device="cpu", prefixes_to_ignore=()):
    """framework/util/load_ckpoint.py:77-92.  Copies INTO the existing parameter storage (the parameters may be views
    of the optimiser's flat buffer), after the same key / shape checks load_state_dict(strict) makes."""
    sd = extract_model_state_dict(checkpoint_fp, model_name, cuda_device, prefixes_to_ignore)
    own = model.state_dict()
    unexpected = [k for k in sd if k not in own]
    if unexpected:
        raise RuntimeError(f"unexpected key(s) in checkpoint for {model_name}: {unexpected}")
    for k, v in sd.items():
        if tuple(v.shape) != tuple(own[k].shape):
            raise RuntimeError(f"size mismatch for {model_name}.{k}: checkpoint {tuple(v.shape)}, model {tuple(own[k].shape)}")
        own[k].copy_(v)
    return sorted(set(own) - set(sd))  # keys the checkpoint did not cover (kept as initialised)


def read_ckpt_info(checkpoint_fp):
    ck = _safe_load(checkpoint_fp, "cpu")
    return int(ck.get("epoch", 0)), int(ck.get("global_step", 0))


def load_from_disk(cfgs, log_dp, epoch=-1, device=None, device_req_free=True, prefixes_to_ignore=()):
    """framework/util/load_ckpoint.py:31-75: pipeline + models with the weights of <log_dp>/ckpoints/..., in eval mode.
    Returns (models, pipeline, epoch, device)."""
    from ..pipelines import load_pipeline
    fp, epoch = find_ckpoint_fp(log_dp, epoch=epoch)
    if not os.path.exists(fp):
        raise FileNotFoundError("Could not find checkpoint {}".format(fp))
    if device is None or isinstance(device, int):
        device = torch.device("cuda", device or 0)
    pipeline = load_pipeline(cfgs, ckpt_info=read_ckpt_info(fp)).to(device)
    for key, m in pipeline.models.items():
        load_ckpoint(m, fp, f"model_{key}", device, prefixes_to_ignore)
        m.eval()
    return pipeline.models, pipeline, pipeline.get_current_epoch(), device


def save_ckpoint(pipeline, checkpoint_fp, optimizer=None, scheduler=None, epoch=None, global_step=None):
    """Writes what Lightning's ModelCheckpoint writes for these pipelines (tensors and plain containers only)."""
    sd = {}
    for key, m in pipeline.models.items():
        for k, v in m.state_dict().items():
            sd[f"model_{key}.{k}"] = v.detach().cpu().clone()
    ck = {"epoch": int(pipeline.get_current_epoch() if epoch is None else epoch),
          "global_step": int(pipeline.train_steps if global_step is None else global_step),
          "state_dict": sd,
          "optimizer_states": [optimizer.state_dict()] if optimizer is not None else [],
          "lr_schedulers": [scheduler.state_dict()] if scheduler is not None else []}
    if ck["optimizer_states"]:
        st = ck["optimizer_states"][0]["state"]
        for s in st.values():
            for k in ("exp_avg", "exp_avg_sq"):
                s[k] = s[k].detach().cpu()
    os.makedirs(os.path.dirname(os.path.abspath(checkpoint_fp)), exist_ok=True)
    torch.save(ck, checkpoint_fp)
    return checkpoint_fp
