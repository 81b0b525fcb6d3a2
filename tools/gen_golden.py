#!/usr/bin/env python3
"""Generate golden input/output vectors by running the REFERENCE's own hot-path modules.

Run in the build container only (the reference never travels to the GPU box):

    PYTHONPATH=/root/reference PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden.py

It imports (read-only) from /root/reference:
  semantic.models.rs_semantic.{RSSemanticNeRF, inference}, baseline.models.satnerf.{SatNeRF, inference},
  semantic.components.rendering.RSSemanticRendering, baseline.components.rendering.SatNeRFRendering,
  baseline.components.loss.*, semantic.components.loss.*
and writes data-only fixtures (inputs, expected outputs, loss values, gradients) to tests/golden/*.npz.

Weights are NOT drawn from torch's RNG: every parameter is overwritten from the NumPy stream of
oracle.snerf_oracle.init_params_numpy(cfg, seed) so that tests can regenerate them from (cfg, seed).
The jitter of sample_rays (framework/components/rendering.py:108) is made reproducible by replacing
torch.rand_like with a function that returns the stored uniform tensor while render_rays runs.
"""
import json
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = os.environ.get("SNERF_REFERENCE", "/root/reference")
sys.path.insert(0, REF)
sys.dont_write_bytecode = True

from oracle import snerf_oracle as O  # noqa: E402  (parameter stream + synthetic batches only)

from semantic.models.rs_semantic import RSSemanticNeRF, inference as sem_inference  # noqa: E402
from baseline.models.satnerf import SatNeRF, inference as sat_inference  # noqa: E402
from semantic.components.rendering import RSSemanticRendering  # noqa: E402
from baseline.components.rendering import SatNeRFRendering  # noqa: E402
from baseline.components.loss import SatNerfLoss, SNerfLoss, DepthLoss  # noqa: E402
from semantic.components.loss import SemanticLoss, SemanticUncertaintyLoss, SemanticCarRegLoss  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def ref_cfgs(cfg: O.OracleCfg):
    pipe = types.SimpleNamespace(
        fc_layers=cfg.fc_layers, fc_units=cfg.fc_units, fc_use_full_features=cfg.fc_use_full_features,
        fc_skips=list(cfg.fc_skips), activation_function=cfg.activation_function,
        t_embedding_tau=cfg.t_embedding_tau, mapping_pos_n_freq=cfg.mapping_pos_n_freq, mapping_dir_n_freq=4,
        use_tj_instead_of_beta=cfg.use_tj_instead_of_beta, use_tj_for_s=cfg.use_tj_for_s,
        semantic_activation_function=cfg.semantic_activation_function,
        use_separate_beta_for_s=cfg.use_separate_beta_for_s,
        use_separate_tj_for_semantic=cfg.use_separate_tj_for_semantic,
        render_chunk_size=cfg.render_chunk_size, n_samples=cfg.n_samples, sc_lambda=cfg.sc_lambda)
    return types.SimpleNamespace(pipeline=pipe)


def build_reference(cfg: O.OracleCfg, seed: int):
    cfgs = ref_cfgs(cfg)
    if cfg.model == "semantic":
        model = RSSemanticNeRF(cfgs, types.SimpleNamespace(semantic_n_classes=cfg.n_classes))
        renderer = RSSemanticRendering(cfgs, inference=sem_inference)
    else:
        model = SatNeRF(cfgs, layers=cfg.fc_layers, feat=cfg.fc_units, skips=list(cfg.fc_skips),
                        t_embedding_dims=cfg.t_embedding_tau, siren=cfg.siren)
        renderer = SatNeRFRendering(cfgs)
    params = O.init_params_numpy(cfg, seed)
    sd = model.state_dict()
    assert list(sd.keys()) == list(params.keys()), (list(sd.keys()), list(params.keys()))
    for k in sd:
        assert tuple(sd[k].shape) == params[k].shape, (k, sd[k].shape, params[k].shape)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
    emb = torch.nn.Embedding(cfg.t_embedding_vocab, cfg.t_embedding_tau)
    with torch.no_grad():
        emb.weight.copy_(torch.from_numpy(O.init_embedding_numpy(cfg, seed)))
    models = {"coarse": model, "t": emb}
    if cfg.model == "semantic" and cfg.use_separate_tj_for_semantic:
        emb_s = torch.nn.Embedding(cfg.t_embedding_vocab, cfg.t_embedding_tau)
        with torch.no_grad():
            emb_s.weight.copy_(torch.from_numpy(O.init_embedding_numpy(cfg, seed + 1)))
        models["t_s"] = emb_s
    return cfgs, models, renderer, params


def render_with_u(renderer, models, rays, extras, u):
    """Run the reference's BaseRenderer.render_rays with its rand_like draw replaced by ``u``."""
    orig = torch.rand_like
    calls = []

    def fake(x, *a, **k):
        calls.append(tuple(x.shape))
        assert tuple(x.shape) == tuple(u.shape)
        return u.clone()

    torch.rand_like = fake
    try:
        res = renderer.render_rays(models, rays, extras)
    finally:
        torch.rand_like = orig
    assert len(calls) == 1, calls
    return res


def losses_for_epoch(cfg, res, batch, epoch, depth_res=None):
    """The gate combinations of semantic/components/training_step.py:22-92 applied with the
    reference's own loss modules."""
    ld = {}
    if epoch < cfg.first_beta_epoch:
        _, d = SNerfLoss(lambda_sc=cfg.sc_lambda)(res, batch["rgbs"])
    else:
        _, d = SatNerfLoss(lambda_sc=cfg.sc_lambda)(res, batch["rgbs"])
    ld.update(d)
    if depth_res is not None:
        _, d = DepthLoss(lambda_ds=cfg.ds_lambda)(depth_res, batch["depths"], batch["depth_weights"])
        ld.update(d)
    if cfg.model == "semantic":
        if epoch < cfg.first_beta_epoch or not cfg.use_beta_for_s:
            _, d = SemanticLoss(cfg.lambda_s, cfg.car_index, ignore_car_index=cfg.ignore_car_index)(
                res, batch["semantic"], batch["mask"])
        else:
            _, d = SemanticUncertaintyLoss(cfg.lambda_s, cfg.car_index, detach_beta_for_s=cfg.detach_beta_for_s,
                                           ignore_car_index=cfg.ignore_car_index)(res, batch["semantic"], batch["mask"])
        ld.update(d)
        if cfg.use_car_reg_loss and epoch >= cfg.car_reg_loss_start:
            _, d = SemanticCarRegLoss(cfg.lambda_c, cfg.car_index)(res, batch["semantic"], batch["mask"])
            ld.update(d)
    return ld


def make_case(name, cfg: O.OracleCfg, n_rays, seed, epoch, with_depth=False, store_params=True,
              per_sample=True, grad_mode="full", adam_steps=0, mask_frac=None, car_prob=0.2, full_grads=(), yard_noise=()):
    torch.manual_seed(0)
    cfgs, models, renderer, params = build_reference(cfg, seed)
    b = O.synthetic_batch(n_rays, cfg.n_samples, seed=seed + 100, n_classes=max(cfg.n_classes, 2), car_prob=car_prob)
    if mask_frac is not None:
        rng = np.random.default_rng(seed + 5)
        b["mask"] = rng.uniform(size=n_rays) > mask_frac
    bt = O.batch_to_torch(b)
    fix = {f"in_{k}": v for k, v in b.items()}

    res = render_with_u(renderer, models, bt["rays"], bt["extras"], bt["u"])
    depth_res = None
    if with_depth:
        bd = O.synthetic_batch(n_rays, cfg.n_samples, seed=seed + 200)
        for k in ("rays", "extras", "u"):
            fix[f"in_depth_{k}"] = bd[k]
        bdt = O.batch_to_torch(bd)
        depth_res = render_with_u(renderer, models, bdt["rays"], bdt["extras"], bdt["u"])
        fix["out_depth_depth_coarse"] = depth_res["depth_coarse"].detach().numpy()
    ld = losses_for_epoch(cfg, res, bt, epoch, depth_res)
    loss = sum(ld.values())
    for m in models.values():
        m.zero_grad()
    loss.backward()

    per_ray_keys = {"rgb_coarse", "depth_coarse", "semantic_logits_coarse", "semantic_label_coarse"}
    for k, v in res.items():
        if per_sample or k in per_ray_keys or k in ("weights_coarse", "sigmas_coarse", "weights_sc_coarse",
                                                    "sun_sc_coarse", "transparency_sc_coarse", "beta_coarse"):
            fix[f"out_{k}"] = v.detach().numpy()
    for k, v in ld.items():
        fix[f"loss_{k}"] = np.float64(v.item())
    fix["loss_total"] = np.float64(loss.item())

    grads = {k: p.grad.detach().numpy().copy() for k, p in models["coarse"].named_parameters()}
    grads["model_t.weight"] = models["t"].weight.grad.detach().numpy().copy()
    if "t_s" in models and models["t_s"].weight.grad is not None:
        grads["model_t_s.weight"] = models["t_s"].weight.grad.detach().numpy().copy()
    for k, g in grads.items():
        if grad_mode == "full":
            fix[f"grad_{k}"] = g
        else:  # norms + a strided sample (full-width case: weights regenerate from the seed)
            fix[f"gradnorm_{k}"] = np.float64(np.sqrt((g.astype(np.float64) ** 2).sum()))
            fix[f"gradsample_{k}"] = g.reshape(-1)[:: max(1, g.size // 64)][:64].copy()
            if k in full_grads:   # the reference's FULL gradient of selected tensors (first / skip trunk layer, a head's first and last layer, the embedding)
                fix[f"grad_{k}"] = g
    if store_params:
        for k, v in params.items():
            fix[f"param_{k}"] = v

    for eps in yard_noise:
        # Yardsticks for the REDUCED-precision mode, made by the reference: the same case from weights carrying `eps` relative Gaussian
        # noise -- the worst absolute deviation of any stored output, the worst relative deviation of a loss term and the worst relative
        # L2 deviation of a parameter gradient from the clean run above.
        torch.manual_seed(0)
        _, models_n, renderer_n, _ = build_reference(cfg, seed)
        rng = np.random.default_rng(seed + 99)
        with torch.no_grad():
            for p_ in models_n["coarse"].parameters():
                p_.mul_(torch.from_numpy(1.0 + float(eps) * rng.standard_normal(tuple(p_.shape))).float())
        res_n = render_with_u(renderer_n, models_n, bt["rays"], bt["extras"], bt["u"])
        ld_n = losses_for_epoch(cfg, res_n, bt, epoch, None)
        sum(ld_n.values()).backward()
        worst_out = max(float((res_n[k[4:]].detach() - torch.from_numpy(fix[k])).abs().max()) for k in fix
                        if k.startswith("out_") and k != "out_semantic_label_coarse" and k[4:] in res_n)
        worst_loss = max(abs(ld_n[k].item() - ld[k].item()) / max(1.0, abs(ld[k].item())) for k in ld_n)
        gn = {k: p_.grad.detach().numpy() for k, p_ in models_n["coarse"].named_parameters()}
        worst_grad = max(float(np.linalg.norm((gn[k] - grads[k]).ravel()) / (np.linalg.norm(grads[k].ravel()) + 1e-30)) for k in gn
                         if float(np.abs(grads[k]).max()) > 0)
        fix[f"yard_out_abs_noise_{eps}"] = np.float64(worst_out)
        fix[f"yard_loss_rel_noise_{eps}"] = np.float64(worst_loss)
        fix[f"yard_grad_rel_noise_{eps}"] = np.float64(worst_grad)
        print(f"{name}: reference under {eps} weight noise: outputs {worst_out:.2e} abs, loss terms {worst_loss:.2e} rel, gradients {worst_grad:.2e} rel L2")

    if adam_steps:
        # a18: Adam(lr=5e-4, wd=0) trajectory on the reference model (base_ray_pipeline.py:246-254)
        plist = [p for m in models.values() for p in m.parameters()]
        opt = torch.optim.Adam(plist, lr=5e-4, weight_decay=0)
        traj = [loss.item()]
        opt.step()
        for _ in range(adam_steps):
            opt.zero_grad()
            r = render_with_u(renderer, models, bt["rays"], bt["extras"], bt["u"])
            l = sum(losses_for_epoch(cfg, r, bt, epoch).values())
            traj.append(l.item())
            l.backward()
            opt.step()
        fix["adam_traj"] = np.array(traj, dtype=np.float64)

    meta = dict(name=name, seed=seed, epoch=epoch, n_rays=n_rays, with_depth=with_depth,
                cfg={k: (list(v) if isinstance(v, tuple) else v) for k, v in vars(cfg).items()},
                torch=torch.__version__, reference="wagnva/semantic-nerf-for-satellite-data@2025-03-21")
    fix["meta_json"] = np.array(json.dumps(meta))
    path = os.path.join(OUT, f"{name}.npz")
    np.savez_compressed(path, **fix)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB, loss={loss.item():.6f}, terms={ {k: round(v.item(), 6) for k, v in ld.items()} }")


def inference_case(name, cfg: O.OracleCfg, n_rays, seed):
    """Seam 3: reference inference(model, cfgs, xyz, z_vals, ...) on explicit xyz / z_vals."""
    cfgs, models, renderer, params = build_reference(cfg, seed)
    rng = np.random.default_rng(seed + 300)
    S = cfg.n_samples
    z = np.sort(rng.uniform(0.0, 1.5, (n_rays, S)).astype(np.float32), axis=1)
    xyz = rng.uniform(-1.2, 1.2, (n_rays, S, 3)).astype(np.float32)
    sun = rng.standard_normal((n_rays, 3)).astype(np.float32)
    sun /= np.linalg.norm(sun, axis=1, keepdims=True)
    t = rng.standard_normal((n_rays, cfg.t_embedding_tau)).astype(np.float32)
    with torch.no_grad():
        if cfg.model == "semantic":
            r = sem_inference(models["coarse"], cfgs, torch.from_numpy(xyz), torch.from_numpy(z), rays_d=None,
                              sun_d=torch.from_numpy(sun), rays_t=torch.from_numpy(t), rays_t_s=None)
        else:
            r = sat_inference(models["coarse"], cfgs, torch.from_numpy(xyz), torch.from_numpy(z), rays_d=None,
                              sun_d=torch.from_numpy(sun), rays_t=torch.from_numpy(t))
    fix = dict(in_xyz=xyz, in_z=z, in_sun=sun, in_t=t)
    for k, v in r.items():
        fix[f"out_{k}"] = v.numpy()
    for k, v in params.items():
        fix[f"param_{k}"] = v
    meta = dict(name=name, seed=seed, n_rays=n_rays, cfg={k: (list(v) if isinstance(v, tuple) else v) for k, v in vars(cfg).items()})
    fix["meta_json"] = np.array(json.dumps(meta))
    path = os.path.join(OUT, f"{name}.npz")
    np.savez_compressed(path, **fix)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")


def trajectory_case(name, cfg: O.OracleCfg, n_rays, seed, steps, final_params=("fc_net.8.weight",), car_prob=0.2, noise_rel=None):
    """a16 + a18 composed over MANY steps: the reference's renderer, its loss modules under the gates of
    semantic/components/training_step.py:22-92 and stock torch.optim.Adam(lr 5e-4, wd 0) (base_ray_pipeline.py:246-254) on ONE
    fixed batch (epoch 0 of a config whose first_beta_epoch is 0, so a training loop that counts epochs itself sees the same gates).
    Stored: the inputs, the total loss and every loss term BEFORE each of the `steps` optimiser steps and after the last one
    (steps + 1 values), and the final values of `final_params` (+ the embedding).  Parameters regenerate from (cfg, seed)."""
    torch.manual_seed(0)
    cfgs, models, renderer, params = build_reference(cfg, seed)
    if noise_rel:      # a yardstick run: the same trajectory from weights carrying `noise_rel` relative noise (only its loss curve is stored)
        rng = np.random.default_rng(seed + 99)
        with torch.no_grad():
            for p in models["coarse"].parameters():
                p.mul_(torch.from_numpy(1.0 + noise_rel * rng.standard_normal(tuple(p.shape))).float())
    b = O.synthetic_batch(n_rays, cfg.n_samples, seed=seed + 100, n_classes=max(cfg.n_classes, 2), car_prob=car_prob)
    bt = O.batch_to_torch(b)
    fix = {f"in_{k}": v for k, v in b.items()}
    plist = [p for m in models.values() for p in m.parameters()]
    opt = torch.optim.Adam(plist, lr=5e-4, weight_decay=0)
    traj, terms = [], {}
    for it in range(steps + 1):
        opt.zero_grad()
        r = render_with_u(renderer, models, bt["rays"], bt["extras"], bt["u"])
        ld = losses_for_epoch(cfg, r, bt, 0)
        l = sum(ld.values())
        traj.append(l.item())
        for k, v in ld.items():
            terms.setdefault(k, []).append(v.item())
        if it == steps:
            break
        l.backward()
        opt.step()
    if noise_rel:
        return np.array(traj, dtype=np.float64)
    fix["traj_total"] = np.array(traj, dtype=np.float64)
    for k, v in terms.items():
        fix[f"traj_{k}"] = np.array(v, dtype=np.float64)
    named = dict(models["coarse"].named_parameters())
    for k in final_params:
        fix[f"final_{k}"] = named[k].detach().numpy().copy()
    fix["final_model_t.weight"] = models["t"].weight.detach().numpy().copy()
    meta = dict(name=name, seed=seed, epoch=0, n_rays=n_rays, steps=steps,
                cfg={k: (list(v) if isinstance(v, tuple) else v) for k, v in vars(cfg).items()},
                torch=torch.__version__, reference="wagnva/semantic-nerf-for-satellite-data@2025-03-21")
    fix["meta_json"] = np.array(json.dumps(meta))
    return fix, traj


def save_trajectory(name, fix, traj, steps, noisy=None):
    if noisy:     # {"1e-3": curve}: the reference's own loss curve from weights with that much relative noise (yardstick of the one-plane mode)
        for k, v in noisy.items():
            fix[f"yard_total_noise_{k}"] = v
    path = os.path.join(OUT, f"{name}.npz")
    np.savez_compressed(path, **fix)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB, loss {traj[0]:.6f} -> {traj[-1]:.6f} over {steps} steps" +
          ("".join(f"; noise {k}: max |dloss| {np.abs(v - fix['traj_total']).max():.2e}" for k, v in (noisy or {}).items())))


def convergence_case(name, cfg: O.OracleCfg, seed, n_bank, n_test, batch, steps, eval_every, scene_seed=3, perturb_rel=(0.0, 1e-6, 1e-3)):
    """The BASELINE metric's quality half (PSNR / mIoU / altitude error) where no real scene exists: the reference -- its renderer,
    its loss modules under the epoch gates of semantic/components/training_step.py:22-92, torch.optim.Adam(lr 5e-4) with
    StepLR(1, 0.9) per epoch (base_ray_pipeline.py:246-269, framework/util/train_util.py:45-60) -- TRAINED for `steps` steps on the
    learnable synthetic scene of oracle.synthetic_scene (contiguous batches of the bank, no shuffle, the jitter of step i drawn by
    oracle.scene_jitter), evaluated on held-out rays every `eval_every` steps.  Run once per entry of `perturb_rel`: from the seeded
    initial weights, and from those weights times (1 + eps N(0, 1)) -- the spread between the runs is what noise of that size in the weights
    alone does to these curves (1e-6: fp32 rounding; 1e-3: the 11-bit operands of the reduced-precision mode), i.e. the yardstick for a
    build that follows the reference within its arithmetic.  Stored: the curves
    only; scene, weights and jitter regenerate from the seeds."""
    train, test = O.synthetic_scene(n_bank, n_test, seed=scene_seed, n_classes=cfg.n_classes)
    spe = max(1, n_bank // batch)
    fix = {}
    u_test = torch.from_numpy(O.scene_jitter(seed, -1, n_test, cfg.n_samples))
    tt = {k: torch.from_numpy(v) for k, v in test.items()}
    for run, eps in enumerate(perturb_rel):
        torch.manual_seed(0)
        cfgs, models, renderer, params = build_reference(cfg, seed)
        if eps:
            rng = np.random.default_rng(seed + 99)
            with torch.no_grad():
                for p in models["coarse"].parameters():
                    p.mul_(torch.from_numpy(1.0 + eps * rng.standard_normal(tuple(p.shape))).float())
        plist = [p for m in models.values() for p in m.parameters()]
        opt = torch.optim.Adam(plist, lr=5e-4, weight_decay=0)
        sched = torch.optim.lr_scheduler.StepLR(opt, step_size=1, gamma=0.9)
        curves = {k: [] for k in ("step", "loss", "psnr", "acc", "miou", "depth_mae")}

        def evaluate(step, loss):
            with torch.no_grad():
                r = render_with_u(renderer, models, tt["rays"], tt["extras"], u_test)
            m = O.scene_metrics(r["rgb_coarse"].numpy(), r["depth_coarse"].numpy(), r["semantic_logits_coarse"].numpy(), test, cfg.car_index)
            curves["step"].append(step); curves["loss"].append(loss)
            for k, v in m.items():
                curves[k].append(v)

        last = float("nan")
        for it in range(steps):
            if it % eval_every == 0:
                evaluate(it, last)
            epoch, k = divmod(it, spe)
            idx = (np.arange(k * batch, (k + 1) * batch)) % n_bank
            bt = {"rays": torch.from_numpy(train["rays"][idx]), "extras": torch.from_numpy(train["extras"][idx]),
                  "rgbs": torch.from_numpy(train["rgbs"][idx]), "semantic": torch.from_numpy(train["semantic"][idx]),
                  "mask": torch.from_numpy(train["mask"][idx])}
            u = torch.from_numpy(O.scene_jitter(seed, it, batch, cfg.n_samples))
            opt.zero_grad()
            r = render_with_u(renderer, models, bt["rays"], bt["extras"], u)
            l = sum(losses_for_epoch(cfg, r, bt, epoch).values())
            l.backward()
            opt.step()
            last = l.item()
            if (it + 1) % spe == 0:
                sched.step()
        evaluate(steps, last)
        for k, v in curves.items():
            fix[f"run{run}_{k}"] = np.array(v, dtype=np.float64)
        print(f"{name} run {run} (eps {eps}): " + ", ".join(f"{k} {curves[k][0]:.3f} -> {curves[k][-1]:.3f}" for k in ("psnr", "acc", "miou", "depth_mae")))
    meta = dict(name=name, seed=seed, scene_seed=scene_seed, n_bank=n_bank, n_test=n_test, batch=batch, steps=steps, eval_every=eval_every,
                perturb_rel=list(perturb_rel), cfg={k: (list(v) if isinstance(v, tuple) else v) for k, v in vars(cfg).items()},
                torch=torch.__version__, reference="wagnva/semantic-nerf-for-satellite-data@2025-03-21")
    fix["meta_json"] = np.array(json.dumps(meta))
    path = os.path.join(OUT, f"{name}.npz")
    np.savez_compressed(path, **fix)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")


FULL_GRADS_SEM = ("fc_net.0.weight", "fc_net.8.weight", "sun_v_net.0.weight", "semantic_prediction.2.weight", "model_t.weight")


def main():
    os.makedirs(OUT, exist_ok=True)
    if len(sys.argv) > 1 and sys.argv[1] == "--trajectories":   # only the round-5 additions (the other fixtures regenerate array-identically)
        # I: long optimiser trajectories of the whole composed step -- 25 steps at W = 32 with L_t on, 10 steps at the full width
        make_trajectories()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "--full-width":     # G only (+ the yardsticks of the REDUCED-precision mode, round 5)
        make_case("sem_siren_full", O.OracleCfg(), 16, seed=8, epoch=2, store_params=False, per_sample=False,
                  grad_mode="sample", full_grads=FULL_GRADS_SEM, yard_noise=("1e-3", "3e-4"))
        return
    if len(sys.argv) > 1 and sys.argv[1] == "--convergence":    # J: the reference trained on the learnable synthetic scene
        make_convergence()
        return
    small = dict(fc_units=32, n_samples=16, render_chunk_size=200)  # chunk < P: exercises the chunk loop
    # A: default semantic config at reduced width, beta loss active
    make_case("sem_siren_small", O.OracleCfg(**small), 48, seed=1, epoch=2, adam_steps=3)
    # B: relu trunk, no sigmoid on the semantic head, epoch 0 (plain MSE), masked rays
    make_case("sem_relu_small", O.OracleCfg(activation_function="relu", semantic_activation_function="none",
                                            ignore_car_index=False, **small), 48, seed=2, epoch=0, mask_frac=0.3)
    # C: beta_s head + t_j into the semantic head + beta-weighted CE + L_t
    make_case("sem_variants_small", O.OracleCfg(use_tj_for_s=True, use_separate_beta_for_s=True, use_beta_for_s=True,
                                                use_car_reg_loss=True, **small), 48, seed=3, epoch=3)
    # D: t_j instead of beta into rgb head, separate t_s embedding for semantics, sc off, full-width heads
    make_case("sem_tj_small", O.OracleCfg(use_tj_instead_of_beta=True, use_tj_for_s=True,
                                          use_separate_tj_for_semantic=True, sc_lambda=0.0,
                                          fc_use_full_features=True, **small), 40, seed=4, epoch=0)
    # E: L_t alone on the default head set (config 3's loss set)
    make_case("sem_cartreg_small", O.OracleCfg(use_car_reg_loss=True, **small), 64, seed=5, epoch=3)
    # F: baseline SatNeRF (config 1 shape family) with the depth-ray pass
    make_case("satnerf_small", O.OracleCfg(model="satnerf", fc_units=32, n_samples=8, render_chunk_size=100),
              40, seed=6, epoch=2, with_depth=True)
    make_case("satnerf_relu_small", O.OracleCfg(model="satnerf", activation_function="relu", fc_units=32, n_samples=8),
              40, seed=7, epoch=0)
    # G: full width (W=512, S=64), few rays; weights regenerate from the seed
    make_case("sem_siren_full", O.OracleCfg(), 16, seed=8, epoch=2, store_params=False, per_sample=False,
              grad_mode="sample", full_grads=FULL_GRADS_SEM, yard_noise=("1e-3", "3e-4"))
    make_case("satnerf_full_c1", O.OracleCfg(model="satnerf", n_samples=32), 16, seed=9, epoch=2,
              store_params=False, per_sample=False, grad_mode="sample")
    # H: seam-3 inference on explicit xyz/z_vals
    inference_case("inference_sem_small", O.OracleCfg(**small), 24, seed=10)
    inference_case("inference_satnerf_small", O.OracleCfg(model="satnerf", fc_units=32, n_samples=8), 24, seed=11)
    make_trajectories()
    make_convergence()


def make_convergence():
    convergence_case("converge_small", O.OracleCfg(fc_units=64, n_samples=32, render_chunk_size=40960), seed=14,
                     n_bank=10240, n_test=1024, batch=256, steps=int(os.environ.get("SNERF_CONV_STEPS", "400")), eval_every=50)


def make_trajectories():
    small = dict(fc_units=32, n_samples=16, render_chunk_size=200)
    c = O.OracleCfg(first_beta_epoch=0, use_car_reg_loss=True, car_reg_loss_start=0, **small)
    save_trajectory("traj25_small", *trajectory_case("traj25_small", c, 48, seed=12, steps=25), 25)
    # the full-width trajectory also from weights with 1e-3 / 1e-4 relative noise: the reference-made yardstick for the one-plane mode at
    # W = 512 (where its trunk runs as one persistent launch)
    c = O.OracleCfg(first_beta_epoch=0)
    fix, traj = trajectory_case("traj10_full", c, 16, seed=13, steps=10)
    noisy = {k: trajectory_case("traj10_full", c, 16, seed=13, steps=10, noise_rel=float(k)) for k in ("1e-3", "1e-4")}
    save_trajectory("traj10_full", fix, traj, 10, noisy)


if __name__ == "__main__":
    main()
