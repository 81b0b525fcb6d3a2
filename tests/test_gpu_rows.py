"""GPU tests that close the SURVEY section-8 rows the round-1 review marked partial, plus the optimiser and
operand-scaling checks it asked for:

  a16  depth-ray branch of the training step (baseline/components/training_step.py:30-54), before and after ds_drop
  f1   GpuRayBank on the device: bit-identical to an index_select of the host copy, dtypes kept, 2-rank union
  f2   batched_inference / lean_inference VALUES against the oracle (chunked, full-frame jitter tensor)
  f4   the device-side validation metrics on the device
  val  validation_step / TrainLoop.validate (base_ray_pipeline.py:101-193) against the oracle + numpy restatements
  adam FlatAdam across first_beta_epoch == torch.optim.Adam (zero gradients, as the reference produces); None-gradient
       skipping and per-parameter step counts against torch.optim.Adam at the optimiser level
  tail heavy-tailed gradients (near-opaque rays next to thousands of almost-silent ones, far from 0.5 to 50)
"""
import os

import numpy as np
import pytest
import torch

from oracle import snerf_oracle as O
from tests.helpers import load_fixture, fixture_batch, max_abs, rel_err, check_validation_metrics
from tests.test_gpu_pipeline import _pipeline_for, _batch_to_dev, DEV, OUT_TOL, LOSS_RTOL, GRAD_REL_TOL, GRAD_ABS_ESCAPE, ROOT

pytestmark = pytest.mark.gpu
ROW_TOL = 4e-6      # per-ray gradient rows down to 2^-20 of the loudest ray (test_heavy_tailed_gradients): measured 3.9e-7 with
                    # per-block exponents (round 1's per-tensor scales: 1.2e-4 on the rows at the floor)


def _rand_sequence(monkeypatch, tensors):
    """the renderer draws its jitter with torch.rand: hand out the fixture's tensors in call order"""
    seq = [t.to(DEV) for t in tensors]

    def fake(*a, **k):
        return seq.pop(0).clone()
    monkeypatch.setattr(torch, "rand", fake)
    return seq


# ---------------------------------------------------------------------------------------------------------------
# a16: depth supervision branch
# ---------------------------------------------------------------------------------------------------------------
def test_training_step_depth_branch_matches_reference_fixture(monkeypatch):
    """depth_enabled = True (the reference default): while train_steps < ds_drop the step renders the depth rays in a
    second pipeline() call, adds DepthLoss and merges loss_dict (satnerf_small.npz stores the reference's coarse_ds,
    total and gradients WITH the depth term); from ds_drop on the term is gone."""
    z, meta, cfg = load_fixture("satnerf_small")
    assert meta["with_depth"]
    b = fixture_batch(z)
    bd = fixture_batch(z, "in_depth_")
    pipe, _ = _pipeline_for(cfg, b["rays"].shape[0], meta["seed"], depth_enabled=True)
    assert pipe.ds_drop == 25            # round(0.25 * max_train_steps = 100), baseline/pipelines/satnerf.py:26-29
    pipe.current_epoch = meta["epoch"]
    batch = _batch_to_dev(b)
    batch["depth"] = {"rays": bd["rays"].to(DEV), "extras": bd["extras"].to(DEV),
                      "depths": b["depths"].to(DEV).view(-1, 1), "weights": b["depth_weights"].to(DEV)}
    seq = _rand_sequence(monkeypatch, [b["u"], bd["u"]])
    out = pipe.training_step(batch, 0)
    assert not seq, "the depth pass did not draw its own jitter"
    assert pipe.logged["train/depth_loss_activated"] == 1.0
    terms = {k[len("train/"):]: float(v) for k, v in pipe.logged.items() if k.startswith("train/coarse_")}
    ref = {k[5:]: float(z[k]) for k in z.files if k.startswith("loss_") and k != "loss_total"}
    assert set(terms) == set(ref) and "coarse_ds" in terms, (sorted(terms), sorted(ref))
    for k, v in ref.items():
        assert abs(terms[k] - v) <= LOSS_RTOL * max(1.0, abs(v)), (k, terms[k], v)
    assert abs(float(out["loss"]) - float(z["loss_total"])) <= LOSS_RTOL * max(1.0, abs(float(z["loss_total"])))
    out["loss"].backward()
    grads = {k: p.grad for k, p in pipe.model_coarse.named_parameters()}
    grads["model_t.weight"] = pipe.model_t.weight.grad
    n = 0
    for k in z.files:
        if k.startswith("grad_"):
            g, r = grads[k[5:]], z[k]
            g = torch.zeros(r.shape) if g is None else g.cpu()
            assert rel_err(g, r) <= GRAD_REL_TOL or max_abs(g, r) <= 1e-7 + GRAD_ABS_ESCAPE * float(np.abs(r).max()), (k, rel_err(g, r))
            n += 1
    assert n >= 20
    # ds_noweights: every depth ray weighs 1 (training_step.py:40-44)
    po = O.to_torch(O.init_params_numpy(cfg, meta["seed"]))
    emb = torch.from_numpy(O.init_embedding_numpy(cfg, meta["seed"]))
    dres = O.render_rays(po, emb, cfg, bd["rays"], bd["extras"], bd["u"])
    want = float(O.depth_loss(dres, b["depths"], torch.ones_like(b["depths"]), cfg)["coarse_ds"])
    pipe.cfgs.pipeline.ds_noweights = True
    pipe.logged.clear()
    _rand_sequence(monkeypatch, [b["u"], bd["u"]])
    pipe.training_step(batch, 1)
    assert abs(float(pipe.logged["train/coarse_ds"]) - want) <= LOSS_RTOL * max(1.0, abs(want))
    pipe.cfgs.pipeline.ds_noweights = False
    # past the drop point: no depth pass (one jitter draw only), no coarse_ds, total = the fixture's total - coarse_ds
    pipe.train_steps = 30
    pipe.logged.clear()
    seq = _rand_sequence(monkeypatch, [b["u"], bd["u"]])
    out2 = pipe.training_step(batch, 2)
    assert len(seq) == 1 and pipe.logged["train/depth_loss_activated"] == 0.0
    assert "train/coarse_ds" not in pipe.logged
    want2 = float(z["loss_total"]) - float(z["loss_coarse_ds"])
    assert abs(float(out2["loss"]) - want2) <= LOSS_RTOL * max(1.0, abs(want2))


# ---------------------------------------------------------------------------------------------------------------
# f1: ray bank on the device
# ---------------------------------------------------------------------------------------------------------------
def test_ray_bank_on_device():
    from snerf_amd.framework.datasets import GpuRayBank
    host = GpuRayBank.synthetic(5000, n_images=7, seed=3)
    bank = GpuRayBank.synthetic(5000, n_images=7, seed=3, device=DEV)
    assert all(v.device.type == "cuda" for v in bank.t.values())
    for step in (0, 3, 26):                       # 26: second epoch -> a new device-drawn permutation
        got = bank.batch(step, 192)
        assert bank._perm.device.type == "cuda" and bank._perm.dtype == torch.int64
        it = step % bank.steps_per_epoch(192)
        idx = bank._perm[it * 192:(it + 1) * 192].cpu()
        assert sorted(bank._perm.cpu().tolist()) == list(range(5000))       # a permutation
        for k, v in got.items():
            assert v.device.type == "cuda" and v.dtype == host.t[k].dtype, k
            assert torch.equal(v.cpu(), host.t[k].index_select(0, idx)), k   # bit for bit the host rows
        assert got["semantic"].dtype == torch.uint8 and got["semantic_sparsity_mask"].dtype == torch.bool
        parts = [bank.batch(step, 192, r, 2) for r in range(2)]               # two ranks: union == global batch
        for k in got:
            assert torch.equal(torch.cat([p[k] for p in parts], 0), got[k]), k
    assert not torch.equal(bank.batch(0, 192)["rays"], bank.batch(26, 192)["rays"])
    # another bank with the same seed draws the same permutation (what every rank relies on)
    twin = GpuRayBank.synthetic(5000, n_images=7, seed=3, device=DEV)
    assert torch.equal(twin.batch(26, 192)["rays"], bank.batch(26, 192)["rays"])
    unshuffled = bank.batch(2, 192, shuffle=False)
    assert torch.equal(unshuffled["rays"].cpu(), host.t["rays"][384:576])
    img = bank.image(1, 1000, 1, 3)               # validation image 1, rank 1 of 3: rows 1334..1667
    assert torch.equal(img["rays"].cpu(), host.t["rays"][1334:1668])


# ---------------------------------------------------------------------------------------------------------------
# f2: full-frame inference values
# ---------------------------------------------------------------------------------------------------------------
def test_batched_and_lean_inference_values_vs_oracle():
    """eval/utils/util.py:13-42 at a chunk size that cuts the frame into ragged chunks, jitter pinned for the whole
    frame: EVERY returned tensor against the oracle (1e-4; labels exact outside the margin), lean == batched."""
    from snerf_amd.eval.utils.util import batched_inference, lean_inference
    cfg = O.OracleCfg(fc_units=64, n_samples=24, render_chunk_size=100)
    pipe, params = _pipeline_for(cfg, 64, 5)
    b = O.batch_to_torch(O.synthetic_batch(333, 24, seed=15))
    rays, extras, u = b["rays"].to(DEV), b["extras"].to(DEV), b["u"].to(DEV)
    ro = {"perturb_rand": u}
    bi = batched_inference(pipe.cfgs, pipe.renderer, pipe.models, rays, extras, render_options=ro)
    ora = O.render_rays(O.to_torch(params), torch.from_numpy(O.init_embedding_numpy(cfg, 5)), cfg, b["rays"], b["extras"], b["u"])
    ora.pop("_z_vals")
    assert set(bi) == set(ora)
    for k, v in ora.items():
        assert bi[k].shape == v.shape and not bi[k].requires_grad, k
        if k == "semantic_label_coarse":
            top2 = ora["semantic_logits_coarse"].topk(2, dim=-1).values
            sure = (top2[:, 0] - top2[:, 1]) > 2 * OUT_TOL
            assert bi[k].dtype == torch.int64 and torch.equal(bi[k].cpu()[sure], v[sure])
        else:
            assert max_abs(bi[k].cpu(), v) <= OUT_TOL, (k, max_abs(bi[k].cpu(), v))
    keys = tuple(k for k in ora)
    lean = lean_inference(pipe.cfgs, pipe.renderer, pipe.models, rays, extras, keys=keys, render_options=ro)
    for k in keys:
        assert torch.equal(lean[k], bi[k]), k


_SHARD_WORKER = r"""
import os, sys
sys.path.insert(0, {root!r})
import torch, torch.distributed as dist
from snerf_amd import parallel
rank, world, dev = parallel.init_distributed(backend="gloo")
from tests.test_gpu_rows import _frame_case
from snerf_amd.eval.utils.util import sharded_lean_inference
from snerf_amd.eval.extract_pointcloud import extract_pointcloud
pipe, rays, extras, ro = _frame_case()
res = sharded_lean_inference(pipe.cfgs, pipe.renderer, pipe.models, rays, extras, keys=("rgb_coarse", "depth_coarse", "semantic_label_coarse", "weights_coarse"), render_options=ro)
pc = extract_pointcloud(pipe.cfgs, pipe.renderer, pipe.models, rays, extras, render_options=ro, sharded=True)
lo, hi = res.pop("_rows")
assert (lo, hi) == parallel.frame_shard(rays.shape[0]) and res["weights_coarse"].shape[0] == hi - lo
torch.save({{"res": {{k: v.cpu() for k, v in res.items()}}, "rows": (lo, hi), "pc": {{k: v.cpu() for k, v in pc.items()}}}}, {out!r} + str(rank))
dist.barrier()
"""


def _frame_case():
    cfg = O.OracleCfg(fc_units=64, n_samples=24, render_chunk_size=100)
    pipe, _ = _pipeline_for(cfg, 64, 5)
    b = O.batch_to_torch(O.synthetic_batch(333, 24, seed=15))       # 333 rays over 2 ranks: 167 + 166, chunks of 100
    return pipe, b["rays"].to(DEV), b["extras"].to(DEV), {"perturb_rand": b["u"].to(DEV)}


def test_rank_sharded_full_frame_inference_equals_single_rank(tmp_path):
    """SURVEY 8(e), last row: lean_inference / extract_pointcloud with the frame's rays sharded over 2 ranks (gloo, both on this
    GPU): each rank renders its contiguous slice, rgb / depth / label are all-gathered -- the frame EVERY rank ends up with
    equals the single-rank frame bit for bit; per-sample results stay local (this rank's rows)."""
    import subprocess, sys
    from snerf_amd.eval.utils.util import lean_inference
    from snerf_amd.eval.extract_pointcloud import extract_pointcloud
    pipe, rays, extras, ro = _frame_case()
    keys = ("rgb_coarse", "depth_coarse", "semantic_label_coarse", "weights_coarse")
    one = lean_inference(pipe.cfgs, pipe.renderer, pipe.models, rays, extras, keys=keys, render_options=ro)
    pc1 = extract_pointcloud(pipe.cfgs, pipe.renderer, pipe.models, rays, extras, render_options=ro)
    res = str(tmp_path / "shard.pt")
    script = tmp_path / "worker.py"
    script.write_text(_SHARD_WORKER.format(root=ROOT, out=res))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29641", WORLD_SIZE="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK="0"), cwd=ROOT) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=300) == 0
    seen = 0
    for r in range(2):
        two = torch.load(res + str(r), weights_only=True)
        lo, hi = two["rows"]
        seen += hi - lo
        for k in ("rgb_coarse", "depth_coarse", "semantic_label_coarse"):
            assert torch.equal(two["res"][k], one[k].cpu()), (r, k)
        assert torch.equal(two["res"]["weights_coarse"], one["weights_coarse"].cpu()[lo:hi])
        for k, v in pc1.items():
            assert torch.equal(two["pc"][k], v.cpu()), (r, k)
    assert seen == rays.shape[0]


# ---------------------------------------------------------------------------------------------------------------
# f4: validation metrics on the device
# ---------------------------------------------------------------------------------------------------------------
def test_validation_metrics_on_device():
    """the same definitions check as the CPU test, with every tensor on the GPU (bincount confusion matrix, masked PSNR
    reduction).  parity unpinned: torchmetrics / kornia are absent, so the yardstick is the numpy restatement of
    semantic/components/metrics.py:11-87 and eval/utils/metrics.py:8-18."""
    check_validation_metrics(DEV)


# ---------------------------------------------------------------------------------------------------------------
# validation step
# ---------------------------------------------------------------------------------------------------------------
def test_validation_step_and_loop_vs_oracle(monkeypatch):
    """a synthetic 64 x 64 "image" through pipeline.validation_step: loss_dict + total, PSNR, semantic accuracy,
    confusion counts, mIoU and uncertainty-at-transient equal the oracle's render of the same rays + numpy
    restatements of the metric definitions; TrainLoop.validate averages the images and run_pipeline calls it at the
    epoch boundary named by check_val_every_n_epoch."""
    from snerf_amd.framework.pipelines import TrainLoop, run_pipeline
    from snerf_amd.framework.datasets import GpuRayBank
    cfg = O.OracleCfg(fc_units=64, n_samples=16, first_beta_epoch=0, render_chunk_size=1500)
    HW = 64 * 64
    pipe, params = _pipeline_for(cfg, 256, 7)
    b = O.batch_to_torch(O.synthetic_batch(2 * HW, 16, seed=31, car_prob=0.05))
    pipe.datasets["rgb_test"] = GpuRayBank({"rays": b["rays"], "extras": b["extras"], "rgbs": b["rgbs"],
                                            "semantic": b["semantic"].to(torch.uint8),
                                            "semantic_sparsity_mask": b["mask"]}, device=DEV)
    u = b["u"].to(DEV)
    state = {"img": 0}
    monkeypatch.setattr(pipe, "_val_render_options", lambda split: {"perturb_rand": u[state["img"] * HW:(state["img"] + 1) * HW]})
    po = O.to_torch(params)
    emb = torch.from_numpy(O.init_embedding_numpy(cfg, 7))
    per_image = []
    for i in range(2):
        state["img"] = i
        sl = slice(i * HW, (i + 1) * HW)
        batch = dict(pipe.datasets["rgb_test"].image(i, HW), split="test")
        out = pipe.validation_step(batch, i)
        ora = O.render_rays(po, emb, cfg, b["rays"][sl], b["extras"][sl], b["u"][sl])
        ld = O.satnerf_loss(ora, b["rgbs"][sl], cfg)
        for k, v in ld.items():
            assert abs(float(out[k]) - float(v)) <= LOSS_RTOL * max(1.0, abs(float(v))), (k, float(out[k]), float(v))
        tot = float(sum(ld.values()))
        assert abs(float(out["loss"]) - tot) <= LOSS_RTOL * max(1.0, abs(tot))
        want_psnr = float(-10 * torch.log10(((ora["rgb_coarse"] - b["rgbs"][sl]) ** 2).mean()))
        assert abs(float(out["psnr"]) - want_psnr) <= 1e-3
        assert max_abs(out["results"]["rgb_coarse"].cpu(), ora["rgb_coarse"]) <= OUT_TOL
        # metrics: definitions restated in numpy on the HIP labels (exact), and the labels themselves vs the oracle
        pred = out["results"]["semantic_label_coarse"].cpu().numpy()
        gt = b["semantic"][sl, 0].numpy()
        agree = float((pred == ora["semantic_label_coarse"].numpy()).mean())
        assert agree >= 0.995, agree
        counts = np.zeros((5, 5))
        np.add.at(counts, (gt, pred), 1)
        assert np.array_equal(out["confusion_counts"].cpu().numpy(), counts)
        assert abs(float(out["semantic_accuracy"]) - float((gt == pred).mean())) <= 1e-6
        iou = [counts[c, c] / (counts[c].sum() + counts[:, c].sum() - counts[c, c]) if (counts[c].sum() + counts[:, c].sum()) > 0
               else np.nan for c in range(5)]
        assert abs(float(out["mIoU"]) - float(np.nanmean(iou))) <= 1e-9
        comp = (ora["weights_coarse"].unsqueeze(-1) * ora["beta_coarse"]).sum(-2)[:, 0].numpy()
        car = gt == 4
        assert car.any() and abs(float(out["uncertainty_at_transient"]) - comp[car].mean()) <= 1e-4
        per_image.append((float(out["loss"]), float(out["psnr"]), float(out["mIoU"]), float(out["semantic_accuracy"]), counts))
    assert abs(float(pipe.logged["test/psnr"]) - per_image[1][1]) <= 1e-6
    # the loop: means over the images, split-wide confusion matrix
    cfgs = pipe.cfgs
    loop = TrainLoop(pipe, cfgs, torch.device(DEV))
    calls = {"n": 0}

    def opts(split):
        i = calls["n"]
        calls["n"] += 1
        return {"perturb_rand": u[i * HW:(i + 1) * HW]}
    monkeypatch.setattr(pipe, "_val_render_options", opts)
    val = loop.validate(rays_per_image=HW)
    assert calls["n"] == 2
    for j, k in enumerate(("test/loss", "test/psnr", "test/mIoU", "test/semantic_accuracy")):
        want = 0.5 * (per_image[0][j] + per_image[1][j])
        assert abs(val[k] - want) <= 1e-4 * max(1.0, abs(want)), (k, val[k], want)
    tot_counts = per_image[0][4] + per_image[1][4]
    rows = tot_counts.sum(1, keepdims=True)
    assert np.allclose(val["test/confusion_matrix"].numpy(), np.divide(tot_counts, rows, out=np.zeros_like(tot_counts), where=rows > 0), atol=1e-6)
    # run_pipeline validates after every check_val_every_n_epoch-th epoch (framework/pipelines.py:316-318)
    monkeypatch.setattr(pipe, "_val_render_options", lambda split: {"perturb": 0})
    seen = []
    cfgs.run.check_val_every_n_epoch = 2
    spe = loop.steps_per_epoch                     # 2048 synthetic rays / batch 256 = 8
    run_pipeline(pipe, cfgs, torch.device(DEV), max_steps=4 * spe, on_validation=lambda step, v: seen.append((step, v)))
    assert [s for s, _ in seen] == [2 * spe - 1, 4 * spe - 1]
    assert all(np.isfinite(v["test/psnr"]) and "test/mIoU_split" in v for _, v in seen)


# ---------------------------------------------------------------------------------------------------------------
# Adam across first_beta_epoch
# ---------------------------------------------------------------------------------------------------------------
def test_fused_adam_across_first_beta_epoch_equals_torch_adam(monkeypatch):
    """Default config: first_beta_epoch = 2.  In the REFERENCE the beta head and the transient embedding receive exact
    ZERO gradients (not None) during epochs 0-1 -- the model returns one concatenated tensor that inference() slices
    (semantic/models/rs_semantic.py:71-96) -- so torch.optim.Adam steps them with a zero update and their step counters
    run with everybody else's (probed on the imported reference: all 45 parameters carry state after one step at
    epoch 0).  The build does the same: the trajectory across the switch equals stock torch.optim.Adam driven by the
    same gradients, the late parameters do not move before the switch, and all step counts are equal."""
    from snerf_amd.framework.pipelines import TrainLoop
    cfg = O.OracleCfg(fc_units=32, n_samples=8, first_beta_epoch=2)

    def run(torch_adam, steps):
        if torch_adam:
            monkeypatch.setenv("SNERF_TORCH_ADAM", "1")
        else:
            monkeypatch.delenv("SNERF_TORCH_ADAM", raising=False)
        pipe, _ = _pipeline_for(cfg, 512, 4, max_steps=1000)      # 2048 rays / 512 -> 4 steps per epoch
        loop = TrainLoop(pipe, pipe.cfgs, torch.device(DEV))
        torch.manual_seed(0)
        for s in range(steps):
            loop.step(s)
        return pipe, loop

    fresh, _ = _pipeline_for(cfg, 512, 4, max_steps=1000)
    late = ["model_coarse.beta_from_xyz.0.weight", "model_coarse.beta_from_xyz.0.bias", "model_coarse.beta_from_xyz.2.weight",
            "model_coarse.beta_from_xyz.2.bias", "model_t.weight"]
    p8, l8 = run(False, 8)                                          # end of epoch 1: beta loss still off
    for n in late:
        assert torch.equal(dict(p8.named_parameters())[n], dict(fresh.named_parameters())[n]), n   # zero gradient: no movement
        i = [k for k, _ in p8.named_parameters()].index(n)
        assert float(l8.optimizer.flat_g[l8.optimizer.offsets[i]:l8.optimizer.offsets[i] + 4].abs().max()) == 0.0
    assert set(l8.optimizer.steps) == {8}                           # ... but their steps are counted, as torch does
    n_steps = 11
    pt, lt = run(True, n_steps)
    ph, lh = run(False, n_steps)
    st = lt.optimizer.state_dict()["state"]
    assert len(st) == len(lh.optimizer.params) and {int(float(v["step"])) for v in st.values()} == {n_steps}
    assert set(lh.optimizer.steps) == {n_steps}
    for n, p in ph.named_parameters():
        ref = dict(pt.named_parameters())[n]
        assert max_abs(p.detach().cpu(), ref.detach().cpu()) <= 2e-6 + 1e-4 * float(ref.abs().max()), n
    moved = (dict(ph.named_parameters())[late[3]] - dict(fresh.named_parameters())[late[3]]).abs()
    assert float(moved.max()) > 1e-5                                # the beta head trains once its loss is on


def test_fused_adam_none_gradients_and_per_parameter_steps():
    """FlatAdam keeps torch.optim.Adam's per-parameter bookkeeping: a parameter whose .grad is None is skipped (no
    moment update, no step), state_dict() omits never-stepped parameters and carries one `step` per parameter, and
    load_state_dict() accepts differing / missing steps (a checkpoint torch.optim.Adam wrote, and vice versa)."""
    from snerf_amd.optim import FlatAdam
    torch.manual_seed(5)
    shapes = [(64, 48), (48,), (7, 3), (5,), (33, 9)]
    base = [torch.randn(s, device=DEV) for s in shapes]
    ph = [torch.nn.Parameter(t.clone()) for t in base]
    pt = [torch.nn.Parameter(t.clone()) for t in base]
    oh, ot = FlatAdam(ph, lr=1e-2), torch.optim.Adam(pt, lr=1e-2)
    quiet = {2, 3}                                                  # no gradient during the first 4 steps
    for step in range(9):
        oh.zero_grad()
        ot.zero_grad()
        for i, (a, b) in enumerate(zip(ph, pt)):
            if step < 4 and i in quiet:
                continue
            g = torch.randn(a.shape, device=DEV, generator=None) * (0.1 + i)
            a.grad = g.clone()
            b.grad = g.clone()
        oh.step()
        ot.step()
        for a, b in zip(ph, pt):
            assert max_abs(a.detach().cpu(), b.detach().cpu()) <= 1e-6 * (1 + float(b.abs().max())), step
        if step == 2:
            sd = oh.state_dict()
            assert sorted(sd["state"]) == [0, 1, 4] and sorted(ot.state_dict()["state"]) == [0, 1, 4]
    assert oh.steps == [9, 9, 5, 5, 9] and oh.step_count == 9
    tsd = ot.state_dict()
    assert [int(float(tsd["state"][i]["step"])) for i in range(5)] == oh.steps
    hsd = oh.state_dict()
    for i in range(5):
        assert rel_err(hsd["state"][i]["exp_avg"].cpu(), tsd["state"][i]["exp_avg"].cpu()) <= 1e-4
        assert rel_err(hsd["state"][i]["exp_avg_sq"].cpu(), tsd["state"][i]["exp_avg_sq"].cpu()) <= 1e-4
    # torch's checkpoint into FlatAdam and FlatAdam's into torch: the next step agrees again
    p2 = [torch.nn.Parameter(t.detach().clone()) for t in pt]
    o2 = FlatAdam(p2, lr=1e-2)
    o2.load_state_dict(tsd)
    assert o2.steps == oh.steps
    p3 = [torch.nn.Parameter(t.detach().clone()) for t in pt]
    o3 = torch.optim.Adam(p3, lr=1e-2)
    o3.load_state_dict(hsd)
    for ps, o in ((p2, o2), (p3, o3), (pt, ot)):
        for i, p in enumerate(ps):
            p.grad = torch.full(p.shape, 0.01 * (i + 1), device=DEV)
        o.step()
    for a, b, c in zip(p2, p3, pt):
        assert max_abs(a.detach().cpu(), c.detach().cpu()) <= 1e-6 * (1 + float(c.abs().max()))
        assert max_abs(b.detach().cpu(), c.detach().cpu()) <= 1e-6 * (1 + float(c.abs().max()))
    early = FlatAdam([torch.nn.Parameter(t.clone()) for t in base], lr=1e-2)
    early.load_state_dict(sd)                                       # written before parameters 2, 3 had a gradient
    assert early.steps == [3, 3, 0, 0, 3]


# ---------------------------------------------------------------------------------------------------------------
# heavy-tailed gradients
# ---------------------------------------------------------------------------------------------------------------
def test_heavy_tailed_gradients():
    """A batch that stresses the fp16 operand scaling: far bounds from 0.5 to 50 (sample spacing, hence alpha, spans two
    orders of magnitude: near-opaque rays next to almost empty ones) and per-ray loss weights from 1 down to 1e-9 (a few
    loud rays among thousands of almost-silent ones), at the headline width.  Required: every parameter gradient within
    2e-3 relative L2 of the fp32 oracle, and the PER-RAY input gradient d loss / d t (the rows of dX of the head layer,
    summed over a ray's samples) of every ray down to 2^-20 of the loudest within 4e-6 relative of the oracle -- i.e. a
    silent ray's gradient row keeps its own precision next to a loud one."""
    from tests.test_gpu_kernels import _gpu_params, _hip_render, _dev
    dev = _dev()
    cfg = O.OracleCfg(n_samples=64)
    N, n_sub = 2048, 192
    pn = O.init_params_numpy(cfg, 41)
    emb_np = O.init_embedding_numpy(cfg, 41)
    bn = O.synthetic_batch(N, 64, seed=42)
    rng = np.random.default_rng(43)
    bn["rays"][:, 7] = np.exp(rng.uniform(np.log(0.5), np.log(50.0), N)).astype(np.float32)
    b = O.batch_to_torch(bn)
    idx = torch.arange(0, N, N // n_sub)[:n_sub]
    wts = torch.from_numpy(np.exp(rng.uniform(np.log(1e-9), 0.0, n_sub)).astype(np.float32))
    wts[:4] = 1.0                                                             # a few loud rays
    wts[4:8] = 2.0 ** -20                                                     # and rows right at the stated floor

    def loss_of(res, w, t_rows):
        col = ((res["rgb_coarse"] - 0.3) ** 2).sum(-1) + 0.1 * (res["weights_coarse"].unsqueeze(-1) * res["beta_coarse"]).sum((-1, -2)) \
            + 0.05 * res["semantic_logits_coarse"].square().sum(-1) + 0.01 * res["sun_sc_coarse"].sum((-1, -2))
        return (w * col).sum()

    gp = _gpu_params(pn, dev, requires_grad=True)
    emb_g = torch.from_numpy(emb_np).to(dev).requires_grad_(True)
    ts = b["extras"][:, 3].long()
    t_g = emb_g[ts.to(dev)].detach().requires_grad_(True)                      # per-ray t rows as a leaf: d loss / d t per ray

    class _Rows:                                                             # _hip_render indexes the embedding by ts
        def __getitem__(self, _):
            return t_g
    hip = _hip_render(cfg, gp, _Rows(), b, dev)
    hip.pop("_z_vals")
    loss_of({k: v[idx.to(dev)] for k, v in hip.items() if k != "semantic_label_coarse"}, wts.to(dev), None).backward()

    po = O.to_torch(pn, requires_grad=True)
    t_o = torch.from_numpy(emb_np)[ts[idx]].clone().requires_grad_(True)

    class _RowsO:
        def __getitem__(self, _):
            return t_o
    # the oracle's render_rays also looks the embedding up by ts: give it the same per-ray leaf
    ora = O.render_rays(po, _RowsO(), cfg, b["rays"][idx], b["extras"][idx], b["u"][idx])
    ora.pop("_z_vals")
    t_end = ora["transparency_coarse"].detach()[:, -1]          # light left in front of the last sample
    assert float(t_end.min()) < 1e-6 and float(t_end.max()) > 0.5, "the batch should mix opaque and thin rays"
    loss_of({k: v for k, v in ora.items() if k != "semantic_label_coarse"}, wts, None).backward()
    for k in po:
        err = rel_err(gp[k].grad.cpu(), po[k].grad)
        assert err <= GRAD_REL_TOL, (k, err)
    dt_h, dt_o = t_g.grad[idx.to(dev)].cpu().double(), t_o.grad.double()
    row = dt_o.norm(dim=1)
    live = row >= row.max() * 2.0 ** -20
    assert int(live.sum()) >= 40
    row_err = (dt_h - dt_o).norm(dim=1)[live] / row[live]
    # per-(128 x 128)-block exponents: max 3.9e-7, median 2.0e-7 (per-tensor scales, round 1: 1.2e-4 on the rows at the floor)
    print("heavy-tail per-ray gradient rows: max rel err %.3g, median %.3g over %d live rows" % (float(row_err.max()), float(row_err.median()), int(live.sum())))
    assert float(row_err.max()) <= ROW_TOL, (float(row_err.max()), float(row_err.median()))
    # rays outside the subset got exactly zero
    mask = torch.ones(N, dtype=torch.bool)
    mask[idx] = False
    assert float(t_g.grad.cpu()[mask].abs().max()) == 0.0


@pytest.mark.parametrize("vocab", [50, 96])
def test_embedding_rows_forward_backward(vocab):
    """ops.embed_rows == nn.Embedding (reference: models["t"](ts), semantic/components/rendering.py:35-46): rows bit-equal,
    the table gradient equal to torch's scatter-add to fp32 summation order, and bit-identical from run to run.  vocab 96:
    BASELINE configs[4] raises `t_embedding_vocab` to cover four scenes' images."""
    from snerf_amd import ops
    torch.manual_seed(7)
    emb = torch.nn.Embedding(vocab, 4).to(DEV)
    ref = torch.nn.Embedding(vocab, 4).to(DEV)
    ref.load_state_dict(emb.state_dict())
    idx = torch.randint(0, vocab, (4099,), device=DEV)
    idx[:7] = vocab - 1
    w = torch.randn(4099, 4, device=DEV)
    rows = ops.embed_rows(emb, idx)
    want = ref(idx)
    assert torch.equal(rows, want)
    (rows * w).sum().backward()
    (want * w).sum().backward()
    g1 = emb.weight.grad.clone()
    assert max_abs(g1.cpu(), ref.weight.grad.cpu()) <= 1e-5 * float(ref.weight.grad.abs().max())
    emb.weight.grad = None
    (ops.embed_rows(emb, idx) * w).sum().backward()
    assert torch.equal(emb.weight.grad, g1)
    # float indices as the extras column carries them (rays' image index), through the renderer's own conversion
    assert torch.equal(ops.embed_rows(emb, idx.float().long()), want)
    # an index outside the table: torch raises; the kernel marks the row (NaN), it does not invent zeros
    bad = ops.embed_rows(emb, torch.tensor([0, vocab, -1], device=DEV))
    assert torch.equal(bad[0], want.new_tensor(emb.weight[0].tolist())) and bool(torch.isnan(bad[1:]).all())
