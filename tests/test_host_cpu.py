"""CPU tests of the host-side mirror: ray column access, TOML/pydantic config chain, plugin loader,
GPU-ray-bank sharding and the multi-process (gloo, world_size 2) gradient all-reduce path."""
import os
import subprocess
import sys

import pytest


def test_mfma_mode_mapping():
    """precision / float32_matmul_precision / mfma_precision -> matrix arithmetic (ops.mfma_mode)"""
    from types import SimpleNamespace as NS
    from snerf_amd import ops, _lib
    assert ops.mfma_mode(NS(precision=32), None) == "f16x2"
    assert ops.mfma_mode(NS(precision=16), None) == "bf16"
    assert ops.mfma_mode(NS(precision=32, mfma_precision="fp32"), None) == "fp32"
    for run, want in (("highest", "split3"), ("high", "split2"), ("medium", "bf16")):
        assert ops.mfma_mode(NS(precision=32, mfma_precision="auto"), NS(float32_matmul_precision=run)) == want
    d = ops.ModelSpec(mfma="split2").desc(16, 8)
    assert d.flags & _lib.FLAG_BF16X3
    assert ops.ModelSpec().desc(16, 8).flags == _lib.FLAG_F16X2          # default: fp32-class on fp16 planes
    assert ops.ModelSpec(mfma="split3").desc(16, 8).flags == 0

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_ray_component_names():
    from snerf_amd.framework.components.rays import ray_component_fn, extras_component_fn
    rays = torch.arange(16.).reshape(2, 8)
    assert ray_component_fn(rays, "origins").tolist() == [[0, 1, 2], [8, 9, 10]]
    assert ray_component_fn(rays, "directions").shape == (2, 3)
    assert ray_component_fn(rays, "near").tolist() == [[6], [14]]
    assert ray_component_fn(rays, "fars").tolist() == [[7], [15]]
    ray_component_fn(rays, "far", value=torch.ones(2, 1))
    assert rays[:, 7].tolist() == [1, 1]
    ex = torch.arange(8.).reshape(2, 4)
    assert extras_component_fn(ex, "sun_d").shape == (2, 3) and extras_component_fn(ex, "ts").tolist() == [[3], [7]]
    with pytest.raises(KeyError):
        ray_component_fn(rays, "bogus")


def test_config_chain_and_plugin_loader(tmp_path):
    from snerf_amd.framework.configs import load_configs
    from snerf_amd.framework.pipelines import load_pipeline
    run = tmp_path / "run.toml"
    run.write_text('max_train_steps = 400\nshuffle_dataset = true\nsynthetic_rays = 4096\ndataset_name = "JAX_068"\n')
    pl = tmp_path / "pipeline.toml"
    pl.write_text('pipeline = "snerf_amd.semantic.pipelines.rs_semantic.RSSemanticPipeline"\nn_samples = 64\n'
                  'batch_size = 1024\nsc_lambda = 0.05\nignore_car_index = true\nuse_car_reg_loss = true\nlambda_c = 0.1\n'
                  'fc_skips = [4]\nactivation_function = "siren"\n')
    cfgs = load_configs(str(run), str(pl))
    pc = cfgs.pipeline
    # defaults of the reference's config classes (nerf.py:63-88, snerf.py:67-68, satnerf.py:115-124, rs_semantic.py:125-141)
    assert (pc.fc_units, pc.fc_layers, pc.learnrate, pc.t_embedding_tau, pc.first_beta_epoch) == (512, 8, 5e-4, 4, 2)
    assert pc.lambda_s == 0.04 and pc.car_reg_loss_start == 3 and pc.depth_supervision_drop == 0.25
    pipe = load_pipeline(cfgs)
    assert type(pipe).__name__ == "RSSemanticPipeline" and pipe.ds_drop == 100
    keys = list(pipe.state_dict().keys())
    assert keys[0].startswith("model_coarse.fc_net.0.") and "model_t.weight" in keys
    assert hasattr(pipe, "car_reg_loss") and hasattr(pipe, "uncertainty_semantic_loss")
    with pytest.raises(FileNotFoundError):
        load_configs(str(tmp_path / "missing.toml"), str(pl))
    with pytest.raises(Exception):
        from snerf_amd.framework.configs import MainConfig
        MainConfig(run={}, pipeline={"pipeline": "snerf_amd.semantic.pipelines.rs_semantic.RSSemanticPipeline",
                                     "activation_function": "tanh"})


def test_ray_bank_sharding_covers_global_batch():
    from snerf_amd.framework.datasets import GpuRayBank, shard_bounds
    bank = GpuRayBank.synthetic(1000, n_images=5, seed=3)
    assert bank.steps_per_epoch(96) == 10
    full = bank.batch(7, 96)
    parts = [bank.batch(7, 96, r, 4) for r in range(4)]
    for k in full:
        assert torch.equal(torch.cat([p[k] for p in parts], 0), full[k]), k
    assert full["semantic"].dtype == torch.uint8 and full["semantic_sparsity_mask"].dtype == torch.bool
    # a new epoch reshuffles; the same (epoch, step) is reproducible
    assert not torch.equal(bank.batch(7, 96)["rays"], bank.batch(17, 96)["rays"])
    assert torch.equal(bank.batch(17, 96)["rays"], GpuRayBank.synthetic(1000, n_images=5, seed=3).batch(17, 96)["rays"])
    with pytest.raises(ValueError):
        shard_bounds(10, 0, 4)


_WORKER = r"""
import sys
sys.path.insert(0, {root!r})
import torch, torch.distributed as dist
from snerf_amd import parallel
rank, world, dev = parallel.init_distributed(backend="gloo")
assert world == 2 and dev.type == "cpu"
torch.manual_seed(0)
ps = [torch.nn.Parameter(torch.zeros(s)) for s in ((3, 5), (7,), (2, 2, 2))]
for i, p in enumerate(ps):
    p.grad = torch.full_like(p, float(rank + 1) * (i + 1))
buf = parallel.allreduce_gradients(ps)
for i, p in enumerate(ps):
    assert torch.equal(p.grad, torch.full_like(p, 3.0 * (i + 1))), p.grad
assert buf.numel() == 15 + 7 + 8
t = torch.tensor([1.0, 2.0]) * (rank + 1)
assert parallel.allreduce_sum_(t).tolist() == [3.0, 6.0]
from snerf_amd.framework.datasets import GpuRayBank
bank = GpuRayBank.synthetic(512, seed=1)
mine = bank.batch(3, 64, rank, world)["rays"]
gathered = [torch.empty_like(mine) for _ in range(world)]
dist.all_gather(gathered, mine)
assert torch.equal(torch.cat(gathered, 0), bank.batch(3, 64)["rays"])
dist.barrier()
print("rank", rank, "ok")
"""


def test_two_rank_gloo_gradient_allreduce_and_sharding(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(_WORKER.format(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29633", WORLD_SIZE="2", CUDA_VISIBLE_DEVICES="",
               HIP_VISIBLE_DEVICES="")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), cwd=ROOT)
             for r in range(2)]
    for p in procs:
        assert p.wait(timeout=240) == 0


def test_model_forward_is_fused_only():
    """the per-point nn.Module forward is not a fallback path: it refuses to run"""
    import types
    from snerf_amd.baseline.models.satnerf import SatNeRF
    pc = types.SimpleNamespace(fc_use_full_features=False)
    m = SatNeRF(types.SimpleNamespace(pipeline=pc), feat=32, t_embedding_dims=4)
    with pytest.raises(NotImplementedError):
        m(torch.zeros(4, 3))


def test_checkpoint_utilities_cpu(tmp_path):
    """Lightning-layout reader (framework/util/load_ckpoint.py): model-prefix extraction, epoch lookup, and the
    safe-loader rule -- a checkpoint that needs unpickling of arbitrary objects is refused, never executed."""
    import torch
    from snerf_amd.framework.util import load_ckpoint as lc
    d = tmp_path / "logs" / "ckpoints"
    d.mkdir(parents=True)
    sd = {"model_coarse.fc_net.0.weight": torch.ones(2, 3), "model_coarse.sigma_from_xyz.0.bias": torch.zeros(1),
          "model_t.weight": torch.full((4, 2), 2.0)}
    for ep in (0, 3, 12):
        torch.save({"epoch": ep, "global_step": 10 * ep, "state_dict": sd}, d / f"epoch={ep}.ckpt")
    fp, ep = lc.find_ckpoint_fp(str(tmp_path / "logs"), -1)
    assert ep == 12 and fp.endswith("epoch=12.ckpt")
    assert lc.find_ckpoint_fp(str(tmp_path / "logs"), 3)[0].endswith("epoch=3.ckpt")
    assert lc.read_ckpt_info(fp) == (12, 120)
    got = lc.extract_model_state_dict(fp, "model_coarse")
    assert sorted(got) == ["fc_net.0.weight", "sigma_from_xyz.0.bias"]
    assert list(lc.extract_model_state_dict(fp, "model_t")) == ["weight"]
    assert list(lc.extract_model_state_dict(fp, "model_coarse", prefixes_to_ignore=["fc_net"])) == ["sigma_from_xyz.0.bias"]
    m = torch.nn.Module()
    m.weight = torch.nn.Parameter(torch.zeros(4, 2))
    assert lc.load_ckpoint(m, fp, "model_t") == [] and float(m.weight.sum()) == 16.0
    m.weight = torch.nn.Parameter(torch.zeros(5, 2))
    with pytest.raises(RuntimeError, match="size mismatch"):
        lc.load_ckpoint(m, fp, "model_t")

    # any object a pickle would have to instantiate (here: a non-allow-listed class) makes the safe loader refuse
    import fractions
    torch.save({"epoch": 0, "global_step": 0, "state_dict": sd, "hyper_parameters": fractions.Fraction(1, 3)}, d / "last.ckpt")
    with pytest.raises(RuntimeError, match="weights_only"):
        lc.read_ckpt_info(str(d / "last.ckpt"))


def test_validation_metrics_match_reference_definitions():
    """semantic_error / accuracy (filter_idx rows count as right, denominator = all rays), row-normalised confusion
    matrix, mIoU with nanmean over absent classes, beta-at-transient, masked PSNR -- against plain numpy restatements
    of semantic/components/metrics.py:11-87 and eval/utils/metrics.py:8-18."""
    import numpy as np
    import torch
    from snerf_amd.semantic.components import metrics as M
    from snerf_amd.eval.utils.metrics import mse, psnr
    rng = np.random.default_rng(5)
    N, S, Cn = 500, 8, 6                       # class 5 never occurs: NaN IoU -> skipped
    gt = rng.integers(0, 5, size=(N, 1)).astype(np.uint8)
    pred = np.where(rng.random(N) < 0.7, gt[:, 0], rng.integers(0, 5, size=N)).astype(np.int64)
    res = {"semantic_label_coarse": torch.from_numpy(pred), "rgb_coarse": torch.zeros(N, 3),
           "weights_coarse": torch.from_numpy(rng.random((N, S)).astype(np.float32)),
           "beta_coarse": torch.from_numpy(rng.random((N, S, 1)).astype(np.float32))}
    tg = torch.from_numpy(gt)
    err = (gt[:, 0] != pred).astype(np.float32)
    assert abs(float(M.semantic_accuracy(res, tg)) - (1 - err.sum() / N)) < 1e-6
    err4 = np.where(gt[:, 0] == 4, 0.0, err)
    assert abs(float(M.semantic_accuracy(res, tg, filter_idx=4)) - (1 - err4.sum() / N)) < 1e-6
    assert M.semantic_error(res["semantic_label_coarse"], tg).shape == tg.shape
    counts = np.zeros((Cn, Cn))
    for g, p in zip(gt[:, 0], pred):
        counts[g, p] += 1
    cm_counts = M.confusion_matrix_values(res, tg, Cn, normalize=None).numpy()
    assert np.array_equal(cm_counts, counts)
    cm = M.confusion_matrix_values(res, tg, Cn).numpy()
    rows = counts.sum(1, keepdims=True)
    assert np.allclose(cm, np.divide(counts, rows, out=np.zeros_like(counts), where=rows > 0), atol=1e-6)
    ious = np.array([counts[c, c] / (counts[c].sum() + counts[:, c].sum() - counts[c, c]) if
                     (counts[c].sum() + counts[:, c].sum()) > 0 else np.nan for c in range(Cn)])
    assert abs(float(M.semantic_mIoU(cm_counts)) - np.nanmean(ious)) < 1e-9
    w, b = res["weights_coarse"].numpy(), res["beta_coarse"].numpy()
    comp = (w[..., None] * b).sum(-2)[:, 0]
    car = gt[:, 0] == 3
    assert abs(float(M.uncertainty_at_transient(res, tg, 3)) - comp[car].sum() / car.sum()) < 1e-5
    a, c = torch.rand(40, 3), torch.rand(40, 3)
    mask = torch.rand(40) > 0.5
    assert abs(float(psnr(a, c, mask)) - float(-10 * torch.log10(((a - c) ** 2)[mask].mean()))) < 1e-6
    assert mse(a, c, reduction="none").shape == (40, 3)
