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
    assert ops.mfma_mode(NS(precision=16), None) == "f16x1"               # the reference's half-precision knob (baseline/pipelines/nerf.py:65)
    assert ops.mfma_mode(NS(precision=32, mfma_precision="bf16"), None) == "f16x1"   # BASELINE.json's name for the same mode
    for run, want in (("highest", "f16x2"), ("high", "f16x1"), ("medium", "f16x1")):
        assert ops.mfma_mode(NS(precision=32, mfma_precision="auto"), NS(float32_matmul_precision=run)) == want
    assert ops.ModelSpec(mfma="f16x1").desc(16, 8).flags == _lib.FLAG_F16X1 == ops.ModelSpec(mfma="bf16").desc(16, 8).flags
    assert ops.ModelSpec().desc(16, 8).flags == 0                        # default: no arithmetic bit = f16x2 (C callers too)

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
# rank-sharded full-frame inference (SURVEY 8(e) last row): frame_shard + all_gather of the per-ray results, ragged tails and
# an empty shard included; per-sample results stay local
from snerf_amd.eval.utils.util import shard_and_gather
for n in (1001, 64, 1):
    def rows(lo, hi, n=n):
        idx = torch.arange(lo, hi)
        return {{"rgb_coarse": torch.stack([idx.float(), idx.float() * 2, idx.float() + 0.5], 1), "depth_coarse": idx.float() / 7,
                "semantic_label_coarse": idx % 5, "weights_coarse": idx.float()[:, None].repeat(1, 4)}}
    got = shard_and_gather(rows, n)
    want = rows(0, n)
    lo, hi = got["_rows"]
    assert (lo, hi) == parallel.frame_shard(n) and hi - lo == max(0, min(-(-n // world), n - rank * -(-n // world)))
    for k in ("rgb_coarse", "depth_coarse", "semantic_label_coarse"):
        assert got[k].dtype == want[k].dtype and torch.equal(got[k], want[k]), (n, k)
    assert torch.equal(got["weights_coarse"], want["weights_coarse"][lo:hi])
assert [parallel.frame_shard(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 9), (9, 10)]
assert [parallel.frame_shard(2, r, 4) for r in range(4)] == [(0, 1), (1, 2), (2, 2), (2, 2)]
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
    of semantic/components/metrics.py:11-87 and eval/utils/metrics.py:8-18 (tests/helpers.py; the GPU twin is
    tests/test_gpu_rows.py::test_validation_metrics_on_device)."""
    from tests.helpers import check_validation_metrics
    check_validation_metrics("cpu")


def test_siren_init_ranges():
    """baseline/models/commons.py:5-18 as applied by rs_semantic.py:239-243 / satnerf.py:140-160: trunk and sun-visibility
    weights ~ U(+-sqrt(6 / fan_in)), their first layers ~ U(+-1 / fan_in); biases and every other head keep PyTorch's
    default Linear init (|.| <= 1 / sqrt(fan_in))."""
    import math
    from snerf_amd.framework.configs import MainConfig
    from snerf_amd.framework.pipelines import load_pipeline
    torch.manual_seed(3)
    cfgs = MainConfig(run={"synthetic_rays": 256}, pipeline={"pipeline": "snerf_amd.semantic.pipelines.rs_semantic.RSSemanticPipeline",
                                                             "fc_units": 128, "batch_size": 64, "depth_enabled": False})
    m = load_pipeline(cfgs).model_coarse
    for seq in (m.fc_net, m.sun_v_net):
        first = True
        for layer in seq:
            if not isinstance(layer, torch.nn.Linear):
                continue
            n = layer.weight.shape[1]
            bound = 1.0 / n if first else math.sqrt(6.0 / n)
            w = layer.weight.detach()
            assert float(w.abs().max()) <= bound * (1 + 1e-6), (n, first)
            if w.numel() >= 4096:     # the range is actually used (uniform: max close to the bound, std = bound / sqrt(3))
                assert float(w.abs().max()) >= 0.98 * bound and abs(float(w.std()) - bound / math.sqrt(3)) <= 0.03 * bound
            assert float(layer.bias.detach().abs().max()) <= 1.0 / math.sqrt(n) + 1e-6
            first = False
    for head in (m.rgb_from_xyzdir, m.semantic_prediction, m.beta_from_xyz, m.sky_color):
        for layer in head:
            if isinstance(layer, torch.nn.Linear):
                n = layer.weight.shape[1]
                assert float(layer.weight.detach().abs().max()) <= 1.0 / math.sqrt(n) + 1e-6
    # relu networks keep the default init everywhere (the sine initialisers are only applied for siren)
    cfgs = MainConfig(run={"synthetic_rays": 256}, pipeline={"pipeline": "snerf_amd.semantic.pipelines.rs_semantic.RSSemanticPipeline",
                                                             "fc_units": 128, "batch_size": 64, "depth_enabled": False,
                                                             "activation_function": "relu"})
    m = load_pipeline(cfgs).model_coarse
    w = m.fc_net[2].weight.detach()
    assert float(w.abs().max()) <= 1.0 / math.sqrt(w.shape[1]) + 1e-6


def test_ray_bank_per_image_sizes():
    """A test bank that carries each image's H*W hands over the reference's images (one per validation step,
    framework/pipelines.py:120-129), whole or as contiguous rank slices; synthetic banks cut equal slices."""
    from snerf_amd.framework.datasets import GpuRayBank
    n = 20
    t = {"rays": torch.arange(n, dtype=torch.float32)[:, None].repeat(1, 8), "extras": torch.zeros(n, 4)}
    bank = GpuRayBank(t, image_sizes=[5, 9, 6])
    assert bank.n_images() == 3
    assert bank.image(1)["rays"][:, 0].tolist() == list(range(5, 14))
    parts = [bank.image(1, None, r, 2)["rays"][:, 0].tolist() for r in range(2)]
    assert parts[0] + parts[1] == list(range(5, 14)) and len(parts[0]) == 5
    with pytest.raises(ValueError):
        bank.image(0, None, 0, 8)                      # 5 rays cannot feed 8 ranks
    with pytest.raises(ValueError):
        GpuRayBank(t, image_sizes=[5, 9])              # does not cover the bank
    plain = GpuRayBank(t)
    assert plain.n_images(6) == 3 and plain.image(2, 6)["rays"][:, 0].tolist() == list(range(12, 18))
    with pytest.raises(ValueError):
        plain.n_images()


def test_loss_plans_merge_by_term_groups():
    """loss_ops.merge_plans: modules whose terms do not collide become one configuration (field by field from the module that owns the
    group), the targets are united, the loss_dict keys concatenated; colliding owners or different target tensors do not merge."""
    import torch
    from snerf_amd.loss_ops import LossSpec, merge_plans
    gt, labels, mask = torch.zeros(4, 3), torch.zeros(4, 1, dtype=torch.long), torch.ones(4, dtype=torch.bool)
    color = (LossSpec(color_mode=2, has_sc=True, sc_lambda=0.05), {"gt_rgb": gt}, ["coarse_color", "coarse_logbeta", "coarse_sc_term2", "coarse_sc_term3"])
    sem = (LossSpec(sem_mode=1, ignore_index=4, lambda_s=0.04, n_classes=5), {"labels": labels, "mask": mask}, ["coarse_semantic"])
    car = (LossSpec(car_reg=True, car_label=4, lambda_c=0.1), {"labels": labels, "mask": mask}, ["coarse_car_reg_loss"])
    spec, aux, keys = merge_plans([color, sem, car])
    assert spec == LossSpec(color_mode=2, has_sc=True, sc_lambda=0.05, sem_mode=1, ignore_index=4, lambda_s=0.04, n_classes=5,
                            car_reg=True, car_label=4, lambda_c=0.1)
    assert aux["gt_rgb"] is gt and aux["labels"] is labels and aux["mask"] is mask
    assert keys == color[2] + sem[2] + car[2]
    assert merge_plans([color]) is color
    assert merge_plans([color, (LossSpec(color_mode=1), {"gt_rgb": gt}, ["coarse_color"])]) is None          # two colour losses
    assert merge_plans([sem, (car[0], {"labels": labels.clone(), "mask": mask}, car[2])]) is None              # different label tensors
    nomask = (car[0], {"labels": labels, "mask": None}, car[2])
    assert merge_plans([sem, nomask]) is None                                                                  # one call has one mask
