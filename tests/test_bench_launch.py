"""`python bench.py --gpus N` must be able to launch its own ranks (the driver's scaling run starts it as a plain command):
the parent starts N rank processes before it touches the GPU, relays rank 0's single JSON line and fails if a rank fails."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=600):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):   # a plain command: no launcher environment
        env.pop(k, None)
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, cwd=ROOT, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]          # ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_bench_spawns_its_ranks_rehearsal_gloo_cpu():
    """CPU rig: the launch / rendezvous / collective / timing protocol without kernels (gloo)."""
    d = _run(["--gpus", "2", "--rehearsal", "--steps", "2", "--warmup", "1", "--rays", "64"])
    assert d["n_gpus"] == 2 and d["config"]["distributed"]["world_size"] == 2
    assert d["config"]["distributed"]["backend"] == "gloo" and d["rehearsal"] is True
    assert d["config"]["global_batch"] == 128 and d["steps"] == 2 and d["scaling"] == "weak"


def test_bench_preset_c3_two_rank_rehearsal():
    """BASELINE.json configs[2] by name: `bench.py --config c3` brings its own rank count (2), per-GPU shape (8192 / 2 rays x 96 samples)
    and arithmetic (the reduced-precision mode) -- launched as a plain command, collectives rehearsed on the CPU (gloo)."""
    d = _run(["--config", "c3", "--rehearsal", "--steps", "2", "--warmup", "1"])
    assert d["n_gpus"] == 2 and d["config"]["distributed"]["world_size"] == 2 and d["rehearsal"] is True
    c = d["config"]
    assert c["preset"] == "c3" and c["rays_per_gpu"] == 4096 and c["samples"] == 96 and c["global_batch"] == 8192 and c["mfma"] == "f16x1"
    # the other presets resolve without a launch: shape / ranks / arithmetic as BASELINE.json names them
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    want = {"c1": (512, 32, 1, "f16x2"), "c2": (4096, 64, 1, "f16x2"), "c3": (8192, 96, 2, "f16x1"), "c4": (16384, 128, 8, "f16x2"), "c5": (32768, 128, 8, "f16x1")}
    for k, (rays, S, gpus, mfma) in want.items():
        pz = b.PRESETS[k]
        assert (pz["rays"], pz["samples"], pz["gpus"], pz["mfma"]) == (rays, S, gpus, mfma), k
    assert b.PRESETS["c3"]["car_reg"] and b.PRESETS["c5"]["frame"] and b.PRESETS["c1"]["model"] == "satnerf"
    # a preset's pipeline configuration builds (no GPU needed): L_t on for c3, the raised vocabulary for c5, the baseline pipeline for c1
    c3 = b.make_cfgs(4096, 96, 2, "f16x1", car_reg=True)
    assert c3.pipeline.use_car_reg_loss and c3.pipeline.car_reg_loss_start == 0 and c3.pipeline.batch_size == 8192
    assert b.make_cfgs(4096, 128, 8, "f16x1", vocab=96).pipeline.t_embedding_vocab == 96
    assert b.make_cfgs(512, 32, 1, model="satnerf").pipeline.pipeline.endswith("SatNeRFPipeline")


def test_bench_rank_failure_is_reported():
    env = dict(os.environ, SNERF_BENCH_FAIL_RANK="1")   # rank 1 dies before the rendezvous: the parent must end rank 0 and fail
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearsal", "--steps", "1", "--warmup", "0"],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0


@pytest.mark.gpu
def test_bench_presets_run_on_one_gpu():
    """every preset's step runs (one rank on this GPU, two steps): c1 the baseline SatNeRF pipeline, c3 the one-plane mode with L_t, c5
    with the rank-sharded full-frame leg"""
    common = ["--gpus", "1", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-eager-gpu-baseline", "--no-inference", "--no-profile", "--no-reduced"]
    d = _run(["--config", "c1"] + common)
    assert d["config"]["preset"] == "c1" and d["config"]["rays_per_gpu"] == 512 and d["config"]["samples"] == 32 and d["value"] > 0
    d = _run(["--config", "c3"] + common)
    assert d["config"]["rays_per_gpu"] == 4096 and d["config"]["samples"] == 96 and d["dtype"].startswith("f16 (REDUCED") and "L_t" in d["config"]["workload"]
    d = _run(["--config", "c5"] + common)
    assert d["config"]["samples"] == 128 and d["inference_sharded"]["rays"] == 640 * 640 and d["inference_sharded"]["rays_per_s"] > 0
    assert d["inference_sharded"]["frame_rows_on_rank0"] == 640 * 640


@pytest.mark.gpu
def test_bench_two_ranks_plain_command_on_one_gpu():
    """The real training step with two self-spawned ranks sharing the card (gloo rehearsal backend; production = RCCL, one GPU per rank)."""
    d = _run(["--gpus", "2", "--rays", "256", "--samples", "16", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
              "--no-eager-gpu-baseline", "--no-inference", "--no-profile"], {"SNERF_DIST_BACKEND": "gloo"})
    assert d["n_gpus"] == 2 and d["config"]["distributed"]["world_size"] == 2 and d["config"]["global_batch"] == 512
    assert d["value"] > 0 and d["data"] == "synthetic"
    # the exchange is timed apart from the compute (HIP events around the flat gradient all-reduce of every timed step)
    ar = d["config"]["distributed"]["allreduce_ms_per_step"]
    assert ar["mean"] >= 0 and ar["max"] >= ar["median"] and ar["bytes"] > 0
    assert len(d["timing"]["step_ms"]) == 2 and len(d["timing"]["host_step_ms"]) == 2
