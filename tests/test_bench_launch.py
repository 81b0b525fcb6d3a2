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


def test_bench_rank_failure_is_reported():
    env = dict(os.environ, SNERF_BENCH_FAIL_RANK="1")   # rank 1 dies before the rendezvous: the parent must end rank 0 and fail
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearsal", "--steps", "1", "--warmup", "0"],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0


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
