"""Fused HIP Adam (snerf_adam_step behind optim.FlatAdam) and Lightning-layout checkpoints -- SURVEY 8(f) rank 3."""
import os

import numpy as np
import pytest
import torch

from tests.helpers import max_abs  # noqa: E402

from oracle import snerf_oracle as O

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def test_fused_adam_matches_torch_adam():
    """5 steps of the one-launch Adam on ragged parameter shapes == torch.optim.Adam (CPU, fp32) to fp32 rounding;
    the exchanged state_dicts load into each other."""
    from snerf_amd.optim import FlatAdam, StepLR
    g = torch.Generator().manual_seed(3)
    shapes = [(7, 5), (33,), (128, 63), (1,), (2, 3, 5)]
    ref_p = [torch.nn.Parameter(torch.randn(*s, generator=g)) for s in shapes]
    hip_p = [torch.nn.Parameter(p.detach().clone().to(DEV)) for p in ref_p]
    ref = torch.optim.Adam(ref_p, lr=5e-4, weight_decay=0)
    ref_s = torch.optim.lr_scheduler.StepLR(ref, step_size=1, gamma=0.9)
    hip = FlatAdam(hip_p, lr=5e-4, weight_decay=0)
    hip_s = StepLR(hip, step_size=1, gamma=0.9)
    for it in range(5):
        ref.zero_grad()
        hip.zero_grad(set_to_none=False)   # in-place mode: .grad stays a view of the flat bucket
        for a, b in zip(ref_p, hip_p):
            gr = torch.randn(a.shape, generator=g) * (10.0 ** (it - 3))
            a.grad = gr.clone()
            b.grad.add_(gr.to(DEV))          # accumulate in place into the flat view, as autograd does
        ref.step()
        hip.step()
        if it % 2 == 1:
            ref_s.step()
            hip_s.step()
        assert abs(ref.param_groups[0]["lr"] - hip.param_groups[0]["lr"]) < 1e-12
    for a, b in zip(ref_p, hip_p):
        assert torch.allclose(a.detach(), b.detach().cpu(), rtol=2e-6, atol=1e-7), float((a.detach() - b.detach().cpu()).abs().max())
    # a gradient tensor that replaced the view (p.grad = fresh tensor) is folded in on step()
    hip.zero_grad(); ref.zero_grad()
    for a, b in zip(ref_p, hip_p):
        gr = torch.randn(a.shape, generator=g)
        a.grad = gr.clone(); b.grad = gr.to(DEV)
    ref.step(); hip.step()
    for a, b in zip(ref_p, hip_p):
        assert torch.allclose(a.detach(), b.detach().cpu(), rtol=2e-6, atol=1e-7)
    # state_dict interop, both directions
    sd = hip.state_dict()
    ref2 = torch.optim.Adam([torch.nn.Parameter(p.detach().cpu().clone()) for p in hip_p], lr=1.0)
    ref2.load_state_dict(sd)
    assert abs(ref2.param_groups[0]["lr"] - hip.param_groups[0]["lr"]) < 1e-12
    assert int(ref2.state_dict()["state"][0]["step"]) == 6
    hip2 = FlatAdam([torch.nn.Parameter(p.detach().clone()) for p in hip_p], lr=1.0)
    hip2.load_state_dict(ref.state_dict())
    assert hip2.step_count == 6
    for i, p in enumerate(ref_p):
        o = hip2.offsets[i]
        assert torch.allclose(hip2.exp_avg[o:o + p.numel()].cpu(), ref.state_dict()["state"][i]["exp_avg"].reshape(-1))


def test_fused_adam_refuses_bad_arguments():
    from snerf_amd import _lib
    from snerf_amd.optim import FlatAdam
    with pytest.raises(RuntimeError):
        FlatAdam([torch.nn.Parameter(torch.zeros(4))])          # CPU parameters: no fallback
    with pytest.raises(ValueError):
        FlatAdam([torch.nn.Parameter(torch.zeros(4, device=DEV))], weight_decay=0.1)
    L = _lib.lib()
    t = torch.zeros(8, device=DEV)
    assert L.snerf_adam_step(t.data_ptr(), t.data_ptr(), t.data_ptr(), t.data_ptr(), 6, 1e-3, 0.9, 0.999, 1e-8, 1, 1.0, None) != 0
    assert L.snerf_adam_step(t.data_ptr(), t.data_ptr(), t.data_ptr(), t.data_ptr(), 8, 1e-3, 0.9, 0.999, 1e-8, 0, 1.0, None) != 0
    assert L.snerf_adam_step(None, t.data_ptr(), t.data_ptr(), t.data_ptr(), 8, 1e-3, 0.9, 0.999, 1e-8, 1, 1.0, None) != 0


def _loop(seed=4, max_steps=100):
    from tests.test_gpu_pipeline import _pipeline_for
    from snerf_amd.framework.pipelines import TrainLoop
    cfg = O.OracleCfg(fc_units=32, n_samples=16, first_beta_epoch=0)
    pipe, _ = _pipeline_for(cfg, 64, seed, max_steps=max_steps)
    pipe.log_metrics = False
    return TrainLoop(pipe, pipe.cfgs, DEV)


def test_checkpoint_resume_continues_the_same_trajectory(tmp_path):
    """train 3 steps, save (Lightning layout), resume in a fresh pipeline: steps 4-5 give the same losses and
    weights as the uninterrupted run; the file reads with torch.load(weights_only=True)."""
    from snerf_amd.framework.util import load_ckpoint as lc
    a = _loop()
    for s in range(3):
        torch.manual_seed(100 + s)
        a.step(s)
    fp = a.save_ckpoint(str(tmp_path / "run" / "ckpoints" / "epoch=0.ckpt"))
    ck = torch.load(fp, weights_only=True)
    assert ck["global_step"] == 3 and set(ck) >= {"epoch", "global_step", "state_dict", "optimizer_states", "lr_schedulers"}
    assert all(k.split(".")[0] in ("model_coarse", "model_t") for k in ck["state_dict"])
    assert lc.find_ckpoint_fp(str(tmp_path / "run"), -1) == (fp, 0)
    assert lc.read_ckpt_info(fp) == (0, 3)
    cont = []
    for s in range(3, 5):
        torch.manual_seed(100 + s)
        cont.append(float(a.step(s)["loss"]))
    b = _loop(seed=9)                       # different initial weights: everything must come from the file
    assert b.load_ckpoint(fp) == 3
    res = []
    for s in range(3, 5):
        torch.manual_seed(100 + s)
        res.append(float(b.step(s)["loss"]))
    assert np.allclose(cont, res, rtol=0, atol=1e-6), (cont, res)
    for (n, p), (_, q) in zip(a.pipeline.named_parameters(), b.pipeline.named_parameters()):
        assert torch.allclose(p, q, rtol=0, atol=1e-7), n
    # evaluation-side loading (load_from_disk) and prefix filters
    models, pipe, epoch, dev = lc.load_from_disk(b.cfgs, str(tmp_path / "run"), epoch=0, device=0)
    assert epoch == 0 and not models["coarse"].training
    sd = lc.extract_model_state_dict(fp, "model_coarse", "cpu", prefixes_to_ignore=["fc_net"])
    assert sd and not any(k.startswith("fc_net") for k in sd)
    only = lc.extract_model_state_dict(fp, "model_coarse", "cpu", prefixes_to_load=["fc_net"])
    assert only and all(k.startswith("fc_net") for k in only)


@pytest.mark.parametrize("env", [{"SNERF_PREFETCH": "0"}, {"SNERF_MAX_LEAD": "0"}, {"SNERF_MAX_LEAD": "1"}, {"SNERF_MERGE_LOSSES": "0"}])
def test_trainloop_host_side_switches_do_not_change_the_trajectory(env, monkeypatch):
    """Batch prefetch on a data stream, the bound on the host's lead and the merged loss call are scheduling: the same seeded steps give
    the same losses (bit for bit for the first two; the merged loss call sums its terms in one kernel instead of module by module) and
    the same weights, across an epoch boundary (a new permutation of the bank drawn by the prefetching side) and a jump in the step
    index (a prefetched batch that is not the one asked for)."""
    import importlib
    from snerf_amd import loss_ops
    def run():
        loop = _loop(seed=11)
        spe = loop.steps_per_epoch
        order = list(range(0, spe + 2)) + [3 * spe + 1, 3 * spe + 2]      # consecutive steps over an epoch boundary, then a jump
        losses = []
        for s in order:
            torch.manual_seed(500 + s)
            losses.append(float(loop.step(s)["loss"]))
        torch.cuda.synchronize()
        return losses, [p.detach().clone() for p in loop.pipeline.parameters()]
    base_l, base_w = run()
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    monkeypatch.setattr(loss_ops, "_MERGE", env.get("SNERF_MERGE_LOSSES", "1") != "0")
    got_l, got_w = run()
    if "SNERF_MERGE_LOSSES" in env:
        assert np.allclose(base_l, got_l, rtol=2e-6, atol=0), (base_l, got_l)
        for a_, b_ in zip(base_w, got_w):
            assert torch.allclose(a_, b_, rtol=0, atol=2e-6)
    else:
        assert base_l == got_l, (base_l, got_l)
        for a_, b_ in zip(base_w, got_w):
            assert torch.equal(a_, b_)


def test_pass_workspaces_are_leased_not_reallocated():
    """Round 3's driver-timed bench lost 240 ms to a hipMalloc of a 10 GB pass workspace INSIDE the timed steps: torch's caching
    allocator had carved a result tensor out of the idle block while the previous step's results were still referenced.  Workspaces are
    now leased from ops' pool (device, stream, size): steps that hold the previous step's results must neither allocate device
    memory nor grow the reserved bytes, and main / sc pass get the same two tensors back every step."""
    from snerf_amd import ops
    ops.release_workspaces()
    loop = _loop()
    held = []
    for s in range(3):                      # warm-up: the pool fills, torch's small blocks settle
        torch.manual_seed(s)
        held = [loop.step(s)]
    torch.cuda.synchronize()
    idle = {k: [t.data_ptr() for t in v] for k, v in ops._WS_FREE.items()}
    assert len(idle) >= 2 and all(len(v) == 1 for v in idle.values()), idle          # one idle workspace per (stream, size): main, sc
    m0 = torch.cuda.memory_stats(DEV)
    for s in range(3, 9):
        torch.manual_seed(s)
        held.append(loop.step(s))           # results of EVERY step stay referenced while the next one allocates
    torch.cuda.synchronize()
    m1 = torch.cuda.memory_stats(DEV)
    assert {k: [t.data_ptr() for t in v] for k, v in ops._WS_FREE.items()} == idle    # the same tensors came back
    big = max(k[2] for k in idle)
    assert m1["num_alloc_retries"] == m0["num_alloc_retries"] and m1["num_ooms"] == m0["num_ooms"]
    # held result tensors may make torch take small blocks; nothing of a workspace's size is ever requested again
    assert m1["reserved_bytes.all.current"] - m0["reserved_bytes.all.current"] < big, (m0["reserved_bytes.all.current"], m1["reserved_bytes.all.current"], big)
    ops.release_workspaces()
    assert not ops._WS_FREE


def test_trainloop_gradients_live_in_the_flat_bucket():
    loop = _loop()
    torch.manual_seed(0)
    loop.step(0)
    opt = loop.optimizer
    assert all(p.grad.data_ptr() == g.data_ptr() for p, g in zip(opt.params, opt._gviews))
    assert all(p.data_ptr() == opt.flat_p.data_ptr() + 4 * o for p, o in zip(opt.params, opt.offsets))
    assert float(opt.flat_g.abs().sum()) > 0 and opt.step_count == 1


_LOOP_WORKER = r"""
import os, sys
sys.path.insert(0, {root!r})
import torch, torch.distributed as dist
from snerf_amd import parallel
rank, world, dev = parallel.init_distributed(backend="gloo")
from tests.test_gpu_optim_ckpt import _loop
torch.rand = lambda *a, **k: torch.zeros(*a, device=k.get("device"), dtype=k.get("dtype", torch.float32))
loop = _loop()
assert (loop.rank, loop.world) == (rank, 2)
losses = [float(loop.step(s)["loss"].detach()) for s in range(3)]
if rank == 0:
    torch.save({{"losses": losses, "params": {{n: p.detach().cpu() for n, p in loop.pipeline.named_parameters()}}}}, {out!r})
dist.barrier()
"""


def test_trainloop_two_ranks_equal_one_rank(tmp_path, monkeypatch):
    """TrainLoop on 2 ranks (gloo, both on this GPU; each renders its half of every global batch, the flat gradient
    bucket is all-reduced, FlatAdam steps) == TrainLoop on 1 rank with the whole batch, after 3 optimiser steps."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    monkeypatch.setattr(torch, "rand", lambda *a, **k: torch.zeros(*a, device=k.get("device"), dtype=k.get("dtype", torch.float32)))
    one = _loop()
    ref_losses = [float(one.step(s)["loss"].detach()) for s in range(3)]
    ref = {n: p.detach().cpu() for n, p in one.pipeline.named_parameters()}
    monkeypatch.undo()
    res = str(tmp_path / "loop.pt")
    script = tmp_path / "worker.py"
    script.write_text(_LOOP_WORKER.format(root=root, out=res))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29633", WORLD_SIZE="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK="0"), cwd=root) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=300) == 0
    two = torch.load(res, weights_only=True)
    # the reported loss is the rank-local one (its own rays, global denominators); weights must agree
    for n, v in ref.items():
        assert torch.allclose(two["params"][n], v, rtol=0, atol=2e-5), (n, float((two["params"][n] - v).abs().max()))
    assert all(np.isfinite(two["losses"])) and all(np.isfinite(ref_losses))


def _one_backward(loop, step=0, seed=321):
    """the loss of one training step of the loop's pipeline (its own batch sampler, fixed jitter)"""
    pl = loop.pipeline
    pl.current_epoch = 0
    batch = {"rgb": loop.bank.batch(step, loop.global_batch, loop.rank, loop.world, shuffle=loop.shuffle)}
    torch.manual_seed(seed)
    return pl.training_step(batch, step)["loss"]


def test_gradient_sinks_equal_autograd_accumulation(monkeypatch):
    """The passes may add their parameter gradients straight into FlatAdam's bucket (ops.accumulate_into_sinks, the training
    loop's backward) instead of returning them to autograd: same seed, same step -> the flat gradient buffer is the same
    to fp32 summation order, and bit-identical from run to run either way."""
    from snerf_amd import ops
    a, b = _loop(seed=9), _loop(seed=9)
    for lp in (a, b):
        lp.optimizer.zero_grad()
    with ops.accumulate_into_sinks():
        _one_backward(a).backward()
    ops.wait_grad_sinks()
    monkeypatch.setattr(ops, "_SINKS_ON", False)
    _one_backward(b).backward()
    torch.cuda.synchronize()
    ga, gb = a.optimizer.flat_g.clone(), b.optimizer.flat_g.clone()
    assert float(ga.abs().max()) > 0
    assert max_abs(ga.cpu(), gb.cpu()) <= 1e-6 * float(gb.abs().max())
    monkeypatch.setattr(ops, "_SINKS_ON", True)
    a.optimizer.zero_grad()
    with ops.accumulate_into_sinks():
        _one_backward(a).backward()
    ops.wait_grad_sinks()
    torch.cuda.synchronize()
    assert torch.equal(a.optimizer.flat_g, ga), "sink path is not bit-reproducible"


def test_autograd_grad_contract_with_flat_adam():
    """Outside ops.accumulate_into_sinks the passes keep the autograd contract even when every .grad is a view of FlatAdam's
    bucket: torch.autograd.grad returns the gradients and leaves .grad untouched; so does a plain backward() (which then
    accumulates through AccumulateGrad as usual)."""
    lp = _loop(seed=11)
    lp.optimizer.zero_grad()
    params = lp.params
    before = lp.optimizer.flat_g.clone()
    grads = torch.autograd.grad(_one_backward(lp), params, allow_unused=True)
    torch.cuda.synchronize()
    assert torch.equal(lp.optimizer.flat_g, before), "autograd.grad wrote into .grad"
    got = [g for g in grads if g is not None]
    assert len(got) >= len(params) - 2 and all(bool(torch.isfinite(g).all()) for g in got)
    assert max(float(g.abs().max()) for g in got) > 0
    # the same gradients arrive in .grad through a plain backward
    _one_backward(lp).backward()
    torch.cuda.synchronize()
    for p, g in zip(params, grads):
        if g is not None:
            assert max_abs(p.grad.cpu(), g.cpu()) <= 1e-6 * max(1e-12, float(g.abs().max()))
