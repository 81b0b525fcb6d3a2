#!/usr/bin/env python3
"""Headline benchmark: train rays/s of the semantic Sat-NeRF hot path on synthetic ray batches.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One step = one full optimiser step of BASELINE.json configs[1] ("JAX_068 semantic pipeline, 4096 rays x 64
samples, fp32") per GPU: on-device batch sampling -> main + solar-correction forward -> SatNerfLoss +
solar correction + SemanticLoss (fused HIP loss kernels) -> backward -> flat gradient all-reduce (RCCL,
N > 1) -> Adam.  Weak scaling: 4096 rays per GPU.  Prints ONE JSON line (rank 0).

Besides the contract fields the line carries, all measured in the same run (N = 1):
  roofline             dominant kernel (the K-contiguous dense-layer GEMM): algorithmic FLOPs / HIP-event duration against the
                       dense fp16 MFMA peak (`frac`), the matrix-pipe occupancy of the three-product arithmetic next to it
  cpu_baseline         the CPU oracle on the host cores, bounded sample
  reference_gpu_eager  the same restatement on cuda:0 with stock PyTorch-ROCm eager ops (north_star's denominator)
  inference            forward-only rays/s: lean full-frame path and the reference-shaped batched_inference
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import warnings  # noqa: E402

import torch  # noqa: E402

# two-stream backward: autograd notes that .grad accumulation happens on another stream than the producer (it
# synchronises correctly; the note is about CUDA-graph capture)
warnings.filterwarnings("ignore", message="The AccumulateGrad node's stream does not match")

FLOPS_PER_SAMPLE_TRAIN = 31_453_696  # main 3*F_m + sc (F_s + 2*F_s-branch), fc_units=512
# configs[0], baseline SatNeRF (2,629,632 weights; its sc pass needs trunk + sigma + feats + sun_v = 2,363,904 of them): the same rule
FLOPS_PER_SAMPLE_TRAIN_SATNERF = 3 * 2 * 2_629_632 + 3 * 2 * 2_363_904 - 1024
FP32_MFMA_PEAK_TFLOPS = 157.3        # /opt/skills/guides/MI355X_MICROARCH.md "Peak FP32 (matrix)"
BF16_MFMA_PEAK_TFLOPS = 2500.0       # same guide: "Peak BF16/FP16 MFMA ~2.5 PF dense"
# What a bare loop of the product's v_mfma_f32_32x32x16_f16 SUSTAINS on real operands under the 1,400 W package cap (the 2.5 PF are a
# zero-operand figure: profiles/r05/power_cap.md, tools/ablate/mfma_power.hip; 1,677 / 1,707 / 1,712 TFLOP/s on two boxes).  Reported next
# to `peak`, never instead of it.
FP16_MFMA_SUSTAINED_TFLOPS = 1700.0
HBM_ACHIEVABLE_BPS = 6.3e12          # same guide: achievable HBM3E streaming rate (peak 8 TB/s)
ALGORITHMIC_BYTES_PER_RAY = 3700     # rays + targets in, per-ray results out, parameters amortised


# BASELINE.json `configs`, launchable by name (--config cN).  `rays` is the GLOBAL batch the configuration names and `gpus` the rank
# count it names: the per-GPU shape (rays / gpus) is what a rank runs, whatever --gpus the launch actually has (weak scaling).
PRESETS = {
    "c1": {"name": "configs[0]: JAX_004 baseline SatNeRF, 512 rays x 32 samples (the reference's CPU plumbing case; here on the HIP path)",
           "rays": 512, "samples": 32, "gpus": 1, "mfma": "f16x2", "model": "satnerf"},
    "c2": {"name": "configs[1]: JAX_068 semantic pipeline, 4096 rays x 64 samples, 1 GPU, fp32 (the headline)",
           "rays": 4096, "samples": 64, "gpus": 1, "mfma": "f16x2"},
    "c3": {"name": "configs[2]: JAX_214 semantic + transient regularisation L_t, 8192 rays x 96 samples, reduced precision, 2 GPUs",
           "rays": 8192, "samples": 96, "gpus": 2, "mfma": "f16x1", "car_reg": True},
    "c4": {"name": "configs[3]: JAX_260 semantic, 16384 rays x 128 samples, fp32, 8 GPUs",
           "rays": 16384, "samples": 128, "gpus": 8, "mfma": "f16x2"},
    "c5": {"name": "configs[4]: four DFC2019 scenes concatenated, 32768 rays x 128 samples, reduced precision, 8 GPUs + rank-sharded full-frame inference",
           "rays": 32768, "samples": 128, "gpus": 8, "mfma": "f16x1", "vocab": 96, "frame": True},
}
FLOPS_PER_SAMPLE_FWD = 5_640_704     # forward of the semantic model, main pass (the reference's inference cost per sample)


def make_cfgs(rays_per_gpu, samples, world, mfma="f16x2", car_reg=False, vocab=50, model="semantic"):
    from snerf_amd.framework.configs import MainConfig
    if model == "satnerf":       # configs[0]: the baseline pipeline (no positional encoding, no semantic head; baseline/pipelines/satnerf.py)
        pipeline = {"pipeline": "snerf_amd.baseline.pipelines.satnerf.SatNeRFPipeline", "n_samples": samples, "batch_size": rays_per_gpu * world,
                    "render_chunk_size": 1 << 22, "learnrate": 5e-4, "fc_units": 512, "fc_layers": 8, "fc_skips": [4], "activation_function": "siren",
                    "sc_lambda": 0.05, "t_embedding_vocab": vocab, "t_embedding_tau": 4, "first_beta_epoch": 0, "depth_enabled": False,
                    "mfma_precision": mfma}
        run = {"max_train_steps": 1 << 30, "synthetic_rays": max(1 << 20, rays_per_gpu * world * 4), "synthetic_images": 19,
               "synthetic_seed": 0, "shuffle_dataset": True}
        return MainConfig(run=run, pipeline=pipeline)
    pipeline = {
        "pipeline": "snerf_amd.semantic.pipelines.rs_semantic.RSSemanticPipeline",
        # configs/pipelines/rs_semantic.toml values
        "n_samples": samples, "batch_size": rays_per_gpu * world, "render_chunk_size": 1 << 22,
        "learnrate": 5e-4, "fc_units": 512, "fc_layers": 8, "fc_skips": [4], "activation_function": "siren",
        "mapping_pos_n_freq": 10, "sc_lambda": 0.05, "t_embedding_vocab": vocab, "t_embedding_tau": 4,
        "lambda_s": 0.04, "semantic_activation_function": "sigmoid", "ignore_car_index": True,
        # steady state of training: beta loss active (epoch >= first_beta_epoch), depth rays dropped
        # (after 25 % of the steps, baseline/pipelines/satnerf.py:26-29) -- SURVEY.md 8(d)
        "first_beta_epoch": 0, "depth_enabled": False, "mfma_precision": mfma,
    }
    if car_reg:      # configs[2]: + the transient regularisation L_t, active from the first epoch on (the bench never leaves epoch 0)
        pipeline.update(use_car_reg_loss=True, car_reg_loss_start=0)
    run = {"max_train_steps": 1 << 30, "synthetic_rays": max(1 << 20, rays_per_gpu * world * 4), "synthetic_images": min(19, vocab - 1) if vocab <= 50 else 76,
           "synthetic_seed": 0, "shuffle_dataset": True}
    return MainConfig(run=run, pipeline=pipeline)


def cpu_baseline(samples, seconds_budget=25.0):
    """The CPU oracle (oracle/snerf_oracle.py, the restatement pinned to the reference by golden vectors)
    timed on the host cores on a bounded sample of the same workload: full train step, W=512."""
    from oracle import snerf_oracle as O
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    affinity = cores
    cores = min(cores, 16)  # the GPU box's CPU share for one GPU (16 cores); more threads only oversubscribe that quota
    torch.set_num_threads(cores)
    cfg = O.OracleCfg(n_samples=samples)
    n = 512
    p = O.to_torch(O.init_params_numpy(cfg, 0), requires_grad=True)
    emb = torch.from_numpy(O.init_embedding_numpy(cfg, 0)).requires_grad_(True)
    b = O.batch_to_torch(O.synthetic_batch(n, samples, seed=0))
    O.train_step(p, emb, cfg, b, epoch=2)  # warm-up
    t0 = time.time()
    reps = 0
    while reps < 1 or (time.time() - t0) < seconds_budget / 2 and reps < 8:
        O.train_step(p, emb, cfg, b, epoch=2)
        reps += 1
    dt = (time.time() - t0) / reps
    return {"value": n / dt, "unit": "train rays/s", "cores": cores, "kind": "port", "cpu_model": cpu_model(), "affinity_cores": affinity,
            "threads_note": "torch intra-op threads = min(affinity, 16): one GPU's CPU share on the GPU box is 16 cores",
            "sample": f"{reps} full train steps (main+sc fwd, losses, bwd) of {n} rays x {samples} samples, "
                      f"fc_units=512, oracle/snerf_oracle.py on torch CPU fp32, {dt:.2f} s/step"}


def eager_gpu_baseline(rays, samples, device, steps=3):
    """the same restatement on cuda:0 with stock PyTorch-ROCm eager ops, chunked like
    the reference (render_chunk_size = 40960 points) -- the denominator of the north-star '>= 10x the reference
    single-GPU PyTorch rays/s'.  Reported next to the result; never part of `value`."""
    from oracle import snerf_oracle as O
    cfg = O.OracleCfg(n_samples=samples)
    p = {k: v.to(device).requires_grad_(True) for k, v in O.to_torch(O.init_params_numpy(cfg, 0)).items()}
    emb = torch.from_numpy(O.init_embedding_numpy(cfg, 0)).to(device).requires_grad_(True)
    b = {k: v.to(device) for k, v in O.batch_to_torch(O.synthetic_batch(rays, samples, seed=0)).items()}
    opt = torch.optim.Adam(list(p.values()) + [emb], lr=5e-4)

    def step():
        opt.zero_grad(set_to_none=True)
        b["u"] = torch.rand(rays, samples, device=device)
        res = O.render_rays(p, emb, cfg, b["rays"], b["extras"], b["u"])
        O.total_loss(O.training_losses(res, b, cfg, 2)).backward()
        opt.step()
    step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    return {"value": rays / dt, "unit": "train rays/s", "ms_per_step": dt * 1e3, "kind": "port on GPU, stock PyTorch-ROCm eager ops (fp32, 'highest' matmul precision)",
            "sample": f"{steps} full train steps of {rays} rays x {samples} samples, render_chunk_size 40960 points"}


def power_probe(loop, step0, seconds=1.5):
    """Clock and package power WHILE the step runs, on this box: rocm-smi sampled from a thread beside ~`seconds` of extra steps AFTER the
    timed region (read-only; nothing here touches the headline).  The step is bound by the package power cap (profiles/r05/power_cap.md):
    the line carries the clock the firmware held and the watts it drew, so a reader can tell a slow box from a slow build."""
    import re
    import subprocess
    import threading
    samples, stop = [], threading.Event()

    def smi():
        while not stop.is_set():
            try:
                o = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=5).stdout
                c = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", o)
                w = re.search(r"Power \(W\): ([\d.]+)", o)
                if c and w:
                    samples.append((time.perf_counter(), int(c.group(1)), float(w.group(1))))
            except Exception:
                return
            time.sleep(0.1)

    try:
        th = threading.Thread(target=smi, daemon=True)
        th.start()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 0
        while time.perf_counter() - t0 < seconds:
            for _ in range(4):
                loop.step(step0 + n); n += 1
            torch.cuda.synchronize()
        t1 = time.perf_counter()
        stop.set()
        th.join(timeout=6)
        mid = sorted((c, w) for t, c, w in samples if t0 + 0.4 * (t1 - t0) <= t <= t1)
        if not mid:
            return None, n
        return {"sclk_mhz": mid[len(mid) // 2][0], "package_power_w": sorted(w for _, w in mid)[len(mid) // 2], "samples": len(mid),
                "nominal_sclk_mhz": 2400, "power_cap_w": 1400,
                "how": "rocm-smi --showclocks --showpower from a thread beside %d extra steps after the timed region (median of the samples in the last 60 %%)" % n}, n
    except Exception:
        stop.set()
        return None, 0


def inference_rates(pipe, cfgs, device, samples, n_rays=40960 * 4):
    """Forward-only rays/s (SURVEY 8(d) asks for it next to the training number): one "image" of n_rays rays through
    (a) lean_inference -- rgb + depth + label only, no solar-correction pass, written in place per chunk -- and
    (b) batched_inference, the reference-shaped full result dict incl. the sc pass (eval/utils/util.py:13-42);
    render_chunk_size = the reference default 40960 rays."""
    from snerf_amd.eval.utils.util import batched_inference, lean_inference
    from snerf_amd.framework.datasets import GpuRayBank
    bank = GpuRayBank.synthetic(n_rays, 19, 5, 123, device=device)
    rays, extras = bank.t["rays"], bank.t["extras"]
    old = cfgs.pipeline.render_chunk_size
    cfgs.pipeline.render_chunk_size = 40960
    out = {}
    try:
        keys = ("rgb_coarse", "depth_coarse", "semantic_label_coarse") if pipe.models["coarse"].spec.n_classes > 0 else ("rgb_coarse", "depth_coarse")
        for name, fn in (("lean", lambda: lean_inference(cfgs, pipe.renderer, pipe.models, rays, extras, keys=keys)),
                         ("batched", lambda: batched_inference(cfgs, pipe.renderer, pipe.models, rays, extras))):
            fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            r = fn()
            torch.cuda.synchronize()
            out[f"{name}_rays_per_s"] = n_rays / (time.perf_counter() - t0)
            del r
            torch.cuda.empty_cache()
    finally:
        cfgs.pipeline.render_chunk_size = old
    mode = pipe.models["coarse"].spec.mfma
    for rr in ("r05", "r04"):       # HBM bytes per ray: PMC passes of their own (tools/pmc_inference.sh), newest committed summary of this mode
        pi = os.path.join(ROOT, "profiles", rr, "pmc_inference.json" if mode == "f16x2" else "pmc_inference_f16x1.json")
        if os.path.isfile(pi) and samples == 64:
            try:
                pj = json.load(open(pi))
                out["hbm_bytes_per_ray"] = pj["lean"]["hbm_bytes_per_ray"]
                out["hbm_bytes_source"] = {"file": os.path.relpath(pi, ROOT), "how": pj.get("how")}
                break
            except Exception:
                pass
    # forward-only work against the matrix peak: the reference's forward FLOPs per sample (all heads) x samples x rays/s, algorithmic
    # (the three products of the default arithmetic are not counted); the lean path leaves out what the frame does not ask for
    out["frac"] = out["lean_rays_per_s"] * samples * FLOPS_PER_SAMPLE_FWD / (BF16_MFMA_PEAK_TFLOPS * 1e12)
    out["mode"] = mode
    out.update(unit="rays/s", rays=n_rays, samples=samples, render_chunk_size=40960,
               lean="rgb + depth + semantic_label, main pass only", batched="all results of render_rays incl. solar-correction pass")
    return out

def reduced_precision_leg(rays, samples, device, steps=10, warmup=4):
    """one training configuration in the REDUCED-precision mode (f16x1: one fp16 plane, one MFMA product): the reference's
    `precision = 16` knob (baseline/pipelines/nerf.py:65), BASELINE.json configs[2] (S = 96, + L_t) and configs[4] (S = 128)"""
    from snerf_amd import ops
    from snerf_amd.framework.pipelines import load_pipeline, TrainLoop
    ops.release_workspaces()
    torch.cuda.empty_cache()
    cfgs = make_cfgs(rays, samples, 1, "f16x1", car_reg=samples == 96)   # configs[2]: semantic + transient regularisation L_t
    pipe = load_pipeline(cfgs)
    pipe.log_metrics = False
    loop = TrainLoop(pipe, cfgs, device)
    step = 0
    for _ in range(warmup):
        out = loop.step(step); step += 1
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = loop.step(step); step += 1
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    res = {"mode": "f16x1", "dtype": "f16 (one block-scaled fp16 plane; REDUCED)", "rays": rays, "samples": samples, "rays_per_s": rays / dt,
           "ms_per_step": dt * 1e3, "steps": steps, "final_loss": float(out["loss"].detach()),
           "config": {96: "configs[2] shape (semantic + L_t)", 128: "configs[4] per-GPU shape"}.get(
               samples, "the headline shape (configs[1]) in the mode the reference's template default float32_matmul_precision = 'high' maps to under mfma_precision = 'auto'")}
    if samples != 96:       # full-frame forward-only rendering (eval/extract_pointcloud.py) in the same mode: configs[4]'s second half at S = 128
        del loop, out
        ops.release_workspaces()
        torch.cuda.empty_cache()
        inf = inference_rates(pipe, cfgs, device, samples, n_rays=40960 * 2)
        res["lean_inference_rays_per_s"] = inf["lean_rays_per_s"]
        res["lean_inference_rays"] = inf["rays"]
        res["lean_inference_frac"] = inf["frac"]
        if "hbm_bytes_per_ray" in inf:
            res["lean_inference_hbm_bytes_per_ray"] = inf["hbm_bytes_per_ray"]
        res["lean_inference_note"] = "trunk + feats as ONE persistent launch with the activation tile resident in LDS (csrc/bsp_trunk.hip)"
        loop = out = None
    del loop, pipe, out
    ops.release_workspaces()
    torch.cuda.empty_cache()
    return res


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes of this script (one GPU per LOCAL_RANK,
    rendezvous on 127.0.0.1), relay rank 0's JSON line, fail if any rank fails.  Runs BEFORE this process touches the GPU
    (a process that has initialised HIP must not exec or fork workers that use it), and never re-executes itself: the
    children are ordinary subprocesses."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    import tempfile
    procs = []
    out0 = tempfile.TemporaryFile()
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL between processes needs it on this driver
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=out0 if r == 0 else subprocess.DEVNULL))
    bad = []
    while not bad and any(p.poll() is None for p in procs):   # a rank that dies leaves the others in a collective: end them
        time.sleep(0.2)
        bad = [(r, p.returncode) for r, p in enumerate(procs) if p.poll() not in (None, 0)]
    if bad:
        for p in procs:
            if p.poll() is None:
                p.kill()          # exactly the processes started above
    for p in procs:
        p.wait()
    out0.seek(0)
    sys.stdout.write(out0.read().decode())
    sys.stdout.flush()
    if bad:
        raise SystemExit(f"bench.py: rank(s) failed: {bad}")


def rehearsal(args):
    """The multi-rank protocol of the bench without the HIP path (CPU rigs, `--rehearsal`): rendezvous, warm-up, K timed
    "steps" that are only the step's two collectives (16-float loss sums/counts, flat gradient bucket), barrier, max over
    ranks, rank 0's JSON line.  Labelled as such: its `value` says nothing about the kernels."""
    import torch.distributed as dist
    from snerf_amd import parallel
    if os.environ.get("SNERF_BENCH_FAIL_RANK") == os.environ.get("RANK", "0"):   # test hook: a rank that dies at start-up
        raise SystemExit(3)
    rank, world, device = parallel.init_distributed("gloo")   # the rehearsal's tensors live on the host whatever the box has
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    bucket = torch.zeros(2_826_766 + 200, dtype=torch.float32)   # the flat gradient bucket
    sums = torch.zeros(16, dtype=torch.float32)

    def step():
        parallel.allreduce_sum_(sums)
        parallel.allreduce_sum_(bucket)
    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if world > 1:
        dist.barrier()
    tmax = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    if rank == 0:
        dt = float(tmax.item())
        print(json.dumps({
            "metric": f"train rays/sec ({args.rays} rays x {args.samples} samples)", "value": args.rays * world * args.steps / dt, "unit": "rays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "none", "data": "rehearsal (collectives only, no kernels)", "rehearsal": True,
            "config": {"workload": "launch / rendezvous / collective protocol only", "preset": args.config, "mfma": args.mfma,
                       "rays_per_gpu": args.rays, "samples": args.samples,
                       "global_batch": args.rays * world, "parallelism": f"dp{world}",
                       "distributed": {"backend": dist.get_backend() if world > 1 else None, "world_size": world,
                                       "gradient_bucket_floats": int(bucket.numel())}}}))
    if world > 1:
        dist.destroy_process_group()


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="ranks (default: 1, or the rank count the --config names)")
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default=None, choices=sorted(PRESETS), help="a BASELINE.json configuration by name: c1 ... c5 (per-GPU shape, arithmetic, "
                    "loss set and rank count of configs[0] ... configs[4]); without it: --rays / --samples / --mfma (default = c2, the headline)")
    ap.add_argument("--rays", type=int, default=None, help="rays per GPU (default 4096)")
    ap.add_argument("--samples", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="skip the per-GEMM HIP-event timing")
    ap.add_argument("--serial-passes", action="store_true", help="run main and sc pass on one stream (as the roofline phase does)")
    ap.add_argument("--mfma", default=None, choices=["f16x2", "f16x1"],
                    help="matrix arithmetic: f16x2 (default, headline: fp32-class); f16x1 = the REDUCED-precision mode of BASELINE configs[2]/[4] "
                         "(one fp16 plane; reported under its own dtype, never as the fp32 headline)")
    ap.add_argument("--no-reduced", action="store_true", help="skip the reduced-precision legs (4096 x 96 and 4096 x 128 in f16x1)")
    ap.add_argument("--no-eager-gpu-baseline", action="store_true", help="skip the stock-PyTorch-on-GPU denominator (3 steps)")
    ap.add_argument("--no-inference", action="store_true", help="skip the forward-only (full-frame inference) leg")
    ap.add_argument("--rehearsal", action="store_true", help="multi-rank protocol only (collectives, no kernels): CPU rigs")
    args = ap.parse_args()
    preset = PRESETS.get(args.config or "", None)
    if preset is not None:
        if args.rays is not None or args.samples is not None or args.mfma is not None:
            raise SystemExit("--config fixes the shape and the arithmetic: drop --rays / --samples / --mfma")
        args.rays, args.samples, args.mfma = preset["rays"] // preset["gpus"], preset["samples"], preset["mfma"]
        if args.gpus is None:
            args.gpus = preset["gpus"]
    args.gpus = args.gpus or 1
    args.rays, args.samples, args.mfma = args.rays or 4096, args.samples or 64, args.mfma or "f16x2"
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args.gpus)   # no launcher: be the launcher (before anything touches the GPU)
    if args.rehearsal:
        return rehearsal(args)

    import snerf_amd  # noqa: F401
    from snerf_amd import _lib, parallel
    from snerf_amd.framework.pipelines import load_pipeline, TrainLoop

    rank, world, device = parallel.init_distributed()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run "
                         f"--nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    L = _lib.lib()
    if args.serial_passes:
        from snerf_amd.semantic.components import rendering as _r0
        _r0.OVERLAP_SC_PASS = False
    torch.manual_seed(0)
    pz = preset or {}
    cfgs = make_cfgs(args.rays, args.samples, world, args.mfma, car_reg=pz.get("car_reg", False), vocab=pz.get("vocab", 50), model=pz.get("model", "semantic"))
    pipe = load_pipeline(cfgs)
    pipe.log_metrics = False  # the reference logs per-step scalars lazily; no host sync inside the timed region
    loop = TrainLoop(pipe, cfgs, device)
    if world > 1:
        loop.exchange_events = []     # HIP events around every step's gradient all-reduce (read after the timed region)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # per-step marks on the compute stream (no host sync inside the timed region) and the host's wall clock at the same points:
    # the line carries BOTH series in issue order, so a stall can be placed (which step) and attributed (device or host).
    import gc
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    host_t = [0.0] * (args.steps + 1)
    gc_was = gc.isenabled()
    step = 0
    for w in range(args.warmup):
        if w == max(0, args.warmup - 2):
            # The harness's own housekeeping goes HERE, behind a synchronize of its own and in front of the last warm-up steps -- not
            # between the warm-up and the timed region: gc.collect() takes tens of ms with torch loaded, the idle GPU drops its
            # clocks meanwhile and the first timed step then ran 3.3 ms long while they ramped (tools/first_step_probe.py: 29.5 ms
            # after 50 ms of idling, 26.4 ms after a bare synchronize).  A training run never idles the device like that.
            torch.cuda.synchronize()
            gc.collect()
            gc.freeze()          # nothing allocated before this point is ever scanned again
            gc.disable()         # no collector pause inside the timed region (re-enabled right after it)
        out = loop.step(step)      # held like in the timed loop: the warm-up must have the timed region's memory profile
        step += 1
    if args.warmup == 0:
        gc.collect(); gc.freeze(); gc.disable()
    barrier()
    mem0 = torch.cuda.memory_stats(device)
    t0 = time.perf_counter()
    marks[0].record()
    host_t[0] = t0
    for i in range(args.steps):
        out = loop.step(step)
        marks[i + 1].record()
        host_t[i + 1] = time.perf_counter()
        step += 1
    t_issued = time.perf_counter()
    barrier()
    dt = time.perf_counter() - t0
    exchange_ms = None
    if loop.exchange_events:
        ev = loop.exchange_events[-args.steps:]          # the timed steps' events (the warm-up's come first)
        exchange_ms = [a.elapsed_time(b) for a, b in ev]
        loop.exchange_events = None
    if gc_was:
        gc.enable()
    mem1 = torch.cuda.memory_stats(device)
    step_ms_issue = [marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)]
    host_step_ms = [(host_t[i + 1] - host_t[i]) * 1e3 for i in range(args.steps)]
    step_ms = sorted(step_ms_issue)
    median_ms = step_ms[len(step_ms) // 2] if len(step_ms) % 2 else 0.5 * (step_ms[len(step_ms) // 2 - 1] + step_ms[len(step_ms) // 2])
    i_max = max(range(args.steps), key=lambda i: step_ms_issue[i])
    timing = {
        "step_ms": [round(x, 3) for x in step_ms_issue],             # device time between consecutive step marks, issue order
        "host_step_ms": [round(x, 3) for x in host_step_ms],         # host wall time to ISSUE each step (launches are asynchronous)
        "device_sum_ms": sum(step_ms_issue), "wall_ms": dt * 1e3, "host_issue_ms": (t_issued - t0) * 1e3,
        "max_step_ms": step_ms_issue[i_max], "max_step_index": i_max,
        "host_max_step_ms": max(host_step_ms), "host_max_step_index": max(range(args.steps), key=lambda i: host_step_ms[i]),
        "allocator_delta": {k: int(mem1.get(k, 0) - mem0.get(k, 0)) for k in
                            ("num_alloc_retries", "num_device_alloc", "num_device_free", "num_ooms", "reserved_bytes.all.current")},
        "gc": "collected + frozen + disabled in front of the last two warm-up steps (the device does not idle between warm-up and timed region)",
    }
    # Roofline phase (after the timed region, so the headline number carries no event overhead): the same steps with
    # the main and solar-correction passes serialised -- in the timed region their kernels overlap on two HIP streams,
    # which would stretch every per-kernel duration -- and every GEMM launch bracketed by HIP events on its stream.
    prof = None
    prof_steps = 0
    if not args.no_profile:
        from snerf_amd.semantic.components import rendering as _rend
        saved = _rend.OVERLAP_SC_PASS
        _rend.OVERLAP_SC_PASS = False
        try:
            for _ in range(2):
                loop.step(step); step += 1
            torch.cuda.synchronize()
            L.snerf_profile_begin()
            prof_steps = max(1, min(args.steps, 10))
            for _ in range(prof_steps):
                loop.step(step); step += 1
            torch.cuda.synchronize()
            prof = _lib.SnerfProfile()
            _lib.check(L.snerf_profile_end(C.byref(prof)), "snerf_profile_end")
        finally:
            _rend.OVERLAP_SC_PASS = saved
    observed = None
    if world == 1 and not args.no_profile:
        observed, n_extra = power_probe(loop, step)
        step += n_extra
    gloo = world > 1 and torch.distributed.get_backend() == "gloo"   # rehearsal rigs; RCCL reduces on the device
    tmax = torch.tensor([dt], dtype=torch.float64, device="cpu" if gloo else device)
    if world > 1:
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
    dt = float(tmax.item())
    loss = float(out["loss"].detach())

    if pz.get("frame"):
        # configs[4]'s second half: full-frame inference a la eval/extract_pointcloud.py with the frame's rays sharded over the ranks
        # (eval/utils/util.py: sharded_lean_inference -- every rank renders ceil(n / world) rays, one all_gather per per-ray result)
        from snerf_amd.eval.utils.util import sharded_lean_inference
        from snerf_amd.framework.datasets import GpuRayBank
        from snerf_amd import ops as _ops
        del out
        _ops.release_workspaces()
        torch.cuda.empty_cache()
        n_frame = 640 * 640
        fb = GpuRayBank.synthetic(n_frame, 19, 5, 321, device=device)
        old_chunk = cfgs.pipeline.render_chunk_size
        cfgs.pipeline.render_chunk_size = 40960
        try:
            sharded_lean_inference(cfgs, pipe.renderer, pipe.models, fb.t["rays"][:40960 * world], fb.t["extras"][:40960 * world])
            barrier()
            tf0 = time.perf_counter()
            fr = sharded_lean_inference(cfgs, pipe.renderer, pipe.models, fb.t["rays"], fb.t["extras"])
            barrier()
            tf = time.perf_counter() - tf0
        finally:
            cfgs.pipeline.render_chunk_size = old_chunk
        tfm = torch.tensor([tf], dtype=torch.float64, device="cpu" if gloo else device)
        if world > 1:
            torch.distributed.all_reduce(tfm, op=torch.distributed.ReduceOp.MAX)
        frame_leg = {"rays": n_frame, "samples": args.samples, "world_size": world, "rays_per_s": n_frame / float(tfm.item()), "ms": float(tfm.item()) * 1e3,
                     "results": "rgb + depth + semantic_label of the WHOLE frame on every rank", "render_chunk_size": 40960,
                     "frame_rows_on_rank0": int(fr["rgb_coarse"].shape[0])}
        del fr, fb
    else:
        frame_leg = None
    if rank != 0:
        return
    mode = pipe.models["coarse"].spec.mfma
    rays_total = args.rays * world * args.steps
    value = rays_total / dt
    flops_step_gpu = (FLOPS_PER_SAMPLE_TRAIN_SATNERF if pz.get("model") == "satnerf" else FLOPS_PER_SAMPLE_TRAIN) * args.rays * args.samples
    reduced = {"f16x1": "f16 (REDUCED: one block-scaled fp16 plane)"}
    line = {
        "metric": f"train rays/sec ({args.rays} rays x {args.samples} samples)", "value": value, "unit": "rays/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "ms_per_step_median": median_ms,
        "higher_is_better": True, "scaling": "weak", "timing": timing,
        # BASELINE.md holds no published number for this metric (the reference publishes none): null by the contract.  The
        # north-star ratio against the reference's single-GPU PyTorch path measured in THIS run is `vs_reference_gpu_eager`.
        "vs_baseline": None,
        "dtype": reduced.get(mode, "f32(f16x2)"),   # fp32-class arithmetic on 16-bit matrix cores: see `arithmetic`
        "arithmetic": {
            "f16x2": "fp32-class: activations stored as two fp16 planes with one power-of-two exponent per 128 x 128 block (same bytes as "
                     "fp32), products hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_f16, fp32 accumulate",
            "f16x1": "REDUCED (the reference's precision = 16 runs): the same block-scaled tensors with ONE fp16 plane (11 significant bits, 2 bytes "
                     "per element), one product per contraction step on v_mfma_f32_32x32x16_f16, fp32 accumulate"}[mode],
        "data": "synthetic",
        "config": {"workload": ((preset["name"] + f" -- per GPU {args.rays} rays x {args.samples} samples, {args.mfma}; ") if preset else
                                "JAX_068 semantic pipeline (configs[1]): ") +
                               ("SatNeRF fc_units=512 x 8 layers (no encoding, no semantic head), main + solar-correction pass, SatNerfLoss + sc, "
                                if pz.get("model") == "satnerf" else
                                "RSSemanticNeRF fc_units=512 x 8 layers, C=5, " + ("" if preset else f"{args.rays} rays x {args.samples} samples per GPU, fp32, ") +
                                "main + solar-correction pass, SatNerfLoss + sc + SemanticLoss(ignore car)" + (" + L_t" if pz.get("car_reg") else "") + ", ") +
                               "Adam lr 5e-4; synthetic rays (SURVEY 8d), random-init SIREN weights",
                   "preset": args.config,
                   "rays_per_gpu": args.rays, "samples": args.samples, "global_batch": args.rays * world,
                   "parallelism": f"dp{world}", "final_loss": loss,
                   "streams": "main pass and solar-correction pass on two HIP streams" if not args.serial_passes else "single stream"},
    }
    if world > 1:   # what the collective layer actually was (the driver can check the rank count RCCL saw)
        import torch.distributed as dist
        line["config"]["distributed"] = {"backend": dist.get_backend(), "world_size": dist.get_world_size(),
                                         "nccl_version": ".".join(map(str, torch.cuda.nccl.version())) if dist.get_backend() == "nccl" else None,
                                         "gradient_bucket_floats": int(getattr(loop.optimizer, "numel", 0))}
        if exchange_ms:   # compute vs exchange, from the first SCALE record on: device time of the flat gradient all-reduce per step (rank 0)
            xs = sorted(exchange_ms)
            line["config"]["distributed"]["allreduce_ms_per_step"] = {"mean": sum(xs) / len(xs), "median": xs[len(xs) // 2], "max": xs[-1],
                                                                       "bytes": 4 * int(getattr(loop.optimizer, "numel", 0)),
                                                                       "note": "HIP events on the compute stream around dist.all_reduce(flat gradient bucket); "
                                                                               "includes waiting for the slowest rank's backward"}
    step_tflops = flops_step_gpu * args.steps / dt / 1e12  # per GPU, algorithmic (SURVEY 8d figure)
    if prof is not None:
        # dominant kernel = variant 0: the K-contiguous dense-layer launches (forward X.W^T and dX), ~63 % of device time
        ms, fl, n = prof.ms[0], prof.flops[0], prof.launches[0]
        alg = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0          # algorithmic: 2 I J K per launch (SURVEY 8d's FLOPs)
        mult, peak = {"f16x2": 3.0, "f16x1": 1.0}[mode], BF16_MFMA_PEAK_TFLOPS
        # HBM-side bytes per launch of the dominant kernel: NOT measured in this run (PMC passes need rocprofv3 runs of their own);
        # read from the newest committed PMC summary of this workload, with its provenance next to it
        traffic, traffic_source = None, None
        step_hbm, step_src = None, None
        for rr in ("r05", "r04", "r03", "r02"):
            tf = os.path.join(ROOT, "profiles", rr, "pmc_hbm_traffic.json")
            if os.path.isfile(tf) and mode == "f16x2" and args.rays == 4096 and args.samples == 64:
                try:
                    tj = json.load(open(tf))
                    traffic = tj.get("kc_bytes_per_launch")
                    traffic_source = {"file": f"profiles/{rr}/pmc_hbm_traffic.json", "commit": tj.get("commit"), "collected": tj.get("provenance", tj.get("note"))}
                    # whole-step HBM bytes: every kernel's (read + written) x its launches per step, from the same PMC passes
                    ks = tj.get("kernels", {}).get("bench step", {})
                    nsteps = tj.get("bench_steps_profiled")
                    if ks and not nsteps:      # older summaries: infer the profiled step count from the one-launch-per-step kernel
                        nsteps = next((v["launches"] for k, v in ks.items() if k.startswith("adam_kernel")), None)
                    if ks and nsteps:
                        step_hbm = sum((v["read_bytes"] + v["written_bytes"]) * v["launches"] for v in ks.values()) / nsteps
                        step_src = {"file": f"profiles/{rr}/pmc_hbm_traffic.json", "steps_profiled": nsteps,
                                    "how": "sum over kernels of (FETCH_SIZE + WRITE_SIZE bytes per launch) x launches per step; counters calibrated on a known copy"}
                    break
                except Exception:
                    traffic = None
        line["roofline"] = {
            "bound": "mfma", "achieved": alg, "peak": peak, "unit": "TFLOP/s",
            "frac": alg / peak,                              # ALGORITHMIC fraction: SURVEY 8(d) FLOPs / dense 16-bit MFMA peak
            "frac_mfma_issued": alg * mult / peak,           # matrix-pipe occupancy: `mult` 16-bit MFMA products per fp32 product
            "mfma_products_per_fp32_product": mult, "traffic": traffic, "traffic_source": traffic_source,
            "kernel": ("snerf::bsp::gemm_kc_kernel<PL> (persistent workgroups, two per CU, 128 x 256 tiles drawn from per-XCD counters; activations = "
                       "fp16 planes by LDS-DMA, W = fragment-ordered planes straight from L2 as the MFMA A operand; "
                       f"{int(mult)} x v_mfma_f32_32x32x16_f16 per 32x32x16 block; one-pass sine epilogue on the accumulators, planes + block "
                       "exponents out through LDS strips)"),
            "vs_fp32_mfma_peak": alg / FP32_MFMA_PEAK_TFLOPS,
            "power_capped": {"sustained_peak": FP16_MFMA_SUSTAINED_TFLOPS, "frac_of_sustained": alg / FP16_MFMA_SUSTAINED_TFLOPS,
                             "frac_mfma_issued_of_sustained": alg * mult / FP16_MFMA_SUSTAINED_TFLOPS,
                             "step_mfma_floor_ms": flops_step_gpu * mult / (FP16_MFMA_SUSTAINED_TFLOPS * 1e12) * 1e3,
                             "observed_in_this_run": observed,
                             "source": "profiles/r05/power_cap.md: the step holds the 1,400 W package cap at sclk ~1.57 GHz (2.4 nominal); a register-resident "
                                       "32x32x16 fp16 MFMA loop on real operands sustains 1.68-1.71 PFLOP/s there (2.48 on zeros); measured on MI355X, not from the guide"},
            "launches": int(n), "avg_launch_ms": ms / max(n, 1),
            "measured_over": f"{prof_steps} extra steps after the timed region, main and sc pass serialised (SNERF_OVERLAP_SC=0 behaviour), "
                             "HIP events on the launch stream around every GEMM launch",
            "per_variant": {(_lib.PROFILE_VARIANTS[v]): {
                "launches": int(prof.launches[v]), "avg_ms": prof.ms[v] / max(prof.launches[v], 1),
                "algorithmic_tflops": (prof.flops[v] / (prof.ms[v] * 1e-3) / 1e12) if prof.ms[v] > 0 else 0.0}
                for v in range(4) if prof.launches[v] > 0},
            "whole_step_algorithmic_tflops": step_tflops, "whole_step_frac": step_tflops / peak,
            # the whole step's memory account next to the matrix one: HBM bytes per step (PMC, committed summary) and the time
            # they take at the guide's achievable HBM rate -- the stored-activation, layer-per-launch design sits between this
            # floor and the three-product matrix floor at the same time
            "step_hbm_bytes": step_hbm, "step_hbm_source": step_src,
            "step_hbm_floor_ms": (step_hbm / HBM_ACHIEVABLE_BPS * 1e3) if step_hbm else None,
            "step_algorithmic_hbm_bytes": ALGORITHMIC_BYTES_PER_RAY * args.rays,
            "step_mfma_floor_ms": flops_step_gpu * mult / (peak * 1e12) * 1e3,
        }
    else:
        line["roofline"] = {"bound": "mfma", "achieved": step_tflops, "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                            "frac": step_tflops / BF16_MFMA_PEAK_TFLOPS, "traffic": None,
                            "kernel": "whole step (algorithmic FLOPs / wall time); per-kernel timing disabled"}
    if frame_leg is not None:
        line["inference_sharded"] = frame_leg
    if world == 1 and not args.no_inference:
        line["inference"] = inference_rates(pipe, cfgs, device, args.samples)
    if world == 1 and not args.no_reduced and mode == "f16x2" and preset is None:
        # BASELINE configs[2] / [4] name reduced precision (the reference's `precision = 16`): the same step in the one-plane mode at
        # their per-GPU shapes, and at the headline shape (what the reference's template default float32_matmul_precision = "high" maps to
        # under mfma_precision = "auto").  Reported here, never in `value`.
        del loop, pipe
        line["reduced_precision"] = [reduced_precision_leg(4096, S, device) for S in (64, 96, 128)]
        loop = pipe = None
    if world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(args.samples)
    if world == 1 and not args.no_eager_gpu_baseline:
        loop = pipe = None
        from snerf_amd import ops as _ops
        _ops.release_workspaces()
        torch.cuda.empty_cache()
        line["reference_gpu_eager"] = eager_gpu_baseline(args.rays, args.samples, device)
        line["vs_reference_gpu_eager"] = value / line["reference_gpu_eager"]["value"]
    print(json.dumps(line))


if __name__ == "__main__":
    main()
