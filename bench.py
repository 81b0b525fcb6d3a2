#!/usr/bin/env python3
"""Headline benchmark: train rays/s of the semantic Sat-NeRF hot path on synthetic ray batches.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One step = one full optimiser step of BASELINE.json configs[1] ("JAX_068 semantic pipeline, 4096 rays x 64
samples, fp32") per GPU: on-device batch sampling -> main + solar-correction forward -> SatNerfLoss +
solar correction + SemanticLoss (fused HIP loss kernels) -> backward -> flat gradient all-reduce (RCCL,
N > 1) -> Adam.  Weak scaling: 4096 rays per GPU.  Prints ONE JSON line (rank 0).
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
FP32_MFMA_PEAK_TFLOPS = 157.3        # /opt/skills/guides/MI355X_MICROARCH.md "Peak FP32 (matrix)"
BF16_MFMA_PEAK_TFLOPS = 2500.0       # same guide: "Peak BF16/FP16 MFMA ~2.5 PF dense"


def make_cfgs(rays_per_gpu, samples, world, mfma="f16x2"):
    from snerf_amd.framework.configs import MainConfig
    pipeline = {
        "pipeline": "snerf_amd.semantic.pipelines.rs_semantic.RSSemanticPipeline",
        # configs/pipelines/rs_semantic.toml values
        "n_samples": samples, "batch_size": rays_per_gpu * world, "render_chunk_size": 1 << 22,
        "learnrate": 5e-4, "fc_units": 512, "fc_layers": 8, "fc_skips": [4], "activation_function": "siren",
        "mapping_pos_n_freq": 10, "sc_lambda": 0.05, "t_embedding_vocab": 50, "t_embedding_tau": 4,
        "lambda_s": 0.04, "semantic_activation_function": "sigmoid", "ignore_car_index": True,
        # steady state of training: beta loss active (epoch >= first_beta_epoch), depth rays dropped
        # (after 25 % of the steps, baseline/pipelines/satnerf.py:26-29) -- SURVEY.md 8(d)
        "first_beta_epoch": 0, "depth_enabled": False, "mfma_precision": mfma,
    }
    run = {"max_train_steps": 1 << 30, "synthetic_rays": max(1 << 20, rays_per_gpu * world * 4), "synthetic_images": 19,
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
    cores = min(cores, 16)  # the GPU box's CPU share for one GPU; more threads only oversubscribe the quota
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
    return {"value": n / dt, "unit": "train rays/s", "cores": cores, "kind": "port",
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--rays", type=int, default=4096, help="rays per GPU")
    ap.add_argument("--samples", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="skip the per-GEMM HIP-event timing")
    ap.add_argument("--serial-passes", action="store_true", help="run main and sc pass on one stream (as the roofline phase does)")
    ap.add_argument("--mfma", default="f16x2", choices=["split3", "f16x2", "fp32", "split2", "bf16", "split3_bwd2"],
                    help="matrix arithmetic: f16x2 (default, headline) and split3 are fp32-class; split2 / split3_bwd2 / bf16 are the REDUCED-precision "
                         "modes of BASELINE configs[2]/[4] (reported under their own dtype, never as the fp32 headline)")
    ap.add_argument("--eager-gpu-baseline", action="store_true", help="also time the oracle with stock PyTorch ops on the GPU")
    args = ap.parse_args()

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
    cfgs = make_cfgs(args.rays, args.samples, world, args.mfma)
    pipe = load_pipeline(cfgs)
    pipe.log_metrics = False  # the reference logs per-step scalars lazily; no host sync inside the timed region
    loop = TrainLoop(pipe, cfgs, device)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    step = 0
    for _ in range(args.warmup):
        loop.step(step)
        step += 1
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = loop.step(step)
        step += 1
    barrier()
    dt = time.perf_counter() - t0
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
    gloo = world > 1 and torch.distributed.get_backend() == "gloo"   # rehearsal rigs; RCCL reduces on the device
    tmax = torch.tensor([dt], dtype=torch.float64, device="cpu" if gloo else device)
    if world > 1:
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
    dt = float(tmax.item())
    loss = float(out["loss"].detach())

    if rank != 0:
        return
    mode = pipe.models["coarse"].spec.mfma
    rays_total = args.rays * world * args.steps
    value = rays_total / dt
    flops_step_gpu = FLOPS_PER_SAMPLE_TRAIN * args.rays * args.samples
    line = {
        "metric": "train rays/sec (4096 rays x 64 samples)", "value": value, "unit": "rays/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None,
        # storage and accumulation are fp32 in every mode; `arithmetic` says how the products are formed
        "dtype": {"f16x2": "f32", "split3": "f32", "fp32": "f32", "split3_bwd2": "f32 forward / REDUCED backward",
                  "split2": "REDUCED (fp32 as two bf16, ~16 bits)", "bf16": "bf16 (REDUCED)"}[mode],
        "arithmetic": {"split3": "fp32 storage and accumulate; products on bf16 MFMA via 3-plane splits, fp32-level accuracy",
                       "f16x2": "fp32 storage and accumulate; products on fp16 MFMA via 2-plane splits of power-of-two-scaled operands, fp32-level accuracy",
                       "fp32": "v_mfma_f32_32x32x2_f32", "split2": "REDUCED: f32 storage/accumulate, operands as 2 bf16 planes (~16 bits, torch 'high')",
                       "bf16": "REDUCED: bf16 operands, f32 accumulate and storage (torch 'medium' / precision=16)",
                       "split3_bwd2": "f32 forward (3-plane splits, fp32-level results); REDUCED backward: 2 bf16 planes (~16 bits) in dX / dW"}[mode],
        "data": "synthetic",
        "config": {"workload": "JAX_068 semantic pipeline (configs[1]): RSSemanticNeRF fc_units=512 x 8 layers, C=5, "
                               f"{args.rays} rays x {args.samples} samples per GPU, fp32, main + solar-correction pass, "
                               "SatNerfLoss + sc + SemanticLoss(ignore car), Adam lr 5e-4; synthetic rays (SURVEY 8d), "
                               "random-init SIREN weights",
                   "rays_per_gpu": args.rays, "samples": args.samples, "global_batch": args.rays * world,
                   "parallelism": f"dp{world}", "final_loss": loss,
                   "streams": "main pass and solar-correction pass on two HIP streams" if not args.serial_passes else "single stream"},
    }
    step_tflops = flops_step_gpu * args.steps / dt / 1e12  # per GPU, algorithmic (SURVEY 8d figure)
    if prof is not None:
        x6 = mode != "fp32"
        # dominant kernel = variant 0: the K-contiguous launches (forward X.W^T and dX; gemm_wide_kernel), ~61 % of device time
        ms, fl, n = prof.ms[0], prof.flops[0], prof.launches[0]
        fp32_eq = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        fam_ms = sum(prof.ms[v] for v in range(3)); fam_fl = sum(prof.flops[v] for v in range(3))
        # split kernel: the contraction at this accuracy IS `mult` 16-bit MFMA products per fp32 product (3 on fp16
        # planes of scaled operands, 6 on bf16 planes), so the kernel's algorithmic work is mult x (2 I J K) flops,
        # priced against the dense bf16/fp16 MFMA peak
        mult, peak = ({"split3": 6.0, "f16x2": 3.0, "split2": 3.0, "bf16": 1.0, "split3_bwd2": 4.0}[mode], BF16_MFMA_PEAK_TFLOPS) if x6 else (1.0, FP32_MFMA_PEAK_TFLOPS)
        achieved = fp32_eq * mult
        # HBM bytes per launch of that kernel from the PMC passes (2 x FETCH_SIZE + WRITE_SIZE, the gfx950 correction of
        # the MI355X guide), measured on this exact workload: profiles/r01/pmc_hbm_traffic.md
        traffic = 1.298e9 if (mode == "f16x2" and args.rays == 4096 and args.samples == 64) else None
        line["roofline"] = {
            "bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak, "traffic": traffic,
            "kernel": (("snerf::gemm_wide_kernel (128x256 tile; two fp16 planes of power-of-two-scaled fp32 operands, hh + hl + lh = 3 x "
                        "v_mfma_f32_32x32x16_f16 per 32x32x16 block, fp32 accumulate; forward X.W^T and dX launches; the two launches per "
                        "step whose width does not fit run the 128x128 tile of gemm_x6_kernel and are averaged in)" if mode == "f16x2" else
                        "snerf::gemm_x6_kernel<false,true,NP,128> (split-bf16: NP bf16 planes per fp32 operand, 6 / 3 / 1 x v_mfma_f32_32x32x16_bf16 per "
                        "32x32x16 block for NP = 3 / 2 / 1, fp32 accumulate; forward X.W^T and dX launches)") if x6 else
                       "snerf::gemm_kernel<128,128,64,64,false,false> (v_mfma_f32_32x32x2_f32)"),
            "fp32_equivalent_tflops": fp32_eq, "vs_fp32_mfma_peak": fp32_eq / FP32_MFMA_PEAK_TFLOPS,
            "launches": int(n), "avg_launch_ms": ms / max(n, 1),
            "measured_over": f"{prof_steps} extra steps after the timed region, main and sc pass serialised (SNERF_OVERLAP_SC=0 behaviour)",
            "all_128x128_gemms": {"ms_per_step": fam_ms / max(prof_steps, 1), "fp32_equivalent_tflops": fam_fl / (fam_ms * 1e-3) / 1e12 if fam_ms > 0 else 0.0},
            "per_variant": {(_lib.PROFILE_VARIANTS[v]): {
                "launches": int(prof.launches[v]), "avg_ms": prof.ms[v] / max(prof.launches[v], 1),
                "fp32_equivalent_tflops": (prof.flops[v] / (prof.ms[v] * 1e-3) / 1e12) if prof.ms[v] > 0 else 0.0}
                for v in range(4) if prof.launches[v] > 0},
            "whole_step_fp32_equivalent_tflops": step_tflops, "whole_step_vs_fp32_mfma_peak": step_tflops / FP32_MFMA_PEAK_TFLOPS,
        }
    else:
        line["roofline"] = {"bound": "mfma", "achieved": step_tflops, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                            "frac": step_tflops / FP32_MFMA_PEAK_TFLOPS, "traffic": None,
                            "kernel": "whole step (algorithmic FLOPs / wall time); per-kernel timing disabled"}
    if world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(args.samples)
    if world == 1 and args.eager_gpu_baseline:
        del loop, pipe
        torch.cuda.empty_cache()
        line["torch_eager_gpu_baseline"] = eager_gpu_baseline(args.rays, args.samples, device)
        line["torch_eager_gpu_baseline"]["speedup_of_value"] = value / line["torch_eager_gpu_baseline"]["value"]
    print(json.dumps(line))


if __name__ == "__main__":
    main()
