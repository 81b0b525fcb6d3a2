"""Data parallelism: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on ROCm).

The path shards over rays with ONE exchange step per optimiser step: a sum all-reduce of one flat
gradient bucket (2.83 M fp32 = 11.3 MB; ring over xGMI ~0.13 ms).  Loss means with data-dependent
denominators are made shard-invariant by all-reducing the 16-float sums/counts vector of the fused
loss between its two phases (loss_ops.py), so gradients are summed, never averaged."""
import os

import torch
import torch.distributed as dist


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def init_distributed(backend: str = None):
    """Reads RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torch.distributed.run). Returns (rank, world, device)."""
    ws = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_cuda = torch.cuda.is_available()
    if use_cuda and local >= torch.cuda.device_count():
        # rehearsal rigs only (several gloo ranks sharing one card): production launches have one GPU per rank
        if (backend or os.environ.get("SNERF_DIST_BACKEND")) != "gloo":
            raise RuntimeError(f"LOCAL_RANK {local} but only {torch.cuda.device_count()} GPU(s) visible")
        local = local % torch.cuda.device_count()
    device = torch.device(f"cuda:{local}") if use_cuda else torch.device("cpu")
    if use_cuda:
        torch.cuda.set_device(device)
    if ws > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        backend = backend or os.environ.get("SNERF_DIST_BACKEND") or ("nccl" if use_cuda else "gloo")
        dist.init_process_group(backend, rank=rank, world_size=ws)
    return rank, ws, device


def _all_reduce(t: torch.Tensor):
    if t.is_cuda and dist.get_backend() == "gloo":  # test rigs only: gloo ranks sharing one GPU
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)


def allreduce_sum_(t: torch.Tensor) -> torch.Tensor:
    if world()[1] > 1:
        _all_reduce(t)
    return t


def frame_shard(n: int, rank: int = None, world_size: int = None):
    """contiguous slice [lo, hi) of a frame of n rays for this rank: ceil(n / world) rays per rank, the last ranks take the ragged
    tail (possibly nothing).  SURVEY 8(e): "shard the H*W rays of each image across ranks"."""
    r, w = world()
    rank = r if rank is None else rank
    world_size = w if world_size is None else world_size
    per = -(-n // world_size)
    lo = min(rank * per, n)
    return lo, min(lo + per, n)


def allgather_rows(local: torch.Tensor, n_total: int) -> torch.Tensor:
    """Rows of a per-ray result, sharded by frame_shard, -> the full (n_total, ...) tensor on EVERY rank: one all_gather of
    equal ceil(n / world)-row pieces (short shards are padded, the padding is cut off again).  Single process: the input."""
    _, w = world()
    if w == 1:
        return local
    per = -(-n_total // w)
    piece = local
    if local.shape[0] != per:
        piece = local.new_zeros((per,) + tuple(local.shape[1:]))
        piece[:local.shape[0]] = local
    piece = piece.contiguous()
    hop = piece.is_cuda and dist.get_backend() == "gloo"      # test rigs only: gloo ranks sharing one GPU
    src = piece.cpu() if hop else piece
    parts = [torch.empty_like(src) for _ in range(w)]
    dist.all_gather(parts, src)
    full = torch.cat(parts, 0)[:n_total]
    return full.to(local.device) if hop else full


def allreduce_gradients(params, flat_buffer: torch.Tensor = None) -> torch.Tensor:
    """Sum-all-reduce every .grad through one flat bucket (a single RCCL call): one cat kernel in, one fused
    multi-tensor copy out. Returns the bucket."""
    params = [p for p in params if p.grad is not None]
    if world()[1] == 1 or not params:
        return flat_buffer
    grads = [p.grad for p in params]
    flat = torch.cat([g.reshape(-1) for g in grads])
    _all_reduce(flat)
    torch._foreach_copy_(grads, [v.view_as(g) for v, g in zip(flat.split([g.numel() for g in grads]), grads)])
    return flat
