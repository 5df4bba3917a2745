"""Multi-GPU glue of the T2S path: one process per GPU.

Sampling shards embarrassingly: the diffusion loop has NO data-path collective: every
series is independent through all steps, so rank r samples global rows [lo, hi) with the Philox
stream keyed by the GLOBAL row index -- results are bit-identical for any number of GPUs.  Training
is data parallel with ONE all-reduce of the flat gradient bucket per step (t2ms_amd/train.py).
torch.distributed (backend "nccl" = RCCL over xGMI on GPUs, "gloo" on CPU for tests) carries the
barrier, the max-over-ranks wall time, the seed broadcast, that gradient all-reduce and the
optional final gather of the (B, L) series.

Three environment overrides exist for rehearsing the N>1 code on ONE GPU: T2S_DIST_BACKEND=gloo replaces
RCCL (which refuses two ranks on one device; device tensors are then staged through host memory by the
helpers below) and T2S_SHARE_GPU=1 maps LOCAL_RANK onto the visible devices modulo their count
(tests/test_two_ranks_one_gpu.py); T2S_FORCE_DIST=1 builds the process group at world size 1, so the same
collectives run through RCCL itself on one rank (tests/test_rccl_one_rank.py).
"""
from __future__ import annotations

import os
from typing import List, Optional, Tuple

import torch


def env_world() -> Tuple[int, int, int]:
    """(rank, local_rank, world_size) from the torchrun environment (1 process if unset)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def local_device_index() -> int:
    """The GPU this rank drives: LOCAL_RANK (one process per GPU).  With T2S_SHARE_GPU=1 several
    ranks may share a device (LOCAL_RANK modulo the visible device count) -- rehearsal only."""
    _, local_rank, _ = env_world()
    n = torch.cuda.device_count()
    if n > 0 and local_rank >= n:
        if os.environ.get("T2S_SHARE_GPU") != "1":
            raise RuntimeError(f"LOCAL_RANK={local_rank} but only {n} GPU(s) are visible: launch one rank per GPU "
                               f"(set T2S_SHARE_GPU=1 to rehearse several ranks on one device)")
        return local_rank % n
    return local_rank


def init(backend: Optional[str] = None, device: Optional[torch.device] = None):
    """Initialise torch.distributed when WORLD_SIZE > 1; returns the module or None.
    T2S_FORCE_DIST=1 initialises the process group at WORLD_SIZE = 1 too: a one-rank RCCL communicator is legal, and
    it is how the collectives of the N > 1 code (device-side all-reduce of the gradient bucket, all_gather, barrier,
    object broadcast) run through RCCL itself on a one-GPU box (tests/test_rccl_one_rank.py)."""
    rank, _, world = env_world()
    if world == 1 and os.environ.get("T2S_FORCE_DIST", "0") in ("", "0"):
        return None
    import torch.distributed as dist
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    if backend is None:
        backend = "nccl" if (device is not None and device.type == "cuda") else "gloo"
    backend = os.environ.get("T2S_DIST_BACKEND", backend)
    if not dist.is_initialized():
        kw = {"device_id": device} if backend == "nccl" and device is not None else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return dist


def _staged(dist, t: torch.Tensor) -> bool:
    """gloo moves host memory: device tensors are staged through the CPU (rehearsal / tests only)."""
    return t.is_cuda and dist.get_backend() == "gloo"


def all_reduce_sum(dist, t: torch.Tensor) -> torch.Tensor:
    """In-place SUM all-reduce of `t` (RCCL on device memory; staged through the host under gloo)."""
    if dist is None:
        return t
    if _staged(dist, t):
        h = t.detach().cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def broadcast_int(dist, value: int, src: int = 0) -> int:
    """The int rank `src` holds, on every rank (host-side control value: seeds, counters)."""
    if dist is None:
        return int(value)
    box = [int(value)]
    dist.broadcast_object_list(box, src=src)
    return int(box[0])


def shard_rows(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous rows [lo, hi) of a `total`-row batch owned by `rank` (remainder to low ranks)."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def barrier(dist, device: Optional[torch.device] = None) -> None:
    if device is not None and device.type == "cuda":
        torch.cuda.synchronize(device)
    if dist is not None:
        dist.barrier()
    if device is not None and device.type == "cuda":
        torch.cuda.synchronize(device)


def max_over_ranks(dist, value: float, device: Optional[torch.device] = None) -> float:
    if dist is None:
        return value
    on = device if (device is not None and dist.get_backend() != "gloo") else "cpu"
    t = torch.tensor([value], dtype=torch.float64, device=on)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_rows(dist, local: torch.Tensor, total: int, rank: int, world: int) -> Optional[torch.Tensor]:
    """Concatenate per-rank (rows_r, ...) tensors on rank 0 in global row order (None elsewhere).
    Shards may be ragged; uses all_gather on tensors padded to the largest shard."""
    if dist is None:
        return local
    sizes = [shard_rows(total, r, world) for r in range(world)]
    maxrows = max(hi - lo for lo, hi in sizes)
    src = local.detach().cpu() if _staged(dist, local) else local
    pad = torch.zeros((maxrows,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    pad[: src.shape[0]] = src
    outs: List[torch.Tensor] = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(outs, pad)
    if rank != 0:
        return None
    return torch.cat([o[: hi - lo] for o, (lo, hi) in zip(outs, sizes)], dim=0).to(local.device)


def gather_ragged(dist, local: torch.Tensor, counts: List[int], rank: int) -> Optional[List[torch.Tensor]]:
    """Per-rank (counts[r], ...) tensors -> on rank 0 the list of every rank's tensor (None elsewhere).  `counts` is known
    on every rank (the shard sizes are a function of the job, not of the data), a rank may hold zero rows and still joins.
    ONE all_gather of tensors padded to the largest count."""
    if dist is None:
        return [local]
    if local.shape[0] != counts[rank]:
        raise ValueError(f"gather_ragged: rank {rank} holds {local.shape[0]} rows, expected {counts[rank]}")
    maxrows = max(max(counts), 1)
    src = local.detach().cpu() if _staged(dist, local) else local.detach()
    pad = torch.zeros((maxrows,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    pad[: src.shape[0]] = src
    outs: List[torch.Tensor] = [torch.empty_like(pad) for _ in counts]
    dist.all_gather(outs, pad)
    if rank != 0:
        return None
    return [o[:c] for o, c in zip(outs, counts)]
