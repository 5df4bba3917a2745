"""Multi-GPU glue for the sampling path: one process per GPU, batches shard embarrassingly.

The diffusion loop has NO data-path collective: every series is independent
through all steps, so rank r samples global rows [lo, hi) with the Philox stream keyed by the
GLOBAL row index -- results are bit-identical for any number of GPUs.  torch.distributed
(backend "nccl" = RCCL over xGMI on GPUs, "gloo" on CPU for tests) is used only for the
barrier, the max-over-ranks wall time and the optional final gather of the (B, L) series.
"""
from __future__ import annotations

import os
from typing import List, Optional, Tuple

import torch


def env_world() -> Tuple[int, int, int]:
    """(rank, local_rank, world_size) from the torchrun environment (1 process if unset)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init(backend: Optional[str] = None, device: Optional[torch.device] = None):
    """Initialise torch.distributed when WORLD_SIZE > 1; returns the module or None."""
    rank, _, world = env_world()
    if world == 1:
        return None
    import torch.distributed as dist
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend is None:
        backend = "nccl" if (device is not None and device.type == "cuda") else "gloo"
    if not dist.is_initialized():
        kw = {"device_id": device} if backend == "nccl" and device is not None else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return dist


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
    t = torch.tensor([value], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_rows(dist, local: torch.Tensor, total: int, rank: int, world: int) -> Optional[torch.Tensor]:
    """Concatenate per-rank (rows_r, ...) tensors on rank 0 in global row order (None elsewhere).
    Shards may be ragged; uses all_gather on tensors padded to the largest shard."""
    if dist is None:
        return local
    sizes = [shard_rows(total, r, world) for r in range(world)]
    maxrows = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((maxrows,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    outs: List[torch.Tensor] = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(outs, pad)
    if rank != 0:
        return None
    return torch.cat([o[: hi - lo] for o, (lo, hi) in zip(outs, sizes)], dim=0)
