"""Data-parallel batch sharding for the DPT path: one process per GPU, replicated weights, contiguous batch shards,
and ONE collective - an all-gather of the output depth maps (RCCL over xGMI when the backend is "nccl").

The reference has no multi-GPU code at all; every image is independent (eval mode, no cross-sample op,
reference dpt_model.py:57-83), so nothing on the data path needs a collective except handing every rank the full
[B,H,W] result. The helpers are backend-agnostic so the N>1 logic is covered on CPU with gloo (tests/test_parallel.py).
"""

from __future__ import annotations

import os

import torch
import torch.distributed as dist


def env_rank_world() -> tuple[int, int, int]:
    """(rank, world_size, local_rank) from the torch.distributed.run environment (1-process defaults)."""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def init_distributed(backend: str | None = None) -> tuple[int, int, int]:
    """Initialise the default process group from env:// (no-op for a single process)."""
    rank, world, local_rank = env_rank_world()
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        kwargs = {}
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            kwargs["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend=backend, init_method="env://", rank=rank, world_size=world, **kwargs)
    return rank, world, local_rank


def shard_bounds(global_batch: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous [start, stop) slice of the global batch owned by `rank` (ragged tails go to the low ranks)."""
    base, extra = divmod(global_batch, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def all_gather_maps(local_maps: torch.Tensor, world: int, out: torch.Tensor | None = None) -> torch.Tensor:
    """Every rank contributes [b,H,W] (equal b) and receives the rank-ordered concatenation [world*b,H,W]."""
    if world == 1:
        return local_maps
    local_maps = local_maps.contiguous()
    if out is None:
        out = torch.empty((world * local_maps.shape[0], *local_maps.shape[1:]), dtype=local_maps.dtype, device=local_maps.device)
    dist.all_gather_into_tensor(out, local_maps)
    return out


def all_gather_ragged(local_maps: torch.Tensor, global_batch: int, rank: int, world: int) -> torch.Tensor:
    """All-gather for shard sizes that differ by at most one image: pad to the largest shard, gather, then drop pads."""
    if world == 1:
        return local_maps
    bmax = -(-global_batch // world)
    pad = bmax - local_maps.shape[0]
    if pad:
        local_maps = torch.cat((local_maps, local_maps.new_zeros((pad, *local_maps.shape[1:]))), dim=0)
    gathered = all_gather_maps(local_maps, world).view(world, bmax, *local_maps.shape[1:])
    parts = []
    for r in range(world):
        s, e = shard_bounds(global_batch, r, world)
        parts.append(gathered[r, : e - s])
    return torch.cat(parts, dim=0)


class DataParallelDepth:
    """forward(global batch on this rank's view) = local shard through the model + all-gather of the depth maps."""

    def __init__(self, model, rank: int, world: int):
        self.model, self.rank, self.world = model, rank, world
        self._out = None

    def forward_shard(self, local_images: torch.Tensor) -> torch.Tensor:
        y = self.model(local_images)
        if self.world == 1:
            return y
        shape = (self.world * y.shape[0], *y.shape[1:])
        if self._out is None or self._out.shape != shape or self._out.dtype != y.dtype:
            self._out = torch.empty(shape, dtype=y.dtype, device=y.device)
        return all_gather_maps(y, self.world, self._out)
