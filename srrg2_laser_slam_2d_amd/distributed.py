"""Multi-GPU layout of a batch of alignments: independent units, no data-path collective.

The reference runs loop-closure / relocalisation candidates in a sequential loop
(MultiLoopDetectorBruteForce2D, configurations/stage_segway_double_config_MULTI.json:964-986); here they are
sharded over one process per GPU.  The only exchange is the one-off RCCL broadcast of the shared local map
(1.6 MB at 1e5 points, 16 MB at 1e6) and, if the caller wants all results everywhere, a final all_gather.
"""
from __future__ import annotations

import numpy as np


def shard_range(n_items: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous block partition [lo, hi) of n_items over `world` ranks (sizes differ by at most 1)."""
    lo = (n_items * rank) // world
    hi = (n_items * (rank + 1)) // world
    return lo, hi


def broadcast_map(map_points, n_points: int, local_rank: int = 0, src: int = 0, device: str | None = None):
    """Rank `src` passes the map as float32 [N, 4]; every rank gets it as a device tensor.
    With an initialised process group this is one broadcast over RCCL/xGMI (gloo on CPU in tests)."""
    import torch
    import torch.distributed as dist
    if device is None:
        device = f"cuda:{local_rank}" if torch.cuda.is_available() else "cpu"
    if dist.is_available() and dist.is_initialized():
        if dist.get_rank() == src:
            t = torch.from_numpy(np.ascontiguousarray(map_points, np.float32)).to(device)
        else:
            t = torch.empty((n_points, 4), dtype=torch.float32, device=device)
        dist.broadcast(t, src=src)
        return t
    return torch.from_numpy(np.ascontiguousarray(map_points, np.float32)).to(device)


def gather_results(local: np.ndarray, device: str | None = None) -> np.ndarray:
    """all_gather of per-rank result rows (equal row counts per rank), rank order."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local
    if device is None:
        device = "cuda" if torch.cuda.is_available() and dist.get_backend() == "nccl" else "cpu"
    t = torch.from_numpy(np.ascontiguousarray(local)).to(device)
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return torch.cat(out, 0).cpu().numpy()
