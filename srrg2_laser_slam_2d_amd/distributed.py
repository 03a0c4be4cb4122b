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


def shard_by_work(work, world: int) -> list[tuple[int, int]]:
    """Contiguous partition of the candidates 0 .. len(work) over `world` ranks by WORK instead of by count: boundary r sits where the running sum
    of `work` (lsm2d_estimate_work: what each alignment will stream) passes r / world of the total -- with the exact culling an alignment's time
    follows that number (33-59 % of the map's chunks survive on configs[1]), so equal counts are not equal work.  Deterministic: every rank derives
    the same partition from the same estimates, no collective.  All-zero or empty work falls back to shard_range.  Returns [(lo, hi)] * world."""
    w = np.maximum(np.asarray(work, np.float64).ravel(), 0.0)
    n = len(w)
    tot = float(w.sum())
    if n == 0 or not tot > 0.0:
        return [shard_range(n, r, world) for r in range(world)]
    cum = np.concatenate([[0.0], np.cumsum(w)])
    cuts = [0]
    for r in range(1, world):
        k = int(np.searchsorted(cum, tot * r / world, side="left"))        # first prefix that reaches the target ...
        if k > 0 and abs(cum[k - 1] - tot * r / world) <= abs(cum[min(k, n)] - tot * r / world):
            k -= 1                                                          # ... or the one before it, whichever is nearer
        cuts.append(min(max(k, cuts[-1]), n))
    cuts.append(n)
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


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


def cross_rank_check(scan_points, scan_offsets, scan_index, x0, poses, n_beams: int, align_fn, n_check: int = 16, device: str | None = None):
    """Sharding must change WHERE an alignment runs, never its result.  Every rank contributes its first ``n_check`` candidates
    (scan, initial guess) and the poses it got for them; rank 0 re-aligns each rank's candidates alone with
    ``align_fn(clouds: list[np.ndarray], x0: np.ndarray[n, 3]) -> np.ndarray[n, 3]`` and compares BIT FOR BIT.
    Returns (n_ranks_identical, world, n_checked) on rank 0, None elsewhere.  Collectives: three all_gathers of fixed-size rows."""
    import torch.distributed as dist
    world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
    rank = dist.get_rank() if world > 1 or (dist.is_available() and dist.is_initialized()) else 0
    ncheck = min(n_check, len(x0))
    # counts travel as int32 next to the number of candidates a rank really checks: an empty scan stays in its place (row i of the clouds
    # belongs to row i of x0 / poses), and a rank with fewer than n_check candidates says so itself
    pts = np.zeros((n_check, n_beams, 4), np.float32); cnt = np.zeros((n_check + 2, 1), np.int32); head = np.zeros((n_check, 6), np.float32)
    too_long = 0
    for i in range(ncheck):
        si = int(scan_index[i]) if scan_index is not None else i
        sc = scan_points[scan_offsets[si]:scan_offsets[si + 1]]
        if len(sc) > n_beams:        # no room for it in the fixed-size rows: every rank must learn of it TOGETHER (a rank that raised here on its own
            too_long = max(too_long, len(sc)); continue      # would leave the others waiting in the collectives below until the transport times out)
        pts[i, : len(sc)] = sc; cnt[i, 0] = len(sc)
    cnt[n_check, 0] = ncheck; cnt[n_check + 1, 0] = too_long
    head[:ncheck, :3] = poses[:ncheck]; head[:ncheck, 3:] = x0[:ncheck]
    all_pts = gather_results(pts.reshape(n_check * n_beams, 4), device).reshape(world, n_check, n_beams, 4)
    all_cnt = gather_results(cnt, device).reshape(world, n_check + 2).astype(np.int64)
    all_head = gather_results(head, device).reshape(world, n_check, 6)
    if all_cnt[:, n_check + 1].max() > 0:      # validated collectively: the same exception on every rank, after the last collective
        bad = int(np.argmax(all_cnt[:, n_check + 1]))
        raise ValueError("cross_rank_check: rank %d holds a scan of %d points, more than the n_beams = %d rows reserved for it" % (bad, all_cnt[bad, n_check + 1], n_beams))
    if rank != 0:
        return None
    same = 0
    for r in range(world):
        nr = int(all_cnt[r, n_check])
        clouds = [np.ascontiguousarray(all_pts[r, i, : all_cnt[r, i]]) for i in range(nr)]
        got = align_fn(clouds, np.ascontiguousarray(all_head[r, :nr, 3:]))
        same += int(np.array_equal(np.asarray(got, np.float32), all_head[r, :nr, :3]))
    return same, world, ncheck
