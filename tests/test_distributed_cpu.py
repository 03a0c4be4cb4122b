"""N > 1 path on the CPU: gloo, world_size 2 -- alignment sharding, map broadcast, result gather."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from srrg2_laser_slam_2d_amd import distributed, synth


def test_shard_ranges_partition_exactly():
    for n in (0, 1, 7, 1000, 65536):
        for world in (1, 2, 3, 8):
            r = [distributed.shard_range(n, k, world) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(r, r[1:]))
            sizes = [hi - lo for lo, hi in r]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    w = synth.make_world(0)
    m = synth.make_map(w, 5000) if rank == 0 else None
    t = distributed.broadcast_map(m, 5000, device="cpu")
    ref = synth.make_map(w, 5000)
    assert t.shape == (5000, 4) and np.array_equal(t.numpy(), ref)
    # each rank "aligns" its shard of 11 candidates; results are gathered in rank order
    lo, hi = distributed.shard_range(12, rank, world)
    local = np.arange(lo, hi, dtype=np.float32)[:, None] * np.ones((1, 3), np.float32)
    allr = distributed.gather_results(local, device="cpu")
    assert np.array_equal(allr[:, 0], np.arange(12, dtype=np.float32))
    # max-over-ranks timing reduction used by bench.py
    tt = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    assert tt.item() == float(world)
    dist.barrier()
    dist.destroy_process_group()
    open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")


def test_gloo_world_size_2(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok0").exists() and (tmp_path / "ok1").exists()


def _align_worker(rank, world, port, out_dir):
    """Every rank ALIGNS its shard of a candidate sweep (with the CPU oracle standing in for the device -- tests may use it) and the
    ranks then run bench.py's cross-rank check: rank 0 re-aligns each rank's first candidates alone and compares bit for bit."""
    from oracle import pyoracle as po
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    world_geom = synth.make_world(0)
    m = synth.make_map(world_geom, 4000) if rank == 0 else None
    map_pts = distributed.broadcast_map(m, 4000, device="cpu").numpy()
    n_total, n_beams = 5 * world, 181      # five candidates per rank
    lo, hi = distributed.shard_range(n_total, rank, world)
    wl = synth.make_workload(hi - lo, 4000, seed=0, n_beams=n_beams, pose_seed_offset=rank, world=world_geom, map_points=np.zeros((0, 4), np.float32))
    sp = po.slice_params(canvas_cols=n_beams)

    def align_alone(clouds, x0):
        return np.stack([po.align(po.aligner_params(8), [sp], [c], [map_pts], x)["pose"] for c, x in zip(clouds, x0)])
    mine = align_alone([wl.scan_points[wl.scan_offsets[i]:wl.scan_offsets[i + 1]] for i in range(hi - lo)], wl.x0)
    err = np.abs(mine - wl.x_true); assert err[:, :2].max() < 1e-3                        # the shard really was aligned
    chk = distributed.cross_rank_check(wl.scan_points, wl.scan_offsets, None, wl.x0, mine, n_beams, align_alone, n_check=4, device="cpu")
    if rank == 0:
        assert chk == (world, world, 4), chk
        # ... and a rank whose poses were tampered with is caught
    bad = mine.copy(); bad[0, 0] += 1e-6 if rank == 1 else 0.0
    chk2 = distributed.cross_rank_check(wl.scan_points, wl.scan_offsets, None, wl.x0, bad, n_beams, align_alone, n_check=4, device="cpu")
    if rank == 0:
        assert chk2 == (world - 1, world, 4), chk2
    # the sweep's consumer sees every candidate's pose, in candidate order
    pad = max(distributed.shard_range(n_total, r, world)[1] - distributed.shard_range(n_total, r, world)[0] for r in range(world))
    rows = np.zeros((pad, 3), np.float32); rows[: hi - lo] = mine
    allp = distributed.gather_results(rows, device="cpu")
    assert allp.shape == (pad * world, 3) and np.array_equal(allp[rank * pad: rank * pad + (hi - lo)], mine)
    dist.barrier(); dist.destroy_process_group()
    open(os.path.join(out_dir, f"aligned{rank}"), "w").write("ok")


def test_gloo_world_size_2_shards_align_and_cross_check(tmp_path):
    port = _free_port()
    mp.spawn(_align_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "aligned0").exists() and (tmp_path / "aligned1").exists()


def test_gloo_world_size_8_shards_align_and_cross_check(tmp_path):
    """The node the driver's scaling run uses has EIGHT ranks: the same flow -- map broadcast from rank 0, per-rank shards aligned, the cross-rank bit check with a
    tampered rank caught, the sweep's gather in candidate order -- at world size 8 (gloo on the CPU, five candidates per rank)."""
    port = _free_port()
    mp.spawn(_align_worker, args=(8, port, str(tmp_path)), nprocs=8, join=True)
    assert all((tmp_path / ("aligned%d" % r)).exists() for r in range(8))


def test_rank_affinity_masks_partition_the_allowed_cores():
    """bench.py pins every rank to its own slice of the cores the process may use (by LOCAL_RANK, before anything touches the GPU): disjoint slices that cover
    the allowed set when there are at least as many cores as ranks, the whole set for everybody otherwise."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    allowed = list(range(3, 67))
    masks = [bench.rank_affinity(allowed, r, 8) for r in range(8)]
    assert all(len(m) == 8 for m in masks) and sorted(c for m in masks for c in m) == allowed
    assert bench.rank_affinity([5, 9], 1, 8) == [5, 9] and bench.rank_affinity(allowed, 0, 1) == allowed
    masks = [bench.rank_affinity(list(range(10)), r, 3) for r in range(3)]
    assert sorted(c for m in masks for c in m) == list(range(10)) and all(len(m) >= 3 for m in masks)


def test_shard_by_work_partitions_exactly_and_balances():
    """distributed.shard_by_work: contiguous, exhaustive, deterministic; cumulative work per rank within one candidate's worth of the ideal."""
    rng = np.random.default_rng(3)
    for n in (1, 5, 1000, 65536):
        w = rng.integers(150, 320, size=n)
        for world in (1, 2, 3, 8):
            sh = distributed.shard_by_work(w, world)
            assert len(sh) == world and sh[0][0] == 0 and sh[-1][1] == n and all(a[1] == b[0] for a, b in zip(sh, sh[1:])) and all(lo <= hi for lo, hi in sh)
            assert sh == distributed.shard_by_work(w.copy(), world)
            if n >= 100 * world:
                tot = [int(w[lo:hi].sum()) for lo, hi in sh]
                assert max(tot) - w.sum() / world <= 2 * w.max()
    # heavy candidates at one end: equal counts would give one rank most of the work
    w = np.concatenate([np.full(500, 300), np.full(500, 100)])
    sh = distributed.shard_by_work(w, 2)
    by_count = [int(w[lo:hi].sum()) for lo, hi in (distributed.shard_range(1000, r, 2) for r in range(2))]
    by_work = [int(w[lo:hi].sum()) for lo, hi in sh]
    assert max(by_work) < max(by_count) and max(by_work) <= 100300
    # no usable estimate: the count partition
    assert distributed.shard_by_work(np.zeros(10), 3) == [distributed.shard_range(10, r, 3) for r in range(3)]
    assert distributed.shard_by_work([], 2) == [(0, 0), (0, 0)]


def test_bench_refuses_more_gpus_than_devices_without_a_launcher():
    """`python bench.py --gpus 8` with no launcher in front must never print a line that claims one GPU: it starts its own ranks -- or, where the box
    has fewer devices than ranks (this CPU container has none), exits non-zero before anything runs."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LSM2D_BENCH_BACKEND")}
    if torch.cuda.device_count() >= 8:
        return
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "1"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "n_gpus" not in r.stdout and "device(s) visible" in (r.stderr + r.stdout)
    # a launcher's world that disagrees with --gpus is refused as well (it used to pass when WORLD_SIZE was 1)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1"], env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "n_gpus" not in r.stdout
