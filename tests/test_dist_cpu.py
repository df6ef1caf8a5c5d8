"""world_size-2 gloo tests of the pose-batch sharding on CPU: map broadcast,
contiguous pose blocks, chunk-overlapped all-gather and its global ordering.  The march itself
is stood in by a NumPy function of the pose (the GPU kernel is covered by the -m gpu tests)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT
from pyracecarsimulator_amd import maps
from pyracecarsimulator_amd.distributed import (BucketedIndexGather, ShardedScan, broadcast_map,
                                                chunk_bounds, shard_range)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_ranges(poses, B):
    j = np.arange(B, dtype=np.float32)[None, :]
    return (poses[:, 0:1] * 1000 + poses[:, 1:2] * 10 + poses[:, 2:3] + j * 1e-3).astype(np.float32).ravel()


def _worker(rank, world, port, n_total, B, chunks, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g0 = maps.make_maze(64, cell=16, wall=2, seed=3, origin=(1.0, -2.0, 0.25)) if rank == 0 else None
        g = broadcast_map(g0, 0)
        poses_all = maps.sample_free_poses(g, n_total, 5)
        lo, hi = shard_range(n_total, rank, world)
        mine = poses_all[lo:hi]
        scan = ShardedScan(hi - lo, B, "cpu", n_chunks=chunks)

        def compute(clo, chi, view, stream=0):
            view.copy_(torch.from_numpy(_fake_ranges(mine[clo:chi], B)))

        scan.step(compute)
        scan.finish()
        got = scan.global_order().numpy().copy()
        q.put((rank, g.occ.sum(), g.origin, g.resolution, got, len(scan.chunks)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("chunks", [1, 4])
def test_sharded_scan_world2_matches_single_process(chunks):
    world, n_total, B = 2, 24, 37
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, B, chunks, q))
             for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    g = maps.make_maze(64, cell=16, wall=2, seed=3, origin=(1.0, -2.0, 0.25))
    want = _fake_ranges(maps.sample_free_poses(g, n_total, 5), B)
    for rank, occ_sum, origin, res_, got, n_chunks in res:
        assert occ_sum == g.occ.sum() and origin == g.origin and res_ == g.resolution
        assert n_chunks == chunks
        assert np.array_equal(got, want), "rank %d: gathered ranges are not in global pose order" % rank


def test_chunk_bounds_and_shards():
    assert chunk_bounds(4096, 4) == [(0, 1024), (1024, 2048), (2048, 3072), (3072, 4096)]
    assert chunk_bounds(10, 4) == [(0, 5), (5, 10)]          # reduced until it divides
    assert chunk_bounds(7, 4) == [(0, 7)]
    assert chunk_bounds(3, 8) == [(0, 1), (1, 2), (2, 3)]
    assert [shard_range(10, r, 3) for r in range(3)] == [(0, 4), (4, 7), (7, 10)]


def _bucket_worker(rank, world, port, n_items, every, n_steps, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        bg = BucketedIndexGather(n_items, every, "cpu")
        seen = []
        for step in range(n_steps):
            v = bg.slot_view()
            v.copy_(torch.arange(n_items, dtype=torch.int32) + 1000 * step + 100000 * rank)
            bg.step_done()
            if (step + 1) % every == 0:                  # a full bucket went out: read it once complete
                bg.flush()
                seen.append(bg.latest().clone().numpy())
        bg.flush()
        if n_steps % every:
            seen.append(bg.latest().clone().numpy())
        q.put((rank, seen))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("every,n_steps", [(1, 5), (4, 8), (4, 10), (8, 3)])
def test_bucketed_index_gather_world2(every, n_steps):
    """bench.py's N>1 exchange: every step's crash indices reach every rank, `every` steps per
    collective, partly filled last bucket included."""
    world, n_items = 2, 6
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bucket_worker, args=(r, world, port, n_items, every, n_steps, q))
             for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, seen in res:
        got = np.concatenate(seen, axis=1)               # (world, n_steps, n_items)
        assert got.shape == (world, n_steps, n_items)
        for r in range(world):
            for step in range(n_steps):
                assert np.array_equal(got[r, step], np.arange(n_items) + 1000 * step + 100000 * r)


def _depth_worker(rank, world, port, n_total, B, depth, n_steps, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = maps.make_maze(64, cell=16, wall=2, seed=3)
        lo, hi = shard_range(n_total, rank, world)
        scan = ShardedScan(hi - lo, B, "cpu", n_chunks=2, depth=depth)
        got = []
        slots = []
        for k in range(n_steps):
            poses = maps.sample_free_poses(g, n_total, 100 + k)[lo:hi]

            def compute(clo, chi, view, stream=0, poses=poses):
                view.copy_(torch.from_numpy(_fake_ranges(poses[clo:chi], B)))

            slots.append(scan.step(compute))
            if len(slots) == depth:                     # read the oldest slot before it is reused
                scan.finish()
                for sl in slots:
                    got.append(scan.global_order(sl).numpy().copy())
                slots = []
        scan.finish()
        for sl in slots:
            got.append(scan.global_order(sl).numpy().copy())
        q.put((rank, got))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("depth,n_steps", [(2, 4), (3, 5)])
def test_sharded_scan_keeps_several_steps_in_flight(depth, n_steps):
    """bench.py's N>1 schedule: `depth` steps in flight, each on its own slot (own local and gathered
    buffers); a slot only waits for the gathers issued from it `depth` steps earlier.  Every step's
    gathered ranges equal the single-process scan of that step's poses, in global pose order."""
    world, n_total, B = 2, 20, 33
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_depth_worker, args=(r, world, port, n_total, B, depth, n_steps, q))
             for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    g = maps.make_maze(64, cell=16, wall=2, seed=3)
    for rank, got in res:
        assert len(got) == n_steps
        for k in range(n_steps):
            want = _fake_ranges(maps.sample_free_poses(g, n_total, 100 + k), B)
            assert np.array_equal(got[k], want), (rank, k)


def _mode_worker(rank, world, port, n_total, B, mode, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = maps.make_maze(64, cell=16, wall=2, seed=3)
        lo, hi = shard_range(n_total, rank, world)
        scan = ShardedScan(hi - lo, B, "cpu", n_chunks=2, depth=2, mode=mode, root=1, max_range_m=15.0)
        out = []
        for k in range(3):
            poses = maps.sample_free_poses(g, n_total, 200 + k)[lo:hi]

            def compute(clo, chi, view, stream=0, poses=poses, k=k):
                # values in [0, 15] m like real ranges (plus one below 0 and one above the cap)
                v = np.abs(_fake_ranges(poses[clo:chi], B)) % 15.0
                v[0], v[-1] = -0.25, 15.5
                view.copy_(torch.from_numpy(v.astype(np.float32)))

            sl = scan.step(compute)
            scan.finish()
            out.append(None if sl.gathered is None else scan.global_order(sl).numpy().copy())
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["root", "ranges_u16"])
def test_sharded_scan_exchange_modes_world2(mode):
    """--gather root: only the consumer rank receives (bit-exact); --gather ranges_u16: every rank
    receives 16-bit ranges, within max/131070 of the float32 values clamped to [0, max]."""
    world, n_total, B = 2, 12, 40
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_mode_worker, args=(r, world, port, n_total, B, mode, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    g = maps.make_maze(64, cell=16, wall=2, seed=3)
    for k in range(3):
        poses_all = maps.sample_free_poses(g, n_total, 200 + k)
        want = []
        for r in range(world):
            lo, hi = shard_range(n_total, r, world)
            for clo, chi in ((0, (hi - lo) // 2), ((hi - lo) // 2, hi - lo)):
                v = np.abs(_fake_ranges(poses_all[lo:hi][clo:chi], B)) % 15.0
                v[0], v[-1] = -0.25, 15.5
                want.append(v.astype(np.float32))
        want = np.concatenate(want)
        if mode == "root":
            assert res[0][k] is None                      # rank 0 is not the consumer: holds nothing
            assert np.array_equal(res[1][k], want)
        else:
            for r in range(world):
                got = res[r][k]
                assert np.abs(got - np.clip(want, 0.0, 15.0)).max() <= 15.0 / 131070 * 1.01 + 2e-6
                assert got.min() == 0.0 and got.max() == np.float32(15.0)


def _fake_reduced(poses, mode, group):
    """Stand-ins for the two reductions of a scanned block: one int32 per roll-out of `group` poses /
    one float32 per pose, functions of the poses only (so the global answer is known on one process)."""
    if mode == "crash":
        v = (poses[:, 0] * 100 + poses[:, 1] * 7).astype(np.int64).reshape(-1, group)
        return (v.sum(axis=1) % 1000 - 500).astype(np.int32)
    return (poses[:, 0] * 0.25 - poses[:, 2]).astype(np.float32)


def _reduced_worker(rank, world, port, n_total, B, mode, group, depth, every, n_steps, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = maps.make_maze(64, cell=16, wall=2, seed=3)
        lo, hi = shard_range(n_total, rank, world)
        n_items = (hi - lo) // group if mode == "crash" else hi - lo
        scan = ShardedScan(hi - lo, B, "cpu", depth=depth, mode=mode, n_items=n_items, every=every)
        assert scan.reduced and not scan.gather and scan.exchange and len(scan.chunks) == 1
        per_slot = [[] for _ in range(depth)]             # steps a slot has taken since its last read-out
        got = {}
        for k in range(n_steps):
            poses = maps.sample_free_poses(g, n_total, 300 + k)[lo:hi]

            def compute(clo, chi, view, stream, res, poses=poses):
                assert (clo, chi) == (0, hi - lo) and view.numel() == (hi - lo) * B
                view.copy_(torch.from_numpy(_fake_ranges(poses, B)))
                res.copy_(torch.from_numpy(_fake_reduced(poses, mode, group)))

            sl = scan.step(compute)
            per_slot[(k % depth)].append(k)
            if k % 3 == 2 or k == n_steps - 1:            # read out at uneven points: partly filled buckets too
                scan.finish()
                for s_i, slot in enumerate(scan.slots):
                    steps = per_slot[s_i]
                    if not steps:
                        continue
                    r = scan.results(slot)
                    # the LAST exchanged bucket of the slot holds its most recent len(...) % every (or every) steps
                    tail = steps[-r.shape[1]:]
                    for j, step_no in enumerate(tail):
                        got[step_no] = r[:, j, :].reshape(-1).numpy().copy()
                    per_slot[s_i] = []
        q.put((rank, got))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode,depth,every,n_steps", [("crash", 1, 1, 3), ("crash", 2, 2, 7), ("steer", 3, 4, 8),
                                                        ("steer", 2, 1, 5)])
def test_sharded_scan_reduced_modes_world2(mode, depth, every, n_steps):
    """--gather crash / steer: the ranges stay with the rank that computed them; the per-roll-out crash
    index / per-pose steering angle of every step reaches every rank in GLOBAL order, bucketed per slot."""
    world, n_total, B, group = 2, 24, 19, 4
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_reduced_worker, args=(r, world, port, n_total, B, mode, group, depth, every,
                                                        n_steps, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    g = maps.make_maze(64, cell=16, wall=2, seed=3)
    for rank in range(world):
        got = res[rank]
        assert got, "rank %d read nothing" % rank
        for k, v in got.items():
            want = _fake_reduced(maps.sample_free_poses(g, n_total, 300 + k), mode, group)
            assert v.dtype == want.dtype and np.array_equal(v, want), (rank, k)
    # every step whose bucket was still the slot's latest at a read-out point was seen; with every == 1 and
    # read-outs every third step that is at least the steps right before each read-out
    assert (n_steps - 1) in res[0] and (n_steps - 1) in res[1]


def test_scaling_model_arithmetic():
    """bench.py's `scaling_model` (distributed.scaling_model) on cfg2's numbers: 4096 x 1081 rays per GPU, 167 Grays/s
    per GPU marching, the fused crash test at 96 % and scan + FollowGap at 74 % of that: the literal all-gather of ranges
    is xGMI-bound below 1x, half the bytes double it, the reduced modes scale with the march."""
    from pyracecarsimulator_amd.distributed import exchange_bytes, scaling_model
    rays, poses, groups = 4096 * 1081, 4096, 32
    assert exchange_bytes("ranges", 8, rays, poses, groups) == (8 * 4 * rays, 7 * 4 * rays)
    assert exchange_bytes("ranges_u16", 8, rays, poses, groups)[1] == 7 * 2 * rays
    assert exchange_bytes("crash", 8, rays, poses, groups) == (8 * 4 * 32, 7 * 4 * 32)
    assert exchange_bytes("steer", 2, rays, poses, groups) == (2 * 4 * 4096, 4 * 4096)
    assert exchange_bytes("none", 8, rays, poses, groups) == (0, 0)
    m = scaling_model(167000.0, {"crash": 160400.0, "steer": 124000.0}, rays, poses, groups)
    assert set(m) == {"ranges", "ranges_u16", "root", "crash", "steer", "none"}
    r = m["ranges"]
    assert r["bound"] == "xgmi" and r["ingress_bytes_per_gpu_per_step_at_8"] == 7 * 4 * rays
    assert r["xgmi_floor_ms"] == pytest.approx(7 * 4 * rays / (7 * 76.5e9) * 1e3, rel=1e-3)          # 0.2315 ms
    assert r["modelled_speedup_8gpu"] == pytest.approx(8 * (rays / 0.2315e-3 / 1e6) / 167000.0, rel=2e-2)   # ~0.92
    assert m["ranges_u16"]["modelled_speedup_8gpu"] == pytest.approx(2 * r["modelled_speedup_8gpu"], rel=2e-2)
    assert m["root"]["modelled_speedup_8gpu"] == r["modelled_speedup_8gpu"]
    assert m["crash"]["bound"] == "march" and m["crash"]["modelled_speedup_8gpu"] == pytest.approx(8 * 0.9605, rel=1e-2)
    assert m["steer"]["bound"] == "march" and m["steer"]["modelled_speedup_8gpu"] == pytest.approx(8 * 0.7425, rel=1e-2)
    assert m["none"]["modelled_speedup_8gpu"] == 8.0
    # a slow enough march is not xGMI-bound even for the ranges (the 1 Grays/s BASELINE.md was written for)
    assert scaling_model(1000.0, {}, rays, poses, groups)["ranges"]["bound"] == "march"


def test_rank_blocks_of_a_seeded_batch_and_the_baseline_batch_layout():
    """A rank generates only its own block, and the blocks tile the batch one GPU would draw;
    cfg4 / cfg5 shard BASELINE.json's GLOBAL batch, the other configs fix the poses per GPU."""
    from pyracecarsimulator_amd import workloads as W
    w = W.cfg2()
    whole = W.make_poses(w, n_poses=1000)
    parts = [W.rank_poses(w, 1000, r, 8) for r in range(8)]
    assert np.array_equal(np.concatenate(parts), whole)
    assert not np.array_equal(W.rank_poses(w, 1000, 0, 8, seed=w.pose_seed + 1), parts[0])
    assert W.batch_layout(W.cfg2(), 8) == (4096, 32768, "weak")
    assert W.batch_layout(W.cfg3(), 4) == (65536, 262144, "weak")
    assert W.batch_layout(W.cfg4(), 8) == (131072, 1 << 20, "strong")
    assert W.batch_layout(W.cfg5(), 8) == (32768, 262144, "strong")
    assert W.batch_layout(W.cfg5(), 8, poses_per_gpu=256) == (256, 2048, "weak")
    assert W.batch_layout(W.cfg4(), 1) == (1 << 20, 1 << 20, "weak")
