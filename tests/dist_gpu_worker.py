"""Worker of tests/test_gpu_dist.py: launched by torch.distributed.run.  With at least WORLD_SIZE visible
devices every rank takes ITS OWN device (LOCAL_RANK) and the process group is RCCL ("nccl") — the real
multi-GPU path; on a box with fewer devices (the 1-GPU gpurun box) every rank uses cuda:0 over gloo.
Map broadcast -> each rank scans its contiguous pose block with the PRODUCT (libscan_amd.so) -> chunked
all-gather of the ranges, two steps in flight on two streams -> rank r saves what it gathered, in global
pose order, and a rank<r>_env.json saying which backend / device it ran on."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out_dir, n_total, B = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    mode = sys.argv[4] if len(sys.argv) > 4 else "ranges"
    import torch
    import torch.distributed as dist
    from pyracecarsimulator_amd import maps, range_libc
    from pyracecarsimulator_amd.distributed import ShardedScan, broadcast_map, shard_range
    from pyracecarsimulator_amd.pipeline import concurrent_streams

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    # (device_count() does not initialise the GPU; one device per rank whenever the box has them)
    one_per_rank = torch.cuda.device_count() >= world and world > 1
    di = int(os.environ.get("LOCAL_RANK", "0")) if one_per_rank else 0
    torch.cuda.set_device(di)
    dev = torch.device("cuda", di)
    if one_per_rank:
        dist.init_process_group("nccl", device_id=dev)
    else:
        dist.init_process_group("gloo")
    assert dist.get_world_size() == world and dist.get_rank() == rank
    import json
    with open(os.path.join(out_dir, "rank%d_env.json" % rank), "w") as fh:
        json.dump({"backend": dist.get_backend(), "device": di, "world": dist.get_world_size(),
                   "visible_devices": torch.cuda.device_count()}, fh)
    g0 = maps.make_maze(512, cell=40, wall=3, p=0.45, seed=17, origin=(1.0, -2.0, 0.25)) if rank == 0 else None
    g = broadcast_map(g0, 0, dev)
    omap = range_libc.PyOMap(g, device=di)
    meth = range_libc.PyRayMarchingGPU(omap, 300)
    meth.set_noise(0.02, 99, 0)                              # noise keyed by the GLOBAL ray id
    fov = 4.71
    steps = []
    for k in range(2):
        poses_all = maps.sample_free_poses(g, n_total, 5 + k)
        lo, hi = shard_range(n_total, rank, world)
        steps.append((lo, torch.from_numpy(np.ascontiguousarray(poses_all[lo:hi])).to(dev)))
    lo, hi = shard_range(n_total, rank, world)
    streams = concurrent_streams(2)
    if mode in ("crash", "steer"):
        reduced_modes(out_dir, mode, rank, world, dev, meth, steps, lo, hi, B, fov, streams)
        dist.barrier()
        dist.destroy_process_group()
        return
    scan = ShardedScan(hi - lo, B, dev, n_chunks=3, gather=True, streams=streams if len(streams) == 2 else None,
                       depth=2, mode=mode, root=world - 1, max_range_m=300 * g.resolution)
    slots = []
    for k in range(2):
        lo_k, d_p = steps[k]

        def compute(clo, chi, view, sptr, d_p=d_p, lo_k=lo_k):
            meth.set_noise(0.02, 99, (lo_k + clo) * B)
            meth.calc_range_fan_device(d_p.data_ptr() + clo * 12, chi - clo, fov, B, view.data_ptr(), stream=sptr)

        slots.append(scan.step(compute))
    scan.finish()
    torch.cuda.synchronize()
    for k, sl in enumerate(slots):
        if sl.gathered is not None:                          # ('root': only the consumer rank holds the batch)
            np.save(os.path.join(out_dir, "rank%d_step%d.npy" % (rank, k)), scan.global_order(sl).cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


def reduced_modes(out_dir, mode, rank, world, dev, meth, steps, lo, hi, B, fov, streams):
    """--gather crash / steer with the product: the bound (prepared C call) path of ShardedScan, two slots on two
    streams, buckets of 2 steps; three steps so that the last bucket of slot 0 is full and slot 1's is partly
    filled.  Every rank saves what it gathered per step (global order)."""
    import torch
    from pyracecarsimulator_amd import racecar as RC
    from pyracecarsimulator_amd.distributed import ShardedScan
    from pyracecarsimulator_amd.followgap import PyFollowGap
    GROUP = 20
    n = hi - lo
    meth.set_noise(0.0, 0, 0)
    meth.set_option("slots", 2)                              # the pipelined kernel shape bench.py runs
    meth.set_option("grid_mult", 3)
    edge = RC.edge_distances(B, -fov / 2.0, fov / B, 0.275, RC.DEFAULT_CAR["width"], RC.DEFAULT_CAR["wb"])
    d_edge = torch.from_numpy(edge).to(dev)
    fg = PyFollowGap(10, 15.0, RC.DEFAULT_CAR["max_steer_ang"], 0.004, device=dev.index)
    sc = ShardedScan(n, B, dev, n_chunks=1, gather=True, streams=streams if len(streams) == 2 else None, depth=2,
                     mode=mode, n_items=n // GROUP if mode == "crash" else n, every=2)
    ptrs = [steps[0][1].data_ptr(), steps[1][1].data_ptr()]          # slot k scans batch k
    if mode == "crash":
        sc.bind_crash(meth, ptrs, fov, GROUP, d_edge.data_ptr(), 0.001)
    else:
        sc.bind_steer(meth, fg, ptrs, fov)
    for _ in range(5):                                       # slot 0: steps 0, 2, 4; slot 1: steps 1, 3
        sc.step()
    sc.finish()
    torch.cuda.synchronize()
    for k, sl in enumerate(sc.slots):
        r = sc.results(sl)                                   # (world, filled, n_items): slot 0's last bucket holds
        assert r.shape[0] == world and r.shape[1] == (1 if k == 0 else 2), r.shape     # step 4; slot 1's steps 1, 3
        for j in range(r.shape[1]):
            np.save(os.path.join(out_dir, "rank%d_slot%d_row%d.npy" % (rank, k, j)), r[:, j, :].reshape(-1).cpu().numpy())


if __name__ == "__main__":
    main()
