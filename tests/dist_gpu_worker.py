"""Worker of tests/test_gpu_dist.py: launched by torch.distributed.run, every rank on cuda:0 over
gloo (a 1-GPU box): map broadcast -> each rank scans its contiguous pose block with the PRODUCT
(libscan_amd.so) -> chunked all-gather of the ranges, two steps in flight on two streams ->
rank r saves what it gathered, in global pose order."""
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
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo")
    g0 = maps.make_maze(512, cell=40, wall=3, p=0.45, seed=17, origin=(1.0, -2.0, 0.25)) if rank == 0 else None
    g = broadcast_map(g0, 0, dev)
    omap = range_libc.PyOMap(g, device=0)
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


if __name__ == "__main__":
    main()
