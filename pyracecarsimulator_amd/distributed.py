"""Pose-batch sharding of the scan path over the GPUs of one node.

One process per GPU (``torch.distributed``; backend "nccl" is RCCL over xGMI on ROCm, "gloo"
in the CPU tests).  Rays are independent, so the only exchanges are the two BASELINE.json's
north_star names:

* once per map: broadcast of the occupancy bytes from rank 0 (each rank then builds its own
  distance transform / tables on its GPU);
* per batch: all-gather of the ranges, issued per CHUNK of the local pose block so that the
  gather of chunk k runs on RCCL's stream while the march kernel of chunk k+1 runs on the
  compute stream (xGMI is point-to-point: a ring all-gather is per-link bound, and for
  4 B/ray it costs more than the kernel, so overlap is what matters).  Exchange modes
  (``ShardedScan(mode=...)``): ``"ranges"`` the float32 all-gather north_star names;
  ``"ranges_u16"`` the same collective on 16-bit fixed-point ranges (half the bytes, LOSSY:
  <= 0.11 mm at 15 m, opt-in and labelled); ``"root"`` a gather to ONE consumer rank (the
  reference's consumer is a single MCTS process: 7/8 of the GPUs then receive nothing);
* the REDUCED exchanges — what the reference's consumers of a scanned batch actually read:
  ``"crash"`` the fused per-roll-out crash test (``Car::isCrashed`` over ``scanMany``'s output,
  scripts/racecar_simulator_v2.py:146-167, consumed at scripts/mcts.py:237-245: ONE int32 per
  roll-out) and ``"steer"`` Follow-the-Gap on every scan (``fg.eval(lidar)``, scripts/mcts.py:262-267:
  ONE float32 per pose).  The ranges stay on the GPU that computed them; 4 B per roll-out / per pose
  are all-gathered in buckets of several steps (``BucketedIndexGather``), per slot, on the slot's
  stream — the exchange that scales over xGMI where 4 B per RAY cannot.

The reference has no collective anywhere (single process, single GPU: SURVEY §2.1).
"""
from __future__ import annotations

import numpy as np

from .workloads import shard_range


def broadcast_map(gmap_or_none, src=0, device=None):
    """Rank ``src`` passes a ``maps.GridMap``; every rank returns an equal GridMap.
    Metadata travels as a small float64 tensor, the grid as uint8."""
    import torch
    import torch.distributed as dist
    from .maps import GridMap

    rank = dist.get_rank()
    dev = device if device is not None else "cpu"
    meta = torch.zeros(6, dtype=torch.float64, device=dev)
    if rank == src:
        g = gmap_or_none
        meta = torch.tensor([g.rows, g.cols, g.resolution, *g.origin], dtype=torch.float64,
                            device=dev)
    dist.broadcast(meta, src)
    rows, cols = int(meta[0].item()), int(meta[1].item())
    if rank == src:
        occ = torch.from_numpy(np.ascontiguousarray(gmap_or_none.occ, dtype=np.uint8)).to(dev)
    else:
        occ = torch.empty((rows, cols), dtype=torch.uint8, device=dev)
    dist.broadcast(occ, src)
    name = gmap_or_none.name if rank == src else "map"
    return GridMap(occ.cpu().numpy(), float(meta[2].item()),
                   (float(meta[3].item()), float(meta[4].item()), float(meta[5].item())), name)


#: xGMI of a fully connected 8-GPU MI355X node: one link per peer, ~153 GB/s per link both directions together
XGMI_LINKS = 7
XGMI_LINK_GBS = 153.0 / 2

EXCHANGE_MODES = ("ranges", "ranges_u16", "root", "crash", "steer", "none")


def exchange_bytes(mode: str, world: int, rays_per_gpu: int, poses_per_gpu: int, groups_per_gpu: int):
    """(bytes all GPUs send per step, bytes the busiest GPU RECEIVES per step) of an exchange mode on ``world``
    GPUs: 4 B per ray (``ranges``, ``root``: into the one consumer), 2 B per ray (``ranges_u16``), 4 B per roll-out
    (``crash``), 4 B per pose (``steer``)."""
    per = {"ranges": 4 * rays_per_gpu, "ranges_u16": 2 * rays_per_gpu, "root": 4 * rays_per_gpu,
           "crash": 4 * groups_per_gpu, "steer": 4 * poses_per_gpu, "none": 0}[mode]
    return per * world, per * (world - 1)


def scaling_model(march_mrays_per_gpu: float, local_mrays_per_gpu: dict, rays_per_gpu: int, poses_per_gpu: int,
                  groups_per_gpu: int, target_world: int = 8, links: int = XGMI_LINKS,
                  link_gbs: float = XGMI_LINK_GBS):
    """What each exchange mode can reach on ``target_world`` GPUs of one node, from rates measured on the GPUs at
    hand: a GPU computes at ``local_mrays_per_gpu[mode]`` (the per-GPU rate of the mode's LOCAL work; modes that only
    move ranges overlap their exchange with the marches and compute at the march rate) unless the xGMI ingress of
    the mode's bytes — into the busiest GPU, ``links`` links at ``link_gbs`` GB/s per direction — is slower; the
    speed-up is against ONE GPU marching without exchange.  ``*_per_gpu`` are the per-GPU batch AT ``target_world``
    GPUs.  Pure arithmetic (unit-tested on the CPU box); bench.py prints it on every N>1 line."""
    peak = links * link_gbs * 1e9
    out = {}
    for md in EXCHANGE_MODES:
        ingress = exchange_bytes(md, target_world, rays_per_gpu, poses_per_gpu, groups_per_gpu)[1]
        floor_s = ingress / peak
        local = float(local_mrays_per_gpu.get(md, march_mrays_per_gpu))
        xg = (rays_per_gpu / floor_s / 1e6) if floor_s > 0 else float("inf")
        out[md] = {"ingress_bytes_per_gpu_per_step_at_%d" % target_world: int(ingress),
                   "xgmi_floor_ms": round(floor_s * 1e3, 5), "per_gpu_local_mrays_s": round(local, 1),
                   "bound": "xgmi" if xg < local else "march",
                   "modelled_speedup_%dgpu" % target_world: round(target_world * min(local, xg) / march_mrays_per_gpu, 2)}
    return out


def chunk_bounds(n_local: int, n_chunks: int):
    """Split a local block of ``n_local`` poses into <= n_chunks equal chunks (the last rank-
    uniform size is required by all_gather_into_tensor, so n_local must divide evenly or the
    chunk count is reduced until it does)."""
    c = max(1, min(n_chunks, n_local))
    while n_local % c:
        c -= 1
    step = n_local // c
    return [(i * step, (i + 1) * step) for i in range(c)]


class _Slot:
    __slots__ = ("local", "gathered", "stream", "handles", "views", "dsts", "calls", "sptr", "local_q",
                 "gathered_q", "views_q", "dsts_q", "dst_lists", "decoded", "bucket", "rcall")


#: exchange modes whose payload is a per-roll-out / per-pose result instead of the ranges
REDUCED_MODES = ("crash", "steer")


class ShardedScan:
    """Scan a global pose batch with every rank taking a contiguous block.

    ``compute(lo, hi, out_view, stream)`` must enqueue the scan of local poses [lo, hi) into
    ``out_view`` (a (hi-lo)*num_rays float32 tensor) on ``stream`` (a raw stream pointer, 0 on
    CPU) — on the GPU that is ``method.calc_range_fan_device``; the CPU tests pass a NumPy stand-in.

    ``depth`` consecutive steps are kept in flight: step k runs on slot ``k % depth`` (own output
    buffers, own stream when ``streams`` are given), and only waits for the gathers that used
    that slot ``depth`` steps earlier, so the all-gathers of step k run on RCCL's stream while
    the marches of steps k+1 .. k+depth-1 run on theirs.  ``finish()`` waits for everything
    (call it before reading results or stopping a clock).

    Reduced modes (``mode="crash"`` / ``"steer"``): a step scans the local block and reduces it on the
    same stream — ``crash``: ``rl_check_collision_groups_device`` (first crashed pose of every roll-out
    of ``group`` poses, int32), ``steer``: ``rl_followgap_eval_device`` (one float32 steering angle per
    pose) — into the slot's bucket; ``n_items`` results per step (roll-outs / poses of the local
    block), ``every`` steps per all-gather.  ``compute(lo, hi, out_view, stream, result_view)`` gets
    the bucket row to fill as a fifth argument.  ``results(slot)`` returns the slot's last exchanged
    bucket as (world, steps, n_items); row ``[:, s, :].reshape(-1)`` is step s in GLOBAL order.
    """

    def __init__(self, n_local: int, num_rays: int, device, n_chunks: int = 4, gather=True,
                 depth: int = 1, streams=None, gather_single_rank: bool = False, mode: str = "ranges",
                 root: int = 0, max_range_m: float = 15.0, n_items: int = 0, every: int = 8):
        import torch
        import torch.distributed as dist
        self.torch = torch
        self.dist = dist
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.n_local, self.num_rays = n_local, num_rays
        if mode not in ("ranges", "ranges_u16", "root") + REDUCED_MODES:
            raise ValueError("mode must be 'ranges', 'ranges_u16', 'root', 'crash' or 'steer'")
        self.mode, self.root, self.max_range_m = mode, int(root), float(max_range_m)
        self.device = device
        self.reduced = mode in REDUCED_MODES
        # (gather_single_rank: run the collectives even in a one-rank group — exercises the RCCL code
        #  path, stream ordering included, on a box with one GPU)
        exchange = gather and (self.world > 1 or (gather_single_rank and dist.is_initialized()))
        self.gather = exchange and not self.reduced           # the RANGES are exchanged
        self.exchange = exchange                              # anything is exchanged
        self.chunks = chunk_bounds(n_local, n_chunks if self.gather else 1)
        if self.reduced:
            self.n_items = int(n_items) if n_items else n_local
            self.every = max(1, int(every))
        if streams is not None:
            depth = max(depth, len(streams))
        self.depth = max(1, int(depth))
        self.slots = []
        u16 = self.gather and mode == "ranges_u16"
        holds_all = self.gather and (mode != "root" or self.rank == self.root)
        self._dev_index = (device.index if getattr(device, "index", None) is not None else 0) \
            if getattr(device, "type", "cpu") == "cuda" else -1
        for k in range(self.depth):
            sl = _Slot()
            sl.local = torch.empty(n_local * num_rays, dtype=torch.float32, device=device)
            # chunk-major gather buffer: [chunk][rank][poses_in_chunk * num_rays]
            # (ranges_u16: the collective moves int16 bit patterns; `gathered` holds the decoded float32)
            sl.gathered = (torch.empty(self.world * n_local * num_rays, dtype=torch.float32,
                                       device=device) if holds_all else None)
            sl.local_q = torch.empty(n_local * num_rays, dtype=torch.int16, device=device) if u16 else None
            sl.gathered_q = (torch.empty(self.world * n_local * num_rays, dtype=torch.int16, device=device)
                             if u16 else None)
            sl.stream = streams[k % len(streams)] if streams else None
            sl.handles = []
            sl.decoded = True
            # per chunk: the slice of `local` a march fills and, when gathering, where its all-gather
            # lands — built once, a step only walks them
            sl.views = [sl.local[lo * num_rays:hi * num_rays] for lo, hi in self.chunks]

            def _dsts(buf):
                return [buf[ci * self.world * (hi - lo) * num_rays:(ci + 1) * self.world * (hi - lo) * num_rays]
                        for ci, (lo, hi) in enumerate(self.chunks)]
            sl.dsts = _dsts(sl.gathered) if holds_all else None
            sl.views_q = [sl.local_q[lo * num_rays:hi * num_rays] for lo, hi in self.chunks] if u16 else None
            sl.dsts_q = _dsts(sl.gathered_q) if u16 else None
            # "root": the consumer rank's destination split per source rank (dist.gather wants a list)
            sl.dst_lists = ([list(d.chunk(self.world)) for d in sl.dsts]
                            if (self.gather and mode == "root" and self.rank == self.root) else None)
            sl.calls = None
            sl.rcall = None
            sl.sptr = sl.stream.cuda_stream if sl.stream is not None else 0
            # reduced modes: the slot's own double-buffered bucket of per-step results, exchanged on its stream
            sl.bucket = (BucketedIndexGather(self.n_items, self.every, device,
                                             dtype=torch.int32 if mode == "crash" else torch.float32,
                                             stream=sl.stream, exchange=exchange)
                         if self.reduced else None)
            self.slots.append(sl)
        self.tick = 0
        self.last = self.slots[0]
        self._bound = None
        self._noise = None

    def bind(self, method, d_poses_ptr, fov: float, noise=None):
        """Fix the scan a step performs — ``method.calc_range_fan_device`` of the local poses at
        device address ``d_poses_ptr`` into the slot's buffer — so that ``step()`` without a
        ``compute`` callback is one prepared C call per chunk (no per-step tensor slicing, pointer
        look-ups or ctypes argument objects: a step costs the host ~7 us instead of ~10, which is what
        a short burst of steps sees between its first and its last launch).
        ``d_poses_ptr``: one address, or one PER SLOT (``depth`` of them): every step in flight then
        scans its own pose batch, as consecutive MCTS roll-out batches do.
        ``noise``: (std, seed, first global ray id of this rank's block) when the method adds range noise and the
        block is scanned in MORE THAN ONE chunk: the noise is keyed by the global ray id, so every chunk call must
        start at its own ray offset (``rl_set_noise`` in front of it) to reproduce the unchunked scan."""
        from . import _lib
        raw = _lib.raw("rl_calc_range_fan_device")
        self._noise = None
        if noise is not None and noise[0] > 0 and len(self.chunks) > 1:
            self._noise = (_lib.raw("rl_set_noise"), float(noise[0]), int(noise[1]), int(noise[2]))
        B = self.num_rays
        ptrs = list(d_poses_ptr) if isinstance(d_poses_ptr, (list, tuple)) else [d_poses_ptr] * self.depth
        if len(ptrs) != self.depth:
            raise ValueError("one pose address per slot (%d) expected, got %d" % (self.depth, len(ptrs)))
        for sl, pp in zip(self.slots, ptrs):
            base = sl.local.data_ptr()
            sl.calls = [(pp + lo * 12, hi - lo, base + lo * B * 4) for lo, hi in self.chunks]
        self._bound = (raw, method._h, float(fov), _lib.check)

    def bind_crash(self, method, d_poses_ptr, fov: float, group: int, d_edge_ptr: int, crash_thresh: float,
                   keep_ranges: bool = True):
        """mode "crash": a step is ``rl_check_collision_groups_device`` of the local poses (roll-outs of
        ``group`` consecutive poses, ``n_items`` = n_local / group of them) straight into the slot's
        bucket row.  ``keep_ranges``: the ranges are also stored to the slot's local buffer (False: the
        ray-marching methods then never store a range at all)."""
        from . import _lib
        if self.mode != "crash":
            raise ValueError("bind_crash needs mode='crash'")
        if self.n_local % int(group) or self.n_items != self.n_local // int(group):
            raise ValueError("n_items must be n_local / group")
        ptrs = list(d_poses_ptr) if isinstance(d_poses_ptr, (list, tuple)) else [d_poses_ptr] * self.depth
        if len(ptrs) != self.depth:
            raise ValueError("one pose address per slot (%d) expected, got %d" % (self.depth, len(ptrs)))
        raw = _lib.raw("rl_check_collision_groups_device")
        for sl, pp in zip(self.slots, ptrs):
            sl.rcall = (pp, sl.local.data_ptr() if keep_ranges else None)
        self._bound = ("crash", raw, method._h, float(fov), int(group), int(d_edge_ptr), float(crash_thresh), _lib.check)

    def bind_steer(self, method, followgap, d_poses_ptr, fov: float):
        """mode "steer": a step is ``rl_calc_range_fan_device`` of the local poses into the slot's local
        buffer + ``rl_followgap_eval_device`` of those scans, on the same stream, into the bucket row."""
        from . import _lib
        if self.mode != "steer":
            raise ValueError("bind_steer needs mode='steer'")
        if self.n_items != self.n_local:
            raise ValueError("n_items must be n_local (one steering angle per pose)")
        ptrs = list(d_poses_ptr) if isinstance(d_poses_ptr, (list, tuple)) else [d_poses_ptr] * self.depth
        if len(ptrs) != self.depth:
            raise ValueError("one pose address per slot (%d) expected, got %d" % (self.depth, len(ptrs)))
        for sl, pp in zip(self.slots, ptrs):
            sl.rcall = (pp, sl.local.data_ptr())
        # FollowGap reads the scan back on the same stream at once: keep the ranges in the L2 (plain stores; the
        # default non-temporal stores cost this mode 4.6 %, profiles/r04/nt_store_ab.txt)
        self._steer_method, self._steer_nt_store = method, int(method.get_info("nt_store"))
        method.set_option("nt_store", 0)       # (unbind() / close() restores the method's own setting)
        self._bound = ("steer", _lib.raw("rl_calc_range_fan_device"), method._h, float(fov),
                       _lib.raw("rl_followgap_eval_device"), followgap._h, _lib.check)

    def _chunk_noise(self, h, ci):
        """Key the next chunk call's noise at the chunk's own global ray id; after the LAST chunk of a step the
        method is back at the block's offset (std, seed, base) the caller bound it with — a later direct scan
        with the same method is keyed as if ShardedScan had never touched it."""
        raw_noise, std, seed, base = self._noise
        rc = raw_noise(h, std, seed, base + self.chunks[ci][0] * self.num_rays)
        if rc:
            from . import _lib
            _lib.check(rc)

    def _restore_noise(self, h):
        raw_noise, std, seed, base = self._noise
        rc = raw_noise(h, std, seed, base)
        if rc:
            from . import _lib
            _lib.check(rc)

    def unbind(self):
        """Give the bound method back as it was handed over: ``bind_steer`` switches its range stores to plain
        ones (``nt_store`` 0) for the FollowGap kernel that reads them next — a caller that goes on to plain scans
        with the same method wants the non-temporal stores back."""
        m = getattr(self, "_steer_method", None)
        if m is not None:
            m.set_option("nt_store", self._steer_nt_store)
            self._steer_method = None
        self._bound = None

    close = unbind

    # the first slot's buffers (depth 1: the only ones)
    @property
    def local(self):
        return self.last.local

    @property
    def gathered(self):
        return self.last.gathered

    def _on(self, sl):
        # the current stream only matters to the collectives (they order themselves against it); a
        # step without gathers hands the slot's stream to `compute` explicitly and skips the switch
        import contextlib
        return (self.torch.cuda.stream(sl.stream) if (sl.stream is not None and self.gather)
                else contextlib.nullcontext())

    def _step_reduced(self, sl, compute):
        b = sl.bucket
        if compute is not None:
            compute(0, self.n_local, sl.views[0], sl.sptr, b.slot_view())
        elif self._bound[0] == "crash":
            _, raw, h, fov, group, d_edge, thresh, check = self._bound
            pp, rp = sl.rcall
            rc = raw(h, pp, self.n_items, group, fov, self.num_rays, d_edge, thresh, b.slot_ptr(), rp, sl.sptr)
            if rc:
                check(rc)
        else:
            _, raw_fan, h, fov, raw_fg, g, check = self._bound
            pp, rp = sl.rcall
            rc = raw_fan(h, pp, self.n_local, fov, self.num_rays, rp, None, None, sl.sptr)
            if not rc:
                rc = raw_fg(g, rp, self.n_local, self.num_rays, b.slot_ptr(), sl.sptr)
            if rc:
                check(rc)
        b.step_done()
        self.last = sl
        return sl

    def step(self, compute=None):
        sl = self.slots[self.tick % self.depth]
        self.tick += 1
        if self.reduced:
            return self._step_reduced(sl, compute)
        if compute is None and not self.gather:      # bound scan, nothing to exchange: the lean path
            raw, h, fov, check = self._bound
            for ci, (pp, cnt, op) in enumerate(sl.calls):
                if self._noise is not None:
                    self._chunk_noise(h, ci)
                rc = raw(h, pp, cnt, fov, self.num_rays, op, None, None, sl.sptr)
                if rc:
                    check(rc)
            if self._noise is not None:
                self._restore_noise(h)
            self.last = sl
            return sl
        with self._on(sl):
            for h in sl.handles:          # gathers of the step that used this slot `depth` steps ago
                h.wait()
            sl.handles = []
            self._decode(sl)              # (ranges_u16: what that step gathered becomes float32 now)
            for ci, (lo, hi) in enumerate(self.chunks):
                view = sl.views[ci]
                if compute is None:
                    raw, hm, fov, check = self._bound
                    pp, cnt, op = sl.calls[ci]
                    if self._noise is not None:
                        self._chunk_noise(hm, ci)
                    rc = raw(hm, pp, cnt, fov, self.num_rays, op, None, None, sl.sptr)
                    if rc:
                        check(rc)
                else:
                    compute(lo, hi, view, sl.sptr)
                if self.gather:
                    sl.handles.append(self._exchange(sl, ci, view))
            if compute is None and self._noise is not None:
                self._restore_noise(self._bound[1])
        self.last = sl
        return sl

    def _exchange(self, sl, ci, view):
        """Issue the collective of chunk ``ci`` (asynchronous; ordered behind the march on the current
        stream) and return its work handle."""
        if self.mode == "ranges":
            return self.dist.all_gather_into_tensor(sl.dsts[ci], view, async_op=True)
        if self.mode == "root":
            return self.dist.gather(view, sl.dst_lists[ci] if self.rank == self.root else None, dst=self.root,
                                    async_op=True)
        # ranges_u16: encode on the slot's stream, exchange 2 B per ray; decoded lazily (global_order / finish)
        from . import _lib
        vq = sl.views_q[ci]
        if self._dev_index >= 0:
            _lib.check(_lib.lib().rl_ranges_to_u16_device(self._dev_index, view.data_ptr(), view.numel(),
                                                          self.max_range_m, vq.data_ptr(), sl.sptr or None))
        else:                                   # CPU tests (gloo): the same arithmetic in torch
            q = self.torch.round(view.clamp(0.0, self.max_range_m) * (65535.0 / self.max_range_m))
            vq.copy_((q.to(self.torch.int32) - ((q >= 32768).to(self.torch.int32) << 16)).to(self.torch.int16))
        sl.decoded = False
        # (neither RCCL nor gloo has a 16-bit integer type: the collective moves the same bytes as uint8)
        return self.dist.all_gather_into_tensor(sl.dsts_q[ci].view(self.torch.uint8), vq.view(self.torch.uint8),
                                                async_op=True)

    def _decode(self, sl):
        """ranges_u16: gathered 16-bit values -> float32 ``gathered`` (after the gathers of the slot)."""
        if self.mode != "ranges_u16" or sl.decoded or not self.gather:
            return
        if self._dev_index >= 0:
            from . import _lib
            _lib.check(_lib.lib().rl_ranges_from_u16_device(self._dev_index, sl.gathered_q.data_ptr(),
                                                            sl.gathered_q.numel(), self.max_range_m,
                                                            sl.gathered.data_ptr(), sl.sptr or None))
        else:
            q = sl.gathered_q.to(self.torch.int32) & 0xffff
            sl.gathered.copy_(q.to(self.torch.float32) * (self.max_range_m / 65535.0))
        sl.decoded = True

    def finish(self, end_events=None):
        """Every enqueued march and gather of every slot is ordered before what follows on the slot
        streams (device-side order; follow with a device synchronisation before reading on the host).
        ``end_events``: optional list of torch events, one per slot — event k is recorded on slot k's
        stream behind its last work (a caller timing a region takes the latest of them)."""
        for k, sl in enumerate(self.slots):
            if self.reduced:
                sl.bucket.flush()
            else:
                with self._on(sl):
                    for h in sl.handles:
                        h.wait()
                    sl.handles = []
                    self._decode(sl)
            if end_events is not None and sl.stream is not None:
                end_events[k].record(sl.stream)

    def results(self, slot=None):
        """Reduced modes: the slot's most recently exchanged bucket as a (world, steps, n_items) tensor
        (call after ``finish()``, which also makes the CURRENT stream wait for the slot streams: kernels
        enqueued on it afterwards see the exchanged bucket; a host read still needs a synchronisation);
        ``[:, s, :].reshape(-1)`` is step s of that bucket in global order."""
        if not self.reduced:
            raise RuntimeError("results(): only for the reduced modes %r" % (REDUCED_MODES,))
        return (slot or self.last).bucket.latest()

    def global_order(self, slot=None):
        """Gathered ranges re-ordered to global pose order: rank-major blocks, i.e. exactly what
        one GPU scanning the whole batch writes.  Returns a (world*n_local*num_rays,) tensor."""
        sl = slot or self.last
        if not self.gather:
            return sl.local
        if sl.gathered is None:
            raise RuntimeError("mode 'root': only rank %d holds the gathered ranges" % self.root)
        if not sl.decoded:
            raise RuntimeError("call finish() before reading ranges_u16 results")
        B, W = self.num_rays, self.world
        per = (self.chunks[0][1] - self.chunks[0][0]) * B
        g = sl.gathered.view(len(self.chunks), W, per)
        return g.permute(1, 0, 2).reshape(-1)


__all__ = ["broadcast_map", "chunk_bounds", "ShardedScan", "shard_range", "BucketedIndexGather", "REDUCED_MODES",
           "exchange_bytes", "scaling_model", "EXCHANGE_MODES", "XGMI_LINKS", "XGMI_LINK_GBS"]


class BucketedIndexGather:
    """Exchange of small per-step results (the int32 crash index of every roll-out, the float32
    steering angle of every pose) in buckets of ``every`` steps, double-buffered.

    ``slot_view()`` (a tensor view) / ``slot_ptr()`` (its device address) say where the current step
    must write its ``n_items`` values (a row of the bucket being filled); ``step_done()`` advances and,
    when a bucket is full, issues ONE async all-gather of the whole bucket — on the GPU it runs on
    RCCL's stream while the next bucket's marches run on the compute stream.  ``flush()`` exchanges a
    partly filled bucket and waits for everything; ``latest()`` returns the last completed gather as
    (world, steps_in_bucket, n_items).  A collective per 30-us step would cost more in host time and
    stream events than it moves.

    ``stream``: the torch stream the producing kernels run on (a slot stream of ``ShardedScan``): the
    collective is issued with that stream current, so it is ordered behind the kernels that filled the
    bucket, and a buffer is only refilled after the gather that read it (two buckets earlier) has been
    waited for on that stream.  ``exchange=False`` (or a one-rank world): the bucket is copied instead.
    """

    def __init__(self, n_items: int, every: int, device, dtype=None, stream=None, exchange=None):
        import torch
        import torch.distributed as dist
        self.torch = torch
        self.dist = dist
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.exchange = (self.world > 1) if exchange is None else (bool(exchange) and dist.is_initialized())
        self.n_items, self.every = int(n_items), max(1, int(every))
        dtype = dtype or torch.int32
        self.stream = stream
        self.local = [torch.zeros(self.every * self.n_items, dtype=dtype, device=device) for _ in range(2)]
        self.gathered = [torch.zeros(self.world * self.every * self.n_items, dtype=dtype, device=device)
                         for _ in range(2)]
        if getattr(self.local[0], "is_cuda", False):
            # the zero fills above run on the CURRENT stream; the kernels that write the rows and the collective's
            # copy-back run on other streams, unordered against it — a fill that is still queued (a GPU shared by
            # several processes: the 8-rank dry runs) would land AFTER the results and wipe them
            torch.cuda.current_stream(self.local[0].device).synchronize()
        row = self.n_items * self.local[0].element_size()
        self._ptrs = [[buf.data_ptr() + k * row for k in range(self.every)] for buf in self.local]
        self.pending = [None, None]
        self.tick = 0
        self._last = None

    def _on(self):
        import contextlib
        return self.torch.cuda.stream(self.stream) if self.stream is not None else contextlib.nullcontext()

    def _reuse(self, k):
        if self.pending[k] is not None:      # the gather issued two buckets ago read this buffer
            with self._on():
                self.pending[k].wait()
            self.pending[k] = None

    def slot_view(self):
        b, slot = divmod(self.tick, self.every)
        k = b & 1
        if slot == 0:
            self._reuse(k)
        return self.local[k][slot * self.n_items:(slot + 1) * self.n_items]

    def slot_ptr(self):
        """Device address of ``slot_view()`` (prepared C calls: no tensor slicing per step)."""
        b, slot = divmod(self.tick, self.every)
        k = b & 1
        if slot == 0:
            self._reuse(k)
        return self._ptrs[k][slot]

    def _issue(self, k, filled):
        with self._on():
            if self.exchange:
                self.pending[k] = self.dist.all_gather_into_tensor(self.gathered[k], self.local[k], async_op=True)
            else:
                self.gathered[k].copy_(self.local[k], non_blocking=True)
        self._last = (k, filled)

    def step_done(self):
        b, slot = divmod(self.tick, self.every)
        self.tick += 1
        if slot == self.every - 1:
            self._issue(b & 1, self.every)

    def flush(self):
        b, slot = divmod(self.tick, self.every)
        if slot:                                            # a partly filled bucket: exchange it too
            k = b & 1
            self._reuse(k)
            self._issue(k, slot)
            self.tick += self.every - slot
        for k in range(2):
            self._reuse(k)
        # the waits above are enqueued on the producing stream (a torch side stream is non-blocking with respect to
        # the default stream): order the CURRENT stream behind it, so that a reader enqueued there after flush()
        # sees the exchanged bucket.  (A host read — .cpu(), .item() on another stream — still synchronises itself.)
        if self.stream is not None and getattr(self.local[0], "is_cuda", False):
            self.torch.cuda.current_stream(self.local[0].device).wait_stream(self.stream)

    def latest(self):
        """(world, filled_steps, n_items) view of the most recently issued bucket.  Call after ``flush()``: it orders
        the current stream behind the exchange; reading the view from the host or from a third stream needs that
        stream's own synchronisation."""
        if self._last is None:
            return None
        k, filled = self._last
        return self.gathered[k].view(self.world, self.every, self.n_items)[:, :filled]
