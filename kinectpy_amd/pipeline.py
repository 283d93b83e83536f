"""Per-frame multi-sensor pipeline: the frame loop of the reference's DataProcessor
(preprocessing/data.py:35-61, 127-161) for one group of sensors held by one GPU:

    extract (depth -> masked, gated, compacted clouds)           a1-a4
    register each sub sensor onto the group's master (voxel 35 -> normals -> point-to-plane ICP)  a11-a14
    transform the sub clouds, fuse, filter_outliers                a17, a6-a8

and, across GPUs, the exchange of parallel.py.  Everything on the device goes through kinectpy_amd.ops.
"""
from collections import deque
from concurrent.futures import ThreadPoolExecutor
from dataclasses import dataclass
import contextlib
import os
from typing import List, Optional

import numpy as np
import torch

from . import ops, parallel


# kpx_frame_params.icp_mode; "fixed" (round 5): no registration inside the frame -- the reference registers on its first frame only
# (preprocessing/data.py:35-41) and every later frame reuses the transforms: the pipeline's init_transforms ARE the transforms
ICP_MODES = {"p2p": 0, "p2plane": 1, "fixed": 2}


@dataclass
class PipelineParams:
    reg_voxel: float = 35.0          # preprocessing/registration.py:35,69
    normals_nn: int = 40             # registration.py:24
    icp_max_dist: float = 100.0      # registration.py:75
    icp_mode: str = "p2plane"        # registration.py:83
    icp_max_iteration: int = 30      # Open3D default criteria
    filt_voxel: float = 10.0         # BASELINE.json config 3: voxel 1 cm on mm data
    filt_k: int = 20
    filt_ratio: float = 2.0
    gate: float = 750.0              # preprocessing/data.py:170-171
    poll_interval: int = 1


class SensorGroupPipeline:
    def __init__(self, xy_table, init_transforms: List[np.ndarray], params: Optional[PipelineParams] = None,
                 cloud_capacity: int = 0):
        self.p = params or PipelineParams()
        self.xy = ops._dev(xy_table, torch.float32).reshape(-1)
        self.init = [np.asarray(T, dtype=np.float64) for T in init_transforms]
        self.capacity = cloud_capacity
        self._xchg = None
        self.last = {}

    def step(self, depth: torch.Tensor, rgb: torch.Tensor):
        """depth (S, n_px) u16, rgb (S, n_px, 3) u8 (background zeroed), both resident on the device.
        Returns (points (K,3) f32, colours (K,3) f32, transforms (S,4,4) f64 sub->group master)."""
        p = self.p
        S = depth.shape[0]
        # both extractions are queued before the first count is read back; the person clouds are only needed after the
        # registration, so their counts are fetched then (no stall)
        fixed = p.icp_mode == "fixed"
        if not fixed:
            fp, _, _, fcnt = ops.depth_to_cloud(depth, self.xy, None, S, False, False, sync=False)             # registration input
        mp, mc, _, mcnt = ops.depth_to_cloud(depth, self.xy, rgb, S, True, True, gate=p.gate, sync=False)     # person clouds
        Ts = [np.eye(4)]
        stats = []
        downs = []
        if fixed:                                           # data.py:44-61 after the first frame: the stored transforms
            Ts += [np.asarray(T, dtype=np.float64) for T in self.init]
        else:
            fk = ops._count(fcnt)
            # -- registration: every sub onto the master, exactly execute_point_to_plane_registration
            downs = [d[0] for d in ops.voxel_downsample_batch([fp[i, :fk[i]] for i in range(S)], p.reg_voxel)]
            tn = ops.estimate_normals(downs[0], 2.0 * p.reg_voxel, p.normals_nn) if p.icp_mode == "p2plane" else None
        if S > 1 and not fixed:
            rs = ops.icp_batch(downs[1:], downs[0], p.icp_max_dist, self.init, p.icp_mode, tn, p.icp_max_iteration)
            for r in rs:
                Ts.append(r["transformation"])
                stats.append((r["iterations"], r["fitness"], r["inlier_rmse"]))
        # -- transform + fuse + filter
        mk = ops._count(mcnt)
        masked = [(mp[i, :mk[i]], mc[i, :mk[i]]) for i in range(S)]
        # pcd.transform(T_i) + np.vstack + voxel_down_sample (data.py:44-61) in one pass, on the fp64 values of the moved points
        vp, vc = ops.fuse_voxel_downsample([m[0] for m in masked], [m[1] for m in masked], Ts, p.filt_voxel)
        keep, _, _ = ops.sor(vp, p.filt_k, p.filt_ratio)
        out_p, out_c, _ = ops.select_by_index([vp, vc], keep, trusted=True)
        self.last = {"icp": stats, "n_down": [int(d.shape[0]) for d in downs], "n_masked": [int(m[0].shape[0]) for m in masked],
                     "n_fused": int(sum(mk)), "n_voxel": int(vp.shape[0]), "n_out": int(out_p.shape[0])}
        return out_p, out_c, np.stack(Ts)

    def exchange(self, out_p, out_c, Ts, to_global: np.ndarray):
        """Fuse across GPUs: clouds are moved into the global master frame with `to_global` (group master ->
        global master, from calibration) and all-gathered together with the composed transforms
        (parallel.CloudExchange: one collective per frame, message sized from the previous frame's counts)."""
        if self._xchg is None:
            self._xchg = parallel.CloudExchange(self.capacity)
        gp = ops.transform(out_p, to_global) if not np.allclose(to_global, np.eye(4)) else out_p
        comp = torch.as_tensor(np.stack([to_global @ T for T in Ts]))
        return self._xchg(gp, out_c, comp)


class SensorShardPipeline:
    """The north-star partition of the reference's DataProcessor across GPUs (BASELINE.json configs[3] / [4]):
    sensor g lives on GPU g (with fewer GPUs than sensors a rank owns a contiguous block of them, same code path).

    calibration (preprocessing/data.py:127-161 -- the master's cloud is the target of every sub device's registration):
        every rank   depth -> full cloud -> voxel_down_sample(35) of its own sensors
        rank 0       + normals of the master's down-sampled cloud, then ONE broadcast of (cloud, normals)
        every rank   execute_point_to_plane_registration of its own sub sensors onto the master cloud (kpx_icp_batch)
    per frame (data.py:35-61):
        every rank   depth + person mask -> masked, gated clouds of its sensors
                     ONE all-gather of those clouds (unmoved: exact float32 of int16 data), in sensor order (= np.vstack of
                     data.py:55-58), with their counts and 4x4 transforms in the header rows; pcd.transform(T_i) happens
                     inside the fused voxel grid, in fp64 (kpx_fuse_voxel_downsample)
        filter_outliers on the FUSED cloud (data.py:61), `fused_filter`:
          "rank0"    rank 0 filters alone (the others return None and go on with the next frame);
          "sharded"  every rank down-samples the fused cloud (cheap, identical everywhere), searches the k neighbours of its
                     own slab of the grid order only (kpx_sor_partial), ONE all-gather of the slabs' mean distances
                     (8 B per point), then the statistics and the keep list with the one-GPU kernels (kpx_sor_finish): the
                     result is on every rank and bit-identical to the single-GPU filter.

    step() = calibration + frame (the bench's unit: unproject + filter + ICP).  `ops_module` is the operator namespace
    (kinectpy_amd.ops; the CPU rehearsal of the multi-rank logic in tests/ injects its own)."""

    def __init__(self, xy_table, n_sensors: int, init_transforms: List[np.ndarray], params: Optional[PipelineParams] = None,
                 group=None, fused_filter: str = "sharded", cloud_capacity: int = 64 * 1024, ops_module=None, rank=None, world=None):
        self.ops = ops_module or ops
        self.p = params or PipelineParams()
        if self.p.icp_mode == "fixed":
            raise ValueError('icp_mode "fixed" (no registration in the frame) is a one-GPU form: SensorGroupPipeline / NativeFramePipeline')
        self.group = group
        self.world = parallel.world_size(group) if world is None else world
        self.rank = (torch.distributed.get_rank() if torch.distributed.is_initialized() else 0) if rank is None else rank
        self.n_sensors = int(n_sensors)
        if self.world > self.n_sensors:
            raise ValueError(f"{self.world} ranks for {self.n_sensors} sensors: a rank needs at least one sensor")
        self.sensors = parallel.shard_sensors(self.n_sensors, self.rank, self.world)     # contiguous: rank order = sensor order
        self.k_max = -(-self.n_sensors // self.world)                                  # transforms per rank in the exchange header
        if len(init_transforms) != self.n_sensors - 1:
            raise ValueError("init_transforms: one 4x4 per sub sensor (sensors 1 .. n-1)")
        self.init = [np.asarray(T, dtype=np.float64) for T in init_transforms]
        if fused_filter not in ("rank0", "sharded"):
            raise ValueError("fused_filter must be 'rank0' or 'sharded'")
        self.fused_filter = fused_filter
        self.xy = self.ops._dev(xy_table, torch.float32).reshape(-1)
        self._bcast = parallel.MasterBroadcast(cloud_capacity, group)
        self._xchg = parallel.SensorExchange(max(4096, cloud_capacity * self.k_max), self.k_max, group)     # the same on every rank
        self.transforms = None            # (len(self.sensors), 4, 4) sub -> master of the sensors this rank owns
        self.last = {}
        self.order = None                 # (parallel.CollectiveOrder, frame number) while a FrameStream runs this step

    def _turn(self, stage):
        """the collective of `stage` in the rank's global order (FrameStream with several frames in flight), else a no-op"""
        if self.order is None:
            return contextlib.nullcontext()
        return self.order[0].turn(self.order[1], stage)

    # -- calibration ------------------------------------------------------------------------------------------------
    def calibrate(self, depth: torch.Tensor):
        """depth (S_local, n_px) u16 of this rank's sensors -> transforms (S_local, 4, 4) f64 (identity for the master)"""
        return self._calibrate(depth, None)[0]

    def _calibrate(self, depth, queue_next):
        o, p = self.ops, self.p
        S = depth.shape[0]
        assert S == len(self.sensors)
        fp, _, _, fcnt = o.depth_to_cloud(depth, self.xy, None, S, False, False, sync=False)
        nxt = queue_next() if queue_next is not None else None     # queued behind the extraction, before the first read-back
        fk = o._count(fcnt)
        downs = [d[0] for d in o.voxel_downsample_batch([fp[i, :fk[i]] for i in range(S)], p.reg_voxel)]
        owns_master = self.sensors[0] == 0
        master = downs[0] if owns_master else None
        tn = o.estimate_normals(master, 2.0 * p.reg_voxel, p.normals_nn) if (owns_master and p.icp_mode == "p2plane") else None
        with self._turn(0):
            master, tn = self._bcast(master, tn, device=fp.device)                      # collective 1 (world > 1)
        subs = downs[1:] if owns_master else downs
        ids = [g for g in self.sensors if g != 0]
        Ts = [np.eye(4)] if owns_master else []
        stats = []
        if subs:
            rs = o.icp_batch(subs, master, p.icp_max_dist, [self.init[g - 1] for g in ids], p.icp_mode, tn, p.icp_max_iteration)
            for r in rs:
                Ts.append(r["transformation"])
                stats.append((r["iterations"], r["fitness"], r["inlier_rmse"]))
        self.transforms = np.stack(Ts)
        self.last = {"icp": stats, "n_down": [int(d.shape[0]) for d in downs], "n_master": int(master.shape[0])}
        return self.transforms, nxt

    # -- frame --------------------------------------------------------------------------------------------------------
    def fuse(self, depth: torch.Tensor, rgb: torch.Tensor, extracted=None):
        """depth (S_local, n_px) u16, rgb (S_local, n_px, 3) u8 -> (points, colours, transforms (n_sensors, 4, 4)) of the
        fused, filtered frame; points / colours are None on ranks > 0 with fused_filter == "rank0"."""
        o, p = self.ops, self.p
        if self.transforms is None:
            raise RuntimeError("calibrate() first (or load transforms)")
        S = depth.shape[0]
        mp, mc, _, mcnt = extracted if extracted is not None else o.depth_to_cloud(depth, self.xy, rgb, S, True, True, gate=p.gate, sync=False)
        mk = o._count(mcnt)
        local = [(mp[i, :mk[i]], mc[i, :mk[i]]) for i in range(S)]
        with self._turn(1):
            segs, all_T, counts = self._xchg(local, self.transforms)                      # collective 2 (world > 1)
        owned = [len(parallel.shard_sensors(self.n_sensors, r, self.world)) for r in range(self.world)]
        Ts = np.concatenate([all_T[r, :owned[r]] for r in range(self.world)])             # sensor order
        clouds = [seg for r in range(self.world) for seg in segs[r][:owned[r]]]
        self.last.update(n_masked=[int(k) for k in mk], n_fused=int(counts.sum()), counts=[int(c) for c in counts.sum(1)])
        dist_on = parallel.collectives_on(self.group)
        if dist_on and self.fused_filter == "rank0" and self.rank != 0:
            return None, None, Ts
        # pcd.transform(T_i) + np.vstack + voxel_down_sample (data.py:44-61) in one pass, on the fp64 values of the moved points
        vp, vc = o.fuse_voxel_downsample([c[0] for c in clouds], [c[1] for c in clouds], Ts, p.filt_voxel)
        M = int(vp.shape[0])
        if dist_on and self.fused_filter == "sharded" and M > 0:
            rows = -(-M // self.world)                                                    # slab r = grid-order positions [r rows, (r+1) rows)
            q0, q1 = min(M, self.rank * rows), min(M, (self.rank + 1) * rows)
            part, order = o.sor_partial(vp, p.filt_k, q0, q1)
            with self._turn(2):
                slabs = parallel.allgather_slabs(part, rows, self.group)                   # collective 3
            keep, _, _ = o.sor_finish(slabs.reshape(-1)[:M], order, p.filt_ratio)
        else:
            if self.order is not None:
                self.order[0].skip(self.order[1], 2)                                       # this frame has no third collective
            if hasattr(o, "sor_select"):                                                   # filter + selection without a count read-back between them
                out_p, out_c, keep, _ = o.sor_select(vp, vc, p.filt_k, p.filt_ratio)
                self.last.update(n_voxel=M, n_out=int(out_p.shape[0]))
                return out_p, out_c, Ts
            keep, _, _ = o.sor(vp, p.filt_k, p.filt_ratio)
        out_p, out_c, _ = o.select_by_index([vp, vc], keep, trusted=True)
        self.last.update(n_voxel=M, n_out=int(out_p.shape[0]))
        return out_p, out_c, Ts

    def step(self, depth: torch.Tensor, rgb: torch.Tensor):
        """calibration + frame on the same depth images (the bench's unit).  Both extractions are queued before the first
        count is read back; the person clouds are only needed after the registration, so their counts are fetched then."""
        o, p = self.ops, self.p
        S = depth.shape[0]
        _, masked = self._calibrate(depth, lambda: o.depth_to_cloud(depth, self.xy, rgb, S, True, True, gate=p.gate, sync=False))
        return self.fuse(depth, rgb, masked)


class NativeFramePipeline:
    """The same step as SensorShardPipeline.step on ONE GPU, as a single native call (kpx_frame_step): the host side of the
    frame loop runs in C++ inside the library, the interpreter only hands the frame over -- so several frames in flight on
    host threads (FrameStream) do not queue up behind the GIL."""

    def __init__(self, xy_table, n_sensors: int, init_transforms: List[np.ndarray], params: Optional[PipelineParams] = None,
                 out_ring: int = 0):
        """out_ring > 0: the fused clouds are written into a ring of that many pre-allocated output buffers owned by this
        pipeline (no allocator traffic per frame); a result then stays valid until `out_ring` more frames have been stepped
        through THIS pipeline -- clone what has to live longer.  0: fresh buffers per frame."""
        self.p = params or PipelineParams()
        self.n_sensors = int(n_sensors)
        self._ring, self._ring_n, self._ring_k = [], int(out_ring), 0
        if len(init_transforms) != self.n_sensors - 1:
            raise ValueError("init_transforms: one 4x4 per sub sensor (sensors 1 .. n-1)")
        self.init = [np.asarray(T, dtype=np.float64) for T in init_transforms]
        self.xy = ops._dev(xy_table, torch.float32).reshape(-1)
        p = self.p
        self._c = ops.FrameParams(p.reg_voxel, p.icp_max_dist, p.filt_voxel, p.filt_ratio, p.gate, p.normals_nn,
                                  ICP_MODES[p.icp_mode], p.icp_max_iteration, p.filt_k)
        self.last = {}

    def step(self, depth: torch.Tensor, rgb: torch.Tensor):
        """depth (S, n_px) u16, rgb (S, n_px, 3) u8: device tensors, or (pinned) host tensors -- then the copy to the device is
        part of the frame, on the frame's stream (kpx_frame_step_host)"""
        S = self.n_sensors
        out = None
        if self._ring_n:
            if len(self._ring) < self._ring_n:
                rows = S * int(depth.numel() // S)
                self._ring.append(tuple(torch.empty((rows, 3), dtype=torch.float32, device=self.xy.device) for _ in range(2)))
            out = self._ring[self._ring_k % len(self._ring)]
            self._ring_k += 1
        out_p, out_c, Ts, info = ops.frame_step(depth, rgb, self.xy, self.init, self._c, out=out)
        self.last = {"icp": [(int(info[32 + i]), None, None) for i in range(1, S)], "n_down": [int(v) for v in info[:S]],
                     "n_masked": [int(v) for v in info[16:16 + S]], "n_fused": int(info[16:16 + S].sum()), "n_voxel": int(info[48]),
                     "n_out": int(out_p.shape[0])}
        return out_p, out_c, Ts


class NativeShardPipeline:
    """SensorShardPipeline.step with the host side of the frame AND its three collectives inside the library
    (kpx_frame_step_sharded): sensor g on GPU g, master broadcast, per-rank registration, all-gather of the masked clouds,
    fused fp64 transform + voxel, sharded (or rank-0) filter.  `comm`: parallel.NativeComm (RCCL on a real node; the staged and
    in-process transports rehearse the same loop).  depth / rgb of THIS rank's sensors, device or (pinned) host tensors."""

    def __init__(self, xy_table, n_sensors: int, init_transforms: List[np.ndarray], params: Optional[PipelineParams] = None,
                 comm=None, fused_filter: str = "sharded", out_ring: int = 0):
        if comm is None:
            raise ValueError("NativeShardPipeline needs a communicator (parallel.NativeComm.rccl / staged / local)")
        self.p = params or PipelineParams()
        if self.p.icp_mode == "fixed":
            raise ValueError('icp_mode "fixed" (no registration in the frame) is a one-GPU form: SensorGroupPipeline / NativeFramePipeline')
        self.comm = comm
        self.n_sensors = int(n_sensors)
        if comm.world > self.n_sensors:
            raise ValueError(f"{comm.world} ranks for {self.n_sensors} sensors: a rank needs at least one sensor")
        self.sensors = parallel.shard_sensors(self.n_sensors, comm.rank, comm.world)
        if len(init_transforms) != self.n_sensors - 1:
            raise ValueError("init_transforms: one 4x4 per sub sensor (sensors 1 .. n-1)")
        self.init = [np.asarray(T, dtype=np.float64) for T in init_transforms]
        if fused_filter not in ("rank0", "sharded", "round_robin"):
            raise ValueError("fused_filter must be 'rank0', 'sharded' or 'round_robin'")
        self.fused_filter = fused_filter
        self.xy = ops._dev(xy_table, torch.float32).reshape(-1)
        p = self.p
        self._c = ops.FrameParams(p.reg_voxel, p.icp_max_dist, p.filt_voxel, p.filt_ratio, p.gate, p.normals_nn,
                                  ICP_MODES[p.icp_mode], p.icp_max_iteration, p.filt_k)
        self._ring, self._ring_n, self._ring_k = [], int(out_ring), 0
        self.order = None                 # (parallel.NativeCollectiveOrder, frame number) while a FrameStream runs this step
        self.native_order = True
        self.retries = 0
        self.last = {}

    def step(self, depth: torch.Tensor, rgb: torch.Tensor):
        """-> (points, colours, transforms) -- points / colours empty on ranks > 0 with fused_filter == "rank0" -- or the RETRY
        marker when this frame runs under a FrameStream (which submits it again in program order); serially it is retried here."""
        from . import _lib
        S = self.n_sensors
        out = None
        if self._ring_n:
            if len(self._ring) < self._ring_n:
                rows = S * int(depth.numel() // max(1, len(self.sensors)))
                self._ring.append(tuple(torch.empty((rows, 3), dtype=torch.float32, device=self.xy.device) for _ in range(2)))
            out = self._ring[self._ring_k % len(self._ring)]
            self._ring_k += 1
        order, frame = self.order if self.order is not None else (None, None)
        while True:
            res = ops.frame_step_sharded(self.comm, order, frame, depth, rgb, self.xy, S, self.init, self._c,
                                         {"sharded": 0, "rank0": 1, "round_robin": 2}[self.fused_filter], out=out)
            if res is not _lib.RETRY:
                break
            self.retries += 1
            if order is not None:
                if self._ring_n:
                    self._ring_k -= 1     # the ring slot belongs to the logical frame: the resubmitted attempt writes the same buffer
                return _lib.RETRY
        out_p, out_c, Ts, info = res
        self.last = {"icp": [(int(info[32 + i]), None, None) for i in range(1, S)], "n_down": [int(v) for v in info[:S]],
                     "n_masked": [int(v) for v in info[16:16 + S]], "n_fused": int(info[16:16 + S].sum()), "n_voxel": int(info[48]),
                     "n_out": int(out_p.shape[0])}
        return out_p, out_c, Ts


class FrameStream:
    """Several frames of a stream in flight.  A frame is a chain of short, mostly latency-bound kernels (the ICP loop alone
    is ~75 dependent launches that keep a fraction of the CUs busy), and consecutive frames are independent, so `depth`
    of them run side by side: each on its own host thread (the library calls release the GIL; its lanes, workspaces and
    progress words are per thread) and its own HIP stream.  Results come back in submission order."""

    def __init__(self, pipe, depth: int = 2):
        """pipe: one pipeline shared by the slots (single-GPU pipelines keep no per-frame state), or a list with one pipeline
        per slot -- required for SensorShardPipeline on several ranks: every slot needs its own communicator
        (parallel.new_group()), because the collectives of different frames in flight interleave differently on each rank."""
        self.pipes = list(pipe) if isinstance(pipe, (list, tuple)) else None
        self.depth = len(self.pipes) if self.pipes else max(1, int(depth))
        self.pipe = self.pipes[0] if self.pipes else pipe
        self.device = torch.cuda.current_device()
        self.streams = [torch.cuda.Stream(device=self.device) for _ in range(self.depth)]
        self.pool = ThreadPoolExecutor(max_workers=self.depth)
        self.pending = deque()
        self.submitted = 0
        # several ranks: the frames' collectives are issued in one global order on every rank (parallel.CollectiveOrder)
        native = self.pipes is not None and all(getattr(p_, "native_order", False) for p_ in self.pipes)
        ordered = self.pipes is not None and self.depth > 1 and all(hasattr(p_, "order") for p_ in self.pipes) and \
            (native or parallel.collectives_on(getattr(self.pipes[0], "group", None)))
        # (round 5: the order's lookahead is its own number -- frame f's exchange goes behind the broadcast of frame f + lookahead, not
        # f + depth - 1: with every slot busy that frame cannot start before f - 1 has been collected)
        look = int(os.environ.get("KPX_ORDER_LOOKAHEAD", "2"))
        self.order = parallel.CollectiveOrder(max(1, min(self.depth, look + 1)), native=native) if ordered else None

    def _run(self, slot, depth, rgb, frame=None):
        torch.cuda.set_device(self.device)
        stream = self.streams[slot]
        pipe = self.pipes[slot] if self.pipes else self.pipe
        try:
            with torch.cuda.stream(stream):
                if not depth.is_cuda and not isinstance(pipe, (NativeFramePipeline, NativeShardPipeline)):
                    # frames handed over in pinned host memory: the copy is part of the frame (the native loop stages them itself)
                    depth = depth.to(self.device, non_blocking=True)
                    rgb = rgb.to(self.device, non_blocking=True)
                if frame is not None:
                    pipe.order = (self.order, frame)
                out = pipe.step(depth, rgb)
                stream.synchronize()             # the outputs are consumed on the caller's stream
        finally:
            if frame is not None:
                pipe.order = None
                self.order.finish(frame)         # stages the frame did not use (or did not reach) are passed
        return out

    def full(self) -> bool:
        return len(self.pending) >= self.depth

    def submit(self, depth: torch.Tensor, rgb: torch.Tensor):
        """queue one frame (call pop() first when full())"""
        assert not self.full()
        slot = self.submitted % self.depth
        self.submitted += 1
        frame = self.order.submit() if self.order is not None else None
        fut = self.pool.submit(self._run, slot, depth, rgb, frame)
        fut.frame, fut.args = frame, (slot, depth, rgb)
        self.pending.append(fut)

    def pop(self):
        """-> (points, colours, transforms) of the oldest frame in flight.  The tensors were allocated on the frame's side
        stream: they are handed to the caller's current stream with record_stream(), so that the caching allocator does not
        give their blocks back to the side stream (whose next frame would overwrite them) while kernels the caller queued
        asynchronously are still reading them."""
        from . import _lib
        fut = self.pending.popleft()
        while True:
            if self.order is not None:
                self.order.block(fut.frame)      # no frame can be submitted before this one is done: see CollectiveOrder
            try:
                out = fut.result()
            finally:
                if self.order is not None:
                    self.order.block(None)
            if out is not _lib.RETRY:
                break
            # a message outgrew its capacity on EVERY rank alike (kpx_frame_step_sharded): the frame runs again under a new frame
            # number, submitted here -- the same point of the main thread's program on every rank
            slot, depth, rgb = fut.args
            frame = self.order.submit() if self.order is not None else None
            args = fut.args
            fut = self.pool.submit(self._run, slot, depth, rgb, frame)
            fut.frame, fut.args = frame, args
        cur = torch.cuda.current_stream(self.device)
        for t in out:
            if isinstance(t, torch.Tensor) and t.is_cuda:
                t.record_stream(cur)
        return out

    @property
    def last(self):
        return self.pipe.last

    def close(self):
        while self.pending:
            self.pop()
        self.pool.shutdown()
        from . import _lib
        for s_ in self.streams:                  # the slots' scratch (hundreds of MB per frame workspace) goes with the stream
            _lib.release_workspace(s_.cuda_stream, self.device)


class NativeFrameStream:
    """Several frames in flight, scheduled inside the library (kpx_stream: C++ worker threads, one HIP stream and one slice of one
    workspace per slot; the interpreter only hands frames over and takes results).  Same interface as FrameStream -- submit / pop /
    full / pending / close / last -- over a NativeFramePipeline (one GPU) or a list of NativeShardPipelines, one per slot (several
    GPUs: every slot needs its own communicator).  `depth` frames run side by side; on one GPU up to 2 x depth may be queued (a free
    worker takes the oldest).  Results are written into a ring of capacity + depth output buffers owned by the stream: a result stays
    valid until `depth` more frames have been submitted after its pop -- clone what has to live longer."""

    def __init__(self, pipe, depth: int = 4):
        import ctypes as C
        from . import _lib as L
        self.pipes = list(pipe) if isinstance(pipe, (list, tuple)) else None
        self.depth = len(self.pipes) if self.pipes else max(1, int(depth))
        self.pipe = self.pipes[0] if self.pipes else pipe
        p0 = self.pipe
        if not isinstance(p0, (NativeFramePipeline, NativeShardPipeline)):
            raise TypeError("NativeFrameStream runs the native frame loops (NativeFramePipeline / NativeShardPipeline)")
        self.sharded = isinstance(p0, NativeShardPipeline)
        if self.sharded and self.pipes is None:
            raise ValueError("several ranks: one NativeShardPipeline (one communicator) per frame slot")
        lib = L.load()
        self._lib, self._L, self._C = lib, L, C
        self.S = p0.n_sensors
        self.S_local = len(p0.sensors) if self.sharded else self.S
        self.n_px = int(p0.xy.numel() // 2)
        self.dev = p0.xy.device
        rank, world = (p0.comm.rank, p0.comm.world) if self.sharded else (0, 1)
        nbytes = int(lib.kpx_stream_workspace_bytes(self.S, rank, world, self.n_px, self.depth))
        if nbytes <= 0:
            raise ValueError("kpx_stream_workspace_bytes: bad shape")
        self._ws = torch.empty(nbytes, dtype=torch.uint8, device=self.dev)
        init = np.ascontiguousarray(np.stack([ops._T(T) for T in p0.init])) if self.S > 1 else np.zeros((1, 4, 4))
        comms = None
        if self.sharded:
            comms = (C.c_void_p * self.depth)(*[getattr(p_.comm.handle, "value", p_.comm.handle) for p_ in self.pipes])
        fused = {"sharded": 0, "rank0": 1, "round_robin": 2}[p0.fused_filter] if self.sharded else 0
        h = C.c_void_p()
        with torch.cuda.device(self.dev):
            torch.cuda.synchronize()                  # the table and the workspace exist before a worker's stream touches them
            L.check(lib.kpx_stream_create(L.ptr(p0.xy), self.n_px, self.S, L.hptr(init), C.byref(p0._c), self.depth, comms, fused,
                                          C.c_void_p(self._ws.data_ptr()), nbytes, C.byref(h)))
        self.handle = h
        rows = self.S * self.n_px
        self.capacity = int(lib.kpx_stream_capacity(h))      # frames submit() takes before a pop(): 2 x depth on one GPU
        self._ring = [tuple(torch.empty((rows, 3), dtype=torch.float32, device=self.dev) for _ in range(2)) for _ in range(self.capacity + self.depth)]
        self.pending = deque()
        self.submitted = 0
        self._h_count = np.zeros(1, dtype=np.int32)
        self.last = {}

    def full(self) -> bool:
        return len(self.pending) >= self.capacity

    def submit(self, depth: torch.Tensor, rgb: torch.Tensor):
        """queue one frame (call pop() first when full()): depth (S_local, n_px) u16, rgb (S_local, n_px, 3) u8, device tensors or
        pinned host tensors; they are kept alive until the frame has been popped"""
        assert not self.full()
        L, C = self._L, self._C
        host = not depth.is_cuda
        if depth.dtype != torch.uint16 or rgb.dtype != torch.uint8 or not (depth.is_contiguous() and rgb.is_contiguous()):
            raise ValueError("NativeFrameStream.submit: contiguous uint16 depth / uint8 rgb tensors")
        if int(depth.numel()) != self.S_local * self.n_px or int(rgb.numel()) != 3 * self.S_local * self.n_px:
            raise ValueError("NativeFrameStream.submit: a frame is (sensors, n_px) depth + (sensors, n_px, 3) rgb")
        out = self._ring[self.submitted % len(self._ring)]
        L.check(self._lib.kpx_stream_submit(self.handle, C.c_void_p(depth.data_ptr()), C.c_void_p(rgb.data_ptr()), 1 if host else 0, L.ptr(out[0]), L.ptr(out[1])))
        self.submitted += 1
        self.pending.append((depth, rgb, out))

    def pop(self):
        """-> (points, colours, transforms) of the oldest frame in flight; the rows are complete when this returns"""
        L = self._L
        _, _, out = self.pending[0]
        h_T = np.zeros((self.S, 4, 4))
        h_info = np.zeros(64, dtype=np.int32)
        rc = self._lib.kpx_stream_pop(self.handle, self._h_count.ctypes.data_as(self._C.c_void_p), L.hptr(h_T), h_info.ctypes.data_as(self._C.c_void_p))
        self.pending.popleft()
        if self.sharded:
            for p_ in self.pipes:
                if p_.comm.error is not None:
                    e, p_.comm.error = p_.comm.error, None
                    raise e
        L.check(rc)
        k, S = int(self._h_count[0]), self.S
        self.last = {"icp": [(int(h_info[32 + i]), None, None) for i in range(1, S)], "n_down": [int(v) for v in h_info[:S]],
                     "n_masked": [int(v) for v in h_info[16:16 + S]], "n_fused": int(h_info[16:16 + S].sum()), "n_voxel": int(h_info[48]), "n_out": k}
        return out[0][:k], out[1][:k], h_T

    def stats(self):
        """frames finished, whether the registrations run through the device's ICP engine, its iteration launches and ticks so far"""
        out = np.zeros(4, dtype=np.uint64)
        self._L.check(self._lib.kpx_stream_stats(self.handle, out.ctypes.data_as(self._C.c_void_p)))
        return {"frames": int(out[0]), "icp_engine": bool(out[1]), "engine_launches": int(out[2]), "engine_ticks": int(out[3])}

    def close(self):
        if self.handle is not None:
            while self.pending:
                try:
                    self.pop()
                except Exception:                      # noqa: BLE001 -- closing: the frames' errors were the caller's to collect
                    pass
            self._lib.kpx_stream_destroy(self.handle)
            self.handle = None
            self._ws = None

    def __del__(self):
        try:
            self.close()
        except Exception:                              # noqa: BLE001
            pass
