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
from typing import List, Optional

import numpy as np
import torch

from . import ops, parallel


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
        fp, _, _, fcnt = ops.depth_to_cloud(depth, self.xy, None, S, False, False, sync=False)                 # registration input
        mp, mc, _, mcnt = ops.depth_to_cloud(depth, self.xy, rgb, S, True, True, gate=p.gate, sync=False)     # person clouds
        fk = ops._count(fcnt)
        # -- registration: every sub onto the master, exactly execute_point_to_plane_registration
        downs = [d[0] for d in ops.voxel_downsample_batch([fp[i, :fk[i]] for i in range(S)], p.reg_voxel)]
        tn = ops.estimate_normals(downs[0], 2.0 * p.reg_voxel, p.normals_nn) if p.icp_mode == "p2plane" else None
        Ts = [np.eye(4)]
        stats = []
        if S > 1:
            rs = ops.icp_batch(downs[1:], downs[0], p.icp_max_dist, self.init, p.icp_mode, tn, p.icp_max_iteration)
            for r in rs:
                Ts.append(r["transformation"])
                stats.append((r["iterations"], r["fitness"], r["inlier_rmse"]))
        # -- transform + fuse + filter
        mk = ops._count(mcnt)
        masked = [(mp[i, :mk[i]], mc[i, :mk[i]]) for i in range(S)]
        pts = [masked[0][0]] + [ops.transform(masked[i][0], Ts[i]) for i in range(1, S)]
        fused_p = torch.cat(pts, 0)
        fused_c = torch.cat([m[1] for m in masked], 0)
        vp, vc, _ = ops.voxel_downsample(fused_p, p.filt_voxel, fused_c)
        keep, _, _ = ops.sor(vp, p.filt_k, p.filt_ratio)
        out_p, out_c, _ = ops.select_by_index([vp, vc], keep, trusted=True)
        self.last = {"icp": stats, "n_down": [int(d.shape[0]) for d in downs], "n_masked": [int(m[0].shape[0]) for m in masked],
                     "n_fused": int(fused_p.shape[0]), "n_out": int(out_p.shape[0])}
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


class FrameStream:
    """Several frames of a stream in flight.  A frame is a chain of short, mostly latency-bound kernels (the ICP loop alone
    is ~75 dependent launches that keep a fraction of the CUs busy), and consecutive frames are independent, so `depth`
    of them run side by side: each on its own host thread (the library calls release the GIL; its lanes, workspaces and
    progress words are per thread) and its own HIP stream.  Results come back in submission order."""

    def __init__(self, pipe: SensorGroupPipeline, depth: int = 2):
        self.pipe, self.depth = pipe, max(1, int(depth))
        self.device = torch.cuda.current_device()
        self.streams = [torch.cuda.Stream(device=self.device) for _ in range(self.depth)]
        self.pool = ThreadPoolExecutor(max_workers=self.depth)
        self.pending = deque()
        self.submitted = 0

    def _run(self, stream, depth, rgb):
        torch.cuda.set_device(self.device)
        with torch.cuda.stream(stream):
            out = self.pipe.step(depth, rgb)
            stream.synchronize()                 # the outputs are consumed on the caller's stream
        return out

    def full(self) -> bool:
        return len(self.pending) >= self.depth

    def submit(self, depth: torch.Tensor, rgb: torch.Tensor):
        """queue one frame (call pop() first when full())"""
        assert not self.full()
        stream = self.streams[self.submitted % self.depth]
        self.submitted += 1
        self.pending.append(self.pool.submit(self._run, stream, depth, rgb))

    def pop(self):
        """-> (points, colours, transforms) of the oldest frame in flight.  The tensors were allocated on the frame's side
        stream: they are handed to the caller's current stream with record_stream(), so that the caching allocator does not
        give their blocks back to the side stream (whose next frame would overwrite them) while kernels the caller queued
        asynchronously are still reading them."""
        out = self.pending.popleft().result()
        cur = torch.cuda.current_stream(self.device)
        for t in out:
            if isinstance(t, torch.Tensor) and t.is_cuda:
                t.record_stream(cur)
        return out

    def close(self):
        while self.pending:
            self.pop()
        self.pool.shutdown()
