import os, sys, time, numpy as np, torch
sys.path.insert(0, os.getcwd())
from kinectpy_amd import ops
from kinectpy_amd.pipeline import PipelineParams
from kinectpy_amd.utils import synth
P = PipelineParams()
xy, depth_h, rgb_h, inits, _ = synth.sensor_ring(4, 1)
depth = torch.as_tensor(depth_h[0]).cuda()
fp, _, _, fcnt = ops.depth_to_cloud(depth, xy, None, 4, False, False, sync=False)
fk = ops._count(fcnt)
downs = [d[0] for d in ops.voxel_downsample_batch([fp[i, :fk[i]] for i in range(4)], P.reg_voxel)]
tn = ops.estimate_normals(downs[0], 2.0 * P.reg_voxel, P.normals_nn)
for kk in (4, 6, 8, 10, 12, 14, 16, 20, 30):
    for sync_each in (False, True):
        for _ in range(3):
            r = ops.icp_batch(downs[1:], downs[0], P.icp_max_dist, inits, P.icp_mode, tn, kk)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            r = ops.icp_batch(downs[1:], downs[0], P.icp_max_dist, inits, P.icp_mode, tn, kk)
            if sync_each: torch.cuda.synchronize()
        torch.cuda.synchronize()
        print("max_iteration", kk, "sync each call" if sync_each else "back to back  ", [x["iterations"] for x in r], f"{(time.perf_counter() - t0) / 20 * 1e6:8.1f} us per call", flush=True)
