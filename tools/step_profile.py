"""dev-only: wall time of each phase of one pipeline step (with syncs)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from kinectpy_amd import ops
from kinectpy_amd.pipeline import PipelineParams
xy, depth_h, rgb_h, inits, truth, _ = bench.make_group(0, 1, 4, 1)
depth = torch.as_tensor(depth_h[0]).cuda(); rgb = torch.as_tensor(rgb_h[0]).cuda(); xyd = torch.as_tensor(xy).cuda().reshape(-1)
p = PipelineParams(); S = 4
def T(name, fn, reps=5):
    fn(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): out = fn()
    torch.cuda.synchronize(); print(f"{name:40s} {(time.perf_counter()-t)/reps*1e3:8.3f} ms"); return out
full = T("depth_to_cloud full x4", lambda: ops.depth_to_cloud(depth, xyd, None, S, False, False))
masked = T("depth_to_cloud masked x4", lambda: ops.depth_to_cloud(depth, xyd, rgb, S, True, True))
T("voxel 35 x4, one after the other", lambda: [ops.voxel_downsample(full[i][0], 35.0)[0] for i in range(S)])
downs = T("voxel 35 x4, batch (what the pipeline calls)", lambda: [d[0] for d in ops.voxel_downsample_batch([full[i][0] for i in range(S)], 35.0)])
tn = T("normals", lambda: ops.estimate_normals(downs[0], 70.0, 40))
rs = T("icp_batch x3", lambda: ops.icp_batch(downs[1:], downs[0], 100.0, inits, "p2plane", tn, 30), reps=3)
Ts = [np.eye(4)] + [r["transformation"] for r in rs]
vp = T("fused transform + stack + voxel 10", lambda: ops.fuse_voxel_downsample([m[0] for m in masked], [m[1] for m in masked], Ts, 10.0))
keep = T("sor k=20", lambda: ops.sor(vp[0], 20, 2.0))
out = T("select", lambda: ops.select_by_index([vp[0], vp[1]], keep[0], trusted=True))
# one registration alone vs. the batch of three: how much of icp_batch is the latency chain of a single problem
for i in range(1, S):
    r = T(f"icp alone, sub {i}", lambda: ops.icp(downs[i], downs[0], 100.0, inits[i - 1], "p2plane", tn, 30), reps=3)
    print("    iterations", r["iterations"])
T("nn_search sub1 -> master (target prep + source prep + one sweep)", lambda: ops.nn_search(downs[1], downs[0], inits[0]), reps=5)
