"""The ICP chain in one launch (icp_chain_kernel, KPX_ICP_CHAIN=1) against a launch per iteration (=0), child processes on one box: the
bench frame's three registrations as one batch and its longest one alone, 30 iterations; host wall per call and the transforms compared
bit for bit.    python tools/icp_chain_ab.py"""
import os, subprocess, sys
code = r'''
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
out = {}
for name, srcs, ini in (("three", downs[1:], inits), ("one", [downs[3]], [inits[2]])):
    for kk in (30, 12):
        for _ in range(5):
            r = ops.icp_batch(srcs, downs[0], P.icp_max_dist, ini, P.icp_mode, tn, kk)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(40):
            r = ops.icp_batch(srcs, downs[0], P.icp_max_dist, ini, P.icp_mode, tn, kk)
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / 40 * 1e6
        out[f"{name}_{kk}"] = np.stack([np.asarray(x["transformation"], dtype=np.float64) for x in r])
        print(os.environ.get("KPX_ICP_CHAIN", "1"), name, "max_iteration", kk, "iterations", [x["iterations"] for x in r], "rows", [int(s.shape[0]) for s in srcs], f"{us:8.1f} us per call")
np.savez(sys.argv[1], **out)
'''
files = []
for chain in ("1", "0"):
    f = f"/tmp/icp_chain_ab_{chain}.npz"
    r = subprocess.run([sys.executable, "-c", code, f], env={**os.environ, "KPX_ICP_CHAIN": chain}, capture_output=True, text=True)
    print(r.stdout, r.stderr[-1500:] if r.returncode else "")
    files.append(f)
import numpy as np
a, b = np.load(files[0]), np.load(files[1])
for k in a.files:
    print("transforms bit-identical:", k, np.array_equal(a[k], b[k]))
