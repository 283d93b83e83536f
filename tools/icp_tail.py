"""per-launch time of the ICP chain's TAIL, free of event pairs: the bench's longest registration (30 iterations) alone, run to
max_iteration = 12, 18, 24, 30 -- the differences are whole launches of the late iterations.  KPX_ICP_CERT=0/1 in child processes."""
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
res = {}
for kk in (12, 18, 24, 30):
    for _ in range(5):
        ops.icp_batch([downs[3]], downs[0], P.icp_max_dist, [inits[2]], P.icp_mode, tn, kk)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(40):
        r = ops.icp_batch([downs[3]], downs[0], P.icp_max_dist, [inits[2]], P.icp_mode, tn, kk)
    torch.cuda.synchronize()
    res[kk] = (time.perf_counter() - t0) / 40 * 1e6
    print(os.environ.get("KPX_ICP_CERT", "1"), "max_iteration", kk, "iterations", r[0]["iterations"], f"{res[kk]:8.1f} us per registration")
print(os.environ.get("KPX_ICP_CERT", "1"), "per launch: iterations 12-18 %.2f us, 18-24 %.2f us, 24-30 %.2f us" % ((res[18] - res[12]) / 6, (res[24] - res[18]) / 6, (res[30] - res[24]) / 6))
'''
for cert in ("1", "0"):
    r = subprocess.run([sys.executable, "-c", code], env={**os.environ, "KPX_ICP_CERT": cert}, capture_output=True, text=True)
    print(r.stdout, r.stderr[-800:] if r.returncode else "")
