"""the frame's three registrations as ONE batch, 8 calls (for a kernel trace: tools/icp_batch_trace.sh prints every launch's duration)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for _ in range(8):
    r = ops.icp_batch(downs[1:1 + n], downs[0], P.icp_max_dist, inits[:n], P.icp_mode, tn, 30)
    torch.cuda.synchronize()
print("iterations", [x["iterations"] for x in r])
