"""Self-check of the row certificates of the culled ICP sweep: run with KPX_ICP_CERT_CHECK=1 -- certified rows are searched all the
same and the search's winner is compared with the kept partner (kpx_prof_icp_cert).  Cases: the registrations of
test_icp_batch_equals_single_problems (both modes) and the bench's three registrations.

    KPX_ICP_CERT_CHECK=1 python tools/cert_check.py
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kinectpy_amd import ops  # noqa: E402
from kinectpy_amd.pipeline import PipelineParams  # noqa: E402
from kinectpy_amd.utils import synth  # noqa: E402
from oracle import oracle as O  # noqa: E402

assert os.environ.get("KPX_ICP_CERT_CHECK") == "1", "set KPX_ICP_CERT_CHECK=1"
base = synth.frame_cloud()
src, tgt, T = synth.icp_pair(12000, base)
tn = O.estimate_normals(tgt, 70.0, 40)[0].astype(np.float32)
srcs = [src, src[:7001], O.transform(src, synth.t_star())[:9000]]
inits = [np.eye(4), np.eye(4), np.linalg.inv(synth.t_star())]
ops.prof_icp_cert()
for mode, nrm in (("p2p", None), ("p2plane", tn)):
    for i in range(3):
        b = ops.icp_batch([srcs[i]], tgt, 100.0, [inits[i]], mode, nrm, 12)[0]
        torch.cuda.synchronize()
        print(mode, i, "iterations", b["iterations"], "fitness", b["fitness"], ops.prof_icp_cert())
P = PipelineParams()
xy, depth_h, rgb_h, inits4, _ = synth.sensor_ring(4, 1)
depth = torch.as_tensor(depth_h[0]).cuda()
fp, _, _, fcnt = ops.depth_to_cloud(depth, xy, None, 4, False, False, sync=False)
fk = ops._count(fcnt)
downs = [d[0] for d in ops.voxel_downsample_batch([fp[i, :fk[i]] for i in range(4)], P.reg_voxel)]
tn4 = ops.estimate_normals(downs[0], 2.0 * P.reg_voxel, P.normals_nn)
for i in range(3):
    b = ops.icp_batch([downs[1 + i]], downs[0], P.icp_max_dist, [inits4[i]], P.icp_mode, tn4, P.icp_max_iteration)[0]
    torch.cuda.synchronize()
    print("bench registration", i, "iterations", b["iterations"], ops.prof_icp_cert())

print("per-iteration state of the bench's longest registration (max_iteration = k):")
prev = (0, 0)
for kk in range(1, P.icp_max_iteration + 1):
    b = ops.icp_batch([downs[3]], downs[0], P.icp_max_dist, [inits4[2]], P.icp_mode, tn4, kk)[0]
    torch.cuda.synchronize()
    c = ops.prof_icp_cert()
    print(f"  k<={kk:2d} it {b['iterations']:2d} certified {c['certified'] - prev[0]:6d} searched {c['searched'] - prev[1]:6d}  last_motion {c.get('last_motion', float('nan')):9.4f} "
          f"motion {c.get('motion', float('nan')):9.3f} skin {c.get('skin', float('nan')):6.2f}  rmse {b['inlier_rmse']:.6f} fitness {b['fitness']:.6f}")
    prev = (c['certified'], c['searched'])
