"""dev-only: per-iteration timing of the correspondence sweeps on the bench clouds"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from kinectpy_amd import ops
xy, depth_h, rgb_h, inits, truth, _ = bench.make_group(0, 1, 4, 1)
full = ops.depth_to_cloud(torch.as_tensor(depth_h[0]).cuda(), xy, None, 4, False, False)
downs = [ops.voxel_downsample(f[0], 35.0)[0] for f in full]
tn = ops.estimate_normals(downs[0], 70.0, 40)
for i in (1, 2, 3):
    for rep in range(2):
        ops.prof_begin(256)
        r = ops.icp(downs[i], downs[0], 100.0, inits[i - 1], "p2plane", tn, 30)
        torch.cuda.synchronize()
        p = ops.prof_end()
    print(f"sub {i}: it={r['iterations']} f64 sweeps {p['nn_mfma'][1]} avg {p['nn_mfma'][0]/max(p['nn_mfma'][1],1)*1e3:.1f} us | f32 sweeps {p['nn_screen'][1]} avg {p['nn_screen'][0]/max(p['nn_screen'][1],1)*1e3:.1f} us")
import time
for mode in ("p2plane",):
    torch.cuda.synchronize(); t = time.time()
    for _ in range(5):
        ops.icp_batch(downs[1:], downs[0], 100.0, inits, mode, tn, 30)
    torch.cuda.synchronize(); print("batch of 3:", (time.time() - t) / 5 * 1e3, "ms")
