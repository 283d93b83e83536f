"""Dense all-pairs engine at BASELINE config 2 (100k x 100k): TFLOP/s of nn_mfma_kernel (HIP events on its stream via the library's
profiler) and a bit-exactness check against the culled engine.  python tools/nn_dense_probe.py [n]"""
import os
import sys

os.environ.setdefault("KPX_NN_SCREEN", "0")       # registrations below: every iteration on the fp64 sweep (no f32 screening)

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kinectpy_amd import ops  # noqa: E402
from kinectpy_amd.utils import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
src, tgt, _ = synth.icp_pair(n)
s, t = torch.as_tensor(src).cuda(), torch.as_tensor(tgt).cuda()
ops.nn_engine("culled")
ri, rd = ops.nn_search(s, t, np.eye(4))
prev = ops.nn_engine("dense")
for rep in range(2):
    gi, gd = ops.nn_search(s, t, np.eye(4))
torch.cuda.synchronize()
print("bit-exact vs culled engine:", bool(torch.equal(gi, ri) and torch.equal(gd, rd)))
ops.prof_stride(1)
ops.prof_begin(256)
for _ in range(5):
    ops.nn_search(s, t, np.eye(4))
torch.cuda.synchronize()
pr = ops.prof_end()
for k in ("nn_mfma", "nn_screen"):
    ms, cnt, work = pr[k]
    if cnt:
        print(f"{k}: {cnt} launches, {ms / cnt:.4f} ms each, {work / ms / 1e9:.2f} TFLOP/s = {work / ms / 1e9 / 78.6:.3f} of 78.6")
ops.nn_engine(prev)

# the warm case: the dense sweep inside a registration (every iteration after the first bounds its rows with the previous partner)

ops.nn_engine("culled")
ref = ops.icp(s, t, 100.0, None, "p2p", None, 30)
ops.nn_engine("dense")
ops.icp(s, t, 100.0, None, "p2p", None, 30)
torch.cuda.synchronize()
ops.prof_stride(1)
ops.prof_begin(1024)
g = ops.icp(s, t, 100.0, None, "p2p", None, 30)
torch.cuda.synchronize()
pr = ops.prof_end()
print("registration: iterations", g["iterations"], ref["iterations"], "T equal to the culled engine's within", float(np.abs(g["transformation"] - ref["transformation"]).max()))
for k in ("nn_mfma", "nn_screen"):
    ms, cnt, work = pr[k]
    if cnt:
        print(f"in a 30-iteration registration  {k}: {cnt} launches, {ms / cnt:.4f} ms each, {work / ms / 1e9:.2f} TFLOP/s = {work / ms / 1e9 / 78.6:.3f} of 78.6")
ops.nn_engine(prev)
