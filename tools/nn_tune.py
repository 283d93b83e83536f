"""dev-only: time the NN sweep for a few shapes / launch plans"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from kinectpy_amd import ops
from kinectpy_amd.utils import synth
base = synth.frame_cloud()
for n in (30000, 100000):
    src, tgt, T = synth.icp_pair(n, base)
    s, t = torch.as_tensor(src).cuda(), torch.as_tensor(tgt).cuda()
    for it in range(2):
        ops.prof_begin(64)
        for _ in range(5):
            ops.nn_search(s, t, np.eye(4))
        torch.cuda.synchronize()
        ms, cnt, fl = ops.prof_end()["nn_mfma"]
    print(f"blocks={os.environ.get('KPX_NN_BLOCKS','4096'):>6s} n={n}: main sweep {ms/cnt*1e3:8.1f} us  {fl/ms/1e9:6.2f} TFLOP/s")
    r = ops.icp(s, t, 100.0, None, "p2p", None, 10, want_corr=False)
    ops.prof_begin(64); r = ops.icp(s, t, 100.0, None, "p2p", None, 10); torch.cuda.synchronize(); ms, cnt, fl = ops.prof_end()["nn_mfma"]
    print(f"   icp 10 it: {cnt} sweeps, avg {ms/cnt*1e3:8.1f} us  {fl/ms/1e9:6.2f} TFLOP/s")
