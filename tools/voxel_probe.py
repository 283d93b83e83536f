"""voxel_down_sample at BASELINE config 3 sizes: one 1M-point cloud (with and without colours) and 64 clouds with colours.
python tools/voxel_probe.py [clouds]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kinectpy_amd import ops  # noqa: E402
from kinectpy_amd.utils import synth  # noqa: E402


def timed(fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(reps):
        e0.record(); out = fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts)), out


nc = int(sys.argv[1]) if len(sys.argv) > 1 else 64
c3 = torch.as_tensor(synth.filter_cloud(1_000_000)).cuda()
col = torch.rand_like(c3)
ms, out = timed(lambda: ops.voxel_downsample(c3, 10.0))
print(f"one 1M cloud, no colours : {ms * 1e3:8.1f} us   {out[0].shape[0]} voxels   {(12e6 + 12 * out[0].shape[0]) / ms / 1e6:7.1f} GB/s algorithmic")
ms, out = timed(lambda: ops.voxel_downsample(c3, 10.0, col))
print(f"one 1M cloud, colours    : {ms * 1e3:8.1f} us   {(24e6 + 24 * out[0].shape[0]) / ms / 1e6:7.1f} GB/s algorithmic")
clouds = [(c3 + float(k)).contiguous() for k in range(nc)]
cols = [torch.rand_like(c3) for _ in range(nc)]
ms, outs = timed(lambda: ops.voxel_downsample_batch(clouds, 10.0, cols), reps=3, warm=1)
m_tot = sum(int(o[0].shape[0]) for o in outs)
print(f"{nc} x 1M clouds, colours : {ms:8.3f} ms   {ms * 1e3 / nc:7.1f} us per cloud   {(24e6 * nc + 24 * m_tot) / ms / 1e6:7.1f} GB/s algorithmic = {(24e6 * nc + 24 * m_tot) / ms / 1e6 / 8000:.3f} of 8 TB/s")

# the same clouds in the order a sensor delivers them (rows of 5 mm in y, ascending x): neighbours in memory are neighbours in space
h = c3.cpu().numpy()
scan = torch.as_tensor(h[np.lexsort((h[:, 0], np.floor(h[:, 1] / 5.0)))]).cuda()
clouds_s = [(scan + float(k)).contiguous() for k in range(nc)]
ms, outs_s = timed(lambda: ops.voxel_downsample_batch(clouds_s, 10.0, cols), reps=3, warm=1)
m_s = sum(int(o[0].shape[0]) for o in outs_s)
print(f"{nc} x 1M clouds in scan order : {ms:8.3f} ms   {ms * 1e3 / nc:7.1f} us per cloud   {(24e6 * nc + 24 * m_s) / ms / 1e6:7.1f} GB/s algorithmic = {(24e6 * nc + 24 * m_s) / ms / 1e6 / 8000:.3f} of 8 TB/s   (voxels {m_s} vs {m_tot})")
