"""segment_plane(30, 30, 2000) at BASELINE config 3 sizes: the scoring kernel (library profiler) and the whole call.
python tools/plane_probe.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kinectpy_amd import ops  # noqa: E402
from kinectpy_amd.utils import synth  # noqa: E402

c3 = torch.as_tensor(synth.filter_cloud(1_000_000)).cuda()
for n in (146_000, 1_000_000):
    cloud = c3[:n].contiguous()
    for _ in range(2):
        pl, inl = ops.segment_plane(cloud, 30.0, 30, 2000, probability=1.0, seed=7)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(5):
        e0.record(); ops.segment_plane(cloud, 30.0, 30, 2000, probability=1.0, seed=7); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ops.prof_stride(1); ops.prof_begin(64)
    for _ in range(5):
        ops.segment_plane(cloud, 30.0, 30, 2000, probability=1.0, seed=7)
    torch.cuda.synchronize()
    ms, cnt, _ = ops.prof_end()["plane_score"]
    flop = 8.0 * 2000 * n
    print(f"n = {n}: call {np.median(ts):.3f} ms, scoring kernel {ms / cnt * 1e3:.1f} us = {flop / (ms / cnt) / 1e9:.2f} TFLOP/s-equivalent (8 flop per point and hypothesis), "
          f"{int(inl.shape[0])} inliers, plane {np.round(pl, 6)}")
