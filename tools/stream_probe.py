"""unproject_u16 (256 frames) and transform (64M points): GB/s algorithmic.  python tools/stream_probe.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kinectpy_amd import ops  # noqa: E402
from kinectpy_amd.utils import synth  # noqa: E402


def timed(fn, reps=7, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(reps):
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))


xy = synth.xy_table()
F = 256
depth = torch.as_tensor(np.tile(synth.render_depth(xy=xy), (F, 1))).cuda()
xyd = torch.as_tensor(xy).cuda()
ms = timed(lambda: ops.unproject_u16(depth, xyd, F))
print(f"unproject_u16 {F} frames: {ms:.4f} ms  {F * 576 * 640 * 8 / ms / 1e6:.0f} GB/s")
big = torch.rand((64_000_000, 3), device="cuda") * 3000
out = torch.empty_like(big)
ms = timed(lambda: ops.transform(big, synth.t_star(), out=out))
print(f"transform 64M points (out of place): {ms:.4f} ms  {64e6 * 24 / ms / 1e6:.0f} GB/s")
ms = timed(lambda: ops.transform(big, synth.t_star(), out=big))
print(f"transform 64M points (in place)    : {ms:.4f} ms  {64e6 * 24 / ms / 1e6:.0f} GB/s")
ms = timed(lambda: out.copy_(big))
print(f"device copy 768 MB                 : {ms:.4f} ms  {64e6 * 24 / ms / 1e6:.0f} GB/s")
