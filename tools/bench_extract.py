"""Only the extract rows of tools/bench_kernels.py (unproject / fused depth -> cloud at 256 frames)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kinectpy_amd import ops  # noqa: E402
from kinectpy_amd.utils import synth  # noqa: E402
from tools.bench_kernels import N_PX, report, timed  # noqa: E402

dev = torch.device("cuda")
xy = synth.xy_table()
base_d, person = synth.render_depth(xy=xy, return_person=True)
xyd = torch.as_tensor(xy).to(dev)
for F in (4, 32, 256):
    depth = torch.as_tensor(np.tile(base_d, (F, 1))).to(dev)
    ms, _ = timed(lambda: ops.unproject_u16(depth, xyd, F))
    report(f"unproject_u16, {F} frames", ms, F * N_PX * 8, frames=F)
    ms, r = timed(lambda: ops.depth_to_cloud(depth, xyd, None, F, False, False, sync=False) if "sync" in ops.depth_to_cloud.__code__.co_varnames else ops.depth_to_cloud(depth, xyd, None, F, False, False))
    report(f"depth_to_cloud no colour, {F} frames", ms, frames=F)
