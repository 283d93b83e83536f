"""The masked + gated + coloured extract at 256 frames (for rocprofv3 --kernel-trace --stats)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kinectpy_amd import ops  # noqa: E402
from kinectpy_amd.utils import synth  # noqa: E402
from tools.bench_kernels import N_PX, report, timed  # noqa: E402

dev = torch.device("cuda")
F = 256
xy = synth.xy_table()
base_d, person = synth.render_depth(xy=xy, return_person=True)
rgb1 = synth.mask_rgb(person)
depth = torch.as_tensor(np.tile(base_d, (F, 1))).to(dev)
rgb = torch.as_tensor(np.tile(rgb1, (F, 1, 1))).to(dev)
xyd = torch.as_tensor(xy).to(dev)
ms, r = timed(lambda: ops.depth_to_cloud(depth, xyd, rgb, F, True, True))
kept = sum(int(t[0].shape[0]) for t in r) if isinstance(r, list) else 0
report("depth_to_cloud mask+gate+colour", ms, F * N_PX * 5 + kept * 28, frames=F, kept=kept)
