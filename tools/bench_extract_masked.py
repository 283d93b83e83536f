"""The batch extract rows of bench.py alone (256 frames, device time by HIP events, no host split): depth_to_cloud plain and
masked + gated + coloured, rgbd_compact from int16 XYZ.  A/B switches: KPX_MEDIAN_FRAME=0 (two radix passes instead of the
one-block-per-frame histogram), KPX_ONEPASS_BATCH=0 (count -> scan -> scatter instead of the frame-major one-pass kernel)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import ev_timed  # noqa: E402
from kinectpy_amd import ops  # noqa: E402
from kinectpy_amd.utils import synth  # noqa: E402

N_PX = 576 * 640
dev = torch.device("cuda")
F = 256
xy = synth.xy_table()
base_d, person = synth.render_depth(xy=xy, return_person=True)
depth = torch.as_tensor(np.tile(base_d, (F, 1))).to(dev)
rgb = torch.as_tensor(np.tile(synth.mask_rgb(person), (F, 1, 1))).to(dev)
xyd = torch.as_tensor(xy).to(dev)
xyz = ops.unproject_u16(depth, xyd, F)
tag = f"median_frame={os.environ.get('KPX_MEDIAN_FRAME', 'default')} onepass_batch={os.environ.get('KPX_ONEPASS_BATCH', 'default')}"
for name, fn, byts in (
        ("depth_to_cloud, no colour", lambda: ops.depth_to_cloud(depth, xyd, None, F, False, False, sync=False), 2),
        ("depth_to_cloud, mask + gate + colour", lambda: ops.depth_to_cloud(depth, xyd, rgb, F, True, True, sync=False), 5),
        ("rgbd_compact from int16 XYZ", lambda: ops.rgbd_compact(xyz, rgb, F, True, True, want_idx=False, sync=False), 9)):
    ms, res = ev_timed(torch, fn, reps=5, warm=2)
    kept = int(res[3].sum().item())
    alg = F * N_PX * byts + kept * (12 if byts == 2 else 24)
    print(f"{tag:50s} {name:40s} {ms:8.4f} ms  {alg / ms / 1e6:8.1f} GB/s  frac {alg / ms / 1e6 / 8000:6.4f}  kept {kept}")
