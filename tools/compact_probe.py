"""The three extraction / compaction rows of bench.roofline_targets on 256 frames, GB/s algorithmic, with a checksum of the
outputs (so that two library variants can be compared in one call).  python tools/compact_probe.py [frames [rows, e.g. 02]]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kinectpy_amd import ops  # noqa: E402
from kinectpy_amd.utils import synth  # noqa: E402

N_PX = 576 * 640


def timed(fn, reps=7, warm=2):
    for _ in range(warm):
        r = fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(reps):
        e0.record(); r = fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return float(np.median(ts)), r


def digest(res):
    pts, col, _, cnt = res
    s = [int(cnt.sum().item())]
    for t in (pts, col):
        if t is None:
            continue
        for f in (0, len(cnt) - 1):
            k = int(cnt[f].item())
            s.append(round(float(t[f][:k].double().sum().item()), 3))
    return s


F = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ROWS = sys.argv[2] if len(sys.argv) > 2 else "01234"
xy = synth.xy_table()
base_d, person = synth.render_depth(xy=xy, return_person=True)
depth = torch.as_tensor(np.tile(base_d, (F, 1))).cuda()
rgb = torch.as_tensor(np.tile(synth.mask_rgb(person), (F, 1, 1))).cuda()
xyd = torch.as_tensor(xy).cuda()
xyz = ops.unproject_u16(depth, xyd, F)
if "0" in ROWS:
    ms, res = timed(lambda: ops.depth_to_cloud(depth, xyd, None, F, False, False, sync=False))
    kept = int(res[3].sum().item())
    print(f"depth_to_cloud no colour      {ms:.4f} ms  {(F * N_PX * 2 + kept * 12) / ms / 1e6:7.0f} GB/s  kept {kept}  {digest(res)}")
if "1" in ROWS:
    ms, res = timed(lambda: ops.depth_to_cloud(depth, xyd, rgb, F, True, True, sync=False))
    kept = int(res[3].sum().item())
    print(f"depth_to_cloud mask+gate+col  {ms:.4f} ms  {(F * N_PX * 5 + kept * 24) / ms / 1e6:7.0f} GB/s  kept {kept}  {digest(res)}")
if "2" in ROWS:
    ms, res = timed(lambda: ops.depth_to_cloud(depth, xyd, rgb, F, True, False, sync=False))
    kept = int(res[3].sum().item())
    print(f"depth_to_cloud mask+col       {ms:.4f} ms  {(F * N_PX * 5 + kept * 24) / ms / 1e6:7.0f} GB/s  kept {kept}  {digest(res)}")
if "3" in ROWS:
    ms, res = timed(lambda: ops.rgbd_compact(xyz, rgb, F, True, True, want_idx=False, sync=False), reps=5)
    kept = int(res[3].sum().item())
    print(f"rgbd_compact mask+gate+col    {ms:.4f} ms  {(F * N_PX * 9 + kept * 24) / ms / 1e6:7.0f} GB/s  kept {kept}  {digest(res)}")
if "4" in ROWS:
    ms, res = timed(lambda: ops.rgbd_compact(xyz, None, F, False, False, want_idx=False, sync=False), reps=5)
    kept = int(res[3].sum().item())
    print(f"rgbd_compact plain            {ms:.4f} ms  {(F * N_PX * 6 + kept * 12) / ms / 1e6:7.0f} GB/s  kept {kept}  {digest(res)}")
