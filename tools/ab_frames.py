"""Run the native frame step over the bench's 8 synthetic frames and save every output (for A/B runs of library switches through
the environment: the files of two runs must be identical).   python tools/ab_frames.py out.npz"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kinectpy_amd.pipeline import NativeFramePipeline, PipelineParams  # noqa: E402
from kinectpy_amd.utils import synth  # noqa: E402

xy, depth, rgb, inits, _ = synth.sensor_ring(4, 8)
pipe = NativeFramePipeline(xy, 4, inits, PipelineParams())
out = {}
for f in range(8):
    p, c, T = pipe.step(torch.as_tensor(depth[f]).cuda(), torch.as_tensor(rgb[f]).cuda())
    out[f"p{f}"], out[f"c{f}"], out[f"T{f}"] = p.cpu().numpy(), c.cpu().numpy(), np.asarray(T)
    out[f"it{f}"] = np.array([x[0] for x in pipe.last["icp"]])
    print(f, out[f"it{f}"], p.shape[0])
np.savez(sys.argv[1], **out)
