"""Run the native frame step over the bench's 8 synthetic frames and save every output (for A/B runs of library switches through
the environment: the files of two runs must be identical).   python tools/ab_frames.py out.npz [frames_in_flight [rounds]]
With frames in flight the 8 frames run `rounds` times through pipeline.FrameStream (one pipeline per slot)."""
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
overlap = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 1
if overlap > 1:
    from kinectpy_amd.pipeline import FrameStream  # noqa: E402
    d, c = torch.as_tensor(depth).cuda(), torch.as_tensor(rgb).cuda()
    fs = FrameStream([NativeFramePipeline(xy, 4, inits, PipelineParams()) for _ in range(overlap)])
    got = []
    for k in range(8 * rounds):
        if fs.full():
            got.append(fs.pop())
        fs.submit(d[k % 8], c[k % 8])
    while fs.pending:
        got.append(fs.pop())
    fs.close()
    for k, (p, cc, T) in enumerate(got):
        out[f"p{k}"], out[f"c{k}"], out[f"T{k}"] = p.cpu().numpy(), cc.cpu().numpy(), np.asarray(T)
    np.savez(sys.argv[1], **out)
    sys.exit(0)
for f in range(8):
    p, c, T = pipe.step(torch.as_tensor(depth[f]).cuda(), torch.as_tensor(rgb[f]).cuda())
    out[f"p{f}"], out[f"c{f}"], out[f"T{f}"] = p.cpu().numpy(), c.cpu().numpy(), np.asarray(T)
    out[f"it{f}"] = np.array([x[0] for x in pipe.last["icp"]])
    print(f, out[f"it{f}"], p.shape[0])
np.savez(sys.argv[1], **out)
