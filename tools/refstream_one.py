"""the reference's frame loop after its first frame (KPX_ICP_FIXED, filter_outliers' defaults) through kpx_stream, 60 frames, for a kernel trace:
bash tools/ktrace.sh refstream tools/refstream_one.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kinectpy_amd.pipeline import NativeFramePipeline, NativeFrameStream, PipelineParams
from kinectpy_amd.utils import synth
xy, depth, rgb, inits, truth = synth.sensor_ring(4, 4)
d, c = torch.as_tensor(depth).cuda(), torch.as_tensor(rgb).cuda()
P = PipelineParams(icp_mode="fixed", filt_voxel=0.02, filt_k=200, filt_ratio=3.0)
fs = NativeFrameStream(NativeFramePipeline(xy, 4, [np.asarray(T) for T in truth], P), 4)
for k in range(60):
    if fs.full():
        fs.pop()
    fs.submit(d[k % 4], c[k % 4])
while fs.pending:
    fs.pop()
fs.close()
torch.cuda.synchronize()
