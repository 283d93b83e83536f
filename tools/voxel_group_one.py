"""one group of eight 1M-point clouds through voxel_downsample_batch for a kernel trace: python tools/voxel_group_one.py [scan|random]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kinectpy_amd import ops
from kinectpy_amd.utils import synth
h = synth.filter_cloud(1_000_000)
if len(sys.argv) > 1 and sys.argv[1] == "scan":
    h = h[np.lexsort((h[:, 0], np.floor(h[:, 1] / 5.0)))]
c3 = torch.as_tensor(h).cuda()
clouds = [(c3 + float(k)).contiguous() for k in range(8)]
cols = [torch.rand_like(c3) for _ in range(8)]
for _ in range(4):
    ops.voxel_downsample_batch(clouds, 10.0, cols)
torch.cuda.synchronize()
