import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kinectpy_amd import ops
from kinectpy_amd.utils import synth
c3 = torch.as_tensor(synth.filter_cloud(1_000_000)).cuda()
vp = ops.voxel_downsample(c3, 10.0)[0]
for _ in range(3):
    keep, stats, _ = ops.sor(vp, 20, 2.0)
torch.cuda.synchronize()
print(vp.shape, keep.shape)
