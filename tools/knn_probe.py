"""SOR / normals on the pipeline's cloud sizes, for a rocprofv3 --pmc pass."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from kinectpy_amd import ops
from kinectpy_amd.utils import synth
c = synth.filter_cloud(400_000)
vp = ops.voxel_downsample(torch.as_tensor(c).cuda(), 35.0)[0][:30000].contiguous()
for _ in range(3):
    ops.sor(vp[:23000].contiguous(), 20, 2.0)
    ops.estimate_normals(vp, 70.0, 40)
torch.cuda.synchronize()
