import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kinectpy_amd import ops
from kinectpy_amd.utils import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
c3 = torch.as_tensor(synth.filter_cloud(1_000_000)).cuda()
clouds = [(c3 + float(k)).contiguous() for k in range(n)]
cols = [torch.rand_like(c3) for _ in range(n)]
for _ in range(6):
    ops.voxel_downsample_batch(clouds, 10.0, cols)
torch.cuda.synchronize()
