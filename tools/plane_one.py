import os, sys
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from kinectpy_amd import ops
from kinectpy_amd.utils import synth
c3 = torch.as_tensor(synth.filter_cloud(1_000_000)).cuda()
for _ in range(6):
    ops.segment_plane(c3, 30.0, 30, 2000, probability=1.0, seed=7)
torch.cuda.synchronize()
