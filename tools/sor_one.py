"""one remove_statistical_outlier call per case for a kernel trace (tools/ktrace.sh tools/sor_one.py sor): config 3's cloud after
voxel_down_sample(10) with (20, 2.0), a fused 4-sensor cloud with filter_outliers' defaults (200, 3.0)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kinectpy_amd import ops
from kinectpy_amd.utils import synth
dev = torch.device("cuda")
v1 = ops.voxel_downsample(torch.as_tensor(synth.filter_cloud(1_000_000)).to(dev), 10.0)[0]
fv = ops.voxel_downsample(torch.as_tensor(synth.frame_cloud()).to(dev), 10.0)[0]
which = sys.argv[1] if len(sys.argv) > 1 else "20"
for _ in range(5):
    if which == "20":
        ops.sor(v1, 20, 2.0)
    else:
        ops.sor(fv, 200, 3.0)
torch.cuda.synchronize()
