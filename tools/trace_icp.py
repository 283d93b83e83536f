"""Development aid: run one 100k x 100k registration with KPX_ICP_TRACE=1 to see the per-iteration sweep choice."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kinectpy_amd import ops  # noqa: E402
from kinectpy_amd.utils import synth  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "p2p"
src, tgt, _ = synth.icp_pair(100_000)
s, t = torch.as_tensor(src).cuda(), torch.as_tensor(tgt).cuda()
tn = ops.estimate_normals(t, 70.0, 40) if mode == "p2plane" else None
for _ in range(2):
    r = ops.icp(s, t, 100.0, None, mode, tn, 30)
print(r["iterations"], r["fitness"])
