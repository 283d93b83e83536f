import os, sys
os.environ["KPX_ICP_CHAIN_DUMP"] = "1"
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kinectpy_amd import ops
from kinectpy_amd.utils import synth
from oracle import oracle as O
base = synth.frame_cloud()
src, tgt, T = synth.icp_pair(12000, base)
batch = ops.icp_batch([src, src], tgt, 100.0, [np.eye(4), np.eye(4)], "p2p", None, 12)
print([b["iterations"] for b in batch], [b["fitness"] for b in batch])
