"""The MFMA kernels once each, for a `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES ...` pass (profiles/rNN/pmc_mfma.csv):
feature matching (fp64 MFMA, K = 36), the all-pairs correspondence sweeps of the dense engine (fp64 + fp32 screening) and the
culled sweep."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kinectpy_amd import ops  # noqa: E402
from kinectpy_amd.utils import synth  # noqa: E402

dev = torch.device("cuda")
rng = np.random.default_rng(0)
fa = torch.as_tensor(rng.random((25000, 33))).to(dev)
fb = torch.as_tensor(rng.random((27000, 33))).to(dev)
for _ in range(3):
    ops.feature_nn(fa, fb)
src, tgt, _ = synth.icp_pair(100_000)
s, t = torch.as_tensor(src).to(dev), torch.as_tensor(tgt).to(dev)
for eng in ("dense", "dense_fp64", "culled"):      # dense_fp64: every iteration on the fp64 sweep (its warm, chunked form after the first)
    ops.nn_engine(eng)
    ops.icp(s, t, 100.0, None, "p2p", None, 12 if eng == "dense_fp64" else 6)
ops.nn_engine("culled")
# the batch form of the culled sweep (one launch per iteration for all registrations: what the frame pipeline runs)
subs = [s[: 30000 + 500 * i].contiguous() for i in range(3)]
tn = ops.estimate_normals(t[:31000].contiguous(), 70.0, 40)
ops.icp_batch(subs, t[:31000].contiguous(), 100.0, [np.eye(4)] * 3, "p2plane", tn, 6)
torch.cuda.synchronize()
