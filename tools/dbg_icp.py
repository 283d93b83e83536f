import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from kinectpy_amd import ops
from kinectpy_amd.utils import synth
base = synth.filter_cloud(200000) if hasattr(synth, "filter_cloud") else None
src, tgt, T = synth.icp_pair(20000)
s, t = torch.as_tensor(src).cuda(), torch.as_tensor(tgt).cuda()
for it in (10, 16, 30):
    r = ops.icp(s, t, 100.0, None, "p2p", None, it, want_corr=True)
    idx = r["idx"].cpu().numpy(); d2 = r["d2"].cpu().numpy()
    ok = d2 < 1e4
    print(it, r["iterations"], r["count"], int(ok.sum()), int(idx[ok].astype(np.int64).sum()), float(d2[ok].sum()))
    np.save(f"gpurun_out/dbg_{os.environ.get('KPX_NN_ENGINE','local')}_{it}.npy", np.stack([idx.astype(np.float64), d2]))
