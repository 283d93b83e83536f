import os, sys, subprocess
import numpy as np
code = r'''
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from kinectpy_amd import ops
from kinectpy_amd.utils import synth
from oracle import oracle as O
base = synth.frame_cloud()
src, tgt, T = synth.icp_pair(12000, base)
tn = O.estimate_normals(tgt, 70.0, 40)[0].astype(np.float32)
srcs = [src, src[:7001], O.transform(src, synth.t_star())[:9000]]
inits = [np.eye(4), np.eye(4), np.linalg.inv(synth.t_star())]
for mode, nrm in (("p2p", None), ("p2plane", tn)):
    batch = ops.icp_batch(srcs, tgt, 100.0, inits, mode, nrm, 12)
    for i, (s, i0, b) in enumerate(zip(srcs, inits, batch)):
        one = ops.icp(s, tgt, 100.0, i0, mode, nrm, 12)
        solo = ops.icp_batch([s], tgt, 100.0, [i0], mode, nrm, 12)[0]
        rT, rf, _, rit = O.registration_icp(s, tgt, 100.0, i0, mode, nrm, 12)
        print(os.environ.get("KPX_ICP_CERT", "1"), os.environ.get("KPX_ICP_CHAIN", "1"), mode, i, "it", b["iterations"], solo["iterations"], one["iterations"], rit, "fit", b["fitness"] == one["fitness"], rf == b["fitness"],
              "batch-one %.3e  batch-oracle %.3e  one-oracle %.3e  solo-batch %.3e" % (np.abs(b["transformation"] - one["transformation"]).max(),
              np.abs(b["transformation"] - rT).max(), np.abs(one["transformation"] - rT).max(), np.abs(solo["transformation"] - b["transformation"]).max()))
'''
VAR = os.environ.get("DBG_VAR", "KPX_ICP_CERT")
for cert in ("1", "0"):
    r = subprocess.run([sys.executable, "-c", code], env={**os.environ, VAR: cert}, capture_output=True, text=True)
    print(r.stdout, r.stderr[-1500:] if r.returncode else "")
