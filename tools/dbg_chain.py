import os, sys, subprocess
code = r'''
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from kinectpy_amd import ops
from kinectpy_amd.utils import synth
from oracle import oracle as O
base = synth.frame_cloud()
src, tgt, T = synth.icp_pair(12000, base)
tn = O.estimate_normals(tgt, 70.0, 40)[0].astype(np.float32)
s2 = O.transform(src, synth.t_star())[:9000]
i2 = np.linalg.inv(synth.t_star())
cases = {"A,B,C": ([src, src[:7001], s2], [np.eye(4), np.eye(4), i2]), "C,B,A": ([s2, src[:7001], src], [i2, np.eye(4), np.eye(4)]),
         "A,A": ([src, src], [np.eye(4), np.eye(4)]), "A,C": ([src, s2], [np.eye(4), i2]), "A": ([src], [np.eye(4)])}
ref = {}
for name, (srcs, inits) in cases.items():
    for mode, nrm in (("p2p", None), ("p2plane", tn)):
        batch = ops.icp_batch(srcs, tgt, 100.0, inits, mode, nrm, 12)
        print(os.environ.get("TAG"), name, mode, "iterations", [b["iterations"] for b in batch], "fitness", [round(b["fitness"], 6) for b in batch])
'''
for tag, env in (("chain", {}), ("chain nocert", {"KPX_ICP_CERT": "0"}), ("chain nolight", {"KPX_ICP_LIGHT_SKIP": "0"}), ("launches", {"KPX_ICP_CHAIN": "0"})):
    r = subprocess.run([sys.executable, "-c", code], env={**os.environ, **env, "TAG": tag}, capture_output=True, text=True)
    print(r.stdout, r.stderr[-1500:] if r.returncode else "")
