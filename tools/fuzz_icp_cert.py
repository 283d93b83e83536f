"""Random registrations with the row certificates on (default) and off (KPX_ICP_CERT=0), child processes, same seeds: sources of 3k-30k
rows on targets of 5k-40k points (clouds large enough for the late iterations to be mostly certified), both modes, several
correspondence distances and iteration caps, batches of 1-3 -- every output compared bit for bit.    python tools/fuzz_icp_cert.py [cases] [seed]"""
import os, subprocess, sys
code = r'''
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from kinectpy_amd import ops
from kinectpy_amd.utils import synth
from oracle import oracle as O
cases, seed = int(sys.argv[2]), int(sys.argv[3])
rng = np.random.default_rng(seed)
base = synth.frame_cloud()
out = {}
for case in range(cases):
    m = int(rng.integers(5000, 40000))
    tgt = np.ascontiguousarray(base[rng.choice(len(base), m, replace=False)], dtype=np.float32)
    mode = ("p2p", "p2plane")[int(rng.integers(2))]
    tn = ops.estimate_normals(torch.as_tensor(tgt).cuda(), 120.0, 30) if mode == "p2plane" else None
    srcs, inits = [], []
    for _ in range(int(rng.integers(1, 4))):
        n = int(rng.integers(3000, 30000))
        T = synth.perturb(np.eye(4), float(rng.uniform(0.2, 3.0)), float(rng.uniform(2, 40.0)), int(rng.integers(1 << 30)))
        pick = base[rng.choice(len(base), n, replace=False)].astype(np.float64) + rng.normal(0, 1.5, (n, 3))
        srcs.append(np.ascontiguousarray(O.transform(pick.astype(np.float32), np.linalg.inv(T)), dtype=np.float32))
        inits.append(np.eye(4))
    md = float(rng.choice([30.0, 60.0, 100.0]))
    iters = int(rng.integers(8, 45))
    r = ops.icp_batch(srcs, tgt, md, inits, mode, tn, iters)
    out[f"T{case}"] = np.stack([x["transformation"] for x in r])
    out[f"s{case}"] = np.array([[x["fitness"], x["inlier_rmse"], x["iterations"], x["count"]] for x in r])
np.savez(sys.argv[1], **out)
print("iterations", [int(v[:, 2].max()) for k, v in out.items() if k.startswith("s")])
'''
cases = sys.argv[1] if len(sys.argv) > 1 else "24"
seed = sys.argv[2] if len(sys.argv) > 2 else "5"
files = []
for cert in ("1", "0"):
    f = f"/tmp/fuzz_icp_cert_{cert}.npz"
    r = subprocess.run([sys.executable, "-c", code, f, cases, seed], env={**os.environ, "KPX_ICP_CERT": cert, "KPX_ICP_CHAIN_LOCK": "0"}, capture_output=True, text=True)
    print("KPX_ICP_CERT=" + cert, r.stdout.strip()[-400:], r.stderr[-1500:] if r.returncode else "")
    if r.returncode:
        sys.exit(1)
    files.append(f)
import numpy as np
a, b = np.load(files[0]), np.load(files[1])
bad = [k for k in a.files if not np.array_equal(a[k], b[k])]
print("cases", cases, "differences:", bad)
sys.exit(1 if bad else 0)
