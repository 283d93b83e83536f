"""Random batches through the one-launch ICP chain and through a launch per iteration, in one process (kpx_icp_chain): sources of 1 .. 3000
rows, 1 .. 7 registrations per batch, both estimation modes, correspondence distances from tight to loose, 0 .. 40 iterations, random
initial transforms -- every output compared bit for bit; every fifth case also against the CPU oracle (iterations, fitness, T within
1e-8) where the problem is well posed (at least 12 correspondences: below that the update step solves a singular system, and the oracle's
LDL^T and the kernel's Gauss-Jordan return different arbitrary answers).    python tools/fuzz_icp_chain.py [cases] [seed]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kinectpy_amd import ops
from kinectpy_amd.utils import synth
from oracle import oracle as O

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 120
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
base = synth.frame_cloud()
if ops.icp_chain(-1) == 0:
    sys.exit("KPX_ICP_CHAIN=0")
bad = skipped = 0
start = ops.icp_chain(-2)
for case in range(cases):
    m = int(rng.integers(50, 4000))
    tgt = np.ascontiguousarray(base[rng.choice(len(base), m, replace=False)], dtype=np.float32)
    mode = ("p2p", "p2plane")[int(rng.integers(2))]
    tn = ops.estimate_normals(torch.as_tensor(tgt).cuda(), 120.0, 30) if mode == "p2plane" else None
    cnt = int(rng.integers(1, 8))
    srcs, inits = [], []
    for _ in range(cnt):
        n = int(rng.integers(1, 3000)) if rng.random() < 0.8 else int(rng.integers(1, 70))
        T = synth.perturb(np.eye(4), float(rng.uniform(0, 3.0)), float(rng.uniform(0, 40.0)), int(rng.integers(1 << 30)))
        pick = tgt[rng.choice(m, n, replace=True)].astype(np.float64) + rng.normal(0, 2.0, (n, 3))
        srcs.append(np.ascontiguousarray(O.transform(pick.astype(np.float32), np.linalg.inv(T)), dtype=np.float32))
        inits.append(np.eye(4) if rng.random() < 0.5 else synth.perturb(np.eye(4), 1.0, 10.0, int(rng.integers(1 << 30))))
    md = float(rng.choice([20.0, 60.0, 150.0]))
    iters = int(rng.integers(0, 41))
    got = {}
    for form in (1, 0):
        ops.icp_chain(form)
        got[form] = ops.icp_batch(srcs, tgt, md, inits, mode, tn, iters)
    ops.icp_chain(1)
    for i, (a, b) in enumerate(zip(got[1], got[0])):
        same = a["iterations"] == b["iterations"] and a["fitness"] == b["fitness"] and a["inlier_rmse"] == b["inlier_rmse"] and \
            np.array_equal(a["transformation"], b["transformation"])
        if not same:
            bad += 1
            print("MISMATCH case", case, "problem", i, mode, "rows", len(srcs[i]), "targets", m, "max_dist", md, "iterations", iters, a["iterations"], b["iterations"],
                  a["fitness"], b["fitness"], np.abs(a["transformation"] - b["transformation"]).max())
    if case % 5 == 0:
        tnh = None if tn is None else tn.cpu().numpy()
        for s, i0, a in zip(srcs, inits, got[1]):
            rT, rf, _, rit = O.registration_icp(s, tgt, md, i0, mode, tnh, iters)
            if min(rf, a["fitness"]) * len(s) < 12:          # fewer correspondences than make the 6 x 6 system (or the Kabsch SVD) well posed: the
                skipped += 1                                 # update is arbitrary on both sides (different eliminations of a singular system)
                continue
            if not (rit == a["iterations"] and rf == a["fitness"] and np.abs(rT - a["transformation"]).max() < 1e-8 * max(1.0, np.abs(rT[:3, 3]).max())):
                bad += 1
                print("ORACLE case", case, mode, "rows", len(s), "targets", m, "max_dist", md, "iterations", iters, rit, a["iterations"], rf, a["fitness"],
                      np.abs(rT - a["transformation"]).max())
print("cases", cases, "chains launched", ops.icp_chain(-2) - start, "mismatches", bad, "(oracle comparisons skipped as ill-posed:", skipped, ")")
sys.exit(1 if bad else 0)
