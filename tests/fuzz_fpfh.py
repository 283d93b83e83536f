"""Fuzzer (not collected by pytest): FPFH features and 33-D matching against the oracle.   python tests/fuzz_fpfh.py [cases] [seed]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kinectpy_amd import ops
from oracle import oracle
oracle.build()
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for case in range(cases):
    n = int(rng.integers(5, 6000))
    kind = int(rng.integers(0, 4))
    if kind == 0: p = rng.normal(scale=300, size=(n, 3))
    elif kind == 1: p = np.stack([rng.uniform(-1000, 1000, n), rng.uniform(-800, 800, n), rng.normal(scale=3, size=n)], -1)
    elif kind == 2:
        c = rng.uniform(-2000, 2000, size=(4, 3)); p = c[rng.integers(0, 4, n)] + rng.normal(scale=40, size=(n, 3))
    else: p = rng.integers(-15, 15, size=(n, 3)).astype(np.float64) * 20
    p = p.astype(np.float32)
    r_n, r_f, nn = float(rng.choice([60, 150])), float(rng.choice([100, 300])), int(rng.choice([5, 40, 100]))
    nrm = ops.estimate_normals(p, r_n, 30).cpu().numpy()
    got = ops.fpfh(p, nrm, r_f, nn).cpu().numpy()
    want, _ = oracle.fpfh(p, nrm, r_f, nn)
    badrows = np.abs(got - want).max(1) > 1e-6
    if badrows.mean() > 5e-3 or not np.allclose(got[~badrows], want[~badrows], rtol=1e-8, atol=1e-8):
        bad += 1; print("FPFH mismatch", case, n, kind, r_f, nn, float(badrows.mean()), flush=True)
    m = int(rng.integers(1, 3000))
    fb = want[rng.choice(n, min(m, n), replace=False)] + (rng.random((min(m, n), 33)) < 0.02) * rng.random((min(m, n), 33))
    gi = ops.feature_nn(want, fb).cpu().numpy()
    ri = oracle.feature_nn(want, fb)
    if not np.array_equal(gi, ri):
        bad += 1; print("FEATURE NN mismatch", case, n, m, int((gi != ri).sum()), flush=True)
print("cases", cases, "mismatching", bad)
