"""Fuzzer (not collected by pytest): segment_plane against the oracle on random clouds.   python tests/fuzz_plane.py [cases] [seed]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kinectpy_amd import ops
from oracle import oracle
oracle.build()
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for case in range(cases):
    n = int(rng.integers(3, 30000))
    kind = int(rng.integers(0, 5))
    if kind == 0:      # plane + noise + outliers
        p = np.stack([rng.uniform(-2000, 2000, n), rng.normal(scale=rng.uniform(0.5, 20), size=n) + 900, rng.uniform(500, 3500, n)], -1)
        p[: n // 10] = rng.uniform(-2000, 3500, size=(n // 10, 3))
    elif kind == 1:    # blob
        p = rng.normal(scale=300, size=(n, 3))
    elif kind == 2:    # two planes
        p = np.stack([rng.uniform(-1000, 1000, n), np.where(rng.random(n) < 0.5, 0.0, 400.0) + rng.normal(scale=2, size=n), rng.uniform(0, 2000, n)], -1)
    elif kind == 3:    # lattice (exact ties in distances)
        p = rng.integers(-30, 30, size=(n, 3)).astype(np.float64) * 10
    else:              # line (degenerate: every sample is collinear)
        t = rng.uniform(-3000, 3000, n); p = np.stack([t, 0.5 * t, 0.25 * t], -1)
    p = p.astype(np.float32)
    rn = int(rng.choice([3, 4, 10, 30])); rn = min(rn, n)
    thr = float(rng.choice([1.0, 5.0, 30.0, 100.0])); iters = int(rng.choice([1, 50, 500, 2000])); prob = float(rng.choice([0.9, 0.999, 0.99999999, 1.0]))
    seed = int(rng.integers(0, 1000))
    try:
        rpl, ridx = oracle.segment_plane(p, thr, rn, iters, prob, seed)
        rerr = None
    except RuntimeError as e:
        rerr = str(e)
    try:
        gpl, gidx = ops.segment_plane(p, thr, rn, iters, seed=seed, probability=prob) if False else ops.segment_plane(p, thr, rn, iters, prob, seed)
        gerr = None
    except RuntimeError as e:
        gerr = str(e)
    if (rerr is None) != (gerr is None):
        bad += 1; print("ERROR BEHAVIOUR differs", case, n, kind, rn, rerr, gerr, flush=True); continue
    if rerr is not None: continue
    gi = gidx.cpu().numpy() if hasattr(gidx, "cpu") else np.asarray(gidx)
    if not np.array_equal(gi, ridx) or np.abs(np.asarray(gpl) - np.asarray(rpl)).max() > 1e-7 * max(1.0, np.abs(rpl).max()):
        bad += 1; print("PLANE mismatch", case, n, kind, rn, thr, iters, prob, seed, len(gi), len(ridx), gpl, rpl, flush=True)
print("cases", cases, "mismatching", bad)
