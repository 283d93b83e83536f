"""Fuzzer (not collected by pytest): the correspondence search against the oracle on extreme random clouds.
    python tests/fuzz_nn.py [cases] [seed]        (KPX_NN_ENGINE=dense fuzzes the all-pairs engine)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kinectpy_amd import ops
from oracle import oracle
oracle.build()
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
def cloud(n):
    kind = int(rng.integers(0, 8))
    scale = float(10.0 ** rng.uniform(-3, 6))
    if kind == 0: p = rng.normal(size=(n, 3))
    elif kind == 1: p = np.stack([rng.uniform(-1, 1, n), rng.uniform(-1, 1, n), rng.normal(scale=1e-3, size=n)], -1)
    elif kind == 2:
        c = rng.uniform(-5, 5, size=(rng.integers(1, 6), 3)); p = c[rng.integers(0, len(c), n)] + rng.normal(scale=3e-3, size=(n, 3))
    elif kind == 3: p = rng.integers(-20, 20, size=(n, 3)).astype(np.float64)
    elif kind == 4:
        t = rng.uniform(-3, 3, n); p = np.stack([t, 0.5 * t, np.full(n, 0.7)], -1)
    elif kind == 5: p = np.repeat(rng.normal(size=(1, 3)), n, 0)                    # all identical
    elif kind == 6: p = np.concatenate([rng.normal(size=(n - n // 2, 3)) * 1e-3, rng.normal(size=(n // 2, 3)) * 10])   # dense core + halo
    else: p = rng.uniform(-1, 1, size=(n, 3)) ** 5
    return (p * scale + rng.uniform(-3, 3, size=3) * scale).astype(np.float32), kind
bad = 0
for case in range(cases):
    n, m = int(rng.integers(1, 20000)), int(rng.integers(1, 20000))
    (src, ks), (tgt, kt) = cloud(n), cloud(m)
    if rng.random() < 0.3: tgt = (tgt.astype(np.float64) * (src.std() + 1e-9) / (tgt.std() + 1e-9)).astype(np.float32)
    T = np.eye(4); T[:3, 3] = rng.normal(size=3) * float(src.std())
    ri, rd, _ = oracle.nn(src, T, tgt, grid=(n * m > 4e7))
    gi, gd = ops.nn_search(src, tgt, T)
    gi, gd = gi.cpu().numpy(), gd.cpu().numpy()
    nb = int(((gi != ri) | (gd != rd)).sum())
    if nb:
        bad += 1
        b = np.flatnonzero((gi != ri) | (gd != rd))[0]
        print("MISMATCH case", case, "n,m", n, m, "kinds", ks, kt, "nbad", nb, "row", b, gi[b], gd[b], ri[b], rd[b], flush=True)
print("cases", cases, "mismatching", bad)
