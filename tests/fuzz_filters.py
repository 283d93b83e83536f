"""Fuzzer (not collected by pytest): voxel filter, statistical outlier removal and normals against the oracle on extreme
random clouds.    python tests/fuzz_filters.py [cases] [seed]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kinectpy_amd import ops
from oracle import oracle
oracle.build()
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
def cloud(n):
    kind = int(rng.integers(0, 8))
    scale = float(10.0 ** rng.uniform(-2, 4))
    if kind == 0: p = rng.normal(size=(n, 3))
    elif kind == 1: p = np.stack([rng.uniform(-1, 1, n), rng.uniform(-1, 1, n), rng.normal(scale=1e-3, size=n)], -1)
    elif kind == 2:
        c = rng.uniform(-5, 5, size=(rng.integers(1, 6), 3)); p = c[rng.integers(0, len(c), n)] + rng.normal(scale=3e-3, size=(n, 3))
        p[: max(1, n // 40)] = rng.uniform(-8, 8, size=(max(1, n // 40), 3))
    elif kind == 3: p = rng.integers(-12, 12, size=(n, 3)).astype(np.float64)
    elif kind == 4:
        t = rng.uniform(-3, 3, n); p = np.stack([t, 0.5 * t, np.full(n, 0.7)], -1)
    elif kind == 5: p = np.repeat(rng.normal(size=(1, 3)), n, 0)
    elif kind == 6: p = np.concatenate([rng.normal(size=(n - n // 2, 3)) * 1e-3, rng.normal(size=(n // 2, 3)) * 10])
    else: p = rng.uniform(-1, 1, size=(n, 3)) ** 5
    return (p * scale + rng.uniform(-3, 3, size=3) * scale).astype(np.float32), kind, scale
bad = 0
for case in range(cases):
    n = int(rng.integers(1, 40000))
    p, kind, scale = cloud(n)
    ext = float(np.ptp(p, axis=0).max()) + 1e-6
    voxel = ext / float(rng.choice([3, 30, 300, 3000]))
    try:
        rp = oracle.voxel_downsample(p, voxel)[0]
        gp = ops.voxel_downsample(p, voxel)[0].cpu().numpy()
        if not np.array_equal(gp, rp): bad += 1; print("VOXEL mismatch", case, n, kind, voxel, flush=True)
    except RuntimeError as e:
        pass
    k, ratio = int(rng.choice([1, 8, 20, 64, 200])), float(rng.choice([0.3, 1.0, 2.5]))
    gi, gs, ga = ops.sor(p, k, ratio, want_avg=True)
    ri, rs, ra = oracle.sor(p, k, ratio)
    ga = ga.cpu().numpy()
    if not np.allclose(ga, ra, rtol=1e-11, atol=0):
        bad += 1; w = np.argmax(np.abs(ga - ra) / np.maximum(np.abs(ra), 1e-300)); print("SOR avg mismatch", case, n, kind, k, w, ga[w], ra[w], flush=True)
    radius, nn = ext / float(rng.choice([5, 50, 500])), int(rng.choice([3, 10, 40, 128]))
    gn = ops.estimate_normals(p, radius, nn).cpu().numpy().astype(np.float64)
    rn, cov, cnt = oracle.estimate_normals(p, radius, nn)
    A = np.zeros((len(p), 3, 3))
    A[:, 0, 0], A[:, 1, 1], A[:, 2, 2] = cov[:, 0], cov[:, 3], cov[:, 5]
    A[:, 0, 1] = A[:, 1, 0] = cov[:, 1]; A[:, 0, 2] = A[:, 2, 0] = cov[:, 2]; A[:, 1, 2] = A[:, 2, 1] = cov[:, 4]
    w = np.linalg.eigvalsh(A)
    well = (cnt >= 3) & ((w[:, 1] - w[:, 0]) > 1e-2 * np.maximum(w[:, 2], 1e-30)) & (w[:, 2] > 1e-12 * scale * scale)
    dots = np.abs((gn * rn).sum(1))
    if not np.allclose(gn[cnt < 3], [0, 0, 1]) or (well.any() and (dots[well] < 1 - 1e-4).any()):
        bad += 1; print("NORMALS mismatch", case, n, kind, radius, nn, int((dots[well] < 1 - 1e-4).sum()), flush=True)
print("cases", cases, "mismatching", bad)
