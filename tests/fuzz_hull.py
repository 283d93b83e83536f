"""Fuzzer (not collected by pytest): hull vertices / oriented bounding box against the oracle (bit-exact vertex set, same
status on degenerate clouds) and, for clouds in general position, against Qhull (scipy).   python tests/fuzz_hull.py [cases] [seed]"""
import os, sys
import numpy as np, torch
from scipy.spatial import ConvexHull
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kinectpy_amd import ops
from oracle import oracle
oracle.build()
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
KINDS = ["blob", "box", "sphere", "lattice", "cylinder", "near_plane", "int_mm", "dups", "line", "plane", "tiny", "two_clusters"]
bad, qh_diff = 0, {}
for case in range(cases):
    kind = KINDS[int(rng.integers(0, len(KINDS)))]
    n = int(rng.integers(1, 6000))
    if kind == "blob":
        p = rng.normal(size=(n, 3)) * rng.uniform(0.1, 1000, 3) + rng.uniform(-3000, 3000, 3)
    elif kind == "box":
        p = rng.uniform(-1, 1, size=(n, 3)) * rng.uniform(1, 2000, 3)
    elif kind == "sphere":
        n = min(n, 1500); p = rng.normal(size=(n, 3)); p /= np.linalg.norm(p, axis=1)[:, None] + 1e-30; p *= rng.uniform(1, 3000)
    elif kind == "lattice":
        p = rng.integers(0, int(rng.integers(2, 12)), size=(n, 3)).astype(np.float64) * float(rng.choice([1, 10, 0.5]))
    elif kind == "cylinder":
        a = rng.uniform(0, 2 * np.pi, n); p = np.stack([np.cos(a) * 300, rng.uniform(-900, 900, n), np.sin(a) * 300], -1)
    elif kind == "near_plane":
        p = np.stack([rng.uniform(-2000, 2000, n), rng.normal(scale=1e-3, size=n) + 900, rng.uniform(500, 3500, n)], -1)
    elif kind == "int_mm":
        p = np.round(rng.normal(size=(n, 3)) * [300, 900, 200] + [0, 0, 2000])
    elif kind == "dups":
        m = max(1, n // 3); q = rng.normal(size=(m, 3)) * 500; p = q[rng.integers(0, m, n)]
    elif kind == "line":
        t = rng.uniform(-3000, 3000, n); p = np.stack([t, 0.5 * t, 0.25 * t], -1)
    elif kind == "plane":
        p = np.stack([rng.uniform(-2000, 2000, n), np.full(n, 900.0), rng.uniform(500, 3500, n)], -1)
    elif kind == "tiny":
        n = int(rng.integers(1, 8)); p = rng.normal(size=(n, 3)) * 100
    else:
        p = np.concatenate([rng.normal(size=(n // 2, 3)) * 50, rng.normal(size=(n - n // 2, 3)) * 50 + [5000, 0, 0]])
    f64 = bool(rng.integers(0, 2))
    p = p.astype(np.float64 if f64 else np.float32)
    n = len(p)
    try:
        ref = oracle.hull_vertices_f64(p.astype(np.float64)); rerr = None
    except RuntimeError as e:
        ref, rerr = None, str(e)
    obb, flags = ops.obb_batch(torch.as_tensor(p)[None], want_vertices=True, check=False)
    st = float(obb[0, 15])
    if rerr is not None:
        want = -1 if "degenerate" in rerr else -2
        if st != want:
            bad += 1; print("STATUS differs", case, kind, n, rerr, st, flush=True)
        continue
    got = np.flatnonzero(flags[0].cpu().numpy())
    if not np.array_equal(got, ref):
        bad += 1; print("VERTEX SET differs", case, kind, n, len(got), len(ref), flush=True); continue
    if st not in (len(ref), -3.0):
        bad += 1; print("COUNT differs", case, kind, n, st, len(ref), flush=True); continue
    if st > 0 and kind not in ("lattice", "sphere", "box"):        # well-separated principal axes: compare the box too
        R, c, ext = oracle.obb_from_vertices(p.astype(np.float64)[ref])
        row = obb[0].cpu().numpy()
        w = np.sort(np.linalg.eigvalsh(np.cov(p.astype(np.float64)[ref].T)))[::-1]
        if min(w[0] - w[1], w[1] - w[2]) > 1e-3 * w[0]:
            sc = np.abs(p).max() + 1
            if not (np.allclose(row[:9].reshape(3, 3), R, atol=1e-6) and np.allclose(row[9:12], c, atol=1e-6 * sc) and np.allclose(row[12:15], ext, rtol=1e-6, atol=1e-9 * sc)):
                bad += 1; print("BOX differs", case, kind, n, np.abs(row[:9].reshape(3, 3) - R).max(), flush=True)
    if kind in ("blob", "box", "sphere", "cylinder", "two_clusters", "lattice", "int_mm", "dups") and n >= 4:
        try:
            uniq, first = np.unique(p.astype(np.float64), axis=0, return_index=True)
            q = np.sort(first[ConvexHull(uniq).vertices])
            if not np.array_equal(q, ref):
                qh_diff[kind] = qh_diff.get(kind, 0) + 1
        except Exception:
            pass
print(f"{cases} cases, {bad} mismatches against the oracle; vertex sets that differ from Qhull by kind: {qh_diff}")
sys.exit(1 if bad else 0)
