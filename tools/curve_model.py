"""CPU-only model of the culled search's per-wave work on the bench's clouds (numpy + scipy, no GPU): how many column tiles and
groups a wave of 16 rows reaches under the exact per-row box test, for rows / columns ordered along the Z-curve or the Hilbert
curve, and what a sphere test would let through instead.  Its Z-curve figures reproduce the kernel's own statistics
(tools/icp_probe.py --waves: tiles per wave p90 9-10, p99 13; groups kept p90 5-6, p99 8-9), which is what makes the other columns
worth reading.  DESIGN.md section 5 (round 3) quotes it.        python tools/curve_model.py"""
import os
import sys

import numpy as np
from scipy.spatial import cKDTree

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kinectpy_amd.utils import synth  # noqa: E402

xy, depth, rgb, inits, truth = synth.sensor_ring(4, 1)
xyt = xy.reshape(-1, 2)


def cloud(s):
    d = depth[0, s].astype(np.float64)
    valid = (d > 0) & np.isfinite(xyt).all(1)
    return np.stack([xyt[:, 0] * d, xyt[:, 1] * d, d], 1)[valid]


def voxel(pts, v=35.0):
    """voxel_down_sample(35 mm): mean point and integer voxel index of every occupied voxel"""
    org = pts.min(0)
    idx = np.floor((pts - org) / v).astype(np.int64)
    key = (idx[:, 0] << 40) | (idx[:, 1] << 20) | idx[:, 2]
    u, inv = np.unique(key, return_inverse=True)
    cnt = np.bincount(inv)
    mean = np.stack([np.bincount(inv, pts[:, a]) / cnt for a in range(3)], 1)
    return mean, np.stack([(u >> 40), (u >> 20) & 0xFFFFF, u & 0xFFFFF], 1)


def zkey(vi, bits=10):
    k = np.zeros(len(vi), np.int64)
    o = 0
    for q in range(bits):
        for a in range(3):
            k |= ((vi[:, a] >> q) & 1) << o
            o += 1
    return k


def hkey(vi, bits=10):
    """Skilling's axes -> transpose, as kpx_morton.h hilbert30 / kpx_voxel.hip voxel_hcode"""
    X = vi.astype(np.int64).copy()
    M = 1 << (bits - 1)
    Q = M
    while Q > 1:
        P = Q - 1
        for i in range(3):
            m = (X[:, i] & Q) != 0
            X[m, 0] ^= P
            t = (X[~m, 0] ^ X[~m, i]) & P
            X[~m, 0] ^= t
            X[~m, i] ^= t
        Q >>= 1
    for i in range(1, 3):
        X[:, i] ^= X[:, i - 1]
    t = np.zeros(len(X), np.int64)
    Q = M
    while Q > 1:
        t ^= np.where((X[:, 2] & Q) != 0, Q - 1, 0)
        Q >>= 1
    for i in range(3):
        X[:, i] ^= t
    k = np.zeros(len(X), np.int64)
    for b in range(bits - 1, -1, -1):
        for i in range(3):
            k = (k << 1) | ((X[:, i] >> b) & 1)
    return k


def pct(a, q):
    return float(np.percentile(a, q))


def evaluate(name, keyf):
    tm, tv = voxel(cloud(0))
    sm, sv = voxel(cloud(1))
    tgt = tm[np.argsort(keyf(tv), kind="stable")]
    src = sm[np.argsort(keyf(sv), kind="stable")]
    T = np.asarray(truth[0]).reshape(4, 4)
    src = src @ T[:3, :3].T + T[:3, 3]
    d, _ = cKDTree(tgt).query(src)
    r = np.minimum(d, 100.0) * 1.0001 + 1e-3           # a late iteration's bound: the partner's distance, at most max_dist
    nt = len(tgt) // 16
    tiles = tgt[: nt * 16].reshape(nt, 16, 3)
    lo, hi = tiles.min(1), tiles.max(1)
    cb = (lo + hi) / 2
    rb = np.linalg.norm(hi - cb, axis=1)
    ng = nt // 16
    glo, ghi = lo[: ng * 16].reshape(ng, 16, 3).min(1), hi[: ng * 16].reshape(ng, 16, 3).max(1)
    tl, gl, ex, sp = [], [], [], []
    for w in range(len(src) // 16):
        P, R = src[w * 16:(w + 1) * 16], r[w * 16:(w + 1) * 16]
        ex.append((P.max(0) - P.min(0)).max())
        g = np.maximum(0, np.maximum(lo[None] - P[:, None], P[:, None] - hi[None]))
        tl.append(((g ** 2).sum(2) <= (R ** 2)[:, None]).any(0).sum())
        g = np.maximum(0, np.maximum(glo[None] - P[:, None], P[:, None] - ghi[None]))
        gl.append(((g ** 2).sum(2) <= (R ** 2)[:, None]).any(0).sum())
        sp.append((np.linalg.norm(P[:, None] - cb[None], axis=2) <= R[:, None] + rb[None]).any(0).sum())
    tl, gl, ex, sp = map(np.array, (tl, gl, ex, sp))
    print(f"{name}: tiles per wave  mean {tl.mean():.2f}  p90 {pct(tl, 90):.0f}  p99 {pct(tl, 99):.0f}  max {tl.max()};  "
          f"groups per wave  mean {gl.mean():.2f}  p90 {pct(gl, 90):.0f}  p99 {pct(gl, 99):.0f}  max {gl.max()}")
    print(f"{' ' * len(name)}  extent of a wave's 16 rows  p50 {pct(ex, 50):.0f}  p90 {pct(ex, 90):.0f}  p99 {pct(ex, 99):.0f} mm;  "
          f"tile half-diagonal  p50 {np.median(rb):.0f}  p90 {pct(rb, 90):.0f} mm;  "
          f"sphere test instead of the box test: tiles per wave  mean {sp.mean():.2f}  p90 {pct(sp, 90):.0f}  p99 {pct(sp, 99):.0f}  max {sp.max()}")


evaluate("Z-curve", zkey)
evaluate("Hilbert", hkey)
