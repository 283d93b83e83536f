"""Fuzzer (not collected by pytest): the batched entry points against one call per item and the oracle.
    python tests/fuzz_batch.py [cases] [seed]
* depth_to_cloud / rgbd_compact on random frame batches (many frames, concentrated depth values: the run-aggregated
  histogram, NaN table entries, sparse masks) against the oracle, bit for bit;
* voxel_downsample_batch (1..9 clouds, ragged sizes, empty clouds, clustered clouds with hundreds of points per voxel)
  against the oracle, bit for bit;
* icp_batch (one launch per iteration) against icp (two kernels per iteration) on both engines: identical results."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kinectpy_amd import ops
from kinectpy_amd.utils import synth
from oracle import oracle
oracle.build()
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
npy = lambda t: t.cpu().numpy()
bad = 0
checks = {"extract": 0, "voxel": 0, "icp": 0}
base = synth.filter_cloud(60000)
for case in range(cases):
    # ---- extract
    n = int(rng.choice([8, 512, 1000, 4096, 5003, 36864]))
    F = int(rng.choice([1, 2, 7, 17, 40]))
    t = rng.normal(scale=0.5, size=(n, 2)).astype(np.float32)
    t[rng.random(n) < rng.choice([0.0, 0.01, 0.3])] = np.nan
    centre = int(rng.integers(400, 5000))
    d = np.clip(rng.normal(centre, rng.choice([0.5, 3.0, 300.0]), size=(F, n)), 0, 65535).astype(np.uint16)      # a few depth bins hold everything
    d[rng.random((F, n)) < rng.uniform(0.0, 0.6)] = 0
    rgb = rng.integers(0, 255, size=(F, n, 3)).astype(np.uint8)
    rgb[rng.random((F, n)) < rng.uniform(0.0, 0.95)] = 0
    cm, dg, wi = bool(rng.integers(0, 2)), bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    use_rgb = bool(rng.integers(0, 2)) or cm
    res = ops.depth_to_cloud(d, t, rgb if use_rgb else None, F, cm, dg, want_idx=wi)
    xyz = np.stack([oracle.unproject_u16(d[f], t) for f in range(F)])
    res2 = ops.rgbd_compact(xyz, rgb if use_rgb else None, F, cm, dg, want_idx=wi)
    for f in range(F):
        rp, rc, ri = oracle.rgbd_compact(xyz[f], rgb[f] if use_rgb else None, cm, dg, oracle.median_z(xyz[f]) + 750.0)
        for which, (gp, gc, gi) in (("fused", res[f]), ("xyz", res2[f])):
            checks["extract"] += 1
            ok = np.array_equal(npy(gp), rp) and (not use_rgb or np.array_equal(npy(gc), rc)) and (not wi or np.array_equal(npy(gi), ri))
            if not ok:
                bad += 1; print("EXTRACT differs", case, which, n, F, f, cm, dg, flush=True)
    # ---- voxel batch
    k = int(rng.integers(1, 10))
    clouds = []
    for i in range(k):
        m = int(rng.choice([0, 1, 17, 3000, 20000, 50000]))
        c = base[rng.choice(len(base), m, replace=False)] if m else np.zeros((0, 3), np.float32)
        if m and rng.random() < 0.3:
            c = (c * np.float32(0.02) + np.float32(rng.normal() * 500)).astype(np.float32)               # hundreds of points per voxel
        clouds.append(np.ascontiguousarray(c + np.float32(37.0 * i)))
    v = float(rng.choice([5.0, 10.0, 35.0, 200.0]))
    with_col = bool(rng.integers(0, 2))
    cols = [rng.random(c.shape).astype(np.float32) for c in clouds] if with_col else None
    got = ops.voxel_downsample_batch(clouds, v, cols)
    for i, c in enumerate(clouds):
        if len(c) == 0:
            ok = got[i][0].shape[0] == 0
        else:
            rp, rc, _ = oracle.voxel_downsample(c, v, cols[i] if with_col else None)
            ok = np.array_equal(npy(got[i][0]), rp) and (not with_col or np.array_equal(npy(got[i][1]), rc))
        checks["voxel"] += 1
        if not ok:
            bad += 1; print("VOXEL BATCH differs", case, i, k, len(c), v, flush=True)
    # ---- icp batch against single registrations, both engines
    if case % 4 == 0:
        nt = int(rng.choice([300, 5000, 20000]))
        src, tgt, T = synth.icp_pair(nt, base)
        tn = ops.estimate_normals(tgt, 70.0, 40)
        nprob = int(rng.integers(1, 6))
        srcs = [np.ascontiguousarray(src[: int(rng.integers(1, nt + 1))]) for _ in range(nprob)]
        inits = [np.eye(4) if rng.random() < 0.5 else np.linalg.inv(synth.t_star()) for _ in range(nprob)]
        iters = int(rng.choice([0, 1, 5, 30]))
        md = float(rng.choice([1.0, 30.0, 100.0]))
        for mode, nrm in (("p2p", None), ("p2plane", tn)):
            ref = None
            for eng in ("culled", "dense"):
                ops.nn_engine(eng)
                batch = ops.icp_batch(srcs, tgt, md, inits, mode, nrm, iters)
                single = [ops.icp(s, tgt, md, i0, mode, nrm, iters) for s, i0 in zip(srcs, inits)]
                for b, o in zip(batch, single):
                    checks["icp"] += 1
                    if not (b["iterations"] == o["iterations"] and b["fitness"] == o["fitness"] and np.array_equal(b["transformation"], o["transformation"])):
                        bad += 1; print("ICP BATCH differs from single", case, mode, eng, nt, iters, md, flush=True)
                if ref is None:
                    ref = batch
                else:
                    for b, o, sarr in zip(batch, ref, srcs):
                        if min(b["fitness"], o["fitness"]) * len(sarr) < 12:
                            continue          # a handful of pairs: the 6x6 / Kabsch system is rank-deficient, its solution is rounding noise
                        if not (b["iterations"] == o["iterations"] and b["fitness"] == o["fitness"] and np.abs(b["transformation"] - o["transformation"]).max() < 1e-7):
                            bad += 1; print("ICP engines differ", case, mode, nt, iters, md, b["iterations"], o["iterations"], b["fitness"], o["fitness"],
                                            np.abs(b["transformation"] - o["transformation"]).max(), flush=True)
            ops.nn_engine("culled")
print("cases", cases, "comparisons", checks, "mismatching", bad)
sys.exit(1 if bad else 0)
