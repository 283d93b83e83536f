"""What does float32 cloud storage (the GPU product's contract) change against the reference's float64
storage (utils/io.py:29-41, preprocessing/data.py:55-56)?  TEST INFRASTRUCTURE (oracle/): runs BASELINE configs 2, 3 and a
config-4 frame step through the oracle in both storage modes and reports the largest coordinate difference and the number
of index decisions that disagree (voxel membership, SOR keep list, slab split, plane inliers, ICP correspondences).

    python -m oracle.storage_deviation [--full] [--out profiles/r02/storage_deviation.json]
"""
import argparse
import json
import sys
import time

import numpy as np

from . import oracle as O


def _sets(a, b):
    """symmetric difference size of two index lists"""
    return int(len(np.setxor1d(np.asarray(a), np.asarray(b))))


def config2(n, base=None):
    """registration of two n-point clouds; the source is NOT float32-representable (rigidly moved + noise in float64), as a
    cloud is after any pcd.transform in the reference"""
    from kinectpy_amd.utils import synth
    if base is None:
        base = synth.frame_cloud()
    r1, r2 = np.random.default_rng(1), np.random.default_rng(2)
    n = min(n, len(base))
    tgt = base[r1.choice(len(base), n, replace=False)].astype(np.float64)
    src0 = base[r2.choice(len(base), n, replace=False)].astype(np.float64)
    Ti = np.linalg.inv(synth.t_star())
    src = src0 @ Ti[:3, :3].T + Ti[:3, 3] + r2.normal(scale=1.0, size=src0.shape)
    out = {}
    for mode in ("p2p", "p2plane"):
        res = {}
        for st in ("f32", "f64"):
            with O.storage(st):
                tn = O.estimate_normals(tgt, 70.0, 40)[0] if mode == "p2plane" else None
                trace = []
                T, fit, rmse, it = O.registration_icp(src, tgt, 100.0, None, mode, tn, 30, grid=True, trace=trace)
                res[st] = (T, fit, rmse, it, trace)
        a, b = res["f32"], res["f64"]
        k = min(len(a[4]), len(b[4]))
        corr_diff = [int((a[4][i][1] != b[4][i][1]).sum()) for i in range(k)]
        out[mode] = {"n": n, "iterations": [a[3], b[3]], "fitness_abs_diff": abs(a[1] - b[1]), "rmse_abs_diff": abs(a[2] - b[2]),
                     "T_max_abs_diff": float(np.abs(a[0] - b[0]).max()), "correspondences_differing_per_iteration_max": max(corr_diff),
                     "correspondences_differing_total": int(sum(corr_diff)), "searches_compared": k}
    return out


def config3(n):
    """the 1M-point filter chain: voxel 10 -> SOR(20, 2) -> slab -> segment_plane(30, 30, 2000, seed 7) -> SOR(50, 0.3).
    The input is rigidly moved in float64 first (a fused cloud of registered sensors), so it is not float32-representable"""
    from kinectpy_amd.utils import synth
    raw = synth.filter_cloud(n).astype(np.float64)
    a, b = np.deg2rad(1.7), np.deg2rad(-0.9)
    Ry = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
    Rz = np.array([[np.cos(b), -np.sin(b), 0], [np.sin(b), np.cos(b), 0], [0, 0, 1]])
    cloud = raw @ (Ry @ Rz).T + np.array([12.345678, -7.654321, 3.14159])
    res = {}
    for st in ("f32", "f64"):
        with O.storage(st):
            vp, _, _, cnt = O.voxel_downsample(cloud, 10.0, return_counts=True)
            keep, stats, avg = O.sor(vp, 20, 2.0)
            fl, aux = O.floor_removal(vp[keep], 200.0, 30.0, 30, 2000, 7, 50, 0.30)
            res[st] = dict(vp=vp, cnt=cnt, keep=keep, stats=stats, lower=aux["lower"], plane=aux["plane"], inl=aux["inliers"], sor2=aux["sor_idx"], out=fl)
    x, y = res["f32"], res["f64"]
    # A voxel that gains or loses a point changes the NUMBER of voxels and shifts every later index, so the two runs are
    # compared geometrically: a point of one run "is" the point of the other run within 1e-3 mm (10x the float32 spacing).
    from scipy.spatial import cKDTree

    def unmatched(p, q, tol=1e-3):
        d, _ = cKDTree(np.asarray(q, dtype=np.float64)).query(np.asarray(p, dtype=np.float64), k=1)
        return int((d > tol).sum())

    vx, vy = x["vp"].astype(np.float64), y["vp"]
    rep = {"n": n, "voxels": [int(len(vx)), int(len(vy))],
           "voxel_means_without_partner_within_1e-3mm": [unmatched(vx, vy), unmatched(vy, vx)],
           "sor1_threshold_rel_diff": abs(x["stats"][2] - y["stats"][2]) / y["stats"][2],
           "sor1_kept": [int(len(x["keep"])), int(len(y["keep"]))],
           "sor1_kept_without_partner": [unmatched(vx[x["keep"]], vy[y["keep"]]), unmatched(vy[y["keep"]], vx[x["keep"]])],
           "plane_max_abs_diff": float(np.abs(x["plane"] - y["plane"]).max()),
           "plane_inliers": [int(len(x["inl"])), int(len(y["inl"]))],
           "out_points": [int(len(x["out"])), int(len(y["out"]))],
           "out_points_without_partner": [unmatched(x["out"], y["out"]), unmatched(y["out"], x["out"])]}
    d, _ = cKDTree(vy).query(vx, k=1)
    rep["voxel_mean_max_abs_diff_mm_of_matched"] = float(d[d <= 1e-3].max())
    return rep


def config4(scale):
    """one 4-sensor frame step (extract -> register -> transform -> fuse -> voxel + SOR)"""
    from kinectpy_amd.pipeline import PipelineParams
    from kinectpy_amd.utils import synth
    xy, depth, rgb, inits, _ = synth.sensor_ring(4, 1, synth.small_xy(scale) if scale > 1 else None)
    res = {}
    for st in ("f32", "f64"):
        with O.storage(st):
            res[st] = O.pipeline_step(xy, depth[0], rgb[0], inits, PipelineParams())
    x, y = res["f32"], res["f64"]
    same = len(x[3]["voxel"]) == len(y[3]["voxel"])
    return {"pixels_per_sensor": int(depth.shape[2]), "T_max_abs_diff": float(max(np.abs(a - b).max() for a, b in zip(x[2], y[2]))),
            "icp_iterations": [[s[0] for s in x[3]["icp"]], [s[0] for s in y[3]["icp"]]],
            "fused_points": [int(len(x[3]["fused"])), int(len(y[3]["fused"]))],
            "fused_max_abs_diff_mm": float(np.abs(x[3]["fused"].astype(np.float64) - y[3]["fused"]).max()),
            "voxels": [int(len(x[3]["voxel"])), int(len(y[3]["voxel"]))],
            "voxel_mean_max_abs_diff_mm": float(np.abs(x[3]["voxel"].astype(np.float64) - y[3]["voxel"]).max()) if same else None,
            "sor_keep_differing": _sets(x[3]["keep"], y[3]["keep"]) if same else None, "out_points": [int(len(x[0])), int(len(y[0]))]}


def report(full=False):
    t0 = time.time()
    rep = {"storage_modes": "f32 = product contract (float32 clouds, fp64 decisions); f64 = the reference's float64 clouds",
           "config2": config2(100_000 if full else 20_000), "config3": config3(1_000_000 if full else 150_000),
           "config4_step": config4(1 if full else 4)}
    rep["seconds"] = round(time.time() - t0, 1)
    rep["threads"] = O.num_threads()
    return rep


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--full", action="store_true")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    r = report(a.full)
    txt = json.dumps(r, indent=1)
    print(txt)
    if a.out:
        open(a.out, "w").write(txt + "\n")
