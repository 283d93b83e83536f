"""Per-operator measurement at BASELINE.json sizes: whole-operator device time from
HIP events on the launch stream, algorithmic bytes / flops per SURVEY 8d, fraction of the roofline.
Writes one JSON line per operator (stdout); profiles/rNN/kernels.json keeps the output.

    python tools/bench_kernels.py [--quick]
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kinectpy_amd import ops  # noqa: E402
from kinectpy_amd.utils import synth  # noqa: E402

HBM_PEAK = 8000.0      # GB/s spec (6.3 TB/s measured achievable, MI355X_MICROARCH.md)
FP64_MFMA_PEAK = 78.6  # TFLOP/s vendor
N_PX = 576 * 640


def timed(fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(reps):
        e0.record()
        out = fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts)), out


def report(name, ms, nbytes=None, flops=None, **extra):
    row = {"op": name, "ms": round(ms, 4)}
    if nbytes is not None:
        gbs = nbytes / (ms * 1e-3) / 1e9
        row.update(algorithmic_bytes=int(nbytes), GBps=round(gbs, 1), frac_hbm=round(gbs / HBM_PEAK, 4))
    if flops is not None:
        tf = flops / (ms * 1e-3) / 1e12
        row.update(flops=int(flops), TFLOPs=round(tf, 2), frac_fp64_mfma=round(tf / FP64_MFMA_PEAK, 4))
    row.update(extra)
    print(json.dumps(row), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda")
    F = 64 if a.quick else 256
    xy = synth.xy_table()
    base_d, person = synth.render_depth(xy=xy, return_person=True)
    rgb1 = synth.mask_rgb(person)
    depth = torch.as_tensor(np.tile(base_d, (F, 1))).to(dev)
    rgb = torch.as_tensor(np.tile(rgb1, (F, 1, 1))).to(dev)
    xyd = torch.as_tensor(xy).to(dev)

    # ---- extract (config 1, batched so the working set exceeds the 256 MiB Infinity Cache)
    ms, xyz = timed(lambda: ops.unproject_u16(depth, xyd, F))
    report("unproject_u16 (a1)", ms, F * N_PX * 8, frames=F)
    ms, res = timed(lambda: ops.depth_to_cloud(depth, xyd, None, F, False, False, sync=False))
    kept = int(res[3].sum().item())
    report("depth_to_cloud no colour (a1+a3)", ms, F * N_PX * 2 + kept * 12, frames=F, kept=kept)
    ms, res = timed(lambda: ops.depth_to_cloud(depth, xyd, rgb, F, True, True, sync=False))
    kept = int(res[3].sum().item())
    report("depth_to_cloud mask+gate+colour (a1+a3+a4)", ms, F * N_PX * 5 + kept * 24, frames=F, kept=kept)
    ms, res = timed(lambda: ops.rgbd_compact(xyz, rgb, F, True, True, want_idx=False, sync=False))
    kept = int(res[3].sum().item())
    report("rgbd_compact from int16 XYZ (a3+a4)", ms, F * N_PX * 9 + kept * 24, frames=F, kept=kept)
    del depth, rgb, xyz, res

    # ---- container ops
    n_big = 16_000_000 if a.quick else 64_000_000
    big = torch.rand((n_big, 3), device=dev) * 3000
    T = synth.t_star()
    ms, out = timed(lambda: ops.transform(big, T, out=big))
    report("transform (a17)", ms, n_big * 24, points=n_big)
    idx = torch.randperm(n_big, device=dev)[: n_big // 2].to(torch.int32).sort().values
    # (two rows: bench.py's roofline_targets times the gather alone -- an index list the caller vouches for, `trusted=True`; the default
    # call first validates the list on the device, Open3D's SelectByIndex semantics for arbitrary lists: range check, duplicates, order --
    # which is the 0.07-of-peak figure profiles/r03/kernels.json showed beside bench.py's 0.45)
    ms, _ = timed(lambda: ops.select_by_index([big], idx, trusted=True))
    report("select_by_index gather (trusted ascending list: what bench.py times)", ms, n_big // 2 * (12 + 12 + 4), points=n_big // 2)
    ms, _ = timed(lambda: ops.select_by_index([big], idx))
    report("select_by_index with the list validated on the device (the default call)", ms, n_big // 2 * (12 + 12 + 4), points=n_big // 2)
    ms, hs = timed(lambda: ops.halfspace_select(big, [0.1, -0.9, 0.2, 300.0]))
    report("halfspace_select (a19)", ms, n_big * 12 + int(hs.shape[0]) * 4, points=n_big)
    ms, (lo_, up_) = timed(lambda: ops.slab_split(big, 200.0))
    report("slab_split max(y)-200 (a18: bbox pass + one selection pass feeding both lists)", ms, n_big * 12 + n_big * 4, points=n_big,
           lower=int(lo_.shape[0]))
    del big, idx, out

    # ---- filter chain (config 3)
    c3 = torch.as_tensor(synth.filter_cloud(1_000_000)).to(dev)
    col = torch.rand_like(c3)
    ms, (vp, vc, _) = timed(lambda: ops.voxel_downsample(c3, 10.0, col))
    report("voxel_down_sample 1M, 10 mm, colours (a7)", ms, 24 * c3.shape[0] + 24 * vp.shape[0], n=int(c3.shape[0]), m=int(vp.shape[0]))
    ops.prof_begin(64)
    ms, (keep, stats, _) = timed(lambda: ops.sor(vp, 20, 2.0), reps=3, warm=1)
    pk = ops.prof_end()["sor_knn"]
    report("remove_statistical_outlier k=20 (a8)", ms, 12 * vp.shape[0] + 16 * keep.shape[0], n=int(vp.shape[0]), kept=int(keep.shape[0]),
           knn_kernel_ms=round(pk[0] / max(pk[1], 1), 3))
    ms, (keep200, _, _) = timed(lambda: ops.sor(vp[:100000].contiguous(), 200, 3.0), reps=2, warm=1)
    report("remove_statistical_outlier k=200 on 100k (filter_outliers default)", ms, 12 * 100000 + 16 * keep200.shape[0], n=100000)
    cloud = vp[keep.long()].contiguous()
    lo, up = ops.slab_split(cloud, 200.0)
    floor = cloud[lo.long()].contiguous()
    ops.prof_begin(64)
    ms, (plane, inl) = timed(lambda: ops.segment_plane(floor, 30.0, 30, 2000, seed=7), reps=3, warm=1)
    pk = ops.prof_end()["plane_score"]
    report("segment_plane 30/30/2000 (a21)", ms, 12 * floor.shape[0] * 2 + 4 * inl.shape[0], flops=None, n=int(floor.shape[0]),
           inliers=int(inl.shape[0]), score_kernel_ms=round(pk[0] / max(pk[1], 1), 3), score_GFLOPs=round(2000 * floor.shape[0] * 8 / (pk[0] / max(pk[1], 1) * 1e-3) / 1e9, 1))
    ms, nrm = timed(lambda: ops.estimate_normals(vp[:100000].contiguous(), 70.0, 40), reps=3, warm=1)
    report("estimate_normals r=70 nn=40 on 100k (a11)", ms, 24 * 100000, n=100000)

    # ---- config 3 as a whole: filter_outliers(voxel 10, k 20, ratio 2) + floor removal (floor_removal.py:61-73), host wall time
    from kinectpy_amd.geometry import PointCloud as _PC
    from kinectpy_amd.floor_removal import remove_floor
    from kinectpy_amd.preprocessing.filtering import filter_outliers
    import time as _time

    def chain():
        pc = _PC(c3)
        f = filter_outliers(pc, 20, 2.0, 10.0)
        return remove_floor(f, seed=7)
    chain(); torch.cuda.synchronize(); t0 = _time.perf_counter()
    outc = chain(); torch.cuda.synchronize()
    report("config 3 chain: voxel 10 + SOR(20,2) + slab + segment_plane + SOR(50,0.3) on 1M points (host wall time)",
           (_time.perf_counter() - t0) * 1e3, n_in=int(c3.shape[0]), n_out=int(len(outc.points)))

    # ---- global registration (rows a11-a13) on two cluttered views, voxel 35
    xy2, ex = synth.xy_table(), synth.clutter()
    views = []
    for i, seed in ((0, 100), (1, 101)):
        dep = synth.render_depth(synth.camera_pose(i, 16), seed=seed, xy=xy2, extra=ex)
        pcl = ops.depth_to_cloud(dep, xy2, None, 1, False, False)[0][0]
        views.append(ops.voxel_downsample(pcl, 35.0)[0])
    nrm = [ops.estimate_normals(v, 70.0, 40) for v in views]
    ms, f0 = timed(lambda: ops.fpfh(views[0], nrm[0], 175.0, 40), reps=3, warm=1)
    report("compute_fpfh_feature r=175 nn=40 (a11)", ms, n=int(views[0].shape[0]))
    f1 = ops.fpfh(views[1], nrm[1], 175.0, 40)
    ms, _ = timed(lambda: ops.feature_nn(f1, f0), reps=3, warm=1)
    report("feature_nn 33-D (a13 matching; fp64 MFMA, K = 36 augmented form)", ms, flops=2.0 * 36 * f1.shape[0] * f0.shape[0], na=int(f1.shape[0]),
           nb=int(f0.shape[0]))
    corr = ops.feature_correspondences(f1, f0, True, 3)
    ms, r = timed(lambda: ops.ransac_corres(views[1], views[0], corr, 52.5, 3, 0.95, 250000, 0.999, 1), reps=2, warm=1)
    report("ransac feature matching 250k it (a13)", ms, corres=int(len(corr)), iterations=r["iterations"], validations=r["validations"],
           fitness=round(r["fitness"], 4))

    from kinectpy_amd.geometry import PointCloud
    from kinectpy_amd.preprocessing.registration import execute_global_registration
    import time
    pcs = []
    for i, seed in ((0, 100), (1, 101)):
        dep = synth.render_depth(synth.camera_pose(i, 16), seed=seed, xy=xy2, extra=ex)
        pcs.append(PointCloud(ops.depth_to_cloud(dep, xy2, None, 1, False, False)[0][0]))
    execute_global_registration(pcs[0], pcs[1], 35, 15, seed=1)
    walls = []
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        Tg = execute_global_registration(pcs[0], pcs[1], 35, 15, seed=1)
        torch.cuda.synchronize()
        walls.append((time.perf_counter() - t0) * 1e3)
    report("execute_global_registration voxel 35, 15 trials x 250k it (registration.py:32-62, host wall time, median of 3)", float(np.median(walls)),
           n_master=int(len(pcs[0].points)), n_sub=int(len(pcs[1].points)), found=Tg is not None)

    # ---- sampler / normaliser
    fused = torch.as_tensor(synth.filter_cloud(260_000)).to(dev)
    ms, _ = timed(lambda: ops.sample_points(fused, 4096, 7))
    report("select_points_randomly 260k -> 4096 (utils/processing.py:259-275)", ms, nbytes=12 * 4096 + 12 * 4096 + 4 * 4096, n=260_000)
    rng = np.random.default_rng(0)
    host = synth.filter_cloud(260_000)
    for B in (1, 32, 256):
        xb = torch.as_tensor(np.stack([host[rng.choice(len(host), 4096, replace=False)] for _ in range(B)]).astype(np.float64)).to(dev)
        ms, (obb, _) = timed(lambda: ops.obb_batch(xb, check=False))
        v = obb[:, 15].cpu().numpy()
        report(f"get_oriented_bounding_box, batch of {B} x 4096 points f64 (utils/normalization.py:38-42)", ms, clouds=B,
               ms_per_cloud=round(ms / B, 4), hull_vertices_mean=float(v.mean()), wraps_mean=float(2 * v.mean() - 4))
        if B == 256:
            yb = torch.as_tensor(rng.normal(size=(B, 32, 3))).to(dev)
            from kinectpy_amd.utils.normalization import OrientedBoundingBox
            M = OrientedBoundingBox.get_rotation_matrix_from_yxz([0, 0, np.pi / 2])
            ms, _ = timed(lambda: ops.normalize_batch(xb, obb, ops.NORM_OBB_ROT_TRANS, M))
            report("obb_rotation_translation_batch apply, 256 x 4096 points f64 (utils/normalization.py:67-97)", ms, nbytes=2 * 24 * B * 4096)
    ms, (obb, _) = timed(lambda: ops.obb_batch(fused, check=False), reps=3, warm=1)
    report("get_oriented_bounding_box, one 260k-point cloud f32", ms, hull_vertices=float(obb[0, 15]))

    # ---- registration (config 2)
    def sweep(prof):
        """the sweep kernel that ran (culled by default, dense with KPX_NN_ENGINE=dense): avg ms, TFLOP/s issued"""
        for name in ("nn_local", "nn_mfma", "nn_screen"):
            ms_k, cnt, work = prof[name]
            if cnt:
                return {"sweep_kernel": name, "sweep_kernel_ms": round(ms_k / cnt, 4), "sweep_launches": cnt,
                        "sweep_TFLOPs_issued": round(work / ms_k / 1e9, 2), "sweep_flops_per_launch": int(work / cnt)}
        return {}

    src, tgt, _ = synth.icp_pair(100_000)
    s, t = torch.as_tensor(src).to(dev), torch.as_tensor(tgt).to(dev)
    dense = 8.0 * len(src) * len(tgt)
    ops.prof_begin(256)
    ms, _ = timed(lambda: ops.nn_search(s, t, np.eye(4)), reps=5, warm=1)
    report("nn_search 100k x 100k, cold", ms, dense_equivalent_flops=int(dense), **sweep(ops.prof_end()))
    ops.prof_begin(256)
    ms, r = timed(lambda: ops.icp(s, t, 100.0, None, "p2p", None, 30), reps=5, warm=1)
    report("registration_icp p2p 100k x 100k, 30 it (a16, config 2)", ms, iterations=r["iterations"], fitness=round(r["fitness"], 5),
           ms_per_iteration=round(ms / (r["iterations"] + 1), 4), dense_equivalent_flops=int(dense * (r["iterations"] + 1)),
           **sweep(ops.prof_end()))
    tn = ops.estimate_normals(t, 70.0, 40)
    ops.prof_begin(256)
    ms, r = timed(lambda: ops.icp(s, t, 100.0, None, "p2plane", tn, 30), reps=5, warm=1)
    report("registration_icp p2plane 100k x 100k (a14)", ms, iterations=r["iterations"], fitness=round(r["fitness"], 5),
           ms_per_iteration=round(ms / (r["iterations"] + 1), 4), dense_equivalent_flops=int(dense * (r["iterations"] + 1)),
           **sweep(ops.prof_end()))


if __name__ == "__main__":
    main()
