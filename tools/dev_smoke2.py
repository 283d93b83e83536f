"""dev-only: SOR / normals / plane / NN / ICP kernels vs the oracle on the GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from kinectpy_amd import ops
from kinectpy_amd.utils import synth
from oracle import oracle as O

def tm(f, *a, **k):
    torch.cuda.synchronize(); t = time.time(); r = f(*a, **k); torch.cuda.synchronize(); return r, time.time() - t

base = synth.frame_cloud()
rng = np.random.default_rng(0)
# ---- SOR
for n, k, ratio in [(5000, 20, 2.0), (60000, 20, 2.0), (20000, 50, 0.3), (20000, 200, 3.0), (17, 20, 1.0)]:
    p = base[rng.choice(len(base), n, replace=False)]
    (gi, gs, ga), dt = tm(ops.sor, p, k, ratio, want_avg=True)
    ri, rs, ra = O.sor(p, k, ratio)
    ga = ga.cpu().numpy(); gs = gs.cpu().numpy()
    print(f"sor n={n} k={k}: idx_equal={np.array_equal(gi.cpu().numpy(), ri)} kept={len(ri)} avg_relerr={np.abs(ga-ra).max()/ra.max():.2e} stats_err={np.abs(gs-np.array(rs)).max():.2e} t={dt*1e3:.1f}ms")
c3 = synth.filter_cloud(1_000_000)
(gp, _, _), dt = tm(ops.voxel_downsample, c3, 10.0); print("voxel 1M", gp.shape, f"{dt*1e3:.1f}ms")
(gi, gs, _), dt = tm(ops.sor, gp, 20, 2.0); print("sor 1M-voxel", gi.shape, f"{dt*1e3:.1f}ms")
t = time.time(); ri, rs, _ = O.sor(gp.cpu().numpy(), 20, 2.0); print("oracle sor", time.time() - t, "idx_equal", np.array_equal(gi.cpu().numpy(), ri))
# ---- normals
p = base[rng.choice(len(base), 30000, replace=False)]
gn, dt = tm(ops.estimate_normals, p, 70.0, 40)
rn, cov, cnt = O.estimate_normals(p, 70.0, 40)
gn = gn.cpu().numpy().astype(np.float64)
dots = np.abs((gn * rn).sum(1))
w = np.linalg.eigvalsh(np.stack([np.stack([cov[:,0],cov[:,1],cov[:,2]],-1), np.stack([cov[:,1],cov[:,3],cov[:,4]],-1), np.stack([cov[:,2],cov[:,4],cov[:,5]],-1)],1))
well = (cnt >= 3) & ((w[:,1]-w[:,0]) > 1e-3 * w[:,2])
print(f"normals: min|dot| well-conditioned={dots[well].min():.12f} frac_well={well.mean():.3f} few={np.sum(cnt<3)} fewmatch={np.allclose(gn[cnt<3],[0,0,1])} t={dt*1e3:.1f}ms")
# ---- plane
for n, rn_, it in [(50000, 30, 2000), (50000, 3, 500), (400000, 30, 2000)]:
    fl = c3[c3[:, 1] >= c3[:, 1].max() - 200][:n]
    (gpl, gidx), dt = tm(ops.segment_plane, fl, 30.0, rn_, it, seed=7)
    rpl, ridx = O.segment_plane(fl, 30.0, rn_, it, seed=7)
    print(f"plane n={len(fl)} rn={rn_}: idx_equal={np.array_equal(gidx.cpu().numpy(), ridx)} inl={len(ridx)} plane_err={np.abs(gpl-rpl).max():.2e} t={dt*1e3:.1f}ms")
# ---- NN + ICP
src, tgt, T = synth.icp_pair(20000, base)
(gi, gd), dt = tm(ops.nn_search, src, tgt, np.eye(4))
ri, rd, rm = O.nn(src, np.eye(4), tgt, grid=True)
print(f"nn 20k: idx_equal={np.array_equal(gi.cpu().numpy(), ri)} d2_equal={np.array_equal(gd.cpu().numpy(), rd)} t={dt*1e3:.2f}ms")
(gi, gd), dt = tm(ops.nn_search, src, tgt, np.linalg.inv(T) @ np.eye(4))
ri, rd, rm = O.nn(src, np.linalg.inv(T), tgt, grid=True)
print(f"nn 20k T: idx_equal={np.array_equal(gi.cpu().numpy(), ri)} d2_equal={np.array_equal(gd.cpu().numpy(), rd)} nmis={(gi.cpu().numpy()!=ri).sum()}")
for mode in ("p2p", "p2plane"):
    tn = O.estimate_normals(tgt, 70.0, 40)[0].astype(np.float32) if mode == "p2plane" else None
    g, dt = tm(ops.icp, src, tgt, 100.0, None, mode, tn)
    rT, rf, rr, rit = O.registration_icp(src, tgt, 100.0, None, mode, tn)
    print(f"icp {mode}: iters {g['iterations']}/{rit} fit {g['fitness']:.6f}/{rf:.6f} rmse {g['inlier_rmse']:.9f}/{rr:.9f} Terr={np.abs(g['transformation']-rT).max():.2e} t={dt*1e3:.1f}ms")
src, tgt, T = synth.icp_pair(100000, base)
(gi, gd), dt = tm(ops.nn_search, src, tgt, np.eye(4)); (gi, gd), dt = tm(ops.nn_search, src, tgt, np.eye(4))
print(f"nn 100k x 100k: {dt*1e3:.2f} ms -> {8e10/dt/1e12:.1f} TFLOP/s (fp64 MFMA peak 78.6)")
ri, rd, rm = O.nn(src, np.eye(4), tgt, grid=True)
print(f"nn 100k: idx_equal={np.array_equal(gi.cpu().numpy(), ri)} d2_equal={np.array_equal(gd.cpu().numpy(), rd)}")
g, dt = tm(ops.icp, src, tgt, 100.0); print(f"icp 100k p2p 30 it: {dt*1e3:.1f} ms", g["iterations"], g["fitness"])
c = rng.choice(len(src), 500, replace=False); corr = np.stack([c, gi.cpu().numpy()[c]], 1).astype(np.int32)
print("kabsch err", np.abs(ops.kabsch(src, tgt, corr) - O.kabsch(src[corr[:,0]], tgt[corr[:,1]])).max())
