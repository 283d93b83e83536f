"""ctypes + NumPy front end of the CPU oracle (TEST INFRASTRUCTURE ONLY -- see kpx_oracle.c header).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
PARITY STATUS: parity unpinned for the Open3D-internal algorithms (no golden vectors exist in the
reference; the hull-vertex part of the oriented bounding box is pinned against Qhull through scipy); the
NumPy-only reference functions are pinned by tests/golden/ref_kat.json and ref_skeleton_fusion.json.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# KPO_BUILD_DIR=_build/asan selects the sanitizer builds (make asan; tools/asan_cpu_suite.sh)
_DIR = os.path.join(_HERE, os.environ.get("KPO_BUILD_DIR", "_build"))
_SOS = {"f32": os.path.join(_DIR, "libkpx_oracle.so"), "f64": os.path.join(_DIR, "libkpx_oracle_f64.so")}
_SO = _SOS["f32"]


def build(force=False):
    src = os.path.join(_HERE, "kpx_oracle.c")
    stale = any(not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src) for so in _SOS.values())
    if force or stale:
        target = ["asan"] if _DIR.endswith("asan") else []
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"] + target, stdout=subprocess.DEVNULL)
    return _SO


# ---- cloud storage mode ---------------------------------------------------------------------------------------------
# "f32" (default): clouds / colours / normals are float32 arrays between stages -- the storage contract of the GPU product
# (DESIGN.md 3).  "f64": they stay float64, as in the reference (utils/io.py:29-41, preprocessing/data.py:55-56: every
# Vector3dVector is float64).  Same C source compiled twice (real_t); `with storage("f64"):` switches every function below.
_STORAGE = "f32"
_libs = {}


class storage:
    def __init__(self, mode):
        if mode not in _SOS:
            raise ValueError("storage mode must be 'f32' or 'f64'")
        self.mode = mode

    def __enter__(self):
        global _STORAGE
        self.prev, _STORAGE = _STORAGE, self.mode
        return self

    def __exit__(self, *exc):
        global _STORAGE
        _STORAGE = self.prev


def storage_mode():
    return _STORAGE


def lib():
    l = _libs.get(_STORAGE)
    if l is None:
        if not os.path.exists(_SOS[_STORAGE]):
            build()
        l = C.CDLL(_SOS[_STORAGE])
        l.kpo_median_i16.restype = C.c_double
        l.kpo_rgbd_compact.restype = C.c_int64
        l.kpo_voxel_downsample.restype = C.c_int64
        l.kpo_sor.restype = C.c_int64
        l.kpo_sor_from_avg.restype = C.c_int64
        l.kpo_hull_vertices.restype = C.c_int64
        assert l.kpo_storage_bytes() == (4 if _STORAGE == "f32" else 8)
        _libs[_STORAGE] = l
    return l


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _rt():
    """dtype of stored clouds in the active storage mode"""
    return np.float32 if _STORAGE == "f32" else np.float64


def _f32(a):
    """cloud array in the active storage dtype (float32 unless `with storage("f64")`)"""
    return np.ascontiguousarray(a, dtype=_rt())


# ------------------------------------------------------------------------------------------------
def philox4x32(ctr, key):
    c = np.asarray(ctr, dtype=np.uint32)
    k = np.asarray(key, dtype=np.uint32)
    o = np.zeros(4, dtype=np.uint32)
    lib().kpo_philox4x32(_p(c), _p(k), _p(o))
    return o


def xy_table_pinhole(H, W, fx, fy, cx, cy):
    xy = np.zeros((H * W, 2), dtype=np.float32)
    lib().kpo_xy_table_pinhole(C.c_int(H), C.c_int(W), C.c_float(fx), C.c_float(fy), C.c_float(cx),
                               C.c_float(cy), _p(xy))
    return xy


def unproject_u16(depth, xy):
    depth = np.ascontiguousarray(depth, dtype=np.uint16).reshape(-1)
    xy = np.ascontiguousarray(xy, dtype=np.float32).reshape(-1, 2)          # the table is float32 in every storage mode
    out = np.zeros((depth.size, 3), dtype=np.int16)
    lib().kpo_unproject_u16(_p(depth), _p(xy), C.c_int64(depth.size), _p(out))
    return out


def median_z(xyz):
    xyz = np.ascontiguousarray(xyz, dtype=np.int16).reshape(-1, 3)
    z = np.ascontiguousarray(xyz[:, 2])
    return float(lib().kpo_median_i16(_p(z), C.c_int64(z.size), C.c_int64(1)))


def rgbd_compact(xyz, rgb=None, use_color_mask=False, use_gate=False, gate_hi=0.0):
    """a3 (+a4 when the flags are set).  Returns points f32 (K,3), colours f32 (K,3)|None, idx i32 (K)."""
    xyz = np.ascontiguousarray(xyz, dtype=np.int16).reshape(-1, 3)
    n = xyz.shape[0]
    if rgb is not None:
        rgb = np.ascontiguousarray(rgb, dtype=np.uint8).reshape(-1, 3)
    pts = np.zeros((n, 3), dtype=_rt())
    col = np.zeros((n, 3), dtype=_rt()) if rgb is not None else None
    idx = np.zeros(n, dtype=np.int32)
    k = lib().kpo_rgbd_compact(_p(xyz), _p(rgb), C.c_int64(n), C.c_int(int(use_color_mask)),
                               C.c_int(int(use_gate)), C.c_double(gate_hi), _p(pts), _p(col), _p(idx))
    return pts[:k].copy(), (col[:k].copy() if col is not None else None), idx[:k].copy()


def transform(pts, T):
    pts = _f32(pts).reshape(-1, 3)
    T = np.ascontiguousarray(T, dtype=np.float64).reshape(4, 4)
    out = np.empty_like(pts)
    lib().kpo_transform(_p(pts), C.c_int64(pts.shape[0]), _p(T), _p(out))
    return out


def rotate(nrm, T):
    nrm = _f32(nrm).reshape(-1, 3)
    T = np.ascontiguousarray(T, dtype=np.float64).reshape(4, 4)
    out = np.empty_like(nrm)
    lib().kpo_rotate(_p(nrm), C.c_int64(nrm.shape[0]), _p(T), _p(out))
    return out


def voxel_downsample(pts, voxel, col=None, nrm=None, return_counts=False):
    pts = _f32(pts).reshape(-1, 3)
    n = pts.shape[0]
    col = _f32(col).reshape(-1, 3) if col is not None else None
    nrm = _f32(nrm).reshape(-1, 3) if nrm is not None else None
    op = np.zeros((max(n, 1), 3), dtype=_rt())
    oc = np.zeros((max(n, 1), 3), dtype=_rt()) if col is not None else None
    on = np.zeros((max(n, 1), 3), dtype=_rt()) if nrm is not None else None
    cnt = np.zeros(max(n, 1), dtype=np.int32)
    m = lib().kpo_voxel_downsample(_p(pts), _p(col), _p(nrm), C.c_int64(n), C.c_double(voxel), _p(op),
                                   _p(oc), _p(on), _p(cnt))
    if m == -1:
        raise RuntimeError("voxel_size <= 0")
    if m == -2:
        raise RuntimeError("voxel_size is too small")
    res = [op[:m].copy(), oc[:m].copy() if oc is not None else None, on[:m].copy() if on is not None else None]
    if return_counts:
        res.append(cnt[:m].copy())
    return tuple(res)


def fuse_voxel_downsample(clouds, cols, Ts, voxel, return_counts=False):
    """preprocessing/data.py:44-61 in one step: cloud c moved by Ts[c] (4x4; identity for the master), stacked in order,
    voxel_down_sample(voxel) of the stack -- every decision on the fp64 value of the moved point (the reference's arrays are
    float64), only the means are stored.  cols: list of colour arrays or None.  -> points, colours | None [, counts]"""
    clouds = [_f32(p).reshape(-1, 3) for p in clouds]
    cnt = len(clouds)
    cols = [_f32(c).reshape(-1, 3) for c in cols] if cols is not None else None
    n = np.array([len(p) for p in clouds], dtype=np.int64)
    total = int(n.sum())
    T = np.ascontiguousarray(np.stack([np.asarray(t, dtype=np.float64).reshape(4, 4) for t in Ts]))
    pp = (C.c_void_p * cnt)(*[p.ctypes.data for p in clouds])
    cc = (C.c_void_p * cnt)(*[c.ctypes.data for c in cols]) if cols is not None else None
    op = np.zeros((max(total, 1), 3), dtype=_rt())
    oc = np.zeros((max(total, 1), 3), dtype=_rt()) if cols is not None else None
    ocnt = np.zeros(max(total, 1), dtype=np.int32)
    lib().kpo_fuse_voxel_downsample.restype = C.c_int64
    m = lib().kpo_fuse_voxel_downsample(C.c_int32(cnt), pp, cc, _p(n), _p(T), C.c_double(voxel), _p(op), _p(oc), _p(ocnt))
    if m == -1:
        raise RuntimeError("voxel_size <= 0")
    if m == -2:
        raise RuntimeError("voxel_size is too small")
    res = [op[:m].copy(), oc[:m].copy() if oc is not None else None]
    if return_counts:
        res.append(ocnt[:m].copy())
    return tuple(res)


def sor(pts, nb_neighbors, std_ratio, brute=False):
    """a8.  Returns keep_idx i32 (K), (mean, std, thr), avg f64 (N)."""
    pts = _f32(pts).reshape(-1, 3)
    n = pts.shape[0]
    idx = np.zeros(max(n, 1), dtype=np.int32)
    st = np.zeros(3, dtype=np.float64)
    avg = np.zeros(max(n, 1), dtype=np.float64)
    k = lib().kpo_sor(_p(pts), C.c_int64(n), C.c_int(nb_neighbors), C.c_double(std_ratio), C.c_int(int(brute)),
                      _p(idx), _p(st), _p(avg))
    if k < 0:
        raise RuntimeError("invalid nb_neighbors / std_ratio")
    return idx[:k].copy(), tuple(st.tolist()), avg[:n].copy()


def sor_from_avg(avg, std_ratio):
    """statistics + keep list of a8 from the per-point mean distances -> keep_idx, (mean, std, thr)"""
    avg = np.ascontiguousarray(avg, dtype=np.float64).reshape(-1)
    n = avg.size
    idx = np.zeros(max(n, 1), dtype=np.int32)
    st = np.zeros(3, dtype=np.float64)
    lib().kpo_sor_from_avg.restype = C.c_int64
    k = lib().kpo_sor_from_avg(_p(avg), C.c_int64(n), C.c_double(std_ratio), _p(idx), _p(st))
    return idx[:k].copy(), tuple(st.tolist())


def hybrid_knn(pts, radius, max_nn):
    pts = _f32(pts).reshape(-1, 3)
    n = pts.shape[0]
    nbr = np.full((n, max_nn), -1, dtype=np.int32)
    cnt = np.zeros(n, dtype=np.int32)
    rc = lib().kpo_hybrid_knn(_p(pts), C.c_int64(n), C.c_double(radius), C.c_int(max_nn), _p(nbr), _p(cnt))
    if rc:
        raise RuntimeError("invalid radius / max_nn")
    return nbr, cnt


def hybrid_knn_d2(pts, radius, max_nn):
    pts = _f32(pts).reshape(-1, 3)
    n = pts.shape[0]
    nbr = np.full((n, max_nn), -1, dtype=np.int32)
    cnt = np.zeros(n, dtype=np.int32)
    d2 = np.zeros((n, max_nn))
    rc = lib().kpo_hybrid_knn_d2(_p(pts), C.c_int64(n), C.c_double(radius), C.c_int(max_nn), _p(nbr), _p(cnt), _p(d2))
    if rc:
        raise RuntimeError("invalid radius / max_nn")
    return nbr, cnt, d2


def fpfh(pts, normals, radius, max_nn):
    """[O3D] compute_fpfh_feature(KDTreeSearchParamHybrid(radius, max_nn)) (registration.py:15-20).
    Returns fpfh (N,33) f64 (Open3D stores the transpose, (33,N)), spfh (N,33)."""
    pts = _f32(pts).reshape(-1, 3)
    nrm = _f32(normals).reshape(-1, 3)
    n = pts.shape[0]
    nbr, cnt, d2 = hybrid_knn_d2(pts, radius, max_nn)
    sp = np.zeros((n, 33))
    fp = np.zeros((n, 33))
    lib().kpo_fpfh(_p(pts), _p(nrm), C.c_int64(n), _p(nbr), _p(cnt), _p(d2), C.c_int(max_nn), _p(sp), _p(fp))
    return fp, sp


def feature_nn(fa, fb):
    fa = np.ascontiguousarray(fa, dtype=np.float64).reshape(-1, 33)
    fb = np.ascontiguousarray(fb, dtype=np.float64).reshape(-1, 33)
    idx = np.zeros(len(fa), dtype=np.int32)
    lib().kpo_feature_nn(_p(fa), C.c_int64(len(fa)), _p(fb), C.c_int64(len(fb)), _p(idx))
    return idx


def feature_correspondences(fs, ft, mutual_filter=True, ransac_n=3):
    """[O3D] registration_ransac_based_on_feature_matching, correspondence stage (mutual filter with the
    fall back to one-way matches when fewer than 3*ransac_n survive)."""
    ij = feature_nn(fs, ft)
    one_way = np.stack([np.arange(len(ij), dtype=np.int32), ij], 1)
    if not mutual_filter:
        return one_way
    ji = feature_nn(ft, fs)
    keep = ji[ij] == np.arange(len(ij))
    mutual = one_way[keep]
    return mutual if len(mutual) >= ransac_n * 3 else one_way


def kabsch_pairs(s, t):
    s = np.ascontiguousarray(s, dtype=np.float64).reshape(-1, 3)
    t = np.ascontiguousarray(t, dtype=np.float64).reshape(-1, 3)
    T = np.zeros(16)
    lib().kpo_kabsch_pairs(_p(s), _p(t), C.c_int(len(s)), _p(T))
    return T.reshape(4, 4)


def ransac_corres(src, tgt, corres, max_dist, ransac_n=3, edge_sim=0.95, max_iter=250000, confidence=0.999, seed=0):
    """[O3D] RegistrationRANSACBasedOnCorrespondence.  Returns T, dict(iterations, validations, fitness, rmse)."""
    src = _f32(src).reshape(-1, 3)
    tgt = _f32(tgt).reshape(-1, 3)
    corres = np.ascontiguousarray(corres, dtype=np.int32).reshape(-1, 2)
    T = np.zeros(16)
    st = np.zeros(4)
    rc = lib().kpo_ransac_corres(_p(src), C.c_int64(len(src)), _p(tgt), C.c_int64(len(tgt)), _p(corres), C.c_int64(len(corres)),
                                 C.c_double(max_dist), C.c_int(ransac_n), C.c_double(edge_sim), C.c_int(max_iter),
                                 C.c_double(confidence), C.c_uint64(seed), _p(T), _p(st))
    if rc:
        return np.eye(4), {"iterations": 0, "validations": 0, "fitness": 0.0, "rmse": 0.0}
    return T.reshape(4, 4), {"iterations": int(st[0]), "validations": int(st[1]), "fitness": float(st[2]), "rmse": float(st[3])}


def covariances(pts, nbr, cnt):
    pts = _f32(pts).reshape(-1, 3)
    n = pts.shape[0]
    cov = np.zeros((n, 6), dtype=np.float64)
    lib().kpo_covariances(_p(pts), C.c_int64(n), _p(nbr), _p(cnt), C.c_int(nbr.shape[1]), _p(cov))
    return cov


def estimate_normals(pts, radius, max_nn):
    """[O3D] estimate_normals(KDTreeSearchParamHybrid) (registration.py:9-13): eigenvector of the
    smallest eigenvalue of the neighbourhood covariance; < 3 neighbours -> (0,0,1).  Sign is
    implementation-defined: compare up to sign.  Returns normals f64 (N,3), cov (N,6), cnt."""
    nbr, cnt = hybrid_knn(pts, radius, max_nn)
    cov = covariances(pts, nbr, cnt)
    A = np.zeros((cov.shape[0], 3, 3))
    A[:, 0, 0], A[:, 0, 1], A[:, 0, 2] = cov[:, 0], cov[:, 1], cov[:, 2]
    A[:, 1, 0], A[:, 1, 1], A[:, 1, 2] = cov[:, 1], cov[:, 3], cov[:, 4]
    A[:, 2, 0], A[:, 2, 1], A[:, 2, 2] = cov[:, 2], cov[:, 4], cov[:, 5]
    w, v = np.linalg.eigh(A)
    nrm = v[:, :, 0].copy()
    nrm[cnt < 3] = [0.0, 0.0, 1.0]
    return nrm, cov, cnt


def ransac_sample(n, ransac_n, seed, h):
    ids = np.zeros(ransac_n, dtype=np.int32)
    lib().kpo_ransac_sample(C.c_int64(n), C.c_int(ransac_n), C.c_uint64(seed), C.c_uint32(h), _p(ids))
    return ids


def plane_fit(pts, ids=None):
    pts = _f32(pts).reshape(-1, 3)
    pl = np.zeros(4)
    if ids is not None:
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        lib().kpo_plane_fit(_p(pts), _p(ids), C.c_int64(ids.size), _p(pl))
    else:
        lib().kpo_plane_fit(_p(pts), None, C.c_int64(pts.shape[0]), _p(pl))
    return pl


def segment_plane(pts, distance_threshold, ransac_n, num_iterations, probability=0.99999999, seed=0,
                  return_hypotheses=False):
    """a21.  Returns plane f64 (4,), inlier idx i32 (K,) [, hyp (iters,6)]."""
    pts = _f32(pts).reshape(-1, 3)
    n = pts.shape[0]
    plane = np.zeros(4)
    idx = np.zeros(max(n, 1), dtype=np.int32)
    cnt = C.c_int64(0)
    hyp = np.zeros((num_iterations, 6)) if return_hypotheses else None
    rc = lib().kpo_segment_plane(_p(pts), C.c_int64(n), C.c_double(distance_threshold), C.c_int(ransac_n),
                                 C.c_int(num_iterations), C.c_double(probability), C.c_uint64(seed), _p(plane),
                                 _p(idx), C.byref(cnt), _p(hyp))
    if rc:
        raise RuntimeError("invalid segment_plane arguments")
    if return_hypotheses:
        return plane, idx[:cnt.value].copy(), hyp
    return plane, idx[:cnt.value].copy()


def nn(src, T, tgt, grid=False):
    """Correspondence search of one ICP iteration.  Returns idx i32 (N), d2 f64 (N), metric f64 (N)."""
    src = _f32(src).reshape(-1, 3)
    tgt = _f32(tgt).reshape(-1, 3)
    T = np.ascontiguousarray(T, dtype=np.float64).reshape(4, 4)
    n = src.shape[0]
    idx = np.zeros(n, dtype=np.int32)
    d2 = np.zeros(n)
    met = np.zeros(n)
    fn = lib().kpo_nn_grid if grid else lib().kpo_nn_brute
    rc = fn(_p(src), C.c_int64(n), _p(T), _p(tgt), C.c_int64(tgt.shape[0]), _p(idx), _p(d2), _p(met))
    if rc:
        raise RuntimeError("empty target")
    return idx, d2, met


def icp_accumulate(src, T, tgt, idx, d2, max_dist, tgt_normals=None):
    src = _f32(src).reshape(-1, 3)
    tgt = _f32(tgt).reshape(-1, 3)
    tn = _f32(tgt_normals).reshape(-1, 3) if tgt_normals is not None else None
    T = np.ascontiguousarray(T, dtype=np.float64).reshape(4, 4)
    out = np.zeros(44)
    lib().kpo_icp_accumulate(_p(src), C.c_int64(src.shape[0]), _p(T), _p(tgt), _p(tn), _p(idx), _p(d2),
                             C.c_double(max_dist), _p(out))
    return out


# ------------------------------------------------------------------------------------------------
def kabsch_from_sums(acc):
    """[O3D] TransformationEstimationPointToPoint(with_scaling=False) == Eigen::umeyama without
    scale (manual_pointcloud_registration.py:90-91): Sigma = E[t s^T] - mu_t mu_s^T,
    R = U diag(1,1,sign(det U det V)) V^T, t = mu_t - R mu_s."""
    n = acc[0]
    if n < 1:
        return np.eye(4)
    mu_s = acc[2:5] / n
    mu_t = acc[5:8] / n
    sigma = acc[8:17].reshape(3, 3) / n - np.outer(mu_t, mu_s)
    U, _, Vt = np.linalg.svd(sigma)
    S = np.eye(3)
    if np.linalg.det(U) * np.linalg.det(Vt) < 0:
        S[2, 2] = -1
    R = U @ S @ Vt
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = mu_t - R @ mu_s
    return T


def kabsch(src_pts, tgt_pts):
    """Umeyama on explicit pairs (compute_transformation with a correspondence list)."""
    s = np.asarray(src_pts, dtype=np.float64)
    t = np.asarray(tgt_pts, dtype=np.float64)
    acc = np.zeros(44)
    acc[0] = len(s)
    acc[2:5] = s.sum(0)
    acc[5:8] = t.sum(0)
    acc[8:17] = (t.T @ s).reshape(-1)
    return kabsch_from_sums(acc)


def _rot_zyx(a, b, g):
    ca, sa, cb, sb, cg, sg = np.cos(a), np.sin(a), np.cos(b), np.sin(b), np.cos(g), np.sin(g)
    Rx = np.array([[1, 0, 0], [0, ca, -sa], [0, sa, ca]])
    Ry = np.array([[cb, 0, sb], [0, 1, 0], [-sb, 0, cb]])
    Rz = np.array([[cg, -sg, 0], [sg, cg, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def p2plane_from_sums(acc):
    """[O3D] TransformationEstimationPointToPlane (registration.py:83): solve (J^T J) x = -J^T r,
    T = [Rz(x2) Ry(x1) Rx(x0) | x3..5]."""
    if acc[0] < 1:
        return np.eye(4)
    A = np.zeros((6, 6))
    q = 17
    for a in range(6):
        for b in range(a, 6):
            A[a, b] = A[b, a] = acc[q]
            q += 1
    rhs = -acc[38:44]
    try:
        x = np.linalg.solve(A, rhs)
    except np.linalg.LinAlgError:
        return np.eye(4)
    T = np.eye(4)
    T[:3, :3] = _rot_zyx(x[0], x[1], x[2])
    T[:3, 3] = x[3:6]
    return T


def registration_icp(src, tgt, max_dist, init=None, mode="p2p", tgt_normals=None, max_iteration=30,
                     relative_fitness=1e-6, relative_rmse=1e-6, grid=True, trace=None):
    """[O3D] registration_icp loop.  The transformed source is always
    T_acc . src_original in fp64 (contract AC1).  Returns T, fitness, rmse, iterations.
    trace (list) receives (T_used, idx, d2) per correspondence search."""
    src = _f32(src).reshape(-1, 3)
    tgt = _f32(tgt).reshape(-1, 3)
    T = np.eye(4) if init is None else np.array(init, dtype=np.float64).reshape(4, 4)
    n = src.shape[0]

    def search(Tc):
        idx, d2, _ = nn(src, Tc, tgt, grid=grid)
        acc = icp_accumulate(src, Tc, tgt, idx, d2, max_dist, tgt_normals if mode == "p2plane" else None)
        if trace is not None:
            trace.append((Tc.copy(), idx, d2))
        cnt = acc[0]
        fit = cnt / n if n else 0.0
        rmse = np.sqrt(acc[1] / cnt) if cnt else 0.0
        return acc, fit, rmse

    acc, fit, rmse = search(T)
    it = 0
    for it in range(1, max_iteration + 1):
        upd = kabsch_from_sums(acc) if mode == "p2p" else p2plane_from_sums(acc)
        T = upd @ T
        acc, nfit, nrmse = search(T)
        done = abs(fit - nfit) < relative_fitness and abs(rmse - nrmse) < relative_rmse
        fit, rmse = nfit, nrmse
        if done:
            break
    return T, fit, rmse, it


# ------------------------------------------------------------------------------------------------
# NumPy-only reference functions (pinned by tests/golden/ref_kat.json)
def equation_plane(p1, p2, p3):
    """floor_removal.py:21-36"""
    x1, y1, z1 = p1
    x2, y2, z2 = p2
    x3, y3, z3 = p3
    a1, b1, c1 = x2 - x1, y2 - y1, z2 - z1
    a2, b2, c2 = x3 - x1, y3 - y1, z3 - z1
    a = b1 * c2 - b2 * c1
    b = a2 * c1 - a1 * c2
    c = a1 * b2 - b1 * a2
    d = -a * x1 - b * y1 - c * z1
    return a, b, c, d


def halfspace_keep_idx(a, b, c, d, pts):
    """floor_removal.py:39-51: keeps points whose plane value is < 0 (label 0)."""
    p = np.asarray(pts, dtype=np.float64)
    val = a * p[:, 0] + b * p[:, 1] + c * p[:, 2] + d
    return np.flatnonzero(~(val >= 0)).astype(np.int32)


def kalman_filter(joint_vals, ri=10, qi=10, fi=1 / 30, hi=1):
    """preprocessing/filtering.py:98-129"""
    x = np.asarray(joint_vals, dtype=np.float64)
    Pi = np.identity(3)
    Fi, Ri, Qi, Hi = fi * np.identity(3), ri * np.identity(3), qi * np.identity(3), hi * np.identity(3)
    xh = x[0]
    out = [xh]
    for i in range(1, len(x)):
        xd = Fi @ xh
        Pd = Fi @ Pi @ Fi.T + Qi
        K = Pd @ Hi.T @ np.linalg.inv(Hi @ Pd @ Hi.T + Ri)
        xh = xd + K @ (x[i] - Hi @ xd)
        Pi = (np.identity(3) - K @ Hi) @ Pd
        out.append(xh)
    return np.array(out)


def transform_joints(vals, T):
    """utils/processing.py:357-383 / extractor.py:109-116: (F,J,3) @ inv(R) + t"""
    v = np.asarray(vals, dtype=np.float64)
    T = np.asarray(T, dtype=np.float64)
    j = v.shape[1] // 3
    return (v.reshape(v.shape[0], j, 3) @ np.linalg.inv(T[:3, :3]) + T[:3, 3]).reshape(v.shape[0], j * 3)


def floor_removal(pts, slab=200.0, thr=30.0, ransac_n=30, iters=2000, seed=0, sor_k=50, sor_ratio=0.30,
                  probability=0.99999999):
    """floor_removal.py:61-73 on a float32 cloud.  Returns (points, dict of intermediates)."""
    pts = _f32(pts).reshape(-1, 3)
    y = pts[:, 1].astype(np.float64)
    cut = y.max() - slab
    lower = np.flatnonzero(y >= cut).astype(np.int32)
    upper = np.flatnonzero(y < cut).astype(np.int32)
    floor = pts[lower]
    plane, inl = segment_plane(floor, thr, ransac_n, iters, probability, seed)
    keep = np.ones(len(floor), dtype=bool)
    keep[inl] = False
    merged = np.concatenate([floor[keep], pts[upper]], axis=0)
    kidx, stats, _ = sor(merged, sor_k, sor_ratio)
    return merged[kidx], {"lower": lower, "upper": upper, "plane": plane, "inliers": inl, "merged": merged,
                          "sor_idx": kidx, "sor_stats": stats}


def num_threads():
    return int(lib().kpo_num_threads())


# ---- sampler / normaliser --------------------------------------------------------
def sample_indices(n, k, seed):
    """seeded stand-in for np.random.choice(n, k, replace=False) (utils/processing.py:270-272)"""
    idx = np.zeros(max(int(k), 0), dtype=np.int32)
    if lib().kpo_sample_indices(C.c_int64(n), C.c_int64(k), C.c_uint64(seed), _p(idx)) != 0:
        raise ValueError("Cannot take a larger sample than population when 'replace=False'")
    return idx


def hull_vertices(pts):
    """indices (ascending) of the extreme points of the cloud (gift wrapping, arithmetic contract AC5)"""
    pts = np.ascontiguousarray(_f32(pts).reshape(-1, 3), dtype=np.float64)
    return hull_vertices_f64(pts)


def hull_vertices_f64(pts):
    pts = np.ascontiguousarray(pts, dtype=np.float64).reshape(-1, 3)
    flag = np.zeros(len(pts), dtype=np.uint8)
    lib().kpo_hull_vertices.restype = C.c_int64
    v = lib().kpo_hull_vertices(_p(pts), C.c_int64(len(pts)), _p(flag))
    if v < 0:
        raise RuntimeError("hull_vertices: degenerate cloud" if v == -1 else "hull_vertices: the wrap did not close")
    return np.flatnonzero(flag)


def obb_from_vertices(verts):
    """OrientedBoundingBox::CreateFromPoints after the hull [O3D, recalled]: cumulant mean / covariance of the hull
    vertices, eigenvectors by descending eigenvalue (third = first x second), box of R^T (v - mean).  Eigenvector signs
    are Eigen's in Open3D and unknowable here: each of the first two columns is given a positive largest component.
    -> R (3,3), center (3,), extent (3,)"""
    v = np.asarray(verts, dtype=np.float64)
    mean = v.mean(0)
    cov = (v[:, :, None] * v[:, None, :]).mean(0) - np.outer(mean, mean)
    w, V = np.linalg.eigh(cov)
    R = V[:, ::-1].copy()
    for c in (0, 1):
        j = int(np.argmax(np.abs(R[:, c])))
        if R[j, c] < 0:
            R[:, c] = -R[:, c]
    R[:, 0] /= np.linalg.norm(R[:, 0])
    R[:, 1] /= np.linalg.norm(R[:, 1])
    R[:, 2] = np.cross(R[:, 0], R[:, 1])
    q = (v - mean) @ R
    lo, hi = q.min(0), q.max(0)
    return R, R @ ((lo + hi) / 2) + mean, hi - lo


def oriented_bounding_box(pts):
    pts = _f32(pts).reshape(-1, 3)
    return obb_from_vertices(pts[hull_vertices(pts)])


def rotation_matrix_from_yxz(rot):
    """Geometry3D::GetRotationMatrixFromYXZ [O3D, recalled]: Ry(rot[0]) Rx(rot[1]) Rz(rot[2])"""
    def ax(axis, a):
        c, s = np.cos(a), np.sin(a)
        return {"x": np.array([[1, 0, 0], [0, c, -s], [0, s, c]]), "y": np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]]),
                "z": np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])}[axis]
    return ax("y", rot[0]) @ ax("x", rot[1]) @ ax("z", rot[2])


def obb_normalization_batch(x, y, obb=None, M=None):
    """utils/normalization.py:16-64 (as written there: the rotation is the constant Rx(pi), the centre is ADDED).
    M: stands in for get_rotation_matrix_from_yxz([0, pi, 0]) (the golden vectors of tests/golden/ref_norm_kat.json record
    the matrix their stub box returned)"""
    x = np.asarray(x, dtype=np.float64)
    x = x[None] if x.ndim == 2 else x
    xs, ys = [], []
    Rc = rotation_matrix_from_yxz([0, np.pi, 0]) if M is None else np.asarray(M, dtype=np.float64)
    for b in range(x.shape[0]):
        R, c, ext = obb[b] if obb is not None else oriented_bounding_box(x[b])
        L = np.max(ext)
        xs.append((np.matmul(x[b], Rc) + c) / L)
        ys.append(((np.matmul(np.asarray(y[b], dtype=np.float64).reshape(-1, 3), Rc) + c) / L).reshape(-1))
    return np.array(xs), np.array(ys)


def obb_rotation_translation_batch(x, y, obb=None):
    """utils/normalization.py:67-97: (p - c) @ R @ Rz(90 deg)"""
    x = np.asarray(x, dtype=np.float64)
    r = rotation_matrix_from_yxz([0, 0, np.pi / 2])
    xs, ys = [], []
    for b in range(x.shape[0]):
        R, c, ext = obb[b] if obb is not None else oriented_bounding_box(x[b])
        xs.append((x[b] - c) @ R @ r)
        ys.append(((np.asarray(y[b], dtype=np.float64).reshape(-1, 3) - c) @ R @ r).reshape(-1))
    return np.array(xs), np.array(ys)


def translation_normalization_batch(x, y, obb=None):
    """utils/normalization.py:100-126: p - c"""
    x = np.asarray(x, dtype=np.float64)
    xs, ys = [], []
    for b in range(x.shape[0]):
        R, c, ext = obb[b] if obb is not None else oriented_bounding_box(x[b])
        xs.append(x[b] - c)
        ys.append((np.asarray(y[b], dtype=np.float64).reshape(-1, 3) - c).reshape(-1))
    return np.array(xs), np.array(ys)


def obb_normalization(points, joints, number_of_joints, obb=None):
    """utils/processing.py:327-354: (p - centre) @ R for the points and the (number_of_joints, 3) joints"""
    points = np.asarray(points, dtype=np.float64)
    R, c, ext = obb if obb is not None else oriented_bounding_box(points)
    j = np.asarray(joints, dtype=np.float64).reshape(number_of_joints, 3)
    return (points - c) @ R, ((j - c) @ R).reshape(number_of_joints * 3)


def normalize_pointcloud(arr, min_range=-1.0, max_range=1.0):
    """utils/processing.py:313-326"""
    arr = np.asarray(arr, dtype=np.float64)
    s = (max_range - min_range) / (np.max(arr) - np.min(arr))
    return arr * s - np.min(arr) * s + min_range


def fuse_skeletons_gradient(skeletons, alpha=1.4, beta=1.4, initial_frame=20):
    """utils/skeleton_fusion.py:21-74 (pinned by tests/golden/ref_skeleton_fusion.json)"""
    sk = np.asarray(skeletons, dtype=np.float64)
    C_, F, J, _ = sk.shape
    out = np.zeros((F, J, 3))
    out[:initial_frame] = np.mean(sk[:, :initial_frame], axis=0)
    for f in range(initial_frame, F):
        for j in range(J):
            last = out[f - 1, j]
            p = [sk[c, f, j] for c in range(3)]
            cen = (p[0] + p[1] + p[2]) / 3
            w = [1.0 / ((np.linalg.norm(q - last) ** alpha) * (np.linalg.norm(q - cen) ** beta)) for q in p]
            out[f, j] = (w[0] * p[0] + w[1] * p[1] + w[2] * p[2]) / (w[0] + w[1] + w[2])
    return out


# ---- coloured ICP ([O3D] ColoredICP, recalled; parity unpinned) -------------------------------------------
def color_gradient(pts, normals, colors, radius, max_nn=30):
    """InitializePointCloudForColoredICP: per point, least squares of the neighbours' intensity differences against their
    offsets projected onto the tangent plane, plus the row (nn - 1) n = 0; fewer than 4 neighbours: zero."""
    pts = _f32(pts).reshape(-1, 3).astype(np.float64)
    nrm = _f32(normals).reshape(-1, 3).astype(np.float64)
    inten = _f32(colors).reshape(-1, 3).astype(np.float64).sum(1) / 3.0
    nbr, cnt = hybrid_knn(pts, radius, max_nn)
    grad = np.zeros_like(pts)
    for k in range(len(pts)):
        nn_ = int(cnt[k])
        if nn_ < 4:
            continue
        j = nbr[k, 1:nn_]
        q = pts[j]
        dp = (q - pts[k]) @ nrm[k]
        A = np.vstack([q - dp[:, None] * nrm[k] - pts[k], (nn_ - 1) * nrm[k]])
        b = np.append(inten[j] - inten[k], 0.0)
        M = A.T @ A
        if np.linalg.det(M) != 0:
            grad[k] = np.linalg.solve(M, A.T @ b)
    return grad


def registration_colored_icp(src, src_colors, tgt, tgt_colors, tgt_normals, max_dist, init=None, lambda_geometric=0.968,
                             max_iteration=30, relative_fitness=1e-6, relative_rmse=1e-6):
    """registration_icp loop with TransformationEstimationForColoredICP as the update -> T, fitness, rmse, iterations"""
    src = _f32(src).reshape(-1, 3)
    tgt = _f32(tgt).reshape(-1, 3)
    n = src.shape[0]
    tn = _f32(tgt_normals).reshape(-1, 3).astype(np.float64)
    Is = _f32(src_colors).reshape(-1, 3).astype(np.float64).sum(1) / 3.0
    It = _f32(tgt_colors).reshape(-1, 3).astype(np.float64).sum(1) / 3.0
    grad = color_gradient(tgt, tgt_normals, tgt_colors, 2.0 * max_dist, 30)
    slg, slp = np.sqrt(lambda_geometric), np.sqrt(1.0 - lambda_geometric)
    T = np.eye(4) if init is None else np.array(init, dtype=np.float64).reshape(4, 4)

    def search(Tc):
        idx, d2, _ = nn(src, Tc, tgt, grid=True)
        ok = d2 < max_dist * max_dist
        cnt = int(ok.sum())
        fit = cnt / n if n else 0.0
        rmse = np.sqrt(d2[ok].sum() / cnt) if cnt else 0.0
        return idx, ok, fit, rmse

    def update(Tc, idx, ok):
        if not ok.any():
            return np.eye(4)
        s = (src[ok].astype(np.float64) @ Tc[:3, :3].T) + Tc[:3, 3]
        j = idx[ok]
        t, nv, g = tgt[j].astype(np.float64), tn[j], grad[j]
        rg = ((s - t) * nv).sum(1)
        sp = s - rg[:, None] * nv
        is0 = (g * (sp - t)).sum(1) + It[j]
        gm = -(g - (g * nv).sum(1)[:, None] * nv)
        JG = slg * np.hstack([np.cross(s, nv), nv])
        JI = slp * np.hstack([np.cross(s, gm), gm])
        rG, rI = slg * rg, slp * (Is[ok] - is0)
        A = JG.T @ JG + JI.T @ JI
        rhs = -(JG.T @ rG + JI.T @ rI)
        try:
            x = np.linalg.solve(A, rhs)
        except np.linalg.LinAlgError:
            return np.eye(4)
        U = np.eye(4)
        U[:3, :3] = _rot_zyx(x[0], x[1], x[2])
        U[:3, 3] = x[3:6]
        return U

    idx, ok, fit, rmse = search(T)
    it = 0
    for it in range(1, max_iteration + 1):
        T = update(T, idx, ok) @ T
        idx, ok, nfit, nrmse = search(T)
        done = abs(fit - nfit) < relative_fitness and abs(rmse - nrmse) < relative_rmse
        fit, rmse = nfit, nrmse
        if done:
            break
    return T, fit, rmse, it


# ---- the frame step of the pipeline (preprocessing/data.py:35-61, 127-161) through the oracle -----------------------------
def pipeline_step(xy, depth, rgb, inits, P):
    """One step of kinectpy_amd.pipeline over S sensors in ONE process: depth (S, n_px) u16, rgb (S, n_px, 3) u8, inits =
    the S-1 initial transforms, P = PipelineParams-like (reg_voxel, normals_nn, icp_max_dist, icp_mode, icp_max_iteration,
    gate, filt_voxel, filt_k, filt_ratio).  Sensor 0 is the master.  -> (points, colours, [T_0 .. T_{S-1}], intermediates)"""
    S = depth.shape[0]
    full, masked = [], []
    for i in range(S):
        xyz = unproject_u16(depth[i], xy)
        full.append(rgbd_compact(xyz)[0])
        p, c, _ = rgbd_compact(xyz, rgb[i], True, True, median_z(xyz) + P.gate)
        masked.append((p, c))
    downs = [voxel_downsample(f, P.reg_voxel)[0] for f in full]
    tn = estimate_normals(downs[0], 2 * P.reg_voxel, P.normals_nn)[0].astype(_rt()) if P.icp_mode == "p2plane" else None
    Ts, icp_stats = [np.eye(4)], []
    for i in range(1, S):
        T, fit, rmse, it = registration_icp(downs[i], downs[0], P.icp_max_dist, inits[i - 1], P.icp_mode, tn, P.icp_max_iteration, grid=True)
        Ts.append(T)
        icp_stats.append((it, fit, rmse))
    # transform + vstack + voxel_down_sample (data.py:44-61) on the fp64 values of the moved points, as the reference's
    # float64 arrays carry them (fuse_voxel_downsample); the stacked cloud itself is kept only for inspection
    vp, vc = fuse_voxel_downsample([m[0] for m in masked], [m[1] for m in masked], Ts, P.filt_voxel)
    pts = np.concatenate([masked[0][0]] + [transform(masked[i][0], Ts[i]) for i in range(1, S)])
    keep, stats, _ = sor(vp, P.filt_k, P.filt_ratio)
    return vp[keep], vc[keep], Ts, {"downs": downs, "normals": tn, "icp": icp_stats, "fused": pts, "masked": masked, "voxel": vp, "keep": keep,
                                   "sor_stats": stats}
