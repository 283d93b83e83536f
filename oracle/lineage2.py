"""Second lineage of the Open3D-internal algorithms on the hot path -- TEST INFRASTRUCTURE ONLY (like everything under oracle/).

NumPy / SciPy restatements written from SURVEY.md Appendix A and section 3.4 (and the published Open3D algorithms they recall),
WITHOUT following oracle/kpx_oracle.c: different search structure (scipy.spatial.cKDTree instead of the oracle's grid), different
linear algebra (numpy.linalg.svd / solve / eigh instead of the oracle's Jacobi / LDL^T), vectorised scoring instead of the oracle's
sequential fma chains.  tests/test_oracle_cpu.py compares the two lineages on the same seeded inputs: where they agree, an
implementation slip in either would have had to be made twice, independently.  What this cannot do is pin Open3D itself (absent
here, unpinned in the reference): both lineages restate the same recalled algorithms -- "two independent restatements agree;
Open3D itself unpinned" (DESIGN.md section 2).

Each function cites the reference call site it stands behind:
  segment_plane        floor_removal.py:70                      [O3D] PointCloud.segment_plane
  registration_icp     preprocessing/registration.py:78-84, manual_pointcloud_registration.py:96-98   [O3D] registration_icp
  estimate_normals     preprocessing/registration.py:9-13       [O3D] estimate_normals(KDTreeSearchParamHybrid)
  compute_fpfh         preprocessing/registration.py:15-20      [O3D] compute_fpfh_feature
  feature_correspondences   preprocessing/registration.py:50-57 [O3D] registration_ransac_based_on_feature_matching (matching step)
  evaluate_registration   preprocessing/registration.py:50-61   [O3D] the scoring of a candidate transform (fitness, inlier_rmse)
  voxel_down_sample    preprocessing/filtering.py:23, registration.py:8,100-101   [O3D] PointCloud.voxel_down_sample
  remove_statistical_outlier   preprocessing/filtering.py:24, floor_removal.py:73   [O3D] PointCloud.remove_statistical_outlier
Clouds arrive as float32 (the storage contract of DESIGN.md section 3); all arithmetic is float64 on the promoted values.
"""
import numpy as np
from scipy.spatial import cKDTree


# ------------------------------------------------------------------------------------------------ plane segmentation
def plane_from_points(pts):
    """least-squares plane through >= 3 points, Appendix A's determinant method -> (a, b, c, d) or zeros for a degenerate sample"""
    p = np.asarray(pts, dtype=np.float64)
    c = p.mean(axis=0)
    r = p - c
    xx, xy, xz = (r[:, 0] * r[:, 0]).sum(), (r[:, 0] * r[:, 1]).sum(), (r[:, 0] * r[:, 2]).sum()
    yy, yz, zz = (r[:, 1] * r[:, 1]).sum(), (r[:, 1] * r[:, 2]).sum(), (r[:, 2] * r[:, 2]).sum()
    det_x, det_y, det_z = yy * zz - yz * yz, xx * zz - xz * xz, xx * yy - xy * xy
    if det_x > det_y and det_x > det_z:
        n = np.array([det_x, xz * yz - xy * zz, xy * yz - xz * yy])
    elif det_y > det_z:
        n = np.array([xz * yz - xy * zz, det_y, xy * xz - yz * xx])
    else:
        n = np.array([xy * yz - xz * yy, xy * xz - yz * xx, det_z])
    nn = np.linalg.norm(n)
    if nn == 0.0:
        return np.zeros(4)
    n = n / nn
    return np.array([n[0], n[1], n[2], -float(n @ c)])


def plane_from_triangle(p0, p1, p2):
    n = np.cross(p1 - p0, p2 - p0)
    nn = np.linalg.norm(n)
    if nn == 0.0:
        return np.zeros(4)
    n = n / nn
    return np.array([n[0], n[1], n[2], -float(n @ p0)])


def segment_plane(pts, distance_threshold, ransac_n, num_iterations, probability, sampler):
    """sampler(h) -> the ransac_n point indices of iteration h (the reference is unseeded; the tests hand over the oracle's Philox
    samples).  -> (plane refitted to the final inliers, ascending inlier indices, iterations evaluated)"""
    p = np.asarray(pts, dtype=np.float64).reshape(-1, 3)
    n = p.shape[0]
    best_fit, best_rmse, best_plane = 0.0, 0.0, np.zeros(4)
    break_iteration = float("inf")
    count = 0
    for h in range(num_iterations):
        if count > break_iteration:
            continue
        ids = np.asarray(sampler(h))
        plane = plane_from_triangle(p[ids[0]], p[ids[1]], p[ids[2]]) if ransac_n == 3 else plane_from_points(p[ids])
        if not plane.any():
            continue
        dist = np.abs(p @ plane[:3] + plane[3])
        inl = dist < distance_threshold
        k = int(inl.sum())
        fit = k / n if k else 0.0
        rmse = float(dist[inl].sum()) / np.sqrt(k) if k else 0.0          # Open3D's "rmse": sum of |d| over sqrt(#inliers)
        if fit > best_fit or (fit == best_fit and rmse < best_rmse):
            best_fit, best_rmse, best_plane = fit, rmse, plane
            if fit < 1.0:
                with np.errstate(divide="ignore"):
                    lim = np.log(1.0 - probability) / np.log(1.0 - fit ** ransac_n) if probability < 1.0 else float("inf")
                break_iteration = float(int(min(lim, float(num_iterations)))) if np.isfinite(lim) else float(num_iterations)
            else:
                break_iteration = 0.0
        count += 1
    dist = np.abs(p @ best_plane[:3] + best_plane[3])
    inliers = np.nonzero(dist < distance_threshold)[0].astype(np.int32)
    refit = plane_from_points(p[inliers]) if inliers.size >= 3 else best_plane
    return refit, inliers, count


# ------------------------------------------------------------------------------------------------ registration_icp
def _rot_zyx(a, b, g):
    ca, sa, cb, sb, cg, sg = np.cos(a), np.sin(a), np.cos(b), np.sin(b), np.cos(g), np.sin(g)
    Rx = np.array([[1, 0, 0], [0, ca, -sa], [0, sa, ca]])
    Ry = np.array([[cb, 0, sb], [0, 1, 0], [-sb, 0, cb]])
    Rz = np.array([[cg, -sg, 0], [sg, cg, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def umeyama_no_scale(s, t):
    """Appendix A: P2P update from matched pairs (rows of s -> rows of t)"""
    mu_s, mu_t = s.mean(axis=0), t.mean(axis=0)
    sigma = (t - mu_t).T @ (s - mu_s) / s.shape[0]
    U, D, Vt = np.linalg.svd(sigma)
    S = np.eye(3)
    if np.linalg.det(U) * np.linalg.det(Vt) < 0:
        S[2, 2] = -1.0
    R = U @ S @ Vt
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = mu_t - R @ mu_s
    return T


def point_to_plane_update(s, t, nt):
    r = ((s - t) * nt).sum(axis=1)
    J = np.hstack([np.cross(s, nt), nt])
    x = np.linalg.solve(J.T @ J, -(J.T @ r))
    T = np.eye(4)
    T[:3, :3] = _rot_zyx(x[0], x[1], x[2])
    T[:3, 3] = x[3:]
    return T


def registration_icp(src, tgt, max_dist, init=None, mode="p2p", tgt_normals=None, max_iteration=30, relative_fitness=1e-6,
                     relative_rmse=1e-6):
    """SURVEY 3.4 -> (T, fitness, inlier_rmse, iterations run)"""
    s0 = np.asarray(src, dtype=np.float64).reshape(-1, 3)
    t = np.asarray(tgt, dtype=np.float64).reshape(-1, 3)
    nt = None if tgt_normals is None else np.asarray(tgt_normals, dtype=np.float64).reshape(-1, 3)
    T = np.eye(4) if init is None else np.array(init, dtype=np.float64).reshape(4, 4)
    tree = cKDTree(t)

    def correspond(Tc):
        p = s0 @ Tc[:3, :3].T + Tc[:3, 3]
        _, j = tree.query(p, k=1)
        d2 = ((p - t[j]) ** 2).sum(axis=1)
        ok = d2 < max_dist * max_dist
        k = int(ok.sum())
        return p, j, ok, (k / s0.shape[0] if s0.shape[0] else 0.0), (np.sqrt(d2[ok].sum() / k) if k else 0.0)

    p, j, ok, fit, rmse = correspond(T)
    it = 0
    for it in range(1, max_iteration + 1):
        if ok.any():
            upd = umeyama_no_scale(p[ok], t[j[ok]]) if mode == "p2p" else point_to_plane_update(p[ok], t[j[ok]], nt[j[ok]])
        else:
            upd = np.eye(4)
        T = upd @ T
        p, j, ok, nfit, nrmse = correspond(T)
        done = abs(fit - nfit) < relative_fitness and abs(rmse - nrmse) < relative_rmse
        fit, rmse = nfit, nrmse
        if done:
            break
    return T, fit, rmse, it


# ------------------------------------------------------------------------------------------------ normals, FPFH, matching
def hybrid_neighbours(pts, radius, max_nn):
    """KDTreeSearchParamHybrid: the max_nn nearest, of those the ones with d^2 < radius^2 (the point itself first, d = 0).
    -> (idx (N, max_nn) padded with -1, count (N,), d2 (N, max_nn))"""
    p = np.asarray(pts, dtype=np.float64).reshape(-1, 3)
    n = p.shape[0]
    k = min(max_nn, n)
    kq = min(n, max_nn + 16)                       # a few more than asked for: on exactly equal distances (integer-millimetre data) the
    _, idx = cKDTree(p).query(p, k=kq)             # tree's choice at the cut is unspecified; the rule is (d2, index) ascending
    idx = idx.reshape(n, kq)
    d2 = ((p[:, None, :] - p[idx]) ** 2).sum(axis=2)
    order = np.lexsort((idx, d2), axis=1)[:, :k]
    idx, d2 = np.take_along_axis(idx, order, 1), np.take_along_axis(d2, order, 1)
    keep = d2 < radius * radius
    cnt = keep.sum(axis=1).astype(np.int32)
    out = np.full((n, max_nn), -1, dtype=np.int32)
    od2 = np.zeros((n, max_nn))
    out[:, :k] = np.where(keep, idx, -1)
    od2[:, :k] = np.where(keep, d2, 0.0)
    return out, cnt, od2


def estimate_normals(pts, radius, max_nn):
    """-> (normals (N,3) up to sign, covariance (N,3,3), neighbour count); fewer than 3 neighbours: (0, 0, 1)"""
    p = np.asarray(pts, dtype=np.float64).reshape(-1, 3)
    idx, cnt, _ = hybrid_neighbours(p, radius, max_nn)
    n = p.shape[0]
    cov = np.zeros((n, 3, 3))
    nrm = np.tile(np.array([0.0, 0.0, 1.0]), (n, 1))
    for i in range(n):
        k = int(cnt[i])
        if k < 3:
            continue
        q = p[idx[i, :k]]
        m = q.mean(axis=0)
        cov[i] = (q.T @ q) / k - np.outer(m, m)          # E[x x^T] - E[x] E[x]^T (Open3D's cumulant form)
        w, v = np.linalg.eigh(cov[i])
        nrm[i] = v[:, 0]
    return nrm, cov, cnt


def _pair_features(p1, n1, p2, n2):
    d = p2 - p1
    dist = np.linalg.norm(d)
    if dist == 0.0:
        return np.zeros(4)
    a1, a2 = float(n1 @ d) / dist, float(n2 @ d) / dist
    if np.arccos(min(1.0, abs(a1))) > np.arccos(min(1.0, abs(a2))):
        n1, n2, d, f2 = n2, n1, -d, -a2
    else:
        f2 = a1
    v = np.cross(d, n1)
    vn = np.linalg.norm(v)
    if vn == 0.0:
        return np.zeros(4)
    v = v / vn
    w = np.cross(n1, v)
    return np.array([np.arctan2(float(w @ n2), float(n1 @ n2)), float(v @ n2), f2, dist])


def compute_fpfh(pts, normals, radius, max_nn):
    """-> (fpfh (N,33), spfh (N,33)); Open3D stores the transposes"""
    p = np.asarray(pts, dtype=np.float64).reshape(-1, 3)
    nr = np.asarray(normals, dtype=np.float64).reshape(-1, 3)
    n = p.shape[0]
    idx, cnt, d2 = hybrid_neighbours(p, radius, max_nn)
    spfh = np.zeros((n, 33))
    for i in range(n):
        k = int(cnt[i])
        if k <= 1:
            continue
        inc = 100.0 / (k - 1)
        for j in idx[i, 1:k]:
            f = _pair_features(p[i], nr[i], p[j], nr[j])
            h = int(np.floor(11 * (f[0] + np.pi) / (2.0 * np.pi)))
            spfh[i, min(max(h, 0), 10)] += inc
            h = int(np.floor(11 * (f[1] + 1.0) * 0.5))
            spfh[i, 11 + min(max(h, 0), 10)] += inc
            h = int(np.floor(11 * (f[2] + 1.0) * 0.5))
            spfh[i, 22 + min(max(h, 0), 10)] += inc
    fpfh = np.zeros((n, 33))
    for i in range(n):
        k = int(cnt[i])
        if k <= 1:
            continue
        acc = np.zeros(33)
        for j, dd in zip(idx[i, 1:k], d2[i, 1:k]):
            if dd == 0.0:
                continue
            acc += spfh[j] / dd
        for b in range(3):
            sm = acc[11 * b:11 * b + 11].sum()
            if sm != 0.0:
                acc[11 * b:11 * b + 11] *= 100.0 / sm
        fpfh[i] = acc + spfh[i]
    return fpfh, spfh


def feature_correspondences(fs, ft, mutual_filter=True, ransac_n=3):
    """nearest target feature of every source feature (and the reverse for the mutual filter; fewer than 3 x ransac_n mutual pairs --
    Open3D: "empirically mutual correspondence set should not be too small" -- : the one-way list).
    -> (K,2) int32 pairs (source, target), and the distance gap of every one-way choice to its runner-up (ties are ambiguous)"""
    fs, ft = np.asarray(fs, dtype=np.float64), np.asarray(ft, dtype=np.float64)
    d_st, j = cKDTree(ft).query(fs, k=min(2, len(ft)))
    d_ts, i = cKDTree(fs).query(ft, k=min(2, len(fs)))
    j1 = j[:, 0] if j.ndim == 2 else j
    i1 = i[:, 0] if i.ndim == 2 else i
    one_way = np.stack([np.arange(len(fs)), j1], axis=1).astype(np.int32)
    gap_st = (d_st[:, 1] - d_st[:, 0]) if j.ndim == 2 else np.full(len(fs), np.inf)
    gap_ts = (d_ts[:, 1] - d_ts[:, 0]) if i.ndim == 2 else np.full(len(ft), np.inf)
    if not mutual_filter:
        return one_way, gap_st, gap_ts
    mutual = one_way[i1[j1] == np.arange(len(fs))]
    return (mutual if len(mutual) >= 3 * ransac_n else one_way), gap_st, gap_ts


# ------------------------------------------------------------------------------------------------ the filter pair (a7, a8)
def voxel_down_sample(pts, voxel_size, colours=None):
    """[O3D] PointCloud.voxel_down_sample (filtering.py:23, registration.py:8, 100-101), SURVEY Appendix A: origin = min_bound - v / 2,
    index = floor((p - origin) / v) per axis, per voxel the MEAN of its points (and colours), accumulated in float64 in ascending
    point order; output in ascending (ix, iy, iz) (the build's documented order).  -> points f32 (M, 3), colours f32 (M, 3) | None,
    counts (M,)"""
    if not voxel_size > 0.0:
        raise RuntimeError("voxel_size <= 0")
    p = np.asarray(pts, dtype=np.float32).reshape(-1, 3).astype(np.float64)
    origin = p.min(axis=0) - 0.5 * voxel_size
    idx = np.floor((p - origin) / voxel_size).astype(np.int64)
    uniq, inv = np.unique(idx, axis=0, return_inverse=True)            # rows sorted lexicographically = ascending (ix, iy, iz)
    inv = inv.reshape(-1)
    sums = np.zeros((len(uniq), 3), dtype=np.float64)
    np.add.at(sums, inv, p)                                            # unbuffered: one add after the other, in point order
    cnt = np.bincount(inv, minlength=len(uniq))
    out_c = None
    if colours is not None:
        c = np.asarray(colours, dtype=np.float32).reshape(-1, 3).astype(np.float64)
        cs = np.zeros((len(uniq), 3), dtype=np.float64)
        np.add.at(cs, inv, c)
        out_c = (cs / cnt[:, None]).astype(np.float32)
    return (sums / cnt[:, None]).astype(np.float32), out_c, cnt


def remove_statistical_outlier(pts, nb_neighbors, std_ratio):
    """[O3D] PointCloud.remove_statistical_outlier (filtering.py:24, floor_removal.py:73), SURVEY row a8: k nearest neighbours with
    the point itself among them (distance 0), avg_i = mean of the distances over the min(k, N) neighbours returned, mu and the
    Bessel-corrected sigma over the points with avg > 0... here: over all points that have neighbours, keep avg_i > 0 and
    avg_i < mu + r sigma, indices ascending.  -> keep indices, (mu, sigma, threshold), avg"""
    if nb_neighbors < 1 or not std_ratio > 0.0:
        raise RuntimeError("invalid nb_neighbors / std_ratio")
    p = np.asarray(pts, dtype=np.float32).reshape(-1, 3).astype(np.float64)
    n = len(p)
    k = min(int(nb_neighbors), n)
    d, _ = cKDTree(p).query(p, k=k)
    d = d.reshape(n, k)
    avg = d.sum(axis=1) / k
    valid = np.ones(n, dtype=bool)                                     # every point has itself as a neighbour
    mu = avg[valid].sum() / valid.sum()
    sigma = np.sqrt(((avg[valid] - mu) ** 2).sum() / (valid.sum() - 1)) if valid.sum() > 1 else 0.0
    thr = mu + std_ratio * sigma
    keep = np.nonzero((avg > 0.0) & (avg < thr))[0]
    return keep.astype(np.int64), (mu, sigma, thr), avg


def fuse_voxel_down_sample(clouds, colours, transforms, voxel_size):
    """data.py:44-61: every sensor's cloud moved by its 4x4 (`pcd.transform(T)`: float64 points), the moved clouds stacked in order
    (`np.vstack`), `voxel_down_sample` of the stack.  The stack is float64 in the reference, so the voxel index and the means are taken
    from the float64 values of the moved points (never rounded to float32 in between).  -> points f32, colours f32 | None, counts"""
    moved = []
    for p, T in zip(clouds, transforms):
        p = np.asarray(p, dtype=np.float32).reshape(-1, 3).astype(np.float64)
        T = np.asarray(T, dtype=np.float64).reshape(4, 4)
        moved.append(p @ T[:3, :3].T + T[:3, 3])
    q = np.vstack(moved)
    origin = q.min(axis=0) - 0.5 * voxel_size
    idx = np.floor((q - origin) / voxel_size).astype(np.int64)
    uniq, inv = np.unique(idx, axis=0, return_inverse=True)
    inv = inv.reshape(-1)
    cnt = np.bincount(inv, minlength=len(uniq))
    sums = np.zeros((len(uniq), 3), dtype=np.float64)
    np.add.at(sums, inv, q)
    out_c = None
    if colours is not None:
        c = np.vstack([np.asarray(x, dtype=np.float32).reshape(-1, 3).astype(np.float64) for x in colours])
        cs = np.zeros((len(uniq), 3), dtype=np.float64)
        np.add.at(cs, inv, c)
        out_c = (cs / cnt[:, None]).astype(np.float32)
    return (sums / cnt[:, None]).astype(np.float32), out_c, cnt


def evaluate_registration(src, tgt, max_dist, T):
    """[O3D] evaluate_registration / the scoring step of registration_ransac_based_on_feature_matching (registration.py:50-61 keeps the
    trial of the highest `fitness`): every source point moved by T, its nearest target point from a k-d tree, inliers = distance below
    max_dist.  -> fitness (inliers / source points), inlier_rmse, inlier count"""
    s = np.asarray(src, dtype=np.float32).reshape(-1, 3).astype(np.float64)
    t = np.asarray(tgt, dtype=np.float32).reshape(-1, 3).astype(np.float64)
    T = np.asarray(T, dtype=np.float64).reshape(4, 4)
    p = s @ T[:3, :3].T + T[:3, 3]
    d, _ = cKDTree(t).query(p, k=1)
    ok = d < max_dist
    k = int(ok.sum())
    return (k / len(s) if len(s) else 0.0), (float(np.sqrt((d[ok] ** 2).sum() / k)) if k else 0.0), k
