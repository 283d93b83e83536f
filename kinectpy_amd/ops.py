"""Tensor-level operators: one thin function per C-ABI entry point of libkinectpx.so.

Inputs/outputs are torch-ROCm tensors (device memory containers).  Functions that produce a
data-dependent number of rows read the device count (one host sync) and return trimmed views.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib as L

GATE_MM = 750.0          # preprocessing/data.py:170-171
FLAG_COLOR_MASK = 1
FLAG_DEPTH_GATE = 2


def _dev(x, dtype):
    if isinstance(x, torch.Tensor):
        return x.to(device=L.device(), dtype=dtype).contiguous()
    return torch.as_tensor(np.ascontiguousarray(x), dtype=dtype).to(L.device()).contiguous()


def _count(t):
    """Device count word(s) -> Python ints; negative values are kpx_status codes raised by kernels."""
    v = t.cpu().tolist()
    for c in v:
        if c < 0:
            raise L.KinectPxError("voxel_size is too small" if c == -3 else f"kpx device error {c}")
    return v


def _T(T):
    T = np.ascontiguousarray(np.asarray(T, dtype=np.float64).reshape(4, 4))
    return T


# ---- extract --------------------------------------------------------------------------------------
def unproject_u16(depth, xy_table, frames=1):
    """a1.  depth u16 [frames*n_px], xy f32 [n_px*2] -> int16 [frames, n_px, 3]."""
    lib = L.load()
    depth = _dev(depth, torch.uint16).reshape(-1)
    xy = _dev(xy_table, torch.float32).reshape(-1)
    n_px = xy.numel() // 2
    assert depth.numel() == frames * n_px
    out = torch.empty((frames, n_px, 3), dtype=torch.int16, device=depth.device)
    L.check(lib.kpx_unproject_u16(L.ptr(depth), L.ptr(xy), n_px, frames, L.ptr(out), L.stream_ptr()))
    return out


def median_i16(v, n, stride, frames=1):
    lib = L.load()
    out = torch.empty(frames, dtype=torch.float64, device=v.device)
    ws, wsz = L.workspace(lib.kpx_median_workspace_bytes(frames))
    L.check(lib.kpx_median_i16(L.ptr(v), n, stride, frames, L.ptr(out), ws, wsz, L.stream_ptr()))
    return out


def rgbd_compact(xyz, rgb=None, frames=1, color_mask=False, depth_gate=False, gate=GATE_MM, want_idx=True, sync=True):
    """a3 (+a4).  xyz int16 [frames, n, 3]; rgb u8 [frames, n, 3] or None.
    Returns per-frame lists (points f32 (K,3), colours f32 (K,3)|None, idx i32 (K)|None) or, with sync=False, the padded buffers
    and the device count tensor (no host round trip)."""
    lib = L.load()
    xyz = _dev(xyz, torch.int16).reshape(frames, -1, 3)
    n = xyz.shape[1]
    rgb_t = _dev(rgb, torch.uint8).reshape(frames, n, 3) if rgb is not None else None
    dev = xyz.device
    flags = (FLAG_COLOR_MASK if (color_mask and rgb is not None) else 0) | (FLAG_DEPTH_GATE if depth_gate else 0)
    med = median_i16(xyz.reshape(-1)[2:], n, 3, frames) if depth_gate else None   # z column, frame stride 3n
    pts = torch.empty((frames, n, 3), dtype=torch.float32, device=dev)
    col = torch.empty((frames, n, 3), dtype=torch.float32, device=dev) if rgb_t is not None else None
    idx = torch.empty((frames, n), dtype=torch.int32, device=dev) if want_idx else None
    cnt = torch.empty(frames, dtype=torch.int32, device=dev)
    ws, wsz = L.workspace(lib.kpx_compact_workspace_bytes(n, frames))
    L.check(lib.kpx_rgbd_compact(L.ptr(xyz), L.ptr(rgb_t), n, frames, flags, L.ptr(med), float(gate), L.ptr(pts),
                                 L.ptr(col), L.ptr(idx), L.ptr(cnt), ws, wsz, L.stream_ptr()))
    if not sync:
        return pts, col, idx, cnt
    ks = _count(cnt)
    return [(pts[f, :k], col[f, :k] if col is not None else None, idx[f, :k] if idx is not None else None)
            for f, k in enumerate(ks)]


def depth_to_cloud(depth, xy_table, rgb=None, frames=1, color_mask=False, depth_gate=False, gate=GATE_MM,
                   want_idx=False, sync=True):
    """Fused a1+a3+a4.  Returns per-frame (points, colours|None, idx|None) or, with sync=False,
    the padded buffers and the device count tensor."""
    lib = L.load()
    depth = _dev(depth, torch.uint16).reshape(frames, -1)
    n = depth.shape[1]
    xy = _dev(xy_table, torch.float32).reshape(-1)
    rgb_t = _dev(rgb, torch.uint8).reshape(frames, n, 3) if rgb is not None else None
    dev = depth.device
    flags = (FLAG_COLOR_MASK if (color_mask and rgb is not None) else 0) | (FLAG_DEPTH_GATE if depth_gate else 0)
    pts = torch.empty((frames, n, 3), dtype=torch.float32, device=dev)
    col = torch.empty((frames, n, 3), dtype=torch.float32, device=dev) if rgb_t is not None else None
    idx = torch.empty((frames, n), dtype=torch.int32, device=dev) if want_idx else None
    cnt = torch.empty(frames, dtype=torch.int32, device=dev)
    ws, wsz = L.workspace(lib.kpx_depth_to_cloud_workspace_bytes(n, frames))
    L.check(lib.kpx_depth_to_cloud(L.ptr(depth), L.ptr(xy), L.ptr(rgb_t), n, frames, flags, float(gate), L.ptr(pts),
                                   L.ptr(col), L.ptr(idx), L.ptr(cnt), ws, wsz, L.stream_ptr()))
    if not sync:
        return pts, col, idx, cnt
    ks = _count(cnt)
    return [(pts[f, :k], col[f, :k] if col is not None else None, idx[f, :k] if idx is not None else None)
            for f, k in enumerate(ks)]


# ---- container ops ----------------------------------------------------------------------------------
def transform(pts, T, out=None):
    lib = L.load()
    pts = _dev(pts, torch.float32).reshape(-1, 3)
    out = torch.empty_like(pts) if out is None else out
    T = _T(T)
    L.check(lib.kpx_transform(L.ptr(pts), pts.shape[0], L.hptr(T), L.ptr(out), L.stream_ptr()))
    return out


def rotate(nrm, T, out=None):
    lib = L.load()
    nrm = _dev(nrm, torch.float32).reshape(-1, 3)
    out = torch.empty_like(nrm) if out is None else out
    T = _T(T)
    L.check(lib.kpx_rotate(L.ptr(nrm), nrm.shape[0], L.hptr(T), L.ptr(out), L.stream_ptr()))
    return out


def joints_affine(x, A, t):
    """a5: x (rows,3) f64 @ A (3,3) + t"""
    lib = L.load()
    x = _dev(x, torch.float64).reshape(-1, 3)
    out = torch.empty_like(x)
    A = np.ascontiguousarray(A, dtype=np.float64).reshape(3, 3)
    t = np.ascontiguousarray(t, dtype=np.float64).reshape(3)
    L.check(lib.kpx_joints_affine_f64(L.ptr(x), x.shape[0], L.hptr(A), L.hptr(t), L.ptr(out), L.stream_ptr()))
    return out


def bounds(pts):
    """min x, y, z, max x, y, z of an (n,3) f32 cloud as a float64[6] device tensor (Open3D get_min_bound / get_max_bound;
    floor_removal.py:65-66 takes max(y))."""
    lib = L.load()
    pts = _dev(pts, torch.float32).reshape(-1, 3)
    if pts.shape[0] == 0:
        raise L.KinectPxError("bounds: empty cloud")
    out = torch.empty(6, dtype=torch.float64, device=pts.device)
    ws, wsz = L.workspace(lib.kpx_bounds_workspace_bytes())
    L.check(lib.kpx_bounds(L.ptr(pts), pts.shape[0], L.ptr(out), ws, wsz, L.stream_ptr()))
    return out


def select_by_index(attrs, idx, invert=False, trusted=False, want_bounds=False):
    """attrs: list of up to three (n,3) f32 tensors (None allowed).  Returns list of selected tensors.
    [O3D] SelectByIndex has mask semantics: the result is in ascending original order without duplicates whatever the
    order of idx.  trusted=True is for index lists that come straight from another operator of this library (the keep
    list of sor, the inliers of segment_plane: ascending, duplicate-free, in range): it skips the range check (two
    reductions and a host sync) and selects with one gather."""
    lib = L.load()
    attrs = list(attrs) + [None] * (3 - len(attrs))
    ref = next(a for a in attrs if a is not None)
    n = ref.shape[0]
    idx = _dev(idx, torch.int32).reshape(-1)
    k = idx.numel()
    if k and n and not trusted:
        lo, hi = int(idx.min()), int(idx.max())
        if lo < 0 or hi >= n:
            raise L.KinectPxError("select_by_index: index out of range")
    # trusted lists are ascending and duplicate-free by construction: a plain gather equals Open3D's mask selection
    mode = 1 if invert else (0 if trusted else 2)
    if want_bounds:
        # the gather that also leaves the selected cloud's bounds (attrs[0] = the points); returns (outs, bounds or None)
        if mode or k == 0 or attrs[0] is None:
            return select_by_index(attrs, idx, invert, trusted), None
        outs = [torch.empty((k, 3), dtype=torch.float32, device=ref.device) if a is not None else None for a in attrs]
        bb = torch.empty(6, dtype=torch.float64, device=ref.device)
        ws, wsz = L.workspace(lib.kpx_select_workspace_bytes(n))
        L.check(lib.kpx_select_by_index_bounds(L.ptr(attrs[0]), L.ptr(attrs[1]), L.ptr(attrs[2]), n, L.ptr(idx), k,
                                               L.ptr(outs[0]), L.ptr(outs[1]), L.ptr(outs[2]), L.ptr(bb), ws, wsz, L.stream_ptr()))
        return outs, bb
    m = n if mode else k
    outs = [torch.empty((m, 3), dtype=torch.float32, device=ref.device) if a is not None else None for a in attrs]
    cnt = torch.empty(1, dtype=torch.int32, device=ref.device) if mode else None        # written by the compaction's scan
    ws, wsz = L.workspace(lib.kpx_select_workspace_bytes(n))
    L.check(lib.kpx_select_by_index(L.ptr(attrs[0]), L.ptr(attrs[1]), L.ptr(attrs[2]), n, L.ptr(idx), k, mode,
                                    L.ptr(outs[0]), L.ptr(outs[1]), L.ptr(outs[2]), L.ptr(cnt), ws, wsz,
                                    L.stream_ptr()))
    if mode:
        m = _count(cnt)[0]
        outs = [o[:m] if o is not None else None for o in outs]
    return outs


def halfspace_select(pts, plane):
    lib = L.load()
    pts = _dev(pts, torch.float32).reshape(-1, 3)
    n = pts.shape[0]
    idx = torch.empty(max(n, 1), dtype=torch.int32, device=pts.device)
    cnt = torch.zeros(1, dtype=torch.int32, device=pts.device)
    pl = np.ascontiguousarray(plane, dtype=np.float64).reshape(4)
    ws, wsz = L.workspace(lib.kpx_select_workspace_bytes(n))
    L.check(lib.kpx_halfspace_select(L.ptr(pts), n, L.hptr(pl), L.ptr(idx), L.ptr(cnt), ws, wsz, L.stream_ptr()))
    return idx[:_count(cnt)[0]]


def slab_split(pts, slab, bounds=None):
    """bounds: the cloud's float64[6] device bounds when they are known (ops.bounds, select_by_index(want_bounds=True)):
    max(y) is read from them instead of from a pass over the points."""
    lib = L.load()
    pts = _dev(pts, torch.float32).reshape(-1, 3)
    n = pts.shape[0]
    lo = torch.empty(n, dtype=torch.int32, device=pts.device)
    up = torch.empty(n, dtype=torch.int32, device=pts.device)
    cnt = torch.zeros(2, dtype=torch.int32, device=pts.device)
    ws, wsz = L.workspace(lib.kpx_select_workspace_bytes(n))
    if bounds is not None:
        if bounds.dtype != torch.float64 or bounds.numel() != 6 or bounds.device != pts.device:
            raise L.KinectPxError("slab_split: bounds must be a float64[6] tensor on the cloud's device")
        L.check(lib.kpx_slab_split_bounded(L.ptr(pts), n, float(slab), L.ptr(bounds), L.ptr(lo), C.c_void_p(cnt.data_ptr()), L.ptr(up),
                                           C.c_void_p(cnt.data_ptr() + 4), ws, wsz, L.stream_ptr()))
    else:
        L.check(lib.kpx_slab_split(L.ptr(pts), n, float(slab), L.ptr(lo), C.c_void_p(cnt.data_ptr()), L.ptr(up),
                                   C.c_void_p(cnt.data_ptr() + 4), ws, wsz, L.stream_ptr()))
    c = _count(cnt)
    return lo[:c[0]], up[:c[1]]


# ---- filters ----------------------------------------------------------------------------------------
def voxel_downsample(pts, voxel, col=None, nrm=None):
    lib = L.load()
    pts = _dev(pts, torch.float32).reshape(-1, 3)
    n = pts.shape[0]
    col = _dev(col, torch.float32).reshape(-1, 3) if col is not None else None
    nrm = _dev(nrm, torch.float32).reshape(-1, 3) if nrm is not None else None
    dev = pts.device
    op = torch.empty((max(n, 1), 3), dtype=torch.float32, device=dev)
    oc = torch.empty_like(op) if col is not None else None
    on = torch.empty_like(op) if nrm is not None else None
    cnt = torch.empty(1, dtype=torch.int32, device=dev)                                  # written by the library (0 for an empty cloud)
    ws, wsz = L.workspace(lib.kpx_voxel_workspace_bytes(n))
    L.check(lib.kpx_voxel_downsample(L.ptr(pts), L.ptr(col), L.ptr(nrm), n, float(voxel), L.ptr(op), L.ptr(oc),
                                     L.ptr(on), L.ptr(cnt), ws, wsz, L.stream_ptr()))
    m = _count(cnt)[0]
    return op[:m], (oc[:m] if oc is not None else None), (on[:m] if on is not None else None)


def voxel_downsample_batch(clouds, voxel, cols=None):
    """voxel_down_sample of several independent clouds in one call (processed side by side on the library's internal
    lanes, one read-back for all counts).  Returns a list of (points, colours | None)."""
    lib = L.load()
    clouds = [_dev(p, torch.float32).reshape(-1, 3) for p in clouds]
    cnt = len(clouds)
    if cnt == 0:
        return []
    dev = clouds[0].device
    cols = [_dev(c, torch.float32).reshape(-1, 3) for c in cols] if cols is not None else None
    outs = [torch.empty((max(p.shape[0], 1), 3), dtype=torch.float32, device=dev) for p in clouds]
    ocols = [torch.empty_like(o) for o in outs] if cols is not None else None
    n_arr = np.array([p.shape[0] for p in clouds], dtype=np.int64)
    arr = lambda ts: C.cast((C.c_void_p * cnt)(*[t.data_ptr() for t in ts]), C.c_void_p) if ts is not None else None
    counts = torch.empty(cnt, dtype=torch.int32, device=dev)
    ws, wsz = L.workspace(lib.kpx_voxel_batch_workspace_bytes(cnt, n_arr.ctypes.data_as(C.c_void_p)))
    L.check(lib.kpx_voxel_downsample_batch(cnt, arr(clouds), arr(cols), n_arr.ctypes.data_as(C.c_void_p), float(voxel), arr(outs),
                                           arr(ocols), L.ptr(counts), ws, wsz, L.stream_ptr()))
    ms = _count(counts)
    return [(outs[i][: ms[i]], ocols[i][: ms[i]] if ocols is not None else None) for i in range(cnt)]


def fuse_voxel_downsample(clouds, cols, Ts, voxel):
    """preprocessing/data.py:44-61 in one pass: cloud c moved by Ts[c], stacked in order, voxel_down_sample(voxel) of the
    stack on the fp64 values of the moved points (kpx_fuse_voxel_downsample).  cols: list of colour tensors or None.
    -> (points f32 (M,3), colours f32 (M,3) | None)"""
    lib = L.load()
    clouds = [_dev(p, torch.float32).reshape(-1, 3) for p in clouds]
    cnt = len(clouds)
    dev = clouds[0].device if cnt else L.device()
    cols = [_dev(c, torch.float32).reshape(-1, 3) for c in cols] if cols is not None else None
    n_arr = np.array([p.shape[0] for p in clouds], dtype=np.int64)
    total = int(n_arr.sum())
    T = np.ascontiguousarray(np.stack([_T(t) for t in Ts]))
    arr = lambda ts: C.cast((C.c_void_p * cnt)(*[t.data_ptr() for t in ts]), C.c_void_p) if ts is not None else None
    op = torch.empty((max(total, 1), 3), dtype=torch.float32, device=dev)
    oc = torch.empty_like(op) if cols is not None else None
    d_cnt = torch.empty(1, dtype=torch.int32, device=dev)
    ws, wsz = L.workspace(lib.kpx_fuse_voxel_workspace_bytes(total))
    L.check(lib.kpx_fuse_voxel_downsample(cnt, arr(clouds), arr(cols), n_arr.ctypes.data_as(C.c_void_p), L.hptr(T), float(voxel), L.ptr(op),
                                          L.ptr(oc), L.ptr(d_cnt), ws, wsz, L.stream_ptr()))
    m = _count(d_cnt)[0]
    return op[:m], (oc[:m] if oc is not None else None)


def sor(pts, nb_neighbors, std_ratio, want_avg=False):
    """a8.  Returns keep_idx i32 (K), stats f64 (3) [mean, std, thr] (device), avg f64 (N)|None."""
    lib = L.load()
    pts = _dev(pts, torch.float32).reshape(-1, 3)
    n = pts.shape[0]
    dev = pts.device
    idx = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    cnt = torch.empty(1, dtype=torch.int32, device=dev)
    stats = torch.empty(3, dtype=torch.float64, device=dev) if n else torch.zeros(3, dtype=torch.float64, device=dev)   # written by the statistics kernels
    avg = torch.empty(max(n, 1), dtype=torch.float64, device=dev) if want_avg else None
    ws, wsz = L.workspace(lib.kpx_sor_workspace_bytes(n, int(nb_neighbors)))
    L.check(lib.kpx_sor(L.ptr(pts), n, int(nb_neighbors), float(std_ratio), L.ptr(idx), L.ptr(cnt), L.ptr(stats),
                        L.ptr(avg), ws, wsz, L.stream_ptr()))
    k = _count(cnt)[0]
    return idx[:k], stats, (avg[:n] if avg is not None else None)


def sor_select(pts, attr, nb_neighbors, std_ratio):
    """a8 with both results of `cl, ind = remove_statistical_outlier(...)`: -> kept points f32 (K,3), kept rows of `attr` (K,3) | None,
    keep_idx i32 (K), stats f64 (3) -- one pass, no count read-back between filter and selection (kpx_sor_select)."""
    lib = L.load()
    pts = _dev(pts, torch.float32).reshape(-1, 3)
    n = pts.shape[0]
    dev = pts.device
    attr = _dev(attr, torch.float32).reshape(n, 3) if attr is not None else None
    idx = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    cnt = torch.empty(1, dtype=torch.int32, device=dev)
    stats = torch.empty(3, dtype=torch.float64, device=dev) if n else torch.zeros(3, dtype=torch.float64, device=dev)
    op = torch.empty((max(n, 1), 3), dtype=torch.float32, device=dev)
    oa = torch.empty((max(n, 1), 3), dtype=torch.float32, device=dev) if attr is not None else None
    ws, wsz = L.workspace(lib.kpx_sor_workspace_bytes(n, int(nb_neighbors)))
    L.check(lib.kpx_sor_select(L.ptr(pts), L.ptr(attr), n, int(nb_neighbors), float(std_ratio), L.ptr(op), L.ptr(oa), L.ptr(idx), L.ptr(cnt),
                               L.ptr(stats), ws, wsz, L.stream_ptr()))
    k = _count(cnt)[0]
    return op[:k], (oa[:k] if oa is not None else None), idx[:k], stats


def sor_partial(pts, nb_neighbors, q_begin, q_end, want_order=True):
    """Sharded a8, first half: mean kNN distances of the queries at cell-sorted positions [q_begin, q_end) of the cloud's
    grid order -> (avg f64 (q_end - q_begin) in that order, order i32 (N) sorted position -> point index | None)."""
    lib = L.load()
    pts = _dev(pts, torch.float32).reshape(-1, 3)
    n = pts.shape[0]
    avg = torch.empty(max(int(q_end - q_begin), 1), dtype=torch.float64, device=pts.device)
    order = torch.empty(n, dtype=torch.int32, device=pts.device) if want_order else None
    ws, wsz = L.workspace(lib.kpx_sor_workspace_bytes(n, int(nb_neighbors)))
    L.check(lib.kpx_sor_partial(L.ptr(pts), n, int(nb_neighbors), int(q_begin), int(q_end), L.ptr(avg), L.ptr(order), ws, wsz,
                                L.stream_ptr()))
    return avg[: int(q_end - q_begin)], order


def sor_finish(avg_sorted, order, std_ratio, want_avg=False):
    """Sharded a8, second half: the slabs' mean distances (all N, grid order) -> keep_idx, stats, avg | None as sor()."""
    lib = L.load()
    avg_sorted = _dev(avg_sorted, torch.float64).reshape(-1)
    order = _dev(order, torch.int32).reshape(-1)
    n = order.numel()
    assert avg_sorted.numel() == n
    dev = order.device
    idx = torch.empty(n, dtype=torch.int32, device=dev)
    cnt = torch.empty(1, dtype=torch.int32, device=dev)
    stats = torch.empty(3, dtype=torch.float64, device=dev)
    avg = torch.empty(n, dtype=torch.float64, device=dev) if want_avg else None
    ws, wsz = L.workspace(lib.kpx_sor_finish_workspace_bytes(n))
    L.check(lib.kpx_sor_finish(L.ptr(avg_sorted), L.ptr(order), n, float(std_ratio), L.ptr(idx), L.ptr(cnt), L.ptr(stats), L.ptr(avg),
                               ws, wsz, L.stream_ptr()))
    k = _count(cnt)[0]
    return idx[:k], stats, avg


def estimate_normals(pts, radius, max_nn):
    lib = L.load()
    pts = _dev(pts, torch.float32).reshape(-1, 3)
    n = pts.shape[0]
    out = torch.empty((n, 3), dtype=torch.float32, device=pts.device)
    ws, wsz = L.workspace(lib.kpx_normals_workspace_bytes(n, int(max_nn)))
    L.check(lib.kpx_estimate_normals(L.ptr(pts), n, float(radius), int(max_nn), L.ptr(out), ws, wsz, L.stream_ptr()))
    return out


def segment_plane(pts, distance_threshold, ransac_n, num_iterations, probability=0.99999999, seed=0):
    """a21.  Returns plane (numpy f64 (4,)), inlier idx i32 (K) device tensor."""
    lib = L.load()
    pts = _dev(pts, torch.float32).reshape(-1, 3)
    n = pts.shape[0]
    dev = pts.device
    plane = torch.zeros(4, dtype=torch.float64, device=dev)
    idx = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    cnt = torch.zeros(1, dtype=torch.int32, device=dev)
    ws, wsz = L.workspace(lib.kpx_segment_plane_workspace_bytes(n, int(ransac_n), int(num_iterations)))
    L.check(lib.kpx_segment_plane(L.ptr(pts), n, float(distance_threshold), int(ransac_n), int(num_iterations),
                                  float(probability), C.c_uint64(int(seed)), L.ptr(plane), L.ptr(idx), L.ptr(cnt), ws,
                                  wsz, L.stream_ptr()))
    k = _count(cnt)[0]
    return plane.cpu().numpy(), idx[:k]


# ---- registration -----------------------------------------------------------------------------------
def nn_search(src, tgt, T=None):
    """One correspondence search.  Returns idx i32 (N), d2 f64 (N)."""
    lib = L.load()
    src = _dev(src, torch.float32).reshape(-1, 3)
    tgt = _dev(tgt, torch.float32).reshape(-1, 3)
    dev = src.device
    Td = torch.as_tensor(_T(np.eye(4) if T is None else T)).to(dev)
    n, m = src.shape[0], tgt.shape[0]
    idx = torch.empty(n, dtype=torch.int32, device=dev)
    d2 = torch.empty(n, dtype=torch.float64, device=dev)
    ws, wsz = L.workspace(lib.kpx_nn_workspace_bytes(n, m))
    L.check(lib.kpx_nn_search(L.ptr(src), n, L.ptr(tgt), m, L.ptr(Td), L.ptr(idx), L.ptr(d2), ws, wsz, L.stream_ptr()))
    return idx, d2


def kabsch(src, tgt, corr):
    lib = L.load()
    src = _dev(src, torch.float32).reshape(-1, 3)
    tgt = _dev(tgt, torch.float32).reshape(-1, 3)
    corr = _dev(corr, torch.int32).reshape(-1, 2)
    T = torch.zeros(16, dtype=torch.float64, device=src.device)
    ws, wsz = L.workspace(lib.kpx_kabsch_workspace_bytes(corr.shape[0]))
    L.check(lib.kpx_kabsch(L.ptr(src), L.ptr(tgt), L.ptr(corr), corr.shape[0], L.ptr(T), ws, wsz, L.stream_ptr()))
    return T.cpu().numpy().reshape(4, 4)


def icp(src, tgt, max_dist, init=None, mode="p2p", tgt_normals=None, max_iteration=30, relative_fitness=1e-6,
        relative_rmse=1e-6, want_corr=False, poll_interval=4):
    """registration_icp.  Returns dict(transformation, fitness, inlier_rmse, iterations, count[, idx, d2])."""
    lib = L.load()
    src = _dev(src, torch.float32).reshape(-1, 3)
    tgt = _dev(tgt, torch.float32).reshape(-1, 3)
    dev = src.device
    n, m = src.shape[0], tgt.shape[0]
    tn = _dev(tgt_normals, torch.float32).reshape(-1, 3) if tgt_normals is not None else None
    md = {"p2p": 0, "p2plane": 1}[mode]
    if md == 1 and tn is None:
        raise L.KinectPxError("TransformationEstimationPointToPlane requires target normals")
    res = torch.zeros(20, dtype=torch.float64, device=dev)
    idx = torch.empty(n, dtype=torch.int32, device=dev) if want_corr else None
    d2 = torch.empty(n, dtype=torch.float64, device=dev) if want_corr else None
    init = _T(np.eye(4) if init is None else init)
    ws, wsz = L.workspace(lib.kpx_icp_workspace_bytes(n, m))
    L.check(lib.kpx_icp(L.ptr(src), n, L.ptr(tgt), L.ptr(tn), m, float(max_dist), L.hptr(init), md, int(max_iteration),
                        float(relative_fitness), float(relative_rmse), int(poll_interval), L.ptr(res), L.ptr(idx), L.ptr(d2), ws, wsz,
                        L.stream_ptr()))
    r = res.cpu().numpy()
    out = {"transformation": r[:16].reshape(4, 4).copy(), "fitness": float(r[16]), "inlier_rmse": float(r[17]),
           "iterations": int(r[18]), "count": int(r[19])}
    if want_corr:
        out["idx"], out["d2"] = idx, d2
    return out


def fpfh(pts, normals, radius, max_nn):
    """a11: compute_fpfh_feature.  Returns (N, 33) float64 device tensor."""
    lib = L.load()
    pts = _dev(pts, torch.float32).reshape(-1, 3)
    nrm = _dev(normals, torch.float32).reshape(-1, 3)
    n = pts.shape[0]
    out = torch.zeros((n, 33), dtype=torch.float64, device=pts.device)
    ws, wsz = L.workspace(lib.kpx_fpfh_workspace_bytes(n, int(max_nn)))
    L.check(lib.kpx_fpfh(L.ptr(pts), L.ptr(nrm), n, float(radius), int(max_nn), L.ptr(out), ws, wsz, L.stream_ptr()))
    return out


def feature_nn(fa, fb):
    lib = L.load()
    fa = _dev(fa, torch.float64).reshape(-1, 33)
    fb = _dev(fb, torch.float64).reshape(-1, 33)
    idx = torch.empty(fa.shape[0], dtype=torch.int32, device=fa.device)
    ws, wsz = L.workspace(lib.kpx_feature_nn_workspace_bytes(fa.shape[0], fb.shape[0]))
    L.check(lib.kpx_feature_nn(L.ptr(fa), fa.shape[0], L.ptr(fb), fb.shape[0], L.ptr(idx), ws, wsz, L.stream_ptr()))
    return idx


def feature_correspondences(fs, ft, mutual_filter=True, ransac_n=3):
    """correspondence stage of registration_ransac_based_on_feature_matching -> int32 (C, 2) numpy array"""
    ij = feature_nn(fs, ft).cpu().numpy()
    one_way = np.stack([np.arange(len(ij), dtype=np.int32), ij], 1)
    if not mutual_filter:
        return one_way
    ji = feature_nn(ft, fs).cpu().numpy()
    mutual = one_way[ji[ij] == np.arange(len(ij))]
    return mutual if len(mutual) >= ransac_n * 3 else one_way


def ransac_corres(src, tgt, corres, max_dist, ransac_n=3, edge_similarity=0.95, max_iteration=250000, confidence=0.999, seed=0):
    """a13: RegistrationRANSACBasedOnCorrespondence.  Returns dict(transformation, fitness, inlier_rmse, iterations, validations)."""
    lib = L.load()
    src = _dev(src, torch.float32).reshape(-1, 3)
    tgt = _dev(tgt, torch.float32).reshape(-1, 3)
    corres = _dev(corres, torch.int32).reshape(-1, 2)
    res = np.zeros(20)
    ws, wsz = L.workspace(lib.kpx_ransac_workspace_bytes(src.shape[0], tgt.shape[0]))
    L.check(lib.kpx_ransac_corres(L.ptr(src), src.shape[0], L.ptr(tgt), tgt.shape[0], L.ptr(corres), corres.shape[0], float(max_dist),
                                  int(ransac_n), float(edge_similarity), int(max_iteration), float(confidence), C.c_uint64(int(seed)),
                                  L.hptr(res), ws, wsz, L.stream_ptr()))
    return {"transformation": res[:16].reshape(4, 4).copy(), "fitness": float(res[16]), "inlier_rmse": float(res[17]),
            "iterations": int(res[18]), "validations": int(res[19])}


def icp_batch(srcs, tgt, max_dist, inits, mode="p2p", tgt_normals=None, max_iteration=30, relative_fitness=1e-6,
              relative_rmse=1e-6):
    """Several registrations onto one shared target, software-pipelined on the current stream.
    Returns a list of dicts like icp()."""
    lib = L.load()
    srcs = [_dev(s, torch.float32).reshape(-1, 3) for s in srcs]
    tgt = _dev(tgt, torch.float32).reshape(-1, 3)
    dev = tgt.device
    cnt, m = len(srcs), tgt.shape[0]
    tn = _dev(tgt_normals, torch.float32).reshape(-1, 3) if tgt_normals is not None else None
    md = {"p2p": 0, "p2plane": 1}[mode]
    if md == 1 and tn is None:
        raise L.KinectPxError("TransformationEstimationPointToPlane requires target normals")
    n_arr = np.array([s.shape[0] for s in srcs], dtype=np.int64)
    p_arr = (C.c_void_p * cnt)(*[s.data_ptr() for s in srcs])
    init = np.ascontiguousarray(np.stack([_T(np.eye(4) if T is None else T) for T in inits]))
    res = torch.empty((cnt, 20), dtype=torch.float64, device=dev)
    ws, wsz = L.workspace(lib.kpx_icp_batch_workspace_bytes(cnt, n_arr.ctypes.data_as(C.c_void_p), m))
    L.check(lib.kpx_icp_batch(cnt, C.cast(p_arr, C.c_void_p), n_arr.ctypes.data_as(C.c_void_p), L.ptr(tgt), L.ptr(tn), m,
                              float(max_dist), L.hptr(init), md, int(max_iteration), float(relative_fitness),
                              float(relative_rmse), L.ptr(res), ws, wsz, L.stream_ptr()))
    r = res.cpu().numpy()
    if np.isnan(r[:, 16]).any():      # icp_chain_kernel poisons the results of a chain that gave up its residency wait: THIS call's error
        raise L.KinectPxError("kpx_icp_batch: the one-launch ICP chain of registration(s) %s gave up waiting for its blocks to become resident "
                              "(another process on this GPU?); KPX_ICP_CHAIN=0 selects the launch-per-iteration form" % np.flatnonzero(np.isnan(r[:, 16])).tolist())
    return [{"transformation": r[i, :16].reshape(4, 4).copy(), "fitness": float(r[i, 16]), "inlier_rmse": float(r[i, 17]),
             "iterations": int(r[i, 18]), "count": int(r[i, 19])} for i in range(cnt)]


# ---- the frame loop as one native call ---------------------------------------------------------------------
class FrameParams(C.Structure):
    _fields_ = [("reg_voxel", C.c_double), ("icp_max_dist", C.c_double), ("filt_voxel", C.c_double), ("filt_ratio", C.c_double),
                ("gate", C.c_double), ("normals_nn", C.c_int32), ("icp_mode", C.c_int32), ("icp_max_iteration", C.c_int32),
                ("filt_k", C.c_int32)]


def frame_step(depth, rgb, xy_table, inits, params: FrameParams, out=None):
    """kpx_frame_step: depth (S, n_px) u16, rgb (S, n_px, 3) u8, xy (n_px*2) f32 on the device; inits: (S-1) 4x4.
    -> points f32 (K,3), colours f32 (K,3), transforms f64 (S,4,4) numpy, info int32 (64) numpy.
    depth / rgb may also be HOST tensors (pinned: the copy is asynchronous): kpx_frame_step_host stages them through the
    workspace on the frame's stream -- the frame then starts in host memory, as SURVEY 8(d) defines the end-to-end interval.
    out: optional (points, colours) buffers of S*n_px rows to write into (otherwise allocated)."""
    lib = L.load()
    host = isinstance(depth, torch.Tensor) and not depth.is_cuda
    if host:
        if depth.dtype != torch.uint16 or rgb.dtype != torch.uint8 or not (depth.is_contiguous() and rgb.is_contiguous()):
            raise ValueError("frame_step: host frames must be contiguous uint16 depth / uint8 rgb tensors")
        dev = L.device()
    else:
        depth = _dev(depth, torch.uint16)
        dev = depth.device
    S = int(depth.shape[0])
    depth = depth.reshape(S, -1)
    n_px = int(depth.shape[1])
    rgb = (rgb if host else _dev(rgb, torch.uint8)).reshape(S, n_px, 3)
    xy = _dev(xy_table, torch.float32).reshape(-1)
    init = np.ascontiguousarray(np.stack([_T(T) for T in inits])) if S > 1 else np.zeros((1, 4, 4))
    if out is None:
        out = (torch.empty((S * n_px, 3), dtype=torch.float32, device=dev), torch.empty((S * n_px, 3), dtype=torch.float32, device=dev))
    h_count = np.zeros(1, dtype=np.int32)
    h_T = np.zeros((S, 4, 4))
    h_info = np.zeros(64, dtype=np.int32)
    # one size for both forms (the host form's staging buffers are a few MB on top): a stream that sees frames of either kind
    # never regrows its scratch
    ws, wsz = L.workspace(lib.kpx_frame_step_host_workspace_bytes(S, n_px))
    if host:
        fn, dptr, cptr = lib.kpx_frame_step_host, C.c_void_p(depth.data_ptr()), C.c_void_p(rgb.data_ptr())
    else:
        fn, dptr, cptr = lib.kpx_frame_step, L.ptr(depth), L.ptr(rgb)
    L.check(fn(dptr, cptr, L.ptr(xy), n_px, S, L.hptr(init), C.byref(params), L.ptr(out[0]), L.ptr(out[1]),
               h_count.ctypes.data_as(C.c_void_p), L.hptr(h_T), h_info.ctypes.data_as(C.c_void_p), ws, wsz, L.stream_ptr()))
    k = int(h_count[0])
    return out[0][:k], out[1][:k], h_T, h_info


def frame_step_sharded(comm, order, frame, depth, rgb, xy_table, n_sensors, inits, params: FrameParams, fused_filter=0, out=None):
    """kpx_frame_step_sharded: this rank's share of the frame (depth (S_local, n_px) u16, rgb (S_local, n_px, 3) u8, on the device or
    in host memory), the collectives issued natively through `comm` (parallel.NativeComm).  inits: the n_sensors - 1 initial
    transforms of ALL sub sensors.  -> (points, colours, transforms (n_sensors,4,4), info) or L.RETRY when a message outgrew its
    capacity on every rank alike (run the frame again)."""
    lib = L.load()
    host = isinstance(depth, torch.Tensor) and not depth.is_cuda
    if host:
        if depth.dtype != torch.uint16 or rgb.dtype != torch.uint8 or not (depth.is_contiguous() and rgb.is_contiguous()):
            raise ValueError("frame_step_sharded: host frames must be contiguous uint16 depth / uint8 rgb tensors")
        dev = L.device()
    else:
        depth = _dev(depth, torch.uint16)
        rgb = _dev(rgb, torch.uint8)
        dev = depth.device
    S, S_l = int(n_sensors), int(depth.shape[0])
    n_px = int(depth.numel() // max(S_l, 1))
    xy = _dev(xy_table, torch.float32).reshape(-1)
    init = np.ascontiguousarray(np.stack([_T(T) for T in inits])) if S > 1 else np.zeros((1, 4, 4))
    if out is None:
        out = (torch.empty((S * n_px, 3), dtype=torch.float32, device=dev), torch.empty((S * n_px, 3), dtype=torch.float32, device=dev))
    h_count = np.zeros(1, dtype=np.int32)
    h_T = np.zeros((S, 4, 4))
    h_info = np.zeros(64, dtype=np.int32)
    # one size for device and host frames: a stream that sees both never regrows its scratch
    ws, wsz = L.workspace(lib.kpx_frame_step_sharded_workspace_bytes(S, comm.rank, comm.world, n_px, 1))
    rc = lib.kpx_frame_step_sharded(comm.handle, None if order is None else order.handle, -1 if frame is None else int(frame),
                                    C.c_void_p(depth.data_ptr()), C.c_void_p(rgb.data_ptr()), 1 if host else 0, L.ptr(xy), n_px, S, L.hptr(init),
                                    C.byref(params), int(fused_filter), L.ptr(out[0]), L.ptr(out[1]), h_count.ctypes.data_as(C.c_void_p), L.hptr(h_T),
                                    h_info.ctypes.data_as(C.c_void_p), ws, wsz, L.stream_ptr())
    if comm.error is not None:
        e, comm.error = comm.error, None
        raise e
    if rc == L.RETRY:
        return L.RETRY
    L.check(rc)
    k = int(h_count[0])
    return out[0][:k], out[1][:k], h_T, h_info


# ---- measurement hooks --------------------------------------------------------------------------------
NN_ENGINES = ("culled", "dense", "dense_fp64")


def nn_engine(name=None):
    """Select the correspondence-search implementation ("culled" default, "dense" all-pairs cross-check); returns the
    name of the engine that was active.  Results are identical; see include/kinectpx.h."""
    prev = L.load().kpx_nn_engine(-1 if name is None else NN_ENGINES.index(name))
    if prev < 0:
        L.check(prev)
    return NN_ENGINES[prev]


PROF_KERNELS = ("nn_mfma", "sor_knn", "plane_score", "compact", "nn_screen", "nn_local")


def prof_stride(stride):
    """time only every stride-th launch of each tagged kernel"""
    L.check(L.load().kpx_prof_stride(int(stride)))


def prof_begin(capacity=65536):
    L.check(L.load().kpx_prof_begin(int(capacity)))


def prof_end():
    """-> {kernel: (total_ms, launches, work)}; work = flops (nn_mfma) or algorithmic bytes"""
    ms = np.zeros(len(PROF_KERNELS))
    cnt = np.zeros(len(PROF_KERNELS), dtype=np.int64)
    work = np.zeros(len(PROF_KERNELS))
    L.check(L.load().kpx_prof_end(L.hptr(ms), cnt.ctypes.data_as(C.c_void_p), L.hptr(work)))
    return {k: (float(ms[i]), int(cnt[i]), float(work[i])) for i, k in enumerate(PROF_KERNELS)}


def prof_icp_phases():
    """-> average microseconds per block and phase of the LAST launch of the ICP iteration kernel (profiler armed), the time
    between the first and the last block start (dispatch ramp) and first start -> last end (span)"""
    out = np.zeros(16)
    L.check(L.load().kpx_prof_icp_phases(L.hptr(out)))
    names = ("update_prologue", "row_prep", "sweep", "pair_epilogue", "block_sums")
    d = {k: float(out[i]) for i, k in enumerate(names)}
    d.update(blocks=int(out[5]), dispatch_ramp_us=float(out[6]), span_us=float(out[7]), slowest={k: float(out[8 + i]) for i, k in enumerate(names)},
             longest_block_us=float(out[13]), blocks_skipped=int(out[14]))
    return d


def sort_pairs_u32(keys, vals, end_bit=32):
    """stable sort of (uint32 key, int32 value) pairs by the low end_bit key bits -> (keys, vals)"""
    lib = L.load()
    keys = _dev(keys, torch.uint32).reshape(-1)
    vals = _dev(vals, torch.int32).reshape(-1)
    n = keys.shape[0]
    ko, vo = torch.empty_like(keys), torch.empty_like(vals)
    ws, wsz = L.workspace(lib.kpx_sort_pairs_u32_workspace_bytes(n))
    L.check(lib.kpx_sort_pairs_u32(L.ptr(keys), L.ptr(vals), n, int(end_bit), L.ptr(ko), L.ptr(vo), ws, wsz, L.stream_ptr()))
    return ko, vo


def icp_chain(on=-1):
    """the one-launch form of the culled ICP chain on (1) / off (0) for this process; -1 only asks.  -> the previous setting (-2: -> chains launched by this process so far)"""
    return int(L.load().kpx_icp_chain(int(on)))


def prof_icp_chain():
    """-> (64, 48) uint64: the one-launch ICP chain's clock (KPX_ICP_CHAIN_STAMPS=1; slots in kpx_icp.hip, g_chain_stamp), reset by the call"""
    out = np.zeros((64, 48), dtype=np.uint64)
    L.check(L.load().kpx_prof_icp_chain(out.ctypes.data_as(C.c_void_p)))
    return out


def prof_icp_cert():
    """-> counters of the certificate self-check (KPX_ICP_CERT_CHECK=1): rows certified / searched / certified rows whose search
    disagreed (must be 0) and the first disagreement; cleared by the call"""
    out = np.zeros(8, dtype=np.uint64)
    L.check(L.load().kpx_prof_icp_cert(out.ctypes.data_as(C.c_void_p)))
    if int(out[2]) == 0:        # no disagreement: the chain's state at its last launch
        return dict(certified=int(out[0]), searched=int(out[1]), mismatches=0, iteration=int(out[3]), last_motion=float(out[4:5].view(np.float64)[0]),
                    motion=float(out[5:6].view(np.float64)[0]), skin=float(out[6:7].view(np.float64)[0]))
    return dict(certified=int(out[0]), searched=int(out[1]), mismatches=int(out[2]),
                first=dict(iteration=int(out[3]), row=int(out[4]), kept=int(out[5]), found=int(out[6]), key=float(np.uint32(out[7]).view(np.float32))))


def prof_icp_waves(cap=16384):
    """-> per wave of the LAST sweep launch: dict of numpy arrays (sweep_us, tiles, box_trips, mul_trips, groups_kept, with_partner)"""
    raw = np.zeros((cap, 4), dtype=np.uint64)
    cnt = np.zeros(1, dtype=np.int64)
    L.check(L.load().kpx_prof_icp_waves(raw.ctypes.data_as(C.c_void_p), cap, cnt.ctypes.data_as(C.c_void_p)))
    raw = raw[:int(cnt[0])]
    c = raw[:, 2]
    return dict(start=raw[:, 0].astype(np.int64), sweep_us=(raw[:, 1].astype(np.int64) - raw[:, 0].astype(np.int64)) * 0.01,
                tiles=(c & np.uint64(0xFFFF)).astype(np.int64), box_trips=((c >> np.uint64(16)) & np.uint64(0xFFFF)).astype(np.int64),
                mul_trips=((c >> np.uint64(32)) & np.uint64(0xFFFF)).astype(np.int64), groups_kept=((c >> np.uint64(48)) & np.uint64(0xFFFF)).astype(np.int64),
                with_partner=raw[:, 3].astype(np.int64))


# ---- sampler / normaliser ---------------------------------------------------------
NORM_OBB, NORM_OBB_ROT_TRANS, NORM_TRANSLATE, NORM_OBB_ROT = 0, 1, 2, 3
_OBB_ERRORS = {-1: "fewer than 3 distinct points, or all points on one line", -2: "the convex hull did not close",
               -3: "the convex hull is flat (Qhull: initial simplex is flat)"}


def sample_points(pts, k, seed, want_points=True):
    """seeded select_points_randomly (utils/processing.py:259-275) -> (points f32 (k,3) | None, idx i32 (k))"""
    lib = L.load()
    pts = _dev(pts, torch.float32).reshape(-1, 3)
    n, k = pts.shape[0], int(k)
    if k > n:
        raise ValueError("Cannot take a larger sample than population when 'replace=False'")
    out = torch.empty((k, 3), dtype=torch.float32, device=pts.device) if want_points else None
    idx = torch.empty(k, dtype=torch.int32, device=pts.device)
    ws, wsz = L.workspace(lib.kpx_sample_workspace_bytes(n))
    L.check(lib.kpx_sample_points(L.ptr(pts), n, k, C.c_uint64(int(seed) & (2 ** 64 - 1)), L.ptr(out), L.ptr(idx), ws, wsz,
                                  L.stream_ptr()))
    return out, idx


def obb_batch(x, want_vertices=False, check=True):
    """get_oriented_bounding_box() of every cloud of x: (count, n, 3) or (n, 3), f32 or f64 (other dtypes are read as
    f64).  -> obb f64 (count, 16) on the device [R | centre | extent | hull vertices], u8 (count, n) vertex flags | None.
    check=True reads the status back (one sync) and raises for degenerate clouds, as Qhull does."""
    lib = L.load()
    if not (isinstance(x, torch.Tensor) and x.dtype == torch.float32):
        x = _dev(x, torch.float64)
    x = _dev(x, x.dtype)
    x = x.reshape((-1,) + tuple(x.shape[-2:]))
    count, n = int(x.shape[0]), int(x.shape[1])
    obb = torch.empty((count, 16), dtype=torch.float64, device=x.device)
    flags = torch.empty((count, n), dtype=torch.uint8, device=x.device) if want_vertices else None
    ws, wsz = L.workspace(lib.kpx_obb_workspace_bytes(count, n))
    L.check(lib.kpx_obb_batch(L.ptr(x), 1 if x.dtype == torch.float64 else 0, count, n, L.ptr(obb), L.ptr(flags), ws, wsz,
                              L.stream_ptr()))
    if check and count:
        st = obb[:, 15].cpu()
        bad = torch.nonzero(st < 0).reshape(-1)
        if bad.numel():
            b = int(bad[0])
            raise L.KinectPxError(f"get_oriented_bounding_box: cloud {b}: {_OBB_ERRORS.get(int(st[b]), 'error')}")
    return obb, flags


def normalize_batch(x, obb, mode, M=None):
    """x f64 (count, rows, 3) -> the normalisation `mode` with the boxes of obb_batch (kpx_normalize_batch)"""
    lib = L.load()
    x = _dev(x, torch.float64)
    count, rows = int(x.shape[0]), int(x.shape[1])
    out = torch.empty_like(x)
    Mh = None if M is None else np.ascontiguousarray(M, dtype=np.float64).reshape(3, 3)
    L.check(lib.kpx_normalize_batch(L.ptr(x), count, rows, L.ptr(obb), int(mode), None if Mh is None else L.hptr(Mh), L.ptr(out),
                                    L.stream_ptr()))
    return out


def fuse_skeletons(skeletons, alpha=1.4, beta=1.4, initial_frame=20):
    """utils/skeleton_fusion.py:21-74.  skeletons (cams, frames, joints, 3) -> fused (frames, joints, 3) f64 on the device"""
    lib = L.load()
    sk = _dev(skeletons, torch.float64)
    if sk.dim() != 4 or sk.shape[3] != 3:
        raise ValueError("skeletons must have shape (cameras, frames, joints, 3)")
    cams, frames, joints = int(sk.shape[0]), int(sk.shape[1]), int(sk.shape[2])
    out = torch.empty((frames, joints, 3), dtype=torch.float64, device=sk.device)
    L.check(lib.kpx_fuse_skeletons(L.ptr(sk), cams, frames, joints, float(alpha), float(beta), int(initial_frame), L.ptr(out), L.stream_ptr()))
    return out


# ---- coloured ICP -------------------------------------------------------------------
def color_gradient(pts, normals, colors, radius, max_nn=30):
    """[O3D] InitializePointCloudForColoredICP: tangent-plane intensity gradient per point, (n, 3) f64 on the device"""
    lib = L.load()
    pts = _dev(pts, torch.float32).reshape(-1, 3)
    nrm = _dev(normals, torch.float32).reshape(-1, 3)
    col = _dev(colors, torch.float32).reshape(-1, 3)
    n = pts.shape[0]
    grad = torch.zeros((n, 3), dtype=torch.float64, device=pts.device)
    ws, wsz = L.workspace(lib.kpx_color_gradient_workspace_bytes(n, int(max_nn)))
    L.check(lib.kpx_color_gradient(L.ptr(pts), L.ptr(nrm), L.ptr(col), n, float(radius), int(max_nn), L.ptr(grad), ws, wsz, L.stream_ptr()))
    return grad


def colored_icp(src, src_colors, tgt, tgt_colors, tgt_normals, max_dist, init=None, lambda_geometric=0.968, max_iteration=30,
                relative_fitness=1e-6, relative_rmse=1e-6, poll_interval=4, tgt_gradient=None):
    """[O3D] registration_colored_icp.  Returns dict(transformation, fitness, inlier_rmse, iterations, count)."""
    lib = L.load()
    src = _dev(src, torch.float32).reshape(-1, 3)
    tgt = _dev(tgt, torch.float32).reshape(-1, 3)
    sc = _dev(src_colors, torch.float32).reshape(-1, 3)
    tc = _dev(tgt_colors, torch.float32).reshape(-1, 3)
    tn = _dev(tgt_normals, torch.float32).reshape(-1, 3)
    n, m = src.shape[0], tgt.shape[0]
    if tgt_gradient is None:
        tgt_gradient = color_gradient(tgt, tn, tc, 2.0 * float(max_dist), 30)       # Open3D: KDTreeSearchParamHybrid(2 max_distance, 30)
    tg = _dev(tgt_gradient, torch.float64).reshape(-1, 3)
    res = torch.zeros(20, dtype=torch.float64, device=src.device)
    init = _T(np.eye(4) if init is None else init)
    ws, wsz = L.workspace(lib.kpx_colored_icp_workspace_bytes(n, m))
    L.check(lib.kpx_colored_icp(L.ptr(src), L.ptr(sc), n, L.ptr(tgt), L.ptr(tc), L.ptr(tn), L.ptr(tg), m, float(max_dist), L.hptr(init),
                                float(lambda_geometric), int(max_iteration), float(relative_fitness), float(relative_rmse), int(poll_interval),
                                L.ptr(res), ws, wsz, L.stream_ptr()))
    r = res.cpu().numpy()
    return {"transformation": r[:16].reshape(4, 4).copy(), "fitness": float(r[16]), "inlier_rmse": float(r[17]),
            "iterations": int(r[18]), "count": int(r[19])}
