"""CPU stand-in for kinectpy_amd.ops, backed by the oracle -- TEST INFRASTRUCTURE ONLY.

The multi-rank logic of kinectpy_amd.pipeline.SensorShardPipeline (who owns which sensor, what is broadcast, gathered and
filtered where) has to be rehearsed on this CPU-only container with world_size > 1 over gloo.  The product has no CPU path
(kinectpy_amd.ops raises without the HIP library and a GPU), so the rehearsal injects this namespace as the pipeline's
`ops_module`: the same function names and return shapes, CPU torch tensors, every computation by oracle/.  Nothing under
kinectpy_amd/ imports this file."""
import numpy as np
import torch

from oracle import oracle as O


def _np(t):
    return t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)


def _dev(x, dtype):
    if isinstance(x, torch.Tensor):
        return x.to(dtype=dtype).contiguous()
    return torch.as_tensor(np.ascontiguousarray(x)).to(dtype).contiguous()


def _count(t):
    return [int(v) for v in t.tolist()]


def depth_to_cloud(depth, xy_table, rgb=None, frames=1, color_mask=False, depth_gate=False, gate=750.0, want_idx=False, sync=True):
    depth = _np(depth).reshape(frames, -1)
    xy = _np(xy_table).reshape(-1, 2)
    n = depth.shape[1]
    rgb = _np(rgb).reshape(frames, n, 3) if rgb is not None else None
    pts = torch.zeros((frames, n, 3), dtype=torch.float32)
    col = torch.zeros((frames, n, 3), dtype=torch.float32) if rgb is not None else None
    cnt = torch.zeros(frames, dtype=torch.int32)
    for f in range(frames):
        xyz = O.unproject_u16(depth[f], xy)
        hi = (O.median_z(xyz) + gate) if depth_gate else 0.0
        p, c, _ = O.rgbd_compact(xyz, rgb[f] if rgb is not None else None, bool(color_mask and rgb is not None), bool(depth_gate), hi)
        pts[f, :len(p)] = torch.as_tensor(p)
        if col is not None:
            col[f, :len(p)] = torch.as_tensor(c)
        cnt[f] = len(p)
    assert not sync
    return pts, col, None, cnt


def voxel_downsample_batch(clouds, voxel, cols=None):
    return [(torch.as_tensor(O.voxel_downsample(_np(p), voxel)[0]), None) for p in clouds]


def voxel_downsample(pts, voxel, col=None, nrm=None):
    vp, vc, _ = O.voxel_downsample(_np(pts), voxel, _np(col) if col is not None else None)
    return torch.as_tensor(vp), (torch.as_tensor(vc) if vc is not None else None), None


def fuse_voxel_downsample(clouds, cols, Ts, voxel):
    vp, vc = O.fuse_voxel_downsample([_np(p) for p in clouds], [_np(c) for c in cols] if cols is not None else None, Ts, voxel)
    return torch.as_tensor(vp), (torch.as_tensor(vc) if vc is not None else None)


def estimate_normals(pts, radius, max_nn):
    return torch.as_tensor(O.estimate_normals(_np(pts), radius, max_nn)[0].astype(np.float32))


def icp_batch(srcs, tgt, max_dist, inits, mode="p2p", tgt_normals=None, max_iteration=30):
    out = []
    tn = _np(tgt_normals) if tgt_normals is not None else None
    for s, T0 in zip(srcs, inits):
        T, fit, rmse, it = O.registration_icp(_np(s), _np(tgt), max_dist, T0, mode, tn, max_iteration, grid=True)
        out.append({"transformation": T, "fitness": fit, "inlier_rmse": rmse, "iterations": it})
    return out


def transform(pts, T, out=None):
    return torch.as_tensor(O.transform(_np(pts), T))


def sor(pts, nb_neighbors, std_ratio, want_avg=False):
    keep, stats, avg = O.sor(_np(pts), nb_neighbors, std_ratio)
    return torch.as_tensor(keep), stats, torch.as_tensor(avg)


def sor_partial(pts, nb_neighbors, q_begin, q_end, want_order=True):
    """the oracle has no grid order: the identity stands in for it (slab = a range of point indices)"""
    _, _, avg = O.sor(_np(pts), nb_neighbors, 1.0)
    return torch.as_tensor(avg[q_begin:q_end].copy()), torch.arange(len(avg), dtype=torch.int32)


def sor_finish(avg_sorted, order, std_ratio, want_avg=False):
    a = _np(avg_sorted)
    avg = np.empty_like(a)
    avg[_np(order)] = a
    keep, stats = O.sor_from_avg(avg, std_ratio)
    return torch.as_tensor(keep), stats, None


def select_by_index(attrs, idx, invert=False, trusted=False):
    i = _np(idx).astype(np.int64)
    return [torch.as_tensor(_np(a)[i]) if a is not None else None for a in list(attrs) + [None] * (3 - len(attrs))]
