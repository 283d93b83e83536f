"""Host mirror of the point-cloud functions of the reference's utils/processing.py that follow the path (SURVEY 8f
rank 3): the random sampler and the single-cloud normalisers.  The timestamp / CSV bookkeeping of that file is outside
the path."""
import copy

import numpy as np
import torch

from .. import ops
from ..geometry import PointCloud


def sort_filenames_by_timestamp(list_of_files):
    """utils/processing.py:12-20: sort by the integer made of all digits in the name (in place, as the reference)"""
    list_of_files.sort(key=lambda x: int(''.join(filter(str.isdigit, x))))
    return np.array(list_of_files)


def find_delay_master_sub(timestamps_dataframe):
    """utils/processing.py:23-51: for every sub device (columns after the first), the row of `master_1` whose timestamp is
    closest to the sub's FIRST timestamp (first such row on ties).  Host bookkeeping of the extractor's synchronisation
    step; accepts a pandas DataFrame or a dict of columns."""
    cols = list(getattr(timestamps_dataframe, "columns", timestamps_dataframe.keys()))
    master = np.asarray(timestamps_dataframe["master_1"]).astype(np.int64)
    out = []
    for device in cols[1:]:
        first = int(np.asarray(timestamps_dataframe[device])[0])
        out.append(int(np.argmin(np.abs(first - master))) if master.size else -1)
    return out


def select_points_randomly(pointcloud, number_of_points, seed=None):
    """utils/processing.py:259-275: `number_of_points` of the cloud, without replacement, as a float64 (k,3) array.
    The reference draws from NumPy's global generator; `seed` fixes the draw (default: a seed drawn from that same
    generator, so np.random.seed() keeps governing it)."""
    if seed is None:
        seed = int(np.random.randint(0, 2 ** 31 - 1))
    pts, _ = ops.sample_points(pointcloud._pts, number_of_points, seed)
    return pts.cpu().numpy().astype(np.float64)


def scale_point_cloud(pcd, xs=0.001, ys=0.001, zs=0.001):
    """utils/processing.py:185-197"""
    out = copy.deepcopy(pcd)
    T = np.diag([xs, ys, zs, 1.0])
    out._pts = ops.transform(out._pts, T)
    return out


def statistical_outlier_removal(pcd, nb_neighbors=200, std_ratio=3.0):
    """utils/processing.py:302-310"""
    down = pcd.voxel_down_sample(voxel_size=0.02)
    filtered, _ = down.remove_statistical_outlier(nb_neighbors, std_ratio)
    return filtered


def normalize_pointcloud(pcd, min_range=-1.0, max_range=1.0):
    """utils/processing.py:313-326: one linear map of ALL coordinates onto [min_range, max_range] (in place)"""
    lo = float(np.min(pcd.get_min_bound()))
    hi = float(np.max(pcd.get_max_bound()))
    unit = (max_range - min_range) / (hi - lo)
    T = np.diag([unit, unit, unit, 1.0])
    T[:3, 3] = -lo * unit + min_range
    pcd._pts = ops.transform(pcd._pts, T)
    return pcd


def obb_normalization(points, joints, number_of_joints, _obb=None):
    """utils/processing.py:329-354: (p - centre) @ R of the cloud's oriented bounding box, joints likewise
    (`_obb`: a (1, 16) box tensor instead of the cloud's own -- the pin test feeds the recorded stub box)"""
    x = np.asarray(points, dtype=np.float64)[None]
    obb = _obb if _obb is not None else ops.obb_batch(x)[0]
    j = np.asarray(getattr(joints, "values", joints), dtype=np.float64).reshape(1, number_of_joints, 3)
    xo = ops.normalize_batch(x, obb, ops.NORM_OBB_ROT).cpu().numpy()[0]
    jo = ops.normalize_batch(j, obb, ops.NORM_OBB_ROT).cpu().numpy().reshape(number_of_joints * 3)
    return xo, jo


def save_points_npz(path, pointcloud, number_of_points, seed=None):
    """writes the `.npz` the reference's dataset reads (datasets/kinect_dataset_npz.py:96-97: npz['points'][:N])"""
    np.savez(path, points=select_points_randomly(pointcloud, number_of_points, seed))
