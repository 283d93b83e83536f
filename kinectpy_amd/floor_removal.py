"""Drop-in for KinectPy's floor_removal.py (reference lines 7-78)."""
import numpy as np

from . import ops
from .geometry import PointCloud


def pick_points(pcd):
    """floor_removal.py:7-18 opens an Open3D GUI window: out of scope."""
    raise NotImplementedError("pick_points needs Open3D's interactive visualiser (GUI, out of scope)")


def equation_plane(p1, p2, p3):
    """floor_removal.py:21-36: un-normalised plane through three points (prints it, as the reference)."""
    (x1, y1, z1), (x2, y2, z2), (x3, y3, z3) = p1, p2, p3
    ux, uy, uz = x2 - x1, y2 - y1, z2 - z1
    vx, vy, vz = x3 - x1, y3 - y1, z3 - z1
    a = uy * vz - vy * uz
    b = vx * uz - ux * vz
    c = ux * vy - uy * vx
    d = (- a * x1 - b * y1 - c * z1)
    print("equation of plane is ", a, "x +", b, "y +", c, "z +", d, "= 0.")
    return a, b, c, d


def pcd_above_plane(a, b, c, d, pcd: PointCloud) -> PointCloud:
    """floor_removal.py:39-51: despite the name it keeps the points whose plane value is < 0."""
    idx = ops.halfspace_select(pcd._pts, [a, b, c, d])
    return pcd._select(idx)


def remove_floor(pcd: PointCloud, slab=200, distance_threshold=30, ransac_n=30, num_iterations=2000,
                 nb_neighbors=50, std_ratio=0.30, seed=None) -> PointCloud:
    """Body of the reference's per-file loop (floor_removal.py:61-73): split at max(y)-slab, RANSAC
    plane on the lower slab, drop its inliers, concatenate with the upper part, SOR(50, 0.30)."""
    # max(y) from the cloud's bounds: left by the gather that produced the cloud (remove_statistical_outlier, _select) or one fast pass
    idx_lower, idx_upper = ops.slab_split(pcd._pts, float(slab), bounds=pcd._device_bounds())
    floor = pcd._select(idx_lower)
    _, inliers = floor.segment_plane(distance_threshold=distance_threshold, ransac_n=ransac_n,
                                     num_iterations=num_iterations, seed=seed)
    outlier_cloud = floor.select_by_index(inliers, invert=True)
    filtered = outlier_cloud + pcd._select(idx_upper)
    filtered, _ = filtered.remove_statistical_outlier(nb_neighbors, std_ratio)
    return filtered
