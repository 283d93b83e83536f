"""Drop-in for the frame loop of KinectPy's preprocessing/data.py (DataProcessor, reference lines 14-178),
on in-memory frames: mask + depth gate + compaction per sensor, transform of the sub sensors, fuse,
filter_outliers.  File discovery / .pcd writing stay host-side helpers."""
from typing import List, Optional, Sequence

import numpy as np
import torch

from .. import ops
from ..geometry import PointCloud
from .filtering import filter_outliers
from .registration import execute_global_registration, execute_point_to_plane_registration


def transform_filtered_image_to_pointcloud(filtered_img, depth_img) -> PointCloud:
    """data.py:165-178: keep pixels whose three colour channels are non-zero and whose z is
    <= median(z) + 750 (the `| z <= median - 750` clause is implied), then rgbd_to_pointcloud."""
    depth = np.asarray(depth_img)
    if depth.dtype != np.int16:
        depth = depth.astype(np.int16)
    (pts, col, _), = ops.rgbd_compact(depth.reshape(-1, 3), np.asarray(filtered_img, dtype=np.uint8).reshape(-1, 3), 1,
                                      color_mask=True, depth_gate=True, want_idx=False)
    return PointCloud._make(pts.clone(), col.clone())


def fuse_registered(filtered_pcds: Sequence[PointCloud], registration_transformations: Sequence[np.ndarray]) -> PointCloud:
    """data.py:44-58: device 0 untouched, device i>0 transformed in place by T[i-1]; vstack in device order."""
    pts, cols = [], []
    for i, pcd in enumerate(filtered_pcds):
        if i > 0:
            pcd.transform(registration_transformations[i - 1])
        pts.append(pcd._pts)
        cols.append(pcd._col)
    fused = PointCloud._make(torch.cat(pts, 0), torch.cat(cols, 0) if all(c is not None for c in cols) else None)
    return fused


class DataProcessor:
    """In-memory equivalent of the reference's DataProcessor: `find_registration_transforms` on frame 0
    (data.py:127-161: FPFH-RANSAC global registration, then point-to-plane ICP; caller-supplied initial
    transforms skip the global step) and `process_frame` per frame (data.py:35-61)."""

    def __init__(self, number_of_devices: int, initial_transformations: Optional[List[np.ndarray]] = None):
        self.number_of_devices = number_of_devices
        self.initial_transformations = initial_transformations
        self.registration_transformations: List[np.ndarray] = []

    def find_registration_transforms(self, master_pcd: PointCloud, sub_pcds: Sequence[PointCloud]):
        self.registration_transformations = []
        for i, sub in enumerate(sub_pcds):
            if self.initial_transformations is None:
                init = execute_global_registration(master_pcd, sub)              # data.py:156
                if init is None:
                    raise RuntimeError("execute_global_registration found no transformation (every RANSAC fitness was 0)")
            else:
                init = self.initial_transformations[i]
            self.registration_transformations.append(execute_point_to_plane_registration(master_pcd, sub, init))   # data.py:157
        return self.registration_transformations

    def process_frame(self, filtered_imgs, depth_imgs) -> PointCloud:
        pcds = [transform_filtered_image_to_pointcloud(c, d) for c, d in zip(filtered_imgs, depth_imgs)]
        return filter_outliers(fuse_registered(pcds, self.registration_transformations))
