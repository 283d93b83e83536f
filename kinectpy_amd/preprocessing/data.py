"""Drop-in for KinectPy's preprocessing/data.py (DataProcessor, reference lines 14-178).

`DataProcessor(output_dirs, mask_rcnn_pb_file, mask_rcnn_pbtxt_file)` does what the reference's constructor does: reads
the `<dir>/color/<ts>_rgb.png` + `<dir>/depths/<ts>_depth.dat` pairs of every device in timestamp order (data.py:73-84),
registers every sub device onto the master on frame 0 -- global FPFH-RANSAC registration, then point-to-plane ICP -- and
saves `transformation_master_sub_{i}.npy` next to the master (data.py:127-161), then per frame: person mask, depth gate,
compaction (data.py:87-124, 165-178), pcd.transform of the subs, vstack, filter_outliers (data.py:44-61) and
`<master>/filtered_and_registered_pointclouds/<ts>.pcd` (data.py:64-69).
Mask R-CNN inference itself is out of scope (weights unavailable): pass `mask_fn(img) -> (H, W) mask`.
`DataProcessor.in_memory(n, transforms)` is the same frame loop on arrays already in memory (no files)."""
import logging
import os
from typing import List, Optional, Sequence

import numpy as np
import torch

from .. import ops
from ..geometry import PointCloud
from ..pcd_io import write_point_cloud
from ..utils.io import load_color, load_depth, rgbd_to_pointcloud
from ..utils.processing import sort_filenames_by_timestamp
from .filtering import Filtering, filter_outliers
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
    """data.py:44-58 as written: device 0 untouched, device i>0 transformed in place by T[i-1]; vstack in device order.
    (The clouds of this library are float32: the stacked cloud carries the float32 rounding of the moved points.
    DataProcessor's frame loop does not use it -- see fuse_and_filter.)"""
    pts, cols = [], []
    for i, pcd in enumerate(filtered_pcds):
        if i > 0:
            pcd.transform(registration_transformations[i - 1])
        pts.append(pcd._pts)
        cols.append(pcd._col)
    fused = PointCloud._make(torch.cat(pts, 0), torch.cat(cols, 0) if all(c is not None for c in cols) else None)
    return fused


def fuse_and_filter(filtered_pcds: Sequence[PointCloud], registration_transformations: Sequence[np.ndarray], nb_neighbors: int = 200,
                    std_ratio: float = 3.0, voxel_size: float = 0.02) -> PointCloud:
    """data.py:44-61 in the reference's precision: pcd.transform(T_i), np.vstack and the voxel_down_sample of filter_outliers
    happen in ONE kernel pass on the fp64 values of the moved points (the reference's arrays are float64: utils/io.py:29-41,
    data.py:55-56), then remove_statistical_outlier.  Equal to filter_outliers(fuse_registered(...)) up to the float32
    rounding of the moved points that the latter would feed to the voxel grid."""
    Ts = [np.eye(4)] + [np.asarray(T, dtype=np.float64) for T in registration_transformations[:len(filtered_pcds) - 1]]
    cols = [p._col for p in filtered_pcds]
    vp, vc = ops.fuse_voxel_downsample([p._pts for p in filtered_pcds], cols if all(c is not None for c in cols) else None, Ts, float(voxel_size))
    cloud, _ = PointCloud._make(vp, vc).remove_statistical_outlier(nb_neighbors, std_ratio)
    return cloud


class DataProcessor:
    def __init__(self, output_dirs: List[str], mask_rcnn_pb_file: Optional[str] = None, mask_rcnn_pbtxt_file: Optional[str] = None, *,
                 mask_fn=None, initial_transformations: Optional[List[np.ndarray]] = None, seed: Optional[int] = None,
                 run: bool = True):
        """reference signature (data.py:15-27) + keyword-only extras: `mask_fn` (the person mask, instead of Mask R-CNN),
        `initial_transformations` (skip the global registration: data.py:156), `seed` (the reference's RANSAC is unseeded),
        `run=False` (build the object, call find / process yourself)."""
        self.device_filenames_df = self._create_device_filenames_df(output_dirs)
        self.number_of_devices = len(self.device_filenames_df.columns)
        self.registration_transformations: List[np.ndarray] = []
        self.initial_transformations = initial_transformations
        self.seed = seed
        self.segmentation = None
        self._mask_args = (mask_rcnn_pb_file, mask_rcnn_pbtxt_file, mask_fn)
        if not run:
            return
        self._find_registration_transforms()
        logging.info('Starting to filter background and save filtered point clouds')
        self.segmentation = Filtering(mask_rcnn_pb_file, mask_rcnn_pbtxt_file, mask_fn=mask_fn)
        for file_idx in range(len(self.device_filenames_df)):
            if (file_idx + 1) % 50 == 0:
                print(f'{file_idx + 1} point clouds have been saved')
            registered_pcd = fuse_and_filter(self._filter_pointclouds_and_save(file_idx), self.registration_transformations)
            dst = os.path.join(self.device_filenames_df.columns[0], 'filtered_and_registered_pointclouds',
                               self.device_filenames_df.iloc[file_idx, 0])
            write_point_cloud(dst + '.pcd', registered_pcd)                       # data.py:64-69

    @classmethod
    def in_memory(cls, number_of_devices: int, initial_transformations: Optional[List[np.ndarray]] = None, seed: Optional[int] = None):
        """the frame loop without the directory walk: find_registration_transforms(master, subs) / process_frame(imgs, depths)"""
        self = cls.__new__(cls)
        self.device_filenames_df = None
        self.number_of_devices = number_of_devices
        self.registration_transformations = []
        self.initial_transformations = initial_transformations
        self.seed = seed
        self.segmentation = None
        return self

    # ---- data.py:73-84
    @staticmethod
    def _create_device_filenames_df(output_dirs):
        import pandas as pd
        mapper, color_suffix = {}, '_rgb.png'
        for output_dir in output_dirs:
            names = [x.split(color_suffix)[0] for x in os.listdir(os.path.join(output_dir, 'color'))]
            mapper[output_dir] = sort_filenames_by_timestamp(names)
        return pd.DataFrame(mapper)

    def _device_frame(self, device_idx, file_idx):
        root = self.device_filenames_df.columns[device_idx]
        name = self.device_filenames_df.iloc[file_idx, device_idx]
        return load_color(os.path.join(root, 'color', name)), load_depth(os.path.join(root, 'depths', name))

    # ---- data.py:87-124
    def _filter_pointclouds_and_save(self, file_idx):
        if self.segmentation is None:
            self.segmentation = Filtering(self._mask_args[0], self._mask_args[1], mask_fn=self._mask_args[2])
        clouds = []
        for device_idx in range(self.number_of_devices):
            color, depth = self._device_frame(device_idx, file_idx)
            clouds.append(transform_filtered_image_to_pointcloud(self.segmentation.apply_segmentation(color), depth))
        return clouds

    # ---- data.py:127-161
    def _find_registration_transforms(self):
        master_pcd = rgbd_to_pointcloud(*self._device_frame(0, 0))
        subs = [rgbd_to_pointcloud(*self._device_frame(i, 0)) for i in range(1, self.number_of_devices)]
        self.find_registration_transforms(master_pcd, subs)
        for i, T in enumerate(self.registration_transformations, start=1):
            np.save(os.path.join(self.device_filenames_df.columns[0], f'transformation_master_sub_{i}.npy'), T)

    def find_registration_transforms(self, master_pcd: PointCloud, sub_pcds: Sequence[PointCloud]):
        self.registration_transformations = []
        for i, sub in enumerate(sub_pcds):
            if self.initial_transformations is None:
                init = execute_global_registration(master_pcd, sub, seed=self.seed)            # data.py:156
                if init is None:
                    raise RuntimeError("execute_global_registration found no transformation (every RANSAC fitness was 0)")
            else:
                init = self.initial_transformations[i]
            self.registration_transformations.append(execute_point_to_plane_registration(master_pcd, sub, init))   # data.py:157
        return self.registration_transformations

    def process_frame(self, filtered_imgs, depth_imgs) -> PointCloud:
        """data.py:40-61 for one synchronised frame set already in memory"""
        pcds = [transform_filtered_image_to_pointcloud(c, d) for c, d in zip(filtered_imgs, depth_imgs)]
        return fuse_and_filter(pcds, self.registration_transformations)
