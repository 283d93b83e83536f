"""Drop-in for the computational part of KinectPy's manual_pointcloud_registration.py (reference lines 22-98): the rough
transform from picked point pairs (Umeyama / Kabsch without scale) and the point-to-point ICP refinement.  The interactive
point picking (Open3D's VisualizerWithEditing, :60-67) and the before / after viewers are GUI and out of scope: the picked
indices are passed in."""
import os

import numpy as np

from . import o3d
from .utils.io import load_color, load_depth, rgbd_to_pointcloud


def load_pointclouds(synced_files, root_dirs, frame: int = 1, isdir: bool = True):
    """manual_pointcloud_registration.py:22-42 (with `pcds` defined on the file branch too)"""
    pcds = []
    if isdir:
        for i, device in enumerate(synced_files.columns):
            timestamp = str(int(synced_files[device].iloc[frame]))
            color = load_color(os.path.join(root_dirs[i], 'color', timestamp + '_rgb.png'))
            depth = load_depth(os.path.join(root_dirs[i], 'depths', timestamp + '_depth.dat'))
            pcds.append(rgbd_to_pointcloud(color, depth))
    else:
        for fp in synced_files:
            pcds.append(o3d.io.read_point_cloud(fp))
    return pcds


def pick_points(pcd):
    raise NotImplementedError("pick_points opens Open3D's interactive visualiser (GUI, out of scope): pass picked_id_source / "
                              "picked_id_target to manual_registration")


def manual_registration(pcd_master, pcd_sub, picked_id_source=None, picked_id_target=None, threshold: float = 0.03) -> np.ndarray:
    """manual_pointcloud_registration.py:70-101: rough sub -> master transform from the user's picked point pairs (Umeyama /
    Kabsch without scale, :90-91), refined by point-to-point ICP on the full clouds (:96-98); returns the 4x4.  The sub device's
    cloud is the source, the master's the target.  picked_id_source[i] <-> picked_id_target[i] are indices into the two clouds
    (the reference obtains them from two GUI sessions, :80-81); threshold 0.03 is the reference's literal (:95)."""
    if picked_id_source is None or picked_id_target is None:
        picked_id_source, picked_id_target = pick_points(pcd_sub), pick_points(pcd_master)
    src_ids = np.asarray(picked_id_source, dtype=np.int64).reshape(-1)
    tgt_ids = np.asarray(picked_id_target, dtype=np.int64).reshape(-1)
    if min(len(src_ids), len(tgt_ids)) < 3 or len(src_ids) != len(tgt_ids):
        raise AssertionError("pick at least three points in each cloud, the same number in both")        # the reference asserts (:82-83)
    reg = o3d.pipelines.registration
    point_to_point = reg.TransformationEstimationPointToPoint()
    rough = point_to_point.compute_transformation(pcd_sub, pcd_master, o3d.utility.Vector2iVector(np.stack([src_ids, tgt_ids], 1)))
    refined = reg.registration_icp(pcd_sub, pcd_master, threshold, rough, point_to_point)              # inputs are not modified
    return refined.transformation
