"""Drop-in for the computational part of KinectPy's manual_pointcloud_registration.py (reference lines 22-98): the rough
transform from picked point pairs (Umeyama / Kabsch without scale) and the point-to-point ICP refinement.  The interactive
point picking (Open3D's VisualizerWithEditing, :60-67) and the before / after viewers are GUI and out of scope: the picked
indices are passed in."""
import copy
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
    """manual_pointcloud_registration.py:70-101.  source = the sub device's cloud, target = the master's (:75-76);
    picked_id_source[i] <-> picked_id_target[i] are the user's correspondences (:80-87); returns the 4x4 sub -> master.
    threshold 0.03 is the reference's literal (:95)."""
    source, target = copy.deepcopy(pcd_sub), copy.deepcopy(pcd_master)
    if picked_id_source is None or picked_id_target is None:
        picked_id_source, picked_id_target = pick_points(source), pick_points(target)
    assert len(picked_id_source) >= 3 and len(picked_id_target) >= 3
    assert len(picked_id_source) == len(picked_id_target)
    corr = np.zeros((len(picked_id_source), 2))
    corr[:, 0] = picked_id_source
    corr[:, 1] = picked_id_target
    p2p = o3d.pipelines.registration.TransformationEstimationPointToPoint()
    trans_init = p2p.compute_transformation(source, target, o3d.utility.Vector2iVector(corr))
    reg_p2p = o3d.pipelines.registration.registration_icp(source, target, threshold, trans_init,
                                                           o3d.pipelines.registration.TransformationEstimationPointToPoint())
    return reg_p2p.transformation
