"""Drop-in for KinectPy's preprocessing/registration.py (reference lines 7-114)."""
import copy

import numpy as np

from .. import o3d


def preprocess_point_cloud(pcd, voxel_size, normals_nn=30, fpfh_nn=100, with_fpfh=True):
    """registration.py:7-21: voxel down-sample, normals (radius 2v), FPFH (radius 5v).  with_fpfh=False skips the
    features (execute_point_to_plane_registration discards them)."""
    pcd_down = pcd.voxel_down_sample(voxel_size)
    pcd_down.estimate_normals(o3d.geometry.KDTreeSearchParamHybrid(radius=voxel_size * 2, max_nn=normals_nn))
    pcd_fpfh = None
    if with_fpfh:
        pcd_fpfh = o3d.pipelines.registration.compute_fpfh_feature(
            pcd_down, o3d.geometry.KDTreeSearchParamHybrid(radius=voxel_size * 5, max_nn=fpfh_nn))
    return pcd_down, pcd_fpfh


def prepare_dataset(pcd_master, pcd_sub, voxel_size, normals_nn=40, fpfh_nn=40, with_fpfh=True):
    """registration.py:24-29: source = sub, target = master."""
    source, target = copy.deepcopy(pcd_sub), copy.deepcopy(pcd_master)
    source_down, source_fpfh = preprocess_point_cloud(source, voxel_size, normals_nn, fpfh_nn, with_fpfh)
    target_down, target_fpfh = preprocess_point_cloud(target, voxel_size, normals_nn, fpfh_nn, with_fpfh)
    return source, target, source_down, target_down, source_fpfh, target_fpfh


def execute_global_registration(pcd_master, pcd_sub, voxel_size: int = 35, ransac_n_trials: int = 15, seed=None) -> np.ndarray:
    """registration.py:32-62: ransac_n_trials runs of FPFH feature-matching RANSAC (distance threshold 1.5 v,
    mutual filter, edge-length 0.95 + distance checkers, 250000 iterations, confidence 0.999); the
    transformation of the best fitness is kept (None if every fitness is 0).  The reference recomputes
    prepare_dataset in every trial with identical results; here it is computed once.  seed: base seed of the
    trials (None -> fresh random seeds, the reference's behaviour)."""
    best_fitness = 0
    ransac_transformation = None
    reg = o3d.pipelines.registration
    (source, target, source_down, target_down, source_fpfh, target_fpfh) = prepare_dataset(pcd_master, pcd_sub, voxel_size)
    distance_threshold = voxel_size * 1.5
    for trial in range(ransac_n_trials):
        result_ransac = reg.registration_ransac_based_on_feature_matching(
            source_down, target_down, source_fpfh, target_fpfh, True, distance_threshold,
            reg.TransformationEstimationPointToPoint(False), 3,
            [reg.CorrespondenceCheckerBasedOnEdgeLength(0.95), reg.CorrespondenceCheckerBasedOnDistance(distance_threshold)],
            reg.RANSACConvergenceCriteria(250000, 0.999), seed=None if seed is None else seed + trial)
        if best_fitness < result_ransac.fitness:
            best_fitness = result_ransac.fitness
            ransac_transformation = result_ransac.transformation
    return ransac_transformation


def execute_point_to_plane_registration(pcd_master, pcd_sub, initial_transformation: np.ndarray,
                                        voxel_size: int = 35) -> np.ndarray:
    """registration.py:65-86.  The reference names master `source` and sub `target` and then calls
    prepare_dataset(source, target), which swaps them back: the effective call is
    registration_icp(sub_down, master_down, 100, init, PointToPlane) and the result maps sub -> master."""
    source, target = copy.deepcopy(pcd_master), copy.deepcopy(pcd_sub)
    threshold = 100
    _, _, source_down, target_down, _, _ = prepare_dataset(source, target, voxel_size, with_fpfh=False)
    reg = o3d.pipelines.registration.registration_icp(
        source_down, target_down, threshold, initial_transformation,
        o3d.pipelines.registration.TransformationEstimationPointToPlane())
    return reg.transformation


def execute_colored_ICP_registration(pcd_master, pcd_sub, initial_transformation):
    """registration.py:89-114, as written there: master is the source, sub the target; three scales (voxel 80 / 40 / 20 with
    50 / 30 / 14 iterations), every scale starts again from `initial_transformation`, the last scale's result is returned."""
    source = copy.deepcopy(pcd_master)
    target = copy.deepcopy(pcd_sub)
    voxel_radius = [80, 40, 20]
    max_iter = [50, 30, 14]
    result_icp = None
    for scale in range(len(max_iter)):
        iters = max_iter[scale]
        radius = voxel_radius[scale]
        source_down = source.voxel_down_sample(radius)
        target_down = target.voxel_down_sample(radius)
        source_down.estimate_normals(o3d.geometry.KDTreeSearchParamHybrid(radius=radius * 2, max_nn=30))
        target_down.estimate_normals(o3d.geometry.KDTreeSearchParamHybrid(radius=radius * 2, max_nn=30))
        result_icp = o3d.pipelines.registration.registration_colored_icp(
            source_down, target_down, radius, initial_transformation,
            o3d.pipelines.registration.TransformationEstimationForColoredICP(),
            o3d.pipelines.registration.ICPConvergenceCriteria(relative_fitness=1e-6, relative_rmse=1e-6, max_iteration=iters))
    return result_icp.transformation
