"""Open3D-shaped namespace over the kinectpx hot path, so that KinectPy code written as
`import open3d as o3d` keeps working with `from kinectpy_amd import o3d` for the calls on the
path.  Anything off the path (visualisation)
raises NotImplementedError loudly instead of silently computing on the CPU.
"""
import types

import numpy as np

from . import ops
from .geometry import (KDTreeSearchParamHybrid, KDTreeSearchParamKNN, OrientedBoundingBox, PointCloud, Vector2iVector,
                       Vector3dVector)
from . import pcd_io


class ICPConvergenceCriteria:
    def __init__(self, relative_fitness=1e-6, relative_rmse=1e-6, max_iteration=30):
        self.relative_fitness, self.relative_rmse, self.max_iteration = relative_fitness, relative_rmse, max_iteration


class RegistrationResult:
    def __init__(self, transformation=None, fitness=0.0, inlier_rmse=0.0, correspondence_set=None):
        self.transformation = np.eye(4) if transformation is None else transformation
        self.fitness, self.inlier_rmse = fitness, inlier_rmse
        self.correspondence_set = correspondence_set if correspondence_set is not None else np.zeros((0, 2), np.int32)

    def __repr__(self):
        return (f"RegistrationResult with fitness={self.fitness:e}, inlier_rmse={self.inlier_rmse:e}, "
                f"and correspondence_set size of {len(self.correspondence_set)}")


class TransformationEstimationPointToPoint:
    mode = "p2p"

    def __init__(self, with_scaling=False):
        if with_scaling:
            raise NotImplementedError("with_scaling=True is not on the KinectPy path (registration.py:53 passes False)")
        self.with_scaling = False

    def compute_transformation(self, source, target, corres):
        """manual_pointcloud_registration.py:90-91: Umeyama/Kabsch on picked pairs"""
        return ops.kabsch(source._pts, target._pts, np.asarray(corres, dtype=np.int32))


class TransformationEstimationPointToPlane:
    mode = "p2plane"


class TransformationEstimationForColoredICP:
    mode = "colored"

    def __init__(self, lambda_geometric=0.968):
        self.lambda_geometric = float(lambda_geometric)


def registration_colored_icp(source, target, max_correspondence_distance, init=None, estimation_method=None, criteria=None):
    """preprocessing/registration.py:107-113"""
    est = estimation_method if estimation_method is not None else TransformationEstimationForColoredICP()
    crit = criteria if criteria is not None else ICPConvergenceCriteria()
    if max_correspondence_distance <= 0:
        raise RuntimeError("Invalid max_correspondence_distance.")
    if not target.has_normals():
        raise RuntimeError("TransformationEstimationPointToPlane and TransformationEstimationColoredICP "
                           "require pre-computed normal vectors for target PointCloud.")
    if not (source.has_colors() and target.has_colors()):
        raise RuntimeError("ColoredICP requires colored point clouds.")
    r = ops.colored_icp(source._pts, source._col, target._pts, target._col, target._nrm, float(max_correspondence_distance), init,
                        est.lambda_geometric, crit.max_iteration, crit.relative_fitness, crit.relative_rmse)
    return RegistrationResult(r["transformation"], r["fitness"], r["inlier_rmse"], None)


def registration_icp(source, target, max_correspondence_distance, init=None, estimation_method=None, criteria=None):
    """manual_pointcloud_registration.py:96-98, preprocessing/registration.py:78-84"""
    est = estimation_method if estimation_method is not None else TransformationEstimationPointToPoint()
    crit = criteria if criteria is not None else ICPConvergenceCriteria()
    if max_correspondence_distance <= 0:
        raise RuntimeError("Invalid max_correspondence_distance.")
    tn = None
    if est.mode == "p2plane":
        if not target.has_normals():
            raise RuntimeError("TransformationEstimationPointToPlane and TransformationEstimationColoredICP "
                               "require pre-computed normal vectors for target PointCloud.")
        tn = target._nrm
    r = ops.icp(source._pts, target._pts, float(max_correspondence_distance), init, est.mode, tn, crit.max_iteration,
                crit.relative_fitness, crit.relative_rmse, want_corr=True)
    idx, d2 = r["idx"].cpu().numpy(), r["d2"].cpu().numpy()
    ok = d2 < float(max_correspondence_distance) ** 2
    corr = np.stack([np.flatnonzero(ok).astype(np.int32), idx[ok]], 1)
    return RegistrationResult(r["transformation"], r["fitness"], r["inlier_rmse"], corr)


class Feature:
    """o3d.pipelines.registration.Feature: `.data` is (33, N) float64 like Open3D's; the device copy is (N, 33)."""

    def __init__(self, dev):
        self._dev = dev

    @property
    def data(self):
        return self._dev.cpu().numpy().T.copy()

    def dimension(self):
        return 33

    def num(self):
        return int(self._dev.shape[0])


class CorrespondenceCheckerBasedOnEdgeLength:
    def __init__(self, similarity_threshold=0.9):
        self.similarity_threshold = float(similarity_threshold)


class CorrespondenceCheckerBasedOnDistance:
    def __init__(self, distance_threshold):
        self.distance_threshold = float(distance_threshold)


class RANSACConvergenceCriteria:
    def __init__(self, max_iteration=100000, confidence=0.999):
        self.max_iteration, self.confidence = int(max_iteration), float(confidence)


def compute_fpfh_feature(input, search_param):
    """preprocessing/registration.py:15-20"""
    if not input.has_normals():
        raise RuntimeError("Failed because input point cloud has no normal.")
    return Feature(ops.fpfh(input._pts, input._nrm, search_param.radius, search_param.max_nn))


def registration_ransac_based_on_feature_matching(source, target, source_feature, target_feature, mutual_filter,
                                                  max_correspondence_distance, estimation_method=None, ransac_n=3,
                                                  checkers=(), criteria=None, seed=None):
    """preprocessing/registration.py:50-57.  Open3D's RANSAC is unseeded and runs its iterations in an OpenMP
    loop; ours draws from Philox with `seed` (None -> fresh random seed) and replays the iterations in order."""
    crit = criteria if criteria is not None else RANSACConvergenceCriteria()
    est = estimation_method if estimation_method is not None else TransformationEstimationPointToPoint()
    if est.mode != "p2p":
        raise NotImplementedError("feature-matching RANSAC is used with TransformationEstimationPointToPoint(False) by KinectPy")
    edge, dist_thr = 0.0, float(max_correspondence_distance)
    for c in checkers:
        if isinstance(c, CorrespondenceCheckerBasedOnEdgeLength):
            edge = c.similarity_threshold
        elif isinstance(c, CorrespondenceCheckerBasedOnDistance):
            dist_thr = c.distance_threshold
    if dist_thr != float(max_correspondence_distance):
        raise NotImplementedError("distance checker threshold must equal max_correspondence_distance (as in KinectPy)")
    if seed is None:
        seed = int(np.random.SeedSequence().generate_state(1, dtype=np.uint64)[0])
    # the correspondences depend only on the two feature sets: execute_global_registration calls this 15 times with the
    # same features (registration.py:46-57), so they are computed once and kept on the source Feature
    key = (id(target_feature), bool(mutual_filter), int(ransac_n))
    cached = getattr(source_feature, "_corr_cache", None)
    if cached is not None and cached[0] == key and cached[2] is target_feature:
        corres = cached[1]
    else:
        corres = ops.feature_correspondences(source_feature._dev, target_feature._dev, bool(mutual_filter), int(ransac_n))
        source_feature._corr_cache = (key, corres, target_feature)
    r = ops.ransac_corres(source._pts, target._pts, corres, float(max_correspondence_distance), int(ransac_n), edge,
                          crit.max_iteration, crit.confidence, seed)
    return RegistrationResult(r["transformation"], r["fitness"], r["inlier_rmse"], corres)


def _off_path(name):
    def f(*a, **k):
        raise NotImplementedError(f"{name} is outside the round-1 hot path of kinectpy_amd (SURVEY.md 8f); "
                                  "there is no CPU fallback")
    return f


geometry = types.SimpleNamespace(PointCloud=PointCloud, OrientedBoundingBox=OrientedBoundingBox, KDTreeSearchParamHybrid=KDTreeSearchParamHybrid,
                                 KDTreeSearchParamKNN=KDTreeSearchParamKNN)
utility = types.SimpleNamespace(Vector3dVector=Vector3dVector, Vector2iVector=Vector2iVector)
io = types.SimpleNamespace(read_point_cloud=pcd_io.read_point_cloud, write_point_cloud=pcd_io.write_point_cloud)
pipelines = types.SimpleNamespace(registration=types.SimpleNamespace(
    registration_icp=registration_icp,
    ICPConvergenceCriteria=ICPConvergenceCriteria,
    RegistrationResult=RegistrationResult,
    TransformationEstimationPointToPoint=TransformationEstimationPointToPoint,
    TransformationEstimationPointToPlane=TransformationEstimationPointToPlane,
    compute_fpfh_feature=compute_fpfh_feature,
    registration_ransac_based_on_feature_matching=registration_ransac_based_on_feature_matching,
    Feature=Feature,
    CorrespondenceCheckerBasedOnEdgeLength=CorrespondenceCheckerBasedOnEdgeLength,
    CorrespondenceCheckerBasedOnDistance=CorrespondenceCheckerBasedOnDistance,
    RANSACConvergenceCriteria=RANSACConvergenceCriteria,
    registration_colored_icp=registration_colored_icp,
    TransformationEstimationForColoredICP=TransformationEstimationForColoredICP,
))
visualization = types.SimpleNamespace(VisualizerWithEditing=_off_path("VisualizerWithEditing"),
                                      draw_geometries=_off_path("draw_geometries"))
