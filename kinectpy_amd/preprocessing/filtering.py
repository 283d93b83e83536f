"""Drop-in for KinectPy's preprocessing/filtering.py (reference lines 12-129)."""
import copy

import numpy as np

from ..geometry import PointCloud


def filter_outliers(pcd: PointCloud, nb_neighbors: int = 200, std_ratio: float = 3.0,
                    voxel_size: float = 0.02) -> PointCloud:
    """preprocessing/filtering.py:12-25: voxel down-sample then statistical outlier removal; returns
    the cloud only."""
    down = copy.deepcopy(pcd).voxel_down_sample(voxel_size)
    cloud, _ = down.remove_statistical_outlier(nb_neighbors, std_ratio)
    return cloud


class Filtering:
    """preprocessing/filtering.py:28-95 runs Mask R-CNN through cv2.dnn; the weights are not
    distributable (README.md:17) and DNN inference is outside the hot path.  The
    drop-in therefore takes the person mask from a caller-supplied function and applies the
    reference's convention: background pixels become (0,0,0)."""

    def __init__(self, frozen_graph_fp=None, pbtxt_fp=None, mask_fn=None):
        if mask_fn is None:
            raise NotImplementedError("Mask R-CNN inference is out of scope: pass mask_fn(img) -> bool/uint8 (H,W) mask")
        self.mask_fn = mask_fn

    def apply_segmentation(self, img):
        mask = np.asarray(self.mask_fn(img)).astype(bool)
        img[~mask] = 0
        return img


def kalman_filter(joint_vals: np.ndarray, ri=10, qi=10, fi=1 / 30, hi=1) -> np.ndarray:
    """preprocessing/filtering.py:98-129: constant-gain-structure Kalman filter over N frames of one
    3-D joint (F = fi I, H = hi I, R = ri I, Q = qi I, P0 = I, x0 = first sample).  Sequential 3x3 fp64
    recursion: host side by design (no kernel)."""
    z = np.asarray(joint_vals, dtype=np.float64)
    eye = np.identity(3)
    F, H, R, Q = fi * eye, hi * eye, ri * eye, qi * eye
    P, x = eye, z[0]
    track = [x]
    for obs in z[1:]:
        x_prior = F @ x
        P_prior = F @ P @ F.T + Q
        gain = P_prior @ H.T @ np.linalg.inv(H @ P_prior @ H.T + R)
        x = x_prior + gain @ (obs - H @ x_prior)
        P = (eye - gain @ H) @ P_prior
        track.append(x)
    return np.array(track)
