"""Host mirror of the reference's utils/skeleton_fusion.py: the fusion itself runs on the MI355X
(kpx_fuse_skeletons); the file's __main__ block (CSV bookkeeping and the MoCap viewer) is outside the path."""
import numpy as np

from .. import ops


def fuse_skeletons_gradient(skeletons, alpha: float = 1.4, beta: float = 1.4) -> np.ndarray:
    """utils/skeleton_fusion.py:21-74: gradient- and centroid-weighted average of the joints of three cameras.
    skeletons: (cameras, frames, joints, 3); returns (frames, joints, 3) float64 like the reference."""
    sk = np.asarray(skeletons, dtype=np.float64)
    num_cameras, num_frames, num_joints, num_dims = sk.shape
    return ops.fuse_skeletons(sk, alpha, beta, 20).cpu().numpy()
