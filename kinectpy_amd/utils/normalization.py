"""Host mirror of the reference's utils/normalization.py: the batch normalisers of the training input pipeline, computed
on the MI355X (kpx_obb_batch + kpx_normalize_batch).  Same names, argument meaning and return shapes; the
`tf.numpy_function` wrappers (normalize_obb, ...) are plain callables here since TensorFlow is outside the path.

x: (B, N, 3) points, y: (B, 3K) flattened joints; both come back as float64 NumPy arrays, as in the reference.
"""
import numpy as np
import torch

from .. import ops
from ..geometry import OrientedBoundingBox


def _boxes(x):
    x = np.asarray(x, dtype=np.float64) if not isinstance(x, torch.Tensor) else x
    obb, _ = ops.obb_batch(x)
    return obb


def _apply(x, y, mode, M=None, obb=None):
    """obb: (B, 16) box tensor [R | centre | extent | status] instead of the clouds' own boxes (pin tests)"""
    xd = ops._dev(x, torch.float64)
    obb = _boxes(xd) if obb is None else obb
    B = xd.shape[0]
    yd = ops._dev(np.asarray(y, dtype=np.float64).reshape(B, -1, 3), torch.float64)
    xo = ops.normalize_batch(xd, obb, mode, M)
    yo = ops.normalize_batch(yd, obb, mode, M)
    return xo.cpu().numpy(), yo.cpu().numpy().reshape(B, -1)


def obb_normalization_batch(x, y):
    """utils/normalization.py:16-64: (p @ R_yxz([0, pi, 0]) + centre) / max extent -- as written in the reference."""
    x = np.asarray(x, dtype=np.float64)
    if x.ndim == 2:
        x = x[np.newaxis, :]
    return _apply(x, y, ops.NORM_OBB, OrientedBoundingBox.get_rotation_matrix_from_yxz([0, np.pi, 0]))


def obb_rotation_translation_batch(x, y):
    """utils/normalization.py:67-97: (p - centre) @ R @ Rz(90 deg)"""
    r = OrientedBoundingBox.get_rotation_matrix_from_yxz([0, 0, np.pi / 2])     # scipy R.from_euler('z', 90 deg)
    return _apply(x, y, ops.NORM_OBB_ROT_TRANS, r)


def translation_normalization_batch(x, y):
    """utils/normalization.py:100-126: p - centre"""
    return _apply(x, y, ops.NORM_TRANSLATE)


def _affine_batch(x, y, A):
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    xo = ops.joints_affine(x.reshape(-1, 3), A, np.zeros(3)).cpu().numpy().reshape(x.shape)
    yo = ops.joints_affine(y.reshape(-1, 3), A, np.zeros(3)).cpu().numpy().reshape(y.shape)
    return xo, yo


def scale_batch(x, y, scale=1 / 1000):
    """utils/normalization.py:129-139"""
    return _affine_batch(x, y, np.eye(3) * scale)


def rotate_batch(x, y, degs=None):
    """utils/normalization.py:142-159: one rotation about y for the whole batch.  The reference draws the angle from
    NumPy's global generator (np.random.randint(360)); pass `degs` to fix it."""
    if degs is None:
        degs = np.random.randint(360)
    a = np.deg2rad(float(degs))
    rotation = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
    return _affine_batch(x, y, rotation)


# the reference wraps the functions above in tf.function / tf.numpy_function (utils/normalization.py:161-188)
normalize_obb = obb_normalization_batch
normalize_obb_rotation_translation = obb_rotation_translation_batch
translate = translation_normalization_batch
scale = scale_batch
rotate = rotate_batch

# options/normalization.py:3-7
normalization_options = {
    "obb_normalization": normalize_obb,
    "obb_rotation_translation": normalize_obb_rotation_translation,
    "translation": translate,
}
