"""Drop-in for KinectPy's utils/io.py (reference utils/io.py:6-43)."""
import numpy as np

from .. import ops
from ..geometry import PointCloud

COLOR_SUFFIX = "_rgb.png"
DEPTH_SUFFIX = "_depth.dat"


def load_color(color_fp: str) -> np.ndarray:
    """utils/io.py:6-12: the PNG as an RGB (H, W, 3) uint8 array.  Host file I/O: OpenCV as in the reference when it is
    installed, otherwise Pillow (both decode PNG losslessly, so the array is the same)."""
    if not color_fp.endswith(COLOR_SUFFIX):
        color_fp += COLOR_SUFFIX
    try:
        import cv2
    except ImportError:
        from PIL import Image
        with Image.open(color_fp) as im:
            return np.array(im.convert("RGB"))
    return cv2.cvtColor(cv2.imread(color_fp), cv2.COLOR_BGR2RGB)


def load_depth(depth_fp: str) -> np.ndarray:
    """utils/io.py:15-20: raw int16 (N,3) XYZ millimetres."""
    if not depth_fp.endswith(DEPTH_SUFFIX):
        depth_fp += DEPTH_SUFFIX
    return np.fromfile(depth_fp, dtype=np.int16).reshape(-1, 3)


def rgbd_to_pointcloud(color_img, depth_img) -> PointCloud:
    """utils/io.py:23-43: points = XYZ, colours = rgb/255, drop every pixel with any zero coordinate."""
    depth = np.asarray(depth_img)
    if depth.dtype != np.int16:
        depth = depth.astype(np.int16)
    (pts, col, _), = ops.rgbd_compact(depth.reshape(-1, 3), np.asarray(color_img, dtype=np.uint8).reshape(-1, 3), 1,
                                      color_mask=False, depth_gate=False, want_idx=False)
    return PointCloud._make(pts.clone(), col.clone())
