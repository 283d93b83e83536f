"""Generates tests/golden/ref_norm_kat.json: the NumPy-only arithmetic of the reference's normalisers
(utils/processing.py:313-354, utils/normalization.py:16-159) and find_delay_master_sub (utils/processing.py:23-51),
executed by importing the reference in the build container:

    PYTHONDONTWRITEBYTECODE=1 python -B tests/golden/make_ref_norm_kat.py

Open3D is absent, so `o3d.geometry.PointCloud(...).get_oriented_bounding_box()` is a STUB that returns a box
(R, centre, extent) chosen here and recorded with the vector; `get_rotation_matrix_from_yxz` returns a recorded
matrix too.  What the vectors pin is therefore exactly what the reference computes itself -- the affine steps
around the box -- not Open3D's box.  tensorflow (only its decorators are touched at import) is a MagicMock.
The output is data (inputs + expected outputs), never reference source.
"""
import json
import os
import sys
import warnings
from unittest import mock

import numpy as np

REF = "/root/reference"
sys.dont_write_bytecode = True
warnings.simplefilter("ignore", FutureWarning)


class _Box:
    def __init__(self, R, c, ext, M):
        self.R, self._c, self.extent, self._M = R, c, ext, M

    def get_center(self):
        return self._c

    def get_rotation_matrix_from_yxz(self, rot):
        assert np.allclose(rot, [0, np.pi, 0])          # the only call in the reference (utils/normalization.py:42)
        return self._M


class _Cloud:
    boxes = []          # boxes handed out in call order

    def __init__(self, points=None):
        self.points = np.asarray(points)

    def get_oriented_bounding_box(self):
        return _Cloud.boxes.pop(0)


o3d = mock.MagicMock()
o3d.geometry.PointCloud = _Cloud
o3d.utility.Vector3dVector = lambda a: np.asarray(a)
sys.modules["open3d"] = o3d
for name in ["cv2", "tensorflow", "PIL", "PIL.Image", "PyMoCapViewer", "imghdr", "utils.visualization", "datasets",
             "datasets.kinect_dataset"]:
    sys.modules.setdefault(name, mock.MagicMock())
sys.path.insert(0, REF)

import pandas as pd  # noqa: E402
from utils import normalization as ref_norm  # noqa: E402
from utils import processing as ref_processing  # noqa: E402


def _rot(rng):
    q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
    return q * np.sign(np.linalg.det(q))


def main():
    rng = np.random.default_rng(20250203)
    out = {}

    # normalize_pointcloud (utils/processing.py:313-324): in place, returns the cloud
    cases = []
    for lo, hi in ((-1.0, 1.0), (0.0, 255.0)):
        pts = rng.normal(scale=800, size=(40, 3)).round(2)
        c = _Cloud(pts.copy())
        r = ref_processing.normalize_pointcloud(c, lo, hi)
        cases.append({"pts": pts.tolist(), "min_range": lo, "max_range": hi, "out": np.asarray(r.points).tolist()})
    out["normalize_pointcloud"] = cases

    # obb_normalization (utils/processing.py:327-354): (p - centre) @ R for points and joints
    pts = rng.normal(scale=600, size=(50, 3))
    joints = pd.Series(rng.normal(scale=500, size=12))
    R, c, ext = _rot(rng), rng.normal(scale=300, size=3), np.abs(rng.normal(scale=900, size=3)) + 10
    _Cloud.boxes = [_Box(R, c, ext, None)]
    xo, jo = ref_processing.obb_normalization(pts, joints, 4)
    out["obb_normalization"] = [{"pts": pts.tolist(), "joints": joints.values.tolist(), "number_of_joints": 4,
                                 "R": R.tolist(), "centre": c.tolist(), "extent": ext.tolist(),
                                 "points_out": np.asarray(xo).tolist(), "joints_out": np.asarray(jo).tolist()}]

    # the batch normalisers (utils/normalization.py:16-126)
    B, N, K = 3, 24, 5
    x = rng.normal(scale=700, size=(B, N, 3))
    y = rng.normal(scale=500, size=(B, 3 * K))
    boxes = [(_rot(rng), rng.normal(scale=300, size=3), np.abs(rng.normal(scale=900, size=3)) + 10) for _ in range(B)]
    M = _rot(rng)                # stands for get_rotation_matrix_from_yxz([0, pi, 0]); recorded
    batch = {"x": x.tolist(), "y": y.tolist(), "M": M.tolist(),
             "boxes": [{"R": b[0].tolist(), "centre": b[1].tolist(), "extent": b[2].tolist()} for b in boxes]}
    for name in ("obb_normalization_batch", "obb_rotation_translation_batch", "translation_normalization_batch"):
        _Cloud.boxes = [_Box(b[0], b[1], b[2], M) for b in boxes]
        gx, gy = getattr(ref_norm, name)(x.copy(), y.copy())
        batch[name] = {"x": np.asarray(gx).tolist(), "y": np.asarray(gy).tolist()}
    _Cloud.boxes = [_Box(boxes[0][0], boxes[0][1], boxes[0][2], M)]
    gx, gy = ref_norm.obb_normalization_batch(x[0].copy(), y[:1].copy())      # the 2-D input branch (:31-32)
    batch["obb_normalization_batch_2d"] = {"x": np.asarray(gx).tolist(), "y": np.asarray(gy).tolist()}
    gx, gy = ref_norm.scale_batch(x.copy(), y.copy())
    batch["scale_batch"] = {"x": np.asarray(gx).tolist(), "y": np.asarray(gy).tolist()}
    gx, gy = ref_norm.scale_batch(x.copy(), y.copy(), scale=2.5)
    batch["scale_batch_2p5"] = {"x": np.asarray(gx).tolist(), "y": np.asarray(gy).tolist()}
    with mock.patch.object(np.random, "randint", lambda n: 217):
        gx, gy = ref_norm.rotate_batch(x.copy(), y.copy())
    batch["rotate_batch"] = {"degs": 217, "x": np.asarray(gx).tolist(), "y": np.asarray(gy).tolist()}
    out["batch"] = batch

    # find_delay_master_sub (utils/processing.py:23-51; SURVEY KAT7)
    tables = [
        {"master_1": [1, 4, 7, 10], "sub_1": [5, 8, 11, 14], "sub_2": [1, 4, 7, 10]},
        {"master_1": [100, 133, 166, 200, 233], "sub_1": [150, 183, 216, 250, 283], "sub_2": [240, 270, 300, 330, 360],
         "sub_3": [90, 120, 150, 180, 210]},
        {"master_1": [10, 20, 30], "sub_1": [15, 25, 35]},        # a tie: the first (strictly smaller) index wins
    ]
    out["find_delay_master_sub"] = [{"table": t, "out": [int(v) for v in ref_processing.find_delay_master_sub(pd.DataFrame(t))]}
                                    for t in tables]

    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_norm_kat.json")
    with open(dst, "w") as f:
        json.dump(out, f)
    print("wrote", dst, os.path.getsize(dst), "bytes")


if __name__ == "__main__":
    main()
