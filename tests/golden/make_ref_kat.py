"""Generates tests/golden/ref_kat.json by importing the reference's pure-NumPy functions.

Run ONLY in the build container (needs /root/reference; use `python -B` so no bytecode is written
into the read-only tree):

    PYTHONDONTWRITEBYTECODE=1 python -B tests/golden/make_ref_kat.py

open3d / cv2 / tensorflow / PyMoCapViewer are absent here, so they are replaced by MagicMock
entries in sys.modules; only functions whose arithmetic is NumPy-only are executed
(SURVEY.md 8c).  The output is data (inputs + expected outputs), never reference source.
"""
import json
import os
import sys
import tempfile
from unittest import mock

import numpy as np

REF = "/root/reference"
sys.dont_write_bytecode = True
for name in ["open3d", "cv2", "tensorflow", "PIL", "PIL.Image", "PyMoCapViewer", "imghdr"]:
    sys.modules.setdefault(name, mock.MagicMock())
sys.path.insert(0, REF)

import pandas as pd  # noqa: E402
import floor_removal as ref_floor  # noqa: E402
from preprocessing import filtering as ref_filtering  # noqa: E402
from utils import io as ref_io  # noqa: E402
from utils import processing as ref_processing  # noqa: E402


class _Cloud:
    """duck-typed cloud recording what pcd_above_plane asks select_by_index for"""

    def __init__(self, pts):
        self.points = np.asarray(pts, dtype=np.float64)
        self.selected = None

    def select_by_index(self, idx, invert=False):
        self.selected = (np.asarray(idx).tolist(), bool(invert))
        return self


def main():
    out = {}
    rng = np.random.default_rng(20250202)

    # KAT1 equation_plane (floor_removal.py:21-36)
    cases = [((1, 2, 3), (4, -1, 2), (0.5, 0.25, -7))]
    for _ in range(4):
        p = rng.normal(scale=100, size=(3, 3))
        cases.append(tuple(map(tuple, p.tolist())))
    out["equation_plane"] = [
        {"p": [list(map(float, q)) for q in c], "abcd": list(map(float, ref_floor.equation_plane(*c)))}
        for c in cases
    ]

    # KAT2 pcd_above_plane (floor_removal.py:39-51)
    pts = [[0, 0, 1], [0, 0, -1], [0, 0, 0], [5, 5, -0.5]]
    c = _Cloud(pts)
    ref_floor.pcd_above_plane(0, 0, 1, 0, c)
    out["pcd_above_plane"] = [{"abcd": [0, 0, 1, 0], "pts": pts, "idx": c.selected[0], "invert": c.selected[1]}]
    pts2 = rng.normal(scale=50, size=(64, 3)).round(3)
    abcd = [0.3, -0.8, 0.52, 11.0]
    c = _Cloud(pts2)
    ref_floor.pcd_above_plane(*abcd, c)
    out["pcd_above_plane"].append({"abcd": abcd, "pts": pts2.tolist(), "idx": c.selected[0], "invert": c.selected[1]})

    # KAT3 kalman_filter (preprocessing/filtering.py:98-129)
    x = np.random.default_rng(1234).normal(scale=100, size=(6, 3))
    out["kalman_filter"] = [
        {"x": x.tolist(), "kw": {}, "y": ref_filtering.kalman_filter(x).tolist()},
        {"x": x.tolist(), "kw": {"ri": 3, "qi": 0.5, "fi": 0.9, "hi": 1.1},
         "y": ref_filtering.kalman_filter(x, ri=3, qi=0.5, fi=0.9, hi=1.1).tolist()},
    ]

    # KAT4 transform_joints (utils/processing.py:357-383) == align_skeletons math (extractor.py:109-116)
    th = 0.3
    T = np.eye(4)
    T[:3, :3] = [[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1]]
    T[:3, 3] = [10, -20, 30]
    df = pd.DataFrame(np.arange(12, dtype=np.float64).reshape(2, 6))
    out["transform_joints"] = [{"x": df.values.tolist(), "T": T.tolist(),
                                "y": ref_processing.transform_joints(df, T).values.tolist()}]
    sk = rng.normal(scale=500, size=(5, 96))
    T2 = np.eye(4)
    q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
    T2[:3, :3] = q
    T2[:3, 3] = [120.5, -33.25, 870.0]
    out["transform_joints"].append({"x": sk.tolist(), "T": T2.tolist(),
                                    "y": ref_processing.transform_joints(pd.DataFrame(sk), T2).values.tolist()})

    # KAT6/7 bookkeeping helpers
    names = ["10_rgb.png", "9_rgb.png", "0100_rgb.png"]
    out["sort_filenames_by_timestamp"] = [{"in": names, "out": list(ref_processing.sort_filenames_by_timestamp(names))}]

    # KAT8 load_depth (utils/io.py:15-20): round trip of an int16 (N,3) .dat
    with tempfile.TemporaryDirectory() as d:
        arr = rng.integers(-3000, 6000, size=(257, 3)).astype(np.int16)
        fp = os.path.join(d, "123_depth.dat")
        arr.tofile(fp)
        a = ref_io.load_depth(fp)
        b = ref_io.load_depth(os.path.join(d, "123"))
        out["load_depth"] = [{"shape": list(a.shape), "dtype": str(a.dtype),
                              "equal": bool((a == arr).all() and (b == arr).all())}]

    # a3/a4 mask logic: the NumPy expressions of utils/io.py:36 and preprocessing/data.py:169-171
    # executed verbatim on a small frame through the reference's DataProcessor method with
    # rgbd_to_pointcloud patched to capture its inputs (Open3D containers cannot run here).
    from preprocessing import data as ref_data  # noqa: E402

    cap = {}

    def _capture(color, depth):
        cap["color"] = np.array(color)
        cap["depth"] = np.array(depth)
        return None

    n = 96
    depth = rng.integers(0, 4000, size=(n, 3)).astype(np.int16)
    depth[rng.random(n) < 0.2] = 0
    depth[5, 0] = 0
    depth[6, 1] = 0
    color = rng.integers(0, 4, size=(n // 8, 8, 3)).astype(np.uint8) * 60
    with mock.patch.object(ref_data, "rgbd_to_pointcloud", _capture):
        ref_data.DataProcessor._transform_filtered_image_to_pointcloud(None, color.copy(), depth.copy())
    pts = cap["depth"].astype(np.float64)
    nz = (pts[:, 0] != 0) & (pts[:, 1] != 0) & (pts[:, 2] != 0)      # utils/io.py:36 on the captured arrays
    out["mask_gate_compact"] = [{
        "depth": depth.tolist(), "color": color.reshape(-1, 3).tolist(),
        "median": float(np.median(depth[:, 2])),
        "after_gate_depth": cap["depth"].tolist(), "after_gate_color": cap["color"].tolist(),
        "final_points": pts[nz].tolist(),
        "final_colors": (cap["color"].astype(np.float64).reshape(-1, 3) / 255)[nz].tolist(),
    }]

    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_kat.json")
    with open(dst, "w") as f:
        json.dump(out, f)
    print("wrote", dst, os.path.getsize(dst), "bytes")


if __name__ == "__main__":
    main()
