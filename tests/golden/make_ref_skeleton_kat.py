"""Generates tests/golden/ref_skeleton_fusion.json by importing the reference's fuse_skeletons_gradient
(utils/skeleton_fusion.py:21-74, NumPy-only arithmetic).

Run ONLY in the build container (needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python -B tests/golden/make_ref_skeleton_kat.py

open3d / PyMoCapViewer are absent here and replaced by MagicMock entries in sys.modules.  The output is data
(inputs + expected outputs), never reference source.
"""
import json
import os
import sys
from unittest import mock

import numpy as np

sys.dont_write_bytecode = True
for name in ["open3d", "cv2", "tensorflow", "PyMoCapViewer"]:
    sys.modules.setdefault(name, mock.MagicMock())
sys.path.insert(0, "/root/reference")
import matplotlib  # noqa: E402

matplotlib.use("Agg")
from utils import skeleton_fusion as ref  # noqa: E402


def main():
    rng = np.random.default_rng(20250202)
    cases = []
    for frames, joints, alpha, beta in [(40, 5, 1.4, 1.4), (33, 3, 0.0, 0.0), (25, 4, 2.0, 0.0), (30, 2, 0.0, 1.0), (21, 1, 0.7, 2.5)]:
        truth = np.cumsum(rng.normal(scale=5.0, size=(frames, joints, 3)), axis=0) + rng.normal(scale=300.0, size=(1, joints, 3))
        sk = np.stack([truth + rng.normal(scale=s, size=truth.shape) for s in (3.0, 8.0, 15.0)])
        fused = ref.fuse_skeletons_gradient(sk.copy(), alpha, beta)
        cases.append({"alpha": alpha, "beta": beta, "skeletons": sk.tolist(), "fused": np.asarray(fused).tolist()})
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_skeleton_fusion.json")
    with open(path, "w") as f:
        json.dump({"fuse_skeletons_gradient": cases}, f)
    print("wrote", path, len(cases), "cases")


if __name__ == "__main__":
    main()
