"""PointCloud container exposing exactly the Open3D surface KinectPy's hot path touches
(SURVEY.md 8b / a22), backed by float32 torch-ROCm tensors (device memory containers only;
every operation is a call into libkinectpx.so through kinectpy_amd.ops).

Reference call sites: utils/io.py:28-41 (points/colors get/set through Vector3dVector),
preprocessing/data.py:46-58 (transform in place, np.asarray(points)), preprocessing/filtering.py:23-24,
floor_removal.py:50,61-73 (select_by_index with (K,1) arrays, segment_plane, `+`), registration.py:8-13.
"""
import copy

import numpy as np
import torch

from . import _lib as L
from . import ops


class Vector3dVector:
    """o3d.utility.Vector3dVector stand-in: wraps an (N,3) array; np.asarray() gives float64."""

    def __init__(self, data=None):
        if isinstance(data, Vector3dVector):
            self.t = data.t
        elif data is None:
            self.t = torch.empty((0, 3), dtype=torch.float32, device=L.device())
        elif isinstance(data, torch.Tensor):
            self.t = data.to(device=L.device(), dtype=torch.float32).reshape(-1, 3).contiguous()
        else:
            a = np.asarray(data)
            if a.size and (a.ndim != 2 or a.shape[1] != 3):
                raise RuntimeError("Vector3dVector: expected an (N, 3) array")
            self.t = torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32).reshape(-1, 3)).to(L.device())

    def __array__(self, dtype=None, copy=None):
        a = self.t.cpu().numpy().astype(np.float64)
        return a if dtype is None else a.astype(dtype)

    def __len__(self):
        return int(self.t.shape[0])

    def __getitem__(self, i):
        return np.asarray(self)[i]


class Vector2iVector:
    def __init__(self, data):
        self.a = np.ascontiguousarray(np.asarray(data), dtype=np.int32).reshape(-1, 2)

    def __array__(self, dtype=None, copy=None):
        return self.a if dtype is None else self.a.astype(dtype)

    def __len__(self):
        return len(self.a)


class KDTreeSearchParamHybrid:
    def __init__(self, radius, max_nn):
        self.radius, self.max_nn = float(radius), int(max_nn)


class KDTreeSearchParamKNN:
    def __init__(self, knn=30):
        self.radius, self.max_nn = 1e150, int(knn)


def _idx_array(indices):
    """accepts lists, (K,), (K,1) arrays (np.argwhere output, floor_removal.py:50,65-69) and tensors"""
    if isinstance(indices, torch.Tensor):
        return indices.reshape(-1).to(torch.int32)
    return np.ascontiguousarray(np.asarray(indices).reshape(-1), dtype=np.int32)


class OrientedBoundingBox:
    """the members the reference reads: R, extent, get_center(), get_rotation_matrix_from_yxz()"""

    def __init__(self, center=None, R=None, extent=None):
        self.center = np.zeros(3) if center is None else np.asarray(center, dtype=np.float64)
        self.R = np.eye(3) if R is None else np.asarray(R, dtype=np.float64)
        self.extent = np.zeros(3) if extent is None else np.asarray(extent, dtype=np.float64)

    @classmethod
    def _from_row(cls, row):
        return cls(row[9:12].copy(), row[:9].reshape(3, 3).copy(), row[12:15].copy())

    def get_center(self):
        return self.center

    @staticmethod
    def get_rotation_matrix_from_yxz(rotation):
        """Ry(rotation[0]) @ Rx(rotation[1]) @ Rz(rotation[2]) (utils/normalization.py:40)"""
        a, b, c = (float(v) for v in np.asarray(rotation, dtype=np.float64).reshape(3))
        ca, sa, cb, sb, cc, sc = np.cos(a), np.sin(a), np.cos(b), np.sin(b), np.cos(c), np.sin(c)
        Ry = np.array([[ca, 0, sa], [0, 1, 0], [-sa, 0, ca]])
        Rx = np.array([[1, 0, 0], [0, cb, -sb], [0, sb, cb]])
        Rz = np.array([[cc, -sc, 0], [sc, cc, 0], [0, 0, 1]])
        return Ry @ Rx @ Rz

    def __repr__(self):
        return f"OrientedBoundingBox: center: {tuple(self.center)}, extent: {tuple(self.extent)}"


class PointCloud:
    # _bounds: the cloud's min / max as a float64[6] DEVICE tensor once some kernel has computed them (the gather that produced the
    # cloud, or the first get_*_bound / remove_floor); dropped whenever the points change
    _bounds = None

    def __init__(self, points=None):
        self._pts = Vector3dVector(points).t
        self._col = None
        self._nrm = None
        self._bounds = None

    # ---- attributes ---------------------------------------------------------------------------
    @property
    def points(self):
        return Vector3dVector(self._pts)

    @points.setter
    def points(self, v):
        self._pts = Vector3dVector(v).t
        self._bounds = None

    @property
    def colors(self):
        return Vector3dVector(self._col if self._col is not None else None)

    @colors.setter
    def colors(self, v):
        t = Vector3dVector(v).t
        self._col = t if t.shape[0] else None

    @property
    def normals(self):
        return Vector3dVector(self._nrm if self._nrm is not None else None)

    @normals.setter
    def normals(self, v):
        t = Vector3dVector(v).t
        self._nrm = t if t.shape[0] else None

    def has_points(self):
        return self._pts.shape[0] > 0

    def has_colors(self):
        return self._col is not None and self._col.shape[0] == self._pts.shape[0] and self.has_points()

    def has_normals(self):
        return self._nrm is not None and self._nrm.shape[0] == self._pts.shape[0] and self.has_points()

    def __repr__(self):
        return f"PointCloud with {self._pts.shape[0]} points."

    @classmethod
    def _make(cls, pts, col=None, nrm=None, bounds=None):
        pc = cls.__new__(cls)
        pc._pts, pc._col, pc._nrm, pc._bounds = pts, col, nrm, bounds
        return pc

    def _device_bounds(self):
        """float64[6] device tensor (min x, y, z, max x, y, z), computed once per cloud"""
        if self._bounds is None:
            self._bounds = ops.bounds(self._pts)
        return self._bounds

    def __deepcopy__(self, memo):
        return PointCloud._make(self._pts.clone(), None if self._col is None else self._col.clone(),
                                None if self._nrm is None else self._nrm.clone())

    __copy__ = lambda self: self.__deepcopy__({})

    def _attrs(self):
        return [self._pts, self._col if self.has_colors() else None, self._nrm if self.has_normals() else None]

    # ---- Open3D methods on the path -----------------------------------------------------------
    def get_min_bound(self):
        if not self.has_points():
            return np.zeros(3)
        return self._device_bounds()[:3].cpu().numpy()

    def get_max_bound(self):
        if not self.has_points():
            return np.zeros(3)
        return self._device_bounds()[3:].cpu().numpy()

    def transform(self, T):
        """in place, returns self (preprocessing/data.py:48)"""
        T = np.asarray(T, dtype=np.float64)
        if T.shape != (4, 4):
            raise RuntimeError("transform: expected a 4x4 matrix")
        if self.has_points():
            self._bounds = None
            ops.transform(self._pts, T, out=self._pts)
            if self.has_normals():
                ops.rotate(self._nrm, T, out=self._nrm)
        return self

    def select_by_index(self, indices, invert=False):
        idx = _idx_array(indices)
        if not self.has_points():
            return PointCloud()
        p, c, n = ops.select_by_index(self._attrs(), idx, invert)
        return PointCloud._make(p, c, n)

    def _select(self, idx):
        """select_by_index for an index list produced by this library (ascending, duplicate-free, in range): one gather"""
        (p, c, n), bb = ops.select_by_index(self._attrs(), idx, False, trusted=True, want_bounds=True)
        return PointCloud._make(p, c, n, bb)

    def voxel_down_sample(self, voxel_size):
        if not voxel_size > 0:
            raise RuntimeError("voxel_size <= 0.")
        p, c, n = ops.voxel_downsample(self._pts, float(voxel_size), self._col if self.has_colors() else None,
                                       self._nrm if self.has_normals() else None)
        return PointCloud._make(p, c, n)

    def remove_statistical_outlier(self, nb_neighbors, std_ratio, print_progress=False):
        if nb_neighbors < 1 or std_ratio <= 0:
            raise RuntimeError("Illegal input parameters, the number of neighbors and standard deviation ratio must be positive.")
        if not self.has_points():
            return PointCloud(), np.zeros(0, dtype=np.int32)
        idx, _, _ = ops.sor(self._pts, int(nb_neighbors), float(std_ratio))
        (p, c, n), bb = ops.select_by_index(self._attrs(), idx, False, trusted=True, want_bounds=True)
        return PointCloud._make(p, c, n, bb), idx.cpu().numpy()

    def segment_plane(self, distance_threshold, ransac_n, num_iterations, probability=0.99999999, seed=None):
        """seed: the reference's RANSAC is unseeded (floor_removal.py:70); ours draws from Philox with
        the given seed (None -> a fresh random seed, i.e. the reference's behaviour)."""
        if seed is None:
            seed = int(np.random.SeedSequence().generate_state(1, dtype=np.uint64)[0])
        plane, idx = ops.segment_plane(self._pts, float(distance_threshold), int(ransac_n), int(num_iterations),
                                       float(probability), int(seed))
        return plane, idx.cpu().numpy()

    def estimate_normals(self, search_param=None, fast_normal_computation=True):
        sp = search_param if search_param is not None else KDTreeSearchParamKNN()
        if self.has_points():
            self._nrm = ops.estimate_normals(self._pts, sp.radius, sp.max_nn)       # any max_nn up to KPX_NORMALS_MAX_NN (beyond 128 the fall-back heaps live in the workspace)
        return self

    def get_oriented_bounding_box(self, robust=False):
        """utils/normalization.py:39, 74, 105; utils/processing.py:341"""
        obb, _ = ops.obb_batch(self._pts)
        return OrientedBoundingBox._from_row(obb[0].cpu().numpy())

    def __add__(self, other):
        both_c = self.has_colors() and other.has_colors()
        both_n = self.has_normals() and other.has_normals()
        if not self.has_points():
            both_c, both_n = other.has_colors(), other.has_normals()
        return PointCloud._make(torch.cat([self._pts, other._pts], 0),
                                torch.cat([self._col, other._col], 0) if both_c and self.has_points() else (other._col if both_c else None),
                                torch.cat([self._nrm, other._nrm], 0) if both_n and self.has_points() else (other._nrm if both_n else None))

    def __iadd__(self, other):
        r = self + other
        self._pts, self._col, self._nrm, self._bounds = r._pts, r._col, r._nrm, None
        return self

    def clone(self):
        return copy.deepcopy(self)
