"""Binary .pcd reader/writer for the files the reference exchanges through o3d.io
(preprocessing/data.py:69 writes, floor_removal.py:61,78 reads/writes).  Host-side file I/O only
(out of the hot path); layout follows the PCD v0.7 `DATA binary` form Open3D emits for clouds with
points (+ packed rgb as a float32 field, + normals)."""
import numpy as np

from .geometry import PointCloud


def write_point_cloud(filename, pcd, write_ascii=False, compressed=False, print_progress=False):
    pts = np.asarray(pcd.points).astype(np.float32)
    n = len(pts)
    fields, cols = ["x", "y", "z"], [pts]
    if pcd.has_normals():
        fields += ["normal_x", "normal_y", "normal_z"]
        cols.append(np.asarray(pcd.normals).astype(np.float32))
    if pcd.has_colors():
        c = np.clip(np.round(np.asarray(pcd.colors) * 255.0), 0, 255).astype(np.uint32)
        packed = ((c[:, 0] << 16) | (c[:, 1] << 8) | c[:, 2]).astype(np.uint32).view(np.float32)
        fields.append("rgb")
        cols.append(packed[:, None])
    data = np.concatenate(cols, 1).astype(np.float32) if n else np.zeros((0, len(fields)), np.float32)
    hdr = ("# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\n"
           f"FIELDS {' '.join(fields)}\nSIZE {' '.join(['4'] * len(fields))}\nTYPE {' '.join(['F'] * len(fields))}\n"
           f"COUNT {' '.join(['1'] * len(fields))}\nWIDTH {n}\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS {n}\nDATA binary\n")
    with open(filename, "wb") as f:
        f.write(hdr.encode("ascii"))
        f.write(np.ascontiguousarray(data).tobytes())
    return True


def read_point_cloud(filename, format="auto", remove_nan_points=False, remove_infinite_points=False,
                     print_progress=False):
    with open(filename, "rb") as f:
        raw = f.read()
    meta, off = {}, 0
    while True:
        end = raw.index(b"\n", off)
        line = raw[off:end].decode("ascii", "replace").strip()
        off = end + 1
        if line.startswith("#") or not line:
            continue
        k, _, v = line.partition(" ")
        meta[k] = v.split()
        if k == "DATA":
            break
    fields, n = meta["FIELDS"], int(meta["POINTS"][0])
    sizes = [int(s) for s in meta["SIZE"]]
    if meta["DATA"][0] == "binary":
        if any(s != 4 for s in sizes):
            raise RuntimeError("read_point_cloud: only 4-byte fields are supported")
        data = np.frombuffer(raw, dtype=np.float32, count=n * len(fields), offset=off).reshape(n, len(fields))
    elif meta["DATA"][0] == "ascii":
        data = np.loadtxt(raw[off:].decode().splitlines(), dtype=np.float64, ndmin=2).astype(np.float32)
    else:
        raise RuntimeError("read_point_cloud: compressed PCD is not supported")
    col = {name: i for i, name in enumerate(fields)}
    pcd = PointCloud(data[:, [col["x"], col["y"], col["z"]]])
    if "normal_x" in col:
        pcd.normals = data[:, [col["normal_x"], col["normal_y"], col["normal_z"]]]
    if "rgb" in col:
        p = np.ascontiguousarray(data[:, col["rgb"]]).view(np.uint32)
        pcd.colors = np.stack([(p >> 16) & 255, (p >> 8) & 255, p & 255], 1).astype(np.float64) / 255.0
    return pcd
