"""Binary .pcd reader/writer for the files the reference exchanges through o3d.io
(preprocessing/data.py:69 writes, floor_removal.py:61,78 reads/writes).  Host-side file I/O only (out of the hot path).

What Open3D writes for a legacy PointCloud (io/file_format/FilePCD.cpp [O3D, recalled]; `write_point_cloud(fp, pcd)` with
its defaults write_ascii=False, compressed=False), and what write_point_cloud below emits byte for byte:

    # .PCD v0.7 - Point Cloud Data file format
    VERSION 0.7
    FIELDS x y z [normal_x normal_y normal_z] [rgb]        normals before colour
    SIZE 4 4 4 ...                                          every field a 4-byte
    TYPE F F F ...                                          float32 -- Open3D narrows its float64 arrays on write
    COUNT 1 1 1 ...
    WIDTH <n>
    HEIGHT 1
    VIEWPOINT 0 0 0 1 0 0 0
    POINTS <n>
    DATA binary
    <n records of len(FIELDS) float32, little endian>

`rgb` is ONE float32 field whose 32 bits are the integer (r << 16) | (g << 8) | b, each channel
round(clamp(c, 0, 1) * 255).  The reader also accepts what other PCD writers produce for the same cloud: fields in any
order, extra fields (skipped), `rgb`/`rgba` of TYPE U or I, SIZE 8 (float64) coordinates, and `DATA ascii`."""
import numpy as np

_NP = {("F", 4): np.float32, ("F", 8): np.float64, ("U", 1): np.uint8, ("U", 2): np.uint16, ("U", 4): np.uint32,
       ("I", 1): np.int8, ("I", 2): np.int16, ("I", 4): np.int32, ("U", 8): np.uint64, ("I", 8): np.int64}


def pcd_header(fields, n):
    """the header text of a binary PCD with `fields` float32 columns and n points, exactly as Open3D prints it"""
    k = len(fields)
    return ("# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\n"
            f"FIELDS {' '.join(fields)}\nSIZE {' '.join(['4'] * k)}\nTYPE {' '.join(['F'] * k)}\n"
            f"COUNT {' '.join(['1'] * k)}\nWIDTH {n}\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS {n}\nDATA binary\n")


def pack_rgb(colors):
    """(n,3) colours in [0,1] -> the float32 `rgb` column ((r<<16)|(g<<8)|b bit pattern)"""
    c = np.round(np.clip(np.asarray(colors, dtype=np.float64), 0.0, 1.0) * 255.0).astype(np.uint32)
    return ((c[:, 0] << 16) | (c[:, 1] << 8) | c[:, 2]).astype(np.uint32).view(np.float32)


def unpack_rgb(column):
    """the `rgb` column (float32 bit pattern, or an integer column) -> (n,3) float64 colours in [0,1]"""
    col = np.ascontiguousarray(column)
    p = col.view(np.uint32) if col.dtype == np.float32 else col.astype(np.uint32)
    return np.stack([(p >> 16) & 255, (p >> 8) & 255, p & 255], 1).astype(np.float64) / 255.0


def encode_pcd(points, normals=None, colors=None) -> bytes:
    pts = np.asarray(points, dtype=np.float64).reshape(-1, 3).astype(np.float32)
    n = len(pts)
    fields, cols = ["x", "y", "z"], [pts]
    if normals is not None:
        fields += ["normal_x", "normal_y", "normal_z"]
        cols.append(np.asarray(normals, dtype=np.float64).reshape(-1, 3).astype(np.float32))
    if colors is not None:
        fields.append("rgb")
        cols.append(pack_rgb(np.asarray(colors).reshape(-1, 3))[:, None])
    data = np.concatenate(cols, 1).astype(np.float32) if n else np.zeros((0, len(fields)), np.float32)
    return pcd_header(fields, n).encode("ascii") + np.ascontiguousarray(data).tobytes()


def decode_pcd(raw: bytes):
    """-> (points (n,3) float64, normals (n,3) float64 | None, colors (n,3) float64 | None)"""
    meta, off = {}, 0
    while True:
        end = raw.index(b"\n", off)
        line = raw[off:end].decode("ascii", "replace").strip()
        off = end + 1
        if line.startswith("#") or not line:
            continue
        k, _, v = line.partition(" ")
        meta[k.upper()] = v.split()
        if k.upper() == "DATA":
            break
    fields = meta["FIELDS"]
    n = int(meta["POINTS"][0]) if "POINTS" in meta else int(meta["WIDTH"][0]) * int(meta.get("HEIGHT", ["1"])[0])
    sizes = [int(s) for s in meta.get("SIZE", ["4"] * len(fields))]
    types = meta.get("TYPE", ["F"] * len(fields))
    counts = [int(c) for c in meta.get("COUNT", ["1"] * len(fields))]
    kind = meta["DATA"][0].lower()
    cols = {}
    if kind == "binary":
        dt = []
        for name, t, s, c in zip(fields, types, sizes, counts):
            if (t.upper(), s) not in _NP:
                raise RuntimeError(f"read_point_cloud: unsupported field {name} (TYPE {t} SIZE {s})")
            dt.append((name, np.dtype(_NP[(t.upper(), s)]).newbyteorder("<"), (c,)) if c != 1 else (name, np.dtype(_NP[(t.upper(), s)]).newbyteorder("<")))
        rec = np.frombuffer(raw, dtype=np.dtype(dt), count=n, offset=off)
        cols = {name: rec[name] for name in fields}
    elif kind == "ascii":
        rows = [ln.split() for ln in raw[off:].decode("ascii", "replace").splitlines() if ln.strip()][:n]
        pos = 0
        for name, t, s, c in zip(fields, types, sizes, counts):
            vals = [r[pos] for r in rows]
            if t.upper() == "F":
                cols[name] = np.array(vals, dtype=np.float64).astype(_NP[("F", s)])
            else:
                cols[name] = np.array([int(v) for v in vals], dtype=_NP[(t.upper(), s)])
            pos += c
    else:
        raise RuntimeError("read_point_cloud: compressed PCD is not supported")
    for need in ("x", "y", "z"):
        if need not in cols:
            raise RuntimeError("read_point_cloud: the file has no x / y / z fields")
    pts = np.stack([cols["x"], cols["y"], cols["z"]], 1).astype(np.float64)
    nrm = np.stack([cols["normal_x"], cols["normal_y"], cols["normal_z"]], 1).astype(np.float64) if "normal_x" in cols else None
    key = "rgb" if "rgb" in cols else ("rgba" if "rgba" in cols else None)
    col = unpack_rgb(cols[key]) if key else None
    return pts, nrm, col


def write_point_cloud(filename, pcd, write_ascii=False, compressed=False, print_progress=False):
    """o3d.io.write_point_cloud(filename, pcd) (preprocessing/data.py:69, floor_removal.py:78); binary only"""
    if write_ascii or compressed:
        raise NotImplementedError("write_point_cloud: the reference writes with Open3D's defaults (binary, uncompressed)")
    with open(filename, "wb") as f:
        f.write(encode_pcd(np.asarray(pcd.points), np.asarray(pcd.normals) if pcd.has_normals() else None,
                           np.asarray(pcd.colors) if pcd.has_colors() else None))
    return True


def read_point_cloud(filename, format="auto", remove_nan_points=False, remove_infinite_points=False,
                     print_progress=False):
    """o3d.io.read_point_cloud(filename) (floor_removal.py:61)"""
    from .geometry import PointCloud
    with open(filename, "rb") as f:
        pts, nrm, col = decode_pcd(f.read())
    keep = np.ones(len(pts), dtype=bool)
    if remove_nan_points:
        keep &= ~np.isnan(pts).any(1)
    if remove_infinite_points:
        keep &= ~np.isinf(pts).any(1)
    pcd = PointCloud(pts[keep])
    if nrm is not None:
        pcd.normals = nrm[keep]
    if col is not None:
        pcd.colors = col[keep]
    return pcd
