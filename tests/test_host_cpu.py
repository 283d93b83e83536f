"""CPU suite: host-side logic of the drop-in modules (no device work)."""
import json
import os

import numpy as np
import pytest

KAT = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ref_kat.json")))
NORM_KAT = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ref_norm_kat.json")))


def test_equation_plane_matches_reference_kat(capsys):
    from kinectpy_amd.floor_removal import equation_plane
    for c in KAT["equation_plane"]:
        assert np.allclose(equation_plane(*c["p"]), c["abcd"], rtol=0, atol=0)
    assert "equation of plane is" in capsys.readouterr().out          # the reference prints it


def test_kalman_filter_matches_reference_kat():
    from kinectpy_amd.preprocessing.filtering import kalman_filter
    for c in KAT["kalman_filter"]:
        assert np.array_equal(kalman_filter(np.array(c["x"]), **c["kw"]), np.array(c["y"]))


def test_find_delay_master_sub_matches_reference_kat():
    """utils/processing.py:23-51 (SURVEY KAT7), DataFrame and plain-dict inputs"""
    import pandas as pd
    from kinectpy_amd.utils.processing import find_delay_master_sub
    for c in NORM_KAT["find_delay_master_sub"]:
        assert find_delay_master_sub(pd.DataFrame(c["table"])) == c["out"]
        assert find_delay_master_sub(c["table"]) == c["out"]


def test_load_depth_round_trip(tmp_path):
    from kinectpy_amd.utils.io import load_depth
    arr = np.random.default_rng(0).integers(-3000, 6000, size=(368640, 3)).astype(np.int16)
    fp = tmp_path / "123_depth.dat"
    arr.tofile(fp)
    for name in (str(fp), str(tmp_path / "123")):
        a = load_depth(name)
        assert a.shape == (368640, 3) and a.dtype == np.int16 and np.array_equal(a, arr)


def test_api_signatures_match_the_reference():
    """names, argument order and defaults of SURVEY.md 8b"""
    import inspect
    from kinectpy_amd import floor_removal
    from kinectpy_amd.preprocessing import extractor, filtering, registration
    sig = lambda f: [(p.name, p.default) for p in inspect.signature(f).parameters.values()]
    assert sig(filtering.filter_outliers) == [("pcd", inspect._empty), ("nb_neighbors", 200), ("std_ratio", 3.0), ("voxel_size", 0.02)]
    assert sig(filtering.kalman_filter) == [("joint_vals", inspect._empty), ("ri", 10), ("qi", 10), ("fi", 1 / 30), ("hi", 1)]
    assert sig(registration.preprocess_point_cloud)[:4] == [("pcd", inspect._empty), ("voxel_size", inspect._empty), ("normals_nn", 30), ("fpfh_nn", 100)]
    assert sig(registration.prepare_dataset)[:5] == [("pcd_master", inspect._empty), ("pcd_sub", inspect._empty), ("voxel_size", inspect._empty), ("normals_nn", 40), ("fpfh_nn", 40)]
    assert sig(registration.execute_global_registration)[:4] == [("pcd_master", inspect._empty), ("pcd_sub", inspect._empty), ("voxel_size", 35), ("ransac_n_trials", 15)]
    assert sig(registration.execute_point_to_plane_registration) == [("pcd_master", inspect._empty), ("pcd_sub", inspect._empty), ("initial_transformation", inspect._empty), ("voxel_size", 35)]
    assert [n for n, _ in sig(floor_removal.pcd_above_plane)] == ["a", "b", "c", "d", "pcd"]
    assert [n for n, _ in sig(extractor.MKVFilesProcessing.__init__)][:5] == ["self", "mkv_fps", "output_dirs", "offline_processor_fp", "number_of_joints"]
    assert sig(extractor.MKVFilesProcessing.extract)[:5] == [("self", inspect._empty), ("color", False), ("depth", False), ("skeleton", False), ("pointcloud", False)]


def test_extractor_error_behaviour(tmp_path):
    from kinectpy_amd.preprocessing.extractor import MKVFilesProcessing
    with pytest.raises(FileNotFoundError):
        MKVFilesProcessing(["a.mkv"], [str(tmp_path / "o")], offline_processor_fp=str(tmp_path / "missing.exe"))
    src = lambda fp: (None, [])
    with pytest.raises(Exception, match="two lists"):
        MKVFilesProcessing(["a.mkv"], [], frame_source=src)
    m = MKVFilesProcessing(["a.mkv"], [str(tmp_path / "master_1")], frame_source=src)
    for sub in ("color", "depths", "pointclouds", "skeleton", "filtered_pointclouds", "filtered_and_registered_pointclouds"):
        assert (tmp_path / "master_1" / sub).is_dir()
    with pytest.raises(NotImplementedError, match="depth images"):
        m.extract(depth=True)


def test_synthetic_scene_is_deterministic():
    from kinectpy_amd.utils import synth
    xy = synth.xy_table()
    a, b = synth.render_depth(xy=xy), synth.render_depth(xy=xy)
    assert np.array_equal(a, b) and a.dtype == np.uint16 and a.size == 576 * 640
    assert 0.15 < (a == 0).mean() < 0.45
    E = synth.camera_pose(1, 4)
    assert np.allclose(E[:3, :3] @ E[:3, :3].T, np.eye(3)) and abs(np.linalg.norm(E[:3, 3]) - 2500) < 1e-9


# ---- on-disk formats: the PCD v0.7 binary layout Open3D writes, field by field ------------------------
def test_pcd_header_fields_and_rgb_packing():
    """preprocessing/data.py:69 / floor_removal.py:61,78 exchange clouds through o3d.io: header text, field order (normals before
    rgb), SIZE/TYPE/COUNT, WIDTH = POINTS, HEIGHT 1, VIEWPOINT, DATA binary, float32 records, rgb = (r<<16)|(g<<8)|b as float bits"""
    from kinectpy_amd import pcd_io
    pts = np.array([[1.5, -2.25, 3.0], [1e3, 2e3, -3e3]])
    nrm = np.array([[0.0, 0.0, 1.0], [0.6, 0.8, 0.0]])
    col = np.array([[1.0, 0.5, 0.0], [0.2, 1.7, -0.3]])                 # out-of-range channels clamp
    raw = pcd_io.encode_pcd(pts, nrm, col)
    want = (b"# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z normal_x normal_y normal_z rgb\n"
            b"SIZE 4 4 4 4 4 4 4\nTYPE F F F F F F F\nCOUNT 1 1 1 1 1 1 1\nWIDTH 2\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS 2\nDATA binary\n")
    assert raw.startswith(want) and len(raw) == len(want) + 2 * 7 * 4
    rec = np.frombuffer(raw[len(want):], dtype="<f4").reshape(2, 7)
    assert np.array_equal(rec[:, :3], pts.astype(np.float32)) and np.array_equal(rec[:, 3:6], nrm.astype(np.float32))
    packed = np.ascontiguousarray(rec[:, 6]).view(np.uint32)
    assert packed[0] == (255 << 16) | (128 << 8) | 0                   # round(0.5 * 255) = 128
    assert packed[1] == (51 << 16) | (255 << 8) | 0
    assert pcd_io.pcd_header(["x", "y", "z"], 0).endswith("WIDTH 0\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS 0\nDATA binary\n")
    p, n, c = pcd_io.decode_pcd(raw)
    assert np.array_equal(p, pts.astype(np.float32).astype(np.float64)) and np.array_equal(n, nrm.astype(np.float32).astype(np.float64))
    assert np.array_equal(c, np.array([[255, 128, 0], [51, 255, 0]]) / 255.0)
    p, n, c = pcd_io.decode_pcd(pcd_io.encode_pcd(pts))
    assert n is None and c is None and len(p) == 2


def test_pcd_reader_accepts_other_writers():
    """a PCL-style file of the same cloud: fields in another order, an extra field, rgb as TYPE U, float64 x; and DATA ascii"""
    from kinectpy_amd import pcd_io
    hdr = (b"# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS rgb intensity x y z\nSIZE 4 4 8 4 4\nTYPE U F F F F\n"
           b"COUNT 1 1 1 1 1\nWIDTH 2\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS 2\nDATA binary\n")
    rec = np.zeros(2, dtype=[("rgb", "<u4"), ("i", "<f4"), ("x", "<f8"), ("y", "<f4"), ("z", "<f4")])
    rec["rgb"] = [(10 << 16) | (20 << 8) | 30, 0xFFFFFF]
    rec["x"], rec["y"], rec["z"] = [1.25, -7.0], [2.0, 8.0], [3.0, 9.0]
    p, n, c = pcd_io.decode_pcd(hdr + rec.tobytes())
    assert np.array_equal(p, [[1.25, 2.0, 3.0], [-7.0, 8.0, 9.0]]) and n is None
    assert np.array_equal(c, np.array([[10, 20, 30], [255, 255, 255]]) / 255.0)
    txt = (b"VERSION .7\nFIELDS x y z rgb\nSIZE 4 4 4 4\nTYPE F F F F\nCOUNT 1 1 1 1\nWIDTH 2\nHEIGHT 1\nPOINTS 2\nDATA ascii\n"
           b"0.5 1.5 2.5 4.2108e+06\n-1 -2 -3 0\n")
    p, n, c = pcd_io.decode_pcd(txt)
    assert np.array_equal(p, [[0.5, 1.5, 2.5], [-1, -2, -3]]) and c.shape == (2, 3)
    with pytest.raises(RuntimeError):
        pcd_io.decode_pcd(hdr.replace(b"DATA binary", b"DATA binary_compressed"))


def test_sort_filenames_by_timestamp_kat():
    """SURVEY 8c KAT6 (captured from the reference): the key is the integer of all digits in the name"""
    from kinectpy_amd.utils.processing import sort_filenames_by_timestamp
    assert list(sort_filenames_by_timestamp(["10_rgb.png", "9_rgb.png", "0100_rgb.png"])) == ["9_rgb.png", "10_rgb.png", "0100_rgb.png"]
