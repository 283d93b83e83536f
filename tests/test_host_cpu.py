"""CPU suite: host-side logic of the drop-in modules (no device work)."""
import json
import os

import numpy as np
import pytest

KAT = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ref_kat.json")))


def test_equation_plane_matches_reference_kat(capsys):
    from kinectpy_amd.floor_removal import equation_plane
    for c in KAT["equation_plane"]:
        assert np.allclose(equation_plane(*c["p"]), c["abcd"], rtol=0, atol=0)
    assert "equation of plane is" in capsys.readouterr().out          # the reference prints it


def test_kalman_filter_matches_reference_kat():
    from kinectpy_amd.preprocessing.filtering import kalman_filter
    for c in KAT["kalman_filter"]:
        assert np.array_equal(kalman_filter(np.array(c["x"]), **c["kw"]), np.array(c["y"]))


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
