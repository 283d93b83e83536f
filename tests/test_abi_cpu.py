"""CPU suite: the C-ABI library loads without a GPU, exports every symbol include/kinectpx.h declares,
and rejects invalid arguments before touching the device (no compute calls here)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    from kinectpy_amd import _lib
    if not os.path.exists(_lib.SO_PATH):
        g.build()
    return _lib.load()


def _declared():
    hdr = open(os.path.join(ROOT, "include", "kinectpx.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(kpx_[a-z0-9_]+)\s*\(", hdr)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    from kinectpy_amd import _lib
    names = _declared()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/kinectpx.h but not exported"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature"
    assert sorted(_lib.SIGNATURES) == names


def test_ctypes_signatures_match_the_header():
    """argument count and the C type class of every parameter (pointer / 64-bit integer / 32-bit integer / double / size_t)
    of each prototype in include/kinectpx.h (size_t counts as a 64-bit integer) against kinectpy_amd/_lib.py::SIGNATURES"""
    from kinectpy_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "kinectpx.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    protos = re.findall(r"\b(?:int|size_t|const char \*)\s*\**\s*(kpx_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", hdr, flags=re.S)
    assert len(protos) == len(_lib.SIGNATURES)

    def cls(param):
        param = " ".join(param.split())
        if "*" in param or "_fn " in param:          # kpx_bcast_fn / kpx_allgather_fn: function pointers
            return "ptr"
        for key, name in (("uint64_t", "i64"), ("int64_t", "i64"), ("size_t", "i64"), ("int32_t", "i32"), ("double", "f64"), ("int ", "i32")):
            if key in param + " ":
                return name
        raise AssertionError(param)

    ctype_cls = {C.c_void_p: "ptr", C.c_int64: "i64", C.c_uint64: "i64", C.c_int32: "i32", C.c_int: "i32", C.c_double: "f64", C.c_size_t: "i64"}      # LP64: size_t is a 64-bit integer
    for name, params in protos:
        params = params.strip()
        want = [] if params in ("", "void") else [cls(q) for q in params.split(",")]
        got = [ctype_cls[t] for t in _lib.SIGNATURES[name][1]]
        assert got == want, (name, got, want)


def test_version_and_error_channel(lib):
    assert lib.kpx_version() == 100
    rc = lib.kpx_voxel_downsample(None, None, None, 10, C.c_double(0.0), None, None, None, None, None, 0, None)
    assert rc == -1 and b"voxel_size" in lib.kpx_last_error()
    rc = lib.kpx_sor(None, 10, 0, C.c_double(1.0), None, None, None, None, None, 0, None)
    assert rc == -1 and b"must be positive" in lib.kpx_last_error()
    rc = lib.kpx_sor(None, 10, 5, C.c_double(-1.0), None, None, None, None, None, 0, None)
    assert rc == -1
    rc = lib.kpx_segment_plane(None, 10, C.c_double(30.0), 2, 10, C.c_double(0.9), 0, None, None, None, None, 0, None)
    assert rc == -1 and b"ransac_n" in lib.kpx_last_error()
    rc = lib.kpx_segment_plane(None, 10, C.c_double(30.0), 30, 10, C.c_double(0.9), 0, None, None, None, None, 0, None)
    assert rc == -1 and b"at least" in lib.kpx_last_error()
    init = np.eye(4)
    rc = lib.kpx_icp(None, 5, None, None, 5, C.c_double(1.0), init.ctypes.data_as(C.c_void_p), 1, 30, C.c_double(1e-6),
                     C.c_double(1e-6), 0, None, None, None, None, 0, None)
    assert rc == -1 and b"normal" in lib.kpx_last_error()
    rc = lib.kpx_icp(None, 5, None, None, 5, C.c_double(-1.0), init.ctypes.data_as(C.c_void_p), 0, 30, C.c_double(1e-6),
                     C.c_double(1e-6), 0, None, None, None, None, 0, None)
    assert rc == -1 and b"max_correspondence_distance" in lib.kpx_last_error()


def test_comm_and_order_objects_without_a_gpu(lib):
    """the exchange layer's host objects: a callback communicator, the collectives' issue order, argument checks of the sharded
    frame step -- nothing here touches a device"""
    from kinectpy_amd import _lib
    calls = []
    b = _lib.BCAST_FN(lambda user, buf, n, root, stream: calls.append(("b", n, root)) or 0)
    g = _lib.ALLGATHER_FN(lambda user, s, r, n, stream: calls.append(("g", n)) or 0)
    h = C.c_void_p()
    assert lib.kpx_comm_create_callbacks(1, 4, C.cast(b, C.c_void_p), C.cast(g, C.c_void_p), None, C.byref(h)) == 0
    assert lib.kpx_comm_rank(h) == 1 and lib.kpx_comm_world(h) == 4
    assert lib.kpx_comm_broadcast(h, C.c_void_p(4096), 100, 0, None) == 0 and lib.kpx_comm_allgather(h, C.c_void_p(4096), C.c_void_p(8192), 64, None) == 0
    assert calls == [("b", 100, 0), ("g", 64)]
    assert lib.kpx_comm_broadcast(h, C.c_void_p(4096), 100, 7, None) == -1 and b"bad arguments" in lib.kpx_last_error()
    assert lib.kpx_comm_destroy(h) == 0
    assert lib.kpx_comm_create_callbacks(0, 2, None, None, None, C.byref(h)) == -1              # several ranks need a transport
    assert lib.kpx_comm_create_rccl(None, 0, 1, C.byref(h)) == -1
    # the order: two frames in flight, keys 3 f (broadcast) and 3 (f + depth - 1) + s
    o = C.c_void_p()
    assert lib.kpx_order_create(2, C.byref(o)) == 0
    f0, f1 = C.c_int64(), C.c_int64()
    lib.kpx_order_submit(o, C.byref(f0)); lib.kpx_order_submit(o, C.byref(f1))
    for frame, stage in ((0, 0), (1, 0), (0, 1), (0, 2)):
        lib.kpx_order_turn_begin(o, frame, stage); lib.kpx_order_turn_end(o, frame, stage)
    lib.kpx_order_block(o, 1)                                                                    # the main thread waits for frame 1: its stages may go
    lib.kpx_order_turn_begin(o, 1, 1); lib.kpx_order_turn_end(o, 1, 1)
    lib.kpx_order_finish(o, 1)
    log, n = (C.c_int64 * 16)(), C.c_int64()
    lib.kpx_order_log(o, log, 16, C.byref(n))
    assert list(log[: n.value]) == [0, 3, 4, 5, 7]
    lib.kpx_order_destroy(o)
    # the sharded frame step refuses to run without a communicator / with more ranks than sensors
    assert lib.kpx_frame_step_sharded_workspace_bytes(4, 0, 8, 368640, 0) == 0
    assert lib.kpx_frame_step_sharded_workspace_bytes(8, 7, 8, 368640, 1) > lib.kpx_frame_step_sharded_workspace_bytes(8, 7, 8, 368640, 0) > 0
    rc = lib.kpx_frame_step_sharded(None, None, 0, None, None, 0, None, 368640, 4, None, None, 0, None, None, None, None, None, None, 0, None)
    assert rc == -1 and b"communicator" in lib.kpx_last_error()


def test_workspace_queries_are_pure_host_arithmetic(lib):
    assert lib.kpx_median_workspace_bytes(4) >= 4 * 512 * 4
    assert lib.kpx_compact_workspace_bytes(368640, 8) >= 8 * 180 * 4
    assert lib.kpx_select_workspace_bytes(1000) > 1000
    small, big = lib.kpx_nn_workspace_bytes(1000, 1000), lib.kpx_nn_workspace_bytes(100000, 100000)
    assert 0 < small < big
    assert lib.kpx_icp_workspace_bytes(100000, 100000) == big
    assert lib.kpx_segment_plane_workspace_bytes(100000, 30, 2000) > 2000 * 30 * 4


def test_product_refuses_to_run_without_gpu():
    import torch
    from kinectpy_amd import _lib, ops
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.KinectPxError, match="no CPU fallback"):
        ops.transform(np.zeros((4, 3), np.float32), np.eye(4))


def test_product_never_imports_the_oracle():
    """the oracle is test infrastructure: nothing under kinectpy_amd/ or include/ may reference it"""
    for d, _, files in os.walk(os.path.join(ROOT, "kinectpy_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(d, f)).read()
                assert "kpx_oracle" not in src and "kpo_" not in src and "from oracle" not in src and "import oracle" not in src, f
