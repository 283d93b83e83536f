"""ctypes binding of libkinectpx.so (include/kinectpx.h).

The library is the product: if it is missing, or no MI355X is visible, the functions here raise --
there is no CPU fallback.  PyTorch-ROCm tensors are used only as the device-memory container
(allocation, streams); every computation is a call into the C ABI.
"""
import ctypes as C
import threading
import weakref
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# KPX_LIBRARY: another build of the same library (A/B runs of compile-time variants, e.g. -DKPX_ICP_WAVES=2)
SO_PATH = os.environ.get("KPX_LIBRARY") or os.path.join(_HERE, "libkinectpx.so")

_i64, _i32, _f64, _u64, _vp, _sz = C.c_int64, C.c_int32, C.c_double, C.c_uint64, C.c_void_p, C.c_size_t

# name -> (restype, argtypes); mirrors include/kinectpx.h one to one
SIGNATURES = {
    "kpx_last_error": (C.c_char_p, []),
    "kpx_version": (C.c_int, []),
    "kpx_unproject_u16": (C.c_int, [_vp, _vp, _i64, _i32, _vp, _vp]),
    "kpx_median_workspace_bytes": (_sz, [_i32]),
    "kpx_median_i16": (C.c_int, [_vp, _i64, _i64, _i32, _vp, _vp, _sz, _vp]),
    "kpx_compact_workspace_bytes": (_sz, [_i64, _i32]),
    "kpx_rgbd_compact": (C.c_int, [_vp, _vp, _i64, _i32, _i32, _vp, _f64, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "kpx_depth_to_cloud_workspace_bytes": (_sz, [_i64, _i32]),
    "kpx_depth_to_cloud": (C.c_int, [_vp, _vp, _vp, _i64, _i32, _i32, _f64, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "kpx_transform": (C.c_int, [_vp, _i64, _vp, _vp, _vp]),
    "kpx_rotate": (C.c_int, [_vp, _i64, _vp, _vp, _vp]),
    "kpx_joints_affine_f64": (C.c_int, [_vp, _i64, _vp, _vp, _vp, _vp]),
    "kpx_select_workspace_bytes": (_sz, [_i64]),
    "kpx_select_by_index": (C.c_int, [_vp, _vp, _vp, _i64, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "kpx_halfspace_select": (C.c_int, [_vp, _i64, _vp, _vp, _vp, _vp, _sz, _vp]),
    "kpx_slab_split": (C.c_int, [_vp, _i64, _f64, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "kpx_slab_split_bounded": (C.c_int, [_vp, _i64, _f64, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "kpx_select_by_index_bounds": (C.c_int, [_vp, _vp, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "kpx_bounds_workspace_bytes": (C.c_size_t, []),
    "kpx_bounds": (C.c_int, [_vp, _i64, _vp, _vp, _sz, _vp]),
    "kpx_voxel_workspace_bytes": (_sz, [_i64]),
    "kpx_voxel_downsample": (C.c_int, [_vp, _vp, _vp, _i64, _f64, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "kpx_voxel_batch_workspace_bytes": (_sz, [_i32, _vp]),
    "kpx_voxel_downsample_batch": (C.c_int, [_i32, _vp, _vp, _vp, _f64, _vp, _vp, _vp, _vp, _sz, _vp]),
    "kpx_fuse_voxel_workspace_bytes": (_sz, [_i64]),
    "kpx_fuse_voxel_downsample": (C.c_int, [_i32, _vp, _vp, _vp, _vp, _f64, _vp, _vp, _vp, _vp, _sz, _vp]),
    "kpx_sor_workspace_bytes": (_sz, [_i64, _i32]),
    "kpx_sor": (C.c_int, [_vp, _i64, _i32, _f64, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "kpx_sor_partial": (C.c_int, [_vp, _i64, _i32, _i64, _i64, _vp, _vp, _vp, _sz, _vp]),
    "kpx_sor_finish_workspace_bytes": (_sz, [_i64]),
    "kpx_sor_finish": (C.c_int, [_vp, _vp, _i64, _f64, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "kpx_normals_workspace_bytes": (_sz, [_i64, _i32]),
    "kpx_estimate_normals": (C.c_int, [_vp, _i64, _f64, _i32, _vp, _vp, _sz, _vp]),
    "kpx_segment_plane_workspace_bytes": (_sz, [_i64, _i32, _i32]),
    "kpx_segment_plane": (C.c_int, [_vp, _i64, _f64, _i32, _i32, _f64, _u64, _vp, _vp, _vp, _vp, _sz, _vp]),
    "kpx_nn_workspace_bytes": (_sz, [_i64, _i64]),
    "kpx_nn_search": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _sz, _vp]),
    "kpx_kabsch_workspace_bytes": (_sz, [_i64]),
    "kpx_kabsch": (C.c_int, [_vp, _vp, _vp, _i64, _vp, _vp, _sz, _vp]),
    "kpx_icp_workspace_bytes": (_sz, [_i64, _i64]),
    "kpx_icp": (C.c_int, [_vp, _i64, _vp, _vp, _i64, _f64, _vp, _i32, _i32, _f64, _f64, _i32, _vp, _vp, _vp, _vp, _sz, _vp]),
    "kpx_fpfh_workspace_bytes": (_sz, [_i64, _i32]),
    "kpx_fpfh": (C.c_int, [_vp, _vp, _i64, _f64, _i32, _vp, _vp, _sz, _vp]),
    "kpx_feature_nn_workspace_bytes": (_sz, [_i64, _i64]),
    "kpx_feature_nn": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _vp, _sz, _vp]),
    "kpx_ransac_workspace_bytes": (_sz, [_i64, _i64]),
    "kpx_ransac_corres": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _i64, _f64, _i32, _f64, _i32, _f64, _u64, _vp, _vp, _sz, _vp]),
    "kpx_sample_workspace_bytes": (_sz, [_i64]),
    "kpx_sample_points": (C.c_int, [_vp, _i64, _i64, _u64, _vp, _vp, _vp, _sz, _vp]),
    "kpx_obb_workspace_bytes": (_sz, [_i32, _i64]),
    "kpx_obb_batch": (C.c_int, [_vp, _i32, _i32, _i64, _vp, _vp, _vp, _sz, _vp]),
    "kpx_normalize_batch": (C.c_int, [_vp, _i32, _i64, _vp, _i32, _vp, _vp, _vp]),
    "kpx_color_gradient_workspace_bytes": (_sz, [_i64, _i32]),
    "kpx_color_gradient": (C.c_int, [_vp, _vp, _vp, _i64, _f64, _i32, _vp, _vp, _sz, _vp]),
    "kpx_colored_icp_workspace_bytes": (_sz, [_i64, _i64]),
    "kpx_colored_icp": (C.c_int, [_vp, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _f64, _vp, _f64, _i32, _f64, _f64, _i32, _vp, _vp, _sz, _vp]),
    "kpx_fuse_skeletons": (C.c_int, [_vp, _i32, _i64, _i32, _f64, _f64, _i32, _vp, _vp]),
    "kpx_nn_engine": (C.c_int, [_i32]),
    "kpx_icp_batch_workspace_bytes": (_sz, [_i32, _vp, _i64]),
    "kpx_icp_batch": (C.c_int, [_i32, _vp, _vp, _vp, _vp, _i64, _f64, _vp, _i32, _i32, _f64, _f64, _vp, _vp, _sz, _vp]),
    "kpx_prof_begin": (C.c_int, [_i32]),
    "kpx_prof_stride": (C.c_int, [_i32]),
    "kpx_frame_step_workspace_bytes": (_sz, [_i32, _i64]),
    "kpx_frame_step": (C.c_int, [_vp, _vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "kpx_frame_step_host_workspace_bytes": (_sz, [_i32, _i64]),
    "kpx_frame_step_host": (C.c_int, [_vp, _vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "kpx_rccl_load": (C.c_int, [_vp]),
    "kpx_rccl_unique_id": (C.c_int, [_vp]),
    "kpx_comm_create_rccl": (C.c_int, [_vp, _i32, _i32, _vp]),
    "kpx_comm_create_callbacks": (C.c_int, [_i32, _i32, _vp, _vp, _vp, _vp]),
    "kpx_comm_create_replay": (C.c_int, [_i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp]),
    "kpx_comm_destroy": (C.c_int, [_vp]),
    "kpx_comm_rank": (C.c_int, [_vp]),
    "kpx_comm_world": (C.c_int, [_vp]),
    "kpx_comm_broadcast": (C.c_int, [_vp, _vp, _sz, _i32, _vp]),
    "kpx_comm_allgather": (C.c_int, [_vp, _vp, _vp, _sz, _vp]),
    "kpx_copy_bytes": (C.c_int, [_vp, _vp, _sz, _vp, _i32]),
    "kpx_order_create": (C.c_int, [_i32, _vp]),
    "kpx_order_destroy": (C.c_int, [_vp]),
    "kpx_order_submit": (C.c_int, [_vp, _vp]),
    "kpx_order_block": (C.c_int, [_vp, _i64]),
    "kpx_order_turn_begin": (C.c_int, [_vp, _i64, _i32]),
    "kpx_order_turn_end": (C.c_int, [_vp, _i64, _i32]),
    "kpx_order_skip": (C.c_int, [_vp, _i64, _i32]),
    "kpx_order_finish": (C.c_int, [_vp, _i64]),
    "kpx_order_log": (C.c_int, [_vp, _vp, _i64, _vp]),
    "kpx_frame_step_sharded_workspace_bytes": (_sz, [_i32, _i32, _i32, _i64, _i32]),
    "kpx_frame_step_sharded": (C.c_int, [_vp, _vp, _i64, _vp, _vp, _i32, _vp, _i64, _i32, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "kpx_stream_workspace_bytes": (_sz, [_i32, _i32, _i32, _i64, _i32]),
    "kpx_stream_create": (C.c_int, [_vp, _i64, _i32, _vp, _vp, _i32, _vp, _i32, _vp, _sz, _vp]),
    "kpx_stream_submit": (C.c_int, [_vp, _vp, _vp, _i32, _vp, _vp]),
    "kpx_stream_pop": (C.c_int, [_vp, _vp, _vp, _vp]),
    "kpx_stream_pending": (C.c_int, [_vp]),
    "kpx_stream_capacity": (C.c_int, [_vp]),
    "kpx_stream_destroy": (C.c_int, [_vp]),
    "kpx_stream_stats": (C.c_int, [_vp, _vp]),
    "kpx_sor_select": (C.c_int, [_vp, _vp, C.c_int64, C.c_int32, C.c_double, _vp, _vp, _vp, _vp, _vp, _vp, C.c_size_t, _vp]),
    "kpx_sort_pairs_u32_workspace_bytes": (C.c_size_t, [C.c_int64]),
    "kpx_sort_pairs_u32": (C.c_int, [_vp, _vp, C.c_int64, C.c_int32, _vp, _vp, _vp, C.c_size_t, _vp]),
    "kpx_prof_icp_phases": (C.c_int, [_vp]),
    "kpx_prof_icp_waves": (C.c_int, [_vp, C.c_int64, _vp]),
    "kpx_icp_chain": (C.c_int, [C.c_int32]),
    "kpx_prof_icp_cert": (C.c_int, [_vp]),
    "kpx_prof_icp_chain": (C.c_int, [_vp]),
    "kpx_prof_end": (C.c_int, [_vp, _vp, _vp]),
}

_lib = None


class KinectPxError(RuntimeError):
    pass


def load():
    """dlopen the library (torch is imported first so both share one libamdhip64)."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise KinectPxError(
                f"{SO_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(make -C kinectpy_amd/csrc).  There is no CPU fallback.")
        lib = C.CDLL(SO_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)          # AttributeError if the .so lacks a declared symbol
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


BCAST_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int32, C.c_void_p)              # kpx_bcast_fn
ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)       # kpx_allgather_fn
RETRY = 1            # kpx_frame_step_sharded: a message outgrew its capacity on every rank alike


def check(rc):
    """kpx_status -> Python exception (SURVEY 8b error conventions: Open3D raises RuntimeError for
    invalid arguments; ValueError-like conditions map to the same class here)."""
    if rc != 0:
        msg = load().kpx_last_error().decode() or f"kpx error {rc}"
        raise KinectPxError(msg)


_gpu_checked = False


def device():
    global _gpu_checked
    if not _gpu_checked:
        if not torch.cuda.is_available():
            raise KinectPxError("no ROCm device visible: the kinectpx hot path runs on MI355X only (no CPU fallback)")
        torch.cuda.current_device()                 # initialises the context
        _gpu_checked = True
    return torch.device("cuda", torch._C._cuda_getDevice())


def _raw_stream():
    """raw handle of the calling thread's current stream on the current device.  (torch.cuda.current_stream() costs ~30 us per
    call -- it re-checks availability through os.environ -- and the ops ask for the stream ~20 times per frame.)"""
    device()
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


def stream_ptr():
    return C.c_void_p(_raw_stream())


def ptr(t):
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), "kinectpx needs contiguous device tensors"
    return C.c_void_p(t.data_ptr())


def hptr(a):
    """host numpy float64 array -> pointer"""
    return a.ctypes.data_as(C.c_void_p)


class Workspace:
    """Grow-only device scratch buffer owned by the caller side (one per thread of use)."""

    def __init__(self):
        self.buf = None

    def get(self, nbytes):
        nbytes = int(nbytes)
        if self.buf is None or self.buf.numel() < nbytes:
            self.buf = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device())
        return C.c_void_p(self.buf.data_ptr()), C.c_size_t(self.buf.numel())


_ws = {}
_ws_lock = threading.Lock()


def workspace(nbytes):
    """scratch of the calling thread's current stream: operators queued on one stream run in order and may share it, another
    stream gets its own.  Keyed by (device, stream) -- not by host thread: pipeline.FrameStream's worker threads pick frame slots
    arbitrarily, and a (thread, stream) key would grow to depth^2 frame workspaces of hundreds of MB.  Two host threads must not
    queue work on the SAME stream concurrently (they would share the scratch; FrameStream gives every frame in flight its own
    stream and ends each frame with stream.synchronize())."""
    stream = _raw_stream()
    # The default (null) stream is where every new host thread starts: there the key keeps the thread, so that two threads calling
    # operators without a stream of their own never interleave multi-kernel operators over one scratch buffer.
    key = (torch._C._cuda_getDevice(), stream, threading.get_ident() if not stream else 0)
    with _ws_lock:
        ws = _ws.get(key)
        if ws is None:
            ws = _ws[key] = Workspace()
            if not stream:
                # a null-stream scratch belongs to its host thread: it goes when the thread does (a pool of short-lived threads calling
                # operators on the default stream would otherwise leave hundreds of MB per thread behind)
                token = getattr(_tls, "token", None)
                if token is None:
                    token = _tls.token = _ThreadToken()
                weakref.finalize(token, _drop_key, key)
        return ws.get(nbytes)                       # (growing under the lock: a reallocation must not race with another thread's get)


class _ThreadToken:
    """lives in a thread's local storage; collected when the thread ends"""


_tls = threading.local()


def _drop_key(key):
    with _ws_lock:
        _ws.pop(key, None)


def release_workspace(stream_handle, dev=None):
    """drop the scratch of one stream (pipeline.FrameStream.close())"""
    with _ws_lock:
        _ws.pop((torch._C._cuda_getDevice() if dev is None else dev, int(stream_handle), 0), None)
