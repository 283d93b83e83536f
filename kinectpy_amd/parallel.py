"""Multi-GPU layer: one process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm, "gloo" for the
CPU rehearsal in tests).  The path shards by sensor (preprocessing/data.py:96-122 loops the devices with no
cross-talk); the only exchange is the fuse (data.py:44-58): the rigid transforms (128 B each) and the
filtered clouds.  Messages are small and latency-bound on xGMI, so each frame uses exactly ONE collective: an
all-gather of the padded cloud buffer with the fixed-size header (count + transforms, float64 bit patterns)
appended as extra rows (direct peer writes; no ring reduction of payload).
"""
import os
from typing import List, Tuple

import torch
import torch.distributed as dist


# Test hook: run the collectives on a ONE-rank group too (a single-GPU box can then push the north-star partition through RCCL
# itself -- broadcast, all_gather_into_tensor on device tensors, one communicator per frame slot -- instead of short-cutting them)
FORCE_COLLECTIVES = False


def collectives_on(group=None) -> bool:
    return dist.is_initialized() and (world_size(group) > 1 or FORCE_COLLECTIVES)


class CollectiveOrder:
    """One global order for the collectives of the frames in flight on a rank (pipeline.FrameStream with SensorShardPipeline).

    Frames in flight run on host threads; which thread reaches its next collective first depends on timing, so two ranks would
    enqueue the collectives of frames f and f + 1 in different orders.  A collective kernel spins on the device until its peers
    arrive, and two of them -- even on different communicators -- are only safe when the device can run both at once, which
    nothing guarantees (streams share hardware queues).  So every rank issues them in ONE fixed, software-pipelined order,
    independent of timing: stage s (0 master broadcast, 1 cloud exchange, 2 slab all-gather) of frame f has the key
        3 f                      for s = 0,
        3 (f + depth - 1) + s    for s = 1, 2,
    i.e. frame f + 1's broadcast goes before frame f's exchange (it is ready earlier: the registration lies in between), and a
    thread may issue a collective only when every smaller key of the frames submitted so far is done.  If the next frame has not
    been submitted yet and its broadcast key is smaller, the thread waits until the main thread either submits it or waits in
    FrameStream.pop() for THIS frame (then the next frame cannot come before this collective on any rank: the main thread's
    submit / pop sequence is the same program everywhere, and it cannot go on before this frame is done).  Stages a frame does
    not use are passed with skip()."""

    STAGES = 3

    def __new__(cls, depth: int = 1, native: bool = False):
        return super().__new__(NativeCollectiveOrder if (native and cls is CollectiveOrder) else cls)

    def __init__(self, depth: int, native: bool = False):
        import threading
        self.depth = max(1, int(depth))
        self.cv = threading.Condition()
        self.pending = set()           # keys of submitted frames not yet done
        self.submitted = 0
        self.waiting_for = None        # frame the main thread waits for in pop(), else None
        self.log = []                  # keys in the order they were issued (tests)

    def key(self, frame: int, stage: int) -> int:
        return 3 * frame if stage == 0 else 3 * (frame + self.depth - 1) + stage

    def submit(self) -> int:
        """main thread: a new frame enters; -> its frame number"""
        with self.cv:
            f = self.submitted
            self.submitted += 1
            for s_ in range(self.STAGES):
                self.pending.add(self.key(f, s_))
            self.cv.notify_all()
            return f

    def block(self, frame):
        """main thread: waits for `frame` (None: the wait is over)"""
        with self.cv:
            self.waiting_for = frame
            self.cv.notify_all()

    def _may_go(self, k: int, frame: int) -> bool:
        return k == min(self.pending) and (k < 3 * self.submitted or self.waiting_for == frame)

    def turn(self, frame: int, stage: int):
        order, k = self, self.key(frame, stage)

        class _Turn:
            def __enter__(self_inner):
                with order.cv:
                    order.cv.wait_for(lambda: order._may_go(k, frame))
                    order.log.append(k)

            def __exit__(self_inner, *exc):
                with order.cv:
                    order.pending.discard(k)
                    order.cv.notify_all()
                return False
        return _Turn()

    def skip(self, frame: int, stage: int):
        with self.cv:
            self.pending.discard(self.key(frame, stage))
            self.cv.notify_all()

    def finish(self, frame: int):
        """the frame is over (also after an exception): whatever stages it did not reach are passed"""
        with self.cv:
            for s_ in range(self.STAGES):
                self.pending.discard(self.key(frame, s_))
            self.cv.notify_all()


class NativeCollectiveOrder(CollectiveOrder):
    """The same order kept inside the library (kpx_order, kpx_comm.hip): kpx_frame_step_sharded takes its turns in C++ without
    coming back to the interpreter.  Same interface as CollectiveOrder (the tests run against both)."""

    def __init__(self, depth: int, native: bool = True):
        import ctypes as C
        from . import _lib
        self.depth = max(1, int(depth))
        self._L = _lib.load()
        h = C.c_void_p()
        _lib.check(self._L.kpx_order_create(self.depth, C.byref(h)))
        self.handle = h

    def submit(self) -> int:
        import ctypes as C
        f = C.c_int64()
        self._L.kpx_order_submit(self.handle, C.byref(f))
        return int(f.value)

    def block(self, frame):
        self._L.kpx_order_block(self.handle, -1 if frame is None else int(frame))

    def turn(self, frame: int, stage: int):
        order = self

        class _Turn:
            def __enter__(self_inner):
                order._L.kpx_order_turn_begin(order.handle, int(frame), int(stage))

            def __exit__(self_inner, *exc):
                order._L.kpx_order_turn_end(order.handle, int(frame), int(stage))
                return False
        return _Turn()

    def skip(self, frame: int, stage: int):
        self._L.kpx_order_skip(self.handle, int(frame), int(stage))

    def finish(self, frame: int):
        self._L.kpx_order_finish(self.handle, int(frame))

    @property
    def log(self):
        import ctypes as C
        import numpy as np
        n = C.c_int64()
        self._L.kpx_order_log(self.handle, None, 0, C.byref(n))
        out = np.zeros(max(1, n.value), dtype=np.int64)
        self._L.kpx_order_log(self.handle, out.ctypes.data_as(C.c_void_p), out.size, C.byref(n))
        return out[: n.value].tolist()

    def __del__(self):
        try:
            self._L.kpx_order_destroy(self.handle)
        except Exception:
            pass


# ---- kpx_comm: the native communicator of the sharded frame loop (kpx_frame_step_sharded) -------------------------------------
class NativeComm:
    """One communicator over all ranks for the native multi-rank frame loop (one per frame slot).  Transports:
      NativeComm.rccl(group)    RCCL called from C++ on the frame's stream; rank 0 draws the id, torch.distributed carries it
                                (collective: every rank calls it in the same order);
      NativeComm.staged(group)  the bytes travel through torch.distributed on host buffers (gloo): rehearsal of the same loop with
                                several ranks sharing one GPU;
      NativeComm.local(hub, r)  in-process ranks (threads of one process, tests): device-to-device copies around a barrier."""

    def __init__(self, handle, rank, world, keep=()):
        self.handle, self.rank, self.world, self._keep = handle, int(rank), int(world), keep
        self.error = None

    @staticmethod
    def _rccl_path():
        import glob
        cands = glob.glob(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so*")) + glob.glob("/opt/rocm/lib/librccl.so*")
        return cands[0] if cands else None

    @classmethod
    def rccl(cls, group=None):
        """Collective: every rank calls it, in the same order.  Every step that can fail on ONE rank (loading librccl, drawing the
        id, ncclCommInitRank) is followed by an agreement over the default group, so that either every rank returns a communicator
        or every rank raises -- no rank is left inside broadcast_object_list or ncclCommInitRank waiting for a peer that has
        already given up (bench.py falls back to the staged transport on the exception)."""
        import ctypes as C
        from . import _lib
        L = _lib.load()
        dev = _lib.device()
        rank = dist.get_rank() if dist.is_initialized() else 0
        world = world_size(group)
        together = dist.is_initialized() and world > 1

        def agree(err, what):
            """raises on EVERY rank when any rank failed at this step"""
            # (over `group` itself: the id is broadcast over it and its size is the communicator's -- a strict subgroup must not wait for
            # ranks outside it)
            bad = allreduce_max(0.0 if err is None else 1.0, dev, group) if together else (0.0 if err is None else 1.0)
            if bad > 0.0:
                raise RuntimeError(f"NativeComm.rccl: {what} failed on {'this rank: ' + repr(err) if err is not None else 'another rank'}")

        ident, err = C.create_string_buffer(128), None
        try:
            path = cls._rccl_path()
            _lib.check(L.kpx_rccl_load(path.encode() if path else None))
            if rank == 0:
                _lib.check(L.kpx_rccl_unique_id(ident))
        except Exception as e:          # noqa: BLE001 -- the peers have to learn of it before anyone blocks
            err = e
        agree(err, "loading RCCL / drawing the communicator id")
        if together:
            box = [bytes(ident.raw)]
            dist.broadcast_object_list(box, src=0, group=group)
            ident = C.create_string_buffer(box[0], 128)
        h, err = C.c_void_p(), None
        try:
            _lib.check(L.kpx_comm_create_rccl(ident, rank, world, C.byref(h)))
        except Exception as e:          # noqa: BLE001
            err = e
        try:
            agree(err, "ncclCommInitRank")
        except RuntimeError:
            if err is None:             # this rank's communicator exists, a peer's does not: nobody may use it
                try:
                    L.kpx_comm_destroy(h)
                except Exception:       # noqa: BLE001
                    pass
            raise
        return cls(h, rank, world)

    @classmethod
    def _callbacks(cls, rank, world, bcast, allgather):
        import ctypes as C
        from . import _lib
        L = _lib.load()
        self = cls(None, rank, world)

        def guard(fn):
            def run(*a):
                try:
                    fn(*a)
                    return 0
                except BaseException as e:      # an exception must not cross the C frames: keep it for the caller
                    self.error = e
                    return 1
            return run
        b, g = _lib.BCAST_FN(guard(bcast)), _lib.ALLGATHER_FN(guard(allgather))
        h = C.c_void_p()
        _lib.check(L.kpx_comm_create_callbacks(rank, world, C.cast(b, C.c_void_p), C.cast(g, C.c_void_p), None, C.byref(h)))
        self.handle, self._keep = h, (b, g)
        return self

    @classmethod
    def staged(cls, group=None):
        """host-staged transport over torch.distributed (gloo): device -> host buffer -> collective -> device"""
        import ctypes as C
        import numpy as np
        from . import _lib
        L = _lib.load()
        rank = dist.get_rank() if dist.is_initialized() else 0
        world = world_size(group)

        def copy(dst, src, n, stream):
            _lib.check(L.kpx_copy_bytes(C.c_void_p(dst), C.c_void_p(src), n, C.c_void_p(stream), 1))

        def bcast(user, d_buf, n, root, stream):
            h = np.empty(n, dtype=np.uint8)
            if rank == root:
                copy(h.ctypes.data, d_buf, n, stream)
            if world > 1:
                dist.broadcast(torch.from_numpy(h), root, group=group)
            if rank != root:
                copy(d_buf, h.ctypes.data, n, stream)

        def allgather(user, d_send, d_recv, n, stream):
            h, out = np.empty(n, dtype=np.uint8), np.empty((world, n), dtype=np.uint8)
            copy(h.ctypes.data, d_send, n, stream)
            if world > 1:
                parts = [torch.from_numpy(out[r]) for r in range(world)]
                dist.all_gather(parts, torch.from_numpy(h), group=group)
            else:
                out[0] = h
            copy(d_recv, out.ctypes.data, world * n, stream)
        return cls._callbacks(rank, world, bcast, allgather)

    class LocalHub:
        """rendezvous of `world` in-process ranks (one thread each)"""

        def __init__(self, world):
            import threading
            self.world = int(world)
            self.barrier = threading.Barrier(self.world)
            self.slot = [None] * self.world

    @classmethod
    def local(cls, hub, rank, record=None):
        """record: a list that receives, per collective this rank takes part in, a device copy (uint8 tensor) of what the rank SENT
        (None for a broadcast it only received) -- the recordings kpx_comm_create_replay plays back (bench.py --emulate-world)"""
        import ctypes as C
        from . import _lib
        L = _lib.load()

        def copy(dst, src, n, stream):
            _lib.check(L.kpx_copy_bytes(C.c_void_p(dst), C.c_void_p(src), n, C.c_void_p(stream), 1))

        def keep(src, n, stream):
            t = torch.empty(int(n), dtype=torch.uint8, device=_lib.device())
            copy(t.data_ptr(), src, n, stream)
            record.append(t)

        def bcast(user, d_buf, n, root, stream):
            copy(None, None, 0, stream)                       # everything queued before the collective has happened
            if record is not None:
                keep(d_buf, n, stream) if rank == root else record.append(None)
            hub.slot[rank] = d_buf
            hub.barrier.wait()
            if rank != root:
                copy(d_buf, hub.slot[root], n, stream)
            hub.barrier.wait()

        def allgather(user, d_send, d_recv, n, stream):
            copy(None, None, 0, stream)
            if record is not None:
                keep(d_send, n, stream)
            hub.slot[rank] = d_send
            hub.barrier.wait()
            for r in range(hub.world):
                copy(d_recv + r * n, hub.slot[r], n, stream)
            hub.barrier.wait()
        return cls._callbacks(rank, hub.world, bcast, allgather)

    @classmethod
    def replay(cls, rank, world, recordings, first_frame=0, stride=1, per_frame=3):
        """ONE rank of a `world`-rank job, the peers' messages played back (kpx_comm_create_replay): recordings[r] = the list
        NativeComm.local(..., record=) filled on rank r, three entries per frame (master broadcast, cloud exchange, slab all-gather).
        Call n of the communicator is collective n % 3 of frame (first_frame + stride * (n // 3)) % frames."""
        import ctypes as C
        from . import _lib
        L = _lib.load()
        frames = len(recordings[0]) // per_frame           # (per_frame = 2: frames without the slab all-gather -- fused_filter rank0 / round robin)
        n = frames * 3 * world
        ptrs, sizes, keep = (C.c_void_p * n)(), (C.c_size_t * n)(), []
        for f in range(frames):
            for c in range(3):
                for r in range(world):
                    t = recordings[r][per_frame * f + c] if c < per_frame else None
                    i = (f * 3 + c) * world + r
                    ptrs[i] = None if t is None else t.data_ptr()
                    sizes[i] = 0 if t is None else t.numel()
                    keep.append(t)
        h = C.c_void_p()
        _lib.check(L.kpx_comm_create_replay(int(rank), int(world), frames, int(first_frame), int(stride), int(per_frame), C.cast(ptrs, C.c_void_p), C.cast(sizes, C.c_void_p), C.byref(h)))
        return cls(h, rank, world, keep=(keep, ptrs, sizes))

    def close(self):
        from . import _lib
        if self.handle is not None:
            _lib.load().kpx_comm_destroy(self.handle)
            self.handle = None


def init_distributed(backend: str = None) -> Tuple[int, int, int]:
    """-> (rank, world, local_rank).  Reads RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* from the env."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    ndev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = os.environ.get("KPX_DIST_BACKEND") or ("nccl" if ndev else "gloo")
        if ndev:
            torch.cuda.set_device(local % ndev)      # gloo rehearsal: several ranks may share one GPU
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    elif ndev:
        torch.cuda.set_device(local % ndev)
    return rank, world, (local % ndev if ndev else local)


def shard_sensors(n_sensors: int, rank: int, world: int) -> List[int]:
    """Contiguous block of sensors owned by `rank` (sensors are dealt as evenly as possible; with fewer
    GPUs than sensors a rank simply owns several, same code path -- SURVEY.md 8e)."""
    base, rem = divmod(n_sensors, world)
    start = rank * base + min(rank, rem)
    return list(range(start, start + base + (1 if rank < rem else 0)))


def world_size(group=None) -> int:
    return dist.get_world_size(group) if dist.is_initialized() else 1


def new_group():
    """another communicator over all ranks (one per frame slot of pipeline.FrameStream: collectives of different frames in
    flight must not share a communicator, their issue order differs between ranks); None on a single rank"""
    if not collectives_on():
        return None
    return dist.new_group(ranks=list(range(dist.get_world_size())), backend=dist.get_backend())


def _all_gather(t: torch.Tensor):
    """all_gather that stages through the host when the backend is gloo and the tensor lives on a GPU"""
    if dist.get_backend() == "gloo" and t.is_cuda:
        out = [torch.empty(t.shape, dtype=t.dtype) for _ in range(world_size())]
        dist.all_gather(out, t.cpu())
        return [o.to(t.device) for o in out]
    out = [torch.empty_like(t) for _ in range(world_size())]
    dist.all_gather(out, t)
    return out


def allgather_header(header: torch.Tensor) -> torch.Tensor:
    """header: 1-D float64 tensor of equal length on every rank -> (world, len)."""
    if world_size() == 1:
        return header[None]
    return torch.stack(_all_gather(header))


def _all_gather_flat(t: torch.Tensor, group=None) -> torch.Tensor:
    """(...)-> (world, ...) with one all_gather_into_tensor (host-staged under gloo when the tensor is on a GPU)"""
    w = world_size(group)
    if dist.get_backend(group) == "gloo" and t.is_cuda:
        out = torch.empty((w,) + tuple(t.shape), dtype=t.dtype)
        try:
            dist.all_gather_into_tensor(out, t.cpu().contiguous(), group=group)
        except (RuntimeError, NotImplementedError):
            parts = [torch.empty(t.shape, dtype=t.dtype) for _ in range(w)]
            dist.all_gather(parts, t.cpu().contiguous(), group=group)
            out = torch.stack(parts)
        return out.to(t.device)
    out = torch.empty((w,) + tuple(t.shape), dtype=t.dtype, device=t.device)
    try:
        dist.all_gather_into_tensor(out, t.contiguous(), group=group)
    except (RuntimeError, NotImplementedError):
        parts = [torch.empty_like(t) for _ in range(w)]
        dist.all_gather(parts, t.contiguous(), group=group)
        out = torch.stack(parts)
    return out


def _broadcast(t: torch.Tensor, src: int, group=None) -> torch.Tensor:
    """in-place broadcast from global rank `src` (host-staged under gloo when the tensor is on a GPU)"""
    if dist.get_backend(group) == "gloo" and t.is_cuda:
        h = t.cpu()
        dist.broadcast(h, src, group=group)
        if dist.get_rank() != src:
            t.copy_(h)
        return t
    dist.broadcast(t, src, group=group)
    return t


def allgather_clouds(padded: torch.Tensor, count: int, transforms: torch.Tensor, always_collective: bool = False, group=None):
    """padded: (cap, C) float32 buffer whose first min(count, cap) rows are valid (same cap on every rank; `count` may
    exceed it: the header carries the true count so that the caller can resend with room);
    transforms: (k, 4, 4) float64 of this rank's sensors (same k on every rank).
    Returns (cloud (sum counts, C), all transforms (world*k, 4, 4), counts list).
    One collective: the header (count, transforms) travels as float64 bit patterns in extra rows of the buffer.
    `always_collective` runs the collective on a one-rank group too (the single-GPU RCCL test)."""
    k = transforms.shape[0]
    hdr = torch.cat([torch.tensor([float(count)], dtype=torch.float64), transforms.reshape(-1).to("cpu", torch.float64)])
    W = world_size(group)
    if W == 1 and not (always_collective and dist.is_initialized()):
        return padded[:min(count, padded.shape[0])], hdr[1:].reshape(-1, 4, 4).to(padded.device), [int(count)]
    cap, C = padded.shape
    words = hdr.numel() * 2                                     # float32 words carrying the float64 header
    hrows = (words + C - 1) // C
    msg = torch.empty((cap + hrows, C), dtype=torch.float32, device=padded.device)
    msg[:cap] = padded
    tail = torch.zeros(hrows * C, dtype=torch.float32)
    tail[:words] = hdr.view(torch.float32)
    msg[cap:] = tail.reshape(hrows, C).to(padded.device, non_blocking=True)
    allm = _all_gather_flat(msg, group)                         # (world, cap + hrows, C)
    hdrs = allm[:, cap:].reshape(W, -1)[:, :words].cpu().reshape(-1).clone().view(torch.float64).reshape(W, -1)   # one read-back
    counts = [int(c) for c in hdrs[:, 0].tolist()]
    all_T = hdrs[:, 1:].reshape(-1, 4, 4).to(padded.device)
    cloud = torch.cat([allm[r, :min(c, cap)] for r, c in enumerate(counts)], 0)
    assert all_T.shape[0] == k * W
    return cloud, all_T, counts


class CloudExchange:
    """Per-frame fuse exchange with an adaptive message size: the padded message holds `cap` rows, sized from the
    counts every rank saw in the previous frame (+25 %, rounded to 4096: the same value on all ranks); a frame that
    outgrows it is detected from the gathered header (which carries the true counts) and sent again with room."""

    def __init__(self, initial_rows: int, group=None):
        if initial_rows <= 0:
            raise ValueError("cloud_capacity must be set for multi-GPU exchange")
        self.cap = int(initial_rows)
        self.group = group

    def __call__(self, pts: torch.Tensor, col: torch.Tensor, transforms: torch.Tensor):
        n = int(pts.shape[0])
        while True:
            cap = self.cap
            buf = torch.empty((cap, 6), dtype=torch.float32, device=pts.device)
            k = min(n, cap)
            buf[:k, :3] = pts[:k]
            buf[:k, 3:] = col[:k]
            cloud, all_T, counts = allgather_clouds(buf, n, transforms, group=self.group)
            need = max(counts)
            self.cap = max(4096, -(-int(need * 1.25) // 4096) * 4096)      # every rank sees the same counts
            if need <= cap:
                return cloud[:, :3], cloud[:, 3:], all_T, counts


class SensorExchange:
    """Per-frame fuse exchange of the north-star partition (preprocessing/data.py:44-58: the sub devices' clouds are moved by
    their transforms and stacked behind the master's): every rank contributes the masked clouds of the sensors it owns,
    UNMOVED (float32 values of int16 sensor data: exact), together with their point counts and 4x4 transforms -- ONE
    all-gather per frame.  The receiver applies the transforms inside the fused voxel grid in fp64
    (kpx_fuse_voxel_downsample), so no float32 rounding of a moved point ever crosses the wire or decides a voxel.
    Message of a rank: `cap` rows of xyz, `cap` rows of rgb (planar, so that every sensor's cloud is a contiguous (n,3) view
    of the gathered buffer), then the header rows: k_max counts and k_max transforms as float64 bit patterns.  `cap` follows
    the largest total of the previous frame (+25 %, rounded to 4096, the same on all ranks); a frame that outgrows it is
    seen by every rank in the gathered header and sent again with room."""

    def __init__(self, initial_rows: int, k_max: int, group=None):
        self.cap = max(4096, int(initial_rows))
        self.k_max = int(k_max)
        self.group = group

    def __call__(self, clouds, transforms):
        """clouds: [(pts (n_i,3) f32, col (n_i,3) f32)] of this rank's sensors (at most k_max); transforms: (len(clouds),4,4) f64.
        -> segments [[(pts, col)] per rank], T (world, k_max, 4, 4) numpy, counts (world, k_max) numpy int64"""
        import numpy as np
        W, K = world_size(self.group), self.k_max
        ns = [int(p.shape[0]) for p, _ in clouds]
        hdr = np.zeros(K * 17, dtype=np.float64)
        hdr[:len(ns)] = ns
        Tl = np.tile(np.eye(4), (K, 1, 1))
        Tl[:len(ns)] = np.asarray(transforms, dtype=np.float64).reshape(-1, 4, 4)
        hdr[K:] = Tl.reshape(-1)
        if not collectives_on(self.group):
            return [list(clouds)], Tl[None], np.array([ns + [0] * (K - len(ns))], dtype=np.int64)
        dev = clouds[0][0].device
        words = hdr.size * 2
        hrows = -(-words // 3)
        tail = torch.zeros(hrows * 3, dtype=torch.float32)
        tail[:words] = torch.from_numpy(hdr).view(torch.float32)
        while True:
            cap = self.cap
            msg = torch.empty((2 * cap + hrows, 3), dtype=torch.float32, device=dev)
            off = 0
            for (p, c), n in zip(clouds, ns):
                k = max(0, min(n, cap - off))
                msg[off:off + k] = p[:k]
                msg[cap + off:cap + off + k] = c[:k]
                off += k
            msg[2 * cap:] = tail.reshape(hrows, 3).to(dev, non_blocking=True)
            allm = _all_gather_flat(msg, self.group)                                  # (world, 2 cap + hrows, 3)
            h = np.ascontiguousarray(allm[:, 2 * cap:].reshape(W, -1)[:, :words].cpu().numpy()).copy().view(np.float64)   # one read-back
            counts = h[:, :K].astype(np.int64)
            need = int(counts.sum(1).max())
            self.cap = max(4096, -(-int(need * 1.25) // 4096) * 4096)
            if need <= cap:
                segs = []
                for r in range(W):
                    o, row = 0, []
                    for j in range(K):
                        n = int(counts[r, j])
                        row.append((allm[r, o:o + n], allm[r, cap + o:cap + o + n]))          # k_max entries per rank (empty ones included)
                        o += n
                    segs.append(row)
                return segs, h[:, K:].reshape(W, K, 4, 4).copy(), counts


class MasterBroadcast:
    """Calibration broadcast of the north-star partition (preprocessing/data.py:140-157 registers every sub device
    onto the master's cloud): rank 0 sends the master's down-sampled cloud and its normals, ONE collective per calibration.
    The message has `cap` rows of 6 floats (xyz | normal) plus one header row carrying the true count as float64 bits;
    `cap` follows the last count seen (+25 %, rounded to 4096, the same on all ranks); a cloud that outgrows it is detected
    from the header by every rank and sent again with room."""

    def __init__(self, initial_rows: int, group=None, src: int = 0):
        self.cap = max(4096, int(initial_rows))
        self.group, self.src = group, src

    def __call__(self, pts, nrm, device=None):
        """rank src: pts (n,3), nrm (n,3) | None; other ranks: None, None (+ device) -> (pts, nrm | None) on every rank"""
        if not collectives_on(self.group):
            return pts, nrm
        me = dist.get_rank()
        if me == self.src:
            device = pts.device
        while True:
            cap = self.cap
            msg = torch.empty((cap + 1, 6), dtype=torch.float32, device=device)
            if me == self.src:
                n = int(pts.shape[0])
                k = min(n, cap)
                msg[:k, :3] = pts[:k]
                if nrm is not None:
                    msg[:k, 3:] = nrm[:k]
                hdr = torch.tensor([float(n), 1.0 if nrm is not None else 0.0], dtype=torch.float64).view(torch.float32)   # 4 words
                msg[cap, :4] = hdr.to(device)
            _broadcast(msg, self.src, self.group)
            h = msg[cap, :4].cpu().view(torch.float64)                       # one read-back: the receivers need n on the host anyway
            n, has_n = int(h[0]), bool(h[1])
            self.cap = max(4096, -(-int(n * 1.25) // 4096) * 4096)
            if n <= cap:
                if me == self.src:
                    return pts, nrm
                return msg[:n, :3].contiguous(), (msg[:n, 3:].contiguous() if has_n else None)


def allgather_slabs(part: torch.Tensor, rows: int, group=None) -> torch.Tensor:
    """part: this rank's 1-D slab, at most `rows` long (the same `rows` on every rank) -> (world, rows) after one all-gather
    (the tail of a short slab is padding).  Used by the sharded fused filter for the per-slab mean kNN distances."""
    buf = torch.zeros(rows, dtype=part.dtype, device=part.device)
    buf[: part.numel()] = part
    if not collectives_on(group):
        return buf[None]
    return _all_gather_flat(buf, group)


def warm(group, device):
    """create the communicator of `group` now, from the calling thread (a tiny all-reduce): the frame slots use their
    groups from worker threads, and every rank has to build its communicators in the same order"""
    if collectives_on(group):
        t = torch.zeros(1, dtype=torch.float32, device="cpu" if dist.get_backend(group) == "gloo" else device)
        dist.all_reduce(t, group=group)
        if t.is_cuda:
            torch.cuda.synchronize()


def barrier():
    if dist.is_initialized():
        dist.barrier()


def allreduce_max(value: float, device, group=None) -> float:
    if not dist.is_initialized():
        return value
    t = torch.tensor([value], dtype=torch.float64, device="cpu" if dist.get_backend(group) == "gloo" else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())
