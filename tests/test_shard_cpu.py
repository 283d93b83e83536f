"""CPU suite: the north-star multi-GPU partition (sensor g -> rank g, master-cloud broadcast, per-rank registration,
all-gather of the transformed clouds, filter on the FUSED cloud) rehearsed with world_size 2, 3 and 4 over gloo.

The product has no CPU path, so the ranks run kinectpy_amd.pipeline.SensorShardPipeline with tests/oracle_ops.py injected
as the operator namespace (every computation by the oracle): what is under test is the partition itself -- ownership,
collectives, ordering, the sharded fused filter -- and the bar is equality with the SINGLE-process oracle step over the
same sensors (preprocessing/data.py:35-61, 127-161), index arrays bit-exact.  The GPU twin of this test
(tests/test_parity_gpu.py::test_sensor_partition_equals_single_process_oracle) runs the same ranks on the HIP kernels."""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_sensors, mode, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      OMP_NUM_THREADS="2")
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import torch
    import oracle_ops
    from kinectpy_amd import parallel
    from kinectpy_amd.pipeline import PipelineParams, SensorShardPipeline
    from kinectpy_amd.utils import synth
    parallel.init_distributed("gloo")
    mine = parallel.shard_sensors(n_sensors, rank, world)
    xy, depth, rgb, inits, _ = synth.sensor_ring(n_sensors, 1, synth.small_xy(), sensors=mine)
    pipe = SensorShardPipeline(xy, n_sensors, inits, PipelineParams(), fused_filter=mode, cloud_capacity=4096, ops_module=oracle_ops)
    assert pipe.sensors == mine
    out = []
    for rep in range(2):                       # the second step runs with the message sizes learnt from the first
        p, c, Ts = pipe.step(torch.as_tensor(depth[0]), torch.as_tensor(rgb[0]))
        out.append((None if p is None else p.numpy(), None if c is None else c.numpy(), Ts))
    parallel.barrier()
    q.put((rank, out, dict(pipe.last)))
    torch.distributed.destroy_process_group()


def _reference(n_sensors):
    from oracle import oracle as O
    from kinectpy_amd.pipeline import PipelineParams
    from kinectpy_amd.utils import synth
    xy, depth, rgb, inits, _ = synth.sensor_ring(n_sensors, 1, synth.small_xy())
    return O.pipeline_step(xy, depth[0], rgb[0], inits, PipelineParams())


@pytest.mark.parametrize("world,n_sensors,mode", [(2, 4, "sharded"), (4, 4, "sharded"), (4, 4, "rank0"), (3, 4, "sharded"),
                                                  (8, 8, "sharded"), (4, 8, "rank0")])       # (8, 8): BASELINE configs[4]'s partition
def test_sensor_partition_equals_single_process_oracle_cpu(world, n_sensors, mode):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_sensors, mode, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref_p, ref_c, ref_T, aux = _reference(n_sensors)
    assert len(ref_p) > 200                                      # a non-trivial fused cloud
    for rank, out, last in res:
        for p, c, Ts in out:
            assert Ts.shape == (n_sensors, 4, 4)
            assert np.array_equal(Ts, np.stack(ref_T))             # every rank ends up with every sensor's transform, bit-equal
            if mode == "rank0" and rank != 0:
                assert p is None and c is None
                continue
            assert np.array_equal(p, ref_p) and np.array_equal(c, ref_c)
        assert last["n_fused"] == len(aux["fused"]) and sum(last["counts"]) == last["n_fused"]
        if not (mode == "rank0" and rank != 0):
            assert last["n_voxel"] == len(aux["voxel"])


@pytest.mark.parametrize("native", [False, True])
def test_collective_order_is_the_same_on_every_rank_whatever_the_timing(native):
    """parallel.CollectiveOrder (native=True: the library's kpx_order, which kpx_frame_step_sharded drives from C++): frames in flight on host threads with random delays in front of every collective, a main thread
    that submits / pops like bench.py -- every simulated rank issues its collectives in ONE order (the software-pipelined key
    order), including the drain at the end and a frame that skips its third collective"""
    import random
    import threading
    from concurrent.futures import ThreadPoolExecutor
    from kinectpy_amd.parallel import CollectiveOrder
    if native:
        from kinectpy_amd import _lib
        try:
            _lib.load()                 # kpx_order lives in libkinectpx.so (needs the built library and libamdhip64, not a GPU)
        except Exception as e:          # noqa: BLE001
            pytest.skip(f"libkinectpx.so not loadable here: {e}")

    def rank_run(seed, depth, n_frames, skip3):
        rnd = random.Random(seed)
        order = CollectiveOrder(depth, native=native)
        pool = ThreadPoolExecutor(max_workers=depth)

        def frame(f):
            try:
                for stage in range(3):
                    time.sleep(rnd.random() * 0.004)
                    if stage == 2 and f in skip3:
                        order.skip(f, 2)
                        continue
                    with order.turn(f, stage):
                        time.sleep(rnd.random() * 0.001)
            finally:
                order.finish(f)

        pending = []
        for k in range(n_frames):
            if len(pending) >= depth:
                f0, fut = pending.pop(0)
                order.block(f0)
                fut.result()
                order.block(None)
            time.sleep(rnd.random() * 0.003)
            f = order.submit()
            pending.append((f, pool.submit(frame, f)))
        while pending:
            f0, fut = pending.pop(0)
            order.block(f0)
            fut.result()
            order.block(None)
        pool.shutdown()
        return order.log

    import time
    for depth in (1, 2, 3):
        logs = []
        ths = [threading.Thread(target=lambda s_=s_: logs.append(rank_run(s_, depth, 9, {4}))) for s_ in range(4)]
        [t.start() for t in ths]
        [t.join(timeout=60) for t in ths]
        assert len(logs) == 4 and all(lg == logs[0] for lg in logs), (depth, logs)
        if native:                       # and the library's order is the Python one's
            from kinectpy_amd.parallel import CollectiveOrder as CO
            native = False
            assert logs[0] == rank_run(99, depth, 9, {4})
            native = True
        o = CollectiveOrder(depth)
        want = sorted(o.key(f, s_) for f in range(9) for s_ in range(3) if not (s_ == 2 and f == 4))
        assert sorted(logs[0]) == want
        if depth == 2:                       # steady state: the next frame's broadcast goes before this frame's exchange
            assert logs[0][:6] == [o.key(0, 0), o.key(1, 0), o.key(0, 1), o.key(0, 2), o.key(2, 0), o.key(1, 1)]
