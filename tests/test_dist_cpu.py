"""CPU suite: the N>1 exchange path with world_size 2 over gloo (no GPU, no kernels)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from kinectpy_amd import parallel
    r, w, _ = parallel.init_distributed("gloo")
    assert (r, w) == (rank, world)
    sensors = parallel.shard_sensors(5, rank, world)
    cap = 64
    n = 10 + 7 * rank                                   # ragged counts
    buf = torch.zeros((cap, 6), dtype=torch.float32)
    buf[:n] = torch.arange(n * 6, dtype=torch.float32).reshape(n, 6) + 1000 * rank
    Ts = torch.stack([torch.eye(4, dtype=torch.float64) * (1 + rank + 0.1 * k) for k in range(2)])
    cloud, all_T, counts = parallel.allgather_clouds(buf, n, Ts)
    # adaptive exchange: the first message is too small for rank 1 (17 rows), the resend must deliver everything; the
    # second frame then fits at once in the capacity learnt from the first
    xc = parallel.CloudExchange(12)
    px, cx, tx, cntx = xc(buf[:n, :3], buf[:n, 3:], Ts)
    assert cntx == [10, 17] and px.shape == (27, 3) and xc.cap == 4096
    px2, _, _, _ = xc(buf[:n, :3], buf[:n, 3:], Ts)
    assert torch.equal(px, px2) and torch.equal(torch.cat([px, cx], 1), cloud)
    t = parallel.allreduce_max(1.0 + rank, "cpu")
    parallel.barrier()
    q.put((rank, sensors, cloud.numpy(), all_T.numpy(), counts, t))
    torch.distributed.destroy_process_group()


def test_world2_gloo_exchange():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == [0, 1, 2] and res[1][1] == [3, 4]           # sensors dealt contiguously
    for rank, _, cloud, all_T, counts, t in res:
        assert counts == [10, 17] and cloud.shape == (27, 6) and t == 2.0
        assert cloud[0, 0] == 0 and cloud[10, 0] == 1000               # rank order preserved, padding dropped
        assert all_T.shape == (4, 4, 4) and np.isclose(all_T[2, 0, 0], 2.0) and np.isclose(all_T[3, 1, 1], 2.1)
    assert np.array_equal(res[0][2], res[1][2])                         # every rank holds the same fused cloud


def test_shard_sensors_covers_everything():
    from kinectpy_amd import parallel
    for n in (1, 4, 8, 13):
        for w in (1, 2, 4, 8):
            got = sum((parallel.shard_sensors(n, r, w) for r in range(w)), [])
            assert got == list(range(n))
