"""Only the selection rows of tools/bench_kernels.py."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kinectpy_amd import ops  # noqa: E402
from tools.bench_kernels import report, timed  # noqa: E402

n_big = 64_000_000
big = torch.rand((n_big, 3), device="cuda") * 2000 - 1000
ms, hs = timed(lambda: ops.halfspace_select(big, [0.1, -0.9, 0.2, 300.0]))
report("halfspace_select (a19)", ms, n_big * 12 + int(hs.shape[0]) * 4, points=n_big, kept=int(hs.shape[0]))
ms, (lo_, up_) = timed(lambda: ops.slab_split(big, 200.0))
report("slab_split", ms, n_big * 12 + n_big * 4, points=n_big, lower=int(lo_.shape[0]))
ms, bb = timed(lambda: ops.bounds(big))
report("bounds", ms, n_big * 12, points=n_big)
ms, (lo2, up2) = timed(lambda: ops.slab_split(big, 200.0, bounds=bb))
report("slab_split, bounds known", ms, n_big * 12 + n_big * 4, points=n_big, lower=int(lo2.shape[0]), same=bool(torch.equal(lo2, lo_) and torch.equal(up2, up_)))
idx = torch.randperm(n_big, device="cuda")[: n_big // 2].to(torch.int32).sort().values
ms, _ = timed(lambda: ops.select_by_index([big], idx, trusted=True))
report("select_by_index", ms, n_big // 2 * 28)
ms, _ = timed(lambda: ops.select_by_index([big], idx, trusted=True, want_bounds=True))
report("select_by_index + bounds", ms, n_big // 2 * 28)
del idx
small = big[:260_000].contiguous()
ms, hs = timed(lambda: ops.halfspace_select(small, [0.1, -0.9, 0.2, 300.0]), reps=20)
report("halfspace_select 260k", ms, 260_000 * 12 + int(hs.shape[0]) * 4)
