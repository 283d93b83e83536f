"""dev-only: first contact with the GPU box (extract / misc / voxel kernels vs the oracle)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from kinectpy_amd import _lib as L
lib = C.CDLL(L.SO_PATH)
for name, (res, args) in L.SIGNATURES.items():
    if hasattr(lib, name):
        fn = getattr(lib, name); fn.restype, fn.argtypes = res, args
L._lib = lib
from kinectpy_amd import ops
from kinectpy_amd.utils import synth
from oracle import oracle as O
print("torch", torch.__version__, torch.cuda.get_device_name(0), "kpx", lib.kpx_version())
xy = synth.xy_table(); dep = synth.render_depth(xy=xy); rgb = synth.person_mask_rgb(dep)
t0=time.time(); xyz = ops.unproject_u16(dep, xy).cpu().numpy()[0]; print("unproject", time.time()-t0)
ref = O.unproject_u16(dep, xy); print("unproject exact:", np.array_equal(xyz, ref), (xyz!=0).any(1).sum())
for (cm, dg) in [(False, False), (True, False), (False, True), (True, True)]:
    (p, c, i), = ops.rgbd_compact(xyz, rgb, 1, cm, dg)
    med = O.median_z(ref)
    rp, rc, ri = O.rgbd_compact(ref, rgb, cm, dg, med + 750.0)
    print("compact", cm, dg, len(ri), np.array_equal(p.cpu().numpy(), rp), np.array_equal(c.cpu().numpy(), rc), np.array_equal(i.cpu().numpy(), ri))
# batched fused
F = 3
deps = np.stack([synth.render_depth(seed=s, xy=xy) for s in (1, 2, 3)])
rgbs = np.stack([synth.person_mask_rgb(d) for d in deps])
res = ops.depth_to_cloud(deps, xy, rgbs, F, True, True, want_idx=True)
for f in range(F):
    r = O.unproject_u16(deps[f], xy); med = O.median_z(r)
    rp, rc, ri = O.rgbd_compact(r, rgbs[f], True, True, med + 750.0)
    p, c, i = res[f]
    print("fused", f, len(ri), np.array_equal(p.cpu().numpy(), rp), np.array_equal(c.cpu().numpy(), rc), np.array_equal(i.cpu().numpy(), ri))
# transform / voxel / select
base = synth.frame_cloud(); T = synth.t_star()
print("transform exact:", np.array_equal(ops.transform(base, T).cpu().numpy(), O.transform(base, T)))
col = np.random.default_rng(0).random(base.shape).astype(np.float32)
for v in (0.02, 10.0, 35.0):
    gp, gc, _ = ops.voxel_downsample(base, v, col)
    rp, rc, _ = O.voxel_downsample(base, v, col)
    print("voxel", v, len(rp), gp.shape[0], np.array_equal(gp.cpu().numpy(), rp), np.array_equal(gc.cpu().numpy(), rc))
idx = np.random.default_rng(1).choice(len(base), 1000, replace=False).astype(np.int32)
g, = [x for x in ops.select_by_index([torch.as_tensor(base).cuda()], idx) if x is not None]
print("select", np.array_equal(g.cpu().numpy(), base[idx]))
g, = [x for x in ops.select_by_index([torch.as_tensor(base).cuda()], idx, invert=True) if x is not None]
m = np.ones(len(base), bool); m[idx] = False
print("select inv", np.array_equal(g.cpu().numpy(), base[m]))
lo, up = ops.slab_split(base, 200.0)
y = base[:, 1].astype(np.float64); cut = y.max() - 200
print("slab", np.array_equal(lo.cpu().numpy(), np.flatnonzero(y >= cut)), np.array_equal(up.cpu().numpy(), np.flatnonzero(y < cut)))
hs = ops.halfspace_select(base, [0.1, -0.9, 0.2, 300.0])
print("halfspace", np.array_equal(hs.cpu().numpy(), O.halfspace_keep_idx(0.1, -0.9, 0.2, 300.0, base)))
