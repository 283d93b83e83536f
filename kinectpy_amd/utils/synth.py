"""Synthetic Kinect scenes for tests and bench.py.

The reference ships no data and no intrinsics; these generators are ours.  Everything is seeded
NumPy on the host: a box room (floor y=+900 mm, Kinect +Y is down; walls at |x|,|z| = 3000), and a
"person" made of 6 ellipsoids.  Cameras sit on a circle around the person and look at it; depth is
ray-cast per pixel so every camera sees the same world (needed for registration tests).
"""
import numpy as np

H, W = 576, 640
FX = FY = 504.0
CX, CY = 320.0, 288.0

# (centre xyz, radii xyz) in the world frame, person standing on the floor (y=900), mm
_PERSON = [
    ((0.0, -620.0, 0.0), (95.0, 120.0, 100.0)),      # head
    ((0.0, -150.0, 0.0), (210.0, 340.0, 130.0)),     # torso
    ((-290.0, -180.0, 30.0), (60.0, 300.0, 60.0)),   # left arm
    ((300.0, -120.0, -60.0), (60.0, 280.0, 70.0)),   # right arm (asymmetric)
    ((-110.0, 520.0, 10.0), (80.0, 380.0, 85.0)),    # left leg
    ((120.0, 500.0, -30.0), (85.0, 400.0, 80.0)),    # right leg
]
ROOM = 3000.0
FLOOR_Y = 900.0


def xy_table(h=H, w=W, fx=FX, fy=FY, cx=CX, cy=CY, hexagon=True):
    """float32 (h*w, 2) table of (u-cx)/fx, (v-cy)/fy; NaN in the corners mimics the NFOV hexagon."""
    u = np.arange(w, dtype=np.float32)
    v = np.arange(h, dtype=np.float32)
    xt = ((u - np.float32(cx)) / np.float32(fx))[None, :].repeat(h, 0)
    yt = ((v - np.float32(cy)) / np.float32(fy))[:, None].repeat(w, 1)
    xy = np.stack([xt, yt], -1).astype(np.float32)
    if hexagon:
        uu = np.abs(u[None, :] - cx) / (w / 2)
        vv = np.abs(v[:, None] - cy) / (h / 2)
        xy[(uu + 0.55 * vv) > 1.22] = np.nan
    return xy.reshape(-1, 2)


def camera_pose(i, n_cams, radius=2500.0):
    """4x4 camera->world of camera i of n on a circle around the person, looking at it."""
    th = 2.0 * np.pi * i / max(n_cams, 1)
    c, s = np.cos(th), np.sin(th)
    R = np.array([[c, 0.0, s], [0.0, 1.0, 0.0], [-s, 0.0, c]])     # Ry(th)
    pos = R @ np.array([0.0, 0.0, -radius])
    E = np.eye(4)
    E[:3, :3] = R
    E[:3, 3] = pos
    return E


def clutter(n=14, seed=11):
    """Extra ellipsoids ("furniture") scattered asymmetrically over the floor and walls: gives the scene the
    geometric distinctiveness feature-based global registration needs.  [(centre, radii)]"""
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        r = rng.uniform(120.0, 420.0, size=3)
        ang, rad = rng.uniform(0, 2 * np.pi), rng.uniform(900.0, 2300.0)
        c = np.array([rad * np.cos(ang), FLOOR_Y - r[1] * rng.uniform(0.3, 1.0) - rng.uniform(0, 900.0) * (rng.random() < 0.3), rad * np.sin(ang)])
        out.append((tuple(c), tuple(r)))
    return out


def render_depth(E=None, person_shift=(0.0, 0.0, 0.0), seed=20250202, noise=2.0, drop=0.10, xy=None,
                 max_depth=6000.0, return_person=False, extra=()):
    """Ray-casts the scene from camera pose E (camera->world).  Returns uint16 (H*W,) depth in mm
    (camera-frame z), 0 = invalid; with return_person also the boolean "ray hit the person" mask."""
    rng = np.random.default_rng(seed)
    if E is None:
        E = camera_pose(0, 1, 2000.0)
    if xy is None:
        xy = xy_table()
    valid = ~np.isnan(xy[:, 0])
    d_cam = np.stack([np.nan_to_num(xy[:, 0]), np.nan_to_num(xy[:, 1]), np.ones(len(xy), np.float32)], -1).astype(np.float64)
    R, o = E[:3, :3], E[:3, 3]
    d = d_cam @ R.T                                    # world direction, parameter t == camera z
    t_best = np.full(len(xy), np.inf)

    def plane(nv, off):                                 # nv . p = off
        den = d @ nv
        with np.errstate(divide="ignore", invalid="ignore"):
            t = (off - o @ nv) / den
        t[~(t > 1e-6)] = np.inf
        return t

    for nv, off in [((0, 1, 0), FLOOR_Y), ((1, 0, 0), ROOM), ((1, 0, 0), -ROOM), ((0, 0, 1), ROOM), ((0, 0, 1), -ROOM)]:
        t = plane(np.array(nv, dtype=np.float64), off)
        p = o + d * np.where(np.isfinite(t), t, 0.0)[:, None]
        inside = (np.abs(p[:, 0]) <= ROOM + 1) & (np.abs(p[:, 2]) <= ROOM + 1) & (p[:, 1] <= FLOOR_Y + 1) & (p[:, 1] >= -2500)
        t[~inside] = np.inf
        t_best = np.minimum(t_best, t)
    for c, r in extra:                                   # static clutter counts as room
        c, r = np.array(c), np.array(r)
        oc, dd = (o - c) / r, d / r
        A, B, Cq = (dd * dd).sum(1), 2 * (dd @ oc), oc @ oc - 1.0
        disc = B * B - 4 * A * Cq
        t = np.full(len(xy), np.inf)
        ok = disc >= 0
        t[ok] = (-B[ok] - np.sqrt(disc[ok])) / (2 * A[ok])
        t[~(t > 1e-6)] = np.inf
        t_best = np.minimum(t_best, t)
    t_room = t_best.copy()
    for c, r in _PERSON:
        c = np.array(c) + np.array(person_shift)
        r = np.array(r)
        oc = (o - c) / r
        dd = d / r
        A = (dd * dd).sum(1)
        B = 2 * (dd @ oc)
        Cq = oc @ oc - 1.0
        disc = B * B - 4 * A * Cq
        t = np.full(len(xy), np.inf)
        ok = disc >= 0
        t[ok] = (-B[ok] - np.sqrt(disc[ok])) / (2 * A[ok])
        t[~(t > 1e-6)] = np.inf
        t_best = np.minimum(t_best, t)
    z = t_best + rng.normal(scale=noise, size=len(xy))
    bad = ~np.isfinite(t_best) | (z <= 250) | (z >= max_depth) | ~valid | (rng.random(len(xy)) < drop)
    z[bad] = 0
    dep = np.round(z).astype(np.uint16)
    if return_person:
        return dep, (t_best < t_room) & (dep > 0)
    return dep


def mask_rgb(person, seed=7):
    """uint8 (H*W,3) colour image with the background zeroed the way Filtering.apply_segmentation leaves
    it (all-channels-zero == background, preprocessing/data.py:169); stand-in for Mask R-CNN."""
    rng = np.random.default_rng(seed)
    rgb = rng.integers(1, 256, size=(person.size, 3), dtype=np.uint8)
    rgb[~person] = 0
    return rgb


def person_mask_rgb(depth, E=None, seed=7):
    """uint8 (H*W,3) colour image with the background zeroed the way Filtering.apply_segmentation
    leaves it (all-channels-zero == background, preprocessing/data.py:169): here "person" = pixels
    whose depth is closer than the room surfaces would be (cheap stand-in for Mask R-CNN)."""
    rng = np.random.default_rng(seed)
    bg = render_depth(E, seed=0, noise=0.0, drop=0.0, person_shift=(0.0, 1e6, 0.0))
    rgb = rng.integers(1, 256, size=(depth.size, 3), dtype=np.uint8)
    person = (depth > 0) & ((bg == 0) | (depth.astype(np.int32) < bg.astype(np.int32) - 40))
    rgb[~person] = 0
    return rgb


def frame_cloud(seed=20250202, E=None):
    """float32 (K,3) valid points of one rendered frame in the camera frame (host math, fp64)."""
    xy = xy_table()
    dep = render_depth(E, seed=seed, xy=xy)
    z = dep.astype(np.float64)
    ok = dep > 0
    pts = np.stack([np.floor(np.nan_to_num(xy[:, 0]) * z + 0.5), np.floor(np.nan_to_num(xy[:, 1]) * z + 0.5), z], -1)
    ok &= (pts[:, 0] != 0) & (pts[:, 1] != 0)
    return pts[ok].astype(np.float32)


T_STAR = None


def t_star():
    """Config-2 ground truth: Ry(5 deg) Rx(2 deg), t = (30,-10,20) mm."""
    a, b = np.deg2rad(5.0), np.deg2rad(2.0)
    Ry = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
    Rx = np.array([[1, 0, 0], [0, np.cos(b), -np.sin(b)], [0, np.sin(b), np.cos(b)]])
    T = np.eye(4)
    T[:3, :3] = Ry @ Rx
    T[:3, 3] = [30.0, -10.0, 20.0]
    return T


def icp_pair(n=100_000, base=None):
    """Config 2: (source, target, T*) float32 clouds; T* maps source onto target."""
    if base is None:
        base = frame_cloud()
    r1, r2 = np.random.default_rng(1), np.random.default_rng(2)
    n = min(n, len(base))
    tgt = base[r1.choice(len(base), n, replace=False)]
    src0 = base[r2.choice(len(base), n, replace=False)].astype(np.float64)
    T = t_star()
    Ti = np.linalg.inv(T)
    src = src0 @ Ti[:3, :3].T + Ti[:3, 3] + r2.normal(scale=1.0, size=src0.shape)
    return src.astype(np.float32), tgt.astype(np.float32), T


def filter_cloud(n=1_000_000, seed=3):
    """Config 3: 45 % floor, 40 % person, 14 % wall, 1 % uniform outliers; float32 (n,3), mm."""
    rng = np.random.default_rng(seed)
    nf, npn, nw = int(0.45 * n), int(0.40 * n), int(0.14 * n)
    no = n - nf - npn - nw
    floor = np.stack([rng.uniform(-2000, 2000, nf), FLOOR_Y + rng.normal(scale=3.0, size=nf), rng.uniform(1000, 3500, nf)], -1)
    parts = []
    per = np.array_split(np.arange(npn), len(_PERSON))
    for (c, r), ids in zip(_PERSON, per):
        v = rng.normal(size=(len(ids), 3))
        v /= np.linalg.norm(v, axis=1, keepdims=True)
        p = np.array(c) + np.array([0.0, 0.0, 2000.0]) + v * np.array(r) + rng.normal(scale=1.5, size=v.shape)
        parts.append(p)
    person = np.concatenate(parts)
    person[:, 1] = np.minimum(person[:, 1], FLOOR_Y - 1.0)
    wall = np.stack([rng.uniform(-2000, 2000, nw), rng.uniform(-1500, FLOOR_Y, nw), 3500 + rng.normal(scale=3.0, size=nw)], -1)
    out = np.stack([rng.uniform(-2000, 2000, no), rng.uniform(-1500, FLOOR_Y, no), rng.uniform(1000, 3500, no)], -1)
    pts = np.concatenate([floor, person, wall, out])
    return pts[rng.permutation(len(pts))].astype(np.float32)


def coloured_pair(n, seed=0):
    """a smooth colour field painted on the scene: target samples + a displaced, noisy second sampling"""
    rng = np.random.default_rng(seed)
    c = filter_cloud(4 * n)
    c = c[(c[:, 1] < 880)][: 2 * n]                      # drop the floor plane: the person and the wall carry the colour
    field = lambda p: np.stack([0.5 + 0.4 * np.sin(p[:, 0] / 90.0) * np.cos(p[:, 1] / 120.0),
                                0.5 + 0.4 * np.cos(p[:, 2] / 150.0 + p[:, 0] / 200.0),
                                0.5 + 0.3 * np.sin(p[:, 1] / 70.0)], 1).astype(np.float32)
    tgt = c[:n].copy()
    src0 = c[n:2 * n].copy()
    T = t_star()
    Ti = np.linalg.inv(T)
    src = (src0.astype(np.float64) @ Ti[:3, :3].T + Ti[:3, 3]).astype(np.float32)
    return src, field(src0), tgt, field(tgt), T


def perturb(T, deg=3.0, mm=50.0, seed=0):
    """initial guess = ground-truth extrinsic perturbed by 3 deg / 50 mm"""
    rng = np.random.default_rng(seed)
    ax = rng.normal(size=3)
    ax /= np.linalg.norm(ax)
    a = np.deg2rad(deg)
    K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    R = np.eye(3) + np.sin(a) * K + (1 - np.cos(a)) * K @ K
    t = rng.normal(size=3)
    t *= mm / np.linalg.norm(t)
    P = np.eye(4)
    P[:3, :3] = R
    P[:3, 3] = t
    return P @ T


def sensor_ring(n_sensors, n_frames=1, xy=None, sensors=None, first_frame=0):
    """BASELINE configs 4/5: `n_sensors` cameras on a circle around the person, `n_frames` time frames (the person moves
    5 mm per frame).  `sensors`: the global sensor ids to render (default all).  Sensor 0 is the master.
    -> xy, depth (F, len(sensors), n_px) u16, rgb (F, len(sensors), n_px, 3) u8 (person mask colours),
       inits (n_sensors - 1 perturbed sub -> master transforms), truth (the exact ones)"""
    if xy is None:
        xy = xy_table()
    sensors = list(range(n_sensors)) if sensors is None else list(sensors)
    poses = [camera_pose(g, n_sensors) for g in range(n_sensors)]
    n_px = len(xy)
    depth = np.zeros((n_frames, len(sensors), n_px), np.uint16)
    rgb = np.zeros((n_frames, len(sensors), n_px, 3), np.uint8)
    for f in range(n_frames):
        t = first_frame + f
        for i, g in enumerate(sensors):
            d, person = render_depth(poses[g], person_shift=(5.0 * t, 0.0, 0.0), seed=100 + g + 1000 * t, xy=xy, return_person=True)
            depth[f, i] = d
            rgb[f, i] = mask_rgb(person, seed=7 + g)
    Einv = np.linalg.inv(poses[0])
    truth = [Einv @ poses[g] for g in range(1, n_sensors)]
    inits = [perturb(T, seed=g) for g, T in enumerate(truth)]
    return xy, depth, rgb, inits, truth


def small_xy(scale=4):
    """a 1/scale-resolution camera (144 x 160 for scale 4) with the same field of view: CPU-sized frames for rehearsals"""
    return xy_table(h=H // scale, w=W // scale, fx=FX / scale, fy=FY / scale, cx=CX / scale, cy=CY / scale)
