"""Seeded synthetic RGB-D stream, canned instance masks and pre-populated surfel maps.

This is the workload generator SURVEY.md 8(d) specifies for the BASELINE configs that have no
dataset in the container ("640x480 synthetic RGBD stream ...").  It is host-side numpy only and is
shared by the tests, bench.py and the CPU baseline so that every leg sees identical inputs.

Scene: an axis-aligned room (6 x 3 x 6 m) containing seeded boxes and spheres, analytic ray
casting, Lambert-shaded procedural texture.  Camera convention: x right, y down, z forward;
pose = camera-to-world 4x4 (row-major), as `currPose` in the reference
(elasticfusionpublic/Core/src/ElasticFusion.h:301).
"""
from __future__ import annotations

import numpy as np

SEED = 0x1F5
ROOM = np.array([3.0, 1.5, 3.0], dtype=np.float64)  # half extents


class Scene:
    def __init__(self, seed: int = SEED, n_boxes: int = 7, n_spheres: int = 5):
        rng = np.random.RandomState(seed)
        self.boxes = []  # (centre, half)
        for _ in range(n_boxes):
            half = rng.uniform(0.15, 0.45, 3)
            c = np.array([rng.uniform(-2.2, 2.2), 1.5 - half[1], rng.uniform(0.6, 2.4)])
            self.boxes.append((c, half))
        self.spheres = []
        for _ in range(n_spheres):
            r = rng.uniform(0.15, 0.35)
            c = np.array([rng.uniform(-2.0, 2.0), rng.uniform(-0.6, 1.5 - r), rng.uniform(0.8, 2.4)])
            self.spheres.append((c, r))
        self.n_objects = n_boxes + n_spheres
        self.base_col = rng.uniform(0.35, 0.95, (self.n_objects + 1, 3))
        self.obj_class = rng.randint(1, 80, self.n_objects + 1)
        self.light = np.array([0.3, -0.8, -0.5])
        self.light /= np.linalg.norm(self.light)

    # ---- texture: smooth + checker, always > 0
    def albedo(self, p: np.ndarray, obj: np.ndarray) -> np.ndarray:
        # soft-edged checker (tanh) + sinusoids: strong but band-limited gradients, like a real
        # camera image (hard edges break the photometric linearisation at coarse pyramid levels)
        sx = np.sin(2 * np.pi * p[..., 0] / 0.45) * np.sin(2 * np.pi * p[..., 1] / 0.45 + 0.7) * np.sin(2 * np.pi * p[..., 2] / 0.45 + 1.9)
        chk = np.tanh(3.0 * sx)
        wav = 0.5 * np.sin(9.0 * p[..., 0] + 1.3) * np.cos(7.0 * p[..., 1]) + 0.5 * np.sin(8.0 * p[..., 2] + 4.0 * p[..., 0])
        a = 0.55 + 0.28 * chk + 0.15 * wav
        return self.base_col[obj] * a[..., None]

    def cast(self, o: np.ndarray, d: np.ndarray):
        """o: (3,), d: (..., 3) world ray directions (unnormalised).  Returns t, normal, object id (0 = room)."""
        shape = d.shape[:-1]
        t_best = np.full(shape, np.inf)
        n_best = np.zeros(shape + (3,))
        obj = np.zeros(shape, dtype=np.int32)
        with np.errstate(divide="ignore", invalid="ignore"):
            # room: exit point of the enclosing box
            inv = 1.0 / d
            t1 = (-ROOM - o) * inv
            t2 = (ROOM - o) * inv
            tfar = np.maximum(t1, t2)
            ax = np.argmin(tfar, axis=-1)
            t_room = np.take_along_axis(tfar, ax[..., None], -1)[..., 0]
            t_best = t_room
            nrm = np.zeros(shape + (3,))
            sgn = -np.sign(np.take_along_axis(d, ax[..., None], -1)[..., 0])
            np.put_along_axis(nrm, ax[..., None], sgn[..., None], -1)
            n_best = nrm
            k = 1
            for c, half in self.boxes:
                ta = (c - half - o) * inv
                tb = (c + half - o) * inv
                tn = np.minimum(ta, tb)
                tf = np.maximum(ta, tb)
                axn = np.argmax(tn, axis=-1)
                tnear = np.take_along_axis(tn, axn[..., None], -1)[..., 0]
                tfar_b = tf.min(axis=-1)
                hit = (tnear < tfar_b) & (tnear > 1e-6) & (tnear < t_best)
                nb = np.zeros(shape + (3,))
                sg = -np.sign(np.take_along_axis(d, axn[..., None], -1)[..., 0])
                np.put_along_axis(nb, axn[..., None], sg[..., None], -1)
                t_best = np.where(hit, tnear, t_best)
                n_best = np.where(hit[..., None], nb, n_best)
                obj = np.where(hit, k, obj)
                k += 1
            for c, r in self.spheres:
                oc = o - c
                a = (d * d).sum(-1)
                b = 2.0 * (d * oc).sum(-1)
                cc = (oc * oc).sum() - r * r
                disc = b * b - 4 * a * cc
                ts = (-b - np.sqrt(np.maximum(disc, 0))) / (2 * a)
                hit = (disc > 0) & (ts > 1e-6) & (ts < t_best)
                p = o + d * ts[..., None]
                ns = (p - c) / r
                t_best = np.where(hit, ts, t_best)
                n_best = np.where(hit[..., None], ns, n_best)
                obj = np.where(hit, k, obj)
                k += 1
        return t_best, n_best, obj


def trajectory(n_frames: int, amp: float = 0.15, yaw_amp: float = 0.12, pitch_amp: float = 0.05) -> np.ndarray:
    """Closed smooth loop of camera-to-world poses, (n,4,4) float64; <=1 cm and <=0.5 deg per frame for n>=120."""
    poses = np.zeros((n_frames, 4, 4))
    for i in range(n_frames):
        s = 2 * np.pi * i / n_frames
        pos = np.array([amp * np.sin(s), 0.1 + 0.4 * amp * np.sin(2 * s), -1.2 + amp * (np.cos(s) - 1.0)])
        yaw = yaw_amp * np.sin(s + 0.5) - yaw_amp * np.sin(0.5)
        pitch = pitch_amp * np.sin(2 * s)
        cy, sy, cp, sp = np.cos(yaw), np.sin(yaw), np.cos(pitch), np.sin(pitch)
        Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
        Rx = np.array([[1, 0, 0], [0, cp, -sp], [0, sp, cp]])
        T = np.eye(4)
        T[:3, :3] = Ry @ Rx
        T[:3, 3] = pos
        poses[i] = T
    return poses


def _rot(yaw: float, pitch: float, roll: float = 0.0) -> np.ndarray:
    cy, sy, cp, sp, cr, sr = np.cos(yaw), np.sin(yaw), np.cos(pitch), np.sin(pitch), np.cos(roll), np.sin(roll)
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rx = np.array([[1, 0, 0], [0, cp, -sp], [0, sp, cp]])
    Rz = np.array([[cr, -sr, 0], [sr, cr, 0], [0, 0, 1]])
    return Ry @ Rx @ Rz


MOTION_PROFILES = ("still", "slow", "nominal", "fast", "jump", "spin", "dolly", "shake")


def trajectory_profile(kind: str, n_frames: int, seed: int = SEED) -> np.ndarray:
    """Camera-to-world poses (n,4,4) f64 for the parity sweep (tests/test_gpu_sweep.py): from a camera that does not move to one that leaves the tracker's
    correspondence gates (0.10 m / 20 degrees per frame, EF/Utils/RGBDOdometry.h:38-39) -- the view lists are then rebuilt every frame and tracking fails, identically
    on both sides.  Per-frame steps: still 0; slow 2 mm / 0.1 deg; nominal 1 cm / 0.5 deg; fast 4 cm / 4 deg; jump = nominal with ONE step of 15 cm / 25 deg; spin =
    rotation only, 6 deg; dolly = translation along the view only, 5 cm; shake = random sign, 2 cm / 2 deg."""
    rng = np.random.RandomState(seed + 101)
    start = np.array([rng.uniform(-0.3, 0.3), rng.uniform(0.0, 0.2), rng.uniform(-1.4, -1.0)])
    yaw0, pitch0 = rng.uniform(-0.15, 0.15), rng.uniform(-0.05, 0.05)
    dirv = rng.standard_normal(3); dirv /= np.linalg.norm(dirv)
    step = {"still": (0.0, 0.0), "slow": (0.002, 0.1), "nominal": (0.01, 0.5), "fast": (0.04, 4.0), "jump": (0.01, 0.5), "spin": (0.0, 6.0), "dolly": (0.05, 0.0),
            "shake": (0.02, 2.0)}[kind]
    poses = np.zeros((n_frames, 4, 4))
    pos, yaw, pitch, roll = start.copy(), yaw0, pitch0, 0.0
    for i in range(n_frames):
        T = np.eye(4)
        T[:3, :3] = _rot(yaw, pitch, roll)
        T[:3, 3] = pos
        poses[i] = T
        dt, dr = step
        if kind == "jump" and i == n_frames // 2:
            dt, dr = 0.15, 25.0
        sgn = rng.choice([-1.0, 1.0], 3) if kind == "shake" else np.ones(3)
        if kind == "dolly":
            pos = pos + T[:3, 2] * dt
        else:
            pos = pos + dirv * dt * sgn[0]
        yaw += np.deg2rad(dr) * 0.8 * sgn[1]
        pitch += np.deg2rad(dr) * 0.3 * sgn[2]
        roll += np.deg2rad(dr) * 0.5 * (sgn[0] if kind == "shake" else 1.0) if kind in ("fast", "shake", "spin") else 0.0
    return poses


def make_stream_from_poses(poses: np.ndarray, scene: "Scene", w: int, h: int, fx: float, fy: float, cx: float, cy: float, noise_seed: int | None = None):
    """As make_stream, on given camera-to-world poses and a given scene."""
    n = poses.shape[0]
    rng = np.random.RandomState(noise_seed) if noise_seed is not None else None
    rgb = np.zeros((n, h, w, 3), np.uint8)
    dep = np.zeros((n, h, w), np.uint16)
    obj = np.zeros((n, h, w), np.int32)
    for i in range(n):
        rgb[i], dep[i], obj[i] = render(scene, poses[i], w, h, fx, fy, cx, cy, rng)
    inv0 = np.linalg.inv(poses[0])
    return dict(rgb=rgb, depth=dep, obj=obj, poses=np.stack([inv0 @ p for p in poses]), poses_world=poses, scene=scene)


def render(scene: Scene, pose: np.ndarray, w: int, h: int, fx: float, fy: float, cx: float, cy: float,
           noise_rng: np.random.RandomState | None = None):
    """Returns rgb (h,w,3) u8, depth (h,w) u16 mm, obj (h,w) int32."""
    u, v = np.meshgrid(np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64))
    dc = np.stack([(u - cx) / fx, (v - cy) / fy, np.ones_like(u)], -1)
    R, t = pose[:3, :3], pose[:3, 3]
    dw = dc @ R.T
    tt, nrm, obj = scene.cast(t, dw)
    z = tt  # camera-frame z because dc.z == 1
    p = t + dw * tt[..., None]
    lam = np.clip((nrm * (-scene.light)).sum(-1), 0.0, 1.0)
    shade = 0.45 + 0.55 * lam
    colr = scene.albedo(p, obj) * shade[..., None]
    rgb = np.clip(np.round(colr * 255.0), 30, 255).astype(np.uint8)
    if noise_rng is not None:
        z = z + noise_rng.standard_normal(z.shape) * 0.001 * z * z
    d = np.where(np.isfinite(z) & (z < 12.0), np.round(z * 1000.0), 0).astype(np.uint16)
    return rgb, d, obj.astype(np.int32)


def make_stream(n_frames: int, w: int = 640, h: int = 480, fx: float = 528.0, fy: float = 528.0, cx: float = 320.0,
                cy: float = 240.0, noise: bool = True, seed: int = SEED, loop_len: int | None = None):
    """Returns dict(rgb=(n,h,w,3) u8, depth=(n,h,w) u16, obj=(n,h,w) i32, poses=(n,4,4) f64, scene=Scene).
    Poses are expressed relative to the first frame (the tracker starts at identity)."""
    scene = Scene(seed)
    poses = trajectory(loop_len or max(n_frames, 120))[:n_frames]
    rng = np.random.RandomState(seed + 1) if noise else None
    rgb = np.zeros((n_frames, h, w, 3), np.uint8)
    dep = np.zeros((n_frames, h, w), np.uint16)
    obj = np.zeros((n_frames, h, w), np.int32)
    for i in range(n_frames):
        rgb[i], dep[i], obj[i] = render(scene, poses[i], w, h, fx, fy, cx, cy, rng)
    inv0 = np.linalg.inv(poses[0])
    rel = np.stack([inv0 @ p for p in poses])
    return dict(rgb=rgb, depth=dep, obj=obj, poses=rel, poses_world=poses, scene=scene)


def canned_masks(obj: np.ndarray, scene: Scene, max_masks: int = 8, min_area: int = 400):
    """Masks a detector would return for one frame: silhouettes of the visible objects, sorted by
    area descending (build/mask_ori.py:117 of the reference), 0/255 uint8, plus class ids."""
    ids, counts = np.unique(obj[obj > 0], return_counts=True)
    order = np.argsort(-counts, kind="stable")
    masks, classes = [], []
    for k in order[:max_masks]:
        if counts[k] < min_area:
            continue
        masks.append(((obj == ids[k]) * 255).astype(np.uint8))
        classes.append(int(scene.obj_class[ids[k]]))
    if not masks:
        return np.zeros((0,) + obj.shape, np.uint8), np.zeros((0,), np.int32)
    return np.stack(masks), np.asarray(classes, np.int32)


def encode_votes(counts: np.ndarray) -> np.ndarray:
    """(n,96) int -> (n,48) float32 packed as float((a<<16)+b) (src/Core/InstanceFusionCuda.cu:22-39 of the reference)."""
    a = counts[:, 0::2].astype(np.int64)
    b = counts[:, 1::2].astype(np.int64)
    return ((a << 16) + b).astype(np.float32)


def make_map(n: int, scene: Scene, world_from_first: np.ndarray, tick: int, fx: float = 528.0, fy: float = 528.0,
             active_fraction: float | None = None, seed: int = SEED + 7, time_delta: int = 200, order: str = "random"):
    """N surfels sampled on the scene surfaces, expressed in the tracker's world frame (= first
    camera frame).  Returns dict of float32 arrays pc(n,4) nr(n,4) col(n,2) tm(n,2) ic(n,4) votes(n,48)."""
    rng = np.random.RandomState(seed)
    if active_fraction is None:
        active_fraction = min(1.0, 1.5e6 / max(n, 1))
    # area-weighted choice between the six room faces, boxes and spheres
    faces = []  # (area, kind, data)
    for ax in range(3):
        a, b = [k for k in range(3) if k != ax]
        area = 4 * ROOM[a] * ROOM[b]
        for sgn in (-1, 1):
            faces.append((area, "room", (ax, sgn)))
    for bi, (c, half) in enumerate(scene.boxes):
        for ax in range(3):
            a, b = [k for k in range(3) if k != ax]
            for sgn in (-1, 1):
                faces.append((4 * half[a] * half[b], "box", (bi, ax, sgn)))
    for si, (c, r) in enumerate(scene.spheres):
        faces.append((4 * np.pi * r * r, "sph", (si,)))
    areas = np.array([f[0] for f in faces])
    which = rng.choice(len(faces), size=n, p=areas / areas.sum())
    pos = np.zeros((n, 3))
    nrm = np.zeros((n, 3))
    obj = np.zeros(n, np.int32)
    for fi, (_, kind, data) in enumerate(faces):
        idx = np.nonzero(which == fi)[0]
        m = idx.size
        if m == 0:
            continue
        if kind == "room":
            ax, sgn = data
            p = rng.uniform(-1, 1, (m, 3)) * ROOM
            p[:, ax] = sgn * ROOM[ax]
            nn = np.zeros((m, 3))
            nn[:, ax] = -sgn
            pos[idx], nrm[idx] = p, nn
        elif kind == "box":
            bi, ax, sgn = data
            c, half = scene.boxes[bi]
            p = c + rng.uniform(-1, 1, (m, 3)) * half
            p[:, ax] = c[ax] + sgn * half[ax]
            nn = np.zeros((m, 3))
            nn[:, ax] = sgn
            pos[idx], nrm[idx], obj[idx] = p, nn, 1 + bi
        else:
            (si,) = data
            c, r = scene.spheres[si]
            v = rng.standard_normal((m, 3))
            v /= np.linalg.norm(v, axis=1, keepdims=True)
            pos[idx], nrm[idx], obj[idx] = c + r * v, v, 1 + len(scene.boxes) + si
    colr = scene.albedo(pos, obj) * 0.8
    rgbi = np.clip(np.round(colr * 255), 30, 255).astype(np.int64)
    packed = (rgbi[:, 0] << 16) + (rgbi[:, 1] << 8) + rgbi[:, 2]
    inv = np.linalg.inv(world_from_first)
    pos = pos @ inv[:3, :3].T + inv[:3, 3]
    nrm = nrm @ inv[:3, :3].T
    # radius: surfels.glsl getRadius at a nominal 2.5 m observation distance, frontal
    radius = np.sqrt(2.0) * rng.uniform(1.5, 3.5, n) / ((fx + fy) / 2.0)
    conf = rng.uniform(0.0, 20.0, n)
    active = rng.uniform(size=n) < active_fraction
    last = np.where(active, tick - rng.randint(1, 20, n), tick - time_delta - rng.randint(1, 500, n)).astype(np.float64)
    init = np.minimum(last, last - rng.randint(0, 300, n)).astype(np.float64)
    # keep confidently observed surfels stable, unstable ones recent (otherwise clean() drops them)
    conf = np.where((~active) | (last > tick - 15), conf, np.maximum(conf, 10.5))
    # votes: 70 % of the surfels all-zero, the rest with 1-3 non-zero counters in [1, 200]; packed
    # directly as float((a << 16) + b) to keep the host footprint at 192 B per surfel
    votes = np.zeros((n, 48), np.float32)
    voted = rng.uniform(size=n) < 0.3
    vi = np.nonzero(voted)[0]
    for _ in range(3):
        sel = vi[rng.uniform(size=vi.size) < 0.6]
        inst = rng.randint(1, 96, sel.size)
        cnt = rng.randint(1, 200, sel.size).astype(np.int64)
        cur = votes[sel, inst // 2].astype(np.int64)
        a, b = cur >> 16, cur & 0xFFFF
        a = np.where(inst % 2 == 0, cnt, a)
        b = np.where(inst % 2 == 1, cnt, b)
        votes[sel, inst // 2] = ((a << 16) + b).astype(np.float32)
    out = dict(
        pc=np.concatenate([pos, conf[:, None]], 1).astype(np.float32),
        nr=np.concatenate([nrm, radius[:, None]], 1).astype(np.float32),
        col=np.stack([packed.astype(np.float32), np.zeros(n, np.float32)], 1).astype(np.float32),
        tm=np.stack([init, last], 1).astype(np.float32),
        ic=np.zeros((n, 4), np.float32),
        votes=votes,
    )
    if order == "morton":
        # The same surfels in a spatially coherent map order (Morton code of the 8 cm voxel): what a map BUILT by the pipeline looks like -- surfels are appended frame by
        # frame in pixel order, so neighbours in the map are neighbours in space -- whereas "random" (the default, SURVEY.md 8d: "sampled uniformly") is the worst case for
        # every pass that gathers the visible part of the store.
        from .dist import _part1by2_10
        v = np.floor(out["pc"][:, :3] / np.float32(0.08)).astype(np.int64) + 512
        code = _part1by2_10(v[:, 0]).astype(np.int64) | (_part1by2_10(v[:, 1]).astype(np.int64) << 1) | (_part1by2_10(v[:, 2]).astype(np.int64) << 2)
        perm = np.argsort(code, kind="stable")
        out = {k: np.ascontiguousarray(a[perm]) for k, a in out.items()}
    return out
