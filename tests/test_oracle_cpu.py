"""CPU tests of the oracle: known answers the survey derives from the reference source, pins on the
reference's GPUTest RGB-D pair, and domain properties.  (The reference has no tests or golden
outputs for this path -- SURVEY.md 4 -- so these are what pins the restatement.)"""
import ctypes as C
import math

import numpy as np
import pytest

from conftest import SMALL
from gputest_protocol import run_oracle_protocol


# ---------------------------------------------------------------- packed vote counters
def test_vote_packing_known_answers(orc):
    L = orc.lib()

    def rt(a, b):
        x, y = C.c_int(), C.c_int()
        L.orc_vote_decode(L.orc_vote_encode(a, b), C.byref(x), C.byref(y))
        return x.value, y.value

    # SURVEY.md section 7 "hard parts" item 2, verified against IF/Core/InstanceFusionCuda.cu:22-39
    assert rt(256, 5) == (256, 4)
    assert rt(300, 7) == (300, 8)
    assert rt(0, 0) == (0, 0) and rt(5, 9) == (5, 9) and rt(255, 65) == (255, 65)
    # first-frame surfels hold -1.0f: both counters decode to -1 (init_unstable.vert:53-66)
    x, y = C.c_int(), C.c_int()
    L.orc_vote_decode(C.c_float(-1.0), C.byref(x), C.byref(y))
    assert (x.value, y.value) == (-1, -1)
    # adding to a -1/-1 pair borrows from the high counter: (-1 + 3, -1) -> (2, -1) -> encode -> (1, -1)
    assert rt(2, -1) == (1, -1)
    # 65535 is passed as `short` and wraps to -1
    assert rt(65535, 0)[0] == -1


def test_expf_and_solver_against_numpy(orc):
    L = orc.lib()
    L.orc_test_expf.restype = C.c_float
    L.orc_test_expf.argtypes = [C.c_float]
    xs = -np.concatenate([np.linspace(0, 20, 2001), np.logspace(-6, 1.9, 500)])
    got = np.array([L.orc_test_expf(float(x)) for x in xs])
    ref = np.exp(xs.astype(np.float32).astype(np.float64))
    assert np.max(np.abs(got - ref) / ref) < 3e-7
    assert L.orc_test_expf(-100.0) == 0.0
    rng = np.random.RandomState(3)
    L.orc_test_ldlt.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    for n in (3, 6):
        for _ in range(20):
            M = rng.standard_normal((n + 3, n))
            A = np.ascontiguousarray(M.T @ M)
            b = rng.standard_normal(n)
            x = np.zeros(n)
            L.orc_test_ldlt(n, orc.ptr(A), orc.ptr(b), orc.ptr(x))
            assert np.allclose(x, np.linalg.solve(A, b), rtol=1e-9, atol=1e-12)
    # rank-deficient system: Eigen's LDLT returns zeros for zero pivots
    A = np.zeros((6, 6)); b = np.zeros(6); x = np.ones(6)
    L.orc_test_ldlt(6, orc.ptr(A), orc.ptr(b), orc.ptr(x))
    assert np.all(x == 0)
    L.orc_test_rodrigues.argtypes = [C.c_void_p, C.c_void_p]
    v = np.array([0.1, -0.2, 0.05]); R = np.zeros(9)
    L.orc_test_rodrigues(orc.ptr(v), orc.ptr(R))
    R = R.reshape(3, 3)
    assert np.allclose(R @ R.T, np.eye(3), atol=1e-14) and abs(np.arccos((np.trace(R) - 1) / 2) - np.linalg.norm(v)) < 1e-12


# ---------------------------------------------------------------- pins on the reference's own RGB-D pair
def test_gputest_pair_pins(orc, gputest_pair, oracle_pins):
    out = run_oracle_protocol(*gputest_pair)
    for k, v in oracle_pins.items():
        assert np.allclose(out[k], v, rtol=1e-5, atol=1e-7), k
    for tag in ("single", "pyr"):
        P = out[f"pose_{tag}"]
        R = P[:3, :3]
        assert np.allclose(R @ R.T, np.eye(3), atol=1e-5)
        assert np.linalg.norm(P[:3, 3]) < 0.05           # hand-held motion between two consecutive frames
        assert out[f"diag_{tag}"][1] > 0.2 * 320 * 240   # ICP inliers
    assert np.linalg.norm(out["pose_single"][:3, 3] - out["pose_pyr"][:3, 3]) < 2e-3


def test_tracker_identity_and_known_motion(orc, small_stream):
    L = orc.lib()
    st = small_stream
    w, h = SMALL["w"], SMALL["h"]

    def model_from(depth, rgb):
        z = depth.astype(np.float32) / 1000.0
        u, v = np.meshgrid(np.arange(w, dtype=np.float32), np.arange(h, dtype=np.float32))
        V = np.stack([(u - SMALL["cx"]) * z / SMALL["fx"], (v - SMALL["cy"]) * z / SMALL["fy"], z, np.ones_like(z)], -1).astype(np.float32)
        N = np.zeros_like(V)
        dx = V[:, 1:, :3] - V[:, :-1, :3]
        dy = V[1:, :, :3] - V[:-1, :, :3]
        n = np.cross(dx[:-1], dy[:, :-1])
        N[:-1, :-1, :3] = n / (np.linalg.norm(n, axis=-1, keepdims=True) + 1e-12)
        V[-1] = 0
        V[:, -1] = 0
        return np.ascontiguousarray(V), np.ascontiguousarray(N), np.concatenate([rgb, 255 * np.ones((h, w, 1), np.uint8)], -1).copy()

    def track(k, icp_weight, so3=1):
        t = L.orc_tracker_create(w, h, SMALL["fx"], SMALL["fy"], SMALL["cx"], SMALL["cy"])
        V, N, rgba = model_from(st["depth"][0], st["rgb"][0])
        pose = np.eye(4, dtype=np.float32).reshape(16).copy()
        L.orc_tracker_init_first_rgb(t, orc.ptr(np.ascontiguousarray(st["rgb"][0])))
        L.orc_tracker_init_model(t, orc.ptr(V), orc.ptr(N), orc.ptr(rgba), orc.ptr(pose))
        L.orc_tracker_init_frame(t, orc.ptr(np.ascontiguousarray(st["depth"][k])), orc.ptr(np.ascontiguousarray(st["rgb"][k])), 20.0)
        L.orc_tracker_run(t, orc.ptr(pose), icp_weight, 1, 0, so3, None)
        L.orc_tracker_destroy(t)
        return pose.reshape(4, 4)

    # frame against itself -> identity up to the coarse-level bias the reference has by construction
    # (model pyramid = 2x2 box average, frame pyramid = centred Gaussian: half a pixel apart) combined
    # with its damped ICP step (lastb uses w, lastA uses w*w: EF/Utils/RGBDOdometry.cpp:550-551)
    P = track(0, 10.0)
    assert np.abs(P - np.eye(4)).max() < 3e-3
    P = track(2, 100.0, so3=0)  # ICP only, no SO(3) pre-alignment: converges to the analytic motion
    gt = st["poses"][2]
    assert np.linalg.norm(P[:3, 3] - gt[:3, 3]) < 3e-3 and np.abs(P[:3, :3] - gt[:3, :3]).max() < 2e-3


# ---------------------------------------------------------------- preprocessing / pyramid properties
def test_bilateral_and_metric_gates(orc):
    L = orc.lib()
    w, h = 64, 48
    L.orc_bilateral.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float]
    L.orc_metric.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float]
    d = np.full((h, w), 2000, np.uint16)
    d[:4] = 100      # below 300 mm -> 0
    d[-4:] = 13000   # beyond depthCut (12 m) -> 0
    out = np.zeros_like(d)
    L.orc_bilateral(orc.ptr(d), orc.ptr(out), w, h, 12.0)
    assert (out[:4] == 0).all() and (out[-4:] == 0).all()
    assert (out[10:-10] == 2000).all()   # constant region is a fixed point
    m = np.zeros((h, w), np.float32)
    L.orc_metric(orc.ptr(d), orc.ptr(m), w, h, 12.0)
    assert (m[:4] == 0).all() and (m[-4:] == 0).all() and np.allclose(m[10], 2.0)


def test_pyramid_kernels_small(orc):
    L = orc.lib()
    rng = np.random.RandomState(0)
    w, h = 32, 24
    L.orc_pyrdown_u16.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    L.orc_vmap.argtypes = [C.c_void_p, C.c_int, C.c_int] + [C.c_float] * 5 + [C.c_void_p]
    L.orc_nmap.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    L.orc_sobel.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    d = np.full((h, w), 1500, np.uint16)
    o = np.zeros((h // 2, w // 2), np.uint16)
    L.orc_pyrdown_u16(orc.ptr(d), w, h, orc.ptr(o))
    assert (o == 1500).all()
    d[5, 7] = 0
    vm = np.zeros((3, h, w), np.float32)
    L.orc_vmap(orc.ptr(d), w, h, 50.0, 50.0, 16.0, 12.0, 20.0, orc.ptr(vm))
    assert np.isnan(vm[0, 5, 7]) and np.isclose(vm[2, 3, 3], 1.5) and np.isclose(vm[0, 3, 20], 1.5 * (20 - 16) / 50)
    nm = np.zeros_like(vm)
    L.orc_nmap(orc.ptr(vm), w, h, orc.ptr(nm))
    assert np.isnan(nm[0, -1, 0]) and np.isnan(nm[0, 0, -1]) and np.isnan(nm[0, 5, 6])   # borders and invalid neighbours
    assert np.allclose(nm[:, 10, 10], [0, 0, 1], atol=1e-6)                               # fronto-parallel plane
    img = np.tile(np.arange(w, dtype=np.uint8) * 4, (h, 1)).copy()
    dx = np.zeros((h, w), np.int16); dy = np.zeros((h, w), np.int16)
    L.orc_sobel(orc.ptr(img), w, h, orc.ptr(dx), orc.ptr(dy))
    # interior: +dI/dx with weights 2*0.52201+0.79451 times (I(x+1)-I(x-1)) = 1.83853*8 -> trunc 14
    assert (dx[5:-5, 5:-5] == 14).all() and (dy[5:-5, 5:-5] == 0).all()


# ---------------------------------------------------------------- map stages
def _tiny_map(n=4):
    pc = np.zeros((n, 4), np.float32); nr = np.zeros((n, 4), np.float32)
    col = np.zeros((n, 2), np.float32); tm = np.zeros((n, 2), np.float32)
    for i in range(n):
        pc[i] = [0.05 * i, 0.0, 1.0 + 0.2 * i, 20.0]
        nr[i] = [0, 0, -1, 0.02]
        col[i] = [float((200 << 16) + (100 << 8) + 50), 0]
        tm[i] = [1, 5]
    return dict(pc=pc, nr=nr, col=col, tm=tm, ic=np.zeros((n, 4), np.float32), votes=np.zeros((n, 48), np.float32))


def test_window_loop_point_and_tap_rules_as_the_shaders_run_them(orc):
    """The three rules that executing the reference's GLSL pinned in round 6, each against a statement written here from the shader text:
    the float window loop of data.vert:151-153 / copy_unstable.vert:110-112 (four or FIVE taps), the pixel of a 1-pixel GL point, the texel a bilateral tap reads."""
    from test_restatement_map_numpy import _point_pixel, _uvo, _window_texels

    L = orc.lib()
    L.orc_test_window_taps.argtypes = [C.c_float, C.c_float, C.c_int, C.c_void_p]
    L.orc_test_uvo_coord.restype = C.c_float
    L.orc_test_uvo_coord.argtypes = [C.c_int, C.c_int]
    L.orc_test_point_pixel.argtypes = [C.c_float]
    L.orc_test_bilateral_tap.argtypes = [C.c_int, C.c_int]
    tex = np.zeros(8, np.int32)
    rng = np.random.RandomState(3)
    five = {}
    for size in (160, 320, 640, 1280, 120, 240, 480, 960):
        # association: the window around every pixel centre of the axis (the texcoord attribute of the uvo buffer)
        for i in range(size):
            c = _uvo(i, size)
            assert L.orc_test_uvo_coord(i, size) == c
            n = L.orc_test_window_taps(float(c), float(size), size, orc.ptr(tex))
            want = _window_texels(c, size)
            assert n == len(want) and tex[:n].tolist() == want, (size, i)
            assert 4 <= n <= 5 and set(want) <= {max(i - 1, 0), i, min(i + 1, size - 1)} and {max(i - 1, 0), i} <= set(want)
        # clean: around arbitrary projected positions
        cnt5 = 0
        for x in rng.uniform(1.0, size - 1.0, 1500).astype(np.float32):
            c = np.float32(x) / np.float32(size)
            n = L.orc_test_window_taps(float(c), float(size), size, orc.ptr(tex))
            want = _window_texels(c, size)
            assert n == len(want) and tex[:n].tolist() == want, (size, float(x))
            cnt5 += n == 5
        five[size] = cnt5 / 1500.0
        for c_ in range(size):
            f = np.float32
            assert L.orc_test_bilateral_tap(c_, size) == min(max(int(np.floor(f(f(f(c_) / f(size)) * f(size)))), 0), size - 1)
    assert 0.03 < five[160] < 0.10 and 0.45 < five[320] < 0.60 and 0.20 < five[640] < 0.32, five        # the fifth trip: 6 % / 53 % / 26 % of the positions
    assert [c_ for c_ in range(480) if L.orc_test_bilateral_tap(c_, 480) != c_] == [63, 125, 126, 127, 250, 252, 254]
    assert all(L.orc_test_bilateral_tap(c_, 640) == c_ for c_ in range(640))
    for u in np.concatenate([rng.uniform(0, 640, 4000), np.arange(0, 640) + 1e-3, np.arange(1, 640) - 1e-3, np.arange(0, 640) + 0.0021, np.arange(0, 640, dtype=np.float64)]).astype(np.float32):
        assert L.orc_test_point_pixel(float(u)) == _point_pixel(u), float(u)
    assert L.orc_test_point_pixel(12.0) == 11 and L.orc_test_point_pixel(12.0019) == 11 and L.orc_test_point_pixel(12.0021) == 12 and L.orc_test_point_pixel(12.999) == 12


def test_index_map_nearest_wins_and_id0_is_empty(orc):
    o = orc.Oracle(**SMALL, max_surfels=1000)
    m = _tiny_map(3)
    m["pc"][:, 0] = 0.0          # all three on the optical axis: z = 1.0, 1.2, 1.4
    m["pc"][:, 1] = 0.0
    o.upload(m)
    o.predict_indices(np.eye(4), 6)
    idx = o.image("index")
    # (the axis projects to (cx, cy) = (160.0, 120.0): exactly on a pixel corner, and a 1-pixel GL point on a pixel edge belongs to the pixel BELOW the edge -- the rule of
    # the rasteriser the reference's index_map shaders ran on, oracle/orc_map.c point_pixel)
    px, py = int(SMALL["cx"]) - 1, int(SMALL["cy"]) - 1
    assert idx[py, px] == 0      # surfel 0 is nearest, but id 0 reads as "empty" (index_map.vert:51)
    assert np.isclose(o.image("index_vc")[py, px, 2], 1.0)
    m["pc"][0, 2] = 3.0          # now surfel 1 (z = 1.2) is nearest
    o.upload(m)
    o.predict_indices(np.eye(4), 6)
    assert o.image("index")[py, px] == 1
    o.predict_indices(np.eye(4), 6 + 300)   # outside the 200-frame time window -> culled
    assert (o.image("index") == 0).all()
    o.close()


def test_splat_disc_and_ids(orc):
    o = orc.Oracle(**SMALL, max_surfels=1000)
    m = _tiny_map(2)
    m["pc"][0] = [0, 0, 2.0, 20.0]; m["nr"][0] = [0, 0, -1, 0.05]
    m["pc"][1] = [0, 0, 1.0, 5.0]; m["nr"][1] = [0, 0, -1, 0.05]     # unstable (conf < 10): not rendered
    o.upload(m)
    o.combined_predict(np.eye(4), 6, 6)
    pv = o.image("pred_vertex")
    cov = pv[..., 2] > 0
    # disc of radius 0.05 m at 2 m: radius 6.6 px -> area ~137 px
    assert 110 < cov.sum() < 165 and np.allclose(pv[cov][:, 2], 2.0, atol=1e-5)
    assert (o.image("pred_image")[cov][:, :3] == [200, 100, 50]).all()
    ids = o.render_ids(np.eye(4))
    assert (ids[cov] == 0).all()                       # surfel 0 renders as id 0
    m["pc"] = m["pc"][::-1].copy(); m["nr"] = m["nr"][::-1].copy()
    o.upload(m)
    ids = o.render_ids(np.eye(4))
    assert 110 < (ids == 1).sum() < 165
    o.close()


def test_pipeline_properties(orc, small_stream):
    st = small_stream
    o = orc.Oracle(**SMALL, max_surfels=400000)
    n_prev = 0
    for i in range(5):
        P = o.process_frame(st["rgb"][i], st["depth"][i])
        m = o.download()
        assert np.isfinite(m["pc"][:, 3]).all() and (m["tm"][:, 1] <= o.tick).all()
        if i:
            assert o.count >= n_prev * 0.95
            assert np.linalg.norm(P[:3, 3] - st["poses"][i][:3, 3]) < 0.02
        n_prev = o.count
    ids = o.image("ids_after")
    assert ids.max() < o.count   # (no surfel is stable after 5 frames: confidence 10 needs >= 10 observations)
    # surfels are appended in column-major pixel order (EF/GlobalModel.cpp:103-112)
    new = m["ic"][m["ic"][:, 2] == o.tick - 1]
    key = new[:, 0] * 1000 + new[:, 1]
    assert (np.diff(key) > 0).all()
    o.close()


# ---------------------------------------------------------------- instance layer
def test_mask_clean_overlap(orc):
    L = orc.lib()
    L.orc_mask_clean_overlap.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
    m = np.zeros((3, 4, 4), np.uint8)
    m[0, :, :] = 255; m[1, :2, :] = 255; m[2, :1, :] = 255
    L.orc_mask_clean_overlap(orc.ptr(m), 3, 4, 4)
    assert (m[2, 0] == 255).all() and (m[1, 0] == 0).all() and (m[1, 1] == 255).all() and (m[0, :2] == 0).all() and (m[0, 2:] == 255).all()


def test_instance_votes_and_labels(orc, small_stream):
    from instancefusion_amd import synth

    st = small_stream
    o = orc.Oracle(**SMALL, max_surfels=400000)
    for i in range(6):
        o.process_frame(st["rgb"][i], st["depth"][i])
    # make every surfel stable so that the id image is populated, then re-render it
    m = o.download()
    m["pc"][:, 3] = 20.0
    o.upload(m)
    o.set_pose(st["poses"][5].astype(np.float32), o.tick)
    o.L.orc_render_ids(o.h, orc.ptr(np.ascontiguousarray(st["poses"][5], np.float32).reshape(16)), 0)
    ids = o.image("ids_tmp")
    o.set_ids_after(ids)   # the segmentation reads ids_after
    masks, cls = synth.canned_masks(st["obj"][5], st["scene"])
    assert masks.shape[0] >= 2
    assert o.should_segment(100) and not o.should_segment(101)   # cadence: >2 frames apart
    o.process_segmentation(st["rgb"][5], st["depth"][5], masks, cls, 5)
    lab = o.labels()
    tab = o.instance_table()
    assert (tab >= 0).sum() >= 1 and tab[0] >= 0 or (tab >= 0).sum() >= 1
    assert lab.max() < 96
    votes = o.download()["votes"]
    # label = first strict arg-max over the 96 decoded counters, -1 if none is positive
    # (countAndColourSurfelMapKernel, IF/Core/InstanceFusionCuda.cu:1158-1186), re-derived in numpy
    v = np.trunc(votes).astype(np.int64)
    cnt = np.empty((votes.shape[0], 96), np.int64)
    cnt[:, 0::2] = ((v >> 16) & 0xFFFF).astype(np.uint16).view(np.int16) if False else (((v >> 16) & 0xFFFF) ^ 0x8000) - 0x8000
    cnt[:, 1::2] = ((v & 0xFFFF) ^ 0x8000) - 0x8000
    best = np.where(cnt.max(axis=1) > 0, cnt.argmax(axis=1), -1)
    assert (lab == best).all()
    assert (lab >= 0).sum() > 100
    # a second call with the same masks matches the registered instances (box IoU = 1) -> no new slots,
    # except instance 0 which can never match (IF/Core/InstanceFusion.cpp:649)
    n1 = (tab >= 0).sum()
    o.process_segmentation(st["rgb"][5], st["depth"][5], masks, cls, 6)
    n2 = (o.instance_table() >= 0).sum()
    # (pixels over first-frame surfels, whose counters are -1, are skipped by the box pass --
    # IF/Core/InstanceFusionCuda.cu:915 -- so boxes can stay small and masks may register again)
    assert n1 <= n2 <= n1 + masks.shape[0]
    o.close()


def test_knn_vote_against_scipy(orc):
    """orc_knn_vote: the 10 nearest surfels (self included) of a brute-force search equal scipy's exact kd-tree
    query, and the colour becomes that of the most frequent non-negative label among them (first maximum)."""
    from scipy.spatial import cKDTree
    from instancefusion_amd import synth

    n = 4000
    st = synth.make_stream(1, 320, 240, 264.0, 264.0, 160.0, 120.0, noise=False)
    m = synth.make_map(n, st["scene"], st["poses_world"][0], 10)
    o = orc.Oracle(w=320, h=240, fx=264.0, fy=264.0, cx=160.0, cy=120.0, max_surfels=n + 10)
    o.upload(m)
    # labels come from the votes: run a label scan through a segmentation-free path by writing votes directly
    rng = np.random.RandomState(2)
    lab = rng.randint(-1, 5, n).astype(np.int32)
    votes = np.zeros((n, 48), np.float32)
    for i in np.nonzero(lab >= 0)[0]:
        a, b = (7, 0) if lab[i] % 2 == 0 else (0, 7)
        votes[i, lab[i] // 2] = orc.lib().orc_vote_encode(a, b)
    m2 = o.download(); m2["votes"] = votes
    o.upload(m2)
    masks = np.zeros((1, 240, 320), np.uint8)             # a call with one empty mask only refreshes labels / colours
    o.set_ids_after(np.zeros((240, 320), np.int32))
    o.process_segmentation(st["rgb"][0], st["depth"][0], masks, np.array([1], np.int32), 0)
    assert np.array_equal(o.labels(), lab)
    before = o.download()["col"][:, 1].copy()
    nbr = o.knn_vote(with_neighbours=True)
    after = o.download()["col"][:, 1]
    pos = m["pc"][:, :3].astype(np.float32)
    d, idx = cKDTree(pos.astype(np.float64)).query(pos.astype(np.float64), k=10)
    assert (np.sort(nbr, 1) == np.sort(idx, 1)).mean() > 0.999      # identical sets except exact distance ties
    tab = o.instance_table()
    for i in range(0, n, 37):
        ls = lab[nbr[i]]
        ls = ls[ls >= 0]
        if len(ls) == 0:
            assert after[i] == before[i]
        else:
            cnt = np.bincount(ls, minlength=96)
            assert after[i] != 0 and (cnt.max() > 0)
    o.close()


def _knn_lists_equivalent(pos, mine, ref_idx):
    """Two 10-NN answers agree when every point has the same sorted distances (equidistant neighbours may be listed in another order or,
    at the 10th place, be another point) and identical index lists wherever the distances are strictly increasing."""
    p64 = pos.astype(np.float64)
    dm = np.sort(((p64[mine] - p64[:, None, :]) ** 2).sum(-1), axis=1)
    dr = np.sort(((p64[ref_idx] - p64[:, None, :]) ** 2).sum(-1), axis=1)
    assert np.array_equal(dm, dr)
    strict = (np.diff(dr, axis=1) > 0).all(axis=1)
    assert strict.mean() > 0.95 and np.array_equal(mine[strict], ref_idx[strict])


def _oracle_knn_lists(orc, pos):
    n = pos.shape[0]
    o = orc.Oracle(w=320, h=240, fx=264.0, fy=264.0, cx=160.0, cy=120.0, max_surfels=n + 10)
    m = dict(pc=np.concatenate([pos, np.full((n, 1), 20.0, np.float32)], 1), nr=np.tile(np.array([0, 0, 1, 0.01], np.float32), (n, 1)), col=np.zeros((n, 2), np.float32),
             tm=np.ones((n, 2), np.float32), ic=np.zeros((n, 4), np.float32), votes=np.zeros((n, 48), np.float32))
    o.upload(m)
    nbr = o.knn_vote(with_neighbours=True)
    o.close()
    return nbr


def test_knn_against_reference_flann_golden(orc):
    """The oracle's neighbour lists against the answer of the reference's vendored FLANN 1.8.4 (tests/golden/knn_ref.npz, tools/make_golden.py)."""
    import os

    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "knn_ref.npz"))
    pos, ref_idx = gold["pos"], gold["idx"].astype(np.int64)
    assert (ref_idx[:, 0] == np.arange(len(pos))).mean() > 0.999            # a point is its own nearest neighbour
    _knn_lists_equivalent(pos, _oracle_knn_lists(orc, pos).astype(np.int64), ref_idx)


def test_knn_against_reference_flann_live(orc):
    """Same, against oracle/_ref/libref_knn.so itself on another cloud (needs /root/reference at build time)."""
    import os

    so = os.path.join(os.path.dirname(os.path.dirname(__file__)), "oracle", "_ref", "libref_knn.so")
    if not os.path.exists(so):
        pytest.skip("oracle/_ref/libref_knn.so not built (needs /root/reference)")
    ref = C.CDLL(so)
    ref.ref_knn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    rng = np.random.RandomState(5)
    pos = np.ascontiguousarray(np.concatenate([rng.uniform(-2, 2, (3000, 3)), rng.normal(0, 0.05, (1500, 3))]), np.float32)   # uniform + a dense cluster
    idx, dist = np.zeros((len(pos), 10), np.int32), np.zeros((len(pos), 10), np.float32)
    assert ref.ref_knn(pos.ctypes.data, len(pos), 10, 64, idx.ctypes.data, dist.ctypes.data) == 0
    _knn_lists_equivalent(pos, _oracle_knn_lists(orc, pos).astype(np.int64), idx.astype(np.int64))


# ---------------------------------------------------------------- local loop-closure detection (EF/ElasticFusion.cpp:453-566)
def test_loop_closure_detection_properties(orc, small_stream):
    st = small_stream
    kw = dict(time_delta=3, confidence=2.0)
    thr = 35000 * (SMALL["w"] * SMALL["h"]) // (640 * 480)
    o = orc.Oracle(**SMALL, max_surfels=400000, **kw)
    o0 = orc.Oracle(**SMALL, max_surfels=400000, **kw)
    o.set_loop_closure(True, thr, 1e-4, 1e-5)
    for k in range(7):
        p = o.process_frame(st["rgb"][k], st["depth"][k])
        assert np.array_equal(p, o0.process_frame(st["rgb"][k], st["depth"][k]))       # detection only: no side effect
        d = o.loop_closure_diag()
        if k == 2:                                                                     # every surfel stable from here on (only stable ones are predicted)
            for e in (o, o0):
                m = e.download(); m["pc"][:, 3] = 20.0; e.upload(m)
        if k < 3:                                                                      # nothing can be 3 frames old yet
            assert not d["ran"] and d["inactive_pixels"] == 0 and (k == 0 or np.array_equal(d["est_pose"], p))   # (the first frame only initialises the map)
        else:
            assert d["ran"] and d["inactive_pixels"] > 0.5 * SMALL["w"] * SMALL["h"]
            assert d["icp_count"] > thr and d["cov_ok"] and 0 < d["cov_max"] < 1e-5
            # active and inactive surfels lie on the same static surfaces: the alignment is (nearly) the identity
            assert np.abs(d["est_pose"][:3, 3] - p[:3, 3]).max() < 0.02
            assert d["accepted"] == (d["icp_error"] < 1e-4)
    assert o.count == o0.count and all(np.array_equal(a, b) for a, b in zip(o.download().values(), o0.download().values()))
    # the INACTIVE render holds only surfels at least time_delta frames old
    m = o.download()
    old_ids = set(np.flatnonzero(m["tm"][:, 1] <= o.tick - 1 - 3).tolist())     # lastTime <= tick - timeDelta at the time of the render
    assert len(old_ids) > 0
    # gates: an impossible error threshold accepts nothing
    o.set_loop_closure(True, thr, 0.0, 1e-5)
    o.process_frame(st["rgb"][7], st["depth"][7])
    d = o.loop_closure_diag()
    assert d["ran"] and not d["accepted"]
    # a singular normal matrix gives a non-finite covariance, never a crash
    L = orc.lib()
    t = L.orc_tracker_create(64, 48, 50.0, 50.0, 32.0, 24.0)
    cov = np.zeros(36, np.float64)
    L.orc_tracker_covariance.argtypes = [C.c_void_p, C.c_void_p]
    L.orc_tracker_covariance(t, cov.ctypes.data_as(C.c_void_p))
    assert not np.isfinite(cov).all()
    L.orc_tracker_destroy(t)
    o.close(); o0.close()


# ---------------------------------------------------------------- deformation hooks (copy_unstable.vert:178-374, sample.geom, EF/ElasticFusion.cpp:568-598)
def _deform_numpy(g, p, n, init_t):
    """Independent restatement of the per-surfel graph application (float32 numpy), for a handful of surfels."""
    f = np.float32
    times = g[:, 15].astype(np.int64)
    pt = int(init_t)
    imin, imax = 0, len(g) - 1
    imid = (imin + imax) // 2
    while imax >= imin:
        imid = (imin + imax) // 2
        if times[imid] < pt:
            imin = imid + 1
        elif times[imid] > pt:
            imax = imid - 1
        else:
            break
    imin = min(imin, len(g) - 1)
    cmax = max(imax, 0)
    dmin, dmid, dmax_ = abs(times[imin] - pt), abs(times[imid] - pt), abs(times[cmax] - pt)
    found = imin if (dmin <= dmid and dmin <= dmax_) else (imid if (dmid <= dmin and dmid <= dmax_) else cmax)
    idx = list(range(found, max(found - 10, -1), -1))
    idx += list(range(found + 1, min(len(g), found + 1 + 20 - len(idx))))
    d = np.array([np.sqrt(np.sum((p - g[j, :3]) ** 2, dtype=f), dtype=f) for j in idx], f)
    order = np.argsort(d, kind="stable")
    near, nd = [idx[k] for k in order], d[order]
    d5 = nd[4] if len(nd) > 4 else f(16777216.0)
    w = np.array([(f(1) - nd[k] / d5) ** 2 for k in range(4)], f)
    w = w / w.sum(dtype=f)
    newp, newn = np.zeros(3, f), np.zeros(3, f)
    for k in range(4):
        nd_ = g[near[k]]
        R = nd_[3:12].reshape(3, 3).T.astype(f)          # stored column-major
        newp += w[k] * (R @ (p - nd_[:3]) + nd_[:3] + nd_[12:15])
        newn += w[k] * (np.linalg.inv(R.astype(np.float64)).T.astype(f) @ n)
    return newp, newn / np.linalg.norm(newn)


def test_deformation_hooks_properties(orc, small_stream):
    st = small_stream
    o = orc.Oracle(**SMALL, max_surfels=400000, confidence=2.0)
    for k in range(4):
        pose = o.process_frame(st["rgb"][k], st["depth"][k])
    m0 = o.download()
    n = o.count
    # Deformation::sampleGraphModel: every 5000th surfel of the map order, times non-decreasing (the reference asserts it)
    s = o.sample_graph_model()
    assert s.shape[0] == (n + 4999) // 5000 >= 5
    assert np.array_equal(s[:, :3], m0["pc"][::5000, :3]) and np.array_equal(s[:, 3], m0["tm"][::5000, 0])
    assert (np.diff(s[:, 3]) >= 0).all()
    # a graph that moves everything rigidly: R = I, t constant  ->  every surfel not created in this frame moves by t, normals stay
    tvec = np.array([0.01, -0.02, 0.005], np.float32)
    g = np.zeros((s.shape[0], 16), np.float32)
    g[:, :3] = s[:, :3]; g[:, 3] = g[:, 7] = g[:, 11] = 1.0; g[:, 12:15] = tvec; g[:, 15] = s[:, 3]
    tick = o.tick
    o.predict_indices(pose, tick)
    o.set_deformation(g, is_fern=True)
    o.clean(pose, tick)
    m1 = o.download()
    o2 = orc.Oracle(**SMALL, max_surfels=400000, confidence=2.0)          # the same clean without a graph keeps the same surfels
    o2.upload(m0); o2.set_pose(pose, tick); o2.predict_indices(pose, tick); o2.clean(pose, tick)
    m2 = o2.download()
    assert m1["pc"].shape == m2["pc"].shape
    moved = m2["tm"][:, 0] != tick
    assert moved.sum() > 0.9 * len(moved)
    assert np.abs(m1["pc"][moved, :3] - (m2["pc"][moved, :3] + tvec)).max() < 2e-6
    assert np.abs(m1["nr"][moved, :3] - m2["nr"][moved, :3]).max() < 2e-6
    assert np.array_equal(m1["pc"][~moved], m2["pc"][~moved]) and np.array_equal(m1["pc"][:, 3], m2["pc"][:, 3]) and np.array_equal(m1["nr"][:, 3], m2["nr"][:, 3])
    assert np.array_equal(m1["tm"], m2["tm"])                             # is_fern: no time-stamp refresh
    # the graph lives for one clean
    o.predict_indices(pose, tick); o.clean(pose, tick)
    assert np.array_equal(o.download()["pc"], m1["pc"])
    # a general graph (small random rotations / translations): the C restatement against an independent numpy one, surfel by surfel
    rng = np.random.RandomState(4)
    g2 = g.copy()
    for i in range(len(g2)):
        a = rng.uniform(-0.05, 0.05, 3)
        Rx = np.array([[1, 0, 0], [0, math.cos(a[0]), -math.sin(a[0])], [0, math.sin(a[0]), math.cos(a[0])]])
        Ry = np.array([[math.cos(a[1]), 0, math.sin(a[1])], [0, 1, 0], [-math.sin(a[1]), 0, math.cos(a[1])]])
        Rz = np.array([[math.cos(a[2]), -math.sin(a[2]), 0], [math.sin(a[2]), math.cos(a[2]), 0], [0, 0, 1]])
        g2[i, 3:12] = (Rz @ Ry @ Rx).T.reshape(9)                        # column-major
        g2[i, 12:15] = rng.uniform(-0.02, 0.02, 3)
    o2.upload(m0); o2.set_pose(pose, tick); o2.predict_indices(pose, tick); o2.set_deformation(g2, is_fern=True); o2.clean(pose, tick)
    m3 = o2.download()
    o2.upload(m0); o2.set_pose(pose, tick); o2.predict_indices(pose, tick); o2.clean(pose, tick)
    m4 = o2.download()                                                     # same survivors, undeformed
    for i in rng.choice(np.flatnonzero(m4["tm"][:, 0] != tick), 40, replace=False):
        newp, newn = _deform_numpy(g2, m4["pc"][i, :3], m4["nr"][i, :3], m4["tm"][i, 0])
        assert np.abs(m3["pc"][i, :3] - newp).max() < 5e-6 and np.abs(m3["nr"][i, :3] - newn).max() < 5e-6, i
    o.close(); o2.close()


def test_render_project_map_properties(orc, small_stream):
    """renderProjectFrameKernel: black where the id image is empty, the surfel's instance colour elsewhere."""
    from instancefusion_amd import synth

    st = small_stream
    o = orc.Oracle(**SMALL, max_surfels=400000)
    for i in range(4):
        po = o.process_frame(st["rgb"][i], st["depth"][i])
    m = o.download(); m["pc"][:, 3] = 20.0
    o.upload(m); o.set_pose(po, o.tick)
    o.process_frame(st["rgb"][3], st["depth"][3], in_pose=po)
    masks, cls = synth.canned_masks(st["obj"][3], st["scene"])
    o.process_segmentation(st["rgb"][3], st["depth"][3], masks, cls, 3)
    ids, pm, mm = o.image("ids_after"), o.render_project_map(), o.download()
    assert (pm[..., 3] == 1).all() and (pm[ids <= 0][:, :3] == 0).all()
    vis = ids > 0
    c = mm["col"][ids[vis], 1].astype(np.int64)
    assert np.array_equal(pm[vis][:, 0], ((c >> 16) & 255).astype(np.float32) / np.float32(255)) and np.array_equal(pm[vis][:, 2], (c & 255).astype(np.float32) / np.float32(255))
    assert (pm[vis][:, :3].sum(axis=1) > 0).any()
    o.close()


def test_oracle_does_not_depend_on_thread_count(orc, small_stream):
    """The oracle's OpenMP loops keep a fixed summation layout (one f64 partial per image row, rows added in order): poses and maps
    are bit-identical for any number of threads, also oversubscribed ones whose chunk borders fall inside rows."""
    st = small_stream
    out = {}
    for th in (1, 3, 32):
        orc.set_threads(th)
        o = orc.Oracle(**SMALL, max_surfels=400000)
        poses = np.stack([o.process_frame(st["rgb"][i], st["depth"][i]).copy() for i in range(4)])
        out[th] = (poses, o.download())
        o.close()
    orc.set_threads(1)
    for th in (3, 32):
        assert np.array_equal(out[th][0], out[1][0]), th
        assert all(np.array_equal(out[th][1][k], out[1][1][k]) for k in out[1][1]), th


def test_rounding_level_perturbation_grows(orc):
    """Why the tracker's sums are exact: the oracle against ITSELF with every per-row partial sum rounded to f32
    once (6e-8 relative, what separates two f32 summation orders) -- identical for a few frames, then the trajectories part by more
    than the north star's 1e-4 m within ~15 frames of the 640x480 benchmark stream.  Two implementations that do not share their sums
    bit for bit cannot stay within 1e-4 m of each other over a sequence; with the exact sums the HIP path and the oracle do
    (tests/test_gpu_parity.py::test_full_loop_640x480_trajectory_and_labels)."""
    from instancefusion_amd import synth

    W, H, NF = 640, 480, 24
    K = dict(fx=528.0, fy=528.0, cx=320.0, cy=240.0)
    st = synth.make_stream(NF, W, H, noise=True, loop_len=90, **K)
    orc.set_threads(orc.usable_cores())
    L = orc.lib()
    runs = []
    for mode in (0, 2, 1):
        L.orc_set_sum_order(mode)
        o = orc.Oracle(w=W, h=H, max_surfels=2_000_000, **K)
        runs.append(np.stack([o.process_frame(st["rgb"][i], st["depth"][i]).copy() for i in range(NF)]))
        o.close()
    L.orc_set_sum_order(0)
    orc.set_threads(1)
    gap = np.linalg.norm(runs[0][:, :3, 3] - runs[1][:, :3, 3], axis=1)
    assert gap[:3].max() < 1e-7                      # the perturbation itself is tiny ...
    assert gap.max() > 1e-4                          # ... and grows past the parity tolerance within the run
    assert np.array_equal(runs[0], runs[2])          # a different ORDER of the exact sums (rows bottom-up) changes nothing at all


def test_reference_shaped_f32_tree_gap(orc, gputest_pair):
    """How far is the REFERENCE's arithmetic from the exact sums every parity claim rests on?  `orc_set_sum_order(3)` makes the oracle sum the 29 / 11
    products of the normal equations as the reference's kernels do -- f32 products, f32 additions, thread-strided partial sums, shuffle tree, one
    partial per warp, second launch over the block partials, with the launch table's GTX 1080 row (EF/Cuda/reduce.cu:133-185, :397-402;
    EF/Utils/GPUConfig.h:123-126).  Asserted here: (1) stage level, on the reference's own RGB-D pair: the f32-tree sums agree with the exact sums to
    1e-5 of their scale ("29-float sums rel. err <= 1e-5"), so mode 3 is the same quantity; (2) trajectory level, 24 frames of the 640x480
    benchmark stream: identical at first, apart by more than the north star's 1e-4 m within the run, and bounded (< 5 mm).  The committed 90-frame
    record (tools/reference_tree_gap.py -> profiles/r06_reference_tree_gap.json; round 3's record, before the window and point rules were pinned to the reference's shaders: r03_reference_tree_gap.json,
    1.4 mm RMS): 0.92 mm RMS, 2.8 mm max, first frame over 1e-4 m: frame 9, while both runs are equally far from the ground truth (35.1 vs 34.9 mm) -- "within 1e-4 m RMS of the reference" is not a property any second implementation of
    this algorithm can have over a sequence, the reference on another GPU model included; what CAN be asserted is bit-identity with a fixed arithmetic,
    which is what tests/test_gpu_parity.py does."""
    import json
    import os

    from gputest_protocol import HALF_K, protocol_inputs
    from instancefusion_amd import synth

    L = orc.lib()
    # (1) stage level
    h, w = gputest_pair[1].shape
    V, N, rgba, prev, depth_mm, rgb = protocol_inputs(*gputest_pair)
    sums = {}
    for mode in (0, 3):
        L.orc_set_sum_order(mode)
        t = L.orc_tracker_create(w, h, HALF_K["fx"], HALF_K["fy"], HALF_K["cx"], HALF_K["cy"])
        L.orc_tracker_init_first_rgb(t, orc.ptr(prev))
        p0 = np.eye(4, dtype=np.float32).reshape(16).copy()
        L.orc_tracker_init_model(t, orc.ptr(V), orc.ptr(N), orc.ptr(rgba), orc.ptr(p0))
        L.orc_tracker_init_frame(t, orc.ptr(depth_mm), orc.ptr(rgb), 20.0)
        pose = p0.copy()
        L.orc_tracker_run(t, orc.ptr(pose), 10.0, 1, 0, 1, None)
        sums[mode] = pose.reshape(4, 4).copy()
        L.orc_tracker_destroy(t)
    L.orc_set_sum_order(0)
    pair_gap = float(np.linalg.norm(sums[0][:3, 3] - sums[3][:3, 3]))
    assert pair_gap < 2e-4, pair_gap                 # one frame of real sensor data: the two arithmetics agree to a fraction of a millimetre
    # (2) trajectory level
    W, H, NF = 640, 480, 24
    K = dict(fx=528.0, fy=528.0, cx=320.0, cy=240.0)
    st = synth.make_stream(NF, W, H, noise=True, loop_len=90, **K)
    orc.set_threads(orc.usable_cores())
    runs = {}
    for mode in (0, 3):
        L.orc_set_sum_order(mode)
        o = orc.Oracle(w=W, h=H, max_surfels=2_000_000, **K)
        runs[mode] = np.stack([o.process_frame(st["rgb"][i], st["depth"][i]).copy() for i in range(NF)])
        o.close()
    L.orc_set_sum_order(0)
    orc.set_threads(1)
    gap = np.linalg.norm(runs[0][:, :3, 3] - runs[3][:, :3, 3], axis=1)
    assert gap[:4].max() < 1e-5                      # the same algorithm: the first frames agree to micrometres
    assert gap.max() > 1e-4                          # ... and part by more than the north star's tolerance within 24 frames
    assert gap.max() < 5e-3                          # bounded: both stay on the trajectory (the gap is tracking noise, not divergence)
    rec = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r06_reference_tree_gap.json")))
    assert np.allclose(gap, rec["gap_m"][:NF], rtol=0, atol=1e-12) or abs(np.sqrt(np.mean(gap ** 2)) - np.sqrt(np.mean(np.square(rec["gap_m"][:NF])))) < 1e-4   # the committed record is this computation
    print(f"reference-shaped f32 tree vs exact sums: pair {pair_gap:.2e} m; 24 frames: max {gap.max():.2e} m, rms {np.sqrt(np.mean(gap ** 2)):.2e} m; committed 90-frame record rms {rec['gap_rms_m']:.2e} max {rec['gap_max_m']:.2e}")
