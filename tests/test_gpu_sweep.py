"""Parity SWEEP: the HIP path against the CPU oracle over many scene seeds, camera motions and degraded inputs ("What's weak" 2-3).

Everything else in tests/test_gpu_parity.py runs on ONE seeded scene and ONE smooth camera loop (<= 1 cm / 0.5 degrees per frame): the view lists survive ~4 frames,
the correspondence gates of the tracker (0.10 m / 20 degrees, EF/Utils/RGBDOdometry.h:38-39) are never approached, the depth gates (0.3 m, depthCut) are touched by a
handful of pixels.  Here: 24 cases = scene seed x motion profile (from a camera that does not move to one that leaves the gates: the view lists are rebuilt every
frame, tracking fails -- identically on both sides) x input degradation (depth straddling the 0.3 m / 12 m gates, large holes, a saturated frame, a black frame),
8 frames at 320x240 and one segmentation call with superpixels each.  Compared, per case: every pose (bit-equal, <= 1 ulp counted: conftest.assert_pose_equal), the
surfel-id image of every frame (on a handle that compacts every frame, so that slot numbers are the oracle's indices), the map size, and after the call the instance
table, every label and the whole map.  Each case also runs the HIP sequence TWICE with default options and demands bit-identical poses and maps (run-to-run
determinism: the exact sums do not depend on the order in which blocks and atomics arrive), and reads the run-time guard of that exactness
(ifx_tracker_range_exceeded): 0, except where a case is built to drive it.

Further down: bench.py's resident-frame path at 640x480 on four other scenes / motions, the instance table overflowing (> 96 instances over time) on that path, and the
guard driven on purpose."""
import os

import numpy as np
import pytest

from conftest import assert_pose_equal

pytestmark = pytest.mark.gpu

W, H = 320, 240
K = dict(fx=264.0, fy=264.0, cx=160.0, cy=120.0)
NF = 8
MAP_KEYS = ("pc", "nr", "col", "tm", "ic", "votes")
DEGRADE = ("none", "near", "far", "holes", "saturated", "black")
MOTION = ("still", "slow", "nominal", "fast", "jump", "spin", "dolly", "shake")
# (scene seed, motion profile, degradation): 24 seeds; every profile three times, every degradation four times, no pair twice
CASES = [(11 + 7 * i, MOTION[i % 8], DEGRADE[(i + i // 8) % 6]) for i in range(24)]
# IFX_SWEEP_EXTRA=N: N more scene seeds (other pairs of profile and degradation) for a one-off wider run; its outcome is kept under profiles/
CASES += [(1000 + 13 * i, MOTION[(3 * i + i // 8) % 8], DEGRADE[(5 * i + 1 + i // 6) % 6]) for i in range(int(os.environ.get("IFX_SWEEP_EXTRA", "0")))]
CONF = 3.0   # confidence threshold of the sweep's handles (reference default 10): surfels become stable inside the 8 frames, so that the id images, the clean pass's
             # stable-neighbour rules and the segmentation call have something to work on


@pytest.fixture(scope="module")
def ifx():
    import instancefusion_amd as m

    m.lib()
    return m


def degrade(st, kind, seed):
    """Input degradations, in place on copies: depth in mm (0 = invalid), applied from frame 2 on so that the map starts healthy."""
    rgb, dep = st["rgb"].copy(), st["depth"].copy()
    rng = np.random.RandomState(seed + 5)
    for i in range(2, rgb.shape[0]):
        if kind == "near":        # the left half pulled to 0.26 ... 0.62 m: straddles the 0.3 m gate of depth_metric.frag and the bilateral filter's validity test
            d = dep[i, :, : W // 2].astype(np.float64) * 0.135
            dep[i, :, : W // 2] = np.round(d).astype(np.uint16)
        elif kind == "far":       # a band beyond depthCut (12 m) and a band just inside it; 16 m is beyond max_depth_processed too
            dep[i, 40:90, :] = 12200
            dep[i, 90:130, :] = 11900
            dep[i, 200:, :] = 16000
        elif kind == "holes":     # 45 % of the image without depth, in two rectangles that move
            x0 = int(rng.uniform(0, W // 2)); y0 = int(rng.uniform(0, H // 2))
            dep[i, y0 : y0 + H // 2, x0 : x0 + W // 2] = 0
            dep[i, :, -50:] = 0
        elif kind == "saturated" and i == 4:
            rgb[i] = 255
        elif kind == "black" and i == 4:
            rgb[i] = 0
    return rgb, dep


def pose_eq(a, b):
    return np.array_equal(np.asarray(a, np.float32), np.asarray(b, np.float32), equal_nan=True)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("seed,motion,deg", CASES, ids=[f"s{s}-{m}-{d}" for s, m, d in CASES])
def test_seed_sweep_against_the_oracle(ifx, orc, seed, motion, deg):
    from instancefusion_amd import synth

    scene = synth.Scene(seed)
    st = synth.make_stream_from_poses(synth.trajectory_profile(motion, NF, seed), scene, W, H, noise_seed=seed + 1, **K)
    rgb, dep = degrade(st, deg, seed)
    orc.set_threads(orc.usable_cores())
    o = orc.Oracle(w=W, h=H, max_surfels=400000, confidence=CONF, **K)
    g = ifx.ElasticFusion(w=W, h=H, max_surfels=400000, confidence=CONF, **K)      # every option at its default: lazy compaction, cached view lists, fused clean + raster walk, look-ahead off (host frames)
    g2 = ifx.ElasticFusion(w=W, h=H, max_surfels=400000, confidence=CONF, **K)     # the same sequence a second time: run-to-run determinism
    gc = ifx.ElasticFusion(w=W, h=H, max_surfels=400000, confidence=CONF, **K)     # compacts every frame: slot numbers are the oracle's indices, the id images can be compared
    gc.set_option("compact_every_frame", 1)
    gh = ifx.ElasticFusion(w=W, h=H, max_surfels=400000, confidence=CONF, **K)     # the host entry with every next frame announced (ifx_hint_next_frame): look-ahead, parked tracker
    inst, instc = ifx.InstanceFusion(g), ifx.InstanceFusion(gc)
    finite = True
    for i in range(NF):
        po = o.process_frame(rgb[i], dep[i])
        pg, p2, pc = g.processFrame(rgb[i], dep[i]), g2.processFrame(rgb[i], dep[i]), gc.processFrame(rgb[i], dep[i])
        if i + 1 < NF:
            gh.hint_next_frame(rgb[i + 1], dep[i + 1])
        assert pose_eq(gh.processFrame(rgb[i], dep[i]), pg), f"the announced-frame entry differs at frame {i}"
        finite = finite and bool(np.isfinite(po).all())
        if np.isfinite(po).all():
            assert_pose_equal(pg, po, f"{motion}/{deg} frame {i}")
            assert_pose_equal(pc, po, f"{motion}/{deg} frame {i} (compacting handle)")
        else:
            assert pose_eq(pg, po) and pose_eq(pc, po), (i, pg, po)     # a lost tracker is lost the same way
        assert pose_eq(pg, p2), f"two runs of the same HIP sequence differ at frame {i}"
        assert gc.count == o.count and g.count == o.count, (i, g.count, gc.count, o.count)
        assert np.array_equal(gc.image("ids_after"), o.image("ids_after")), i
    masks, cls = synth.canned_masks(st["obj"][NF - 1], scene, min_area=150)
    sp = (seed // 7) % 2 == 0                                      # every other case with the superpixel refinement of the masks
    if masks.shape[0] > 0:
        inst.ProcessSegmentation(rgb[NF - 1], dep[NF - 1], masks, cls, NF - 1, superpixels=sp)
        instc.ProcessSegmentation(rgb[NF - 1], dep[NF - 1], masks, cls, NF - 1, superpixels=sp)
        o.process_segmentation(rgb[NF - 1], dep[NF - 1], masks, cls, NF - 1, flags=2 if sp else 0)
        assert np.array_equal(inst.getInstanceTable(), o.instance_table())
        assert np.array_equal(instc.getInstanceTable(), o.instance_table())
        assert np.array_equal(instc.labels(), o.labels())
        assert np.array_equal(inst.labels(), o.labels())
    assert gh.lookahead_stats()["host_hinted"] == NF - 1
    mg, m2, mc, mo, mh = g.download(), g2.download(), gc.download(), o.download(), gh.download()
    for k in MAP_KEYS:
        assert np.array_equal(mc[k], mo[k], equal_nan=True), k
        assert np.array_equal(mg[k], mo[k], equal_nan=True), k
        if k != "votes":                                          # (g2 and gh ran no segmentation call)
            assert np.array_equal(mg[k] if k != "col" else mg[k][:, 0], m2[k] if k != "col" else m2[k][:, 0], equal_nan=True), k
            assert np.array_equal(mg[k] if k != "col" else mg[k][:, 0], mh[k] if k != "col" else mh[k][:, 0], equal_nan=True), k
    # the exact-sum range guard: nothing in the sweep may leave the range in which the sums are order-independent (a case that did would not be REQUIRED to agree)
    assert g.tracker_range_exceeded() == 0 and gc.tracker_range_exceeded() == 0, (g.tracker_range_exceeded(), motion, deg)
    for x in (g, g2, gc, gh, o):
        x.close()


@pytest.mark.timeout(1200)
@pytest.mark.parametrize("seed,motion,small", [(301, "nominal", 0), (302, "fast", 0), (303, "jump", 0), (304, "shake", 0), (2238, "shake", 0),
                                               (901, "shake", 1), (902, "slow", 1), (903, "still", 1), (904, "nominal", 1)] +
                         [(2000 + 17 * i, MOTION[(i + 1 + i // 8) % 8], 0) for i in range(int(os.environ.get("IFX_SWEEP_EXTRA_640_FROM", "0")), int(os.environ.get("IFX_SWEEP_EXTRA_640", "0")))] +   # (IFX_SWEEP_EXTRA_640=N: a one-off wider run)
                         [(2500 + 23 * i, MOTION[(3 * i + 2) % 8], 2) for i in range(int(os.environ.get("IFX_SWEEP_EXTRA_1280", "0")))])   # (IFX_SWEEP_EXTRA_1280=N: the same at 1280x960 -- tiled rasteriser, persistent coarsest level)
def test_resident_frame_path_at_640x480_other_scenes(ifx, orc, seed, motion, small):
    """bench.py's own frame path -- frames resident in HBM, the next frame announced, its tracker parked behind every frame, default options (lazy compaction, cached view
    lists, fused clean + raster walk, hot records, the id image on the lattice) -- at the benchmark's resolution on OTHER scenes and camera motions than the one every other
    640x480 test uses: 24 frames and a segmentation call on the resident frame; every pose, the instance table, and at the end count, labels and the whole map against the
    oracle.  (`fast` rebuilds the view lists every frame, `jump` leaves the correspondence gates once.)
    Nothing here reads a count between the frames (that forces a list rebuild): the view lists live as long as the motion lets them.  Seed 2238 `shake` and the `small`
    cases (320x240, 32 frames) are maps OLDER THAN 20 FRAMES under lists that survive several frames: the clean pass's age rule (copy_unstable.vert:160-172) removes
    unstable surfels that no list holds -- a list rebuild must apply the rule they have outlived before it takes them in (2238 found that it did not: 519 surfels too many)."""
    import torch

    from instancefusion_amd import synth

    Wb, Hb, NFb = {0: (640, 480, 24), 1: (320, 240, 32), 2: (1280, 960, 14)}[small]
    Kb = dict(fx=528.0 * Wb / 640, fy=528.0 * Wb / 640, cx=Wb / 2.0, cy=Hb / 2.0)
    scene = synth.Scene(seed)
    st = synth.make_stream_from_poses(synth.trajectory_profile(motion, NFb, seed), scene, Wb, Hb, noise_seed=seed + 1, **Kb)
    orc.set_threads(orc.usable_cores())
    d_rgb = torch.from_numpy(st["rgb"]).cuda()
    d_dep = torch.from_numpy(st["depth"].view(np.int16)).cuda()
    torch.cuda.synchronize()
    cap = 6_000_000 if small == 2 else 2_000_000
    g = ifx.ElasticFusion(w=Wb, h=Hb, max_surfels=cap, confidence=CONF, **Kb)
    o = orc.Oracle(w=Wb, h=Hb, max_surfels=cap, confidence=CONF, **Kb)
    inst = ifx.InstanceFusion(g)
    for i in range(NFb):
        if i + 1 < NFb:
            g.hint_next_frame_device(d_rgb[i + 1].data_ptr(), d_dep[i + 1].data_ptr())
        g.enqueue_frame_device(d_rgb[i].data_ptr(), d_dep[i].data_ptr(), i)
        inst.whetherDoSegmentation(100 + i)
        po = o.process_frame(st["rgb"][i], st["depth"][i])
        pg = g.trajectory(1)[0]
        if np.isfinite(po).all():
            assert_pose_equal(pg, po, f"{motion} frame {i}")
        else:
            assert pose_eq(pg, po), i
        if i == NFb - 1:   # (the labels are the last call's: compared right behind it)
            masks, cls = synth.canned_masks(st["obj"][i], scene)
            if masks.shape[0]:   # (the fast camera has left the objects behind by then: no detections, no call)
                inst.ProcessSegmentation(None, None, masks, cls, i, superpixels=False)
                o.process_segmentation(st["rgb"][i], st["depth"][i], masks, cls, i, flags=0)
                assert np.array_equal(inst.getInstanceTable(), o.instance_table()), i
                assert np.array_equal(inst.labels(), o.labels())
    assert g.count == o.count
    mg, mo = g.download(), o.download()
    for k in MAP_KEYS:
        assert np.array_equal(mg[k], mo[k], equal_nan=True), k
    assert g.tracker_range_exceeded() == 0
    g.close(); o.close()


@pytest.mark.timeout(1800)
@pytest.mark.parametrize("seed,motion", [(3001, "nominal")] + [(3100 + 19 * i, MOTION[(2 * i + 1 + i // 4) % 8]) for i in range(int(os.environ.get("IFX_SWEEP_LONG_FROM", "0")), int(os.environ.get("IFX_SWEEP_LONG", "0")))])
def test_long_run_with_calls_on_the_resident_frame_path(ifx, orc, seed, motion):
    """The same path over a LONGER life of the map: 56 frames at 640x480 (the age rule of the clean pass at work from frame 21 on, view lists that live several frames,
    tombstones piling up towards a compaction) with a segmentation call on the resident frame every 9th frame, with and without superpixels in turn -- the gated label
    scan, the superpixels run ahead of a call, votes on a map that keeps changing.  Every pose and every whetherDoSegmentation decision; after every call the instance table and the labels; at the end the whole
    map.  (IFX_SWEEP_LONG=N: N more scenes / motions, a one-off wider run.)"""
    import torch

    from instancefusion_amd import synth

    Wb, Hb, NFb = 640, 480, 56
    Kb = dict(fx=528.0, fy=528.0, cx=320.0, cy=240.0)
    scene = synth.Scene(seed)
    st = synth.make_stream_from_poses(synth.trajectory_profile(motion, NFb, seed), scene, Wb, Hb, noise_seed=seed + 1, **Kb)
    orc.set_threads(orc.usable_cores())
    d_rgb = torch.from_numpy(st["rgb"]).cuda()
    d_dep = torch.from_numpy(st["depth"].view(np.int16)).cuda()
    torch.cuda.synchronize()
    g = ifx.ElasticFusion(w=Wb, h=Hb, max_surfels=2_000_000, confidence=CONF, **Kb)
    g.set_option("compact_divisor", 24)   # (a compaction inside the run: tombstones reach 1 / 24 of the slots once the age rule works)
    o = orc.Oracle(w=Wb, h=Hb, max_surfels=2_000_000, confidence=CONF, **Kb)
    inst = ifx.InstanceFusion(g)
    calls = 0
    for i in range(NFb):
        if i + 1 < NFb:
            g.hint_next_frame_device(d_rgb[i + 1].data_ptr(), d_dep[i + 1].data_ptr())
        g.enqueue_frame_device(d_rgb[i].data_ptr(), d_dep[i].data_ptr(), i)
        want = inst.whetherDoSegmentation(100 + i)
        po = o.process_frame(st["rgb"][i], st["depth"][i])
        assert want == o.should_segment(100 + i), i   # (the decision: empty lattice pixels of the id image, vote mass under it, the cadence)
        pg = g.trajectory(1)[0]
        if np.isfinite(po).all():
            assert_pose_equal(pg, po, f"{motion} frame {i}")
        else:
            assert pose_eq(pg, po), i
        if i % 9 == 8 or i == NFb - 1:
            masks, cls = synth.canned_masks(st["obj"][i], scene)
            if masks.shape[0]:
                sp = calls % 2 == 1
                inst.ProcessSegmentation(None, None, masks, cls, i, superpixels=sp)
                o.process_segmentation(st["rgb"][i], st["depth"][i], masks, cls, i, flags=2 if sp else 0)
                calls += 1
                assert np.array_equal(inst.getInstanceTable(), o.instance_table()), i
                assert np.array_equal(inst.labels(), o.labels()), i
    assert g.count == o.count
    mg, mo = g.download(), o.download()
    for k in MAP_KEYS:
        assert np.array_equal(mg[k], mo[k], equal_nan=True), k
    assert g.tracker_range_exceeded() == 0
    g.close(); o.close()


@pytest.mark.timeout(900)
def test_instance_table_overflow_on_the_resident_frame_path(ifx, orc):
    """> 96 instances over time under bench.py's own frame path: frames resident in HBM, the next frame announced and its tracker parked behind every frame, the
    segmentation call on the resident frame beside it (ProcessSegmentation(None, None, ...)) -- and every call brings masks of NEW classes, so the 96-row instance
    table fills and evicts (IF/Core/InstanceFusion.cpp:836-905) while frames keep coming.  Tables and labels against the oracle at every call, the whole map at the end.
    (After six frames the map is made stable and its votes cleared on both sides -- a map of first-frame surfels carries the reference's -1 votes and registers nothing;
    without superpixel refinement, which at 320x240 rejects most of the synthetic silhouettes: ~6 of 8 masks register per call, the table is full after 14 calls.)"""
    import torch

    from instancefusion_amd import synth

    NFR = 24
    scene = synth.Scene(4242)
    st = synth.make_stream_from_poses(synth.trajectory_profile("nominal", NFR, 4242), scene, W, H, noise_seed=4243, **K)
    orc.set_threads(orc.usable_cores())
    d_rgb = torch.from_numpy(st["rgb"]).cuda()
    d_dep = torch.from_numpy(st["depth"].view(np.int16)).cuda()
    torch.cuda.synchronize()
    g = ifx.ElasticFusion(w=W, h=H, max_surfels=600000, **K)
    o = orc.Oracle(w=W, h=H, max_surfels=600000, **K)
    inst = ifx.InstanceFusion(g)
    evicted, seen, most = False, 0, 0
    po = None
    for i in range(NFR):
        if i == 6:
            m = o.download(); m["pc"][:, 3] = 20.0; m["votes"][:] = 0
            g.upload(m); o.upload(m)
            g.set_pose(po, o.tick); o.set_pose(po, o.tick)
            pg = g.processFrame(st["rgb"][i], st["depth"][i], inPose=po); po = o.process_frame(st["rgb"][i], st["depth"][i], in_pose=po)
            assert_pose_equal(pg, po, "frame 6 (held pose)")
        else:
            if i + 1 < NFR and i + 1 != 6:
                g.hint_next_frame_device(d_rgb[i + 1].data_ptr(), d_dep[i + 1].data_ptr())
            g.enqueue_frame_device(d_rgb[i].data_ptr(), d_dep[i].data_ptr(), i)
            inst.whetherDoSegmentation(100 + i)
            po = o.process_frame(st["rgb"][i], st["depth"][i])
            assert_pose_equal(g.trajectory(1)[0], po, f"frame {i}")
        if i >= 6:
            masks, _ = synth.canned_masks(st["obj"][i], scene, min_area=150)
            nm = masks.shape[0]
            classes = (1 + (seen + np.arange(nm)) % 79).astype(np.int32)       # a new class for every mask of every call: nothing matches an instance of the table
            seen += nm
            if i == 6:
                inst.ProcessSegmentation(st["rgb"][i], st["depth"][i], masks, classes, i)
            else:
                inst.ProcessSegmentation(None, None, masks, classes, i)       # the resident frame, beside the parked tracker of frame i + 1
            o.process_segmentation(st["rgb"][i], st["depth"][i], masks, classes, i, flags=0)
            tg, to = inst.getInstanceTable(), o.instance_table()
            assert np.array_equal(tg, to), i
            assert np.array_equal(inst.labels(), o.labels()), i
            n_used = int((to >= 0).sum())
            if n_used < most - 4:
                evicted = True
            most = max(most, n_used)
    assert seen > 96 and most > 85 and evicted, (seen, most, evicted)
    mg, mo = g.download(), o.download()
    for k in MAP_KEYS:
        assert np.array_equal(mg[k], mo[k]), k
    assert g.tracker_range_exceeded() == 0
    g.close(); o.close()


def test_exact_sum_range_guard_is_driven(ifx, orc):
    """ifx_tracker_range_exceeded: 0 on ordinary frames (and on every case of the sweep above); raised by a frame pair built to leave the working range of a row entry.
    What can leave it at 640x480: the ICP rows cannot (unit normals, arms below 16 m, residuals gated at 0.10 m: every diagonal sum stays below a quarter of its exact
    range by construction), the photometric rows of the 6-DoF step cannot either (the reference's sigma = sqrt(count) weights shrink them by ~1/500); the SO(3)
    pre-alignment's residual can: a frame that flips from black to white puts 255^2 on every one of the 19 200 pixels of the coarsest level, 1.25e9 against the 2^30
    at which the guard -- set at HALF the exact range -- speaks.  What the count says: the sums of such frames are not GUARANTEED order-independent any more."""
    from instancefusion_amd import synth

    Wg, Hg = 640, 480
    Kg = dict(fx=528.0, fy=528.0, cx=320.0, cy=240.0)
    st = synth.make_stream(3, Wg, Hg, noise=True, **Kg)
    g = ifx.ElasticFusion(w=Wg, h=Hg, max_surfels=1_500_000, **Kg)
    for i in range(3):
        g.processFrame(st["rgb"][i], st["depth"][i])
    assert g.tracker_range_exceeded() == 0
    black = np.zeros((Hg, Wg, 3), np.uint8)
    white = np.full((Hg, Wg, 3), 255, np.uint8)
    for k in range(3):
        g.processFrame(black if k % 2 == 0 else white, st["depth"][2])
    n = g.tracker_range_exceeded()
    assert n > 0, "the guard did not see the SO(3) residual of a black -> white flip"
    print(f"range guard: {n} reductions beyond half the exact range on the black / white flip, 0 on the synthetic stream")
    g.close()


def _junk_behind_the_camera(pose, last_time, conf=1.0):
    """One unstable surfel 30 m BEHIND the camera of `pose`: no view list holds it, no render draws it, nothing matches it -- only the clean pass's age rule ever applies."""
    T = np.asarray(pose, np.float64).reshape(4, 4)
    p = T[:3, 3] - 30.0 * T[:3, 2]
    return dict(pc=np.array([p[0], p[1], p[2], conf], np.float32), nr=np.array([0, 0, 1, 0.01], np.float32), col=np.zeros(2, np.float32),
                tm=np.array([last_time, last_time], np.float32), ic=np.zeros(4, np.float32), votes=np.zeros(48, np.float32))


@pytest.mark.timeout(900)
@pytest.mark.parametrize("motion,late,time_delta,path,peek", [
    ("still", 0, 200, "host", 0),       # the age rule removes the junk surfel in the very frame whose scan leaves it out: it is "surfel 0" for that frame's passes still
    ("still", 2, 200, "host", 0),       # ... two frames later, between two scans: the NEXT live surfel reads id 0 from the following frame on
    ("slow", 2, 200, "resident", 0),    # the same on bench.py's frame path (next frame announced, tracker parked)
    ("nominal", 3, 200, "host", 0),     # lists rebuilt every ~4 frames
    ("slow", 2, 200, "host", 2),        # a count is read after the second frame: a forced scan (ifx_vlist_reap) in between
    ("fast", 1, 200, "host", 0),        # lists rebuilt EVERY frame: the scan of the frame whose clean pass removes the junk surfel must leave it "surfel 0" for that frame
    ("fast", 2, 200, "host", 0),
    ("shake", 2, 200, "host", 0),
    ("still", 6, 30, "host", 0),        # timeDelta 30: by the time a scan looks at the junk surfels they are beyond the time window (exempt) -- the reference removed them at age 21
])
def test_surfel_0_outside_the_view_lists(ifx, orc, motion, late, time_delta, path, peek):
    """The reference's "surfel 0" (id 0 = "no surfel": it occludes, but is never associated, counted or voted for) is the FIRST LIVE surfel of the map.  Here a slot that
    no cached view list holds meets the clean pass's age rule (copy_unstable.vert:160-172) only when a scan comes by -- invisible, unless it is the map's first live slot:
    the map is given an unstable surfel 30 m behind the camera as its first slot, whose age passes 20 in frame `late` after the upload, and the stable surfel drawn nearest to the image
    centre as its second, which becomes "surfel 0" the frame after.  Every pose, and at the end count and the whole map (a surfel that reads id 0 is not updated and its
    pixel creates a new one), against the oracle, which cleans every surfel in every frame like the reference.  (ADVICE round 5 for the
    timeDelta case: a second junk surfel in the middle of the map, neither first nor listed.)"""
    import torch

    from instancefusion_amd import synth

    seed, NW, NR = 777, 24, 20 if time_delta != 200 else 9
    scene = synth.Scene(seed)
    st = synth.make_stream_from_poses(synth.trajectory_profile(motion, NW + NR, seed), scene, W, H, noise_seed=seed + 1, **K)
    orc.set_threads(orc.usable_cores())
    o = orc.Oracle(w=W, h=H, max_surfels=400000, confidence=CONF, time_delta=time_delta, **K)
    g = ifx.ElasticFusion(w=W, h=H, max_surfels=400000, confidence=CONF, time_delta=time_delta, **K)
    po = None
    for i in range(NW):
        po = o.process_frame(st["rgb"][i], st["depth"][i])
        assert_pose_equal(g.processFrame(st["rgb"][i], st["depth"][i]), po, f"warm-up frame {i}")
    m = o.download()
    ids = o.image("ids_after")
    F0 = o.tick                                                   # the time of the first frame after the upload
    par = (F0 + late) % 2                                         # only pixels with x % 2 == y % 2 == time % 2 associate (data.vert:131), mostly with the surfel of their own texel
    yy, xx = np.nonzero((ids > 0) & ((np.arange(H)[:, None] % 2) == par) & ((np.arange(W)[None, :] % 2) == par))
    assert yy.size, "no stable surfel in view"
    near = np.argmin((yy - H // 2) ** 2 + (xx - W // 2) ** 2)
    s = int(ids[yy[near], xx[near]])                              # the stable surfel drawn nearest to the image centre, on a pixel that is active in the frame that removes the junk
    assert m["pc"][s, 3] >= CONF
    order = np.concatenate([[s], np.delete(np.arange(m["pc"].shape[0]), s)])
    junk = _junk_behind_the_camera(po, F0 + late - 21)            # age 21 (> 20: removed) at the end of frame F0 + late
    mid = order.shape[0] // 2
    m2 = {}
    for k in MAP_KEYS:
        a = m[k][order]
        m2[k] = np.concatenate([junk[k][None], a[:mid], junk[k][None], a[mid:]]).astype(np.float32)
    o.upload(m2); g.upload(m2)
    o.set_pose(po, F0); g.set_pose(po, F0)
    if path == "resident":
        d_rgb = torch.from_numpy(st["rgb"]).cuda()
        d_dep = torch.from_numpy(st["depth"].view(np.int16)).cuda()
        torch.cuda.synchronize()
    for i in range(NW, NW + NR):
        first = i == NW
        pin = po if first else None                               # the frame behind the upload holds its pose (its prediction is the old map's)
        po = o.process_frame(st["rgb"][i], st["depth"][i], in_pose=pin)
        if path == "host" or first:
            pg = g.processFrame(st["rgb"][i], st["depth"][i], inPose=pin)
        else:
            if i + 1 < NW + NR:
                g.hint_next_frame_device(d_rgb[i + 1].data_ptr(), d_dep[i + 1].data_ptr())
            g.enqueue_frame_device(d_rgb[i].data_ptr(), d_dep[i].data_ptr(), i)
            pg = g.trajectory(1)[0]
        assert_pose_equal(pg, po, f"frame {i - NW} after the upload")
        if peek and i - NW + 1 == peek:
            assert g.count == o.count, (i, g.count, o.count)
    assert g.count == o.count, (g.count, o.count)
    mg, mo = g.download(), o.download()
    for k in MAP_KEYS:
        assert np.array_equal(mg[k], mo[k], equal_nan=True), k
    g.close(); o.close()
