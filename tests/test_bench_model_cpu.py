"""bench.py's byte model (roofline.achieved = algorithmic bytes / measured time): the per-level bookkeeping of the tracker's launches, on the CPU."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_tracker_byte_model_follows_the_persistent_level_mask():
    bench = importlib.import_module("bench")
    P = 640 * 480
    # no level in the persistent kernel: the 19 two-launch iterations average (10 + 5 / 4 + 4 / 16) / 19 of the image; nothing billed to k_gn_level
    assert abs(bench.level_avg(0) - (10 + 5 / 4 + 4 / 16) / 19) < 1e-12
    assert bench.gn_level_bytes(P, 0) == 0.0
    # the default: the coarsest level's 4 iterations in one launch -- 39 B per pixel once + 41 B per pixel and iteration on P / 16 pixels
    assert abs(bench.level_avg(4) - (10 + 5 / 4) / 15) < 1e-12
    assert abs(bench.gn_level_bytes(P, 4) - P / 16 * (39 + 4 * 41)) < 1e-6
    # every level: nothing left for the two-launch form; the average launch of the kernel over the three levels
    assert bench.level_avg(7) == 0.0
    want = sum(P / 4 ** l * (39 + it * 41) for l, it in enumerate((10, 5, 4))) / 3
    assert abs(bench.gn_level_bytes(P, 7) - want) < 1e-6
    # the table entries use the module's current mask
    old = bench.GN_PERSIST
    try:
        bench.GN_PERSIST = 0
        a0 = bench.algorithmic_bytes("icp_residual", 5_000_000, P)
        bench.GN_PERSIST = 4
        a4 = bench.algorithmic_bytes("icp_residual", 5_000_000, P)
        assert a0 == P * bench.level_avg(0) * 70 and a4 == P * bench.level_avg(4) * 70 and a4 > a0 * 0.9
        assert bench.algorithmic_bytes("gn_level", 5_000_000, P) == bench.gn_level_bytes(P, 4)
    finally:
        bench.GN_PERSIST = old


def test_list_kernels_have_byte_entries():
    bench = importlib.import_module("bench")
    P = 640 * 480
    vl = (420_000, 750_000)
    for k in ("index_list", "clean_view", "raster_view", "fuse_update", "project_bbox", "count_colour_px", "cull_frame"):
        assert bench.algorithmic_bytes(k, 5_000_000, P, vl=vl) > 0, k
    assert bench.algorithmic_bytes("raster_view", 5_000_000, P, vl=vl) == (vl[0] + vl[1]) * 44.0
    assert bench.algorithmic_bytes("no_such_kernel", 5_000_000, P, vl=vl) == 0.0
