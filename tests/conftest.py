import os
import sys

import numpy as np
import pytest
import torch  # noqa: F401  (first, so that libifx.so binds to the HIP runtime torch ships)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # no test may hang a GPU box: every test gets a hard limit (pytest-timeout), the RCCL one a short one
    for it in items:
        if it.get_closest_marker("timeout") is None:
            it.add_marker(pytest.mark.timeout(180 if "rccl" in it.name else 600))


POSE_STATS = {"poses": 0, "not_bit_equal": 0, "worst_ulp": 0}


def assert_pose_equal(pg, po, what=""):
    """HIP pose against the oracle's.  The claim is bit-identity: the sums of the normal equations are exact on
    both sides, and what is left -- the f64 rounding of the 6x6 solve (unpivoted LDL^T on the device, pivoted in the oracle) and of
    sin / cos of the increment -- survives the cast to f32 with probability ~1e-8 per value.  Asserted: every entry equal, or one f32
    ulp apart (entries below 2^-20 in magnitude: 2^-43 absolute, an ulp of the entries they are sums of).  The number of poses that
    were not bit-equal is counted in POSE_STATS and printed at the end of the session."""
    a = np.ascontiguousarray(pg, np.float32).reshape(-1)
    b = np.ascontiguousarray(po, np.float32).reshape(-1)
    POSE_STATS["poses"] += 1
    if np.array_equal(a, b):
        return
    POSE_STATS["not_bit_equal"] += 1
    ia = a.view(np.int32).astype(np.int64); ib = b.view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, -(ia & 0x7FFFFFFF), ia); ib = np.where(ib < 0, -(ib & 0x7FFFFFFF), ib)
    ulp = np.abs(ia - ib)
    small = (np.abs(a) < 2.0 ** -20) & (np.abs(b) < 2.0 ** -20) & (np.abs(a.astype(np.float64) - b) <= 2.0 ** -43)
    worst = int(np.where(small, 0, ulp).max())
    POSE_STATS["worst_ulp"] = max(POSE_STATS["worst_ulp"], worst)
    assert worst <= 1, f"pose differs from the oracle's by {worst} ulp {what}: {np.abs(a - b).max():.3e}"


def pytest_terminal_summary(terminalreporter):
    if POSE_STATS["poses"]:
        terminalreporter.write_line(f"pose parity: {POSE_STATS['poses']} poses compared with the oracle, {POSE_STATS['not_bit_equal']} not bit-equal "
                                    f"(worst {POSE_STATS['worst_ulp']} ulp)")


SMALL = dict(w=320, h=240, fx=264.0, fy=264.0, cx=160.0, cy=120.0)


@pytest.fixture(scope="session")
def small_stream():
    from instancefusion_amd import synth

    return synth.make_stream(10, SMALL["w"], SMALL["h"], SMALL["fx"], SMALL["fy"], SMALL["cx"], SMALL["cy"], noise=True)


@pytest.fixture(scope="session")
def gputest_pair():
    d = np.load(os.path.join(ROOT, "tests", "golden", "gputest_pair.npz"))
    return d["c1"], d["d1"], d["c2"], d["d2"]


@pytest.fixture(scope="session")
def oracle_pins():
    return dict(np.load(os.path.join(ROOT, "tests", "golden", "oracle_pins.npz")))


@pytest.fixture(scope="session")
def orc():
    import oracle_lib as ol

    ol.build()
    return ol
