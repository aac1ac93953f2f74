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
