#!/usr/bin/env python3
"""Generates the fixtures under tests/golden/ (run in the development container only).

1. gputest_pair.npz -- the real RGB-D pair the reference ships for its only executable check of
   the tracker (elasticfusionpublic/GPUTest/{1c,1d,2c,2d}.png; depth in 1/5000 m,
   GPUTest.cpp:55,94), sub-sampled 2x (nearest) to keep the fixture small.  Data only.
2. oracle_pins.npz -- outputs of the CPU oracle on that pair (regression pins; the reference
   publishes no expected values for it -- GPUTest asserts nothing).
"""
import os
import sys

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
REF = "/root/reference/elasticfusionpublic/GPUTest"
OUT = os.path.join(ROOT, "tests", "golden")


def main():
    os.makedirs(OUT, exist_ok=True)
    d = {}
    for k in ("1c", "2c"):
        d[k] = np.asarray(Image.open(os.path.join(REF, k + ".png")).convert("RGB"))[::2, ::2].copy()
    for k in ("1d", "2d"):
        a = np.asarray(Image.open(os.path.join(REF, k + ".png")))
        d[k] = a.astype(np.uint16)[::2, ::2].copy()
    np.savez_compressed(os.path.join(OUT, "gputest_pair.npz"), c1=d["1c"], c2=d["2c"], d1=d["1d"], d2=d["2d"])
    from gputest_protocol import run_oracle_protocol

    pins = run_oracle_protocol(d["1c"], d["1d"], d["2c"], d["2d"])
    np.savez_compressed(os.path.join(OUT, "oracle_pins.npz"), **pins)
    print({k: (v.shape, v.dtype) for k, v in pins.items()})


if __name__ == "__main__":
    main()
