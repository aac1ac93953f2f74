#!/usr/bin/env python3
"""Generates the fixtures under tests/golden/ (run in the development container only).

1. gputest_pair.npz -- the real RGB-D pair the reference ships for its only executable check of
   the tracker (elasticfusionpublic/GPUTest/{1c,1d,2c,2d}.png; depth in 1/5000 m,
   GPUTest.cpp:55,94), sub-sampled 2x (nearest) to keep the fixture small.  Data only.
2. oracle_pins.npz -- outputs of the CPU oracle on that pair (regression pins; the reference
   publishes no expected values for it -- GPUTest asserts nothing).
3. slic_ref.npz -- superpixel labels produced by the REFERENCE's own gSLICr per-pixel functions
   (oracle/_ref/libref_slic.so, built by `make -C oracle ref` from /root/reference/src/gSLICr with
   -DCOMPILE_WITHOUT_CUDA; driver loops in oracle/ref_slic.cpp) on both colour images of the pair and
   on one synthetic 640x480 frame, plus the oracle's merged-region ids for them (regression pins of
   mergeSuperPixel, for which the reference holds no expected values).
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
    slic_fixtures(d)


def slic_fixtures(d):
    import ctypes as C

    import oracle_lib as ol
    from instancefusion_amd import synth

    ol.build()
    ref = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_slic.so"))
    ref.ref_slic_segment.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p]
    st = synth.make_stream(2, 640, 480, 528.0, 528.0, 320.0, 240.0, noise=True)
    cases = {"c1": (d["1c"], d["1d"] // 5), "c2": (d["2c"], d["2d"] // 5), "synth": (st["rgb"][1], st["depth"][1])}
    out = {}
    for name, (rgb, dep) in cases.items():
        rgb = np.ascontiguousarray(rgb, np.uint8)
        h, w = rgb.shape[:2]
        seg = np.zeros((h, w), np.int32)
        n = ref.ref_slic_segment(rgb.ctypes.data, w, h, 16, 0.6, 5, seg.ctypes.data)
        assert n == (w // 16) * (h // 16) and seg.max() < 32768
        out[name + "_ref_labels"] = seg.astype(np.int16)
        o = ol.Oracle(w=w, h=h, fx=528.0 * w / 640, fy=528.0 * w / 640, cx=w / 2.0, cy=h / 2.0, max_surfels=1000)
        seg2, fin, info = o.merge_superpixels(dep, seg)
        out[name + "_merge_seg"] = seg2.astype(np.int16)
        out[name + "_merge_final"] = fin.astype(np.int16)
        o.close()
    np.savez_compressed(os.path.join(OUT, "slic_ref.npz"), **out)
    print({k: (v.shape, v.dtype) for k, v in out.items()})
    knn_fixture()


def knn_fixture():
    """4. knn_ref.npz -- the 10 nearest neighbours of every point of a seeded surfel cloud according to the REFERENCE's vendored FLANN 1.8.4
    (oracle/_ref/libref_knn.so: KDTreeSingleIndex, exact search, leaf_max_size 64 as at IF/Core/InstanceFusion.cpp:1085-1087)."""
    import ctypes as C

    from instancefusion_amd import synth

    n = 6000
    st = synth.make_stream(1, 320, 240, 264.0, 264.0, 160.0, 120.0, noise=False)
    pos = np.ascontiguousarray(synth.make_map(n, st["scene"], st["poses_world"][0], 10, seed=77)["pc"][:, :3], np.float32)
    ref = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_knn.so"))
    ref.ref_knn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    idx, dist = np.zeros((n, 10), np.int32), np.zeros((n, 10), np.float32)
    assert ref.ref_knn(pos.ctypes.data, n, 10, 64, idx.ctypes.data, dist.ctypes.data) == 0
    np.savez_compressed(os.path.join(OUT, "knn_ref.npz"), pos=pos, idx=idx.astype(np.int16), dist=dist)
    print("knn_ref", pos.shape, idx.shape)


if __name__ == "__main__":
    main()
