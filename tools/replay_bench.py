#!/usr/bin/env python3
"""End-to-end rate of the C++ host (ifx_replay: log decoding, frames, fern data base, loop-closure callbacks, instance stage without masks) on a synthetic
640x480 .klg that goes forth and back.  Usage: python tools/replay_bench.py [--frames N] [--out DIR]"""
import argparse
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=240)
    ap.add_argument("--unique", type=int, default=40)
    ap.add_argument("--out", default="/tmp/replay_bench")
    ap.add_argument("--res", default="640x480")
    a = ap.parse_args()
    from instancefusion_amd import logio, synth

    os.makedirs(a.out, exist_ok=True)
    W, H = (int(v) for v in a.res.split("x"))
    K = (528.0 * W / 640, 528.0 * H / 480, W / 2.0, H / 2.0)
    t0 = time.time()
    st = synth.make_stream(a.unique, W, H, *K, noise=True)
    klg = os.path.join(a.out, "s.klg")
    wr = logio.RawLogWriter(klg, depth="zlib", image="jpeg", jpeg_quality=90)
    for i in range(a.frames + 1):
        j = i % (2 * a.unique - 2)
        j = j if j < a.unique else 2 * a.unique - 2 - j
        wr.add(33333 * i, st["rgb"][j], st["depth"][j])
    wr.close()
    print(f"log: {a.frames} frames, {os.path.getsize(klg) / 1e6:.1f} MB, generated in {time.time() - t0:.1f} s", flush=True)
    exe = os.path.join(ROOT, "instancefusion_amd", "ifx_replay")
    common = [exe, klg, "--width", str(W), "--height", str(H), "--fx", str(K[0]), "--fy", str(K[1]), "--cx", str(K[2]), "--cy", str(K[3]), "--max-surfels", "3000000"]
    for name, extra in (("closeLoops (reference default): detection + fern data base + optimiser", []), ("--detect-only", ["--detect-only"]),
                        ("--no-close-loops", ["--no-close-loops"]), ("closeLoops, --decode-threads 0 (records decoded when asked for)", ["--decode-threads", "0"]),
                        ("closeLoops, --decode-threads 8", ["--decode-threads", "8"]), ("--no-close-loops --decode-threads 8", ["--no-close-loops", "--decode-threads", "8"]),
                        ("--no-close-loops --no-lookahead (frames handed over one at a time)", ["--no-close-loops", "--no-lookahead"]),
                        ("closeLoops, --no-lookahead", ["--no-lookahead"])):
        r = subprocess.run(common + ["--out", os.path.join(a.out, "m")] + extra, capture_output=True, text=True)
        print(name, "->", (r.stdout.strip().splitlines() or [r.stderr.strip()])[-1], flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
