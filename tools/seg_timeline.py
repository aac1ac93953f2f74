#!/usr/bin/env python3
"""Timeline of ONE segmentation call out of a rocprofv3 kernel trace (`*_kernel_trace.csv`): every dispatch between k_mask_clean_overlap and the label scan,
with its start offset, duration and the idle gap in front of it -- where the host sits in the call ("segmentation call").

    python tools/seg_timeline.py TRACE.csv [call index, default: the last one] > profiles/<tag>_seg_call_timeline.txt
"""
import csv
import sys


def short(n):
    n = n.split("(")[0].replace("void ", "").strip()
    return n.split("<")[0]


def main():
    rows = []
    with open(sys.argv[1]) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Stream_Id", r.get("Queue_Id", ""))))
    rows.sort()
    starts = [i for i, r in enumerate(rows) if r[2].startswith("k_mask_clean_overlap")]
    if not starts:
        sys.exit("no segmentation call in the trace")
    want = int(sys.argv[2]) if len(sys.argv) > 2 else len(starts) - 1
    i0 = starts[want]
    end_names = ("k_count_colour_px", "k_count_colour")
    i1 = next((i for i in range(i0, len(rows)) if rows[i][2] in end_names), len(rows) - 1)
    # the call's own stream only (it runs beside the next frame's tracker), and from its first dispatch: the superpixels are enqueued before the masks are staged
    stream = rows[i0][3]
    rows = [r for r in rows[:i1 + 1] if r[3] == stream]
    i1 = len(rows) - 1
    i0 = max(i for i, r in enumerate(rows) if r[2].startswith("k_mask_clean_overlap"))
    while i0 > 0 and rows[i0][0] - rows[i0 - 1][1] < 300_000:
        i0 -= 1
    t0 = rows[i0][0]
    print(f"segmentation call {want} of {len(starts)}: {i1 - i0 + 1} dispatches, {1e-3 * (rows[i1][1] - t0):.1f} us from the first kernel's start to the label scan's end")
    print(f"{'offset us':>10} {'dur us':>8} {'gap us':>8}  kernel (stream)")
    prev_end = {}
    busy = 0
    last_end_any = t0
    idle = 0.0
    for s, e, n, q in rows[i0:i1 + 1]:
        gap = (s - last_end_any) * 1e-3
        if gap > 0:
            idle += gap
        print(f"{(s - t0) * 1e-3:10.1f} {(e - s) * 1e-3:8.1f} {max(gap, 0.0):8.1f}  {n} ({q})")
        last_end_any = max(last_end_any, e)
        busy += e - s
    print(f"sum of kernel durations {busy * 1e-3:.1f} us (kernels of other streams overlap), idle gaps on the device {idle:.1f} us")


if __name__ == "__main__":
    main()
