#!/usr/bin/env python3
"""Timeline of ONE frame's critical path out of a rocprofv3 kernel trace (`*_kernel_trace.csv`): the dispatches of the main stream between two consecutive k_model_l0
(the first launch of a frame's tracker), with start offset, duration and the idle gap in front of each; plus the per-kernel sums (busy time, gaps) averaged over all
complete frames of the trace.

    python tools/frame_timeline.py TRACE.csv [frame index, default: the middle one] > profiles/<tag>_frame_timeline.txt
"""
import csv
import sys
from collections import defaultdict


def short(n):
    n = n.split("(")[0].replace("void ", "").strip()
    if n.replace(" ", "") in ("k_raster_view<false,true>",):
        return "k_clean_raster_view"
    return n.split("<")[0]


def main():
    rows = []
    with open(sys.argv[1]) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Stream_Id", r.get("Queue_Id", ""))))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if r[2] == "k_model_l0"]
    if len(marks) < 3:
        sys.exit("fewer than three frames in the trace")
    stream = rows[marks[0]][3]
    main_rows = [r for r in rows if r[3] == stream]
    idx = [i for i, r in enumerate(main_rows) if r[2] == "k_model_l0"]
    want = int(sys.argv[2]) if len(sys.argv) > 2 else len(idx) // 2
    # averages over all complete frames
    busy, gaps, cnt = defaultdict(float), defaultdict(float), defaultdict(int)
    gap_list, dur_list = defaultdict(list), defaultdict(list)
    spans = []
    for a, b in zip(idx[:-1], idx[1:]):
        span = main_rows[b][0] - main_rows[a][0]
        if span > 3_000_000:   # (a pause of the host: not a frame period)
            continue
        spans.append(span)
        for j in range(a, b):
            s, e, n, _ = main_rows[j]
            busy[n] += e - s
            cnt[n] += 1
            gaps[n] += max(0, s - main_rows[j - 1][1]) if j > 0 else 0
            gap_list[n].append(max(0, s - main_rows[j - 1][1]) if j > 0 else 0)
            dur_list[n].append(e - s)
    nfr = len(spans)
    print(f"{nfr} frame periods (k_model_l0 to k_model_l0 on the main stream): mean {1e-3 * sum(spans) / nfr:.1f} us, min {1e-3 * min(spans):.1f}, max {1e-3 * max(spans):.1f}")
    med = lambda v: sorted(v)[len(v) // 2]
    print(f"{'kernel':28s} {'per frame':>9s} {'busy us':>9s} {'median dur':>10s} {'median gap':>10s} {'mean gap':>9s}   (per frame: launches x ...)")
    tot_b = tot_g = 0.0
    for n in sorted(busy, key=lambda k_: -busy[k_]):
        if cnt[n] / nfr < 0.5:
            continue
        print(f"{n:28s} {cnt[n] / nfr:9.2f} {1e-3 * busy[n] / nfr:9.2f} {1e-3 * med(dur_list[n]):10.2f} {1e-3 * med(gap_list[n]):10.2f} {1e-3 * gaps[n] / cnt[n]:9.2f}")
        tot_b += busy[n] / nfr
        tot_g += med(gap_list[n]) * cnt[n] / nfr
    print(f"{'total':28s} {'':9s} {1e-3 * tot_b:9.2f} {'':10s} {1e-3 * tot_g:10.2f}  (sum of launches x median gap)")
    spans.sort()
    print(f"median frame period {1e-3 * spans[len(spans) // 2]:.1f} us")
    a, b = idx[want], idx[want + 1]
    t0 = main_rows[a][0]
    print(f"\nframe {want}: {b - a} dispatches, {1e-3 * (main_rows[b][0] - t0):.1f} us")
    print(f"{'offset us':>10} {'dur us':>8} {'gap us':>8}  kernel")
    for j in range(a, b):
        s, e, n, _ = main_rows[j]
        print(f"{1e-3 * (s - t0):10.1f} {1e-3 * (e - s):8.1f} {1e-3 * max(0, s - main_rows[j - 1][1]):8.1f}  {n}")


if __name__ == "__main__":
    main()
