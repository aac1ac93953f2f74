#!/usr/bin/env python3
"""Turns rocprofv3 output into the summaries kept under profiles/.

    python tools/pmc_summary.py stats  RUN_kernel_stats.csv            OUT.csv     # kernel names shortened (no argument lists)
    python tools/pmc_summary.py pmc    FETCH_counter_collection.csv WRITE_counter_collection.csv OUT.json "about text" [bench.json [calibration.json]]
    python tools/pmc_summary.py calib  FETCH_counter_collection.csv expected.json OUT.json                        # tools/pmc_calib.sh
    python tools/pmc_summary.py counters OUT.json PASS1_counter_collection.csv [PASS2 ...]                        # tools/pmc_bound.sh: per-kernel averages of any counters

PMC: FETCH_SIZE and WRITE_SIZE are collected in separate passes (`rocprofv3 --pmc FETCH_SIZE --kernel-trace ...`); values are KB per
launch.  bytes_read = FETCH_SIZE * 1024 * factor, bytes_written = WRITE_SIZE * 1024.  The factor: MI355X_MICROARCH.md (HBM section)
gives 2 for wide (16 B per lane) coalesced streaming reads on gfx950 and calls every other access width uncalibrated; tools/micro/pmc_calib
measures it for this path's patterns (8-B and 4-B streams, the 16-B + 8-B store scan, sparse and random 16-B gathers) and every kernel
entry records the pattern and the factor that was applied to it.
"""
import csv
import json
import sys
from collections import defaultdict


def short(name, keep_template=False):
    """k_name (template arguments dropped: the variants of a kernel are one entry; `keep_template` keeps them, without commas, for the stats table)."""
    n = name.split("(")[0].replace("void ", "").strip()
    if n.replace(" ", "") in ("k_raster_view<false,true>", "k_raster_view<true,true>") and not keep_template:
        return "k_clean_raster_view"   # the fused clean + raster walk of the frame path is timed (and priced) under its own name (bench.py: clean_raster_view)
    if "<" in n:
        n = n.replace(", ", " ") if keep_template else n.split("<")[0]
    return n


def stats(src, dst):
    with open(src) as f, open(dst, "w") as g:
        r = csv.reader(f)
        head = next(r)
        g.write(",".join(head) + "\n")
        for row in r:
            row[0] = short(row[0], keep_template=True)
            g.write(",".join(row) + "\n")


def pmc_table(path, counter):
    acc = defaultdict(lambda: [0, 0.0])
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            k = short(row["Kernel_Name"])
            acc[k][0] += 1
            acc[k][1] += float(row["Counter_Value"])
    return {k: (n, s / n) for k, (n, s) in acc.items()}


# dominant read pattern of a kernel -> which calibrated factor its FETCH_SIZE is multiplied with
PATTERN = {
    # one scan of the store: float4 + float2 of every slot
    "k_cull_frame": "stream24", "k_cull_raster": "stream24", "k_cull_clean": "stream24", "k_index_project": "stream24", "k_alive_flags": "stream8",
    # wide streams
    "k_count_colour": "stream16", "k_compact_scatter": "stream16", "k_max_count": "stream16", "k_clean_table": "stream16", "k_knn_search": "stream16",
    # list-driven passes: a 4-B list entry, then 16-B / 8-B records gathered from the store at ~9 % density
    "k_raster_view": "gather16", "k_clean_raster_view": "gather16", "k_clean_view": "gather16", "k_index_list": "gather16", "k_raster_list": "gather16", "k_clean_list": "gather16", "k_tile_raster": "gather16",
    # per-pixel passes that gather 16-B records through an id / key image
    "k_index_resolve": "gather16r", "k_splat_resolve": "gather16r", "k_fuse_update": "gather16r", "k_project_bbox": "gather16r", "k_count_colour_px": "gather16r",
    "k_associate": "gather16r", "k_vote_update": "gather16r", "k_project_depth": "gather16r",
    # tracker: planar f32 maps, one dword per lane
    "k_icp_residual": "stream4", "k_rgb_step_solve": "stream4", "k_so3_fused": "stream4", "k_model_l0": "stream16", "k_model_down": "stream4", "k_frame_maps": "stream4",
    "k_frame_down": "stream4", "k_bilateral_metric": "stream4", "k_intensity": "stream4",
}


def calib(fetch_csv, expected_json, dst):
    """factor per pattern = known bytes / (FETCH_SIZE * 1024).  For the gathers the known byte count is the one a line-granular fetch moves
    (distinct 64-B lines + the list itself): the factor then says what ONE counted request stands for, and traffic = FETCH_SIZE * factor is in
    real bytes for a gather kernel too."""
    fe = pmc_table(fetch_csv, "FETCH_SIZE")
    with open(expected_json) as f:
        exp = json.load(f)
    out = {"_about": "tools/micro/pmc_calib under rocprofv3 --pmc FETCH_SIZE --kernel-trace: bytes each kernel must read (buffers beyond the Infinity Cache) against the counter.  "
                     "factor = bytes / (FETCH_SIZE x 1024).  Gathers: `factor_lines64` assumes 64-B lines are fetched, `factor_lines128` 128-B ones; the one nearer the streams' "
                     "factor is the consistent reading (a request is tallied at 64 B whatever its size).", "patterns": {}}
    for k, e in exp.items():
        if k not in fe:
            continue
        kb = fe[k][1]
        name = k.replace("k_cal_", "")
        if isinstance(e, dict):
            out["patterns"][name] = dict(FETCH_SIZE_KB=round(kb, 1), entries=e["entries"], algorithmic_bytes=e["algorithmic"], factor_algorithmic=round(e["algorithmic"] / (kb * 1024), 3),
                                         factor_lines64=round(e["lines64"] / (kb * 1024), 3), factor_lines128=round(e["lines128"] / (kb * 1024), 3),
                                         counted_bytes_per_entry=round(kb * 1024 / e["entries"], 1))
        else:
            out["patterns"][name] = dict(FETCH_SIZE_KB=round(kb, 1), bytes=e, factor=round(e / (kb * 1024), 3))
    with open(dst, "w") as g:
        json.dump(out, g, indent=1)


def factors_from(calibration_json):
    """pattern -> (factor, source).  Streams: the measured factor.  Gathers: FETCH_SIZE counts REQUESTS x 64 B; the factor that turns it into bytes moved is the
    streams' own (a request that the stream calibration shows to be tallied at half its size is tallied so for a gather too) unless the calibration says otherwise."""
    fac = {"stream16": (2.0, "guide"), "stream24": (2.0, "guide (uncalibrated pattern)"), "stream8": (2.0, "guide (uncalibrated pattern)"), "stream4": (2.0, "guide (uncalibrated pattern)"),
           "gather16": (2.0, "guide (uncalibrated pattern)"), "gather16r": (2.0, "guide (uncalibrated pattern)")}
    if not calibration_json:
        return fac
    try:
        with open(calibration_json) as f:
            pat = json.load(f)["patterns"]
    except (OSError, ValueError, KeyError):
        return fac
    for k, v in pat.items():
        if "factor" in v:
            fac[k] = (v["factor"], calibration_json)
        elif k.startswith("gather"):
            # which line size makes the gather consistent with the streams?  take the candidate nearer to the stream16 factor
            ref = pat.get("stream16", {}).get("factor", 2.0)
            a, b = v["factor_lines64"], v["factor_lines128"]
            fac[k] = ((a if abs(a - ref) <= abs(b - ref) else b), calibration_json)
    return fac


def pmc(fetch_csv, write_csv, dst, about, bench_json=None, calibration_json=None):
    fe, wr = pmc_table(fetch_csv, "FETCH_SIZE"), pmc_table(write_csv, "WRITE_SIZE")
    fac = factors_from(calibration_json)
    out = {"_about": about + ".  Values are KB per launch (averages).  bytes_read = FETCH_SIZE * 1024 * fetch_factor, bytes_written = WRITE_SIZE * 1024.  fetch_factor per kernel by its "
                     "dominant read pattern (`pattern`): MI355X_MICROARCH.md gives 2 for 16-B-per-lane streams on gfx950; the other patterns are calibrated by tools/micro/pmc_calib "
                     "(`factor_source`).", "factors": {k: dict(factor=v[0], source=v[1]) for k, v in fac.items()}, "kernels": {}}
    if bench_json:   # the workload the counters were collected on: bench.py uses the file only for a matching run
        try:
            with open(bench_json) as f:
                b = json.loads(f.read().strip().splitlines()[-1])
            res = b["config"]["workload"].split("surfel map, ")[1].split(" ")[0]
            out["workload"] = {"res": res, "surfel_slots": b["config"]["surfel_slots"], "surfels_live": b["config"]["surfels_live"], "value_under_profiler": b["value"]}
        except (OSError, ValueError, KeyError, IndexError) as e:
            out["workload_error"] = str(e)
    for k in sorted(fe, key=lambda k_: -fe[k_][0] * fe[k_][1]):
        if not k.startswith("k_"):
            continue
        n, f_kb = fe[k]
        w_kb = wr.get(k, (0, 0.0))[1]
        pat = PATTERN.get(k, "stream16")
        out["kernels"][k] = dict(launches=n, FETCH_SIZE_KB=round(f_kb, 1), WRITE_SIZE_KB=round(w_kb, 1), pattern=pat, fetch_factor=fac[pat][0], bytes_read=int(f_kb * 1024 * fac[pat][0]),
                                 bytes_written=int(w_kb * 1024))
    with open(dst, "w") as g:
        json.dump(out, g, indent=1)


def counters(dst, paths):
    """per kernel: launches and the average per launch of every counter found in the given passes (each pass is its own run of the same command)"""
    acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    for path in paths:
        with open(path) as f:
            for row in csv.DictReader(f):
                k = short(row["Kernel_Name"])
                if not k.startswith("k_"):
                    continue
                c = acc[k][row["Counter_Name"]]
                c[0] += 1
                c[1] += float(row["Counter_Value"])
    out = {"_about": "rocprofv3 --pmc (one pass per counter group, --kernel-trace only) over python3 bench.py; values are averages PER LAUNCH, summed over all shader engines / channels "
                     "by the *_sum forms.  SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_* count quad-cycles per wave (MI355X_MICROARCH.md).", "kernels": {}}
    for k in sorted(acc, key=lambda k_: -sum(v[1] for v in acc[k_].values())):
        e = {"launches": max(v[0] for v in acc[k].values())}
        for c, (n, s_) in sorted(acc[k].items()):
            e[c] = round(s_ / n, 1)
        out["kernels"][k] = e
    with open(dst, "w") as g:
        json.dump(out, g, indent=1)


if __name__ == "__main__":
    if sys.argv[1] == "counters":
        counters(sys.argv[2], sys.argv[3:])
    elif sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3])
    elif sys.argv[1] == "pmc":
        pmc(sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5] if len(sys.argv) > 5 else "", sys.argv[6] if len(sys.argv) > 6 else None, sys.argv[7] if len(sys.argv) > 7 else None)
    elif sys.argv[1] == "calib":
        calib(sys.argv[2], sys.argv[3], sys.argv[4])
    else:
        sys.exit(__doc__)
