#!/usr/bin/env python3
"""Turns rocprofv3 output into the summaries kept under profiles/.

    python tools/pmc_summary.py stats  RUN_kernel_stats.csv            OUT.csv     # kernel names shortened (no argument lists)
    python tools/pmc_summary.py pmc    FETCH_counter_collection.csv WRITE_counter_collection.csv OUT.json "about text"

PMC: FETCH_SIZE and WRITE_SIZE are collected in separate passes (`rocprofv3 --pmc FETCH_SIZE --kernel-trace ...`); values are KB per
launch.  bytes_read = FETCH_SIZE * 1024 * 2 (MI355X_MICROARCH.md, HBM section: on gfx950 the counter reports half of the bytes of wide
streaming reads), bytes_written = WRITE_SIZE * 1024.
"""
import csv
import json
import sys
from collections import defaultdict


def short(name, keep_template=False):
    """k_name (template arguments dropped: the variants of a kernel are one entry; `keep_template` keeps them, without commas, for the stats table)."""
    n = name.split("(")[0].replace("void ", "").strip()
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


def pmc(fetch_csv, write_csv, dst, about, bench_json=None):
    fe, wr = pmc_table(fetch_csv, "FETCH_SIZE"), pmc_table(write_csv, "WRITE_SIZE")
    out = {"_about": about + ".  Values are KB per launch (averages).  bytes_read = FETCH_SIZE * 1024 * 2 (MI355X_MICROARCH.md, HBM section: on gfx950 FETCH_SIZE "
                     "reports half of the bytes of wide coalesced streaming reads; the factor is calibrated for 16-B-per-lane streams only -- tools/micro/stream_ceiling "
                     "under the same counters gives the factor for this store's 8-B + 16-B pattern), bytes_written = WRITE_SIZE * 1024.", "kernels": {}}
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
        out["kernels"][k] = dict(launches=n, FETCH_SIZE_KB=round(f_kb, 1), WRITE_SIZE_KB=round(w_kb, 1), bytes_read=int(f_kb * 1024 * 2), bytes_written=int(w_kb * 1024))
    with open(dst, "w") as g:
        json.dump(out, g, indent=1)


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3])
    elif sys.argv[1] == "pmc":
        pmc(sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5] if len(sys.argv) > 5 else "", sys.argv[6] if len(sys.argv) > 6 else None)
    else:
        sys.exit(__doc__)
