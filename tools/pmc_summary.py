#!/usr/bin/env python3
"""Summarise two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of bench.py into profiles/rNN_pmc_<workload>.json.
   tools/pmc_summary.py <fetch counter_collection.csv> <write counter_collection.csv> <frames profiled> <workload> <out.json>
Counter units and the gfx950 FETCH_SIZE caveat: MI355X_MICROARCH.md (HBM / rocprofv3 section)."""
import csv
import json
import re
import sys
from collections import defaultdict


def per_kernel(path, counter):
    tot, n = defaultdict(float), defaultdict(int)
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            name = row["Kernel_Name"]
            m = re.search(r"mpc::(\w+)", name)
            if not m:
                continue
            tot[m.group(1)] += float(row["Counter_Value"])
            n[m.group(1)] += 1
    return tot, n


def main():
    fetch_csv, write_csv, frames, workload, out = sys.argv[1], sys.argv[2], float(sys.argv[3]), sys.argv[4], sys.argv[5]
    ft, fn = per_kernel(fetch_csv, "FETCH_SIZE")
    wt, wn = per_kernel(write_csv, "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(ft) | set(wt)):
        kernels[k] = {"dispatches_per_frame": round(max(fn.get(k, 0), wn.get(k, 0)) / frames, 2),
                      "FETCH_SIZE_KB_per_frame": round(ft.get(k, 0.0) / frames, 1),
                      "WRITE_SIZE_KB_per_frame": round(wt.get(k, 0.0) / frames, 1)}
    fetch_b = sum(ft.values()) / frames * 1024
    write_b = sum(wt.values()) / frames * 1024
    per_launch = {}
    for k in kernels:                                        # what bench.py reads: roofline.traffic of the dominant kernel, per launch
        launches = max(fn.get(k, 0), wn.get(k, 0), 1)
        fb, wb = ft.get(k, 0.0) / launches * 1024, wt.get(k, 0.0) / launches * 1024
        per_launch[k] = {"launches_profiled": launches, "raw_fetch_bytes_per_launch": int(fb), "write_bytes_per_launch": int(wb),
                         "hbm_bytes_per_launch": int(2 * fb + wb)}
    doc = {
        "workload": workload,
        "collected": "two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of bench.py --no-cpu, kernel-trace/stats not combined",
        "units": "counter values are KB (x1024 = bytes); per frame = per step of the profiled command",
        "per_kernel": kernels,
        "raw_fetch_bytes_per_launch": int(fetch_b),
        "write_bytes_per_launch": int(write_b),
        "gfx950_correction": "MI355X_MICROARCH.md: FETCH_SIZE reads exactly 1/2 of the bytes of wide coalesced 16 B/lane streams; other "
                             "widths uncalibrated.  Most reads here are 16 B/lane, so 2 x FETCH + WRITE is reported (an upper bound).",
        "hbm_bytes_per_step_all_kernels": int(2 * fetch_b + write_b),
    }
    doc.update(per_launch)
    with open(out, "w") as f:
        json.dump(doc, f, indent=1)
    print(json.dumps(per_launch))


if __name__ == "__main__":
    main()
