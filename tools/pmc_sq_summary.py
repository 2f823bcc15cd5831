#!/usr/bin/env python3
"""Per-kernel sums of one rocprofv3 --pmc pass (SQ counters): tools/pmc_sq_summary.py <counter_collection.csv> <out.json>"""
import csv, json, re, sys
from collections import defaultdict
tot = defaultdict(lambda: defaultdict(float))
n = defaultdict(int)
with open(sys.argv[1], newline="") as f:
    for row in csv.DictReader(f):
        m = re.search(r"mpc::(\w+)", row["Kernel_Name"])
        if not m:
            continue
        k = m.group(1)
        tot[k][row["Counter_Name"]] += float(row["Counter_Value"])
        n[(k, row["Counter_Name"])] += 1
out = {}
for k, cs in tot.items():
    d = {c: v for c, v in cs.items()}
    d["dispatches"] = max(n[(k, c)] for c in cs)
    wc = d.get("SQ_WAVE_CYCLES", 0.0)
    if wc:
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM"):
            if c in d:
                d[c + "_share_of_wave_cycles"] = round(d[c] / wc, 4)
    out[k] = d
json.dump({"source": "rocprofv3 --pmc (one pass), bench.py --steps 2 --warmup 1 --no-cpu; sums over all dispatches of a kernel; "
                     "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES cycles (MI355X_MICROARCH.md)",
           "per_kernel": out}, open(sys.argv[2], "w"), indent=1)
for k in sorted(out, key=lambda k: -out[k].get("SQ_WAVE_CYCLES", 0)):
    print(k, {c: v for c, v in out[k].items() if c.endswith("share_of_wave_cycles")}, "MFMA busy", out[k].get("SQ_VALU_MFMA_BUSY_CYCLES"))
