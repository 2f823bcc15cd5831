"""Kernel timeline of a rocprofv3 --kernel-trace csv: python tools/timeline.py <kernel_trace.csv> [first_ms] [last_ms]
Prints start / end (ms since the first pursuit launch) and the stream-less queue id of every kernel in the window."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
lo = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
hi = float(sys.argv[3]) if len(sys.argv) > 3 else 1e9
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = min(int(r["Start_Timestamp"]) for r in rows if "pursuit" in r["Kernel_Name"])
for r in rows:
    a, b = (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6
    if a < lo or a > hi:
        continue
    name = r["Kernel_Name"].split("(")[0].replace("mpc::", "").replace("void ", "")[:40]
    print(f"{a:9.3f} {b:9.3f} {b - a:8.3f}  q{r.get('Queue_Id', '?'):>3}  {name}")
