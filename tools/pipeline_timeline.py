"""Two consecutive frames of bench.py's pipelined call, every kernel the device ran, from a rocprofv3 --kernel-trace csv:
    python tools/pipeline_timeline.py <kernel_trace.csv> <out.txt>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "mp_pursuit_kernel" in r["Kernel_Name"]]
# pursuit launches of the pipeline are followed by the stream assembly before the next pursuit (the device-stage loop's are not)
pipe = [i for k, i in enumerate(idx[:-1]) if any("mp_stream_count" in rows[j]["Kernel_Name"] for j in range(i + 1, idx[k + 1]))]
a, b = pipe[-5], pipe[-3]
t0 = int(rows[a]["Start_Timestamp"])
out = ["# bench.py: two consecutive frames of the timed call in steady state, every kernel the device ran (rocprofv3 --kernel-trace)\n"
       "# start_ms end_ms duration_ms kernel   -- pursuits back to back on one stream (224 workgroups on 256 CUs); behind each, on the "
       "side streams: stream assembly + entropy phase 1 (f), later phase 2 + container copy (f), beside the next pursuits on the CUs "
       "they leave free.  The copyBuffer kernel is the profiler's doing: without rocprofv3 the container's copy is an SDMA copy "
       "(DESIGN.md 4).  Run bench.py with --no-e2e: the legs behind the timed call would be picked up instead\n"]
for r in rows[a:b + 1]:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6
    name = r["Kernel_Name"].split("(")[0].replace("mpc::", "").replace("void ", "")
    out.append(f"{s:8.3f} {e:8.3f} {e - s:7.3f}  {name}\n")
open(sys.argv[2], "w").write("".join(out))
