#!/bin/bash
# CU time of the kernels behind the pursuit: tools/chain_bench.py on 8 CUs of one XCD (ROC_GLOBAL_CU_MASK=0xff), where every kernel
# is throughput-bound -- 8 x the average duration is the CU time a frame's chains take from a pursuit running beside them.
#   tools/chain_cu_time.sh [workload] [mask]
R=$(cd "$(dirname "$0")/.." && pwd)
w=${1:-raise}; m=${2:-0xff}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/chain_cu
ROC_GLOBAL_CU_MASK=$m rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/chain_cu -- python3 $R/tools/chain_bench.py $w 5 2>/dev/null | grep "records ->"
python3 - <<PY
import csv, glob
tot = 0.0
for r in csv.DictReader(open(glob.glob("/tmp/chain_cu/*/*kernel_stats.csv")[0])):
    n = r["Name"]
    if "ent_" in n or "mp_stream" in n:
        calls = int(r["Calls"]); per_frame = float(r["TotalDurationNs"]) / 1e3 / 7.0      # 2 warm-up + 5 timed calls
        tot += per_frame
        print("   %-52s %3d calls  %9.1f us per frame" % (n[:52], calls, per_frame))
print("   total %.1f us per frame on mask $m" % tot)
PY
