#!/bin/bash
# L2 hit rate and L1->L2 read latency of the pursuit kernel at several workgroup counts: tools/l2_probe.sh "256 224 128"
# (why does a workgroup's cost rise 30 % from 128 to 256 workgroups?)  Counters in their own passes, no trace domains.
R=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
for wg in $1; do
  for set in "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" "TCC_REQ_sum TCC_TAG_STALL_sum" "TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum"; do
    tag=$(echo $set | tr ' ' '_')
    rm -rf /tmp/l2p
    MPC_WORKGROUPS=$wg rocprofv3 --pmc $set --output-format csv -d /tmp/l2p -- python3 $R/tools/quick_bench.py raise 2 > /tmp/l2p.log 2>&1
    python3 - "$wg" <<'PY'
import csv, glob, sys, collections
f = glob.glob("/tmp/l2p/*/*counter_collection.csv")
if not f:
    print("wg", sys.argv[1], "no counters"); sys.exit(0)
acc = collections.defaultdict(float); n = 0
for r in csv.DictReader(open(f[0])):
    if "mp_pursuit_kernel" in r["Kernel_Name"]:
        acc[r["Counter_Name"]] += float(r["Counter_Value"])
launches = 5.0
print("wg", sys.argv[1], {k: round(v / launches) for k, v in acc.items()})
PY
  done
done
