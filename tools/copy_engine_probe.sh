#!/bin/bash
# which engine takes the container's device-to-host copy?  tools/copy_engine_probe.sh "-" "GPU_FORCE_BLIT_COPY_SIZE=0" ...
# (a blit copy shows up as __amd_rocclr_copyBuffer kernels in the kernel trace, an SDMA copy does not)
R=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
i=0
for setting in "$@"; do
  i=$((i+1))
  if [ "$setting" = "-" ]; then s=""; else s="$setting"; fi
  out=$R/gpurun_out/copy_probe_$i
  rm -rf $out
  env $s rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $R/tools/chain_bench.py raise 20 > $out.log 2>&1
  echo "[$setting] $(grep 'records -> container' $out.log)"
  python3 - "$out" <<PY
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "copyBuffer" in r["Name"] or "fillBuffer" in r["Name"]:
        print("    %-40s calls %4s avg %8.1f us" % (r["Name"][:40], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
