#!/bin/bash
# kernel-time summary of one bench workload on the GPU box: tools/prof.sh <tag> <workload> [bench args...]
# writes gpurun_out/<tag>/.../*_kernel_stats.csv and prints its head
tag=$1; w=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$tag -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu --workload $w "$@" > $R/gpurun_out/$tag.log 2>&1
cat $R/gpurun_out/$tag/*/*kernel_stats.csv | cut -c1-160 | head -11
