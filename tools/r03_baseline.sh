#!/bin/bash
# round-3 baseline on a fresh box: GPU suite, stamps + residency for the three workloads and the natural frame, quick device-stage times
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
python -m pytest tests -m gpu -x -q > $O/r03_gputests0.log 2>&1 && tail -2 $O/r03_gputests0.log
for w in raise natural 1080p 8k; do
  MPCODEC_LIB=$R/imageexperiments_amd/lib/libmpcodec_stamps.so python tools/quick_bench.py $w 2 > $O/r03_stamps0_$w.log 2>&1
  tail -3 $O/r03_stamps0_$w.log | cut -c1-900
done
for w in raise natural 1080p 8k; do python tools/quick_bench.py $w 10 2>/dev/null; done | tee $O/r03_quick0.log
python bench.py --steps 20 --warmup 5 --no-cpu 2>/dev/null | cut -c1-400
