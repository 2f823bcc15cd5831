#!/bin/bash
# one tuning iteration on a GPU box: parity (GPU suite), stamps + residency, device-stage times.  tools/r03_iter.sh <tag> [fast]
set -e
tag=$1
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
if [ "$2" = "fast" ]; then
  python -m pytest tests/test_gpu_golden_frames.py tests/test_gpu_parity.py -m gpu -x -q > $O/${tag}_gputests.log 2>&1 && tail -2 $O/${tag}_gputests.log
else
  python -m pytest tests -m gpu -x -q > $O/${tag}_gputests.log 2>&1 && tail -2 $O/${tag}_gputests.log
fi
for w in raise natural 1080p; do
  MPCODEC_LIB=$R/imageexperiments_amd/lib/libmpcodec_stamps.so python tools/quick_bench.py $w 2 > $O/${tag}_stamps_$w.log 2>&1
  tail -3 $O/${tag}_stamps_$w.log | cut -c1-900
done
for w in raise natural 1080p 8k; do python tools/quick_bench.py $w 10 2>/dev/null; done | tee $O/${tag}_quick.log
