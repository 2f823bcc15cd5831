#!/bin/bash
# same-box A/B of the device stage: tools/ab_quick.sh "<libA> <libB> ..." [workloads...]   (devices differ by up to 10 %: only
# numbers taken in one gpurun call compare)
LIBS=$1; shift
for w in "${@:-raise natural 1080p 8k}"; do
  for rep in 1 2 3; do
    for lib in $LIBS; do
      echo -n "$(basename $lib) "; MPCODEC_LIB=$lib python tools/quick_bench.py $w 10 2>/dev/null
    done
  done
done
