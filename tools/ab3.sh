#!/bin/bash
# A/B/... several builds of libmpcodec.so on the same GPU: tools/ab3.sh "<lib> <lib> ..." [workloads...]
LIBS=$1; shift
for w in "${@:-1080p raise}"; do
  for rep in 1 2; do
    for lib in $LIBS; do
      MPCODEC_LIB=$lib python bench.py --steps 6 --warmup 2 --no-cpu --workload $w 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w', '$(basename $lib)', d['value'], d['ms_per_step'])"
    done
  done
done
