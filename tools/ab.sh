#!/bin/bash
# A/B two builds of libmpcodec.so in one process-sequence on the same GPU: tools/ab.sh <libA> <libB> [workloads...]
A=$1; B=$2; shift 2
for w in "${@:-1080p raise}"; do
  for rep in 1 2; do
    for lib in $A $B; do
      MPCODEC_LIB=$lib python bench.py --steps 6 --warmup 2 --no-cpu --workload $w 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w', '$(basename $lib)', d['value'], d['ms_per_step'])"
    done
  done
done
