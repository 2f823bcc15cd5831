#!/bin/bash
# A/B an environment setting on the same GPU: tools/ab_env.sh "VAR=value" [workloads...]
SETTING=$1; shift
for w in "${@:-1080p raise 8k}"; do
  for rep in 1 2; do
    for mode in base set; do
      if [ $mode = set ]; then export $SETTING; else unset ${SETTING%%=*}; fi
      python bench.py --steps 6 --warmup 2 --no-cpu --workload $w 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w', '$mode', d['value'], d['ms_per_step'])"
    done
  done
done
