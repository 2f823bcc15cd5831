#!/bin/bash
# same-box A/B of bench.py's pipelined value under environment settings: tools/ab_env_bench.sh "A=1 B=2" "A=3" ...   ("-" = no setting;
# BENCH_ARGS="--workload 1080p --steps 60" for another workload)
for rep in 1 2 3; do
  for setting in "$@"; do
    if [ "$setting" = "-" ]; then s=""; else s="$setting"; fi
    env $s python bench.py --steps 20 --warmup 5 --no-cpu --no-e2e $BENCH_ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$setting]', d['value'], d['ms_per_step'], d['bytes_match_golden'])"
  done
done
