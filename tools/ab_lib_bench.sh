#!/bin/bash
# same-box A/B of bench.py's pipelined value between library builds: tools/ab_lib_bench.sh "<libA> <libB>"   (BENCH_ARGS as in ab_env_bench.sh)
for rep in 1 2 3; do
  for lib in $1; do
    MPCODEC_LIB=$lib python bench.py --steps 20 --warmup 5 --no-cpu --no-e2e $BENCH_ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$(basename $lib)]', d['value'], d['ms_per_step'], d['bytes_match_golden'])"
  done
done
