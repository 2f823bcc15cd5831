#!/bin/bash
# Everything profiles/ holds for one round, in one GPU-box call: tools/profile_round.sh <tag, e.g. r01>
# Outputs land in gpurun_out/<tag>_*; copy the summaries into profiles/ afterwards.
set -e
tag=$1
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for w in 1080p raise; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_stats_$w -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu --workload $w > $O/${tag}_stats_$w.log 2>&1
  cp $O/${tag}_stats_$w/*/*kernel_stats.csv $O/${tag}_kernel_stats_$w.csv
  echo "kernel stats $w done"
done
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${tag}_pmc_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu > $O/${tag}_pmc_fetch.log 2>&1
echo "pmc fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${tag}_pmc_write -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu > $O/${tag}_pmc_write.log 2>&1
echo "pmc write done"
python3 $R/tools/pmc_summary.py $O/${tag}_pmc_fetch/*/*counter_collection.csv $O/${tag}_pmc_write/*/*counter_collection.csv 3 \
  "1920x1080 synthetic, K=8, quality 3.5 (bench.py default)" $O/${tag}_pmc_1080p.json
cp $O/${tag}_pmc_1080p.json $R/profiles/${tag}_pmc_1080p.json      # bench.py reads roofline.traffic from here
cd $R
python3 bench.py --steps 20 --warmup 3 > $O/${tag}_bench_1080p.json 2> $O/${tag}_bench_1080p.err
echo "bench 1080p done"
python3 bench.py --steps 6 --warmup 2 --no-cpu --workload raise > $O/${tag}_bench_raise.json 2>/dev/null
python3 bench.py --steps 6 --warmup 2 --no-cpu --workload 8k > $O/${tag}_bench_8k.json 2>/dev/null
cut -c1-400 $O/${tag}_bench_1080p.json
