#!/bin/bash
# Everything profiles/ holds for one round, in one GPU-box call: tools/profile_round.sh <tag, e.g. r02>
# Only gpurun_out/ travels back from the GPU box: everything lands in gpurun_out/<tag>_*, the summaries to be committed in
# gpurun_out/<tag>_profiles/ -- afterwards, locally:  cp gpurun_out/<tag>_profiles/* profiles/
#   - rocprofv3 --kernel-trace --stats of the device stage (tools/quick_bench.py: the same launches bench.py times with events)
#   - two separate --pmc passes (FETCH_SIZE, WRITE_SIZE) of the same command -> per-kernel HBM bytes per launch
#   - one SQ counter pass (wave cycles / waits / instruction mix of the pursuit kernel)
#   - bench.py lines of the three workloads, the quality sweep of BASELINE configs[2], end-to-end timings
set -e
tag=$1
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
P=$O/${tag}_profiles
mkdir -p $O $P $R/profiles
cd /tmp && export TMPDIR=/tmp
for w in raise 1080p 8k; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_stats_$w -- python3 $R/tools/quick_bench.py $w 10 > $O/${tag}_stats_$w.log 2>&1
  cp $O/${tag}_stats_$w/*/*kernel_stats.csv $P/${tag}_kernel_stats_$w.csv
  echo "kernel stats $w done"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${tag}_pmc_fetch_$w -- python3 $R/tools/quick_bench.py $w 2 > $O/${tag}_pmc_fetch_$w.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${tag}_pmc_write_$w -- python3 $R/tools/quick_bench.py $w 2 > $O/${tag}_pmc_write_$w.log 2>&1
  python3 $R/tools/pmc_summary.py $O/${tag}_pmc_fetch_$w/*/*counter_collection.csv $O/${tag}_pmc_write_$w/*/*counter_collection.csv 5 \
    "tools/quick_bench.py $w 2 (3 warm-up + 2 timed device-stage launches)" $P/${tag}_pmc_$w.json
  cp $P/${tag}_pmc_$w.json $R/profiles/${tag}_pmc_$w.json      # bench.py (below) reads roofline.traffic from profiles/
  echo "pmc $w done"
done
# the whole pipeline of bench.py's value (pursuit, stream assembly, entropy kernels, the container's copy) in one trace
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_stats_pipeline -- python3 $R/bench.py --steps 14 --warmup 4 --no-cpu --no-e2e > $O/${tag}_stats_pipeline.log 2>&1
cp $O/${tag}_stats_pipeline/*/*kernel_stats.csv $P/${tag}_kernel_stats_pipeline_raise.csv
python3 $R/tools/pipeline_timeline.py $O/${tag}_stats_pipeline/*/*kernel_trace.csv $P/${tag}_timeline_pipeline_raise.txt
echo "pipeline stats done"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES \
  --output-format csv -d $O/${tag}_pmc_sq -- python3 $R/tools/quick_bench.py raise 2 > $O/${tag}_pmc_sq.log 2>&1
python3 $R/tools/pmc_sq_summary.py $O/${tag}_pmc_sq/*/*counter_collection.csv $P/${tag}_pmc_sq_raise.json
cp $P/${tag}_pmc_sq_raise.json $R/profiles/${tag}_pmc_sq_raise.json      # bench.py quotes it in roofline.limited_by
echo "pmc sq done"
cd $R
python3 bench.py > $P/${tag}_bench_raise.json 2> $O/${tag}_bench_raise.err
python3 bench.py --workload 1080p --no-cpu > $P/${tag}_bench_1080p.json 2>/dev/null
python3 bench.py --workload 8k --no-cpu > $P/${tag}_bench_8k.json 2>/dev/null
echo "bench lines done"
: > $P/${tag}_quality_sweep.jsonl
for q in 2.0 2.5 3.0 3.5 4.0 4.5 5.0 5.5 6.0; do
  python3 bench.py --quality $q --no-cpu --steps 10 --warmup 2 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print(json.dumps({'quality':$q,'value_Mpix_s':d['value'],'ms_per_step':d['ms_per_step'],'device_stage_Mpix_s':d['device_stage_Mpix_s'],'container_bytes':d['config']['container_bytes'],'bpp':d['config']['bpp'],'mfma_frac':d['roofline']['frac'],'tile_channel_steps':d['roofline']['tile_channel_steps_per_step']}))" >> $P/${tag}_quality_sweep.jsonl
done
echo "quality sweep done"
python3 tools/e2e_timing.py > $P/${tag}_e2e_timing.txt 2>/dev/null
# the frame pipeline's sensitivity to how the runtime maps its streams onto hardware queues
: > $P/${tag}_hw_queues.txt
for nq in 1 2 4 8; do
  GPU_MAX_HW_QUEUES=$nq python3 bench.py --steps 20 --warmup 5 --no-cpu --no-e2e 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('GPU_MAX_HW_QUEUES=$nq value', d['value'], 'Mpix/s ms_per_step', d['ms_per_step'], 'bytes_match_golden', d['bytes_match_golden'])" >> $P/${tag}_hw_queues.txt
done
# one frame at a time: latency of a single mpc_encode_image(_device) call, with the library's own trace
for w in raise 1080p; do python3 tools/single_frame_trace.py $w 2>&1 | grep -E "single frame|trace" | sed "s/^/$w: /" ; done > $P/${tag}_single_frame.txt
echo "queues + single frame done"
cut -c1-600 $P/${tag}_bench_raise.json
