#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace statistics and the two HBM-traffic PMC passes of the default bench
# command, reduced to small text/json summaries under gpurun_out/ (the raw rocpd databases stay on the box).
# usage: scripts/profile_round.sh r01
set -u
R=${1:-r01}
OUT=$PWD/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
W=/tmp/prof_$R; rm -rf $W; mkdir -p $W
STEPS=300; WARM=120   # bench.py's defaults: the same command the bench line comes from
( cd /tmp && rocprofv3 --kernel-trace --stats -d $W/kt -o r -- python $OLDPWD/bench.py --steps $STEPS --warmup $WARM > $OUT/bench_under_rocprof.json 2> $W/kt.log )
python scripts/prof_summary.py $W/kt $((STEPS + WARM)) k_contact_solve $STEPS > $OUT/${R}_kernel_stats_pile32k.txt
python scripts/prof_timeline.py $W/kt $((STEPS + WARM - 50)) > $OUT/${R}_timeline_pile32k.txt 2>&1 || true
python bench.py --stage-timing --no-cpu-baseline > $OUT/${R}_bench_stage_timing.json 2> /dev/null || true
for C in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && rocprofv3 --kernel-trace --pmc $C -d $W/$C -o r -- python $OLDPWD/bench.py --steps 60 --warmup 5 > /dev/null 2> $W/$C.log )
done
python scripts/pmc_traffic.py $W/FETCH_SIZE $W/WRITE_SIZE pile32k > $OUT/traffic.json
( cd /tmp && rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $W/sq -o r -- python $OLDPWD/bench.py --steps 60 --warmup 5 > /dev/null 2> $W/sq.log )
python scripts/pmc_summary.py $W/sq > $OUT/${R}_pmc_sq_counters_pile32k.txt 2>&1 || true
ls -la $OUT | head -30
