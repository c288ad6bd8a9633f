#!/bin/bash
# Runs on the GPU box (via gpurun): for each workload the bench line, rocprofv3 kernel-trace statistics of the same command
# and the two HBM-traffic PMC passes, reduced to small text/json summaries under gpurun_out/ (copy into profiles/; the raw
# rocpd databases stay on the box). The headline workload additionally gets one step's timeline, SQ counters and a
# --stage-timing run.
# usage: scripts/profile_round.sh r02 [workload ...]        (default: pile32k islands256k)
set -u
R=${1:-r02}; shift || true
WLS=${@:-pile32k islands256k}
OUT=$PWD/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
for WL in $WLS; do
  W=/tmp/prof_${R}_$WL; rm -rf $W; mkdir -p $W
  # bench.py settles the scene (120 steps) before the warm-up and the timed steps: SETTLE + WARM + STEPS steps are profiled
  SETTLE=120
  case $WL in islands256k|islands1m) STEPS=60; WARM=10;; polyheap32k) STEPS=100; WARM=10; SETTLE=240;; *) STEPS=300; WARM=20;; esac
  ( cd /tmp && rocprofv3 --kernel-trace --stats -d $W/kt -o r -- python $OLDPWD/bench.py --workload $WL --steps $STEPS --warmup $WARM --north-star none --other-arithmetic-steps 0 --no-cpu-baseline --no-shim > $OUT/${R}_bench_under_rocprof_$WL.json 2> $W/kt.log )
  case $WL in chains16k|ragdolls1k) KN=k_island_velocity;; *) KN=k_contact_solve;; esac
  python scripts/prof_summary.py $W/kt $((SETTLE + STEPS + WARM)) $KN $STEPS > $OUT/${R}_kernel_stats_$WL.txt
  # HBM-side traffic of the velocity-solve kernels of EVERY workload (r04): two PMC passes, timed region only, scaled per algorithmic byte
  case $WL in islands256k|islands1m) PS=20;; *) PS=40;; esac
  for C in FETCH_SIZE WRITE_SIZE; do
    ( cd /tmp && rocprofv3 --kernel-trace --pmc $C -d $W/$C -o r -- python $OLDPWD/bench.py --workload $WL --steps $PS --warmup 5 --north-star none --other-arithmetic-steps 0 --no-cpu-baseline --no-shim > $W/$C.json 2> $W/$C.log )
  done
  python scripts/pmc_traffic.py $W/FETCH_SIZE $W/WRITE_SIZE $WL $W/FETCH_SIZE.json > $OUT/${R}_traffic_$WL.json 2> $OUT/${R}_traffic_$WL.err || true
  # ... and of the other bandwidth kernels of the step (the row preparation had no counter profile), last $PS timed steps
  python scripts/pmc_kernel_traffic.py $W/FETCH_SIZE $W/WRITE_SIZE $PS k_prep_contacts k_np_merge k_np_detect k_pos_seed k_push_links k_bp_pairs > $OUT/${R}_traffic_other_kernels_$WL.json 2> /dev/null || true
  if [ $WL = pile32k ]; then
    python scripts/prof_timeline.py $W/kt $((SETTLE + STEPS + WARM - 50)) > $OUT/${R}_timeline_$WL.txt 2>&1 || true
    python bench.py --stage-timing --north-star none --other-arithmetic-steps 0 --no-cpu-baseline --no-shim > $OUT/${R}_bench_stage_timing.json 2> /dev/null || true
    ( cd /tmp && rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $W/sq -o r -- python $OLDPWD/bench.py --steps 60 --warmup 5 --north-star none --other-arithmetic-steps 0 --no-cpu-baseline --no-shim > /dev/null 2> $W/sq.log )
    python scripts/pmc_summary.py $W/sq > $OUT/${R}_pmc_sq_counters_$WL.txt 2>&1 || true
  fi
done
# one traffic.json keyed by workload (what bench.py reads as roofline.traffic)
ROUND_TAG=$R python - <<'PY'
import glob, json, os
out = {}
for f in sorted(glob.glob("gpurun_out/%s_traffic_*.json" % os.environ.get("ROUND_TAG", "r04"))):
    try:
        j = json.load(open(f)); out[j["workload"]] = j
    except Exception:
        pass
json.dump(out, open("gpurun_out/traffic.json", "w"), indent=1)
PY
ls -la $OUT | head -40
