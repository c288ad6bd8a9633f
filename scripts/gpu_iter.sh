#!/bin/bash
# One short gpurun call of an optimisation iteration: a chosen subset of the GPU tests, A/B against libedynhip_base.so, knob runs,
# optional dataflow traces. usage: scripts/gpu_iter.sh <tag> "<pytest -k expression or ''>" (then sources scripts/runs/<tag>.sh: the run / trace lines of this iteration)
TAG=${1:-it}; KEXPR=${2:-}; mkdir -p gpurun_out/$TAG; export TMPDIR=/tmp
line() { python - "$1" "$2" <<'PY'
import json, sys
try:
    j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print("%-44s %.1f steps/s  %.3f ms/step  solve %.3f ms  points %d colours %d" % (sys.argv[2], j["value"], j["ms_per_step"], j["roofline"]["solve_ms_per_step"], j["config"]["contact_points"], j["config"]["colours"]))
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
}
run() {  # run <name> <workload> [ENV=..]...
  local name=$1 wl=$2; shift 2
  case $wl in islands256k) A="--steps 60 --warmup 10";; polyheap32k) A="--steps 100 --warmup 10";; *) A="";; esac
  env "$@" timeout 600 python bench.py --workload $wl $A --north-star none --other-arithmetic-steps 0 --no-cpu-baseline --no-shim > gpurun_out/$TAG/$name.json 2> gpurun_out/$TAG/$name.err
  line gpurun_out/$TAG/$name.json "$name $*"
}
trace() {  # trace <name> [ENV=..]...
  local name=$1; shift
  env "$@" EDYNHIP_DF_TRACE=/tmp/df_$name.bin EDYNHIP_DFP_TRACE=/tmp/dfp_$name.bin EDYNHIP_DF_TRACE_STEP=200 timeout 200 python bench.py --steps 150 --warmup 100 --no-cpu-baseline --no-shim --north-star none --other-arithmetic-steps 0 > /dev/null 2>&1
  python scripts/df_trace.py /tmp/df_$name.bin > gpurun_out/$TAG/dftrace_velocity_$name.txt 2>&1
  python scripts/df_trace.py /tmp/dfp_$name.bin > gpurun_out/$TAG/dftrace_position_$name.txt 2>&1
  echo "--- trace $name $*"; sed -n '1p;9p' gpurun_out/$TAG/dftrace_velocity_$name.txt; head -4 gpurun_out/$TAG/dftrace_position_$name.txt
}
prof() {  # prof <name> [ENV=..]... : rocprofv3 kernel statistics of the default bench command (per-step averages)
  local name=$1; shift
  local W=/tmp/prof_$name; rm -rf $W; mkdir -p $W
  ( cd /tmp && env "$@" rocprofv3 --kernel-trace --stats -d $W/kt -o r -- python $OLDPWD/bench.py --workload ${PROF_WL:-pile32k} --steps 300 --warmup 20 --north-star none --other-arithmetic-steps 0 --no-cpu-baseline --no-shim > /dev/null 2> $W/kt.log )
  python scripts/prof_summary.py $W/kt 440 k_contact_solve 300 > gpurun_out/$TAG/kernel_stats_$name.txt 2>&1
  echo "--- prof $name $*"; head -${PROF_LINES:-22} gpurun_out/$TAG/kernel_stats_$name.txt | cut -c1-118
}
if [ -n "$KEXPR" ]; then
  timeout 900 python -m pytest tests -m gpu -x -q --durations=30 -k "$KEXPR" > gpurun_out/$TAG/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -45 gpurun_out/$TAG/pytest_gpu.log
fi
BASE=$PWD/edyn_amd/libedynhip_base.so
[ -f scripts/runs/$TAG.sh ] && source scripts/runs/$TAG.sh
