#!/bin/bash
# One gpurun call of an optimisation round: GPU tests (-x), A/B of edyn_amd/libedynhip_base.so against the current build on one box,
# developer knobs of the current build, dataflow traces (velocity + position). Everything lands in gpurun_out/<tag>/.
# usage: scripts/gpu_ab_round.sh <tag> [notests]
TAG=${1:-ab}; mkdir -p gpurun_out/$TAG; export TMPDIR=/tmp
line() { python - "$1" "$2" <<'PY'
import json, sys
try:
    j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print("%-40s %.1f steps/s  %.3f ms/step  solve %.3f ms  points %d colours %d" % (sys.argv[2], j["value"], j["ms_per_step"], j["roofline"]["solve_ms_per_step"], j["config"]["contact_points"], j["config"]["colours"]))
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
}
run() {  # run <name> <workload> [ENV=..]...
  local name=$1 wl=$2; shift 2
  case $wl in islands256k) A="--steps 60 --warmup 10";; polyheap32k) A="--steps 100 --warmup 10";; *) A="";; esac
  env "$@" timeout 600 python bench.py --workload $wl $A --north-star none --other-arithmetic-steps 0 --no-cpu-baseline > gpurun_out/$TAG/$name.json 2> gpurun_out/$TAG/$name.err
  line gpurun_out/$TAG/$name.json "$name $*"
}
if [ "${2:-}" != notests ]; then
  timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/$TAG/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/$TAG/pytest_gpu.log
fi
BASE=$PWD/edyn_amd/libedynhip_base.so
for REP in 1 2; do
  run pile32k_base_$REP pile32k EDYNHIP_LIB=$BASE
  run pile32k_new_$REP pile32k X=0
done
for WL in pile8k mixed32k; do
  run ${WL}_base $WL EDYNHIP_LIB=$BASE
  run ${WL}_new $WL X=0
done
run pile32k_nospec pile32k EDYNHIP_SPECULATE=0
run pile32k_dfp1024 pile32k EDYNHIP_DFP_WAVES=1024
run pile32k_dfp2048 pile32k EDYNHIP_DFP_WAVES=2048
run pile32k_dfp256 pile32k EDYNHIP_DFP_WAVES=256
# stage timing of the new build
timeout 300 python bench.py --stage-timing --north-star none --other-arithmetic-steps 0 --no-cpu-baseline > gpurun_out/$TAG/stage_timing.json 2>/dev/null
python -c "
import json; j=json.loads([l for l in open('gpurun_out/$TAG/stage_timing.json') if l.startswith('{')][-1]); print({k: round(v,4) for k,v in j.get('stages_ms_per_step',{}).items()})"
# traces
EDYNHIP_DF_TRACE=/tmp/df.bin EDYNHIP_DFP_TRACE=/tmp/dfp.bin EDYNHIP_DF_TRACE_STEP=200 timeout 200 python bench.py --steps 150 --warmup 100 --no-cpu-baseline --north-star none --other-arithmetic-steps 0 > /dev/null 2>&1
python scripts/df_trace.py /tmp/df.bin > gpurun_out/$TAG/dftrace_velocity.txt 2>&1; sed -n '1p;8,10p' gpurun_out/$TAG/dftrace_velocity.txt
python scripts/df_trace.py /tmp/dfp.bin > gpurun_out/$TAG/dftrace_position.txt 2>&1; cat gpurun_out/$TAG/dftrace_position.txt | head -30
