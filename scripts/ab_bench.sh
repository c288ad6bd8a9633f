#!/bin/bash
# A/B of two builds on ONE box (boxes of the pool differ by several per cent): edyn_amd/libedynhip_base.so (built from an earlier
# commit, see the commit log) against the current edyn_amd/libedynhip.so, alternating, default bench command without the CPU legs.
# usage: scripts/ab_bench.sh <tag> [workload ...]
TAG=${1:-ab}; shift || true
WLS=${@:-pile32k}
mkdir -p gpurun_out/$TAG
for WL in $WLS; do
  case $WL in islands256k) A="--steps 60 --warmup 10";; polyheap32k) A="--steps 100 --warmup 10";; *) A="";; esac
  for REP in 1 2; do
    for B in base new; do
      if [ $B = base ]; then export EDYNHIP_LIB=$PWD/edyn_amd/libedynhip_base.so; else unset EDYNHIP_LIB; fi
      timeout 600 python bench.py --workload $WL $A --north-star none --other-arithmetic-steps 0 --no-cpu-baseline > gpurun_out/$TAG/${WL}_${B}_$REP.json 2> gpurun_out/$TAG/${WL}_${B}_$REP.err
      python - <<PY
import json
try:
    j = json.loads([l for l in open("gpurun_out/$TAG/${WL}_${B}_$REP.json") if l.startswith("{")][-1])
    print("$WL $B $REP: %.1f steps/s, %.3f ms/step, solve %.3f ms, points %d, colours %d" % (j["value"], j["ms_per_step"], j["roofline"]["solve_ms_per_step"], j["config"]["contact_points"], j["config"]["colours"]))
except Exception as e:
    print("$WL $B $REP: failed", e)
PY
    done
  done
done
unset EDYNHIP_LIB
