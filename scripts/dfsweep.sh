for WL in islands256k mixed32k pile8k; do for T in 1 0; do
echo -n "$WL two_lane $T: "; EDYNHIP_DF_TWOLANE=$T timeout 200 python bench.py --workload $WL --steps 100 --warmup 60 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), {k: round(v,3) for k,v in d['stages_ms_per_step'].items() if k in ('solve_velocity_ms','solve_position_ms','step_ms')})"
done; done
