timeout 400 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for cfg in "1 0" "1 512" "1 2048" "0 0"; do set -- $cfg
echo -n "two_lane $1 waves $2: "; EDYNHIP_DF_TWOLANE=$1 EDYNHIP_DF_WAVES=$2 timeout 100 python bench.py --steps 150 --warmup 100 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), {k: round(v,3) for k,v in d['stages_ms_per_step'].items() if k in ('solve_velocity_ms','solve_position_ms','step_ms')})"
done
