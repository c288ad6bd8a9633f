for cfg in "512 512" "1024 1024" "1024 2048"; do set -- $cfg
echo -n "islands256k waves vel $1 pos $2: "; EDYNHIP_DF_WAVES=$1 EDYNHIP_DFP_WAVES=$2 timeout 200 python bench.py --workload islands256k --steps 100 --warmup 60 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), {k: round(v,3) for k,v in d['stages_ms_per_step'].items() if k in ('solve_velocity_ms','solve_position_ms','step_ms')}, d['config'].get('colours'))"
done
