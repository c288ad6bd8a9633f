for cfg in "64 256" "32 512"; do set -- $cfg
echo -n "wave_lanes $1 waves $2: "; EDYNHIP_DF_WAVELANES=$1 EDYNHIP_DF_WAVES=$2 timeout 100 python bench.py --steps 150 --warmup 100 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['stages_ms_per_step']['solve_velocity_ms'],3))"
done
EDYNHIP_DF_TRACE=/tmp/df.bin EDYNHIP_DF_TRACE_STEP=200 timeout 100 python bench.py --steps 150 --warmup 100 > /dev/null 2>&1
python scripts/df_trace.py /tmp/df.bin | head -8
