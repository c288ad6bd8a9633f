for W in 512 1024; do
EDYNHIP_DF_WAVES=$W EDYNHIP_DF_TRACE=/tmp/df.bin EDYNHIP_DF_TRACE_STEP=200 timeout 100 python bench.py --steps 150 --warmup 100 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['stages_ms_per_step']['solve_velocity_ms'],3))"
python scripts/df_trace.py /tmp/df.bin | sed -n '1p;8,9p;14,26p'
done
