EDYNHIP_DF_TRACE=/tmp/df.bin EDYNHIP_DF_TRACE_STEP=200 timeout 100 python bench.py --steps 150 --warmup 100 --no-cpu-baseline > /dev/null 2>&1
python scripts/df_trace.py /tmp/df.bin | sed -n '1p;8,10p;14,32p'
