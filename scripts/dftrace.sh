EDYNHIP_DF_TRACE=/tmp/df.bin EDYNHIP_DF_TRACE_STEP=200 timeout 100 python bench.py --steps 150 --warmup 100 > /dev/null 2>&1
python scripts/df_trace.py /tmp/df.bin
EDYNHIP_DF_WAVELANES=16 EDYNHIP_DF_WAVES=1024 EDYNHIP_DF_TRACE=/tmp/df.bin EDYNHIP_DF_TRACE_STEP=200 timeout 100 python bench.py --steps 150 --warmup 100 > /dev/null 2>&1
python scripts/df_trace.py /tmp/df.bin | head -14
