for L in 32768 65536; do
EDYNHIP_DF_LANES=$L EDYNHIP_DF_PREDICT=4 EDYNHIP_DF_BACKOFF=4 EDYNHIP_DF_TRACE=/tmp/df_$L.bin EDYNHIP_DF_TRACE_STEP=200 timeout 100 python bench.py --steps 150 --warmup 100 > /dev/null 2>&1
python scripts/df_trace.py /tmp/df_$L.bin
done
