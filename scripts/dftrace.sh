# usage: bash scripts/dftrace.sh [out-file]   (EDYNHIP_LIB selects the build)
EDYNHIP_DF_TRACE=/tmp/df.bin EDYNHIP_DF_TRACE_STEP=200 timeout 100 python bench.py --steps 150 --warmup 100 --no-cpu-baseline --north-star none --other-arithmetic-steps 0 > /dev/null 2>&1
python scripts/df_trace.py /tmp/df.bin | sed -n '1p;8,10p;14,32p'
