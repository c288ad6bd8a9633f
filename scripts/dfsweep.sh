for L in 16384 32768 49152 65536; do for P in 0 4; do
echo -n "lanes $L predict $P backoff 4: "; EDYNHIP_DF_LANES=$L EDYNHIP_DF_PREDICT=$P EDYNHIP_DF_BACKOFF=4 timeout 100 python bench.py --steps 150 --warmup 100 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['stages_ms_per_step']['solve_velocity_ms'],3))"
done; done
