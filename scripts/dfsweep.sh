timeout 120 python scripts/dev_parity.py pile8 30 2>&1 | tail -3
timeout 300 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
for W in 256 512 1024; do
echo -n "pos waves $W: "; EDYNHIP_DFP_WAVES=$W timeout 100 python bench.py --steps 150 --warmup 100 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), {k: round(v,3) for k,v in d['stages_ms_per_step'].items()})"
done
echo -n "pos per-colour: "; EDYNHIP_DATAFLOW_POS=0 timeout 100 python bench.py --steps 150 --warmup 100 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), {k: round(v,3) for k,v in d['stages_ms_per_step'].items()})"
