timeout 400 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for i in 1 2; do echo -n "pile32k default bench: "; timeout 200 python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],4), round(d['roofline']['frac'],3))"; done
