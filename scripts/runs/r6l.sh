# round 6, iteration l: records stored straight into the pinned slot by the pack kernel (EDYNHIP_RECORDS_DIRECT=1) against the copy engine
cd tests/cpp
for REP in 1 2; do
  for M in sequential_exclusive asynchronous_exclusive sequential; do
    timeout 300 ./bench_update 32 120 300 $M | grep "^{" | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('copy   ', r['run'], r['update_steps_per_sec'], r['ratio'], r['ms_per_update'])"
    EDYNHIP_RECORDS_DIRECT=1 timeout 300 ./bench_update 32 120 300 $M | grep "^{" | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('direct ', r['run'], r['update_steps_per_sec'], r['ratio'], r['ms_per_update'])"
  done
done
cd ../..
