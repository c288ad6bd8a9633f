run polyheap32k_new_1 polyheap32k X=0
run polyheap32k_new_2 polyheap32k X=0
run pile32k_new pile32k X=0
( cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_poly/kt -o r -- python $OLDPWD/bench.py --workload polyheap32k --steps 100 --warmup 10 --north-star none --no-cpu-baseline > /dev/null 2> /tmp/prof_poly.log )
python scripts/prof_summary.py /tmp/prof_poly/kt 350 k_contact_solve 100 > gpurun_out/$TAG/kernel_stats_polyheap.txt 2>&1; head -9 gpurun_out/$TAG/kernel_stats_polyheap.txt | cut -c1-118
