for REP in 1 2; do
  run pile32k_base_$REP pile32k EDYNHIP_LIB=$BASE
  run pile32k_new_$REP pile32k X=0
done
run pile32k_prep1 pile32k EDYNHIP_PREP_PER_POINT=0
run pile32k_dfp1024 pile32k EDYNHIP_DFP_WAVES=1024
run pile32k_dfp2048 pile32k EDYNHIP_DFP_WAVES=2048
run pile32k_dfp768 pile32k EDYNHIP_DFP_WAVES=768
for WL in pile8k mixed32k islands256k polyheap32k; do
  run ${WL}_base $WL EDYNHIP_LIB=$BASE
  run ${WL}_new $WL X=0
done
timeout 300 python bench.py --stage-timing --north-star none --no-cpu-baseline > gpurun_out/$TAG/stage_timing.json 2>/dev/null
python -c "
import json; j=json.loads([l for l in open('gpurun_out/$TAG/stage_timing.json') if l.startswith('{')][-1]); print({k: round(v,4) for k,v in j.get('stages_ms_per_step',{}).items()})"
trace d512 X=0
trace d1024 EDYNHIP_DFP_WAVES=1024
