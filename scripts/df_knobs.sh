#!/bin/bash
# developer sweep over the dataflow kernels' launch knobs on one box: steps/s and solve ms of the default bench command per setting
run() { env "$@" timeout 300 python bench.py --north-star none --other-arithmetic-steps 0 --no-cpu-baseline ${WLARGS:-} 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().splitlines()[-1]); print('%-44s %.1f steps/s  solve %.3f ms  %s' % ('$*', j['value'], j['roofline']['solve_ms_per_step'], j['roofline']['kernel'][:22]))"; }
run A=0
run EDYNHIP_DF_NAP=1
run EDYNHIP_DF_NAP=0
run EDYNHIP_DF_LANES=4
run EDYNHIP_DF_LANES=1
run EDYNHIP_DF_WAVES=768
run EDYNHIP_DF_WAVES=512
run EDYNHIP_DFP_WAVES=512
run EDYNHIP_DFP_WAVES=2048
run A=0
