# round 6, iteration t: XCD-local task lists (EDYNHIP_DF_XCD=1) once more on the final build, every BASELINE configuration
run pile32k pile32k
run pile32k_xcd pile32k EDYNHIP_DF_XCD=1
run pile8k pile8k
run pile8k_xcd pile8k EDYNHIP_DF_XCD=1
run mixed32k mixed32k
run mixed32k_xcd mixed32k EDYNHIP_DF_XCD=1
run islands256k islands256k
run islands256k_xcd islands256k EDYNHIP_DF_XCD=1
