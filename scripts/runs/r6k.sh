# round 6, iteration k: island certificate prefers long-lived contacts - how many steps still relabel in full? (calls of k_cc_hook_bodies)
for WL in mixed32k polyheap32k pile32k islands256k; do
  PROF_WL=$WL PROF_LINES=40 prof stable_$WL > /dev/null
  PROF_WL=$WL PROF_LINES=40 prof first_$WL EDYNHIP_LIB=$BASE > /dev/null
  echo "== $WL: stable-first certificate"; grep "k_cc_\|k_contact_solve" gpurun_out/$TAG/kernel_stats_stable_$WL.txt | cut -c1-112
  echo "== $WL: first contact (base build)"; grep "k_cc_\|k_contact_solve" gpurun_out/$TAG/kernel_stats_first_$WL.txt | cut -c1-112
done
for WL in mixed32k polyheap32k pile32k; do run first_$WL $WL EDYNHIP_LIB=$BASE; run stable_$WL $WL; done
