#!/bin/bash
# round 4: block_lists_kernel's workgroup size (EOGS_BL_T=256|512|1024 forces it; default: chosen by entries per block) per regime.
# One line per (regime, size): entries per block, the depth_sort group (= block_lists_kernel) and the step.
cd $GRAFT_REPO_ROOT
for cfg in "1048576 2048 0.01" "1048576 2048 trained" "1048576 1024 0.01" "1048576 1024 trained" "1048576 1024 0.1" "2000000 1024 0.1" "300000 1600 trained" "200000 1024 trained" "1048576 1536 0.01"; do
  for T in 256 512 1024 0; do
    EOGS_BL_T=$T python tools/regime_probe.py $cfg 20 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('[$cfg] T=$T entries/block', d['entries_per_block'], 'blocklists', d['block_lists'], 'depth_sort %.4f'%d['kernels'].get('depth_sort',0), 'step %.4f'%d['ms_per_step'])"
  done
done
