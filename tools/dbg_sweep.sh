#!/bin/bash
# performance experiments: bench at several CSH_DEBUG settings (k_tokens parts switched off), prints k_tokens / k_pack times
for d in ${DBG:-0}; do
  CSH_DEBUG=$d python bench.py --unique 8 --batch 1024 --steps 3 --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_ms']; print('debug=$d', 'k_tokens', k['k_tokens'], 'k_ac_runs', k['k_ac_runs'], 'k_pack', k['k_pack'], 'parity', d['parity_spot_check'])"
done
