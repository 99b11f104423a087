#!/bin/bash
# Runs on the GPU box: the bench frame with different thresholds for the tail kernel (RRT_OPT_TAIL_PATHS).
for t in 0 262144 1048576 2097152 4194304 8388608 16777216; do
  echo "== tail paths $t"
  RRT_TAIL_PATHS_OVERRIDE=$t timeout -k 10 300 python bench.py --steps 3 --warmup 1 --cpu-seconds 0 --no-calibrate "$@" | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('  %.1f Msamples/s  %.1f ms/step  ' % (d['value']*1e-6, d['ms_per_step']) + '  '.join('%s %.1f' % kv for kv in d['kernel_ms_per_step'].items()))
" || exit 1
done
