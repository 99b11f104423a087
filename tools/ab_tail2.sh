#!/bin/bash
# Runs on the GPU box: tail kernel on / off at several sample counts.
for spp in ${SPPS:-4 16 64}; do
for t in ${TAILS:-0 262144 2097152}; do
  echo "== spp $spp tail paths $t"
  RRT_TAIL_PATHS_OVERRIDE=$t timeout -k 10 300 python bench.py --steps 5 --warmup 1 --spp $spp --cpu-seconds 0 --no-calibrate 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('  %.1f Msamples/s  %.2f ms/step  ' % (d['value']*1e-6, d['ms_per_step']) + '  '.join('%s %.1f' % kv for kv in d['kernel_ms_per_step'].items()))
" || exit 1
done
done
