#!/bin/bash
# Runs on the GPU box: bench under values of one environment variable.  usage: tools/ab_env.sh VAR "v1 v2 ..." <bench args>
VAR=$1; VALUES=$2; shift 2
for v in $VALUES; do
  echo "== $VAR=$v"
  env $VAR=$v timeout -k 10 300 python bench.py --steps 5 --warmup 1 --cpu-seconds 0 --no-calibrate "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('  %.1f Msamples/s  %.2f ms/step  ' % (d['value']*1e-6, d['ms_per_step']) + '  '.join('%s %.1f' % kv for kv in d['kernel_ms_per_step'].items()))
" || exit 1
done
