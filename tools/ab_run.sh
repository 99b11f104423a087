#!/bin/bash
# Runs on the GPU box: short bench per variant in build_ab/, printing the per-kernel times.
# usage: tools/ab_run.sh "<bench args>" variant...
ARGS=$1; shift
for v in "$@"; do
  echo "== $v"
  RRT_HIP_LIB=$PWD/build_ab/librrt_hip_$v.so timeout -k 10 300 python bench.py $ARGS --cpu-seconds 0 --no-calibrate | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('  %.1f Msamples/s  %.1f ms/step  ' % (d['value']*1e-6, d['ms_per_step']) + '  '.join('%s %.1f' % kv for kv in d['kernel_ms_per_step'].items()))
" || exit 1
done
