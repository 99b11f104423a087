#!/bin/bash
# Runs on the GPU box: the bench scene at several sample counts, with the queue lengths the host reads back.
for spp in "$@"; do
  echo "== spp $spp"
  RRT_DEBUG_PENDING=1 timeout -k 10 300 python bench.py --steps 1 --warmup 1 --spp $spp --cpu-seconds 0 --no-calibrate 2> gpurun_out/ab_spp.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('  %.1f Msamples/s  %.1f ms/step  ' % (d['value']*1e-6, d['ms_per_step']) + '  '.join('%s %.1f' % kv for kv in d['kernel_ms_per_step'].items()))
" || exit 1
  grep pending gpurun_out/ab_spp.err | tail -n 4
done
