#!/bin/bash
# Runs on the GPU box: the bench frame under different pool shapes (all samples at once, pools that
# hold a quarter / a sixteenth of them, explicit slots per pixel).
for args in "--spp 256" "--spp 256 --max-slots 140000000" "--spp 256 --max-slots 36000000" "--spp 256 --batch-samples 64" "--spp 256 --max-slots 12000000"; do
  echo "== $args"
  timeout -k 10 300 python bench.py --steps 2 --warmup 1 --cpu-seconds 0 --no-calibrate $args | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('  %.1f Msamples/s  %.1f ms/step  ' % (d['value']*1e-6, d['ms_per_step']) + '  '.join('%s %.1f' % kv for kv in d['kernel_ms_per_step'].items()))
" || exit 1
done
