#!/bin/bash
# Runs on the GPU box: shader / memory clock and power as rocm-smi reports them while the bench renders.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
timeout -k 10 200 python3 $REPO/bench.py --steps ${STEPS:-80} --warmup 2 --cpu-seconds 0 --no-calibrate > /tmp/bench_clocks.json 2>/dev/null &
PID=$!
sleep ${DELAY:-12}
for i in 1 2 3 4 5 6 7 8; do
  rocm-smi --showclocks --showpower --showuse 2>/dev/null | grep -E "sclk|mclk|Power|GPU use" | tr '\n' ' '
  echo
  sleep 0.4
done
wait $PID
python3 -c "
import json
d=json.loads(open('/tmp/bench_clocks.json').read().strip().splitlines()[-1]); print('bench: %.1f ms/step' % d['ms_per_step'])"
