#!/bin/bash
# Runs on the GPU box: host-built against device-built hierarchy on a bench workload.  usage: tools/ab_bvh.sh <bench args>
for m in 0 1; do
  echo "== device bvh $m"
  RRT_DEVICE_BVH_OVERRIDE=$m RRT_BVH_TIMING=1 timeout -k 10 600 python bench.py --steps 3 --warmup 1 --cpu-seconds 0 "$@" 2> gpurun_out/ab_bvh.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('  %.1f Msamples/s  %.2f ms/step  upload %.3f s  nodes %d  ' % (d['value']*1e-6, d['ms_per_step'], d['config']['scene_upload_s'], d['config']['bvh_nodes']) + '  '.join('%s %.1f' % kv for kv in d['kernel_ms_per_step'].items()))
for k in d['traversal_kernels'] or []: print('   ', k['kernel'], 'nodes/ray %.2f tris/ray %.2f' % (k['nodes_per_ray'], k['tris_per_ray']))
" || { tail -n 5 gpurun_out/ab_bvh.err; exit 1; }
  grep "bvh build" gpurun_out/ab_bvh.err
done
