#!/bin/bash
# Runs on the GPU box: device-built hierarchies of library variants in build_ab/.  usage: tools/ab_bvh2.sh "<bench args>" variant...
ARGS=$1; shift
for v in "$@"; do
  echo "== $v"
  RRT_HIP_LIB=$PWD/build_ab/librrt_hip_$v.so RRT_DEVICE_BVH_OVERRIDE=1 timeout -k 10 600 python bench.py --steps 3 --warmup 1 --cpu-seconds 0 $ARGS 2> gpurun_out/ab_bvh.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('  %.1f Msamples/s  %.2f ms/step  upload %.3f s  nodes %d  ' % (d['value']*1e-6, d['ms_per_step'], d['config']['scene_upload_s'], d['config']['bvh_nodes']) + '  '.join('%s %.1f' % kv for kv in d['kernel_ms_per_step'].items()))
for k in d['traversal_kernels'] or []: print('   ', k['kernel'], 'nodes/ray %.2f tris/ray %.2f' % (k['nodes_per_ray'], k['tris_per_ray']))
" || { tail -n 5 gpurun_out/ab_bvh.err; exit 1; }
done
