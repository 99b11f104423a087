#!/bin/bash
# Builds named variants of librrt_hip.so for A/B timing:  tools/ab_build.sh name "-DFLAG=1 ..." [name flags]...
set -e
cd "$(dirname "$0")/../rust-raytracer_amd/csrc"
mkdir -p ../../build_ab
while [ $# -ge 2 ]; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-hip-fp32-correctly-rounded-divide-sqrt -fgpu-flush-denormals-to-zero -fno-slp-vectorize -std=c++17 -fPIC -shared -pthread $2 -o ../../build_ab/librrt_hip_$1.so rrt_api.hip bvh_build.cpp bvh_build_device.hip -ldl &
  shift 2
done
wait
ls -la ../../build_ab/
