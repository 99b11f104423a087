#!/bin/bash
# Runs on the GPU box: kernel-trace statistics of one bench step, then the HBM traffic counters of
# the extend kernel in passes of their own.  Results land in gpurun_out/prof_<tag>/.
set -e
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 1 --warmup 1 --spp ${SPP:-256} --cpu-seconds 0 --no-calibrate --no-other-workloads"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/bench.py $ARGS > $OUT/bench_trace.json
timeout -s KILL 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $REPO/bench.py $ARGS > $OUT/bench_fetch.json
timeout -s KILL 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $REPO/bench.py $ARGS > $OUT/bench_write.json
timeout -s KILL 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_l2 -- python3 $REPO/bench.py $ARGS > $OUT/bench_l2.json
find $OUT -name "*.csv" | head -30
