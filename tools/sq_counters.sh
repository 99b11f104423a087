#!/bin/bash
# Runs on the GPU box: issue / occupancy / cache counters of one bench step, one rocprofv3 pass per
# counter group, summed per kernel.  usage: tools/sq_counters.sh [tag]   (SPP=16 by default)
set -e
TAG=${1:-sq}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/sq_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 1 --warmup 0 --spp ${SPP:-16} --cpu-seconds 0 --no-calibrate --no-other-workloads"
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU --output-format csv -d $OUT/p1 -- python3 $REPO/bench.py $ARGS > $OUT/b1.json
timeout -s KILL 300 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_FLAT SQ_WAIT_ANY --output-format csv -d $OUT/p2 -- python3 $REPO/bench.py $ARGS > $OUT/b2.json
timeout -s KILL 300 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/p3 -- python3 $REPO/bench.py $ARGS > $OUT/b3.json
python3 - $OUT <<'PY'
import csv, glob, sys, collections, json
out = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(sys.argv[1] + "/p*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if name.startswith("rrt::"):
            out[name][r["Counter_Name"]] += float(r["Counter_Value"])
json.dump(out, open(sys.argv[1] + "/summary.json", "w"), indent=1, sort_keys=True)
for k, c in sorted(out.items()):
    print(k)
    for n, v in sorted(c.items()):
        print("   %-32s %.4g" % (n, v))
PY
