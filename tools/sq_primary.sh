#!/bin/bash
# Runs on the GPU box: SQ counters of the primary kernel for one library variant.  usage: tools/sq_primary.sh variant
REPO=${GRAFT_REPO_ROOT:-/root/repo}
V=$1
OUT=$REPO/gpurun_out/sqp_$V
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export RRT_HIP_LIB=$REPO/build_ab/librrt_hip_$V.so RRT_DEVICE_BVH_OVERRIDE=${BVH:-1}
ARGS="--steps 1 --warmup 0 --spp 256 --cpu-seconds 0 --no-calibrate --no-other-workloads"
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_INSTS_FLAT --output-format csv -d $OUT/p1 -- python3 $REPO/bench.py $ARGS > $OUT/b1.json || exit 1
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(sys.argv[1] + "/p*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "k_primary" in name or "k_extend" in name:
            out[name][r["Counter_Name"]] += float(r["Counter_Value"])
for k, c in sorted(out.items()):
    print(k, "  ".join("%s %.4g" % kv for kv in sorted(c.items())))
PY
