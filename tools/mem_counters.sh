#!/bin/bash
# Runs on the GPU box: vector-memory pipeline counters (address unit, L1, translation, latencies) of one
# bench step, one rocprofv3 pass per counter group, summed per kernel.  usage: tools/mem_counters.sh [tag]
set -e
TAG=${1:-mem}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/mem_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 1 --warmup 0 --spp ${SPP:-64} --cpu-seconds 0 --no-calibrate --no-other-workloads"
timeout -s KILL 300 rocprofv3 --pmc TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN1_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/p1 -- python3 $REPO/bench.py $ARGS > $OUT/b1.json
timeout -s KILL 300 rocprofv3 --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_TCP_LATENCY_sum TCP_TCC_READ_REQ_LATENCY_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TA_BUSY_sum TD_TD_BUSY_sum TD_TC_STALL_sum --output-format csv -d $OUT/p2 -- python3 $REPO/bench.py $ARGS > $OUT/b2.json
timeout -s KILL 300 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TA_TOTAL_WAVEFRONTS_sum SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $OUT/p3 -- python3 $REPO/bench.py $ARGS > $OUT/b3.json
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
        print("   %-40s %.4g" % (n, v))
PY
