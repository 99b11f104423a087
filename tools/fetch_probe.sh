#!/bin/bash
# Runs on the GPU box: tools/fetch_probe.hip (built beforehand as build_ab/fetch_probe) under rocprofv3, one
# counter pass per mode; prints FETCH_SIZE (KiB) and the L2's request counters per dispatch next to the line count.
set -e
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/fetch_probe
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for size in 25 19; do
  for mode in 0 1 2 3 4; do
    timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/m${mode}_s${size} -- $REPO/build_ab/fetch_probe $mode $size > $OUT/m${mode}_s${size}.txt
    timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $OUT/m${mode}_s${size}_l2 -- $REPO/build_ab/fetch_probe $mode $size > /dev/null
  done
done
python3 - $OUT <<'PY'
import csv, glob, sys, collections
for size in (25, 19):
    for mode in range(5):
        d = "%s/m%d_s%d" % (sys.argv[1], mode, size)
        print(open(d + ".txt").read().strip())
        per = collections.defaultdict(list)
        for f in glob.glob(d + "/*/*_counter_collection.csv") + glob.glob(d + "_l2/*/*_counter_collection.csv"):
            for r in csv.DictReader(open(f)):
                if "gather" in r["Kernel_Name"]:
                    per[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in sorted(per.items()):
            print("    %-14s per dispatch: %s" % (k, ", ".join("%.4g" % x for x in v)))
PY
