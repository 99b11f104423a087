#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/latency
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $REPO/tools/latency_probe.py > $OUT/log.txt 2>&1 || { tail -5 $OUT/log.txt; exit 1; }
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/*/*_kernel_trace.csv")[0]
for r in csv.DictReader(open(f)):
    if "k_intersect" in r["Kernel_Name"]:
        print("k_intersect grid %s  %.1f us" % (r["Grid_Size_X"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
PY
grep "hit fraction" $OUT/log.txt
