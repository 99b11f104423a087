#!/bin/bash
# Runs on the GPU box: per-dispatch kernel durations of one short bench run -> gpurun_out/trace_<tag>.txt
TAG=${1:-x}; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/trace_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $REPO/bench.py --steps 1 --warmup 0 --spp ${SPP:-16} --cpu-seconds 0 --no-calibrate "$@" > $OUT/bench.json || exit 1
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/*/*_kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if "rrt::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
prev_end = t0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("rrt::", "")
    print("%-18s start %9.1f us  dur %8.1f us  gap %6.1f us  grid %s" % (name, (s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, r.get("Grid_Size_X", "")))
    prev_end = e
PY
