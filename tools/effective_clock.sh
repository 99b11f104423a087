#!/bin/bash
# Runs on the GPU box: the shader clock each kernel of one bench frame actually ran at.
# GRBM_GUI_ACTIVE (cycles the graphics engine was busy, summed by rocprofv3 over the 8 XCDs) / 8 / the dispatch's
# wall time from the same CSV; meaningful for dispatches of a few milliseconds and more.
set -e
TAG=${1:-clk}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/clk_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/p -- python3 $REPO/bench.py --steps 2 --warmup 1 --spp ${SPP:-256} --cpu-seconds 0 --no-calibrate > $OUT/bench.json
python3 - $OUT <<'PY'
import csv, glob, sys, collections
rows = collections.defaultdict(lambda: [0.0, 0.0, 0])
for f in glob.glob(sys.argv[1] + "/p/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if not name.startswith("rrt::") or r["Counter_Name"] != "GRBM_GUI_ACTIVE":
            continue
        ns = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        if ns < 1.0e6:
            continue  # (short dispatches read high)
        e = rows[name]
        e[0] += float(r["Counter_Value"]); e[1] += ns; e[2] += 1
for name, (cycles, ns, n) in sorted(rows.items()):
    print("%-26s %3d dispatches over 1 ms, %8.2f ms in all, effective clock %.2f GHz" % (name, n, ns * 1e-6, cycles / 8.0 / ns))
PY
