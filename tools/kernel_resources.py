#!/usr/bin/env python3
"""Compiles the HIP library with -Rpass-analysis=kernel-resource-usage and prints one line per kernel
(registers, spills, scratch, occupancy, LDS).  usage: tools/kernel_resources.py [extra hipcc flags]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "rust-raytracer_amd", "csrc")
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fno-hip-fp32-correctly-rounded-divide-sqrt",
       "-fgpu-flush-denormals-to-zero", "-fno-slp-vectorize", "-std=c++17", "-fPIC", "-shared", "-pthread", "-Rpass-analysis=kernel-resource-usage",
       *sys.argv[1:], "-o", "/tmp/rrt_resources.so", "rrt_api.hip", "bvh_build.cpp", "bvh_build_device.hip", "-ldl"]
out = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"remark: +(.*?) \[-Rpass", line)
    if not m:
        continue
    text = m.group(1).strip()
    if text.startswith("Function Name:"):
        cur = subprocess.run(["c++filt", text.split(":", 1)[1].strip()], capture_output=True, text=True).stdout.strip()
        cur = cur.split("(")[0].replace("void ", "").replace("rrt::", "")
        rows[cur] = {}
    elif cur and ":" in text:
        k, v = text.split(":", 1)
        rows[cur][k.strip()] = v.strip()
print("%-28s %6s %6s %7s %7s %8s %5s %7s" % ("kernel", "VGPR", "SGPR", "vspill", "sspill", "scratch", "occ", "LDS"))
for k, r in rows.items():
    print("%-28s %6s %6s %7s %7s %8s %5s %7s" % (k, r.get("VGPRs"), r.get("SGPRs"), r.get("VGPRs Spill"), r.get("SGPRs Spill"),
                                                 r.get("ScratchSize [bytes/lane]"), r.get("Occupancy [waves/SIMD]"),
                                                 r.get("LDS Size [bytes/block]")))
