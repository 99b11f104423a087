"""Condenses a tools/profile_bench.sh run (gpurun_out/prof_<tag>/) into files under profiles/:

    profiles/<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary, verbatim
    profiles/<tag>_pmc_summary.json   per-kernel sums of FETCH_SIZE / WRITE_SIZE / TCC_HIT / TCC_MISS
    profiles/extend_traffic.json      HBM bytes and memory-side requests of every kernel over one frame, read by bench.py

FETCH_SIZE and WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE tallies 128-byte requests at 64 bytes for
16-byte-per-lane loads (which is what every traversal load is), so it is doubled before use.
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys
import importlib.util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_spec = importlib.util.spec_from_file_location("rrt_build", os.path.join(ROOT, "rust-raytracer_amd", "build.py"))
_build = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_build)
kernel_fingerprint = _build.kernel_fingerprint


def main(tag):
    src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    stats = glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))[0]
    shutil.copy(stats, os.path.join(dst, tag + "_kernel_stats.csv"))
    launches = {}
    avg_ns = {}
    for r in csv.DictReader(open(stats)):
        name = r["Name"].split("(")[0].replace("void ", "")
        launches[name] = int(r["Calls"])
        avg_ns[name] = float(r["AverageNs"])
    summary = collections.defaultdict(dict)
    for sub in ("pmc_fetch", "pmc_write", "pmc_l2"):
        for f in glob.glob(os.path.join(src, sub, "*", "*_counter_collection.csv")):
            for r in csv.DictReader(open(f)):
                name = r["Kernel_Name"].split("(")[0].replace("void ", "")
                if not name.startswith("rrt::"):
                    continue
                summary[name][r["Counter_Name"]] = summary[name].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    out = {}
    for name, c in summary.items():
        n = launches.get(name, 0)
        e = {"launches": n, "avg_launch_us": avg_ns.get(name, 0.0) / 1e3, "counters": c}
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c and n:
            e["hbm_read_bytes_per_launch"] = 2.0 * c["FETCH_SIZE"] * 1024.0 / n
            e["hbm_write_bytes_per_launch"] = c["WRITE_SIZE"] * 1024.0 / n
            e["hbm_bytes_per_launch"] = e["hbm_read_bytes_per_launch"] + e["hbm_write_bytes_per_launch"]
            e["hbm_gb_per_s"] = e["hbm_bytes_per_launch"] / (avg_ns[name] * 1e-9) * 1e-9
        if "TCC_HIT_sum" in c:
            e["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
        out[name] = e
    with open(os.path.join(dst, tag + "_pmc_summary.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    ext = out.get("rrt::k_extend<false>")
    frames = launches.get("rrt::k_primary<false>", 0)  # one primary launch per frame of this workload
    if ext and "hbm_bytes_per_launch" in ext and frames:
        # every kernel of the frame, for the bench line's other roofline entries
        per_kernel = {}
        for name, e in out.items():
            if "hbm_bytes_per_launch" not in e:
                continue
            short = name.replace("rrt::", "").split("<")[0]
            k = per_kernel.setdefault(short, {"hbm_bytes_per_frame": 0.0, "launches_per_frame": 0.0, "us_per_frame": 0.0,
                                              "read_lines_per_frame": 0.0, "write_requests_per_frame": 0.0,
                                              "_hit": 0.0, "_miss": 0.0})
            k["hbm_bytes_per_frame"] += e["hbm_bytes_per_launch"] * e["launches"] / frames
            # memory-side requests of the L2s: one per 128-byte line read (FETCH_SIZE tallies it at 64 bytes, for
            # streams and for scattered 16-byte loads alike: tools/fetch_probe.hip), one per 64 bytes written
            k["read_lines_per_frame"] += e["counters"]["FETCH_SIZE"] * 1024.0 / 64.0 / frames
            k["write_requests_per_frame"] += e["counters"]["WRITE_SIZE"] * 1024.0 / 64.0 / frames
            k["launches_per_frame"] += e["launches"] / frames
            k["us_per_frame"] += e["avg_launch_us"] * e["launches"] / frames
            k["_hit"] += e["counters"].get("TCC_HIT_sum", 0.0)
            k["_miss"] += e["counters"].get("TCC_MISS_sum", 0.0)
        for k in per_kernel.values():
            hit, miss = k.pop("_hit"), k.pop("_miss")
            k["l2_hit_rate"] = hit / (hit + miss) if hit + miss > 0 else None
            k["hbm_gb_per_s"] = k["hbm_bytes_per_frame"] / (k["us_per_frame"] * 1e-6) * 1e-9 if k["us_per_frame"] > 0 else None
            k["requests_g_per_s"] = ((k["read_lines_per_frame"] + k["write_requests_per_frame"]) / (k["us_per_frame"] * 1e-6) * 1e-9
                                     if k["us_per_frame"] > 0 else None)
        with open(os.path.join(dst, "extend_traffic.json"), "w") as f:
            json.dump({"source": tag, "kernel": "k_extend", "workload": "buddha 1920x1080, 256 spp",
                       "hbm_bytes_per_frame": ext["hbm_bytes_per_launch"] * ext["launches"] / frames,
                       "launches_per_frame": ext["launches"] / frames, "avg_launch_us": ext["avg_launch_us"],
                       "l2_hit_rate": ext.get("l2_hit_rate"), "kernels": per_kernel,
                       "kernel_fingerprint": kernel_fingerprint()}, f, indent=1)
    for name, e in sorted(out.items()):
        print("%-28s launches %4d avg %9.1f us  HBM %8.1f GB/s  L2 hit %.3f" % (
            name, e["launches"], e["avg_launch_us"], e.get("hbm_gb_per_s", 0.0), e.get("l2_hit_rate", 0.0)))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r01")
