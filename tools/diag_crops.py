#!/usr/bin/env python3
"""Runs on the GPU box: time of the primary kernel per 240 x 135 crop of the bench frame, for the library given in
RRT_HIP_LIB and the hierarchy selected by RRT_DEVICE_BVH_OVERRIDE (diagnostic for hierarchies with slow regions)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

rr = bench.rr
W, H, SPP = 1920, 1080, 64
scene, rs, camera, _ = bench.load_buddha(W, H)
ctx = rr.Context(0)
if os.environ.get("RRT_DEVICE_BVH_OVERRIDE"):
    ctx.set_option(rr.OPT_DEVICE_BVH, int(os.environ["RRT_DEVICE_BVH_OVERRIDE"]))
ctx.upload_scene(scene)
ctx.set_camera(camera)
ctx.film_alloc(W, H)
ctx.set_option(rr.OPT_TIMING, 1)
for y0 in range(0, H, 135):
    row = []
    for x0 in range(0, W, 240):
        s = rr.make_settings(W, H, rs.depth_limit, rs.clamp, seed=0, crop=(x0, y0, x0 + 240, y0 + 135))
        ctx.render(s, 0, SPP)  # warm
        ctx.reset_stats()
        ctx.render(s, 0, SPP)
        row.append(ctx.stats().ms_primary)
    print(" ".join("%6.2f" % v for v in row))
