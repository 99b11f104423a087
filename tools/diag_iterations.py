#!/usr/bin/env python3
"""Runs on the GPU box: continuation rays, node visits and extend time of the bench frame by depth limit, so that
the differences between successive limits give the cost of each bounce's rays (diagnostic)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

rr = bench.rr
W, H, SPP = 1920, 1080, int(os.environ.get("SPP", "64"))
scene, rs, camera, _ = bench.load_buddha(W, H)
ctx = rr.Context(0)
ctx.upload_scene(scene)
ctx.set_camera(camera)
ctx.film_alloc(W, H)
ctx.set_option(rr.OPT_TAIL_PATHS, 0)
prev = None
print("depth  rays(M)  nodes/ray  tris/ray  extend_ms | this bounce: rays(M) nodes/ray tris/ray ms  Grays/s")
for depth in range(1, 9):
    s = rr.make_settings(W, H, depth, rs.clamp, seed=0)
    ctx.set_option(rr.OPT_COUNT_TRAVERSAL, 1)
    ctx.reset_stats()
    ctx.render(s, 0, SPP)
    st = ctx.stats()
    rays = st.rays_extend - st.rays_primary
    nodes, tris = st.nodes_by_kernel[1], st.tris_by_kernel[1]
    ctx.set_option(rr.OPT_COUNT_TRAVERSAL, 0)
    ctx.set_option(rr.OPT_TIMING, 1)
    ctx.set_option(rr.OPT_SIDE_STREAM, 0)
    ctx.render(s, 0, SPP)
    ctx.reset_stats()
    ctx.render(s, 0, SPP)
    ms = ctx.stats().ms_extend
    ctx.set_option(rr.OPT_TIMING, 0)
    cur = (rays, nodes, tris, ms)
    line = "%5d %8.2f %9.2f %8.2f %9.3f" % (depth, rays / 1e6, nodes / max(rays, 1), tris / max(rays, 1), ms)
    if prev:
        dr, dn, dt, dm = (cur[i] - prev[i] for i in range(4))
        if dr > 0:
            line += " | %8.2f %9.2f %8.2f %7.3f %7.2f" % (dr / 1e6, dn / dr, dt / dr, dm, dr / max(dm, 1e-9) / 1e6)
    print(line, flush=True)
    prev = cur
