#!/usr/bin/env python3
"""Runs on the GPU box: cost of a visibility ray when every path of the bench frame draws the SAME light, against the
scene's five lights drawn at random per path (an upper bound on what sorting the visibility rays by light would buy)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

rr = bench.rr
W, H, SPP = 1920, 1080, 64
scene, rs, camera, _ = bench.load_buddha(W, H)
d = scene.desc()
pos = np.ctypeslib.as_array(d.tri_pos, shape=(d.n_tris, 3, 3)).copy()
nrm = np.ctypeslib.as_array(d.tri_nrm, shape=(d.n_tris, 3, 3)).copy()
mat = np.ctypeslib.as_array(d.tri_mat, shape=(d.n_tris,)).copy()
env = np.ctypeslib.as_array(d.env_rgb8, shape=(d.env_h, d.env_w, 3)).copy()
materials = [d.materials[i] for i in range(d.n_materials)]
lights = [d.lights[i] for i in range(d.n_lights)]
rects = [d.rects[i] for i in range(d.n_rects)]


def variant(keep):
    ls = [lights[i] for i in keep]
    rs_ = []
    for r in rects:
        if r.light in keep:
            rs_.append(rr.make_rect(r.pos[:], r.side_a[:], r.side_b[:], r.material, keep.index(r.light)))
    has_env = any(lights[i].kind == rr.LIGHT_INFINITE for i in keep)
    return rr.SceneArrays(pos, nrm, mat, materials, ls, rs_, env_rgb8=env if has_env else None)


ctx = rr.Context(0)
ctx.set_camera(camera)
ctx.film_alloc(W, H)
ctx.set_option(rr.OPT_TIMING, 1)
s = rr.make_settings(W, H, rs.depth_limit, rs.clamp, seed=0)
for name, keep in (("all five", [0, 1, 2, 3, 4]), ("area 0", [0]), ("area 1", [1]), ("area 2", [2]), ("distant", [3]), ("three areas", [0, 1, 2])):
    ctx.upload_scene(variant(keep))
    ctx.render(s, 0, SPP)
    ctx.reset_stats()
    ctx.render(s, 0, SPP)
    st = ctx.stats()
    print("%-12s connect %6.2f ms for %6.1f M rays = %.3f ns/ray | extend %6.2f ms for %6.1f M = %.3f ns/ray | shade %5.2f ms for %5.1f M" % (
        name, st.ms_connect, st.rays_shadow * 1e-6, st.ms_connect * 1e6 / st.rays_shadow, st.ms_extend,
        (st.rays_extend - st.rays_primary) * 1e-6, st.ms_extend * 1e6 / max(st.rays_extend - st.rays_primary, 1), st.ms_shade,
        st.path_bounces * 1e-6))
