"""Per-depth traversal statistics of the bench workload (counting build of the kernels)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

rr = bench.rr
w, h, spp = 1920, 1080, 4
scene, rs, cam, _ = bench.load_buddha(w, h)
ctx = rr.Context(0)
ctx.upload_scene(scene)
ctx.set_camera(cam)
ctx.film_alloc(w, h)
ctx.set_option(rr.OPT_COUNT_TRAVERSAL, 1)
prev = None
for depth in (1, 2, 3, 4, 6, 12):
    ctx.reset_stats()
    s = rr.make_settings(w, h, depth, rs.clamp, seed=0)
    ctx.render(s, 0, spp)
    st = ctx.stats()
    cur = dict(paths=st.paths, ext=st.rays_extend, sh=st.rays_shadow, b=st.path_bounces, nodes=st.nodes_visited, tris=st.tris_tested)
    line = "depth %2d: ext/path %.3f shadow/path %.3f bounces/path %.3f nodes/ray %.2f tris/ray %.2f" % (
        depth, st.rays_extend / st.paths, st.rays_shadow / st.paths, st.path_bounces / st.paths,
        st.nodes_visited / (st.rays_extend + st.rays_shadow), st.tris_tested / (st.rays_extend + st.rays_shadow))
    if prev:
        dr = (cur["ext"] + cur["sh"]) - (prev["ext"] + prev["sh"])
        line += " | added rays %.1f M: nodes/ray %.1f tris/ray %.1f" % (dr * 1e-6, (cur["nodes"] - prev["nodes"]) / dr, (cur["tris"] - prev["tris"]) / dr)
    print(line)
    prev = cur
