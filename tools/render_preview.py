"""Renders a scene on GPU 0 and writes a tone-mapped PNG (for eyeballing): 
python tools/render_preview.py buddha 480 270 64 gpurun_out/buddha.png"""
import os
import sys
import time

import numpy as np
from PIL import Image

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402  (loads the package by path)

rr = bench.rr


def main():
    name, w, h, spp, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    if name == "buddha":
        scene, rs, cam, _ = bench.load_buddha(w, h)
    else:
        folder = os.path.join(rr.SCENES, name)
        scene = rr.Scene.load_from_folder(folder)
        rs = rr.load_render_settings(folder)
        cam = rr.Camera.from_settings(rs, w, h)
    ctx = rr.Context(0)
    t = time.time()
    ctx.upload_scene(scene)
    print("upload + BVH build %.2f s" % (time.time() - t))
    ctx.set_camera(cam)
    ctx.film_alloc(w, h)
    s = rr.make_settings(w, h, rs.depth_limit, rs.clamp, seed=0)
    t = time.time()
    ctx.render(s, 0, spp)
    film = ctx.film_download()
    dt = time.time() - t
    st = ctx.stats()
    print("%.3f s, %.2f M paths/s, %.1f M rays/s, %.2f bounces/path" % (
        dt, w * h * spp / dt * 1e-6, (st.rays_extend + st.rays_shadow) / dt * 1e-6, st.path_bounces / st.paths))
    Image.fromarray(rr.tonemap(film)).save(out)
    np.save(out + ".npy", film)


if __name__ == "__main__":
    main()
