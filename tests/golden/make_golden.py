"""Writes the golden vectors under tests/golden/ from the CPU oracle.

The reference (Rust) cannot be run here, so these vectors come from the oracle -- the
double-precision restatement in oracle/ -- and pin (a) the oracle against accidental change and
(b) the device against a fixed, committed answer.  Re-run only on purpose:

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import conftest  # noqa: E402
import scenes_util  # noqa: E402
from oracle import binding as ob  # noqa: E402

rr = conftest.rr


def rays_fixture(name, scene_desc, rs, n_primary, seed):
    orc = ob.Oracle(scene_desc)
    rays = scenes_util.ray_set(rr, orc, rs, n_primary, seed)
    prim, t, margin = orc.intersect(rays, with_margin=True)
    np.savez_compressed(os.path.join(HERE, name), rays=rays.astype(np.float32), prim=prim, t=t, margin=margin)
    print(name, rays.shape, "hits", (prim >= 0).sum(), "clear", (margin >= 1e-4).sum())


def film_fixture(name, scene_desc, rs, w, h, spp, depth, seed):
    orc = ob.Oracle(scene_desc)
    cam = rr.Camera.from_settings(rs, w, h)
    orc.set_camera(cam)
    s = rr.make_settings(w, h, depth, rs.clamp, seed=seed)
    film, _ = orc.render(s, 0, spp, threads=8)
    np.save(os.path.join(HERE, name), film)
    print(name, film.shape, "mean radiance", film[..., :3].sum() / film[..., 3].sum())


def main():
    folder = os.path.join(rr.SCENES, "cornell")
    cornell = rr.Scene.load_from_folder(folder)
    crs = rr.load_render_settings(folder)
    rays_fixture("cornell_rays.npz", cornell.desc(), crs, 1200, 1)
    film_fixture("cornell_film_48x48_4spp_seed3.npy", cornell.desc(), crs, 48, 48, 4, crs.depth_limit, 3)

    mixed, mrs, _ = scenes_util.mixed_scene(rr)
    rays_fixture("mixed_rays.npz", mixed.desc(), mrs, 1200, 2)
    film_fixture("mixed_film_80x60_4spp_seed4.npy", mixed.desc(), mrs, 80, 60, 4, mrs.depth_limit, 4)


if __name__ == "__main__":
    main()
