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


def converged_fixture(name, scene_desc, rs, w, h, crop, spp, seeds):
    """High-spp oracle films of a crop window, one per seed: what the converged-image test compares the
    device with (same seed: same random numbers; the second seed gives the oracle's own noise level)."""
    orc = ob.Oracle(scene_desc)
    orc.set_camera(rr.Camera.from_settings(rs, w, h))
    x0, y0, x1, y1 = crop
    films = {}
    for seed in seeds:
        s = rr.make_settings(w, h, rs.depth_limit, rs.clamp, seed=seed, crop=crop)
        film, st = orc.render(s, 0, spp, threads=8, bucket=(16, 16))
        films["seed%d" % seed] = film[y0:y1, x0:x1].copy()
        print(name, "seed", seed, "%.0f paths/s" % (st.paths / st.seconds), "mean", film[..., :3].sum() / film[..., 3].sum())
    np.savez_compressed(os.path.join(HERE, name), width=w, height=h, crop=np.array(crop), spp=spp, depth=rs.depth_limit,
                        clamp=rs.clamp, seeds=np.array(seeds), **films)


def bench_mesh_fixture(name, n_primary, seed):
    """First hits and any-hit answers on the full-size bench mesh (the 1,087,716-triangle stand-in of
    bench.py, scenes/buddha lights and camera): the oracle takes its time to build over a million
    triangles, so its answers are committed."""
    scene, rs = scenes_util.buddha_scene(rr, 966, 563)
    assert scene.desc().n_tris == 1087716
    orc = ob.Oracle(scene.desc())
    rays = scenes_util.ray_set(rr, orc, rs, n_primary, seed)
    prim, t, margin = orc.intersect(rays, with_margin=True)
    rng = np.random.default_rng(seed)
    tmax = rng.uniform(0.02, 3.0, size=len(rays))
    odt = np.concatenate([rays, tmax[:, None]], axis=1).astype(np.float32).astype(np.float64)
    occluded = orc.occluded(odt)
    np.savez_compressed(os.path.join(HERE, name), rays=rays.astype(np.float32), prim=prim, t=t, margin=margin,
                        tmax=odt[:, 6].astype(np.float32), occluded=occluded)
    print(name, rays.shape, "hits", (prim >= 0).sum(), "clear", (margin >= 1e-4).sum(), "occluded", occluded.sum())


def main():
    folder = os.path.join(rr.SCENES, "cornell")
    cornell = rr.Scene.load_from_folder(folder)
    crs = rr.load_render_settings(folder)
    rays_fixture("cornell_rays.npz", cornell.desc(), crs, 1200, 1)
    film_fixture("cornell_film_48x48_4spp_seed3.npy", cornell.desc(), crs, 48, 48, 4, crs.depth_limit, 3)

    mixed, mrs, _ = scenes_util.mixed_scene(rr)
    rays_fixture("mixed_rays.npz", mixed.desc(), mrs, 1200, 2)
    film_fixture("mixed_film_80x60_4spp_seed4.npy", mixed.desc(), mrs, 80, 60, 4, mrs.depth_limit, 4)

    if "--large" in sys.argv:  # minutes of CPU: the converged crops and the full-size bench mesh
        converged_fixture("cornell_converged_64x64_1024spp.npz", cornell.desc(), crs, 256, 256, (96, 112, 160, 176), 1024, (31, 32))
        buddha, brs = scenes_util.buddha_scene(rr)
        converged_fixture("buddha68k_converged_64x64_1024spp.npz", buddha.desc(), brs, 480, 270, (216, 100, 280, 164), 1024, (41, 42))
        bench_mesh_fixture("bench_mesh_rays.npz", 3000, 7)


if __name__ == "__main__":
    main()
