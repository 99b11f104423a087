"""CPU-only checks of the oracle itself: it reproduces the committed golden vectors, and its
renders have the properties any correct scheduler must have (independent of thread count, bucket
size and of how the sample range is cut)."""
import os

import numpy as np
import pytest

import scenes_util

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def mixed(rrt):
    return scenes_util.mixed_scene(rrt)


def test_oracle_reproduces_golden_rays(oracle_binding, cornell, mixed):
    for name, scene in (("cornell_rays.npz", cornell[0]), ("mixed_rays.npz", mixed[0])):
        g = np.load(os.path.join(GOLDEN, name))
        orc = oracle_binding.Oracle(scene.desc())
        prim, t, margin = orc.intersect(g["rays"].astype(np.float64), with_margin=True)
        np.testing.assert_array_equal(prim, g["prim"])
        np.testing.assert_array_equal(t, g["t"])
        np.testing.assert_array_equal(margin, g["margin"])
        plain_prim, plain_t = orc.intersect(g["rays"].astype(np.float64))
        np.testing.assert_array_equal(plain_prim, prim)
        np.testing.assert_array_equal(plain_t, t)


def test_oracle_reproduces_golden_films(rrt, oracle_binding, cornell, mixed):
    cases = (("cornell_film_48x48_4spp_seed3.npy", cornell[0], cornell[1], 48, 48, 3),
             ("mixed_film_80x60_4spp_seed4.npy", mixed[0], mixed[1], 80, 60, 4))
    for name, scene, rs, w, h, seed in cases:
        ref = np.load(os.path.join(GOLDEN, name))
        orc = oracle_binding.Oracle(scene.desc())
        orc.set_camera(rrt.Camera.from_settings(rs, w, h))
        s = rrt.make_settings(w, h, rs.depth_limit, rs.clamp, seed=seed)
        film, st = orc.render(s, 0, 4, threads=4)
        assert st.paths == w * h * 4
        # per-pixel sums are added in sample order whatever the thread count: exactly reproducible
        np.testing.assert_array_equal(film, ref)


def test_oracle_scheduler_independence(rrt, oracle_binding, cornell):
    scene, rs = cornell
    orc = oracle_binding.Oracle(scene.desc())
    orc.set_camera(rrt.Camera.from_settings(rs, 40, 24))
    s = rrt.make_settings(40, 24, 4, rs.clamp, seed=2, crop=(3, 2, 37, 23))
    a, _ = orc.render(s, 0, 6, threads=1, bucket=(128, 128))
    b, _ = orc.render(s, 0, 6, threads=5, bucket=(7, 5))
    np.testing.assert_array_equal(a, b)
    c, _ = orc.render(s, 0, 2, threads=3, bucket=(16, 16))
    c, _ = orc.render(s, 2, 6, threads=3, bucket=(16, 16), film=c)
    np.testing.assert_allclose(c, a, rtol=1e-12)
    assert (a[..., 3][2:23, 3:37] == 6).all()
    outside = np.ones((24, 40), bool)
    outside[2:23, 3:37] = False
    assert (a[outside] == 0).all()


def test_oracle_trace_one_matches_render(rrt, oracle_binding, cornell):
    scene, rs = cornell
    orc = oracle_binding.Oracle(scene.desc())
    orc.set_camera(rrt.Camera.from_settings(rs, 16, 16))
    s = rrt.make_settings(16, 16, rs.depth_limit, rs.clamp, seed=6)
    film, _ = orc.render(s, 3, 4, threads=2)
    for (x, y) in ((0, 0), (7, 9), (15, 15), (8, 13)):
        np.testing.assert_array_equal(orc.trace_one(s, x, y, 3), film[y, x, :3])


def test_clamp_bounds_the_running_sum(rrt, oracle_binding, mixed):
    """tracer.rs:116-120 clamps the running radiance; what a path adds after its last clamp is
    bounded by the light it can collect in one bounce, so no sample is far above the clamp."""
    scene, rs, _ = mixed
    orc = oracle_binding.Oracle(scene.desc())
    orc.set_camera(rrt.Camera.from_settings(rs, 40, 30))
    lo, _ = orc.render(rrt.make_settings(40, 30, 6, 0.25, seed=1), 0, 1, threads=4)
    hi, _ = orc.render(rrt.make_settings(40, 30, 6, 0.0, seed=1), 0, 1, threads=4)
    assert hi[..., :3].max() > 1.0          # unclamped paths do get bright
    assert lo[..., :3].sum() < hi[..., :3].sum()
    # depth 1 never reaches the clamp line of a second bounce: clamp only applies if the path continues
    d1a, _ = orc.render(rrt.make_settings(40, 30, 1, 0.25, seed=1), 0, 1, threads=4)
    d1b, _ = orc.render(rrt.make_settings(40, 30, 1, 1e9, seed=1), 0, 1, threads=4)
    assert (d1a[..., :3] <= d1b[..., :3] + 1e-12).all()


def _filter_table(method, radius):
    """Film::new's 16 x 16 table (src/film.rs:87-112, 326-358), restated in numpy."""
    c = (np.arange(16) + 0.5) * radius / 16
    if method == 1:  # Gaussian, alpha 1.5
        e = np.exp(-1.5 * radius * radius)
        g = np.maximum(np.exp(-1.5 * c * c) - e, 0.0)
        return g[:, None] * g[None, :]
    x = np.abs(2.0 * c / radius)
    b = cc = 0.33
    m = np.where(x > 1.0,
                 ((-b - 6 * cc) * x ** 3 + (6 * b + 30 * cc) * x ** 2 + (-12 * b - 48 * cc) * x + (8 * b + 24 * cc)) / 6,
                 ((12 - 9 * b - 6 * cc) * x ** 3 + (-18 + 12 * b + 6 * cc) * x ** 2 + (6 - 2 * b)) / 6)
    return m[:, None] * m[None, :]


@pytest.mark.parametrize("method", [1, 2])
def test_oracle_filter_splat(rrt, oracle_binding, cornell, method):
    """Film::write_bucket_pixels with a filter (src/film.rs:169-215): the film is the splat of the
    per-sample radiances, whatever the bucket size, and samples near the crop edge reach outside it."""
    scene, rs = cornell
    w, h, radius = 36, 28, 1.8
    orc = oracle_binding.Oracle(scene.desc())
    orc.set_camera(rrt.Camera.from_settings(rs, w, h))
    s = rrt.make_settings(w, h, 3, rs.clamp, seed=6, crop=(4, 3, 30, 25))
    plain, _ = orc.render(s, 0, 1, threads=2)  # filter none, 1 spp: the film holds each sample's radiance
    orc.set_filter(method, radius)
    film, _ = orc.render(s, 0, 1, threads=3, bucket=(8, 8))
    film2, _ = orc.render(s, 0, 1, threads=1, bucket=(128, 128))
    np.testing.assert_allclose(film, film2, rtol=1e-12, atol=1e-15)
    table = _filter_table(method, radius)
    L = oracle_binding.lib()
    expect = np.zeros_like(film)
    for y in range(3, 25):
        for x in range(4, 30):
            fx, fy = x + L.orc_rng(6, y * w + x, 0, 0, 0), y + L.orc_rng(6, y * w + x, 0, 0, 1)
            for py in range(int(np.ceil(fy - radius)), int(np.floor(fy + radius)) + 1):
                for px in range(int(np.ceil(fx - radius)), int(np.floor(fx + radius)) + 1):
                    if 0 <= px < w and 0 <= py < h:
                        ix = min(int(abs((px - fx) / radius * 16)), 15)
                        iy = min(int(abs((py - fy) / radius * 16)), 15)
                        wgt = table[iy, ix]
                        expect[py, px, :3] += plain[y, x, :3] * wgt
                        expect[py, px, 3] += wgt
    np.testing.assert_allclose(film, expect, rtol=1e-9, atol=1e-12)
    assert film[2, 10, 3] != 0.0 and film[0, 0, 3] == 0.0  # one row outside the crop is reached, the corner is not
    orc.set_filter(0)
    again, _ = orc.render(s, 0, 1, threads=2)
    np.testing.assert_array_equal(again, plain)


def test_oracle_first_hits_are_consistent(rrt, oracle_binding, cornell):
    """orc_first_hit: unit shading normals facing the camera ray's side of the surface or not (cornell's
    boxes are closed, seen from outside: facing), albedos out of the scene's material table, and a hit
    exactly where a depth-1 render finds something to shade."""
    scene, rs = cornell
    w, h = 24, 24
    cam = rrt.Camera.from_settings(rs, w, h)
    s = rrt.make_settings(w, h, 1, rs.clamp, seed=2)
    orc = oracle_binding.Oracle(scene.desc())
    orc.set_camera(cam)
    normal, albedo = orc.first_hit_sums(s, w, h, 0, 1)
    hit = normal[..., 3] == 1
    assert hit.mean() > 0.9  # the room fills the view
    np.testing.assert_allclose(np.linalg.norm(normal[hit][:, :3], axis=1), 1.0, atol=1e-9)
    assert (normal[~hit] == 0).all() and (albedo[~hit] == 0).all()
    d = scene.desc()
    table = {tuple(np.round(list(d.materials[i].kd), 6)) for i in range(d.n_materials)}
    seen = {tuple(np.round(a, 6)) for a in albedo[hit][:, :3]}
    assert seen <= table and len(seen) >= 3


def test_bucket_queue_keeps_every_worker_busy(rrt, oracle_binding, cornell):
    """render()'s workers pull 128 x 128 buckets from the film's queue (src/renderer.rs:104-160,
    src/film.rs:134-148): with many more buckets than workers every worker takes some, and the rate
    grows with the number of workers -- what the bench's CPU baseline relies on.  With two buckets for
    four workers (a frame smaller than 2 x 1 buckets) only two of them ever work."""
    scene, rs = cornell
    orc = oracle_binding.Oracle(scene.desc())
    w, h = 512, 384  # 4 x 3 buckets
    orc.set_camera(rrt.Camera.from_settings(rs, w, h))
    s = rrt.make_settings(w, h, 3, rs.clamp, seed=1)
    _, one = orc.render(s, 0, 1, threads=1)
    _, four = orc.render(s, 0, 1, threads=4)
    assert one.buckets == four.buckets == 12
    assert one.threads_busy == 1 and four.threads_busy == 4
    assert four.paths == one.paths == w * h
    if (os.cpu_count() or 1) >= 4:
        assert four.paths / four.seconds > 2.0 * one.paths / one.seconds
    orc.set_camera(rrt.Camera.from_settings(rs, 160, 90))
    _, small = orc.render(rrt.make_settings(160, 90, 3, rs.clamp, seed=1), 0, 1, threads=4)
    assert small.buckets == 2 and small.threads_busy <= 2
