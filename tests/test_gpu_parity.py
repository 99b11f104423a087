"""Device vs oracle on the same inputs, through the C ABI (include/rrt.h).

Stated tolerances (the integrator is floating point; the oracle is f64, the device f32):
* first hit: object id identical and |t_dev - t_ref| <= 1e-5 * t_ref (1e-5 relative, whatever the
  length of the ray: the device decides the object in f32 and then computes the reported distance
  of that one object in f64, see exact_triangle_t) on every ray whose oracle answer is well
  separated (margin >= 1e-4: not within 1e-4 of a triangle edge in barycentric units, not grazing,
  no second candidate within 1e-4 relative distance);
* per-path radiance with identical random numbers: at least 99.8 % of camera paths agree to 1e-3
  relative (+1e-4 absolute); the rest took a different branch at a float/double tie;
* images: summed absolute film difference <= 1 % of the summed film (cornell), 2 % (mixed scene).
"""
import os

import numpy as np
import pytest

import scenes_util

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def mixed(rrt):
    return scenes_util.mixed_scene(rrt)


def _check_first_hit(ctx, rays, ref_prim, ref_t, margin, record=None, name=None, min_clear=0.95):
    dev_prim, dev_t = ctx.intersect(rays)
    clear = margin >= 1e-4
    assert clear.mean() > min_clear
    assert (ref_prim[clear] >= 0).sum() > 500
    np.testing.assert_array_equal(dev_prim[clear], ref_prim[clear])
    hit = clear & (ref_prim >= 0)
    err = np.abs(dev_t[hit] - ref_t[hit])
    rel = err / ref_t[hit]
    print("t: max rel %.3g, 99.9%% rel %.3g, max abs %.3g; %d rays above 1e-5 relative (smallest t among them %.3g)" % (
        rel.max(), np.quantile(rel, 0.999), err.max(), (rel > 1e-5).sum(),
        ref_t[hit][rel > 1e-5].min() if (rel > 1e-5).any() else 0.0))
    worst = np.argmax(rel)
    print("worst ray: object %d, t %.6g, error %.3g" % (ref_prim[hit][worst], ref_t[hit][worst], err[worst]))
    assert (err <= 1e-5 * ref_t[hit]).all(), rel.max()
    aside_agree = float((dev_prim[~clear] == ref_prim[~clear]).mean()) if (~clear).any() else 1.0
    # Rays set aside as ambiguous (within 1e-4 of an edge, grazing, or with a second candidate within 1e-4 of
    # the first) may legitimately pick the neighbouring triangle, but hit-or-miss must still agree on nearly
    # all of them and most ids do: a traversal bug would show here first.
    if (~clear).any():
        same_kind = ((dev_prim[~clear] >= 0) == (ref_prim[~clear] >= 0)).mean()
        print("set aside: %d rays, id agreement on them %.3f, hit/miss agreement %.3f" % ((~clear).sum(), aside_agree, same_kind))
        assert same_kind >= 0.9, same_kind
        if (~clear).sum() >= 20:
            assert aside_agree >= 0.5, aside_agree
    if record is not None:
        record(name, rays=int(len(ref_prim)), clear_rays=int(clear.sum()), set_aside=int((~clear).sum()),
               set_aside_id_agreement=aside_agree, id_agreement_clear=1.0, t_max_rel_err=float(rel.max()),
               t_rays_above_1e5_relative=int((rel > 1e-5).sum()))


def test_first_hit_cornell_golden(cornell, gpu_ctx, parity_record):
    """Fixed ray set with committed oracle answers."""
    g = np.load(os.path.join(GOLDEN, "cornell_rays.npz"))
    gpu_ctx.upload_scene(cornell[0])
    _check_first_hit(gpu_ctx, g["rays"], g["prim"], g["t"], g["margin"], parity_record, "first_hit_cornell_golden")


def test_first_hit_mixed_golden(mixed, gpu_ctx, parity_record):
    """Triangles, quad, plane and spheres in one scene."""
    g = np.load(os.path.join(GOLDEN, "mixed_rays.npz"))
    gpu_ctx.upload_scene(mixed[0])
    _check_first_hit(gpu_ctx, g["rays"], g["prim"], g["t"], g["margin"], parity_record, "first_hit_mixed_golden")
    kinds = set(np.digitize(g["prim"][g["prim"] >= 0], [994, 995, 996]))  # triangles | quad | plane | spheres
    # the light quad lies in the plane y = 5: its box is flat, and under the strict box test (the
    # default reading of the reference's pre-filter) no ray ever finds it
    assert kinds == {0, 2, 3}


def test_first_hit_cornell_live(rrt, oracle_binding, cornell, gpu_ctx, parity_record):
    scene, rs = cornell
    orc = oracle_binding.Oracle(scene.desc())
    gpu_ctx.upload_scene(scene)
    rays = scenes_util.ray_set(rrt, orc, rs, 4000, 17)
    ref_prim, ref_t, margin = orc.intersect(rays, with_margin=True)
    _check_first_hit(gpu_ctx, rays, ref_prim, ref_t, margin, parity_record, "first_hit_cornell_live")


@pytest.mark.parametrize("which", ["cornell", "mixed"])
def test_occluded(rrt, oracle_binding, cornell, mixed, gpu_ctx, which):
    scene, rs = cornell if which == "cornell" else mixed[:2]
    orc = oracle_binding.Oracle(scene.desc())
    gpu_ctx.upload_scene(scene)
    rng = np.random.default_rng(11)
    primary = scenes_util.camera_rays(rrt, rs, 256, 256, 3000, 2)
    sec = scenes_util.bounce_rays(orc, primary, rng).astype(np.float32).astype(np.float64)
    ref_prim, ref_t, margin = orc.intersect(sec, with_margin=True)
    tmax = rng.uniform(0.05, 3.0, size=len(sec))
    rays = np.concatenate([sec, tmax[:, None]], axis=1).astype(np.float32).astype(np.float64)
    ref = orc.occluded(rays)
    dev = gpu_ctx.occluded(rays)
    # leave out rays whose closest hit sits within 1e-4 (relative) of tmax or is not well separated
    near = (ref_prim >= 0) & (np.abs(ref_t - rays[:, 6]) <= 1e-4 * rays[:, 6])
    ok = (margin >= 1e-4) & ~near
    np.testing.assert_array_equal(dev[ok], ref[ok])
    assert ref[ok].sum() > 100 and (ref[ok] == 0).sum() > 100


@pytest.fixture(scope="module")
def buddha(rrt):
    return scenes_util.buddha_scene(rrt)


@pytest.fixture
def builder(request, rrt, gpu_ctx):
    """RRT_OPT_DEVICE_BVH for the uploads of one test: 0 = the host's binned-SAH build, 1 = the build by kernels."""
    gpu_ctx.set_option(rrt.OPT_DEVICE_BVH, request.param)
    yield request.param
    gpu_ctx.set_option(rrt.OPT_DEVICE_BVH, 2)


@pytest.mark.parametrize("builder", [0, 1], indirect=True)
def test_first_hit_and_occlusion_deep_tree(rrt, oracle_binding, buddha, gpu_ctx, parity_record, builder):
    """68 k triangles: traversal stacks spill past their LDS part, leaves of up to 4 triangles,
    rays leaving the surface in all directions.  Against the oracle with the hierarchy of either builder."""
    scene, rs = buddha
    assert scene.desc().n_tris > 60000
    orc = oracle_binding.Oracle(scene.desc())
    gpu_ctx.upload_scene(scene)
    d = scene.desc()
    assert gpu_ctx.bvh_check(np.ctypeslib.as_array(d.tri_pos, shape=(d.n_tris, 9)))[3] == bool(builder)
    rays = scenes_util.ray_set(rrt, orc, rs, 6000, 23)
    ref_prim, ref_t, margin = orc.intersect(rays, with_margin=True)
    _check_first_hit(gpu_ctx, rays, ref_prim, ref_t, margin, parity_record, "first_hit_buddha68k_live" + ("_device_bvh" if builder else ""))
    rng = np.random.default_rng(5)
    tmax = rng.uniform(0.02, 2.0, size=len(rays))
    odt = np.concatenate([rays, tmax[:, None]], axis=1).astype(np.float32).astype(np.float64)
    ref = orc.occluded(odt)
    dev = gpu_ctx.occluded(odt)
    near = (ref_prim >= 0) & (np.abs(ref_t - odt[:, 6]) <= 1e-4 * odt[:, 6])
    ok = (margin >= 1e-4) & ~near
    np.testing.assert_array_equal(dev[ok], ref[ok])
    assert ref[ok].sum() > 100 and (ref[ok] == 0).sum() > 100


@pytest.mark.parametrize("which", ["cornell", "buddha68k"])
def test_rays_around_the_root_box(rrt, oracle_binding, cornell, buddha, gpu_ctx, parity_record, which):
    """Rays that start outside the box around all triangles -- the ones the traversal engine tests against the
    root's box before they set out: aimed at random points of a box 1.3 times as large (a good part misses
    the inner box altogether), at the inner box's corners and edge midpoints, and skimming its faces.  Same
    answers as the oracle, which tests every triangle's own box."""
    scene, _ = cornell if which == "cornell" else buddha
    d = scene.desc()
    pos = np.ctypeslib.as_array(d.tri_pos, shape=(d.n_tris, 3, 3)).astype(np.float64)
    lo, hi = pos.reshape(-1, 3).min(axis=0), pos.reshape(-1, 3).max(axis=0)
    mid, half = 0.5 * (lo + hi), 0.5 * (hi - lo)
    rng = np.random.default_rng(41)
    n = 6000
    dirs = rng.normal(size=(n, 3))
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    org = mid + dirs * (3.0 * np.linalg.norm(half))                      # on a sphere well outside
    target = mid + rng.uniform(-1.3, 1.3, size=(n, 3)) * half
    corners = mid + half * np.array([[sx, sy, sz] for sx in (-1, 0, 1) for sy in (-1, 0, 1) for sz in (-1, 0, 1)], dtype=np.float64)
    k = rng.integers(0, len(corners), size=n // 4)
    target[:n // 4] = corners[k] + rng.normal(scale=1e-6, size=(n // 4, 3)) * np.linalg.norm(half)
    # along a face: origin and target both in the plane of one face of the box
    m = n // 8
    axis = rng.integers(0, 3, size=m)
    side = rng.integers(0, 2, size=m)
    face = np.where(side == 1, hi[axis], lo[axis])
    # (not exactly in it: a direction component of exactly zero on a box's boundary plane is a 0 * inf in the
    # reference's box pre-filter, which the oracle restates and the device's clamped reciprocal does not)
    org[-m:][np.arange(m), axis] = face + rng.normal(scale=1e-4, size=m) * half[axis]
    target[-m:][np.arange(m), axis] = face + rng.normal(scale=1e-4, size=m) * half[axis]
    dd = target - org
    dd /= np.linalg.norm(dd, axis=1, keepdims=True)
    rays = np.concatenate([org, dd], axis=1).astype(np.float32).astype(np.float64)
    orc = oracle_binding.Oracle(d)
    gpu_ctx.upload_scene(scene)
    ref_prim, ref_t, margin = orc.intersect(rays, with_margin=True)
    assert (ref_prim < 0).mean() > 0.1   # the set does hold rays that miss everything
    # The oracle gives a miss the margin 1: it has no candidate to be close to.  A ray aimed at the mesh's own outline
    # to within 1e-6 that misses in f64 may graze in f32 (seen: u = -2e-8), so those count as ambiguous here.
    margin = margin.copy()
    margin[:n // 4][ref_prim[:n // 4] < 0] = 0.0
    # (a quarter of these rays is aimed at edges and corners on purpose: more of them are ambiguous than elsewhere)
    _check_first_hit(gpu_ctx, rays, ref_prim, ref_t, margin, parity_record, "root_box_rays_" + which, min_clear=0.7)
    tmax = rng.uniform(0.5, 6.0, size=n) * np.linalg.norm(half)
    odt = np.concatenate([rays, tmax[:, None]], axis=1).astype(np.float32).astype(np.float64)
    ref = orc.occluded(odt)
    dev = gpu_ctx.occluded(odt)
    near = (ref_prim >= 0) & (np.abs(ref_t - odt[:, 6]) <= 1e-4 * odt[:, 6])
    ok = (margin >= 1e-4) & ~near
    np.testing.assert_array_equal(dev[ok], ref[ok])


@pytest.mark.parametrize("depth", [2, 12])
def test_per_path_radiance_buddha(rrt, oracle_binding, buddha, gpu_ctx, depth, parity_record):
    """The bench workload's scene at small size: environment light with MIS escape rays, distant
    light, three area lights, thin lens, microfacet plastic; 2 samples per pixel so that every slot
    starts a second path after its first."""
    scene, rs = buddha
    w, h, spp = 160, 90, 2
    cam = rrt.Camera.from_settings(rs, w, h)
    s = rrt.make_settings(w, h, depth, rs.clamp, seed=9)
    gpu_ctx.upload_scene(scene)
    gpu_ctx.set_camera(cam)
    gpu_ctx.film_alloc(w, h)
    orc = oracle_binding.Oracle(scene.desc())
    orc.set_camera(cam)
    agree = []
    for smp in range(spp):
        gpu_ctx.film_clear()
        gpu_ctx.render(s, smp, smp + 1)
        dev = gpu_ctx.film_download().astype(np.float64)
        ref, _ = orc.render(s, smp, smp + 1, threads=8)
        agree.append(_path_agreement(dev, ref).mean())
    # and both samples in one call (each slot renders one path after the other)
    gpu_ctx.set_option(rrt.OPT_BATCH_SAMPLES, 1)
    gpu_ctx.film_clear()
    gpu_ctx.render(s, 0, spp)
    both = gpu_ctx.film_download().astype(np.float64)
    gpu_ctx.set_option(rrt.OPT_BATCH_SAMPLES, 0)
    ref_both, st = orc.render(s, 0, spp, threads=8)
    np.testing.assert_array_equal(both[..., 3], ref_both[..., 3])
    rel = np.abs(both[..., :3] - ref_both[..., :3]).sum() / ref_both[..., :3].sum()
    print("depth %d: per-path agreement %s; two-sample film rel diff %.3g; surface hits per path %.2f" % (
        depth, ["%.4f" % a for a in agree], rel, st.path_bounces / st.paths))
    parity_record("per_path_buddha68k_depth%d" % depth, path_agreement_min=min(agree), two_sample_film_rel_diff=rel)
    assert min(agree) >= 0.998
    assert rel <= 1e-3


@pytest.mark.parametrize("builder", [0, 1], indirect=True)
def test_stress_scene(rrt, oracle_binding, gpu_ctx, parity_record, builder):
    """24 cow / lamp / teapot trios (375 k triangles, generated normals, many separate objects),
    depth limit 16: first hits, and per-path radiance with the distant and the area light.  Against the
    oracle with the hierarchy of either builder."""
    _, (scene, cam_args, depth, clamp) = scenes_util.stress_scene(rrt, 24, grid=(4, 3, 2))
    assert scene.desc().n_tris == 24 * 15632 + 2
    w, h = 128, 72
    cam = rrt.Camera(image_width=w, image_height=h, **cam_args)
    orc = oracle_binding.Oracle(scene.desc())
    orc.set_camera(cam)
    gpu_ctx.upload_scene(scene)
    gpu_ctx.set_camera(cam)
    rng = np.random.default_rng(3)
    px = rng.uniform(0, [w, h], size=(3000, 2))
    primary = np.array([np.concatenate(cam.generate_ray(x, y)) for x, y in px])
    sec = scenes_util.bounce_rays(orc, primary, rng)
    rays = np.concatenate([primary, sec]).astype(np.float32).astype(np.float64)
    ref_prim, ref_t, margin = orc.intersect(rays, with_margin=True)
    suffix = "_device_bvh" if builder else ""
    _check_first_hit(gpu_ctx, rays, ref_prim, ref_t, margin, parity_record, "first_hit_stress24" + suffix)
    s = rrt.make_settings(w, h, depth, clamp, seed=4)
    gpu_ctx.film_alloc(w, h)
    gpu_ctx.render(s, 0, 1)
    dev = gpu_ctx.film_download().astype(np.float64)
    ref, st = orc.render(s, 0, 1, threads=8)
    close = _path_agreement(dev, ref)
    print("stress scene: %.4f of paths agree; %.2f surface hits per path; mean ref %.5f dev %.5f" % (
        close.mean(), st.path_bounces / st.paths, ref[..., :3].mean(), dev[..., :3].mean()))
    parity_record("per_path_stress24" + suffix, path_agreement=close.mean(), hits_per_path=st.path_bounces / st.paths)
    assert st.path_bounces / st.paths > 0.5
    assert close.mean() >= 0.998


@pytest.mark.parametrize("builder", [0, 1], indirect=True)
def test_first_hit_far_from_the_origin(rrt, oracle_binding, gpu_ctx, parity_record, builder):
    """A scene thousands of units from the origin with a wide range of sizes: 60 k triangles of 0.05 units in a
    cluster 20 units across, inside 2 k triangles of 100 units spread over 3000 units.  The node corners lie on a
    16-bit grid of the whole scene (a step of 0.06 units here, larger than the small triangles) and the ray's
    distances to the grid are differences of large numbers: ids must still be the oracle's, distances within 1e-5,
    and the hierarchy of either builder must pass the invariant check."""
    rng = np.random.default_rng(21)
    shift = np.array([4000.0, -2500.0, 7000.0])
    small = rng.uniform(-10, 10, size=(60000, 1, 3)) + rng.normal(scale=0.05, size=(60000, 3, 3))
    large = rng.uniform(-1500, 1500, size=(2000, 1, 3)) + rng.normal(scale=100.0, size=(2000, 3, 3))
    pos = (np.concatenate([small, large]) + shift).astype(np.float32)
    n = len(pos)
    nrm = np.cross(pos[:, 1] - pos[:, 0], pos[:, 2] - pos[:, 0]).astype(np.float64)
    nrm /= np.maximum(np.linalg.norm(nrm, axis=1, keepdims=True), 1e-30)
    scene = rrt.SceneArrays(pos, np.repeat(nrm[:, None, :], 3, axis=1).astype(np.float32), np.zeros(n),
                            materials=[rrt.make_material(rrt.MAT_MATTE, kd=(0.5, 0.5, 0.5))])
    gpu_ctx.upload_scene(scene)
    bad, nodes, depth, on_device, _ = gpu_ctx.bvh_check(pos.reshape(-1, 9))
    assert bad == 0 and on_device == bool(builder)
    # half of the rays start inside the small cluster and aim at it, half cross the large one
    o_in = rng.uniform(-12, 12, size=(3000, 3))
    d_in = rng.uniform(-10, 10, size=(3000, 3)) - o_in
    o_out = rng.uniform(-1600, 1600, size=(3000, 3))
    d_out = rng.normal(size=(3000, 3))
    org = np.concatenate([o_in, o_out]) + shift
    d = np.concatenate([d_in, d_out])
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.concatenate([org, d], axis=1).astype(np.float32).astype(np.float64)
    orc = oracle_binding.Oracle(scene.desc())
    ref_prim, ref_t, margin = orc.intersect(rays, with_margin=True)
    assert (ref_prim[:3000] >= 0).sum() > 300 and (ref_prim[3000:] >= 0).sum() > 300
    _check_first_hit(gpu_ctx, rays, ref_prim, ref_t, margin, parity_record,
                     "first_hit_far_from_origin" + ("_device_bvh" if builder else ""), min_clear=0.9)


def _device_film(rrt, ctx, scene, rs, w, h, spp, depth, seed, crop=None):
    cam = rrt.Camera.from_settings(rs, w, h)
    s = rrt.make_settings(w, h, depth, rs.clamp, seed=seed, crop=crop)
    ctx.upload_scene(scene)
    ctx.set_camera(cam)
    ctx.film_alloc(w, h)
    ctx.render(s, 0, spp)
    return ctx.film_download().astype(np.float64), cam, s


def _path_agreement(dev, ref):
    a, b = dev[..., :3], ref[..., :3]
    return np.all(np.abs(a - b) <= 1e-3 * np.abs(b) + 1e-4, axis=-1)


@pytest.mark.parametrize("depth", [1, 2, 8])
def test_per_path_radiance_cornell(rrt, oracle_binding, cornell, gpu_ctx, depth, parity_record):
    """One sample per pixel: the film holds each path's radiance."""
    scene, rs = cornell
    dev, cam, s = _device_film(rrt, gpu_ctx, scene, rs, 96, 96, 1, depth, 3)
    orc = oracle_binding.Oracle(scene.desc())
    orc.set_camera(cam)
    ref, _ = orc.render(s, 0, 1, threads=8)
    np.testing.assert_array_equal(dev[..., 3], ref[..., 3])
    close = _path_agreement(dev, ref)
    print("depth %d: %.4f of paths agree; mean ref %.5f dev %.5f" % (depth, close.mean(), ref[..., :3].mean(), dev[..., :3].mean()))
    parity_record("per_path_cornell_depth%d" % depth, path_agreement=close.mean())
    assert close.mean() >= 0.999
    assert ref[..., :3].mean() > 0


@pytest.mark.parametrize("depth", [1, 3, 6])
def test_per_path_radiance_mixed(rrt, oracle_binding, mixed, gpu_ctx, depth, parity_record):
    """All shape, material and light kinds, thin-lens camera."""
    scene, rs, _ = mixed
    dev, cam, s = _device_film(rrt, gpu_ctx, scene, rs, 120, 90, 1, depth, 5)
    orc = oracle_binding.Oracle(scene.desc())
    orc.set_camera(cam)
    ref, _ = orc.render(s, 0, 1, threads=8)
    np.testing.assert_array_equal(dev[..., 3], ref[..., 3])
    close = _path_agreement(dev, ref)
    print("depth %d: %.4f of paths agree; mean ref %.5f dev %.5f" % (depth, close.mean(), ref[..., :3].mean(), dev[..., :3].mean()))
    parity_record("per_path_mixed_depth%d" % depth, path_agreement=close.mean())
    assert close.mean() >= 0.998


def test_first_hit_buffers(rrt, oracle_binding, mixed, gpu_ctx):
    """RRT_OPT_AOVS: per-pixel sums of the shading normal and the albedo of what the camera rays hit,
    against the oracle's first hits of the same samples (every shape and material kind, thin lens).
    Off by default; cleared with the film; a second call adds to them."""
    scene, rs, _ = mixed
    w, h, spp = 60, 45, 4
    cam = rrt.Camera.from_settings(rs, w, h)
    s = rrt.make_settings(w, h, 3, rs.clamp, seed=5)
    gpu_ctx.upload_scene(scene)
    gpu_ctx.set_camera(cam)
    gpu_ctx.film_alloc(w, h)
    with pytest.raises(rrt.RrtError):
        gpu_ctx.aov_download()
    orc = oracle_binding.Oracle(scene.desc())
    orc.set_camera(cam)
    ref_n, ref_a = orc.first_hit_sums(s, w, h, 0, spp)
    try:
        gpu_ctx.set_option(rrt.OPT_AOVS, 1)
        gpu_ctx.render(s, 0, 1)
        gpu_ctx.render(s, 1, spp)
        film = gpu_ctx.film_download()
        dev_n, dev_a = gpu_ctx.aov_download()
        gpu_ctx.film_clear()
        zero_n, zero_a = gpu_ctx.aov_download()
    finally:
        gpu_ctx.set_option(rrt.OPT_AOVS, 0)
    assert (zero_n == 0).all() and (zero_a == 0).all()
    assert (film[..., 3] == spp).all()
    same_hits = dev_n[..., 3] == ref_n[..., 3]
    print("hit counts agree on %.4f of the pixels; %.3f of the camera rays hit" % (same_hits.mean(), ref_n[..., 3].sum() / (w * h * spp)))
    assert same_hits.mean() >= 0.995 and ref_n[..., 3].sum() > 0.3 * w * h * spp
    ok = same_hits
    np.testing.assert_allclose(dev_a[ok][:, :3], ref_a[ok][:, :3], rtol=1e-5, atol=1e-6)
    # normals: a ray that hits an edge between two faces may take either; all but a few pixels agree closely
    close = np.abs(dev_n[ok][:, :3] - ref_n[ok][:, :3]).max(axis=1) <= 2e-3 * spp
    print("normal sums agree on %.4f of those" % close.mean())
    assert close.mean() >= 0.99
    assert len(set(map(tuple, np.round(ref_a[ref_n[..., 3] == spp][:, :3] / spp, 3)))) >= 4  # several materials in view


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_rect_aabb_modes(rrt, oracle_binding, gpu_ctx, mode):
    """The three readings of the reference's box pre-filter for quads agree between device and oracle."""
    scene = rrt.Scene.load_from_folder(os.path.join(rrt.SCENES, "cornell"))
    rs = rrt.load_render_settings(os.path.join(rrt.SCENES, "cornell"))
    scene.set_rect_aabb_mode(mode)
    dev, cam, s = _device_film(rrt, gpu_ctx, scene, rs, 64, 64, 1, 4, 8)
    orc = oracle_binding.Oracle(scene.desc())
    orc.set_camera(cam)
    ref, _ = orc.render(s, 0, 1, threads=8)
    assert _path_agreement(dev, ref).mean() >= 0.98


def test_film_golden_cornell(rrt, cornell, gpu_ctx):
    ref = np.load(os.path.join(GOLDEN, "cornell_film_48x48_4spp_seed3.npy"))
    scene, rs = cornell
    dev, _, _ = _device_film(rrt, gpu_ctx, scene, rs, 48, 48, 4, rs.depth_limit, 3)
    np.testing.assert_array_equal(dev[..., 3], ref[..., 3])
    rel = np.abs(dev[..., :3] - ref[..., :3]).sum() / np.abs(ref[..., :3]).sum()
    print("relative film difference %.5f" % rel)
    assert rel <= 0.01


def test_film_golden_mixed(rrt, mixed, gpu_ctx):
    ref = np.load(os.path.join(GOLDEN, "mixed_film_80x60_4spp_seed4.npy"))
    scene, rs, _ = mixed
    dev, _, _ = _device_film(rrt, gpu_ctx, scene, rs, 80, 60, 4, rs.depth_limit, 4)
    np.testing.assert_array_equal(dev[..., 3], ref[..., 3])
    rel = np.abs(dev[..., :3] - ref[..., :3]).sum() / np.abs(ref[..., :3]).sum()
    print("relative film difference %.5f" % rel)
    assert rel <= 0.02


def test_image_cornell_live(rrt, oracle_binding, cornell, gpu_ctx):
    scene, rs = cornell
    dev, cam, s = _device_film(rrt, gpu_ctx, scene, rs, 64, 64, 16, rs.depth_limit, 3)
    orc = oracle_binding.Oracle(scene.desc())
    orc.set_camera(cam)
    ref, _ = orc.render(s, 0, 16, threads=8)
    np.testing.assert_array_equal(dev[..., 3], ref[..., 3])
    rel = np.abs(dev[..., :3] - ref[..., :3]).sum() / np.abs(ref[..., :3]).sum()
    print("relative film difference %.5f" % rel)
    assert rel <= 0.01


@pytest.mark.parametrize("method,radius", [(1, 1.8), (2, 2.0), (1, 0.7)])
def test_reconstruction_filters(rrt, oracle_binding, mixed, gpu_ctx, method, radius):
    """Gaussian / Mitchell splat on the device (src/film.rs:169-215) against the oracle's, with a crop
    window so that samples reach outside it, few slots so that the window takes several passes, and
    several samples per pixel.  The device adds with float atomics: same terms, different order."""
    scene, rs, _ = mixed
    w, h, spp = 80, 60, 3
    cam = rrt.Camera.from_settings(rs, w, h)
    s = rrt.make_settings(w, h, 4, rs.clamp, seed=8, crop=(5, 4, 70, 57))
    gpu_ctx.upload_scene(scene)
    gpu_ctx.set_camera(cam)
    gpu_ctx.film_alloc(w, h)
    gpu_ctx.set_option(rrt.OPT_MAX_SLOTS, 64 * 40)
    gpu_ctx.set_filter(method, radius)
    try:
        gpu_ctx.render(s, 0, spp)
        dev = gpu_ctx.film_download().astype(np.float64)
    finally:
        gpu_ctx.set_filter(rrt.FILTER_NONE)
        gpu_ctx.set_option(rrt.OPT_MAX_SLOTS, 0)
    orc = oracle_binding.Oracle(scene.desc())
    orc.set_camera(cam)
    orc.set_filter(method, radius)
    ref, _ = orc.render(s, 0, spp, threads=8)
    # weights do not depend on the paths: they must agree to float rounding everywhere
    np.testing.assert_allclose(dev[..., 3], ref[..., 3], rtol=2e-6, atol=1e-6)
    outside = np.ones((h, w), bool)
    outside[4:57, 5:70] = False
    assert (ref[..., 3][outside] != 0).any()
    rel = np.abs(dev[..., :3] - ref[..., :3]).sum() / np.abs(ref[..., :3]).sum()
    print("filter %d radius %.1f: film rel diff %.3g" % (method, radius, rel))
    assert rel <= 2e-3
    # and the plain film still comes out afterwards
    gpu_ctx.film_clear()
    gpu_ctx.render(s, 0, 1)
    plain = gpu_ctx.film_download()
    assert set(np.unique(plain[..., 3])) == {0.0, 1.0}


def test_device_tonemap(rrt, cornell, gpu_ctx):
    """rrt_film_tonemap (src/film.rs:233-249) against the host tone map of the same film: f32 against
    f64 may differ by one code value where the product lands on an integer."""
    scene, rs = cornell
    dev, cam, s = _device_film(rrt, gpu_ctx, scene, rs, 64, 64, 8, 4, 1, crop=(0, 0, 64, 60))
    img = gpu_ctx.film_tonemap()
    ref = rrt.tonemap(gpu_ctx.film_download())
    assert img.shape == (64, 64, 3) and img.dtype == np.uint8
    d = np.abs(img.astype(int) - ref.astype(int))
    assert d.max() <= 1 and (d != 0).mean() < 0.01
    assert (img[60:] == 0).all() and img[:60].max() > 50  # unrendered rows have no weight: black


def test_crop_and_sample_ranges(rrt, cornell, gpu_ctx):
    """Sample ranges compose: [0,3) + [3,8) == [0,8); a crop only touches its window; batching
    (samples in flight, tile ranges) does not change the result."""
    scene, rs = cornell
    w = h = 64
    cam = rrt.Camera.from_settings(rs, w, h)
    gpu_ctx.upload_scene(scene)
    gpu_ctx.set_camera(cam)
    gpu_ctx.film_alloc(w, h)
    s = rrt.make_settings(w, h, 4, rs.clamp, seed=5, crop=(8, 16, 40, 56))
    gpu_ctx.render(s, 0, 8)
    whole = gpu_ctx.film_download()
    gpu_ctx.film_clear()
    gpu_ctx.render(s, 0, 3)
    gpu_ctx.render(s, 3, 8)
    parts = gpu_ctx.film_download()
    np.testing.assert_allclose(parts, whole, rtol=1e-5, atol=1e-6)
    mask = np.zeros((h, w), bool)
    mask[16:56, 8:40] = True
    assert (whole[..., 3][mask] == 8).all() and (whole[~mask] == 0).all()
    try:
        gpu_ctx.set_option(rrt.OPT_MAX_SLOTS, 64 * 5)  # forces tile ranges and one sample per batch
        gpu_ctx.film_clear()
        gpu_ctx.render(s, 0, 8)
        small = gpu_ctx.film_download()
    finally:
        gpu_ctx.set_option(rrt.OPT_MAX_SLOTS, 0)
    np.testing.assert_allclose(small, whole, rtol=1e-5, atol=1e-6)


def test_full_size_schedule_independence(rrt, buddha, gpu_ctx):
    """The bench frame size (1920 x 1080) on the buddha scene: the film must not depend on how the
    samples are scheduled -- all 24 in flight at once, 4 slots per pixel that each work through 6
    samples (paths regenerated in place), or a pool so small that the window takes several passes --
    and sample ranges must compose.  Differences are float summation order only."""
    scene, rs = buddha
    w, h, spp = 1920, 1080, 24
    cam = rrt.Camera.from_settings(rs, w, h)
    s = rrt.make_settings(w, h, rs.depth_limit, rs.clamp, seed=11)
    gpu_ctx.upload_scene(scene)
    gpu_ctx.set_camera(cam)
    gpu_ctx.film_alloc(w, h)
    gpu_ctx.reset_stats()
    gpu_ctx.render(s, 0, spp)
    whole = gpu_ctx.film_download().astype(np.float64)
    st = gpu_ctx.stats()
    assert st.paths == w * h * spp and (whole[..., 3] == spp).all()
    assert np.isfinite(whole).all() and whole[..., :3].min() >= 0 and whole[..., :3].max() <= spp * rs.clamp * 1.0001
    films = {}
    try:
        gpu_ctx.set_option(rrt.OPT_BATCH_SAMPLES, 4)
        gpu_ctx.film_clear()
        gpu_ctx.render(s, 0, spp)
        films["regenerated"] = gpu_ctx.film_download().astype(np.float64)
        gpu_ctx.set_option(rrt.OPT_BATCH_SAMPLES, 0)
        gpu_ctx.set_option(rrt.OPT_MAX_SLOTS, 3 << 20)  # 1080p has 2.07 M pixels: one slot per pixel, 24 samples each
        gpu_ctx.film_clear()
        gpu_ctx.render(s, 0, spp)
        films["small pool"] = gpu_ctx.film_download().astype(np.float64)
        gpu_ctx.set_option(rrt.OPT_MAX_SLOTS, 1 << 20)  # fewer slots than pixels: tile ranges
        gpu_ctx.film_clear()
        gpu_ctx.render(s, 0, 5)
        gpu_ctx.render(s, 5, spp)
        films["tile ranges, two calls"] = gpu_ctx.film_download().astype(np.float64)
    finally:
        gpu_ctx.set_option(rrt.OPT_BATCH_SAMPLES, 0)
        gpu_ctx.set_option(rrt.OPT_MAX_SLOTS, 0)
    for name, f in films.items():
        np.testing.assert_array_equal(f[..., 3], whole[..., 3])
        rel = np.abs(f[..., :3] - whole[..., :3]).max() / whole[..., :3].max()
        print("%s: max difference %.3g of the film's maximum" % (name, rel))
        assert rel <= 1e-5, name


def test_rounds_of_one_sample_per_slot(rrt, cornell, gpu_ctx):
    """A pool that holds 20 of a pixel's 64 samples: the library renders four rounds of 16 samples, one
    per slot (not slots that work through several samples).  Same paths, same film up to summation order;
    two calls over adjacent sample ranges compose to the same film as well."""
    scene, rs = cornell
    w, h, spp = 160, 120, 64
    cam = rrt.Camera.from_settings(rs, w, h)
    s = rrt.make_settings(w, h, rs.depth_limit, rs.clamp, seed=5)
    gpu_ctx.upload_scene(scene)
    gpu_ctx.set_camera(cam)
    gpu_ctx.film_alloc(w, h)
    gpu_ctx.reset_stats()
    gpu_ctx.render(s, 0, spp)
    whole = gpu_ctx.film_download().astype(np.float64)
    assert gpu_ctx.stats().paths == w * h * spp
    try:
        gpu_ctx.set_option(rrt.OPT_MAX_SLOTS, w * h * 20)
        gpu_ctx.set_option(rrt.OPT_TIMING, 1)  # launches are counted from the timing events
        gpu_ctx.film_clear()
        gpu_ctx.reset_stats()
        gpu_ctx.render(s, 0, spp)
        gpu_ctx.set_option(rrt.OPT_TIMING, 0)
        rounds = gpu_ctx.film_download().astype(np.float64)
        st = gpu_ctx.stats()
        assert st.paths == w * h * spp and st.launches_primary == 4, (st.paths, st.launches_primary)
        gpu_ctx.film_clear()
        gpu_ctx.render(s, 0, 23)
        gpu_ctx.render(s, 23, spp)
        two_calls = gpu_ctx.film_download().astype(np.float64)
    finally:
        gpu_ctx.set_option(rrt.OPT_MAX_SLOTS, 0)
        gpu_ctx.set_option(rrt.OPT_TIMING, 0)
    for name, f in (("rounds", rounds), ("two calls", two_calls)):
        np.testing.assert_array_equal(f[..., 3], whole[..., 3])
        rel = np.abs(f[..., :3] - whole[..., :3]).max() / whole[..., :3].max()
        print("%s: max difference %.3g of the film's maximum" % (name, rel))
        assert rel <= 1e-5, name


def test_side_stream_changes_nothing(rrt, buddha, gpu_ctx):
    """Visibility rays on a stream of their own beside the next iteration's closest-hit rays (the default)
    against everything on one stream: the kernels touch disjoint records, so the films are identical."""
    scene, rs = buddha
    w, h, spp = 480, 270, 16
    cam = rrt.Camera.from_settings(rs, w, h)
    s = rrt.make_settings(w, h, rs.depth_limit, rs.clamp, seed=3)
    gpu_ctx.upload_scene(scene)
    gpu_ctx.set_camera(cam)
    gpu_ctx.film_alloc(w, h)
    films = []
    try:
        for side in (1, 0, 1):
            gpu_ctx.set_option(rrt.OPT_SIDE_STREAM, side)
            gpu_ctx.film_clear()
            gpu_ctx.render(s, 0, spp)
            films.append(gpu_ctx.film_download())
    finally:
        gpu_ctx.set_option(rrt.OPT_SIDE_STREAM, 1)
    assert films[0][..., :3].max() > 0
    np.testing.assert_array_equal(films[0], films[1])
    np.testing.assert_array_equal(films[0], films[2])


def test_cpp_host_render(rrt, cornell, gpu_ctx, tmp_path):
    """The C++ host (Scene::load_from_folder, RenderSettingsFile, Camera, render() with its worker
    thread and ThreadMessage channel, Film::merge_accumulators, tone map, PPM / PFM writers) drives
    the same device library: scenes/cornell as shipped (1000 x 1000, depth 8), 2 samples, against
    the film rendered through the ctypes layer with the same seed."""
    import ctypes as C
    scene, rs = cornell
    prefix = str(tmp_path / "cornell")
    rate = C.c_double()
    dev = (C.c_int * 1)(0)
    rc = rrt.host_lib().rrth_render_folder(os.fsencode(os.path.join(rrt.SCENES, "cornell")), dev, 1, 7, 2,
                                           os.fsencode(prefix), C.byref(rate))
    assert rc == 0, rrt.host_lib().rrth_last_error().decode()
    assert rate.value > 1e6
    w, h = rs.image_width, rs.image_height
    with open(prefix + ".ppm", "rb") as f:
        assert f.readline() == b"P6\n" and f.readline() == b"%d %d\n" % (w, h) and f.readline() == b"255\n"
        ppm = np.frombuffer(f.read(), dtype=np.uint8).reshape(h, w, 3)
    from test_host_and_abi import _read_png
    np.testing.assert_array_equal(_read_png(prefix + ".png"), ppm)
    with open(prefix + ".pfm", "rb") as f:
        assert f.readline() == b"PF\n" and f.readline() == b"%d %d\n" % (w, h) and f.readline() == b"-1.0\n"
        pfm = np.frombuffer(f.read(), dtype="<f4").reshape(h, w, 3)[::-1]
    film, cam, s = _device_film(rrt, gpu_ctx, scene, rs, w, h, 2, rs.depth_limit, 7)
    mean = film[..., :3] / film[..., 3:4]
    np.testing.assert_allclose(pfm, mean, rtol=1e-5, atol=1e-6)
    ref = rrt.tonemap(film.astype(np.float32))
    d = np.abs(ppm.astype(int) - ref.astype(int))
    assert d.max() <= 1 and (d != 0).mean() < 1e-3
    assert ppm.mean() > 10


def test_cpp_host_first_hit_maps(rrt, cornell, gpu_ctx):
    """The C++ host with a film that keeps first hits: the two maps denoise() would hand to OIDN
    (src/denoise.rs:9-23), against the device buffers through the ctypes layer with the same seed."""
    import ctypes as C
    scene, rs = cornell
    w, h, spp = rs.image_width, rs.image_height, 2
    nmap = np.zeros((h, w, 3), dtype=np.float32)
    amap = np.zeros((h, w, 3), dtype=np.float32)
    dev = (C.c_int * 1)(0)
    rc = rrt.host_lib().rrth_render_folder_first_hits(os.fsencode(os.path.join(rrt.SCENES, "cornell")), dev, 1, 7, spp,
                                                      None, None, nmap.ctypes.data, amap.ctypes.data)
    assert rc == 0, rrt.host_lib().rrth_last_error().decode()
    cam = rrt.Camera.from_settings(rs, w, h)
    s = rrt.make_settings(w, h, rs.depth_limit, rs.clamp, seed=7)
    gpu_ctx.upload_scene(scene)
    gpu_ctx.set_camera(cam)
    gpu_ctx.film_alloc(w, h)
    try:
        gpu_ctx.set_option(rrt.OPT_AOVS, 1)
        gpu_ctx.render(s, 0, spp)
        n, a = gpu_ctx.aov_download()
    finally:
        gpu_ctx.set_option(rrt.OPT_AOVS, 0)
    np.testing.assert_allclose(nmap, n[..., :3] / spp, atol=1e-6)
    np.testing.assert_allclose(amap, a[..., :3] / spp, atol=1e-6)
    lengths = np.linalg.norm(nmap, axis=2)
    assert (lengths <= 1.0 + 1e-5).all() and (lengths > 0.99).mean() > 0.8  # mostly one face per pixel
    assert amap.max() <= 1.0 and amap.mean() > 0.1


def test_empty_and_degenerate_inputs(rrt, cornell, gpu_ctx):
    scene, rs = cornell
    # a scene with no triangles at all: only a quad light
    d = scene.desc()
    empty = rrt.SceneArrays(materials=[d.materials[9]], lights=[d.lights[0]],
                            rects=[rrt.make_rect(d.rects[0].pos[:], d.rects[0].side_a[:], d.rects[0].side_b[:], 0, 0)],
                            rect_aabb_mode=rrt.RECT_AABB_OFF)
    gpu_ctx.upload_scene(empty)
    prim, t = gpu_ctx.intersect(np.array([[0, 0, 0, 0, 1, 0], [0, 0, 0, 0, -1, 0]], dtype=np.float32))
    assert prim.tolist() == [0, -1] and t[0] == pytest.approx(0.99)
    # nothing at all
    gpu_ctx.upload_scene(rrt.SceneArrays())
    prim, _ = gpu_ctx.intersect(np.array([[0, 0, 0, 0, 1, 0]], dtype=np.float32))
    assert prim.tolist() == [-1]
    cam = rrt.Camera.from_settings(rs, 16, 16)
    gpu_ctx.set_camera(cam)
    gpu_ctx.film_alloc(16, 16)
    s = rrt.make_settings(16, 16, 4, 10.0)
    gpu_ctx.render(s, 0, 2)          # every path escapes into nothing
    film = gpu_ctx.film_download()
    assert (film[..., 3] == 2).all() and (film[..., :3] == 0).all()
    gpu_ctx.render(s, 5, 5)          # empty sample range: no-op
    s0 = rrt.make_settings(16, 16, 4, 10.0, crop=(4, 4, 4, 9))
    gpu_ctx.render(s0, 0, 1)         # empty window: no-op
    np.testing.assert_array_equal(gpu_ctx.film_download(), film)
    # zero rays
    prim, t = gpu_ctx.intersect(np.zeros((0, 6), dtype=np.float32))
    assert len(prim) == 0
    # bad arguments are reported, not executed
    bad = rrt.make_settings(16, 16, 4, 10.0, crop=(0, 0, 17, 16))
    with pytest.raises(rrt.RrtError):
        gpu_ctx.render(bad, 0, 1)
    with pytest.raises(rrt.RrtError):
        gpu_ctx.render(rrt.make_settings(8, 8, 4, 10.0), 0, 1)  # does not match the film


def test_stats_count_what_happened(rrt, cornell, gpu_ctx):
    scene, rs = cornell
    gpu_ctx.reset_stats()
    _device_film(rrt, gpu_ctx, scene, rs, 32, 32, 2, 3, 1)
    st = gpu_ctx.stats()
    assert st.paths == 32 * 32 * 2
    assert st.paths <= st.rays_extend <= 3 * st.paths
    assert 0 < st.path_bounces <= st.rays_extend
    assert 0 < st.rays_shadow <= st.path_bounces
    assert st.bvh_nodes > 4014 // 12 and st.bvh_bytes > 4014 * 96


def test_film_reduce_single_process(rrt, cornell, gpu_ctx):
    """rrt_film_reduce (one context per GPU in one process, RCCL): disjoint sample ranges on two GPUs sum to
    the film of the whole range, first-hit buffers included, twice in a row (the second call reuses the
    cached communicators).  With one GPU only the argument checks and the one-context case can run."""
    scene, rs = cornell
    w = h = 64
    cam = rrt.Camera.from_settings(rs, w, h)
    s = rrt.make_settings(w, h, 4, rs.clamp, seed=21)
    gpu_ctx.upload_scene(scene)
    gpu_ctx.set_camera(cam)
    gpu_ctx.film_alloc(w, h)
    gpu_ctx.render(s, 0, 6)
    whole = gpu_ctx.film_download()
    rrt.film_reduce([gpu_ctx], 0)  # one context: nothing to do
    np.testing.assert_array_equal(gpu_ctx.film_download(), whole)
    with pytest.raises(rrt.RrtError):
        rrt.film_reduce([gpu_ctx, gpu_ctx], 0)  # two ranks on one device
    with pytest.raises(rrt.RrtError):
        rrt.film_reduce([gpu_ctx], 1)
    if rrt.hip_lib().rrt_device_count() < 2:
        pytest.skip("one GPU: the two-rank reduce cannot run here")
    other = rrt.Context(1)
    try:
        other.upload_scene(scene)
        other.set_camera(cam)
        other.film_alloc(w, h)
        for ctx in (gpu_ctx, other):
            ctx.set_option(rrt.OPT_AOVS, 1)
        for rounds in range(2):
            gpu_ctx.film_clear()
            other.film_clear()
            gpu_ctx.render(s, 0, 3)
            other.render(s, 3, 6)
            a, b = gpu_ctx.film_download(), other.film_download()
            (an, aa), (bn, ba) = gpu_ctx.aov_download(), other.aov_download()
            rrt.film_reduce([gpu_ctx, other], 0)
            np.testing.assert_allclose(gpu_ctx.film_download(), a + b, rtol=1e-6, atol=1e-7)
            np.testing.assert_allclose(gpu_ctx.film_download(), whole, rtol=1e-5, atol=1e-6)
            rn, ra = gpu_ctx.aov_download()
            np.testing.assert_allclose(rn, an + bn, rtol=1e-6, atol=1e-7)
            np.testing.assert_allclose(ra, aa + ba, rtol=1e-6, atol=1e-7)
    finally:
        gpu_ctx.set_option(rrt.OPT_AOVS, 0)
        other.close()


def test_film_reduce_runs_rccl_on_one_gpu(rrt, cornell):
    """The RCCL path of rrt_film_reduce on a machine with a single GPU: a context created with
    RRT_DEBUG_REDUCE_SINGLE_THROUGH_RCCL in the environment sends its one-context reduce through the library
    loading, the communicator cache and ncclReduce (a one-rank communicator), film and first-hit buffers; the
    buffers must come back unchanged, twice (the second call reuses the communicator), and destroying the
    context must tear the communicator down without trouble.  The library exports no function for this."""
    assert not hasattr(rrt.hip_lib(), "rrt_debug_reduce_single_through_rccl")
    scene, rs = cornell
    w = h = 64
    os.environ["RRT_DEBUG_REDUCE_SINGLE_THROUGH_RCCL"] = "1"
    try:
        ctx = rrt.Context(0)
    finally:
        del os.environ["RRT_DEBUG_REDUCE_SINGLE_THROUGH_RCCL"]
    try:
        ctx.upload_scene(scene)
        ctx.set_camera(rrt.Camera.from_settings(rs, w, h))
        ctx.film_alloc(w, h)
        ctx.set_option(rrt.OPT_AOVS, 1)
        ctx.render(rrt.make_settings(w, h, 4, rs.clamp, seed=3), 0, 4)
        film = ctx.film_download()
        normal, albedo = ctx.aov_download()
        assert film[..., 3].min() == 4.0 and np.abs(film[..., :3]).sum() > 0
        for _ in range(2):
            rrt.film_reduce([ctx], 0)
            np.testing.assert_array_equal(ctx.film_download(), film)
            n2, a2 = ctx.aov_download()
            np.testing.assert_array_equal(n2, normal)
            np.testing.assert_array_equal(a2, albedo)
    finally:
        ctx.close()


def test_first_hit_and_occlusion_bench_mesh_golden(rrt, gpu_ctx, parity_record):
    """The full-size bench mesh (1,087,716 triangles, what bench.py renders): first hits and any-hit answers
    against the oracle's, committed as a golden file (tests/golden/make_golden.py --large)."""
    g = np.load(os.path.join(GOLDEN, "bench_mesh_rays.npz"))
    scene, _ = scenes_util.buddha_scene(rrt, 966, 563)
    assert scene.desc().n_tris == 1087716
    gpu_ctx.upload_scene(scene)
    rays = g["rays"].astype(np.float64)
    _check_first_hit_small(gpu_ctx, rays, g["prim"], g["t"], g["margin"], parity_record, "first_hit_bench_mesh_golden")
    odt = np.concatenate([rays, g["tmax"][:, None].astype(np.float64)], axis=1)
    dev = gpu_ctx.occluded(odt)
    near = (g["prim"] >= 0) & (np.abs(g["t"] - odt[:, 6]) <= 1e-4 * odt[:, 6])
    ok = (g["margin"] >= 1e-4) & ~near
    np.testing.assert_array_equal(dev[ok], g["occluded"][ok])
    assert g["occluded"][ok].sum() > 100 and (g["occluded"][ok] == 0).sum() > 100
    parity_record("occluded_bench_mesh_golden", rays=int(ok.sum()), agreement=float((dev[ok] == g["occluded"][ok]).mean()))


def _check_first_hit_small(ctx, rays, ref_prim, ref_t, margin, record, name):
    """_check_first_hit for ray sets in which most camera rays miss (the bench frame): fewer hits are required."""
    dev_prim, dev_t = ctx.intersect(rays)
    clear = margin >= 1e-4
    assert clear.mean() > 0.95 and (ref_prim[clear] >= 0).sum() > 500
    np.testing.assert_array_equal(dev_prim[clear], ref_prim[clear])
    hit = clear & (ref_prim >= 0)
    err = np.abs(dev_t[hit] - ref_t[hit])
    rel = err / ref_t[hit]
    assert (err <= 1e-5 * ref_t[hit]).all(), rel.max()
    record(name, rays=int(len(rays)), clear_rays=int(clear.sum()), hits=int(hit.sum()), id_agreement_clear=1.0,
           t_max_rel_err=float(rel.max()), t_rays_above_1e5_relative=int((rel > 1e-5).sum()))


def _rel_mse(a, b, eps=1e-2):
    """Per-pixel relative mean squared error of image a against b (linear radiance, all channels)."""
    return float(np.mean((a - b) ** 2 / (b * b + eps)))


def _box4_rmse_8bit(rrt, film_a, film_b):
    """RMSE, in units of one 8-bit code value / 255, between the tone-mapped images after a 4 x 4 box filter."""
    a = rrt.tonemap(film_a.astype(np.float32)).astype(np.float64) / 255.0
    b = rrt.tonemap(film_b.astype(np.float32)).astype(np.float64) / 255.0
    h, w = a.shape[:2]
    a = a.reshape(h // 4, 4, w // 4, 4, 3).mean(axis=(1, 3))
    b = b.reshape(h // 4, 4, w // 4, 4, 3).mean(axis=(1, 3))
    return float(np.sqrt(np.mean((a - b) ** 2)))


@pytest.mark.parametrize("which", ["cornell", "buddha68k"])
def test_converged_image(rrt, cornell, buddha, gpu_ctx, parity_record, which):
    """Converged images at 1024 samples per pixel on a 64 x 64 crop (cornell: the sphere and pillars; buddha:
    the statue's body), device against oracle films committed as golden files (two seeds each).

    Stated bounds, on the linear per-pixel mean radiance L = sum / weight:
    * same seed (the device draws the same random numbers as the oracle): relMSE(dev, orc) with eps = 1e-2
      at most 2 % of the oracle's own seed-to-seed relMSE -- the few paths that take another branch at
      an f32/f64 tie (0.05-0.2 %) change the image far less than the Monte Carlo noise does -- and the
      8-bit tone-mapped images after a 4 x 4 box filter differ by an RMSE of at most 0.25/255;
    * another seed (independent noise): relMSE(dev, orc) <= 2 x the oracle's seed-to-seed relMSE, box-filtered
      8-bit RMSE <= 2/255, and no bias: the crop's total radiance agrees with the oracle's within 4 standard
      deviations of the oracle's own seed-to-seed scatter (estimated per pixel from its two films)."""
    g = np.load(os.path.join(GOLDEN, "%s_converged_64x64_1024spp.npz" % which))
    scene, rs = cornell if which == "cornell" else buddha
    w, h, spp = int(g["width"]), int(g["height"]), int(g["spp"])
    x0, y0, x1, y1 = [int(v) for v in g["crop"]]
    seed_a, seed_b = [int(v) for v in g["seeds"]]
    assert int(g["depth"]) == rs.depth_limit and float(g["clamp"]) == rs.clamp
    orc_a, orc_b = g["seed%d" % seed_a], g["seed%d" % seed_b]
    assert (orc_a[..., 3] == spp).all() and (orc_b[..., 3] == spp).all()
    cam = rrt.Camera.from_settings(rs, w, h)
    gpu_ctx.upload_scene(scene)
    gpu_ctx.set_camera(cam)
    gpu_ctx.film_alloc(w, h)
    films = {}
    for seed in (seed_a, seed_b + 1000):
        s = rrt.make_settings(w, h, rs.depth_limit, rs.clamp, seed=seed, crop=(x0, y0, x1, y1))
        gpu_ctx.film_clear()
        gpu_ctx.render(s, 0, spp)
        film = gpu_ctx.film_download().astype(np.float64)[y0:y1, x0:x1]
        assert (film[..., 3] == spp).all()
        films[seed] = film
    mean = lambda f: f[..., :3] / f[..., 3:4]
    la, lb, da, dc = mean(orc_a), mean(orc_b), mean(films[seed_a]), mean(films[seed_b + 1000])
    noise = _rel_mse(lb, la)
    same = _rel_mse(da, la)
    other = _rel_mse(dc, la)
    rmse_same = _box4_rmse_8bit(rrt, films[seed_a], orc_a)
    rmse_other = _box4_rmse_8bit(rrt, films[seed_b + 1000], orc_a)
    rmse_noise = _box4_rmse_8bit(rrt, orc_b, orc_a)
    # bias test on the crop's total: variance of one film's total ~ sum over pixels of (a - b)^2 / 2
    sigma_total = np.sqrt(((la - lb) ** 2).sum() / 2.0)
    z_other = abs(dc.sum() - la.sum()) / (np.sqrt(2.0) * sigma_total)
    z_same = abs(da.sum() - la.sum()) / (np.sqrt(2.0) * sigma_total)
    print("%s: relMSE oracle seed-to-seed %.3g, device same seed %.3g (%.4f of it), device other seed %.3g (%.3f of it); "
          "box-filtered 8-bit RMSE same %.3f/255, other %.3f/255 (oracle's own %.3f/255); total radiance z: other seed %.2f, "
          "same seed %.3f; mean L oracle %.5f device %.5f" % (which, noise, same, same / noise, other, other / noise,
                                                            rmse_same * 255, rmse_other * 255, rmse_noise * 255, z_other,
                                                            z_same, la.mean(), dc.mean()))
    parity_record("converged_%s" % which, spp=spp, relmse_oracle_seed_to_seed=noise, relmse_device_same_seed=same,
                  relmse_device_other_seed=other, rmse8_box4_same_seed=rmse_same * 255, rmse8_box4_other_seed=rmse_other * 255,
                  rmse8_box4_oracle_seed_to_seed=rmse_noise * 255, z_total_other_seed=z_other, z_total_same_seed=z_same,
                  mean_radiance_oracle=la.mean(), mean_radiance_device_other_seed=dc.mean())
    assert noise > 0
    assert same <= 0.02 * noise
    assert rmse_same * 255 <= 0.25
    assert other <= 2.0 * noise
    assert rmse_other * 255 <= 2.0
    assert z_other <= 4.0
    assert z_same <= 0.5


def test_4k_sample_ranges_compose(rrt, buddha, gpu_ctx, parity_record):
    """The 3840 x 2160 configuration on one GPU: the sample ranges [0, 512) and [512, 1024) -- what two of the
    GPUs of a sample-sharded render would take -- add up to the film of [0, 1024) rendered in one call, up to
    float summation order; every pixel gets every sample."""
    scene, rs = buddha
    w, h, spp = 3840, 2160, 1024
    cam = rrt.Camera.from_settings(rs, w, h)
    s = rrt.make_settings(w, h, rs.depth_limit, rs.clamp, seed=2)
    gpu_ctx.upload_scene(scene)
    gpu_ctx.set_camera(cam)
    gpu_ctx.film_alloc(w, h)
    gpu_ctx.reset_stats()
    gpu_ctx.render(s, 0, spp)
    whole = gpu_ctx.film_download()
    assert gpu_ctx.stats().paths == w * h * spp
    assert (whole[..., 3] == spp).all() and np.isfinite(whole).all()
    gpu_ctx.film_clear()
    gpu_ctx.render(s, 0, spp // 2)
    first = gpu_ctx.film_download()
    assert (first[..., 3] == spp // 2).all()
    gpu_ctx.render(s, spp // 2, spp)
    parts = gpu_ctx.film_download()
    np.testing.assert_array_equal(parts[..., 3], whole[..., 3])
    diff = np.abs(parts[..., :3].astype(np.float64) - whole[..., :3]).max() / whole[..., :3].max()
    print("4K: max difference %.3g of the film's maximum" % diff)
    parity_record("compose_4k_1024spp", max_diff_over_film_max=diff)
    assert diff <= 1e-5


@pytest.mark.parametrize("which", ["buddha", "mixed"])
def test_tail_kernel_changes_nothing(rrt, buddha, mixed, gpu_ctx, which):
    """RRT_OPT_TAIL_PATHS: the kernel that runs the last live paths to their end (continuation ray, shading,
    visibility rays, next bounce, all by one lane) uses the traversal engine and the shading code of the
    separate kernels in the same order: the film is the same bit for bit whether it never runs (0), takes over
    late (the default) or right after the second iteration (2^30), and so are the ray and bounce counts."""
    scene, rs = buddha if which == "buddha" else mixed[:2]
    w, h, spp = (480, 270, 16) if which == "buddha" else (240, 180, 8)
    cam = rrt.Camera.from_settings(rs, w, h)
    s = rrt.make_settings(w, h, rs.depth_limit, rs.clamp, seed=13)
    gpu_ctx.upload_scene(scene)
    gpu_ctx.set_camera(cam)
    gpu_ctx.film_alloc(w, h)
    films, stats = [], []
    try:
        for tail in (0, 1 << 30, 4096, 1 << 21):
            gpu_ctx.set_option(rrt.OPT_TAIL_PATHS, tail)
            gpu_ctx.set_option(rrt.OPT_TIMING, 1)  # launches are counted from the timing events
            gpu_ctx.film_clear()
            gpu_ctx.reset_stats()
            gpu_ctx.render(s, 0, spp)
            films.append(gpu_ctx.film_download())
            st = gpu_ctx.stats()
            stats.append((st.paths, st.rays_extend, st.rays_shadow, st.path_bounces))
            # (with a small threshold the queues may run empty between two looks at their lengths: then it never starts)
            assert st.launches_tail == (0 if tail == 0 else 1) or (tail == 4096 and st.launches_tail == 0), (tail, st.launches_tail)
    finally:
        gpu_ctx.set_option(rrt.OPT_TAIL_PATHS, 1 << 21)
        gpu_ctx.set_option(rrt.OPT_TIMING, 0)
    assert films[0][..., :3].max() > 0 and stats[0][3] > stats[0][0] // 8
    for f, st in zip(films[1:], stats[1:]):
        np.testing.assert_array_equal(f, films[0])
        assert st == stats[0]


def _traversal_cost(rrt, ctx, cam, s, spp):
    """Mean nodes and triangles fetched per ray over a small render (counting build of the kernels)."""
    ctx.set_camera(cam)
    ctx.film_alloc(s.width, s.height)
    ctx.set_option(rrt.OPT_COUNT_TRAVERSAL, 1)
    try:
        ctx.reset_stats()
        ctx.render(s, 0, spp)
        st = ctx.stats()
    finally:
        ctx.set_option(rrt.OPT_COUNT_TRAVERSAL, 0)
    rays = st.rays_extend + st.rays_shadow
    return st.nodes_visited / rays, st.tris_tested / rays, ctx.film_download()


@pytest.mark.parametrize("which", ["cornell", "buddha68k", "stress24"])
def test_device_built_hierarchy(rrt, cornell, buddha, gpu_ctx, parity_record, which):
    """RRT_OPT_DEVICE_BVH = 1: the hierarchy built by kernels (Morton order + locally-ordered clustering, 4-wide
    collapse and quantisation on the device) against the host-built one (binned SAH): sound by the same invariant
    check (every triangle inside every box on the way to it, each in exactly one leaf, records in leaf order), the
    same first hits on a ray set, the same film up to the paths that meet an edge (a ray through a shared edge
    takes whichever of the two triangles its traversal order reaches first), and a traversal cost -- nodes
    fetched per ray over a render -- within 15 % of the host tree's."""
    if which == "cornell":
        scene, rs = cornell
        w, h, spp, depth, clamp = 128, 128, 4, rs.depth_limit, rs.clamp
        cam = rrt.Camera.from_settings(rs, w, h)
    elif which == "buddha68k":
        scene, rs = buddha
        w, h, spp, depth, clamp = 240, 135, 4, rs.depth_limit, rs.clamp
        cam = rrt.Camera.from_settings(rs, w, h)
    else:
        _, (scene, cam_args, depth, clamp) = scenes_util.stress_scene(rrt, 24, grid=(4, 3, 2))
        w, h, spp = 160, 90, 4
        cam = rrt.Camera(image_width=w, image_height=h, **cam_args)
    d = scene.desc()
    pos = np.ctypeslib.as_array(d.tri_pos, shape=(d.n_tris, 9)).copy()
    s = rrt.make_settings(w, h, depth, clamp, seed=17)
    rng = np.random.default_rng(4)
    px = rng.uniform(0, [w, h], size=(4000, 2))
    rays = np.array([np.concatenate(cam.generate_ray(x, y)) for x, y in px]).astype(np.float32)
    results = {}
    try:
        for mode in (0, 1):
            gpu_ctx.set_option(rrt.OPT_DEVICE_BVH, mode)
            gpu_ctx.upload_scene(scene)
            bad, nodes, tree_depth, on_device, secs = gpu_ctx.bvh_check(pos)
            assert bad == 0 and on_device == bool(mode), (mode, bad, on_device)
            prim, t = gpu_ctx.intersect(rays)
            odt = np.concatenate([rays, np.full((len(rays), 1), 1.5, np.float32)], axis=1)
            occ = gpu_ctx.occluded(odt)
            nodes_per_ray, tris_per_ray, film = _traversal_cost(rrt, gpu_ctx, cam, s, spp)
            results[mode] = dict(prim=prim, t=t, occ=occ, nodes=nodes, depth=tree_depth, secs=secs, npr=nodes_per_ray,
                                 tpr=tris_per_ray, film=film.astype(np.float64))
    finally:
        gpu_ctx.set_option(rrt.OPT_DEVICE_BVH, 2)
    host, dev = results[0], results[1]
    same = host["prim"] == dev["prim"]
    print("%s: %d triangles; host tree %d nodes depth %d, %.2f nodes + %.2f triangles per ray; device tree %d nodes depth %d, "
          "%.2f + %.2f per ray, built in %.1f ms; first hits agree on %.5f, occlusion on %.5f" % (
              which, d.n_tris, host["nodes"], host["depth"], host["npr"], host["tpr"], dev["nodes"], dev["depth"], dev["npr"],
              dev["tpr"], dev["secs"] * 1e3, same.mean(), (host["occ"] == dev["occ"]).mean()))
    assert same.mean() >= 0.999 and (host["prim"] >= 0).sum() > 500
    np.testing.assert_array_equal(host["t"][same], dev["t"][same])
    assert (host["occ"] == dev["occ"]).mean() >= 0.999
    np.testing.assert_array_equal(host["film"][..., 3], dev["film"][..., 3])
    rel = np.abs(host["film"][..., :3] - dev["film"][..., :3]).sum() / host["film"][..., :3].sum()
    assert rel <= 2e-3, rel
    cost_ratio = (dev["npr"] * 64 + dev["tpr"] * 36) / (host["npr"] * 64 + host["tpr"] * 36)
    parity_record("device_bvh_%s" % which, triangles=int(d.n_tris), first_hit_agreement=same.mean(), film_rel_diff=rel,
                  nodes_per_ray_host=host["npr"], nodes_per_ray_device=dev["npr"], tris_per_ray_host=host["tpr"],
                  tris_per_ray_device=dev["tpr"], traversal_bytes_ratio=cost_ratio, device_build_ms=dev["secs"] * 1e3)
    assert dev["npr"] <= 1.15 * host["npr"], (dev["npr"], host["npr"])
    assert cost_ratio <= 1.15, cost_ratio


def _soup(rng, n, spread=10.0, size=0.05):
    centre = rng.uniform(-spread, spread, size=(n, 1, 3))
    pos = (centre + rng.normal(scale=size, size=(n, 3, 3))).astype(np.float32)
    nrm = np.cross(pos[:, 1] - pos[:, 0], pos[:, 2] - pos[:, 0])
    nrm /= np.maximum(np.linalg.norm(nrm, axis=1, keepdims=True), 1e-30)
    return pos, np.repeat(nrm[:, None, :], 3, axis=1).astype(np.float32)


def test_device_build_failure_falls_back_to_the_host_build(rrt):
    """RRT_OPT_DEVICE_BVH = 2 (the default) builds the hierarchy by kernels from 4 Mi triangles on.  When that
    build fails -- here a context created with RRT_DEBUG_FAIL_DEVICE_BVH in the environment, which refuses every
    device build as if the GPU had no memory left for its temporaries -- the upload does not: the host builder
    runs instead and the scene renders.  Asked for explicitly (= 1), the failure is reported."""
    rng = np.random.default_rng(8)
    n = (4 << 20) + 1000
    pos, nrm = _soup(rng, n, spread=40.0, size=0.08)
    scene = rrt.SceneArrays(pos, nrm, np.zeros(n), materials=[rrt.make_material(rrt.MAT_MATTE, kd=(0.5, 0.5, 0.5))])
    os.environ["RRT_DEBUG_FAIL_DEVICE_BVH"] = "1"
    try:
        ctx = rrt.Context(0)
    finally:
        del os.environ["RRT_DEBUG_FAIL_DEVICE_BVH"]
    try:
        ctx.upload_scene(scene)
        bad, nodes, depth, on_device, _ = ctx.bvh_check(pos.reshape(-1, 9))
        assert bad == 0 and not on_device and nodes > n // 8
        rays = np.concatenate([rng.uniform(-40, 40, size=(2000, 3)), rng.normal(size=(2000, 3))], axis=1).astype(np.float32)
        prim, t = ctx.intersect(rays)
        assert (prim >= 0).sum() > 20
        ctx.set_option(rrt.OPT_DEVICE_BVH, 1)
        with pytest.raises(rrt.RrtError, match="device build refused"):
            ctx.upload_scene(scene)
    finally:
        ctx.close()
    # an ordinary context builds the same scene on the device and finds the same hits
    ctx = rrt.Context(0)
    try:
        ctx.upload_scene(scene)
        bad, _, _, on_device, _ = ctx.bvh_check(pos.reshape(-1, 9))
        assert bad == 0 and on_device
        prim2, t2 = ctx.intersect(rays)
        same = prim == prim2
        assert same.mean() > 0.999
        np.testing.assert_array_equal(t[same], t2[same])
    finally:
        ctx.close()


@pytest.mark.parametrize("kind", ["identical", "zero_area", "strip"])
def test_device_build_on_degenerate_input(rrt, gpu_ctx, kind):
    """Input on which every candidate pair of the clustering has the same surface: a few thousand copies of one
    triangle, triangles collapsed onto points, a strip of equal boxes.  The build by kernels must finish in a few
    dozen rounds with a tree shallow enough for the traversal stack (pairs (0,1), (2,3), ... merge per round; with
    'the lower index wins' one pair merged per round and the tree came out as deep as the input is long),
    and rays must find what the host-built tree finds."""
    rng = np.random.default_rng(12)
    n = 6000
    if kind == "identical":
        pos = np.tile(np.array([[[0.0, 0.0, 0.0], [1.0, 0.0, 0.0], [0.0, 1.0, 0.0]]], dtype=np.float32), (n, 1, 1))
    elif kind == "zero_area":
        pts = rng.integers(-3, 4, size=(n, 1, 3)).astype(np.float32)
        pos = np.tile(pts, (1, 3, 1))
        pos[: n // 10] = _soup(rng, n // 10, spread=3.0, size=0.3)[0]   # and some proper triangles to hit
    else:
        x = np.arange(n, dtype=np.float32)[:, None]
        pos = np.stack([np.concatenate([x, 0 * x, 0 * x], 1), np.concatenate([x + 1, 0 * x, 0 * x], 1),
                        np.concatenate([x, 0 * x + 1, 0 * x], 1)], axis=1)
    nrm = np.tile(np.array([0.0, 0.0, 1.0], dtype=np.float32), (n, 3, 1))
    scene = rrt.SceneArrays(pos, nrm, np.zeros(n), materials=[rrt.make_material(rrt.MAT_MATTE, kd=(0.5, 0.5, 0.5))])
    lo, hi = pos.reshape(-1, 3).min(0), pos.reshape(-1, 3).max(0)
    org = rng.uniform(lo - 1.0, hi + 1.0, size=(3000, 3))
    org[:, 2] = 5.0
    rays = np.concatenate([org, np.tile([0.0, 0.0, -1.0], (3000, 1))], axis=1).astype(np.float32)
    out = {}
    try:
        for mode in (0, 1):
            gpu_ctx.set_option(rrt.OPT_DEVICE_BVH, mode)
            gpu_ctx.upload_scene(scene)
            bad, nodes, depth, on_device, _ = gpu_ctx.bvh_check(pos.reshape(-1, 9))
            assert bad == 0 and on_device == bool(mode)
            assert depth <= 40, (kind, mode, depth)
            out[mode] = gpu_ctx.intersect(rays)
    finally:
        gpu_ctx.set_option(rrt.OPT_DEVICE_BVH, 2)
    # which of several coincident triangles a ray reports depends on the order of the leaves; hit or miss and t do not
    np.testing.assert_array_equal(out[0][0] >= 0, out[1][0] >= 0)
    np.testing.assert_array_equal(out[0][1], out[1][1])
    if kind != "zero_area":
        assert (out[0][0] >= 0).sum() > 100


def test_fifty_million_triangles(rrt, gpu_ctx):
    """The many-object configuration at its full size: 3200 cow / lamp / teapot trios = 50,022,402 triangles,
    depth 16.  No oracle comparison at this size (the 24-trio cut has it); what must hold whatever the size:
    the hierarchy -- built by kernels, the default from 4 Mi triangles on -- passes the host's invariant check
    (every triangle inside every box on the way to it, each in exactly one leaf, records in leaf order), every
    camera path of a 1080p sample is traced, and the film is finite and lit."""
    stress, (scene, cam_args, depth, clamp) = scenes_util.stress_scene(rrt, 3200, grid=(16, 16, 13))
    d = scene.desc()
    assert d.n_tris == 3200 * 15632 + 2
    gpu_ctx.upload_scene(scene)
    try:
        bad, nodes, tree_depth, on_device, secs = gpu_ctx.bvh_check(np.ctypeslib.as_array(d.tri_pos, shape=(d.n_tris, 9)))
        assert bad == 0 and on_device and tree_depth <= 63
        w, h, spp = 1920, 1080, 2
        gpu_ctx.set_camera(rrt.Camera(image_width=w, image_height=h, **cam_args))
        gpu_ctx.film_alloc(w, h)
        gpu_ctx.reset_stats()
        gpu_ctx.render(rrt.make_settings(w, h, depth, clamp, seed=1), 0, spp)
        st = gpu_ctx.stats()
        film = gpu_ctx.film_download()
        assert st.paths == w * h * spp
        assert (film[..., 3] == spp).all() and np.isfinite(film).all()
        assert film[..., :3].mean() > 1e-3 and st.path_bounces > st.paths // 2
        print("50 M triangles: %d nodes, depth %d, built on the device in %.2f s; %.2f surface hits per path" % (
            nodes, tree_depth, secs, st.path_bounces / st.paths))
    finally:
        gpu_ctx.upload_scene(rrt.SceneArrays(materials=[rrt.make_material(rrt.MAT_MATTE, kd=(0.5, 0.5, 0.5))]))  # frees 6 GB
