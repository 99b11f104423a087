"""rrt_intersect_noalloc against rrt_intersect through the C ABI: the same rays, staged through the
library's static device array instead of per-call allocations, must give the same object ids and the
same distances bit for bit (same kernel, same inputs) -- on two calls in a row, the second with a
smaller input, so that what the first call left in the array must not leak into the second."""
import numpy as np
import pytest

import scenes_util

pytestmark = pytest.mark.gpu


def test_noalloc_matches_intersect_twice(rrt, oracle_binding, cornell, gpu_ctx):
    scene, rs = cornell
    gpu_ctx.upload_scene(scene)
    orc = oracle_binding.Oracle(scene.desc())
    rays = scenes_util.ray_set(rrt, orc, rs, 5000, 29).astype(np.float32)
    assert len(rays) > 6000
    for n in (len(rays), 777):  # the second, smaller call runs over the first one's leftovers
        sub = rays[:n] if n == len(rays) else rays[-n:]
        ref_prim, ref_t = gpu_ctx.intersect(sub)
        prim, t = gpu_ctx.intersect_noalloc(sub)
        np.testing.assert_array_equal(prim, ref_prim)
        np.testing.assert_array_equal(t, ref_t)
        assert (prim >= 0).sum() > n // 2
    # and against the oracle on the well separated rays, to the first-hit tolerance
    o_prim, o_t, margin = orc.intersect(rays.astype(np.float64), with_margin=True)
    prim, t = gpu_ctx.intersect_noalloc(rays)
    clear = margin >= 1e-4
    np.testing.assert_array_equal(prim[clear], o_prim[clear])
    hit = clear & (o_prim >= 0)
    assert (np.abs(t[hit] - o_t[hit]) <= 1e-5 * o_t[hit]).all()  # 1e-5 relative


def test_noalloc_limits(rrt, cornell, gpu_ctx):
    scene, _ = cornell
    gpu_ctx.upload_scene(scene)
    prim, _ = gpu_ctx.intersect_noalloc(np.zeros((0, 6), dtype=np.float32))
    assert len(prim) == 0
    rays = np.zeros((rrt.NOALLOC_MAX_RAYS + 1, 6), dtype=np.float32)
    rays[:, 4] = 1.0
    with pytest.raises(rrt.RrtError, match="RRT_NOALLOC_MAX_RAYS"):
        gpu_ctx.intersect_noalloc(rays)
    # exactly the maximum is accepted and agrees with the allocating entry point
    rng = np.random.default_rng(1)
    full = np.concatenate([rng.uniform(-0.9, 0.9, (rrt.NOALLOC_MAX_RAYS, 3)), rng.normal(size=(rrt.NOALLOC_MAX_RAYS, 3))], axis=1)
    full[:, 3:] /= np.linalg.norm(full[:, 3:], axis=1, keepdims=True)
    full = full.astype(np.float32)
    a = gpu_ctx.intersect(full)
    b = gpu_ctx.intersect_noalloc(full)
    np.testing.assert_array_equal(a[0], b[0])
    np.testing.assert_array_equal(a[1], b[1])


def test_noalloc_allocates_no_device_memory(rrt, cornell, gpu_ctx):
    """The entry point's reason to exist: it leaves the device's free memory where it was.  hipMemGetInfo before,
    between and after calls (after a first call that loads what the runtime loads lazily) reports the same free
    bytes every time -- and the allocating rrt_intersect still returns what it took."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")

    def free_bytes():
        free, total = C.c_size_t(), C.c_size_t()
        assert hip.hipMemGetInfo(C.byref(free), C.byref(total)) == 0
        return free.value
    scene, _ = cornell
    gpu_ctx.upload_scene(scene)
    rng = np.random.default_rng(3)
    rays = np.concatenate([rng.uniform(-0.9, 0.9, (50000, 3)), rng.normal(size=(50000, 3))], axis=1).astype(np.float32)
    gpu_ctx.intersect_noalloc(rays)   # warm-up
    gpu_ctx.synchronize()
    before = free_bytes()
    seen = []
    for n in (50000, 123, 50000):
        gpu_ctx.intersect_noalloc(rays[:n])
        seen.append(free_bytes())
    assert seen == [before] * 3, (before, seen)
    gpu_ctx.intersect(rays)
    gpu_ctx.synchronize()
    assert free_bytes() == before
