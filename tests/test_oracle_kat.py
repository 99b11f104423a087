"""The reference's own known-answer tests, run against the CPU oracle (and the host camera).

Every test below restates one #[test] of the reference, with its file:line, and checks the same
inputs against the same expectation.  These 25 tests are everything the reference pins about the
hot path; they are what the oracle is anchored to.
"""
import ctypes as C
import math

import numpy as np
import pytest


@pytest.fixture(scope="module")
def L(oracle_binding):
    return oracle_binding.lib()


def vec(v):
    return (C.c_double * len(v))(*[float(x) for x in v])


# ---- src/objects/triangle.rs:379-426 it_tests_intersects
def test_triangle_it_tests_intersects(L):
    p = vec([-1.0, -1.0, 0.0, 1.0, -1.0, 0.0, 1.0, 1.0, 0.0])
    n = vec([0.6, 0.0, -1.0, 0.0, 0.5, -1.0, 0.4, 0.0, -1.0])
    t = C.c_double()
    point, ns, ng = vec([0] * 3), vec([0] * 3), vec([0] * 3)
    hit = L.orc_triangle_intersect(p, n, vec([0.0, 0.0, -2.0]), vec([0.0, 0.0, 1.0]), C.byref(t), point, ns, ng)
    assert hit == 1
    expected = np.array([0.5, 0.0, -1.0]) / np.linalg.norm([0.5, 0.0, -1.0])
    np.testing.assert_allclose(np.array(ns[:]), expected, rtol=1.0e-6, atol=2.3e-16)
    assert t.value == 2.0


# ---- src/helpers.rs:244-321
def test_fresnel_ratio_low_at_impact_on_normal(L):
    r = L.orc_get_fresnel_ratio(vec([0, 0, -1]), vec([0, 0, 1]), 1.5)
    assert 0.03999999 < r < 0.04000001


def test_fresnel_ratio_one_at_perpendicular_of_normal(L):
    r = L.orc_get_fresnel_ratio(vec([0, 0, -1]), vec([1, 0, 0]), 1.5)
    assert 0.99999999 < r < 1.00000001


def test_fresnel_ratio_half_at_around_82_deg_on_normal(L):
    v = np.array([1.0, 0.0, 0.1259879])
    v /= np.linalg.norm(v)
    r = L.orc_get_fresnel_ratio(vec([0, 0, -1]), vec(v), 1.5)
    assert 0.499999 < r < 0.500001


def test_max_dimension_vec_3(L):
    assert L.orc_max_dimension(vec([1, 3, 2])) == 1


def test_permute(L):
    out = vec([0] * 3)
    L.orc_permute(vec([1, 2, 3]), 2, 1, 0, out)
    assert out[:] == [3.0, 2.0, 1.0]


def test_spherical_phi(L):
    assert abs(L.orc_spherical_phi(vec([0, 0, 1])) - 0.0) < 1e-6
    assert abs(L.orc_spherical_phi(vec([1, 0, 0])) - math.pi / 2) < 1e-6
    assert abs(L.orc_spherical_phi(vec([0, 0, -1])) - math.pi) < 1e-6
    assert abs(L.orc_spherical_phi(vec([-1, 0, 0])) - 3 * math.pi / 2) < 1e-6


def test_spherical_theta(L):
    assert abs(L.orc_spherical_theta(vec([0, 1, 0])) - 0.0) < 1e-6
    assert abs(L.orc_spherical_theta(vec([1, 0, 0])) - math.pi / 2) < 1e-6
    assert abs(L.orc_spherical_theta(vec([0, -1, 0])) - math.pi) < 1e-6


# ---- src/bsdf/helpers/fresnel.rs:76-154
def test_fresnel_dielectric_normal_incidence(L):
    assert L.orc_fresnel_dielectric(1.0, 1.5, 1.0) == pytest.approx(0.04, abs=1e-4)


def test_fresnel_dielectric_no_reflection(L):
    assert L.orc_fresnel_dielectric(1.5, 1.5, 1.0) == pytest.approx(0.0, abs=1e-4)
    assert L.orc_fresnel_dielectric(1.5, 1.5, 0.5) == pytest.approx(0.0, abs=1e-4)


def test_fresnel_dielectric_tir(L):
    assert L.orc_fresnel_dielectric(1.5, 1.0, 0.5) == pytest.approx(1.0, abs=1e-4)


def test_fresnel_dielectric_swap(L):
    assert L.orc_fresnel_dielectric(1.0, 1.5, -1.0) == pytest.approx(0.04, abs=1e-4)


def test_fresnel_dielectric_parallel_perpendicular(L):
    assert L.orc_fresnel_dielectric(1.0, 1.5, 0.5) == pytest.approx(0.08918, abs=1e-3)


def test_fresnel_noop(L):
    # FresnelNoop evaluates to 1 for any cosine: SpecularReflection with it returns R / |cos|
    for wo in ([0.0, 0.0, 1.0], [0.0, 0.6, 0.8], [0.0, 0.6, -0.8]):
        wi, f, pdf = vec([0] * 3), vec([0] * 3), C.c_double()
        L.orc_specular_reflection_sample_f(vec([1, 1, 1]), 0, 1.0, 1.0, vec(wo), wi, C.byref(pdf), f)
        assert f[0] * abs(wi[2]) == pytest.approx(1.0)


# ---- src/bsdf/helpers/microfacet_distribution.rs:225-326
def test_trowbridge_reitz_sample_11(L):
    out = vec([0, 0])
    for args in ((1.0, 0.5, 0.5), (0.5, 0.3, 0.7)):
        L.orc_tr_sample_11(*args, out)
        assert all(math.isfinite(x) for x in out[:])


def test_trowbridge_reitz_sample(L):
    out = vec([0] * 3)
    L.orc_tr_sample(vec([0, 0, 1]), 0.1, 0.1, 0.5, 0.5, out)
    assert abs(np.linalg.norm(out[:]) - 1.0) < 1e-6 and out[2] > 0.0
    wi = np.ones(3) / math.sqrt(3.0)
    L.orc_tr_sample(vec(wi), 0.5, 0.2, 0.2, 0.8, out)
    assert abs(np.linalg.norm(out[:]) - 1.0) < 1e-6 and all(math.isfinite(x) for x in out[:])


def test_sample_wh(L):
    out = vec([0] * 3)
    for ax, ay, visible in ((0.1, 0.1, 0), (0.1, 0.1, 1), (0.5, 0.1, 0)):
        L.orc_tr_sample_wh(vec([0, 0, 1]), ax, ay, visible, 0.9, 0.9, out)
        assert abs(np.linalg.norm(out[:]) - 1.0) < 1e-6 and all(math.isfinite(x) for x in out[:])


def test_pdf(L):
    for visible in (0, 1):
        p = L.orc_tr_pdf(vec([0, 0, 1]), vec([0, 0, 1]), 0.1, 0.1, visible)
        assert p > 0.0 and math.isfinite(p)


# ---- src/bsdf/microfacet_reflection.rs:108-141
def test_microfacet_reflection_f(L):
    out = vec([0] * 3)
    L.orc_microfacet_f(vec([0.9] * 3), 0.5, 0.5, 1.0, 1.5, vec([0, 0, 1]), vec([0, 0, 1]), out)
    assert all(x >= 0.0 and math.isfinite(x) for x in out[:])
    L.orc_microfacet_f(vec([0.9] * 3), 0.5, 0.5, 1.0, 1.5, vec([0.6, 0.0, 0.8]), vec([-0.6, 0.0, 0.8]), out)
    assert out[0] >= 0.0 and math.isfinite(out[0])


# ---- src/bsdf/specular_reflection.rs:44-151
def _spec_sample(L, kind, wo):
    wi, f, pdf = vec([0] * 3), vec([0] * 3), C.c_double()
    L.orc_specular_reflection_sample_f(vec([1, 1, 1]), kind, 1.0, 1.5, vec(wo), wi, C.byref(pdf), f)
    return np.array(wi[:]), pdf.value, np.array(f[:])


def test_specular_reflection_sample_f(L):
    wo = np.ones(3) / math.sqrt(3.0)
    wi, pdf, f = _spec_sample(L, 0, wo)
    np.testing.assert_allclose(wi, [-wo[0], -wo[1], wo[2]])
    assert pdf == pytest.approx(1.0)
    np.testing.assert_allclose(f, np.ones(3) / abs(wi[2]))


def test_specular_reflection_f(L):
    f, pdf = vec([0] * 3), C.c_double()
    wo = np.ones(3) / math.sqrt(3.0)
    L.orc_specular_reflection_f_pdf(vec([1, 1, 1]), vec(wo), vec([-wo[0], -wo[1], wo[2]]), f, C.byref(pdf))
    assert f[:] == [0.0, 0.0, 0.0]


def test_specular_reflection_pdf(L):
    f, pdf = vec([0] * 3), C.c_double()
    wo = np.ones(3) / math.sqrt(3.0)
    L.orc_specular_reflection_f_pdf(vec([1, 1, 1]), vec(wo), vec([-wo[0], -wo[1], wo[2]]), f, C.byref(pdf))
    assert pdf.value == pytest.approx(1.0)


def test_specular_reflection_with_dielectric(L):
    wi, _, f = _spec_sample(L, 1, [0.0, 0.0, 1.0])
    np.testing.assert_allclose(wi, [0.0, 0.0, 1.0])
    np.testing.assert_allclose(f, np.ones(3) * 0.04 / abs(wi[2]), rtol=1e-9)


def test_specular_reflection_with_dielectric_low_angle(L):
    wo = np.array([1.0, 0.0, 0.01])
    wo /= np.linalg.norm(wo)
    wi, _, f = _spec_sample(L, 1, wo)
    np.testing.assert_allclose(wi, [-wo[0], -wo[1], wo[2]])
    expected = L.orc_fresnel_dielectric(1.0, 1.5, wi[2]) / abs(wi[2])
    np.testing.assert_allclose(f, np.ones(3) * expected, rtol=1e-9)


# ---- src/camera.rs:166-259 test (host Camera; the oracle and the device consume its matrices)
def test_camera(rrt):
    cam = rrt.Camera((0.0, 0.0, 1.0), (0.0, 0.0, 0.0), 90.0, 0.0, None, 100, 100)
    _, d = cam.generate_ray(50.0, 50.0)
    np.testing.assert_allclose(d, [0.0, 0.0, -1.0], atol=1e-12)
    _, dl = cam.generate_ray(0.0, 50.0)
    _, dr = cam.generate_ray(100.0, 50.0)
    angle = math.degrees(math.acos(np.clip(np.dot(dl, dr), -1, 1)))
    assert angle == pytest.approx(90.0, rel=1e-5)

    cam = rrt.Camera((0.0, 0.0, 1.0), (0.0, 1.0, 0.0), 90.0, 0.0, None, 100, 100)
    _, d = cam.generate_ray(50.0, 50.0)
    np.testing.assert_allclose(d, np.array([0.0, 1.0, -1.0]) / math.sqrt(2.0), atol=1e-12)
    _, dl = cam.generate_ray(0.0, 50.0)
    _, dr = cam.generate_ray(100.0, 50.0)
    angle = math.degrees(math.acos(np.clip(np.dot(dl, dr), -1, 1)))
    assert angle == pytest.approx(90.0, rel=1e-5)


# ---- the replacement random number generator: Philox4x32-10 against the Random123 known answers
def test_philox_known_answers(L):
    def run(c, k):
        out = (C.c_uint32 * 4)()
        L.orc_philox4x32_10((C.c_uint32 * 4)(*c), (C.c_uint32 * 2)(*k), out)
        return list(out)

    assert run([0] * 4, [0] * 2) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert run([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert run([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_rng_draws_are_open_unit_interval_and_f32_exact(L):
    vals = np.array([L.orc_rng(7, p, s, b, i) for p in range(8) for s in range(4) for b in range(5) for i in range(4)])
    assert (vals > 0).all() and (vals < 1).all()
    assert (vals.astype(np.float32).astype(np.float64) == vals).all()
    assert 0.4 < vals.mean() < 0.6
