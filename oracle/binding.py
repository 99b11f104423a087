"""ctypes access to the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


class OracleStats(C.Structure):
    _fields_ = [("paths", C.c_uint64), ("rays_extend", C.c_uint64), ("rays_shadow", C.c_uint64),
                ("path_bounces", C.c_uint64), ("nodes_visited", C.c_uint64), ("prims_tested", C.c_uint64),
                ("seconds", C.c_double), ("threads", C.c_uint32), ("threads_busy", C.c_uint32),
                ("buckets", C.c_uint32), ("_pad", C.c_uint32)]


_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", HERE])


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        dp = C.POINTER(C.c_double)
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [C.c_void_p]
        L.orc_destroy.argtypes = [C.c_void_p]
        L.orc_destroy.restype = None
        L.orc_set_camera.argtypes = [C.c_void_p, dp, dp, C.c_double, C.c_double]
        L.orc_set_filter.argtypes = [C.c_void_p, C.c_uint32, C.c_double]
        L.orc_render.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                 C.c_void_p, C.POINTER(OracleStats)]
        L.orc_trace_one.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, dp]
        L.orc_first_hit.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, dp, dp]
        L.orc_intersect.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        L.orc_intersect_margin.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_occluded.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        L.orc_philox4x32_10.argtypes = [C.POINTER(C.c_uint32)] * 3
        L.orc_philox4x32_10.restype = None
        L.orc_rng.restype = C.c_double
        L.orc_rng.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
        L.orc_triangle_intersect.argtypes = [dp, dp, dp, dp, dp, dp, dp, dp]
        L.orc_fresnel_dielectric.restype = C.c_double
        L.orc_fresnel_dielectric.argtypes = [C.c_double] * 3
        L.orc_get_fresnel_ratio.restype = C.c_double
        L.orc_get_fresnel_ratio.argtypes = [dp, dp, C.c_double]
        L.orc_max_dimension.argtypes = [dp]
        L.orc_permute.argtypes = [dp, C.c_int, C.c_int, C.c_int, dp]
        L.orc_spherical_phi.restype = C.c_double
        L.orc_spherical_phi.argtypes = [dp]
        L.orc_spherical_theta.restype = C.c_double
        L.orc_spherical_theta.argtypes = [dp]
        L.orc_tr_sample_11.argtypes = [C.c_double, C.c_double, C.c_double, dp]
        L.orc_tr_sample.argtypes = [dp, C.c_double, C.c_double, C.c_double, C.c_double, dp]
        L.orc_tr_sample_wh.argtypes = [dp, C.c_double, C.c_double, C.c_int, C.c_double, C.c_double, dp]
        L.orc_tr_pdf.restype = C.c_double
        L.orc_tr_pdf.argtypes = [dp, dp, C.c_double, C.c_double, C.c_int]
        L.orc_microfacet_f.argtypes = [dp, C.c_double, C.c_double, C.c_double, C.c_double, dp, dp, dp]
        L.orc_specular_reflection_sample_f.argtypes = [dp, C.c_int, C.c_double, C.c_double, dp, dp, dp, dp]
        L.orc_specular_reflection_f_pdf.argtypes = [dp, dp, dp, dp, dp]
        L.orc_tonemap_pixel.argtypes = [dp, C.c_void_p]
        _lib = L
    return _lib


def vec(v):
    return (C.c_double * len(v))(*[float(x) for x in v])


class Oracle:
    """An oracle context over the same rrt_scene_desc the device library takes."""

    def __init__(self, scene_desc):
        self._lib = lib()
        self._desc = scene_desc  # keep the arrays alive
        self._h = self._lib.orc_create(C.byref(scene_desc))
        if not self._h:
            raise RuntimeError("orc_create failed")

    def close(self):
        if self._h:
            self._lib.orc_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_camera(self, camera):
        dp = C.POINTER(C.c_double)
        r2c = np.ascontiguousarray(camera.raster_to_camera, dtype=np.float64)
        c2w = np.ascontiguousarray(camera.camera_to_world, dtype=np.float64)
        rc = self._lib.orc_set_camera(self._h, r2c.ctypes.data_as(dp), c2w.ctypes.data_as(dp), camera.aperture,
                                      camera.focal_distance)
        assert rc == 0

    def set_filter(self, method, radius=0.0):
        rc = self._lib.orc_set_filter(self._h, method, radius)
        assert rc == 0

    def render(self, settings, sample_begin, sample_end, threads=0, bucket=(128, 128), film=None):
        """Returns (film float64 [h][w][4], OracleStats)."""
        if film is None:
            film = np.zeros((settings.height, settings.width, 4), dtype=np.float64)
        st = OracleStats()
        rc = self._lib.orc_render(self._h, C.byref(settings), sample_begin, sample_end, bucket[0], bucket[1], threads,
                                  film.ctypes.data, C.byref(st))
        assert rc == 0, rc
        return film, st

    def trace_one(self, settings, px, py, sample):
        out = (C.c_double * 3)()
        rc = self._lib.orc_trace_one(self._h, C.byref(settings), px, py, sample, out)
        assert rc == 0
        return np.array(out[:])

    def first_hit_sums(self, settings, width, height, sample_begin, sample_end):
        """Per-pixel sums of orc_first_hit over the samples: (normal [h][w][4] with the hit count in w,
        albedo [h][w][4])."""
        normal = np.zeros((height, width, 4))
        albedo = np.zeros((height, width, 4))
        n, a = (C.c_double * 3)(), (C.c_double * 3)()
        for y in range(height):
            for x in range(width):
                for s in range(sample_begin, sample_end):
                    rc = self._lib.orc_first_hit(self._h, C.byref(settings), x, y, s, n, a)
                    assert rc >= 0
                    if rc:
                        normal[y, x, :3] += n[:]
                        normal[y, x, 3] += 1
                        albedo[y, x, :3] += a[:]
        return normal, albedo

    def intersect(self, org_dir, with_margin=False):
        rays = np.ascontiguousarray(org_dir, dtype=np.float64).reshape(-1, 6)
        n = rays.shape[0]
        prim = np.empty(n, dtype=np.int32)
        t = np.empty(n, dtype=np.float64)
        if with_margin:
            margin = np.empty(n, dtype=np.float64)
            rc = self._lib.orc_intersect_margin(self._h, rays.ctypes.data, n, prim.ctypes.data, t.ctypes.data,
                                                margin.ctypes.data)
            assert rc == 0
            return prim, t, margin
        rc = self._lib.orc_intersect(self._h, rays.ctypes.data, n, prim.ctypes.data, t.ctypes.data)
        assert rc == 0
        return prim, t

    def occluded(self, org_dir_tmax):
        rays = np.ascontiguousarray(org_dir_tmax, dtype=np.float64).reshape(-1, 7)
        n = rays.shape[0]
        hit = np.empty(n, dtype=np.uint8)
        rc = self._lib.orc_occluded(self._h, rays.ctypes.data, n, hit.ctypes.data)
        assert rc == 0
        return hit
