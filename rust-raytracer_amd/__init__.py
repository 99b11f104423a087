"""ctypes access to the MI355X path-tracing core.

Three native libraries sit behind this module:

* ``librrt_hip.so``  -- the product: HIP kernels behind the C ABI of ``include/rrt.h``;
* ``librrt_host.so`` -- the C++ host mirror of the reference's loader / Camera / Film
  (``include/rrt_host.h``);
* nothing else.  The CPU oracle lives in ``oracle/`` and is loaded only by tests,
  ``__graft_entry__.smoke()`` and ``bench.py``'s CPU baseline through ``oracle/binding.py``.

There is no CPU fallback: every compute call goes to the HIP library and fails
with ``RrtError`` if that library or a GPU is missing.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SCENES = os.path.join(ROOT, "scenes")


class RrtError(RuntimeError):
    pass


# --------------------------------------------------------------------------- rrt.h structs
class Material(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("_pad", C.c_uint32), ("kd", C.c_double * 3), ("ks", C.c_double * 3),
                ("roughness", C.c_double), ("ior", C.c_double)]


class Light(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("_pad", C.c_uint32), ("intensity", C.c_double * 3), ("v0", C.c_double * 3),
                ("v1", C.c_double * 3), ("v2", C.c_double * 3)]


class Rect(C.Structure):
    _fields_ = [("pos", C.c_double * 3), ("side_a", C.c_double * 3), ("side_b", C.c_double * 3),
                ("material", C.c_uint32), ("light", C.c_int32)]


class Plane(C.Structure):
    _fields_ = [("pos", C.c_double * 3), ("normal", C.c_double * 3), ("material", C.c_uint32), ("_pad", C.c_uint32)]


class Sphere(C.Structure):
    _fields_ = [("center", C.c_double * 3), ("radius", C.c_double), ("material", C.c_uint32), ("_pad", C.c_uint32)]


class SceneDesc(C.Structure):
    _fields_ = [("n_tris", C.c_uint32), ("n_rects", C.c_uint32), ("n_planes", C.c_uint32), ("n_spheres", C.c_uint32),
                ("n_materials", C.c_uint32), ("n_lights", C.c_uint32),
                ("tri_pos", C.POINTER(C.c_float)), ("tri_nrm", C.POINTER(C.c_float)),
                ("tri_mat", C.POINTER(C.c_uint32)), ("rects", C.POINTER(Rect)), ("planes", C.POINTER(Plane)),
                ("spheres", C.POINTER(Sphere)), ("materials", C.POINTER(Material)), ("lights", C.POINTER(Light)),
                ("env_rgb8", C.POINTER(C.c_uint8)), ("env_w", C.c_uint32), ("env_h", C.c_uint32),
                ("rect_aabb_mode", C.c_uint32), ("_pad", C.c_uint32)]


class Settings(C.Structure):
    """rrt_settings: Settings of src/renderer.rs:24-30 plus film geometry and the RNG seed."""
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("crop_x0", C.c_uint32), ("crop_y0", C.c_uint32),
                ("crop_x1", C.c_uint32), ("crop_y1", C.c_uint32), ("depth_limit", C.c_uint32), ("_pad", C.c_uint32),
                ("clamp", C.c_double), ("seed", C.c_uint64)]


class Stats(C.Structure):
    _fields_ = [("paths", C.c_uint64), ("rays_extend", C.c_uint64), ("rays_shadow", C.c_uint64),
                ("path_bounces", C.c_uint64), ("ms_raygen", C.c_double), ("ms_extend", C.c_double),
                ("ms_shade", C.c_double), ("ms_connect", C.c_double), ("ms_film", C.c_double),
                ("launches_extend", C.c_uint64), ("nodes_visited", C.c_uint64), ("tris_tested", C.c_uint64),
                ("bvh_nodes", C.c_uint64), ("bvh_bytes", C.c_uint64),
                ("nodes_by_kernel", C.c_uint64 * 3), ("tris_by_kernel", C.c_uint64 * 3),
                ("ms_primary", C.c_double), ("launches_primary", C.c_uint64), ("rays_primary", C.c_uint64),
                ("ms_tail", C.c_double), ("launches_tail", C.c_uint64),
                ("rays_entered_by_kernel", C.c_uint64 * 3)]


class RenderSettingsFile(C.Structure):
    """rrth_render_settings: render_settings.yaml as main() reads it (src/main.rs:252-296)."""
    _fields_ = [("threads", C.c_uint32), ("depth_limit", C.c_uint32), ("max_samples", C.c_uint32),
                ("image_width", C.c_uint32), ("image_height", C.c_uint32), ("bucket_width", C.c_uint32),
                ("bucket_height", C.c_uint32), ("crop_x0", C.c_uint32), ("crop_y0", C.c_uint32),
                ("crop_x1", C.c_uint32), ("crop_y1", C.c_uint32), ("filter_method", C.c_uint32),
                ("denoise", C.c_uint32), ("has_focal_distance", C.c_uint32), ("clamp", C.c_double),
                ("filter_radius", C.c_double), ("camera_position", C.c_double * 3), ("camera_target", C.c_double * 3),
                ("fov", C.c_double), ("aperture", C.c_double), ("focal_distance", C.c_double),
                ("window_scale", C.c_double)]


MAT_MATTE, MAT_PLASTIC, MAT_MIRROR, MAT_GLASS = 0, 1, 2, 3
LIGHT_POINT, LIGHT_AREA, LIGHT_DISTANT, LIGHT_INFINITE = 0, 1, 2, 3
RECT_AABB_STRICT, RECT_AABB_INCLUSIVE, RECT_AABB_OFF = 0, 1, 2
OPT_TIMING, OPT_BATCH_SAMPLES, OPT_MAX_SLOTS, OPT_COUNT_TRAVERSAL, OPT_SIDE_STREAM, OPT_AOVS, OPT_TAIL_PATHS, OPT_DEVICE_BVH = 1, 2, 3, 4, 5, 6, 7, 8
FILTER_NONE, FILTER_GAUSSIAN, FILTER_MITCHELL = 0, 1, 2
NOALLOC_MAX_RAYS = 65536  # RRT_NOALLOC_MAX_RAYS

# every symbol include/rrt.h and include/rrt_host.h declare
RRT_SYMBOLS = ["rrt_device_count", "rrt_create", "rrt_destroy", "rrt_set_stream", "rrt_upload_scene",
               "rrt_set_camera", "rrt_film_alloc", "rrt_film_bind", "rrt_film_clear", "rrt_film_download",
               "rrt_film_device_ptr", "rrt_render", "rrt_synchronize", "rrt_film_reduce", "rrt_intersect",
               "rrt_occluded", "rrt_set_option", "rrt_get_stats", "rrt_reset_stats", "rrt_last_error",
               "rrt_set_filter", "rrt_film_tonemap", "rrt_bvh_selfcheck", "rrt_aov_download", "rrt_aov_device_ptr",
               "rrt_intersect_noalloc", "rrt_bvh_check_uploaded"]
RRTH_SYMBOLS = ["rrth_last_error", "rrth_scene_load", "rrth_scene_load_with_mesh", "rrth_scene_free",
                "rrth_scene_desc", "rrth_scene_set_rect_aabb_mode", "rrth_scene_set_material",
                "rrth_load_render_settings", "rrth_camera_matrices", "rrth_camera_generate_ray", "rrth_tonemap",
                "rrth_render_folder", "rrth_write_png", "rrth_render_folder_first_hits"]

_hip = None
_host = None


def hip_lib():
    """librrt_hip.so.  Raises RrtError when it has not been built: there is no other compute path."""
    global _hip
    if _hip is None:
        # RRT_HIP_LIB selects another build of the same library (A/B timing of kernel variants)
        path = os.environ.get("RRT_HIP_LIB") or os.path.join(HERE, "librrt_hip.so")
        if not os.path.exists(path):
            raise RrtError("librrt_hip.so is missing: run `python rust-raytracer_amd/build.py`")
        lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
        lib.rrt_create.restype = C.c_void_p
        lib.rrt_create.argtypes = [C.c_int]
        lib.rrt_destroy.argtypes = [C.c_void_p]
        lib.rrt_destroy.restype = None
        lib.rrt_set_stream.argtypes = [C.c_void_p, C.c_void_p]
        lib.rrt_upload_scene.argtypes = [C.c_void_p, C.POINTER(SceneDesc)]
        lib.rrt_set_camera.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_double, C.c_double]
        lib.rrt_film_alloc.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
        lib.rrt_film_bind.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32]
        lib.rrt_film_clear.argtypes = [C.c_void_p]
        lib.rrt_film_download.argtypes = [C.c_void_p, C.c_void_p]
        lib.rrt_film_device_ptr.argtypes = [C.c_void_p]
        lib.rrt_film_device_ptr.restype = C.c_void_p
        lib.rrt_set_filter.argtypes = [C.c_void_p, C.c_uint32, C.c_double]
        lib.rrt_film_tonemap.argtypes = [C.c_void_p, C.c_void_p]
        lib.rrt_aov_download.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        lib.rrt_aov_device_ptr.argtypes = [C.c_void_p]
        lib.rrt_aov_device_ptr.restype = C.c_void_p
        lib.rrt_render.argtypes = [C.c_void_p, C.POINTER(Settings), C.c_uint32, C.c_uint32]
        lib.rrt_synchronize.argtypes = [C.c_void_p]
        lib.rrt_film_reduce.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int]
        lib.rrt_intersect.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        lib.rrt_intersect_noalloc.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        lib.rrt_occluded.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        lib.rrt_set_option.argtypes = [C.c_void_p, C.c_int, C.c_int64]
        lib.rrt_get_stats.argtypes = [C.c_void_p, C.POINTER(Stats)]
        lib.rrt_reset_stats.argtypes = [C.c_void_p]
        lib.rrt_last_error.argtypes = [C.c_void_p]
        lib.rrt_last_error.restype = C.c_char_p
        lib.rrt_bvh_selfcheck.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        lib.rrt_bvh_selfcheck.restype = C.c_int64
        lib.rrt_bvh_check_uploaded.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                                               C.POINTER(C.c_int), C.POINTER(C.c_double)]
        lib.rrt_bvh_check_uploaded.restype = C.c_int64
        _hip = lib
    return _hip


def host_lib():
    global _host
    if _host is None:
        hip_lib()  # librrt_host.so links against it
        path = os.path.join(HERE, "librrt_host.so")
        if not os.path.exists(path):
            raise RrtError("librrt_host.so is missing: run `python rust-raytracer_amd/build.py`")
        lib = C.CDLL(path)
        lib.rrth_last_error.restype = C.c_char_p
        lib.rrth_scene_load.restype = C.c_void_p
        lib.rrth_scene_load.argtypes = [C.c_char_p]
        lib.rrth_scene_load_with_mesh.restype = C.c_void_p
        lib.rrth_scene_load_with_mesh.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(Material)]
        lib.rrth_scene_free.argtypes = [C.c_void_p]
        lib.rrth_scene_free.restype = None
        lib.rrth_scene_desc.argtypes = [C.c_void_p, C.POINTER(SceneDesc)]
        lib.rrth_scene_set_rect_aabb_mode.argtypes = [C.c_void_p, C.c_uint32]
        lib.rrth_scene_set_material.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(Material)]
        lib.rrth_load_render_settings.argtypes = [C.c_char_p, C.POINTER(RenderSettingsFile)]
        dp = C.POINTER(C.c_double)
        lib.rrth_camera_matrices.argtypes = [dp, dp, C.c_double, C.c_double, C.c_int, C.c_double, C.c_uint32,
                                             C.c_uint32, dp, dp, dp]
        lib.rrth_camera_generate_ray.argtypes = [dp, dp, C.c_double, C.c_double, C.c_int, C.c_double, C.c_uint32,
                                                 C.c_uint32, C.c_double, C.c_double, C.c_double, C.c_double, dp, dp]
        lib.rrth_tonemap.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
        lib.rrth_write_png.argtypes = [C.c_char_p, C.c_void_p, C.c_uint32, C.c_uint32]
        lib.rrth_render_folder.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.c_int, C.c_uint64, C.c_int, C.c_char_p,
                                           C.POINTER(C.c_double)]
        lib.rrth_render_folder_first_hits.argtypes = lib.rrth_render_folder.argtypes + [C.c_void_p, C.c_void_p]
        _host = lib
    return _host


def _vec3(v):
    return (C.c_double * 3)(*[float(x) for x in v])


def make_material(kind, kd=(0, 0, 0), ks=(0, 0, 0), roughness=0.0, ior=1.5):
    m = Material()
    m.kind = kind
    m.kd = _vec3(kd)
    m.ks = _vec3(ks)
    m.roughness = roughness
    m.ior = ior
    return m


# --------------------------------------------------------------------------- host mirror
class Scene:
    """Scene::load_from_folder (src/scene.rs:55): a loaded, flattened scene."""

    def __init__(self, handle):
        self._h = handle
        self._desc = None

    @classmethod
    def load_from_folder(cls, folder):
        lib = host_lib()
        h = lib.rrth_scene_load(os.fsencode(folder))
        if not h:
            raise RrtError(lib.rrth_last_error().decode())
        return cls(h)

    @classmethod
    def load_from_folder_with_mesh(cls, folder, positions, normals, material):
        """Lights / environment from the folder, triangles from arrays [n][3][3] (float32)."""
        lib = host_lib()
        pos = np.ascontiguousarray(positions, dtype=np.float32)
        nrm = np.ascontiguousarray(normals, dtype=np.float32)
        assert pos.shape == nrm.shape and pos.ndim == 3 and pos.shape[1:] == (3, 3)
        h = lib.rrth_scene_load_with_mesh(os.fsencode(folder), pos.ctypes.data, nrm.ctypes.data, pos.shape[0],
                                          C.byref(material))
        if not h:
            raise RrtError(lib.rrth_last_error().decode())
        return cls(h)

    def desc(self):
        d = SceneDesc()
        if host_lib().rrth_scene_desc(self._h, C.byref(d)) != 0:
            raise RrtError(host_lib().rrth_last_error().decode())
        d._owner = self  # the pointers in `d` stay valid only while the scene lives
        return d

    def set_rect_aabb_mode(self, mode):
        if host_lib().rrth_scene_set_rect_aabb_mode(self._h, mode) != 0:
            raise RrtError(host_lib().rrth_last_error().decode())

    def set_material(self, index, material):
        if host_lib().rrth_scene_set_material(self._h, index, C.byref(material)) != 0:
            raise RrtError(host_lib().rrth_last_error().decode())

    def close(self):
        if self._h:
            host_lib().rrth_scene_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class SceneArrays:
    """A scene assembled in Python: the same flat arrays Scene::load_from_folder produces, filled in
    by hand (for scenes that have no folder: tests, generated stress scenes)."""

    def __init__(self, tri_pos=None, tri_nrm=None, tri_mat=None, materials=(), lights=(), rects=(), planes=(),
                 spheres=(), env_rgb8=None, rect_aabb_mode=RECT_AABB_STRICT):
        n = 0 if tri_pos is None else len(tri_pos)
        self.tri_pos = np.ascontiguousarray(np.zeros((0, 3, 3)) if tri_pos is None else tri_pos, dtype=np.float32)
        self.tri_nrm = np.ascontiguousarray(np.zeros((0, 3, 3)) if tri_nrm is None else tri_nrm, dtype=np.float32)
        self.tri_mat = np.ascontiguousarray(np.zeros(n) if tri_mat is None else tri_mat, dtype=np.uint32)
        assert self.tri_pos.shape == (n, 3, 3) and self.tri_nrm.shape == (n, 3, 3) and self.tri_mat.shape == (n,)
        self.materials = (Material * max(1, len(materials)))(*materials)
        self.lights = (Light * max(1, len(lights)))(*lights)
        self.rects = (Rect * max(1, len(rects)))(*rects)
        self.planes = (Plane * max(1, len(planes)))(*planes)
        self.spheres = (Sphere * max(1, len(spheres)))(*spheres)
        self.counts = (n, len(rects), len(planes), len(spheres), len(materials), len(lights))
        self.env = None if env_rgb8 is None else np.ascontiguousarray(env_rgb8, dtype=np.uint8)
        self.rect_aabb_mode = rect_aabb_mode

    def desc(self):
        d = SceneDesc()
        d.n_tris, d.n_rects, d.n_planes, d.n_spheres, d.n_materials, d.n_lights = self.counts
        d.tri_pos = self.tri_pos.ctypes.data_as(C.POINTER(C.c_float))
        d.tri_nrm = self.tri_nrm.ctypes.data_as(C.POINTER(C.c_float))
        d.tri_mat = self.tri_mat.ctypes.data_as(C.POINTER(C.c_uint32))
        d.materials = C.cast(self.materials, C.POINTER(Material))
        d.lights = C.cast(self.lights, C.POINTER(Light))
        d.rects = C.cast(self.rects, C.POINTER(Rect))
        d.planes = C.cast(self.planes, C.POINTER(Plane))
        d.spheres = C.cast(self.spheres, C.POINTER(Sphere))
        if self.env is not None:
            d.env_rgb8 = self.env.ctypes.data_as(C.POINTER(C.c_uint8))
            d.env_h, d.env_w = self.env.shape[:2]
        d.rect_aabb_mode = self.rect_aabb_mode
        d._owner = self
        return d


def make_light(kind, intensity=(1, 1, 1), v0=(0, 0, 0), v1=(0, 0, 0), v2=(0, 0, 0)):
    l = Light()
    l.kind = kind
    l.intensity, l.v0, l.v1, l.v2 = _vec3(intensity), _vec3(v0), _vec3(v1), _vec3(v2)
    return l


def make_rect(pos, side_a, side_b, material, light=-1):
    r = Rect()
    r.pos, r.side_a, r.side_b = _vec3(pos), _vec3(side_a), _vec3(side_b)
    r.material, r.light = material, light
    return r


def make_plane(pos, normal, material):
    p = Plane()
    p.pos, p.normal, p.material = _vec3(pos), _vec3(normal), material
    return p


def make_sphere(center, radius, material):
    s = Sphere()
    s.center, s.radius, s.material = _vec3(center), float(radius), material
    return s


def load_render_settings(folder):
    r = RenderSettingsFile()
    if host_lib().rrth_load_render_settings(os.fsencode(folder), C.byref(r)) != 0:
        raise RrtError(host_lib().rrth_last_error().decode())
    return r


class Camera:
    """Camera::new (src/camera.rs:39-101): holds the two matrices the device needs."""

    def __init__(self, position, target, fov, aperture, focal_distance, image_width, image_height):
        self.position, self.target = tuple(position), tuple(target)
        self.fov, self.aperture = float(fov), float(aperture)
        self._fd_in = focal_distance
        self.image_width, self.image_height = int(image_width), int(image_height)
        r2c = (C.c_double * 16)()
        c2w = (C.c_double * 16)()
        fd = C.c_double()
        rc = host_lib().rrth_camera_matrices(_vec3(position), _vec3(target), self.fov, self.aperture,
                                             0 if focal_distance is None else 1,
                                             0.0 if focal_distance is None else float(focal_distance),
                                             self.image_width, self.image_height, r2c, c2w, C.byref(fd))
        if rc != 0:
            raise RrtError(host_lib().rrth_last_error().decode())
        self.raster_to_camera = np.array(r2c[:], dtype=np.float64)
        self.camera_to_world = np.array(c2w[:], dtype=np.float64)
        self.focal_distance = fd.value

    @classmethod
    def from_settings(cls, rs, width=None, height=None):
        return cls(rs.camera_position[:], rs.camera_target[:], rs.fov, rs.aperture,
                   rs.focal_distance if rs.has_focal_distance else None,
                   width or rs.image_width, height or rs.image_height)

    def generate_ray(self, film_x, film_y, lens=(0.0, 0.0)):
        o = (C.c_double * 3)()
        d = (C.c_double * 3)()
        rc = host_lib().rrth_camera_generate_ray(_vec3(self.position), _vec3(self.target), self.fov, self.aperture,
                                                 0 if self._fd_in is None else 1,
                                                 0.0 if self._fd_in is None else float(self._fd_in),
                                                 self.image_width, self.image_height, film_x, film_y, lens[0], lens[1],
                                                 o, d)
        if rc != 0:
            raise RrtError(host_lib().rrth_last_error().decode())
        return np.array(o[:]), np.array(d[:])


def make_settings(width, height, depth_limit, clamp, seed=0, crop=None):
    s = Settings()
    s.width, s.height = width, height
    x0, y0, x1, y1 = crop if crop is not None else (0, 0, width, height)
    s.crop_x0, s.crop_y0, s.crop_x1, s.crop_y1 = x0, y0, x1, y1
    s.depth_limit = depth_limit
    s.clamp = clamp
    s.seed = seed
    return s


def bvh_selfcheck(tri_pos):
    """Builds the hierarchy rrt_upload_scene would build for these triangles ([n][3][3] float32) on the
    host and checks its invariants: (violations, node count, depth).  Needs no GPU."""
    pos = np.ascontiguousarray(tri_pos, dtype=np.float32).reshape(-1, 3, 3)
    n_nodes, depth = C.c_uint32(0), C.c_uint32(0)
    bad = hip_lib().rrt_bvh_selfcheck(pos.ctypes.data, pos.shape[0], C.byref(n_nodes), C.byref(depth))
    if bad < 0:
        raise RrtError("rrt_bvh_selfcheck failed: %d" % bad)
    return int(bad), n_nodes.value, depth.value


def tonemap(sum_rgbw):
    """Film tone map (src/film.rs:233-249): float32 [h][w][4] accumulators -> uint8 [h][w][3]."""
    acc = np.ascontiguousarray(sum_rgbw, dtype=np.float32)
    h, w = acc.shape[:2]
    out = np.zeros((h, w, 3), dtype=np.uint8)
    if host_lib().rrth_tonemap(acc.ctypes.data, w, h, out.ctypes.data) != 0:
        raise RrtError(host_lib().rrth_last_error().decode())
    return out


# --------------------------------------------------------------------------- device context
def film_reduce(contexts, root=0):
    """rrt_film_reduce: sums the films (and first-hit buffers) of one context per GPU onto contexts[root]."""
    arr = (C.c_void_p * len(contexts))(*[c._h for c in contexts])
    rc = hip_lib().rrt_film_reduce(arr, len(contexts), root)
    if rc != 0:
        msg = hip_lib().rrt_last_error(contexts[root]._h if 0 <= root < len(contexts) else None)
        raise RrtError("rrt_film_reduce failed (%d): %s" % (rc, (msg or b"").decode()))


class Context:
    """One GPU's rrt_ctx (include/rrt.h)."""

    def __init__(self, device=0):
        self._lib = hip_lib()
        self._h = self._lib.rrt_create(device)
        if not self._h:
            raise RrtError("rrt_create: " + self._lib.rrt_last_error(None).decode())
        self._film_shape = None

    def _check(self, rc, what):
        if rc != 0:
            raise RrtError("%s failed (%d): %s" % (what, rc, self._lib.rrt_last_error(self._h).decode()))

    def close(self):
        if self._h:
            self._lib.rrt_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, stream_ptr):
        self._check(self._lib.rrt_set_stream(self._h, stream_ptr), "rrt_set_stream")

    def upload_scene(self, scene):
        d = scene.desc() if hasattr(scene, "desc") else scene
        self._check(self._lib.rrt_upload_scene(self._h, C.byref(d)), "rrt_upload_scene")

    def set_camera(self, camera):
        r2c = np.ascontiguousarray(camera.raster_to_camera, dtype=np.float64)
        c2w = np.ascontiguousarray(camera.camera_to_world, dtype=np.float64)
        dp = C.POINTER(C.c_double)
        self._check(self._lib.rrt_set_camera(self._h, r2c.ctypes.data_as(dp), c2w.ctypes.data_as(dp), camera.aperture,
                                             camera.focal_distance), "rrt_set_camera")

    def film_alloc(self, width, height):
        self._check(self._lib.rrt_film_alloc(self._h, width, height), "rrt_film_alloc")
        self._film_shape = (height, width, 4)

    def film_bind(self, device_ptr, width, height):
        self._check(self._lib.rrt_film_bind(self._h, device_ptr, width, height), "rrt_film_bind")
        self._film_shape = (height, width, 4)

    def film_clear(self):
        self._check(self._lib.rrt_film_clear(self._h), "rrt_film_clear")

    def film_download(self):
        out = np.empty(self._film_shape, dtype=np.float32)
        self._check(self._lib.rrt_film_download(self._h, out.ctypes.data), "rrt_film_download")
        return out

    def set_filter(self, method, radius=0.0):
        """Film reconstruction filter (src/film.rs:59-132): FILTER_NONE / FILTER_GAUSSIAN / FILTER_MITCHELL."""
        self._check(self._lib.rrt_set_filter(self._h, method, radius), "rrt_set_filter")

    def film_tonemap(self):
        """The film through the device tone map (src/film.rs:233-249): uint8 [h][w][3]."""
        out = np.empty(self._film_shape[:2] + (3,), dtype=np.uint8)
        self._check(self._lib.rrt_film_tonemap(self._h, out.ctypes.data), "rrt_film_tonemap")
        return out

    def aov_download(self):
        """First-hit sums (OPT_AOVS): normal [h][w][4] = (sum n, hits), albedo [h][w][4] = (sum rgb, 0)."""
        normal = np.empty(self._film_shape[:2] + (4,), dtype=np.float32)
        albedo = np.empty_like(normal)
        self._check(self._lib.rrt_aov_download(self._h, normal.ctypes.data, albedo.ctypes.data), "rrt_aov_download")
        return normal, albedo

    def render(self, settings, sample_begin, sample_end):
        self._check(self._lib.rrt_render(self._h, C.byref(settings), sample_begin, sample_end), "rrt_render")

    def synchronize(self):
        self._check(self._lib.rrt_synchronize(self._h), "rrt_synchronize")

    def intersect(self, org_dir):
        rays = np.ascontiguousarray(org_dir, dtype=np.float32).reshape(-1, 6)
        n = rays.shape[0]
        prim = np.empty(n, dtype=np.int32)
        t = np.empty(n, dtype=np.float32)
        self._check(self._lib.rrt_intersect(self._h, rays.ctypes.data, n, prim.ctypes.data, t.ctypes.data), "rrt_intersect")
        return prim, t

    def intersect_noalloc(self, org_dir):
        """rrt_intersect_noalloc: as intersect(), staged through the library's static device array (at most
        NOALLOC_MAX_RAYS rays)."""
        rays = np.ascontiguousarray(org_dir, dtype=np.float32).reshape(-1, 6)
        n = rays.shape[0]
        prim = np.empty(n, dtype=np.int32)
        t = np.empty(n, dtype=np.float32)
        self._check(self._lib.rrt_intersect_noalloc(self._h, rays.ctypes.data, n, prim.ctypes.data, t.ctypes.data),
                    "rrt_intersect_noalloc")
        return prim, t

    def occluded(self, org_dir_tmax):
        rays = np.ascontiguousarray(org_dir_tmax, dtype=np.float32).reshape(-1, 7)
        n = rays.shape[0]
        hit = np.empty(n, dtype=np.uint8)
        self._check(self._lib.rrt_occluded(self._h, rays.ctypes.data, n, hit.ctypes.data), "rrt_occluded")
        return hit

    def bvh_check(self, tri_pos):
        """rrt_bvh_check_uploaded: (violations, nodes, depth, built on the device?, device build seconds)."""
        pos = np.ascontiguousarray(tri_pos, dtype=np.float32).reshape(-1, 9)
        nodes, depth, dev, secs = C.c_uint32(), C.c_uint32(), C.c_int(), C.c_double()
        bad = self._lib.rrt_bvh_check_uploaded(self._h, pos.ctypes.data, pos.shape[0], C.byref(nodes), C.byref(depth),
                                               C.byref(dev), C.byref(secs))
        if bad < 0:
            self._check(int(bad), "rrt_bvh_check_uploaded")
        return int(bad), nodes.value, depth.value, bool(dev.value), secs.value

    def set_option(self, option, value):
        self._check(self._lib.rrt_set_option(self._h, option, int(value)), "rrt_set_option")

    def stats(self):
        s = Stats()
        self._check(self._lib.rrt_get_stats(self._h, C.byref(s)), "rrt_get_stats")
        return s

    def reset_stats(self):
        self._check(self._lib.rrt_reset_stats(self._h), "rrt_reset_stats")
