"""Diagnostic (GPU box): which light carries the device/oracle difference on the buddha scene."""
import os, sys, ctypes as C, copy
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(root, "tests")); sys.path.insert(0, root)
import importlib.util
spec = importlib.util.spec_from_file_location("rust_raytracer_amd", os.path.join(root, "rust-raytracer_amd", "__init__.py"))
rrt = importlib.util.module_from_spec(spec); sys.modules["rust_raytracer_amd"] = rrt; spec.loader.exec_module(rrt)
from oracle import binding as ob
import scenes_util
scene, rs = scenes_util.buddha_scene(rrt, 60, 40)
w, h = 160, 90
cam = rrt.Camera.from_settings(rs, w, h)
d = scene.desc()
n = d.n_lights
LightArr = type(d.lights.contents) * n
orig = LightArr.from_address(C.addressof(d.lights.contents))
print("lights:", [(orig[i].kind, list(orig[i].intensity)) for i in range(n)])
ctx = rrt.Context(0)
for keep in list(range(n)) + [-1]:
    arr = LightArr()
    for i in range(n):
        C.memmove(C.byref(arr[i]), C.byref(orig[i]), C.sizeof(arr[i]))
        if keep >= 0 and i != keep:
            for k in range(3): arr[i].intensity[k] = 0.0
    d2 = type(d)()
    C.memmove(C.byref(d2), C.byref(d), C.sizeof(d))
    d2.lights = C.cast(arr, type(d.lights))
    d2._owner = (scene, arr)
    ctx.upload_scene_desc(d2) if hasattr(ctx, "upload_scene_desc") else ctx.upload_scene(d2)
    ctx.set_camera(cam); ctx.film_alloc(w, h)
    orc = ob.Oracle(d2); orc.set_camera(cam)
    s = rrt.make_settings(w, h, 1, rs.clamp, seed=9)
    ctx.film_clear(); ctx.render(s, 0, 1)
    dev = ctx.film_download().astype(np.float64)
    ref, st = orc.render(s, 0, 1, threads=8)
    bad = ~np.all(np.abs(dev[..., :3] - ref[..., :3]) <= 1e-3 * np.abs(ref[..., :3]) + 1e-4, axis=-1)
    print("keep light", keep, "bad", bad.sum(), " dev mean %.5f ref mean %.5f" % (dev[..., :3].mean(), ref[..., :3].mean()))
# light 0 only: what do the differing pixels look at?
arr = LightArr()
for i in range(n):
    C.memmove(C.byref(arr[i]), C.byref(orig[i]), C.sizeof(arr[i]))
    if i != 0:
        for k in range(3): arr[i].intensity[k] = 0.0
d2 = type(d)(); C.memmove(C.byref(d2), C.byref(d), C.sizeof(d)); d2.lights = C.cast(arr, type(d.lights))
ctx.upload_scene(d2); ctx.set_camera(cam); ctx.film_alloc(w, h)
orc = ob.Oracle(d2); orc.set_camera(cam)
s = rrt.make_settings(w, h, 1, rs.clamp, seed=9)
ctx.film_clear(); ctx.render(s, 0, 1)
dev = ctx.film_download().astype(np.float64)
ref, st = orc.render(s, 0, 1, threads=8)
bad = ~np.all(np.abs(dev[..., :3] - ref[..., :3]) <= 1e-3 * np.abs(ref[..., :3]) + 1e-4, axis=-1)
ys, xs = np.nonzero(bad)
rays = np.array([np.concatenate(cam.generate_ray(x + 0.5, y + 0.5)) for x, y in zip(xs, ys)])
prim, t = ctx.intersect(rays)
nt = d.n_tris
print("prims under bad pixels (n_tris = %d):" % nt, np.unique(prim, return_counts=True))
for k in range(min(10, len(xs))):
    print("  px", xs[k], ys[k], "prim", prim[k], "t %.3f" % t[k], "dev", np.round(dev[ys[k], xs[k], :3], 4), "ref", np.round(ref[ys[k], xs[k], :3], 4))
allrays = np.array([np.concatenate(cam.generate_ray(x + 0.5, y + 0.5)) for y in range(h) for x in range(w)])
pa, ta = ctx.intersect(allrays)
pa = pa.reshape(h, w)
for code in range(nt, nt + 3):
    m = pa == code
    print("quad", code - nt, "pixels", m.sum(), "bad among them", (bad & m).sum(), " mean dev %.4f ref %.4f" % (dev[m][:, 0].mean() if m.any() else 0, ref[m][:, 0].mean() if m.any() else 0))
