#!/usr/bin/env python3
"""Runs on the GPU box: the bench frame as one pass, against its left and right halves rendered at the same time by two
contexts on two streams with a phase offset (diagnostic: how much of a pass's late, thin iterations another pass's
heavy kernels can hide)."""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

rr = bench.rr
W, H, SPP, FRAMES = 1920, 1080, 256, 6
scene, rs, camera, _ = bench.load_buddha(W, H)


def make_ctx():
    c = rr.Context(0)
    c.upload_scene(scene)
    c.set_camera(camera)
    c.film_alloc(W, H)
    return c


full = rr.make_settings(W, H, rs.depth_limit, rs.clamp, seed=0)
a = make_ctx()
for _ in range(2):
    a.render(full, 0, SPP)
a.synchronize()
t0 = time.perf_counter()
for _ in range(FRAMES):
    a.render(full, 0, SPP)
a.synchronize()
single = (time.perf_counter() - t0) / FRAMES * 1e3
print("one pass per frame: %.2f ms" % single, flush=True)

b = make_ctx()
parts = int(os.environ.get("PARTS", "2"))
for offset_ms in (0.0, 10.0, 20.0, 30.0):
    def worker(ctx, which, delay):
        time.sleep(delay)
        for _ in range(FRAMES):
            for k in range(which, parts, 2):
                x0, x1 = W * k // parts, W * (k + 1) // parts
                s = rr.make_settings(W, H, rs.depth_limit, rs.clamp, seed=0, crop=(x0, 0, x1, H))
                ctx.render(s, 0, SPP)
        ctx.synchronize()
    for c in (a, b):
        c.synchronize()
    ta = threading.Thread(target=worker, args=(a, 0, 0.0))
    tb = threading.Thread(target=worker, args=(b, 1, offset_ms * 1e-3))
    t0 = time.perf_counter()
    ta.start(); tb.start(); ta.join(); tb.join()
    both = (time.perf_counter() - t0) / FRAMES * 1e3
    print("%d parts on two streams, second delayed %4.1f ms: %.2f ms per frame (%.1f %% of one pass)" % (parts, offset_ms, both, 100.0 * both / single), flush=True)
