"""Diagnostic (GPU box): when the pool cannot hold every sample, is it faster to let slots regenerate
paths (one pass) or to render the sample range in rounds of one sample per slot?"""
import os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import bench, torch
W, H, SPP, K = 1920, 1080, 256, int(sys.argv[1]) if len(sys.argv) > 1 else 64
scene, rs, camera, mesh = bench.load_buddha(W, H)
rr = bench.rr
ctx = rr.Context(0)
ctx.upload_scene(scene); ctx.set_camera(camera); ctx.film_alloc(W, H)
ctx.set_option(rr.OPT_MAX_SLOTS, W * H * K + (1 << 20))
s = rr.make_settings(W, H, rs.depth_limit, rs.clamp, seed=0)
for name in ("regenerate", "rounds", "regenerate", "rounds"):
    ctx.film_clear(); ctx.synchronize()
    t0 = time.perf_counter()
    if name == "regenerate":
        ctx.render(s, 0, SPP)
    else:
        for b in range(0, SPP, K):
            ctx.render(s, b, b + K)
    ctx.synchronize()
    dt = time.perf_counter() - t0
    print("K=%d %-10s %.1f ms  %.2f G samples/s" % (K, name, dt * 1e3, W * H * SPP / dt * 1e-9), flush=True)
