"""Diagnostic (GPU box): duration of the closest-hit kernel for small numbers of incoherent rays on the
bench scene -- with few rays in flight the kernel time is the latency of its longest dependent chain.
Run under rocprofv3 --kernel-trace; prints nothing itself."""
import os, sys
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import bench
scene, rs, camera, mesh = bench.load_buddha(640, 360)
ctx = bench.rr.Context(0)
ctx.upload_scene(scene)
rng = np.random.default_rng(1)
for n in (64, 64, 1024, 16384, 262144, 4194304):
    # origins near the statue's surface region, directions uniform on the sphere
    o = rng.uniform([-1.0, 0.2, -1.0], [1.0, 5.5, 1.0], size=(n, 3))
    d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    prim, t = ctx.intersect(np.concatenate([o, d], axis=1))
    print(n, "hit fraction %.2f" % (prim >= 0).mean(), flush=True)
