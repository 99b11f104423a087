"""Deterministic stand-in for scene/buddha/happy-buddha.obj.

The reference snapshot does not contain the Happy Buddha mesh (it is listed in
its .MISSING_LARGE_BLOBS), only the scene's lights, camera and MTL.  This
module generates a statue-like closed surface with exactly the Stanford
model's triangle count (1,087,716) and smooth per-vertex normals, framed the
way the scene's camera expects (camera at (3, 2, -8) looking at (0, 3.2, -0.5),
fov 60: src scene/buddha/render_settings.yaml): a figure about 6.4 units tall
standing at the origin.  Everything is closed-form (no random number
generator), so the same arrays come out on every machine.

If the real OBJ is dropped into scenes/buddha/ the loader uses it instead.
"""
import numpy as np

HAPPY_BUDDHA_TRIANGLES = 1_087_716
_N_THETA, _N_V = 966, 563  # 966 * 563 quads * 2 = 1,087,716 triangles


def _profile(v):
    """Radius of the undisplaced figure along its height, v in [0, 1]."""
    base = 1.25 * np.exp(-((v - 0.02) / 0.10) ** 2)          # pedestal
    legs = 0.95 * np.exp(-((v - 0.22) / 0.16) ** 2)
    belly = 1.20 * np.exp(-((v - 0.48) / 0.15) ** 2)
    chest = 0.85 * np.exp(-((v - 0.68) / 0.10) ** 2)
    head = 0.62 * np.exp(-((v - 0.88) / 0.07) ** 2)
    r = np.maximum.reduce([base, legs, belly, chest, head]) + 0.12
    # close the surface at both ends (a pin hole of radius 1e-3 remains)
    cap = np.sqrt(np.clip(v / 0.015, 0.0, 1.0)) * np.sqrt(np.clip((1.0 - v) / 0.03, 0.0, 1.0))
    return 1e-3 + r * cap


def _surface(theta, v):
    """Positions of the displaced surface; theta, v broadcastable arrays."""
    r = _profile(v)
    # arms / hands raised: two lobes either side at chest height
    arms = 0.45 * np.exp(-((v - 0.74) / 0.09) ** 2) * (np.exp(-((np.cos(theta - 0.9)) - 1.0) ** 2 / 0.08)
                                                       + np.exp(-((np.cos(theta + 0.9 - np.pi)) - 1.0) ** 2 / 0.08))
    # robe folds and fine surface detail: fixed sinusoid sums
    folds = 0.035 * np.sin(9.0 * theta + 14.0 * v + 0.7) * np.sin(23.0 * v + 1.3)
    folds += 0.020 * np.sin(17.0 * theta - 31.0 * v + 2.1)
    detail = 0.006 * np.sin(61.0 * theta + 1.9) * np.sin(173.0 * v + 0.4) + 0.004 * np.sin(97.0 * theta - 211.0 * v)
    cap = np.clip(v / 0.03, 0.0, 1.0) * np.clip((1.0 - v) / 0.04, 0.0, 1.0)
    rr = r + cap * (arms + folds + detail)
    height = 6.4
    # elliptical cross-section (deeper than wide at the belly), slight forward lean of the head
    x = 1.00 * rr * np.cos(theta)
    z = 0.85 * rr * np.sin(theta) - 0.25 * np.exp(-((v - 0.9) / 0.1) ** 2)
    y = height * v + 0.0 * theta
    return np.stack([x, y, z], axis=-1)


def happy_buddha_standin(n_theta=_N_THETA, n_v=_N_V):
    """Returns (positions, normals): float32 arrays [n_tris][3][3], n_tris = 2 * n_theta * n_v."""
    theta = (np.arange(n_theta, dtype=np.float64) * (2.0 * np.pi / n_theta))[None, :]
    v = (np.arange(n_v + 1, dtype=np.float64) / n_v)[:, None]
    p = _surface(theta, v)                                  # [n_v + 1][n_theta][3]
    # smooth normals from central differences of the parametrisation
    h = 1e-5
    dpt = (_surface(theta + h, v) - _surface(theta - h, v)) / (2 * h)
    vlo, vhi = np.clip(v - h, 0.0, 1.0), np.clip(v + h, 0.0, 1.0)
    dpv = (_surface(theta, vhi) - _surface(theta, vlo)) / (vhi - vlo)[..., None]
    n = np.cross(dpv, dpt)
    length = np.linalg.norm(n, axis=-1, keepdims=True)
    n = np.where(length > 1e-20, n / np.maximum(length, 1e-20), np.array([0.0, 1.0, 0.0]))
    p = p.astype(np.float32)
    n = n.astype(np.float32)
    j0 = np.arange(n_theta)
    j1 = (j0 + 1) % n_theta
    a, b = p[:-1][:, j0], p[:-1][:, j1]
    c, d = p[1:][:, j1], p[1:][:, j0]
    na, nb = n[:-1][:, j0], n[:-1][:, j1]
    nc, nd = n[1:][:, j1], n[1:][:, j0]
    # counter-clockwise quad (a, d, c, b) seen from outside -> fan triangles (a, d, c), (a, c, b), as a
    # triangulating OBJ loader would emit them
    pos = np.stack([np.stack([a, d, c], axis=2), np.stack([a, c, b], axis=2)], axis=2)
    nrm = np.stack([np.stack([na, nd, nc], axis=2), np.stack([na, nc, nb], axis=2)], axis=2)
    return np.ascontiguousarray(pos.reshape(-1, 3, 3)), np.ascontiguousarray(nrm.reshape(-1, 3, 3))


def write_obj(path, positions, normals):
    """Writes the mesh as an OBJ (v / vn per corner, no sharing) for tools that want a file."""
    with open(path, "w") as f:
        f.write("mtllib happy-buddha.mtl\no buddha_standin\nusemtl initialShadingGroup\n")
        for t in range(positions.shape[0]):
            for k in range(3):
                f.write("v %.6f %.6f %.6f\n" % tuple(positions[t, k]))
            for k in range(3):
                f.write("vn %.6f %.6f %.6f\n" % tuple(normals[t, k]))
            i = 3 * t + 1
            f.write("f %d//%d %d//%d %d//%d\n" % (i, i, i + 1, i + 1, i + 2, i + 2))
