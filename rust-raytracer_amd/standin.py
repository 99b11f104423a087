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


# ---------------------------------------------------------------- a harder stand-in
# The lathe above is genus 0 and star-shaped about its axis at every height: a ray that leaves it
# rarely comes back.  The scanned statue is not like that (genus 104, deep folds between robe and
# body).  This variant keeps the frame, the height and the triangle count and adds what makes paths
# long: tiers of robe whose profile overhangs (the meridian curve runs up, back down under the ledge
# above and up again, so every tier is an undercut groove), two arm rings and a halo with real
# through-holes, a necklace and a belt of small rings, and triangles emitted in a hashed order
# instead of row by row.  Still closed-form and the same on every machine.
_HARD_PARTS = (("body", 806, 420), ("arm", 300, 120), ("arm", 300, 120), ("halo", 402, 99), ("beads", 58 * 40, 40),
               ("knot", 37, 20))  # sum of u * v = 543,858 quads = 1,087,716 triangles


def _surface_tiered(theta, v):
    """The lathe's surface with 12 overhanging robe tiers between the pedestal and the chest."""
    p = _surface(theta, v)
    window = np.clip((v - 0.10) / 0.05, 0.0, 1.0) * np.clip((0.66 - v) / 0.05, 0.0, 1.0)
    window = window * window * (3.0 - 2.0 * window)
    phase = 2.0 * np.pi * 12.0 * v + 2.0 * np.sin(3.0 * theta)          # the tiers sag and rise around the figure
    # meridian = (radius, height) + (a_r cos, a_y sin): with the height advancing b = 6.4 / (24 pi) = 0.085 per
    # radian of phase, a_y = 0.2 lies between b (from where the curve turns back down: an overhang) and 4.6 b
    # (from where it would cross itself)
    radial = 1.0 + window * 0.10 * np.cos(phase) / np.maximum(np.hypot(p[..., 0], p[..., 2] + 0.0), 1e-6)
    out = np.empty_like(p)
    out[..., 0] = p[..., 0] * radial
    out[..., 2] = p[..., 2] * radial
    out[..., 1] = p[..., 1] + window * 0.20 * np.sin(phase)
    return out


def _lathe(surface, n_theta, n_v):
    theta = (np.arange(n_theta, dtype=np.float64) * (2.0 * np.pi / n_theta))[None, :]
    v = (np.arange(n_v + 1, dtype=np.float64) / n_v)[:, None]
    p = surface(theta, v)
    h = 1e-5
    dpt = (surface(theta + h, v) - surface(theta - h, v)) / (2 * h)
    vlo, vhi = np.clip(v - h, 0.0, 1.0), np.clip(v + h, 0.0, 1.0)
    dpv = (surface(theta, vhi) - surface(theta, vlo)) / (vhi - vlo)[..., None]
    n = np.cross(dpv, dpt)
    length = np.linalg.norm(n, axis=-1, keepdims=True)
    n = np.where(length > 1e-20, n / np.maximum(length, 1e-20), np.array([0.0, 1.0, 0.0]))
    return _quads(p.astype(np.float32), n.astype(np.float32), wrap_rows=False)


def _quads(p, n, wrap_rows):
    """Two triangles per cell of a grid of points [rows][cols][3] that wraps around in its columns (and, for a
    torus, in its rows as well; a lathe has one more row of points than of cells)."""
    j0 = np.arange(p.shape[1])
    j1 = (j0 + 1) % p.shape[1]
    lo, hi = (p, np.roll(p, -1, axis=0)) if wrap_rows else (p[:-1], p[1:])
    nlo, nhi = (n, np.roll(n, -1, axis=0)) if wrap_rows else (n[:-1], n[1:])
    a, b, c, d = lo[:, j0], lo[:, j1], hi[:, j1], hi[:, j0]
    na, nb, nc, nd = nlo[:, j0], nlo[:, j1], nhi[:, j1], nhi[:, j0]
    pos = np.stack([np.stack([a, d, c], axis=2), np.stack([a, c, b], axis=2)], axis=2)
    nrm = np.stack([np.stack([na, nd, nc], axis=2), np.stack([na, nc, nb], axis=2)], axis=2)
    return pos.reshape(-1, 3, 3), nrm.reshape(-1, 3, 3)


def _torus(centre, e1, e2, major, minor, n_u, n_v):
    """A ring around `centre` in the plane of the unit vectors e1, e2: n_u cells around, n_v around the tube."""
    e1, e2 = np.asarray(e1, dtype=np.float64), np.asarray(e2, dtype=np.float64)
    e3 = np.cross(e1, e2)
    alpha = (np.arange(n_u, dtype=np.float64) * (2.0 * np.pi / n_u))[None, :, None]
    beta = (np.arange(n_v, dtype=np.float64) * (2.0 * np.pi / n_v))[:, None, None]
    ring = np.cos(alpha) * e1 + np.sin(alpha) * e2
    n = np.cos(beta) * ring + np.sin(beta) * e3
    p = np.asarray(centre, dtype=np.float64) + major * ring + minor * n
    # rows = around the tube; the winding that makes (a, d, c) counter-clockwise seen from outside
    return _quads(p.astype(np.float32)[::-1], n.astype(np.float32)[::-1], wrap_rows=True)


def happy_buddha_hard(div=1):
    """(positions, normals), float32 [n][3][3]: the harder stand-in, 1,087,716 triangles at div = 1 (every grid
    dimension is divided by `div` for the small copies the tests use)."""
    def dim(k):
        return max(8, k // div)
    parts = []
    arms = 0
    for kind, n_u, n_v in _HARD_PARTS:
        if kind == "body":
            parts.append(_lathe(_surface_tiered, dim(n_u), dim(n_v)))
        elif kind == "arm":
            side = 1.0 if arms == 0 else -1.0
            arms += 1
            parts.append(_torus((side * 1.22, 4.55, 0.05), (1.0, 0.0, 0.0), (0.0, 1.0, 0.0), 0.55, 0.17, dim(n_u), dim(n_v)))
        elif kind == "halo":
            parts.append(_torus((0.0, 5.75, 0.62), (1.0, 0.0, 0.0), (0.0, 1.0, 0.0), 0.95, 0.055, dim(n_u), dim(n_v)))
        elif kind == "knot":
            parts.append(_torus((0.0, 6.42, -0.22), (1.0, 0.0, 0.0), (0.0, 0.0, 1.0), 0.10, 0.04, dim(n_u), dim(n_v)))
        else:
            # 58 rings of 40 x 40 cells: 22 around the neck, 36 around the waist
            n_rings = 58
            per = n_u // n_rings
            for k in range(n_rings):
                if k < 22:
                    a, ring_r, y, zc = 2.0 * np.pi * k / 22, 0.62, 5.05, -0.10
                else:
                    a, ring_r, y, zc = 2.0 * np.pi * (k - 22) / 36, 1.50, 3.10, 0.0
                centre = (ring_r * np.cos(a), y + 0.03 * np.sin(5.0 * a), zc + 0.85 * ring_r * np.sin(a))
                radial = np.array([np.cos(a), 0.0, np.sin(a)])
                up = np.array([0.0, 1.0, 0.0])
                # alternate links lie flat and stand upright, like a chain
                e1, e2 = (radial, up) if k % 2 == 0 else (radial, np.cross(up, radial))
                parts.append(_torus(centre, e1, e2, 0.085, 0.03, dim(per), dim(n_v)))
    pos = np.concatenate([p for p, _ in parts])
    nrm = np.concatenate([n for _, n in parts])
    # hashed order: triangle i of the output is triangle (i * 2654435761) mod n of the construction order
    # (the multiplier is prime and larger than n, so this is a permutation)
    n = len(pos)
    perm = ((np.arange(n, dtype=np.uint64) * np.uint64(2654435761)) % np.uint64(n)).astype(np.int64)
    return np.ascontiguousarray(pos[perm]), np.ascontiguousarray(nrm[perm])


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
