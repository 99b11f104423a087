"""The many-object stress scene: copies of the cow, lamp and teapot models of the reference's scene
folder on a grid -- about 15.6 k triangles per trio, 3200 trios = 50 M triangles at full size.

The reference cannot load the cow and the lamp as they are (src/scene.rs:226-232 requires vertex
normals, which only the teapot file has), so they get area-weighted vertex normals here.  Each
mesh is scaled to a unit box first; trio i sits in cell i of a 16 x 16 x 13 grid (x fastest, then
z, then y) with a seeded jitter and a seeded rotation about the y axis.  Everything is Matte; one
distant light and one area light above the grid, a two-triangle floor below it; depth limit 16.
"""
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
MODELS = os.path.join(os.path.dirname(_HERE), "scenes", "models")
GRID = (16, 16, 13)  # cells in x, z, y


def load_obj(path):
    """Vertices, optional normals and fan-triangulated faces of a Wavefront OBJ (the subset the
    model files use: v, vn, f with v, v/vt, v//vn or v/vt/vn corners; polygons become fans (0, i, i+1),
    as the tobj loader of the reference does)."""
    verts, norms, tri_v, tri_n = [], [], [], []
    with open(path) as f:
        for line in f:
            t = line.split()
            if not t:
                continue
            if t[0] == "v":
                verts.append([float(x) for x in t[1:4]])
            elif t[0] == "vn":
                norms.append([float(x) for x in t[1:4]])
            elif t[0] == "f":
                corners = [c.split("/") for c in t[1:]]
                vi = [int(c[0]) - 1 if int(c[0]) > 0 else len(verts) + int(c[0]) for c in corners]
                has_n = all(len(c) == 3 and c[2] for c in corners)
                ni = [int(c[2]) - 1 if int(c[2]) > 0 else len(norms) + int(c[2]) for c in corners] if has_n else None
                for k in range(1, len(vi) - 1):
                    tri_v.append([vi[0], vi[k], vi[k + 1]])
                    if ni is not None:
                        tri_n.append([ni[0], ni[k], ni[k + 1]])
    verts = np.asarray(verts, dtype=np.float64)
    tri_v = np.asarray(tri_v, dtype=np.int64)
    if norms and len(tri_n) == len(tri_v):
        return verts, tri_v, np.asarray(norms, dtype=np.float64), np.asarray(tri_n, dtype=np.int64)
    return verts, tri_v, None, None


def smooth_normals(verts, tris):
    """Area-weighted vertex normals: the sum of the (unnormalised) face normals around a vertex."""
    p = verts[tris]
    fn = np.cross(p[:, 1] - p[:, 0], p[:, 2] - p[:, 0])
    vn = np.zeros_like(verts)
    for k in range(3):
        np.add.at(vn, tris[:, k], fn)
    length = np.linalg.norm(vn, axis=1, keepdims=True)
    return vn / np.where(length > 0, length, 1.0)


def _unit_mesh(name):
    """Triangle positions and normals [n][3][3] of one model, scaled into the unit box centred on the
    origin in x and z and standing on y = 0."""
    verts, tris, norms, ntris = load_obj(os.path.join(MODELS, name))
    lo, hi = verts.min(axis=0), verts.max(axis=0)
    scale = 1.0 / (hi - lo).max()
    centre = 0.5 * (lo + hi)
    v = (verts - np.array([centre[0], lo[1], centre[2]])) * scale
    p = v[tris]
    n = smooth_normals(v, tris)[tris] if norms is None else norms[ntris]
    # a corner without a usable normal (isolated or degenerate geometry) takes its face's, or straight up
    face = np.cross(p[:, 1] - p[:, 0], p[:, 2] - p[:, 0])
    flen = np.linalg.norm(face, axis=1, keepdims=True)
    face = np.where(flen > 0, face / np.where(flen > 0, flen, 1.0), np.array([0.0, 1.0, 0.0]))
    nlen = np.linalg.norm(n, axis=2, keepdims=True)
    n = np.where(nlen > 1e-12, n / np.where(nlen > 1e-12, nlen, 1.0), face[:, None, :])
    return p, n


def trio():
    """Cow, lamp and teapot side by side in one unit cell: positions and normals [n][3][3]."""
    parts = []
    for name, size, offset in (("cow.obj", 0.45, (-0.24, 0.0, -0.22)), ("lamp.obj", 0.45, (0.24, 0.0, -0.22)),
                               ("teapot_normals.obj", 0.45, (0.0, 0.0, 0.24))):
        p, n = _unit_mesh(name)
        parts.append((p * size + np.asarray(offset), n))
    return np.concatenate([p for p, _ in parts]), np.concatenate([n for _, n in parts])


def stress_mesh(n_trios=3200, seed=1, grid=GRID):
    """Positions and normals (float32 [n][3][3]) of n_trios trios in the first cells of the grid."""
    assert 1 <= n_trios <= grid[0] * grid[1] * grid[2]
    base_p, base_n = trio()
    rng = np.random.default_rng(seed)
    angle = rng.uniform(0.0, 2.0 * np.pi, n_trios)
    jitter = rng.uniform(-0.05, 0.05, (n_trios, 3)) * np.array([1.0, 0.0, 1.0])
    i = np.arange(n_trios)
    cell = np.stack([i % grid[0], i // (grid[0] * grid[1]), (i // grid[0]) % grid[1]], axis=1).astype(np.float64)
    origin = cell * np.array([1.0, 1.1, 1.0]) + jitter
    c, s = np.cos(angle), np.sin(angle)
    rot = np.zeros((n_trios, 3, 3))
    rot[:, 0, 0], rot[:, 0, 2], rot[:, 1, 1], rot[:, 2, 0], rot[:, 2, 2] = c, s, 1.0, -s, c
    nt = len(base_p)
    pos = np.empty((n_trios * nt, 3, 3), dtype=np.float32)
    nrm = np.empty((n_trios * nt, 3, 3), dtype=np.float32)
    step = 64  # trios per chunk: bounds the float64 temporaries
    for a in range(0, n_trios, step):
        b = min(a + step, n_trios)
        r = rot[a:b, None, None]                                       # [k][1][1][3][3]
        p = np.einsum("kabij,tvj->ktvi", r, base_p) + origin[a:b, None, None, :]
        n = np.einsum("kabij,tvj->ktvi", r, base_n)
        pos[a * nt:b * nt] = p.reshape(-1, 3, 3)
        nrm[a * nt:b * nt] = n.reshape(-1, 3, 3)
    return pos, nrm


def stress_scene(rrt, n_trios=3200, seed=1, grid=GRID):
    """(SceneArrays, camera arguments, depth_limit, clamp) of the stress workload."""
    pos, nrm = stress_mesh(n_trios, seed, grid)
    lo, hi = pos.reshape(-1, 3).min(axis=0).astype(np.float64), pos.reshape(-1, 3).max(axis=0).astype(np.float64)
    centre, size = 0.5 * (lo + hi), (hi - lo)
    # a floor under the lowest layer, two triangles, a cell wider than the grid on every side
    x0, x1, z0, z1 = lo[0] - 1.0, hi[0] + 1.0, lo[2] - 1.0, hi[2] + 1.0
    floor = np.array([[[x0, 0, z0], [x1, 0, z1], [x1, 0, z0]], [[x0, 0, z0], [x0, 0, z1], [x1, 0, z1]]], dtype=np.float32)
    pos = np.concatenate([pos, floor])
    nrm = np.concatenate([nrm, np.tile(np.array([0, 1, 0], dtype=np.float32), (2, 3, 1))])
    materials = [rrt.make_material(rrt.MAT_MATTE, kd=(0.65, 0.6, 0.5), roughness=0.0),
                 rrt.make_material(rrt.MAT_MATTE, kd=(0.9, 0.9, 0.9), roughness=20.0)]
    top = hi[1] + 0.5 * max(size[0], size[2]) + 1.0
    side = 0.6 * size[0] + 1.0  # a square: the reference takes |side_b|^2 for the area (objects/rectangle.rs:131-133)
    a, b = (side, 0.0, 0.0), (0.0, 0.0, side)  # a x b = (0, -side^2, 0): the light faces down at the grid
    corner = (centre[0] - 0.5 * side, top, centre[2] - 0.5 * side)
    lights = [rrt.make_light(rrt.LIGHT_DISTANT, intensity=(0.9, 0.9, 0.85), v0=(0.3, 1.0, -0.5)),
              rrt.make_light(rrt.LIGHT_AREA, intensity=(1.0, 1.0, 1.0), v0=corner, v1=a, v2=b)]
    rects = [rrt.make_rect(corner, a, b, material=1, light=1)]
    scene = rrt.SceneArrays(pos, nrm, np.zeros(len(pos), dtype=np.uint32), materials, lights, rects,
                            rect_aabb_mode=rrt.RECT_AABB_INCLUSIVE)
    # from a corner above, far enough for the bounding sphere to fill most of a 50 degree view
    direction = np.array([-0.6, 0.45, -0.75])
    direction /= np.linalg.norm(direction)
    eye = centre + direction * (0.5 * np.linalg.norm(size) / np.tan(np.radians(25.0)))
    camera = dict(position=tuple(eye), target=tuple(centre), fov=50.0, aperture=0.0, focal_distance=None)
    return scene, camera, 16, 10.0
