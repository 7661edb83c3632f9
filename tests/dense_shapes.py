"""Densely sampled TRUE collision shapes of the Aliengo asset (test infrastructure for tests/test_shape_variants.py).

The shipped model represents the URDF's collision primitives by 64 sphere-swept points (robots/common.py: limb boxes as 3-4 spheres along
the link, trunk box as corners + long-edge mid points, cylinders as capsule end spheres) with at most 8 simultaneous contacts.  Here the
same primitives (robots/aliengo.py: _prims(), the URDF after Isaac Gym's fixed-joint collapse, AGC:124-138) are sampled as what they
are: box SURFACES -- every vertex, the 12 edges every <= 4 cm, the 6 faces on a <= 8 cm grid, radius 0 -- capsules (Isaac Gym replaces the
asset's cylinders by capsules, AGC:131) as spheres every <= 2 cm along the axis, the foot spheres as they are: ~600 points, into an oracle
built with a 768-entry point table and the contact cap lifted to 32 (oracle/Makefile: liborc_shapes.so)."""
import numpy as np

from isaacgymloco_amd.robots import aliengo

DEFINES = {"LSIM_MAX_COLLISION_POINTS": 768, "LSIM_MAX_CONTACTS": 32}


def _box_surface(dims, pos, R, edge_step=0.04, face_step=0.08):
    h = 0.5 * np.asarray(dims, dtype=np.float64)
    pts = set()

    def grid(n_lo, length):
        n = max(int(np.ceil(length / n_lo)), 1)
        return np.linspace(-0.5 * length, 0.5 * length, n + 1)
    axes = [grid(edge_step, dims[k]) for k in range(3)]
    for a in range(3):                                     # edges parallel to axis a
        o = [k for k in range(3) if k != a]
        for s0 in (-1, 1):
            for s1 in (-1, 1):
                for t in axes[a]:
                    p = np.zeros(3); p[a] = t; p[o[0]] = s0 * h[o[0]]; p[o[1]] = s1 * h[o[1]]
                    pts.add(tuple(np.round(p, 9)))
    faces = [grid(face_step, dims[k]) for k in range(3)]
    for a in range(3):                                     # faces normal to axis a
        o = [k for k in range(3) if k != a]
        for s in (-1, 1):
            for u in faces[o[0]]:
                for v in faces[o[1]]:
                    p = np.zeros(3); p[a] = s * h[a]; p[o[0]] = u; p[o[1]] = v
                    pts.add(tuple(np.round(p, 9)))
    return [(pos + R @ np.array(p), 0.0) for p in sorted(pts)]


def _capsule(radius, length, pos, R, step=0.02):
    n = max(int(np.ceil(length / step)), 1)
    axis = R[:, 2]
    return [(pos + t * axis, float(radius)) for t in np.linspace(-0.5 * length, 0.5 * length, n + 1)]


def dense_points():
    out = []
    for body, prims in enumerate(aliengo._prims()):
        for kind, dims, pos, R in prims:
            if kind == "sphere":
                out.append((body, pos, float(dims[0])))
            elif kind == "box":
                out += [(body, p, r) for p, r in _box_surface(dims, pos, R)]
            elif kind == "cylinder":
                out += [(body, p, r) for p, r in _capsule(float(dims[0]), float(dims[1]), pos, R)]
    feet = [p for p in out if p[0] in (4, 8, 12, 16)]
    return feet + [p for p in out if p[0] not in (4, 8, 12, 16)]


def build_dense_model(structs, **patterns):
    """the product's model table with the collision points replaced by the dense sampling, in the variant library's (larger) struct"""
    src = aliengo.build_model(**patterns)
    Model = structs["lsim_robot_model"]
    m = Model()
    import ctypes
    Src = type(src)
    for name, _ in Model._fields_:          # field by field as bytes: the two struct classes are distinct ctypes types with equal member layouts
        if name in ("points", "num_collision_points"):
            continue
        fd, fs = getattr(Model, name), getattr(Src, name)
        assert fd.size == fs.size, name
        ctypes.memmove(ctypes.addressof(m) + fd.offset, ctypes.addressof(src) + fs.offset, fd.size)
    pts = dense_points()
    assert len(pts) <= DEFINES["LSIM_MAX_COLLISION_POINTS"], len(pts)
    m.num_collision_points = len(pts)
    for i, (body, pos, rad) in enumerate(pts):
        m.points[i].body = body
        m.points[i].radius = rad
        for k in range(3):
            m.points[i].pos[k] = float(pos[k])
    return m
