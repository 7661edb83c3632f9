"""Shared pieces of the robot model builders: rigid merge of mass properties, rpy rotation, and the rules that turn URDF collision
primitives into the simulator's sphere-swept collision points.  Used by the hand-written Aliengo table
(robots/aliengo.py) and by the generic URDF loader (robots/urdf.py), so both produce the same points from the same primitives."""
import numpy as np

from .. import abi


def merge(parts):
    """Rigidly merge [(mass, com(3), inertia-about-com 3x3)] -> same triple (parallel-axis theorem)."""
    m = sum(p[0] for p in parts)
    c = sum(p[0] * np.asarray(p[1], dtype=np.float64) for p in parts) / m
    I = np.zeros((3, 3))
    for pm, pc, pI in parts:
        d = np.asarray(pc, dtype=np.float64) - c
        I += pI + pm * (np.dot(d, d) * np.eye(3) - np.outer(d, d))
    return m, c, I



def rpy_matrix(rpy):
    r, p, y = rpy
    cr, sr, cp, sp, cy, sy = np.cos(r), np.sin(r), np.cos(p), np.sin(p), np.cos(y), np.sin(y)
    return np.array([[cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
                     [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
                     [-sp, cp * sr, cp * cr]])



def _limb_box_points(dims, pos, R, skip_far_end):
    """spheres along the long axis of a limb's box; the end away from the joint is left to the child link when skip_far_end"""
    ext = np.abs(R @ np.diag(dims))                      # columns = box axes in link coordinates, scaled by the size
    lengths = dims
    a = int(np.argmax(lengths))
    others = [lengths[k] for k in range(3) if k != a]
    r = 0.5 * min(others)
    axis = R[:, a]
    L = lengths[a]
    ends = [pos - 0.5 * L * axis, pos + 0.5 * L * axis]
    near, far = (ends[0], ends[1]) if np.linalg.norm(ends[0]) <= np.linalg.norm(ends[1]) else (ends[1], ends[0])
    n = 3
    ts = [k / n for k in range(n)] if skip_far_end else [k / n for k in range(n + 1)]
    _ = ext
    return [(near + t * (far - near), r) for t in ts]


def collision_points(bodies, foot_name="foot", max_points=None):
    max_points = max_points or abi.DEFINES["LSIM_MAX_COLLISION_POINTS"]
    groups = {"feet": [], "base": [], "calf": [], "thigh": [], "hip": []}
    depth = {}
    for i, b in enumerate(bodies):
        depth[i] = 0 if b["parent"] < 0 else depth[b["parent"]] + 1
    for i, b in enumerate(bodies):
        role = "base" if depth[i] == 0 else ("feet" if foot_name in b["name"] else {1: "hip", 2: "thigh", 3: "calf"}.get(depth[i], "calf"))
        has_child_geometry = any(c["parent"] == i and c["prims"] for c in bodies)
        for kind, dims, pos, R in b["prims"]:
            if kind == "sphere":
                groups[role].append((i, pos, float(dims[0])))
            elif kind == "box" and role == "base":
                h = 0.5 * dims
                a = int(np.argmax(dims))
                for s in np.ndindex(2, 2, 2):
                    sg = np.array([1.0 if v == 0 else -1.0 for v in s])
                    groups[role].append((i, pos + R @ (sg * h), 0.0))
                for s in np.ndindex(2, 2):            # mid points of the four edges parallel to the long axis
                    sg = np.zeros(3)
                    o = [k for k in range(3) if k != a]
                    sg[o[0]] = 1.0 if s[0] == 0 else -1.0
                    sg[o[1]] = 1.0 if s[1] == 0 else -1.0
                    groups[role].append((i, pos + R @ (sg * h), 0.0))
            elif kind == "box":
                for p, r in _limb_box_points(dims, pos, R, skip_far_end=has_child_geometry and role == "calf"):
                    groups[role].append((i, p, r))
            elif kind == "cylinder":
                radius, length = float(dims[0]), float(dims[1])
                axis = R[:, 2]
                if length < 0.75 * radius:               # a thin disc (rotor housings): one sphere
                    groups[role].append((i, pos, radius))
                else:
                    groups[role].append((i, pos - 0.5 * length * axis, radius))
                    groups[role].append((i, pos + 0.5 * length * axis, radius))
    pts = groups["feet"] + groups["base"] + groups["calf"] + groups["thigh"] + groups["hip"]
    return pts[:max_points]


