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



def _limb_box_points(dims, pos, R, skip_far_end, n=3, span=None, wide=False):
    """spheres along the long axis of a limb's box; the end away from the joint is left to the child link when skip_far_end.
    span = (t0, t1): n spheres evenly from t0 to t1 of the length instead (the calf rule below)"""
    lengths = dims
    a = int(np.argmax(lengths))
    others = [lengths[k] for k in range(3) if k != a]
    r = 0.5 * (max(others) if wide else min(others))      # wide: the larger half-width of the cross-section (the calf rule)
    axis = R[:, a]
    L = lengths[a]
    ends = [pos - 0.5 * L * axis, pos + 0.5 * L * axis]
    near, far = (ends[0], ends[1]) if np.linalg.norm(ends[0]) <= np.linalg.norm(ends[1]) else (ends[1], ends[0])
    if span is not None:
        ts = [span[0] + (span[1] - span[0]) * k / (n - 1) for k in range(n)]
    else:
        ts = [k / n for k in range(n)] if skip_far_end else [k / n for k in range(n + 1)]
    return [(near + t * (far - near), r) for t in ts]


def collision_points(bodies, foot_name="foot", max_points=None):
    """The sphere-swept collision points of a robot from its URDF collision primitives, at most LSIM_MAX_COLLISION_POINTS = 64
    (one lane of the narrow phase each).  Rules:
      foot sphere                as it is
      trunk box                  8 corners + the mid points of the four long edges, radius 0
      calf box                   SIX spheres from 12 % to 86 % of the link (round 6; before: 3 at 0, 1/3, 2/3).  Under trained stairs policies the
                                 calves are what brushes risers and tread edges: with 8 mm spheres every 83 mm and nothing between 2/3 of the link
                                 and the foot sphere, `_reward_collision` (LR:1573-1576) counted 0.048 calf contacts per env-step where densely
                                 sampled true shapes count 0.069 (-31 %; base, hip and thigh rates agreed: tools/trained_policy_physics.py,
                                 profiles/r06_trained_policy_physics_*.json).  At 37 mm spacing a tread edge between two spheres is within the
                                 18 mm reach (radius + contact offset) of one of them.  The knee end is the thigh's far-end sphere.
      thigh box                  4 spheres at 0, 1/3, 2/3, 1 of the link
      thick cylinders            capsule end spheres (AGC:131 replaces cylinders by capsules); ONE centre sphere when the cylinder is shorter than its
                                 radius (the hip: 41.8 mm long, r 46 mm -- its two end spheres were 42 mm apart and mostly each other)
      thin discs (rotor housings, length < 0.75 radius)   one sphere -- dropped when the disc's centre lies inside a box of the same link (the four
                                 on the trunk) or the link also has a thick cylinder (the hip's own, which stands 13 mm off the trunk's side): their
                                 volume is the link's other shapes' to within a centimetre, and the calves needed their twelve table entries.
    If a robot still does not fit, its calf spheres are thinned (6 -> 3) before anything is cut."""
    max_points = max_points or abi.DEFINES["LSIM_MAX_COLLISION_POINTS"]
    depth = {}
    for i, b in enumerate(bodies):
        depth[i] = 0 if b["parent"] < 0 else depth[b["parent"]] + 1

    def build(n_calf):
        groups = {"feet": [], "base": [], "calf": [], "thigh": [], "hip": []}
        for i, b in enumerate(bodies):
            role = "base" if depth[i] == 0 else ("feet" if foot_name in b["name"] else {1: "hip", 2: "thigh", 3: "calf"}.get(depth[i], "calf"))
            has_child_geometry = any(c["parent"] == i and c["prims"] for c in bodies)
            boxes = [(dims, pos, R) for kind, dims, pos, R in b["prims"] if kind == "box"]
            thick = any(kind == "cylinder" and float(dims[1]) >= 0.75 * float(dims[0]) for kind, dims, pos, R in b["prims"])
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
                elif kind == "box" and role == "calf" and has_child_geometry and n_calf > 3:
                    for p, r in _limb_box_points(dims, pos, R, True, n=n_calf, span=(0.12, 0.86)):
                        groups[role].append((i, p, r))
                elif kind == "box":
                    for p, r in _limb_box_points(dims, pos, R, skip_far_end=has_child_geometry and role == "calf"):
                        groups[role].append((i, p, r))
                elif kind == "cylinder":
                    radius, length = float(dims[0]), float(dims[1])
                    axis = R[:, 2]
                    if length < 0.75 * radius:               # a thin disc (rotor housings)
                        inside = any(np.all(np.abs(Rb.T @ (np.asarray(pos) - np.asarray(pb))) <= 0.5 * np.asarray(db) + 1e-9) for db, pb, Rb in boxes)
                        if not (inside or thick):
                            groups[role].append((i, pos, radius))
                    elif length < radius:
                        groups[role].append((i, pos, radius))
                    else:
                        groups[role].append((i, pos - 0.5 * length * axis, radius))
                        groups[role].append((i, pos + 0.5 * length * axis, radius))
        return groups["feet"] + groups["base"] + groups["calf"] + groups["thigh"] + groups["hip"]

    for n_calf in (6, 5, 4, 3):
        pts = build(n_calf)
        if len(pts) <= max_points:
            return pts
    return pts[:max_points]
