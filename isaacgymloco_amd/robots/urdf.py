"""URDF -> lsim_robot_model for Unitree-style quadrupeds (trunk + 4 x {hip, thigh, calf, foot}; Aliengo, A1, Go1, ...).

Does what Isaac Gym's asset loader does for the reference (LR:1135-1148 with AGC:117-139 options):
  * `collapse_fixed_joints=True`: every link hanging on a fixed joint is merged rigidly into its parent (mass, centre of mass,
    inertia by the parallel-axis theorem; collision primitives re-expressed in the parent frame), EXCEPT children of joints
    marked `dont_collapse="true"` (the feet), which stay separate bodies without a degree of freedom;
  * bodies are numbered depth first with siblings in alphabetical order (base, FL_hip, FL_thigh, FL_calf, FL_foot, FR_hip, ...),
    degrees of freedom in the same order (LR:1143-1145);
  * `replace_cylinder_with_capsule=True` (AGC:131): cylinders collide as capsules.
The simulator's kinematic model wants zero joint rpy and a 17-body / 12-DoF tree; anything else is rejected loudly.

Collision geometry becomes sphere-swept points by the rules the hand-written Aliengo table follows: sphere -> one
point; the base's box -> 8 corners + the 4 long-edge mid points (radius 0); a limb's box -> spheres along its long axis (radius =
half the smaller cross-section; the distal end is left to the child's geometry); cylinder -> the two capsule end spheres, or one
sphere when it is a thin disc (length < 0.75 radius: the rotor housings).  Points are ordered feet, base, calves, thighs, hips and truncated to
LSIM_MAX_COLLISION_POINTS from the end.
"""
import xml.etree.ElementTree as ET

import numpy as np

from .. import abi
from .common import collision_points, merge, rpy_matrix as _rpy_matrix


def _vec(s, n=3, default=0.0):
    if s is None:
        return np.full(n, default, dtype=np.float64)
    return np.array([float(x) for x in s.split()], dtype=np.float64)


class _Link:
    def __init__(self, el):
        self.name = el.get("name")
        self.parts = []        # [(mass, com(3), inertia 3x3 about the com in link axes)]
        self.prims = []        # [(kind, dims, position(3), rotation 3x3)] in link axes
        ine = el.find("inertial")
        if ine is not None and ine.find("mass") is not None and float(ine.find("mass").get("value")) > 0.0:
            o = ine.find("origin")
            xyz = _vec(o.get("xyz") if o is not None else None)
            R = _rpy_matrix(_vec(o.get("rpy") if o is not None else None))
            it = ine.find("inertia")
            I = np.array([[float(it.get("ixx")), float(it.get("ixy")), float(it.get("ixz"))],
                          [float(it.get("ixy")), float(it.get("iyy")), float(it.get("iyz"))],
                          [float(it.get("ixz")), float(it.get("iyz")), float(it.get("izz"))]])
            self.parts.append((float(ine.find("mass").get("value")), xyz, R @ I @ R.T))
        for c in el.findall("collision"):
            g = c.find("geometry")
            if g is None or len(g) == 0:
                continue
            o = c.find("origin")
            xyz = _vec(o.get("xyz") if o is not None else None)
            R = _rpy_matrix(_vec(o.get("rpy") if o is not None else None))
            shape = list(g)[0]
            if shape.tag == "box":
                self.prims.append(("box", _vec(shape.get("size")), xyz, R))
            elif shape.tag == "cylinder":
                self.prims.append(("cylinder", np.array([float(shape.get("radius")), float(shape.get("length"))]), xyz, R))
            elif shape.tag == "sphere":
                self.prims.append(("sphere", np.array([float(shape.get("radius"))]), xyz, R))
            # meshes carry no analytic shape: ignored (the reference's quadruped URDFs collide with primitives only)

    def absorb(self, child, xyz, R):
        """merge `child` (rigidly attached at xyz / R in this link's frame) into this link"""
        for m, c, I in child.parts:
            self.parts.append((m, xyz + R @ c, R @ I @ R.T))
        for kind, dims, p, Rp in child.prims:
            self.prims.append((kind, dims, xyz + R @ p, R @ Rp))


def parse(path_or_text):
    """-> (bodies, dof_limits): bodies = list of dicts in Isaac Gym order after the fixed-joint collapse"""
    root = ET.parse(path_or_text).getroot() if not str(path_or_text).lstrip().startswith("<") else ET.fromstring(path_or_text)
    links = {l.get("name"): _Link(l) for l in root.findall("link")}
    joints = []
    for j in root.findall("joint"):
        o = j.find("origin")
        ax = j.find("axis")
        lim = j.find("limit")
        joints.append(dict(name=j.get("name"), type=j.get("type"), parent=j.find("parent").get("link"), child=j.find("child").get("link"),
                           xyz=_vec(o.get("xyz") if o is not None else None), rpy=_vec(o.get("rpy") if o is not None else None),
                           axis=_vec(ax.get("xyz")) if ax is not None else np.zeros(3),
                           limit={k: float(v) for k, v in lim.attrib.items()} if lim is not None else {},
                           keep=j.get("dont_collapse", "false").lower() == "true"))
    children = {}
    for j in joints:
        children.setdefault(j["parent"], []).append(j)
    child_names = {j["child"] for j in joints}
    roots = [n for n in links if n not in child_names]
    if len(roots) != 1:
        raise ValueError(f"URDF must have exactly one root link, found {roots}")

    def collapse(name):
        """merge the fixed-joint subtree below `name` into it; returns the surviving child joints [(joint, child link)]"""
        out = []
        for j in children.get(name, []):
            sub = collapse(j["child"])
            if j["type"] == "fixed" and not j["keep"]:
                R = _rpy_matrix(j["rpy"])
                links[name].absorb(links[j["child"]], j["xyz"], R)
                for jj, cn in sub:   # grandchildren re-attach to this link: compose the transforms
                    if np.any(np.abs(j["rpy"]) > 1e-12):
                        raise ValueError("rotated fixed joints above movable joints are not supported")
                    out.append((dict(jj, xyz=j["xyz"] + jj["xyz"]), cn))
            else:
                out.append((j, j["child"]))
        links[name]._kids = out
        return out

    collapse(roots[0])
    bodies, limits = [], []

    def visit(name, parent_idx, joint):
        link = links[name]
        if not link.parts:
            raise ValueError(f"link {name} has no mass after the fixed-joint collapse")
        m, c, I = merge(link.parts)
        idx = len(bodies)
        dof = -1
        if joint is not None:
            if np.any(np.abs(joint["rpy"]) > 1e-9):
                raise ValueError(f"joint {joint['name']}: non-zero rpy is not supported by the simulator's kinematic model")
            if joint["type"] in ("revolute", "continuous"):
                dof = len(limits)
                lm = joint["limit"]
                limits.append((lm.get("lower", -np.pi), lm.get("upper", np.pi), lm.get("velocity", 100.0), lm.get("effort", 1000.0)))
            elif joint["type"] != "fixed":
                raise ValueError(f"joint {joint['name']}: type {joint['type']} not supported")
        bodies.append(dict(name=name, mass=m, com=c, inertia=I, parent=parent_idx, dof=dof, prims=link.prims,
                           joint_pos=joint["xyz"] if joint is not None else np.zeros(3),
                           axis=joint["axis"] if (joint is not None and dof >= 0) else np.zeros(3),
                           joint_name=joint["name"] if joint is not None else None))
        for j, cn in sorted(link._kids, key=lambda t: t[1]):
            visit(cn, idx, j)

    visit(roots[0], -1, None)
    return bodies, limits


def model_from_bodies(bodies, limits, penalize_contacts_on=("thigh", "calf", "base"), terminate_after_contacts_on=("base",), foot_name="foot",
                      base_name_alias="base"):
    """lsim_robot_model from a parsed body list (parse() output or a stored table, robots/tables/*.json)"""
    if len(bodies) != 17 or len(limits) != 12:
        raise ValueError(f"expected 17 bodies / 12 DoF after the fixed-joint collapse, got {len(bodies)} / {len(limits)}")
    expect_parent = [-1] + [p for l in range(4) for p in (0, 1 + 4 * l, 2 + 4 * l, 3 + 4 * l)]
    expect_dof = [-1] + [d for l in range(4) for d in (3 * l, 3 * l + 1, 3 * l + 2, -1)]
    if [b["parent"] for b in bodies] != expect_parent or [b["dof"] for b in bodies] != expect_dof:
        raise ValueError("URDF topology is not trunk + 4 x (hip, thigh, calf, foot)")
    m = abi.LsimRobotModel()
    for i, b in enumerate(bodies):
        mb = m.bodies[i]
        mb.mass = float(b["mass"])
        I = np.asarray(b["inertia"], dtype=np.float64)
        for k in range(3):
            mb.com[k] = float(b["com"][k])
            mb.joint_pos[k] = float(b["joint_pos"][k])
            mb.joint_axis[k] = float(b["axis"][k])
        for k, (r, c) in enumerate(((0, 0), (0, 1), (0, 2), (1, 1), (1, 2), (2, 2))):
            mb.inertia[k] = float(I[r, c])
        mb.parent, mb.dof = b["parent"], b["dof"]
    for j, (lo, hi, vel, eff) in enumerate(limits):
        m.dof_pos_lower[j], m.dof_pos_upper[j], m.dof_vel_limit[j], m.dof_effort_limit[j] = lo, hi, vel, eff
    pts = collision_points(bodies, foot_name)
    m.num_collision_points = len(pts)
    for i, (body, pos, rad) in enumerate(pts):
        m.points[i].body, m.points[i].radius = body, rad
        for k in range(3):
            m.points[i].pos[k] = float(pos[k])
    names = [base_name_alias if i == 0 else b["name"] for i, b in enumerate(bodies)]
    feet = [i for i, n in enumerate(names) if foot_name in n]
    if len(feet) != 4:
        raise ValueError(f"expected 4 feet matching '{foot_name}', found {feet}")
    for k in range(4):
        m.feet_bodies[k] = feet[k]
    m.penalised_body_mask = sum(1 << i for i, n in enumerate(names) if any(p in n for p in penalize_contacts_on))
    m.termination_body_mask = sum(1 << i for i, n in enumerate(names) if any(p in n for p in terminate_after_contacts_on))
    return m, names, [b["joint_name"] for b in bodies if b["dof"] >= 0]


def build_model(path_or_text, **kw):
    """lsim_robot_model from a quadruped URDF; name patterns as in the task config (AGC:117-139 -> body masks, LR:1149-1219)."""
    bodies, limits = parse(path_or_text)
    return model_from_bodies(bodies, limits, **kw)


# ---- stored tables: the collapsed body list as plain data (robots/tables/<name>.json), so that a robot whose URDF is not shipped with
#      the package can still be simulated (tools/gen_robot_tables.py writes them from a URDF)
def table_to_json(bodies, limits):
    def arr(x):
        return np.asarray(x, dtype=np.float64).tolist()
    return {"bodies": [dict(name=b["name"], mass=float(b["mass"]), com=arr(b["com"]), inertia=arr(b["inertia"]), parent=b["parent"], dof=b["dof"],
                            joint_pos=arr(b["joint_pos"]), axis=arr(b["axis"]), joint_name=b["joint_name"],
                            prims=[dict(kind=k, dims=arr(d), pos=arr(p), rot=arr(R)) for k, d, p, R in b["prims"]]) for b in bodies],
            "limits": [list(map(float, l)) for l in limits]}


def table_from_json(d):
    bodies = []
    for b in d["bodies"]:
        b = dict(b)
        b["com"], b["inertia"] = np.array(b["com"]), np.array(b["inertia"])
        b["joint_pos"], b["axis"] = np.array(b["joint_pos"]), np.array(b["axis"])
        b["prims"] = [(p["kind"], np.array(p["dims"]), np.array(p["pos"]), np.array(p["rot"])) for p in b["prims"]]
        bodies.append(b)
    return bodies, [tuple(l) for l in d["limits"]]


def build_model_from_table(name, **kw):
    import json
    import os
    path = name if os.path.exists(name) else os.path.join(os.path.dirname(os.path.abspath(__file__)), "tables", name + ".json")
    with open(path) as f:
        bodies, limits = table_from_json(json.load(f))
    return model_from_bodies(bodies, limits, **kw)
