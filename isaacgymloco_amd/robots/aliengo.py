"""Unitree Aliengo model table (lsim_robot_model) for the simulator.

The numbers are the physical constants of the robot as published in its URDF
(reference copy: legged_gym/resources/robots/aliengo/urdf/aliengo.urdf:306-541 for the
trunk and FR leg; the other legs mirror the signs of y / x).  Isaac Gym loads that URDF
with collapse_fixed_joints=True (AGC:128), which merges the imu and rotor links into
their parents and keeps the feet (dont_collapse, URDF:517): 17 bodies, 12 DoF
(LR:1143-1148).  `build_model()` performs that merge and emits the table; the test
tests/test_model.py re-derives it from the reference URDF when that file is available.

Collision geometry is represented as sphere-swept points per body
(DESIGN.md 4.6), derived from the URDF's collision primitives by the shared rules of
robots/common.py: box corners / edge samples with radius 0, spheres along the limb boxes,
capsule end spheres, one sphere per rotor disc, foot sphere -- 64 points.  Priority order =
order in the table (contacts beyond LSIM_MAX_CONTACTS are dropped from the end): feet,
base, calves, thighs, hips.  robots/urdf.py builds the same table from the URDF file itself.
"""
import numpy as np

from .. import abi
from .common import collision_points as _points_from_prims, merge, rpy_matrix

LEGS = ("FL", "FR", "RL", "RR")  # Isaac Gym DoF/body order (LR:1143-1145)
_SX = {"FL": 1.0, "FR": 1.0, "RL": -1.0, "RR": -1.0}   # front / rear
_SY = {"FL": 1.0, "FR": -1.0, "RL": 1.0, "RR": -1.0}   # left / right

# --- URDF constants (FL-leg sign convention; mirrored per leg below) ---
TRUNK = dict(mass=11.644, com=(0.008811, 0.003839, 0.000273),
             inertia=(0.051944892, 0.001703617, 0.000235941, 0.24693924, 0.000119783, 0.270948307))
IMU = dict(mass=0.001, com=(0.0, 0.0, 0.0), inertia=(0.0001, 0.0, 0.0, 0.000001, 0.0, 0.0001), at=(0.0, 0.0, 0.0))
HIP_ROTOR = dict(mass=0.146, inertia=(0.000138702, 0.0, 0.0, 8.3352e-05, 0.0, 8.3352e-05), at=(0.139985, 0.051, 0.0))
HIP = dict(mass=1.993, com=(-0.022191, 0.015144, -1.5e-05),
           inertia=(0.002446735, -0.00059805, 1.945e-06, 0.003925876, 1.284e-06, 0.004148145))
THIGH_ROTOR = dict(mass=0.146, inertia=(8.3352e-05, 0.0, 0.0, 0.000138702, 0.0, 8.3352e-05), at=(0.0, 0.0298, 0.0))
THIGH = dict(mass=0.639, com=(-0.005607, -0.003877, -0.048199),
             inertia=(0.004173855, 1.0284e-05, -0.000318874, 0.004343802, 0.000109233, 0.000340136))
CALF_ROTOR = dict(mass=0.132, inertia=(0.000145463, 0.0, 0.0, 0.000133031, 0.0, 0.000145463), at=(0.0, -0.0997, 0.0))
CALF = dict(mass=0.207, com=(0.002781, 6.3e-05, -0.142518),
            inertia=(0.002129279, 3.9e-08, 5.757e-06, 0.002141463, -5.16e-07, 3.7583e-05))
FOOT = dict(mass=0.06, com=(0.0, 0.0, 0.0), inertia=(1.6854e-05, 0.0, 0.0, 1.6854e-05, 0.0, 1.6854e-05))

HIP_ORIGIN = (0.2407, 0.051, 0.0)      # trunk -> hip joint (x mirrored front/rear, y left/right)
THIGH_ORIGIN = (0.0, 0.0868, 0.0)      # hip -> thigh joint
CALF_ORIGIN = (0.0, 0.0, -0.25)
FOOT_ORIGIN = (0.0, 0.0, -0.25)
LIMITS = dict(hip=(-0.873, 1.047, 20.0, 44.0), thigh=(-0.524, 3.927, 20.0, 44.0), calf=(-2.775, -0.611, 15.89, 55.0))

# collision primitives (URDF <collision>): trunk box, hip cylinder (-> capsule, AGC:131), thigh/calf boxes, foot sphere
TRUNK_BOX = (0.647, 0.15, 0.112)
HIP_CYL = dict(radius=0.046, length=0.0418, at=(0.0, 0.083, 0.0))       # axis along y
THIGH_BOX = (0.25, 0.0374, 0.043)   # long axis along -z of the link, centred at z = -0.125
CALF_BOX = (0.25, 0.0208, 0.016)
FOOT_RADIUS = 0.0265


def _mat(i6):
    xx, xy, xz, yy, yz, zz = i6
    return np.array([[xx, xy, xz], [xy, yy, yz], [xz, yz, zz]], dtype=np.float64)


def _mirror(part, sx, sy):
    """Mirror com / inertia products of a FL-convention part to another leg."""
    out = dict(part)
    if "com" in part:
        c = part["com"]
        out["com"] = (c[0] * sx, c[1] * sy, c[2])
    xx, xy, xz, yy, yz, zz = part["inertia"]
    out["inertia"] = (xx, xy * sx * sy, xz * sx, yy, yz * sy, zz)
    if "at" in part:
        a = part["at"]
        out["at"] = (a[0] * sx, a[1] * sy, a[2])
    return out


def _hip_signs(leg):
    # URDF hip inertial: com x flips front/rear, y flips left/right (URDF FR: (-0.022191,-0.015144), RR: (+0.022191,-0.015144))
    return _SX[leg], _SY[leg]


def body_table():
    """List of dicts (17 bodies, Isaac order) with merged mass properties."""
    bodies = []
    parts = [(TRUNK["mass"], TRUNK["com"], _mat(TRUNK["inertia"])), (IMU["mass"], IMU["at"], _mat(IMU["inertia"]))]
    for leg in LEGS:
        r = _mirror(HIP_ROTOR, _SX[leg], _SY[leg])
        parts.append((r["mass"], r["at"], _mat(HIP_ROTOR["inertia"])))
    m, c, I = merge(parts)
    bodies.append(dict(name="base", mass=m, com=c, inertia=I, joint_pos=(0, 0, 0), axis=(0, 0, 0), parent=-1, dof=-1))
    for l, leg in enumerate(LEGS):
        sx, sy = _SX[leg], _SY[leg]
        hip = _mirror(HIP, sx, sy)
        # the hip link's own inertial is mirrored in x for rear legs: com x -> -x ... (URDF RR_hip com = (+0.022191, -0.015144))
        hip["com"] = (HIP["com"][0] * sx, HIP["com"][1] * sy, HIP["com"][2])
        tr = _mirror(THIGH_ROTOR, 1.0, sy)
        m, c, I = merge([(hip["mass"], hip["com"], _mat(hip["inertia"])), (tr["mass"], tr["at"], _mat(THIGH_ROTOR["inertia"]))])
        bodies.append(dict(name=f"{leg}_hip", mass=m, com=c, inertia=I,
                           joint_pos=(HIP_ORIGIN[0] * sx, HIP_ORIGIN[1] * sy, 0.0), axis=(1, 0, 0), parent=0, dof=3 * l))
        th = _mirror(THIGH, 1.0, sy)
        cr = _mirror(CALF_ROTOR, 1.0, sy)
        m, c, I = merge([(th["mass"], th["com"], _mat(th["inertia"])), (cr["mass"], cr["at"], _mat(CALF_ROTOR["inertia"]))])
        bodies.append(dict(name=f"{leg}_thigh", mass=m, com=c, inertia=I,
                           joint_pos=(0.0, THIGH_ORIGIN[1] * sy, 0.0), axis=(0, 1, 0), parent=1 + 4 * l, dof=3 * l + 1))
        bodies.append(dict(name=f"{leg}_calf", mass=CALF["mass"], com=np.array(CALF["com"]), inertia=_mat(CALF["inertia"]),
                           joint_pos=CALF_ORIGIN, axis=(0, 1, 0), parent=2 + 4 * l, dof=3 * l + 2))
        bodies.append(dict(name=f"{leg}_foot", mass=FOOT["mass"], com=np.array(FOOT["com"]), inertia=_mat(FOOT["inertia"]),
                           joint_pos=FOOT_ORIGIN, axis=(0, 0, 0), parent=3 + 4 * l, dof=-1))
    return bodies


ROTOR_DISC = dict(radius=0.035, length=0.02)   # hip / thigh / calf rotor housings (URDF <collision> cylinders of the *_rotor links)
_RX90, _RY90 = (np.pi / 2, 0.0, 0.0), (0.0, np.pi / 2, 0.0)   # cylinder axis along y / box long axis along z


def _prims():
    """Collision primitives per body after the fixed-joint collapse, in the order Isaac Gym meets them in the URDF (the link's own
    shape first, then the merged children in URDF joint order: FR, FL, RR, RL): [(kind, dims, position, rotation)] per body."""
    cyl = lambda d, at, rpy: ("cylinder", np.array([d["radius"], d["length"]]), np.array(at, dtype=np.float64), rpy_matrix(rpy))
    box = lambda size, at: ("box", np.array(size, dtype=np.float64), np.array(at, dtype=np.float64), rpy_matrix(_RY90))
    out = [[] for _ in range(17)]
    out[0].append(("box", np.array(TRUNK_BOX), np.zeros(3), np.eye(3)))
    for leg in ("FR", "FL", "RR", "RL"):
        r = _mirror(HIP_ROTOR, _SX[leg], _SY[leg])
        out[0].append(cyl(ROTOR_DISC, r["at"], _RY90))
    for l, leg in enumerate(LEGS):
        sy = _SY[leg]
        out[1 + 4 * l].append(cyl(HIP_CYL, (0.0, HIP_CYL["at"][1] * sy, 0.0), _RX90))
        out[1 + 4 * l].append(cyl(ROTOR_DISC, (0.0, THIGH_ROTOR["at"][1] * sy, 0.0), _RX90))
        out[2 + 4 * l].append(box(THIGH_BOX, (0.0, 0.0, -0.5 * THIGH_BOX[0])))
        out[2 + 4 * l].append(cyl(ROTOR_DISC, (0.0, CALF_ROTOR["at"][1] * sy, 0.0), _RX90))
        out[3 + 4 * l].append(box(CALF_BOX, (0.0, 0.0, -0.5 * CALF_BOX[0])))
        out[4 + 4 * l].append(("sphere", np.array([FOOT_RADIUS]), np.zeros(3), np.eye(3)))
    return out


def collision_points():
    """[(body, (x,y,z), radius)] in priority order, from the URDF's collision primitives by the rules of robots/common.py"""
    bodies = body_table()
    for b, prims in zip(bodies, _prims()):
        b["prims"] = prims
    return _points_from_prims(bodies, "foot")


def build_model(penalize_contacts_on=("thigh", "calf", "base"), terminate_after_contacts_on=("base",), foot_name="foot"):
    """lsim_robot_model for Aliengo (AGC:117-139 name patterns -> body masks, LR:1149-1219)."""
    m = abi.LsimRobotModel()
    bodies = body_table()
    for i, b in enumerate(bodies):
        mb = m.bodies[i]
        mb.mass = b["mass"]
        I = b["inertia"]
        for k in range(3):
            mb.com[k] = float(b["com"][k])
            mb.joint_pos[k] = float(b["joint_pos"][k])
            mb.joint_axis[k] = float(b["axis"][k])
        for k, (r, c) in enumerate(((0, 0), (0, 1), (0, 2), (1, 1), (1, 2), (2, 2))):
            mb.inertia[k] = float(I[r, c])
        mb.parent = b["parent"]
        mb.dof = b["dof"]
    for l in range(4):
        for k, key in enumerate(("hip", "thigh", "calf")):
            lo, hi, vel, eff = LIMITS[key]
            j = 3 * l + k
            m.dof_pos_lower[j], m.dof_pos_upper[j], m.dof_vel_limit[j], m.dof_effort_limit[j] = lo, hi, vel, eff
    pts = collision_points()
    assert len(pts) <= abi.DEFINES["LSIM_MAX_COLLISION_POINTS"]
    m.num_collision_points = len(pts)
    for i, (body, pos, rad) in enumerate(pts):
        m.points[i].body = body
        m.points[i].radius = rad
        for k in range(3):
            m.points[i].pos[k] = pos[k]
    names = [b["name"] for b in bodies]
    feet = [i for i, n in enumerate(names) if foot_name in n]
    for k in range(4):
        m.feet_bodies[k] = feet[k]
    m.penalised_body_mask = sum(1 << i for i, n in enumerate(names) if any(p in n for p in penalize_contacts_on))
    m.termination_body_mask = sum(1 << i for i, n in enumerate(names) if any(p in n for p in terminate_after_contacts_on))
    return m


BODY_NAMES = [b["name"] for b in body_table()]
DOF_NAMES = [f"{leg}_{j}_joint" for leg in LEGS for j in ("hip", "thigh", "calf")]
