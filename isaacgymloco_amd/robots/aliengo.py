"""Unitree Aliengo model table (lsim_robot_model) for the simulator.

The numbers are the physical constants of the robot as published in its URDF
(reference copy: legged_gym/resources/robots/aliengo/urdf/aliengo.urdf:306-541 for the
trunk and FR leg; the other legs mirror the signs of y / x).  Isaac Gym loads that URDF
with collapse_fixed_joints=True (AGC:128), which merges the imu and rotor links into
their parents and keeps the feet (dont_collapse, URDF:517): 17 bodies, 12 DoF
(LR:1143-1148).  `build_model()` performs that merge and emits the table; the test
tests/test_model.py re-derives it from the reference URDF when that file is available.

Collision geometry is represented as sphere-swept points per body
(DESIGN.md "Collision shapes"): box corners / edge samples with radius 0, capsule end
spheres, foot sphere.  Priority order = order in the table (overflow beyond
LSIM_MAX_CONTACTS contacts is dropped from the end): feet, base, calves, thighs, hips.
"""
import numpy as np

from .. import abi

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
CALF_ROTOR_CYL = dict(radius=0.035, at=(0.0, -0.0997, 0.0))              # on the thigh link, axis along y
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


def merge(parts):
    """Rigidly merge [(mass, com(3), inertia-about-com 3x3)] -> same triple (parallel-axis theorem)."""
    m = sum(p[0] for p in parts)
    c = sum(p[0] * np.asarray(p[1], dtype=np.float64) for p in parts) / m
    I = np.zeros((3, 3))
    for pm, pc, pI in parts:
        d = np.asarray(pc, dtype=np.float64) - c
        I += pI + pm * (np.dot(d, d) * np.eye(3) - np.outer(d, d))
    return m, c, I


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


def collision_points():
    """[(body, (x,y,z), radius)] in priority order."""
    pts = []
    for l in range(4):                                   # feet
        pts.append((4 + 4 * l, (0.0, 0.0, 0.0), FOOT_RADIUS))
    hx, hy, hz = (0.5 * v for v in TRUNK_BOX)            # trunk box: 8 corners + mid points of the 4 long edges
    for sx in (1, -1):
        for sy in (1, -1):
            for sz in (-1, 1):
                pts.append((0, (sx * hx, sy * hy, sz * hz), 0.0))
    for sy in (1, -1):
        for sz in (-1, 1):
            pts.append((0, (0.0, sy * hy, sz * hz), 0.0))
    rc = 0.5 * min(CALF_BOX[1], CALF_BOX[2])
    for l in range(4):                                   # calf: 3 spheres along the shank (the foot sphere covers the tip)
        for z in (0.0, -CALF_BOX[0] / 3, -2 * CALF_BOX[0] / 3):
            pts.append((3 + 4 * l, (0.0, 0.0, z), rc))
    rt = 0.5 * min(THIGH_BOX[1], THIGH_BOX[2])
    for l in range(4):                                   # thigh: 4 spheres along the link
        for z in (0.0, -THIGH_BOX[0] / 3, -2 * THIGH_BOX[0] / 3, -THIGH_BOX[0]):
            pts.append((2 + 4 * l, (0.0, 0.0, z), rt))
    for l, leg in enumerate(LEGS):                       # hip capsule end spheres
        sy = _SY[leg]
        for dy in (-0.5 * HIP_CYL["length"], 0.5 * HIP_CYL["length"]):
            pts.append((1 + 4 * l, (0.0, HIP_CYL["at"][1] * sy + dy, 0.0), HIP_CYL["radius"]))
    for l, leg in enumerate(LEGS):                       # calf-rotor capsule on the thigh
        pts.append((2 + 4 * l, (0.0, CALF_ROTOR_CYL["at"][1] * _SY[leg], 0.0), CALF_ROTOR_CYL["radius"]))
    return pts


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
