"""Container-only: write isaacgymloco_amd/robots/tables/go2.json (BASELINE config 5's second robot) and tests/golden/mocap_go2_frames.npz.

The reference ships no Go2 URDF, but it ships Go2 mocap clips (datasets/mocap_motions_go2/*.txt: per frame the 12 joint angles and the toe
positions in the base frame that its retargeting tool computed from a kinematic model of the robot; layout motion_loader.py:26-48).
  * KINEMATICS -- hip origins, thigh offset, thigh and calf lengths, joint axes x / y / y, DoF order -- are FITTED to those clips: least squares
    on the toe positions of all 1274 frames (residual: the 5-decimal text's rounding, 7e-6 m) and then set to the 4-decimal values the fit
    lands on.  That is the part tests/test_model.py pins.
  * INERTIAL values (masses, centres of mass, inertia tensors), joint limits (position / velocity / effort) and collision primitives are the
    values of Unitree's published go2_description, written down from the builder's knowledge of that file: there is no network in the build
    container and no Go2 asset in the reference to check them against, so they are NOMINAL -- stated in the table's "provenance" field.
Only derived numbers are written; no text of any reference file."""
import glob
import json
import os
import sys

import numpy as np
from scipy.optimize import least_squares

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from isaacgymloco_amd.robots import urdf  # noqa: E402

files = sorted(glob.glob("/root/reference/datasets/mocap_motions_go2/*.txt"))
assert len(files) == 13, files
frames = [np.array(json.load(open(f))["Frames"], np.float64) for f in files]
fr = np.concatenate(frames)
q, toe = fr[:, 7:19].reshape(-1, 4, 3), fr[:, 19:31].reshape(-1, 4, 3)
assert np.all(np.sign(toe[:, 2, 1]) > 0) and np.all(np.sign(toe[:, 3, 1]) < 0)          # leg slots FL, FR, RL, RR (Isaac Gym's order)


def rot(axis, a):
    c, s, z, o = np.cos(a), np.sin(a), np.zeros_like(a), np.ones_like(a)
    rows = ([o, z, z], [z, c, -s], [z, s, c]) if axis == "x" else ([c, z, s], [z, o, z], [-s, z, c])
    return np.stack([np.stack(r, -1) for r in rows], -2)


def fk(p, q):
    hx, hy, hz, ty, l1, l2 = p
    out = np.zeros((len(q), 4, 3))
    for leg in range(4):
        sx, sy = (1 if leg < 2 else -1), (1 if leg % 2 == 0 else -1)
        R1 = rot("x", q[:, leg, 0]); R2 = R1 @ rot("y", q[:, leg, 1]); R3 = R2 @ rot("y", q[:, leg, 2])
        out[:, leg] = np.array([sx * hx, sy * hy, hz]) + R1 @ np.array([0, sy * ty, 0]) + R2 @ np.array([0, 0, -l1]) + R3 @ np.array([0, 0, -l2])
    return out


fit = least_squares(lambda p: (fk(p, q) - toe).ravel(), np.array([0.2, 0.05, 0.0, 0.1, 0.2, 0.2]), xtol=1e-15, ftol=1e-15, gtol=1e-15)
print("fitted (hip x, hip y, hip z, thigh offset y, thigh length, calf length):", np.round(fit.x, 6), "max |residual|", np.abs(fit.fun).max())
HX, HY, HZ, TY, L1, L2 = [round(float(v), 4) + 0.0 for v in fit.x]
assert np.abs(fk((HX, HY, HZ, TY, L1, L2), q) - toe).max() < 1.5e-5
assert (HX, HY, abs(HZ), TY, L1, L2) == (0.1934, 0.0465, 0.0, 0.0955, 0.213, 0.213)

# ---- nominal inertial / limit / collision values (Unitree go2_description; see the module docstring)
I3 = lambda xx, xy, xz, yy, yz, zz: np.array([[xx, xy, xz], [xy, yy, yz], [xz, yz, zz]])   # noqa: E731
EYE = np.eye(3)
RY90 = np.array([[0.0, 0.0, 1.0], [0.0, 1.0, 0.0], [-1.0, 0.0, 0.0]])     # rpy (0, pi/2, 0): a box's long axis along the link's z
RX90 = np.array([[1.0, 0.0, 0.0], [0.0, 0.0, -1.0], [0.0, 1.0, 0.0]])     # rpy (pi/2, 0, 0): a cylinder's axis along the link's y
bodies = [dict(name="base", mass=6.921, com=np.array([0.021112, 0.0, -0.005366]), inertia=I3(0.02448, 0.00012166, 0.0014849, 0.098077, -3.12e-05, 0.107),
               parent=-1, dof=-1, joint_pos=np.zeros(3), axis=np.zeros(3), joint_name=None,
               prims=[("box", np.array([0.3762, 0.0935, 0.114]), np.zeros(3), EYE)])]
limits = []
for leg, name in enumerate(("FL", "FR", "RL", "RR")):
    sx, sy = (1.0 if leg < 2 else -1.0), (1.0 if leg % 2 == 0 else -1.0)
    base_idx = 1 + 4 * leg
    bodies.append(dict(name=f"{name}_hip", mass=0.678, com=np.array([-0.0054 * sx, 0.00194 * sy, -0.000105]),
                       inertia=I3(0.00048, -3.01e-06 * sx * sy, 1.11e-06 * sx, 0.000884, -1.42e-06 * sy, 0.000596),
                       parent=0, dof=3 * leg, joint_pos=np.array([sx * HX, sy * HY, HZ]), axis=np.array([1.0, 0.0, 0.0]), joint_name=f"{name}_hip_joint",
                       prims=[("cylinder", np.array([0.046, 0.04]), np.array([0.0, sy * 0.08, 0.0]), RX90)]))
    limits.append((-1.0472, 1.0472, 30.1, 23.7))
    bodies.append(dict(name=f"{name}_thigh", mass=1.152, com=np.array([-0.00374, -0.0223 * sy, -0.0327]),
                       inertia=I3(0.00584, 8.72e-05 * sy, -0.000289, 0.0058, 0.000808 * sy, 0.00103),
                       parent=base_idx, dof=3 * leg + 1, joint_pos=np.array([0.0, sy * TY, 0.0]), axis=np.array([0.0, 1.0, 0.0]),
                       joint_name=f"{name}_thigh_joint", prims=[("box", np.array([0.213, 0.0245, 0.034]), np.array([0.0, 0.0, -0.1065]), RY90)]))
    limits.append((-1.5708, 3.4907, 30.1, 23.7) if leg < 2 else (-0.5236, 4.5379, 30.1, 23.7))
    bodies.append(dict(name=f"{name}_calf", mass=0.154, com=np.array([0.00548, -0.000975 * sy, -0.115]),
                       inertia=I3(0.00108, 3.4e-07 * sy, 1.72e-05, 0.0011, 8.28e-06 * sy, 3.29e-05),
                       parent=base_idx + 1, dof=3 * leg + 2, joint_pos=np.array([0.0, 0.0, -L1]), axis=np.array([0.0, 1.0, 0.0]),
                       joint_name=f"{name}_calf_joint", prims=[("box", np.array([0.213, 0.016, 0.016]), np.array([0.0, 0.0, -0.1065]), RY90)]))
    limits.append((-2.7227, -0.83776, 15.7, 45.43))
    bodies.append(dict(name=f"{name}_foot", mass=0.04, com=np.zeros(3), inertia=9.6e-06 * np.eye(3), parent=base_idx + 2, dof=-1,
                       joint_pos=np.array([0.0, 0.0, -L2]), axis=np.zeros(3), joint_name=f"{name}_foot_fixed",
                       prims=[("sphere", np.array([0.022]), np.zeros(3), EYE)]))
table = urdf.table_to_json(bodies, limits)
table["provenance"] = {
    "kinematics": "joint origins, link lengths, axes and DoF order fitted to the toe positions of the reference's datasets/mocap_motions_go2/*.txt "
                  "(1274 frames x 4 feet, residual 7e-6 m); pinned by tests/test_model.py",
    "inertial_limits_collision": "NOMINAL: Unitree's published go2_description (masses, centres of mass, inertia tensors, joint limits, collision "
                                 "primitives) as the builder knows it; no Go2 asset exists in the reference and the build container has no network, "
                                 "so these numbers are not verified against any file",
    "total_mass_kg": float(sum(b["mass"] for b in bodies))}
out = os.path.join(ROOT, "isaacgymloco_amd", "robots", "tables", "go2.json")
with open(out, "w") as f:
    json.dump(table, f, indent=0)
print(out, "mass", table["provenance"]["total_mass_kg"])

# ---- fixture: every 4th frame of every clip (joint angles + toe positions), values unchanged
joint, toes, clip = [], [], []
for i, x in enumerate(frames):
    x = x[::4].astype(np.float32)
    joint.append(x[:, 7:19]); toes.append(x[:, 19:31]); clip += [i] * len(x)
fx = os.path.join(ROOT, "tests", "golden", "mocap_go2_frames.npz")
np.savez_compressed(fx, joint_pos=np.concatenate(joint), toe_pos_base=np.concatenate(toes), clip=np.array(clip, np.int16),
                    clip_names=np.array([os.path.basename(f) for f in files]))
print(fx, np.concatenate(joint).shape, os.path.getsize(fx), "bytes")
