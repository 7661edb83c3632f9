"""Synthetic simulator states for the BASELINE-size (N = 4096) golden fixtures -- test infrastructure.

The N = 16 fixtures store the injected simulator state of every step (a state the CPU physics produced).  At N = 4096 that is
2.5 MB per step, so the big fixtures store only what the reference computed FROM the state, and both the generator
(tools/gen_golden.py, which feeds the reference's LeggedRobot.step()) and the tests (which feed the oracle, the lane emulator and
the HIP library) rebuild the injected state from this module.  numpy's legacy RandomState streams are frozen by numpy's
compatibility policy, so the arrays are the same wherever they are generated; the fixture keeps a CRC of them all the same.

The states are plausible rather than physical (physics is frozen in these replays): robots near their env origin, moderate
tilts, a few tipped over / out of the terrain border / faster than commanded, every foot in contact with probability 1/2, a few
thigh / calf / base contacts, lateral foot forces that trip the stumble terms."""
import zlib

import numpy as np

FEET = [4, 8, 12, 16]
DEFAULT_DOF = np.array([0.1, 0.8, -1.5, -0.1, 0.8, -1.5, 0.1, 1.0, -1.5, -0.1, 1.0, -1.5], np.float32)   # AGC:38-52 (inputs only)


def _quat_from_rpy(r, p, y):
    cr, sr, cp, sp, cy, sy = np.cos(r / 2), np.sin(r / 2), np.cos(p / 2), np.sin(p / 2), np.cos(y / 2), np.sin(y / 2)
    q = np.stack([sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy, cr * cp * sy - sr * sp * cy, cr * cp * cy + sr * sp * sy], -1)
    return q / np.linalg.norm(q, axis=-1, keepdims=True)


def synth_step_inputs(N, t, env_origins, seed):
    """Inputs of replay step t for N envs: dict(actions, root, dof, body_feet, contact, ep_before, terrain_levels) as float32 / int64 arrays.
    env_origins [N, 3]: the env origins at the time of the step (root positions are drawn around them)."""
    rs = np.random.RandomState(100003 * int(seed) + 7919 * int(t) + 17)
    f32 = np.float32
    actions = rs.normal(0.0, 1.0, (N, 12)).astype(f32)
    if t == 0:
        actions[:] = 0.0                                   # runner start-up: zero-action step (BT:114)
    else:
        actions[::97] *= 300.0                             # exercises clip_actions (LR:129)
    root = np.zeros((N, 13), f32)
    root[:, 0:2] = env_origins[:, 0:2] + rs.uniform(-3.0, 3.0, (N, 2))
    root[:, 2] = env_origins[:, 2] + rs.uniform(0.25, 0.5, N)
    rpy = np.stack([rs.normal(0, 0.15, N), rs.normal(0, 0.15, N), rs.uniform(-np.pi, np.pi, N)], -1)
    tipped = rs.uniform(size=N) < 0.01
    rpy[tipped, 0] += np.pi * 0.9                          # fall-down termination (LR:272-275)
    root[:, 3:7] = _quat_from_rpy(rpy[:, 0], rpy[:, 1], rpy[:, 2]).astype(f32)
    root[:, 7:10] = rs.normal(0, 0.6, (N, 3))
    root[:, 10:13] = rs.normal(0, 0.8, (N, 3))
    fast = rs.uniform(size=N) < 0.01
    root[fast, 7] = np.where(rs.uniform(size=int(fast.sum())) < 0.5, 4.0, -4.0)   # base velocity far beyond the command (LR:266-270)
    out = rs.uniform(size=N) < 0.005
    root[out, 0] = -3.0                                    # out of the terrain border (LR:262-264)
    dof = np.zeros((N, 12, 2), f32)
    dof[:, :, 0] = DEFAULT_DOF[None, :] + rs.normal(0, 0.25, (N, 12))
    dof[:, :, 1] = rs.normal(0, 3.0, (N, 12))
    dof[::53, 3, 1] = 25.0                                 # beyond the soft velocity limit (dof_vel_limits term)
    body_feet = np.zeros((N, 4, 13), f32)
    off = np.array([[0.25, 0.13], [0.25, -0.13], [-0.25, 0.13], [-0.25, -0.13]], f32)
    body_feet[:, :, 0:2] = root[:, None, 0:2] + off[None] + rs.normal(0, 0.05, (N, 4, 2))
    body_feet[:, :, 2] = env_origins[:, None, 2] + rs.uniform(0.0, 0.15, (N, 4))
    body_feet[:, :, 6] = 1.0
    body_feet[:, :, 7:10] = rs.normal(0, 0.8, (N, 4, 3))
    contact = np.zeros((N, 17, 3), f32)
    stance = rs.uniform(size=(N, 4)) < 0.5
    fz = rs.uniform(0.5, 150.0, (N, 4)) * stance
    fz[rs.uniform(size=(N, 4)) < 0.02] = 400.0             # beyond max_contact_force
    contact[:, FEET, 2] = fz
    contact[:, FEET, 0:2] = rs.normal(0, 4.0, (N, 4, 2)) * stance[..., None]
    trip = (rs.uniform(size=(N, 4)) < 0.05) & stance
    if t > 0:                                              # every env trips on one of the steps, so that `rew` shows the stumble slices row by row
        forced = (np.arange(N) % 2) == (t % 2)
        fz[forced, 0] = np.maximum(fz[forced, 0], 5.0)
        contact[forced, FEET[0], 2] = fz[forced, 0]
        trip[forced, 0] = True
    contact[:, FEET, 0] = np.where(trip, 6.0 * fz + 1.0, contact[:, FEET, 0])     # lateral >> vertical: feet_stumble (LR:1590-1608)
    for b in (2, 3, 6, 7, 10, 11, 14, 15):                 # thighs and calves: penalised contacts (LR:1213-1215)
        hit = rs.uniform(size=N) < 0.03
        contact[hit, b, :] = rs.normal(0, 5.0, (int(hit.sum()), 3))
    base_hit = rs.uniform(size=N) < 0.01
    contact[base_hit, 0, 2] = 20.0                         # termination contact (LR:257)
    levels = ((7 * np.arange(N) + t) % 8).astype(np.int64)  # terrain levels 0..7 whatever the init draw was: the stumble terms need level > 3 (LR:1593)
    ep = rs.randint(0, 1003, size=N).astype(np.int64)
    ep[:4] = [499, 999, 1000, 1001]
    if t == 0:
        ep[:] = 0
    return dict(actions=actions, root=root, dof=dof, body_feet=body_feet, contact=contact, ep_before=ep, terrain_levels=levels)


def crc_of_inputs(inp):
    c = 0
    for k in sorted(inp):
        c = zlib.crc32(np.ascontiguousarray(inp[k]).tobytes(), c)
    return np.uint32(c)
