"""Physics anchor from outside the shipped collision model: the 64 sphere-swept points + cap of 8 contacts of the product
against densely sampled TRUE collision shapes (box surfaces, capsules; tests/dense_shapes.py: 574 points) with the cap lifted to 32, both in
the CPU oracle (same solver, same seeds / commands / actions).  PhysX's own narrow phase stays closed; this bounds what the sphere-swept
approximation of the URDF's boxes (AGC:124-138) does to the quantities the reward / termination stack reads: the collision
count of `_reward_collision` (LR:1573-1576), base contacts (termination, LR:257), foot load, base height, survival."""
import numpy as np
import pytest

import dense_shapes as D
from helpers import C, LC, T, make_oracle

FEET = [4, 8, 12, 16]


def _sim(task, dense, N, seed):
    from oracle import oracle
    cfg = C.TASKS[task][0]()
    cfg.domain_rand.push_robots = False
    cfg.domain_rand.disturbance = False
    if not dense:
        return make_oracle(cfg, N, seed=seed)[0]
    lib = oracle.variant("orc_shapes", D.DEFINES)
    structs = oracle.variant_structs(D.DEFINES)
    ter = T.Terrain(cfg.terrain, N, seed=1)
    model = D.build_dense_model(structs, penalize_contacts_on=tuple(cfg.asset.penalize_contacts_on),
                                terminate_after_contacts_on=tuple(cfg.asset.terminate_after_contacts_on), foot_name=cfg.asset.foot_name)
    from isaacgymloco_amd.envs.legged_robot import build_robot_model
    lc = LC.make_lsim_config(cfg, num_envs=N, terrain=ter, model=build_robot_model(cfg.asset), seed=seed)
    return oracle.OracleSim(lc, model, ter.heightsamples, ter.env_origins, library=lib)


def _rollout(task, dense, N=96, steps=150, sigma=0.7, seed=6):
    orc = _sim(task, dense, N, seed)
    orc.reset_all()
    rs = np.random.RandomState(2)
    pen = [2, 3, 6, 7, 10, 11, 14, 15, 0]                              # thighs, calves, base (AGC:126)
    acc = dict(collision=[], base_hit=[], fz=[], height=[], alive=[], contacts=[])
    for t in range(steps):
        orc.step((sigma * rs.normal(0, 1, (N, 12))).astype(np.float32))
        cf = orc.buf["contact_forces"]
        acc["collision"].append((np.linalg.norm(cf[:, pen, :], axis=-1) > 0.1).sum(1).mean())     # _reward_collision's count per env
        acc["base_hit"].append((np.linalg.norm(cf[:, 0, :], axis=-1) > 1.0).mean())               # termination contacts (LR:257)
        acc["fz"].append(cf[:, FEET, 2].sum(1).mean())
        acc["height"].append((orc.buf["root_states"][:, 2] - orc.buf["env_origins"][:, 2]).mean())
        acc["alive"].append(1.0 - orc.buf["reset"].mean())
        acc["contacts"].append(orc.buf["contact_count"][:, 0].mean())
    orc.close()
    return {k: float(np.mean(v[20:])) for k, v in acc.items()}


@pytest.mark.parametrize("task", ["aliengo", "aliengo_stairs"])
def test_true_shapes_vs_sphere_swept_points(task):
    shipped = _rollout(task, dense=False)
    true = _rollout(task, dense=True)
    print(task, "64 points, cap 8 :", {k: round(v, 4) for k, v in shipped.items()})
    print(task, "574 points, cap 32:", {k: round(v, 4) for k, v in true.items()})
    assert true["contacts"] > shipped["contacts"]                      # the dense sampling does see more points in contact
    # bounds ~2-3 x the measured differences
    assert abs(true["collision"] - shipped["collision"]) < 0.25 * max(shipped["collision"], 0.05) + 0.05
    assert abs(true["base_hit"] - shipped["base_hit"]) < 0.02
    assert abs(true["fz"] - shipped["fz"]) < 0.10 * 24.94 * 9.81
    assert abs(true["height"] - shipped["height"]) < 0.015
    assert abs(true["alive"] - shipped["alive"]) < 0.01
