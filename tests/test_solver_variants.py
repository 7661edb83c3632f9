"""The two solvers the library ships, side by side on the CPU oracle: PhysX's published TGS scheme (cfg.sim.physx.solver_type = 1,
4 position iterations, 0 velocity iterations -- the reference's settings LRC:245-248, and the default since round 4) against the build's
earlier solver (solver_type = 0: 8 velocity-level Gauss-Seidel sweeps over the whole 5 ms step) on the same seeds, commands and actions.
PhysX stays closed: this does not pin parity.  It bounds the MODELLING distance between the two solver families on the quantities the
reward / termination stack reads -- vertical load on the feet, feet in contact, base height, foot slip, joint state -- so that "not
comparable" becomes a stated number."""
import numpy as np
import pytest

from helpers import C, make_oracle

FEET = [4, 8, 12, 16]


def _rollout(task, solver_type, N=96, steps=150, sigma=0.5, seed=4):
    cfg = C.TASKS[task][0]()
    cfg.sim.physx.solver_type = solver_type
    cfg.domain_rand.push_robots = False
    cfg.domain_rand.disturbance = False
    orc, lc, model, ter = make_oracle(cfg, N, seed=seed)
    assert lc.solver_type == solver_type
    orc.reset_all()
    rs = np.random.RandomState(1)
    acc = dict(fz=[], contact=[], height=[], slip=[], rew=[], alive=[], qd=[])
    for t in range(steps):
        orc.step((sigma * rs.normal(0, 1, (N, 12))).astype(np.float32))
        cf = orc.buf["contact_forces"][:, FEET, :]
        inc = cf[:, :, 2] > 1.0                                            # the reference's contact flag (LR:207)
        vfoot = orc.buf["rigid_body_states"][:, FEET, 7:9]
        acc["fz"].append(cf[:, :, 2].sum(1).mean())                        # vertical load carried by the feet
        acc["contact"].append(inc.mean())
        acc["height"].append((orc.buf["root_states"][:, 2] - orc.buf["env_origins"][:, 2]).mean())
        acc["slip"].append((np.linalg.norm(vfoot, axis=-1) * inc).sum() / max(inc.sum(), 1))     # what _reward_feet_slide reads (LR:1610-1613)
        acc["rew"].append(orc.buf["rew"].mean())
        acc["alive"].append(1.0 - orc.buf["reset"].mean())
        acc["qd"].append(np.abs(orc.buf["dof_state"][:, :, 1]).mean())
    orc.close()
    return {k: float(np.mean(v[20:])) for k, v in acc.items()}             # after the landing transient of the reset


@pytest.mark.parametrize("task", ["aliengo", "aliengo_stairs"])
def test_tgs_variant_stays_close_on_what_the_rewards_read(task):
    pgs = _rollout(task, 0)
    tgs = _rollout(task, 1)
    print(task, "PGS8:", {k: round(v, 4) for k, v in pgs.items()})
    print(task, "TGS4:", {k: round(v, 4) for k, v in tgs.items()})
    weight = 24.94 * 9.81
    assert 0.5 * weight < pgs["fz"] < 1.6 * weight and 0.5 * weight < tgs["fz"] < 1.6 * weight     # both carry the robot (random actions: hops and falls)
    # bounds ~2-3 x the measured differences (printed above)
    assert abs(tgs["fz"] - pgs["fz"]) < 0.08 * weight
    assert abs(tgs["contact"] - pgs["contact"]) < 0.06
    assert abs(tgs["height"] - pgs["height"]) < 0.015
    assert abs(tgs["slip"] - pgs["slip"]) < 0.08
    assert abs(tgs["alive"] - pgs["alive"]) < 0.02
    assert abs(tgs["qd"] - pgs["qd"]) < 0.15 * pgs["qd"]


def test_tgs_variant_standing_load_and_height():
    """zero actions (PD hold at the default pose) on the flat task: both solvers must carry the robot's weight and settle at the
    same height to a fraction of a millimetre -- the static limit where the two schemes have to agree"""
    from helpers import quiet_cfg
    cfg = quiet_cfg("aliengo")          # no domain randomisation, default initial pose, flat ground
    out = {}
    for name, solver_type in (("pgs", 0), ("tgs", 1)):
        cfg.sim.physx.solver_type = solver_type
        orc, lc, model, ter = make_oracle(cfg, 32, seed=2)
        orc.reset_all()
        for t in range(100):
            orc.step(np.zeros((32, 12), np.float32))
        fz = orc.buf["contact_forces"][:, :, 2].sum(1)                         # all bodies: under Kp = 40 the robot sags until the calves touch down too
        ok = orc.buf["episode_length"] > 50
        out[name] = (float(fz[ok].mean()), float((orc.buf["root_states"][:, 2] - orc.buf["env_origins"][:, 2])[ok].mean()), int(ok.sum()))
        orc.close()
    print("standing: (total vertical contact force [N], base height [m], robots)", out)
    weight = 24.94 * 9.81
    assert out["pgs"][2] == 32 and out["tgs"][2] == 32
    assert abs(out["pgs"][0] - weight) < 0.01 * weight and abs(out["tgs"][0] - weight) < 0.01 * weight
    assert abs(out["pgs"][1] - out["tgs"][1]) < 1e-3
