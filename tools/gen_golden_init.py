"""Container-only: pin the init-time domain randomisation against the reference's own code.

The reference's `LeggedRobot._init_buffers` (LR:999-1028: motor_strength, Kp / Kd / motor-strength factors, payload, COM displacement),
`_process_rigid_shape_props` (LR:506-513: 64 friction buckets) and `_get_env_origins` (LR:1221-1240: initial terrain levels, types,
origins) are RUN under the Isaac Gym stub with their torch draws replaced by the Philox uniforms of the matching (env, 0xFFFFFFFF, tag,
index) -- the same injection tools/gen_golden.py uses for the step -- so that the values compare one for one with what lsim_create /
orc_create draw.  (tools/gen_golden.py copies these buffers from the oracle into the reference and therefore does not test them.)

Output: tests/golden/init_<task>.npz.  Only data is written."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as GG  # noqa: E402  (installs the reference import path; nothing runs at import)

TAG = GG.TAG
INIT_STEPW = 0xFFFFFFFF


def run(name, task, ref_cfg_cls, N=96, seed=1):
    cfg = GG.C.TASKS[task][0]()
    orc, lc, model, terrain = GG.make_oracle(cfg, N, seed=seed)       # supplies the terrain grid / model table the reference env is built on
    ref_cfg = ref_cfg_cls()
    GG.CTX = GG.RngCtx(seed, 0, N)
    GG.CTX.stepw = INIT_STEPW
    # _init_buffers: draw order of LR:999-1028
    dr = ref_cfg.domain_rand
    plan = []
    if getattr(dr, "randomize_motor_strength", False):
        plan.append((TAG["init"], 0, None))
    if dr.randomize_kp:
        plan.append((TAG["init"], 12, None))
    if dr.randomize_kd:
        plan.append((TAG["init"], 13, None))
    if dr.randomize_motor_strength:
        plan.append((TAG["init"], 14, None))
    if dr.randomize_payload_mass:
        plan.append((TAG["init"], 15, None))
    if dr.randomize_com_displacement:
        plan.append((TAG["init"], 16, None))
    GG.CTX.plan = plan
    env, tensors = GG.build_reference_env(ref_cfg, terrain, model, N)
    assert not GG.CTX.plan, GG.CTX.plan
    out = dict(num_envs=np.array(N), seed=np.array(seed), motor_strength=env.motor_strength.numpy().copy(), kp_factors=env.Kp_factors.numpy()[:, 0].copy(),
               kd_factors=env.Kd_factors.numpy()[:, 0].copy(), motor_strength_factors=env.motor_strength_factors.numpy()[:, 0].copy(),
               payload=env.payload.numpy()[:, 0].copy(), com_displacement=env.com_displacement.numpy().copy())
    # friction buckets (LR:506-513), drawn when env 0 is created
    GG.CTX.plan = [(TAG["init"], 19, None), (TAG["init_bucket"], 0, np.arange(64))]
    env._process_rigid_shape_props([], 0)
    assert not GG.CTX.plan
    out["friction"] = env.friction_coeffs.numpy().reshape(N).copy()
    # initial terrain levels / types / origins (LR:1221-1240)
    GG.CTX.plan = [(TAG["init"], 20, None)]
    env._get_env_origins()
    assert not GG.CTX.plan
    out["terrain_levels"] = env.terrain_levels.numpy().copy()
    out["terrain_types"] = env.terrain_types.numpy().copy()
    out["env_origins"] = env.env_origins.numpy().copy()
    np.savez_compressed(os.path.join(GG.GOLDEN, f"init_{name}.npz"), **out)
    print(f"wrote init_{name}.npz:", {k: (v.shape, float(np.min(v)), float(np.max(v))) for k, v in out.items() if v.ndim})
    orc.close()


if __name__ == "__main__":
    GG.install_rng_patches()
    run("aliengo", "aliengo", GG.aliengo_config.AlienGoRoughCfg)
    run("aliengo_stairs", "aliengo_stairs", GG.aliengo_stairs_config.AlienGoStairsCfg)
