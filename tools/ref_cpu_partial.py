"""Container-only measurement (legs B and C): the REFERENCE's own Python on the CPU cores of this container, with warm-up
and repeats (BASELINE.md section 2 held single un-repeated timings).  Neither leg contains physics -- Isaac Gym / PhysX is absent -- so both are
UPPER bounds on what the reference's CPU path (`--sim_device cpu`) could do, not its throughput.
  B  env side   : LeggedRobot.step() of the reference (legged_gym/envs/base/legged_robot.py:122-176) with the simulator calls stubbed to no-ops
                  (tools/refstub) and the simulator state frozen at a realistic one: delay model, 4 x _compute_torques, post_physics_step
                  (callback, 187 + 63 height samples, termination, rewards, termination observations, reset_idx, observations)
  C  learner    : HIMPPO.act x T + process_env_step x T + compute_returns, and HIMPPO.update() (5 epochs x 4 minibatches), rsl_rl as shipped
Writes profiles/ref_cpu_partial.json.   usage: python tools/ref_cpu_partial.py [--quick]"""
import json
import os
import statistics
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
import gen_golden as G          # noqa: E402  (refenv.install(), the stub gym, build_reference_env; the Philox injection is NOT installed here)
from helpers import make_oracle  # noqa: E402
from isaacgymloco_amd.envs import config as C  # noqa: E402
from legged_gym.envs.aliengo import aliengo_stairs_config  # noqa: E402

quick = "--quick" in sys.argv
REPS = 3 if quick else 10


def cpu_model():
    for line in open("/proc/cpuinfo"):
        if line.startswith("model name"):
            return line.split(":", 1)[1].strip()
    return "unknown"


def med(xs):
    return {"median_s": statistics.median(xs), "min_s": min(xs), "max_s": max(xs), "repeats": len(xs)}


def leg_b(N):
    task, ref_cls = "aliengo_stairs", aliengo_stairs_config.AlienGoStairsCfg
    cfg = C.TASKS[task][0]()
    gen, lc, model, terrain = make_oracle(cfg, N, seed=1)
    env, tensors = G.build_reference_env(ref_cls(), terrain, model, N)
    G.sync_initial_state(env, tensors, gen)
    env.reset_idx(torch.arange(N))
    gen.reset_all()
    rs = np.random.RandomState(0)
    for _ in range(3):                              # a realistic frozen state: robots that have landed and moved
        gen.step(rs.normal(0, 1, (N, 12)).astype(np.float32))
    body = np.zeros((N, 17, 13), np.float32)
    body[:, :, :] = gen.buf["rigid_body_states"]
    G.inject_state(tensors, gen.buf["root_states"].copy(), gen.buf["dof_state"].copy(), body, gen.buf["contact_forces"].copy())
    env.episode_length_buf = torch.randint(0, 1000, (N,))
    acts = [torch.from_numpy(rs.normal(0, 1, (N, 12)).astype(np.float32)) for _ in range(4)]
    for i in range(3):
        env.step(acts[i % 4])
    ts = []
    for i in range(REPS):
        t = time.perf_counter()
        env.step(acts[i % 4])
        ts.append(time.perf_counter() - t)
    gen.close()
    out = med(ts)
    out.update(envs=N, env_steps_per_s=N / out["median_s"], what="reference LeggedRobot.step(), physics stubbed out (upper bound)")
    return out


def leg_c(N, T=100):
    from rsl_rl.algorithms import HIMPPO
    from rsl_rl.modules import HIMActorCritic
    tcfg = C.TASKS["aliengo"][1]().to_dict()
    torch.manual_seed(1)
    ac = HIMActorCritic(270, 238, 45, 12, **tcfg["policy"])
    alg = HIMPPO(ac, device="cpu", **tcfg["algorithm"])
    alg.init_storage(N, T, [270], [238], [12])
    reps = max(2, REPS // 5) if N >= 4096 else REPS

    def rollout():
        obs, crit = torch.randn(N, 270), torch.randn(N, 238)
        with torch.inference_mode():
            for _ in range(T):
                alg.act(obs, crit)
                obs, crit = torch.randn(N, 270), torch.randn(N, 238)
                alg.process_env_step(torch.randn(N), torch.rand(N) < 0.01, {"time_outs": torch.zeros(N, dtype=torch.bool)}, crit)
            alg.compute_returns(crit)
    coll, upd = [], []
    for i in range(reps + 1):                     # the first round is the warm-up
        t = time.perf_counter(); rollout(); c = time.perf_counter() - t
        t = time.perf_counter(); alg.update(); u = time.perf_counter() - t
        if i > 0:
            coll.append(c); upd.append(u)
    return {"envs": N, "steps_per_env": T, "rollout_side": med(coll), "update": med(upd),
            "what": "reference HIMPPO: act x T + process_env_step x T + compute_returns (no simulator); update() = 5 epochs x 4 minibatches"}


if __name__ == "__main__":
    torch.set_num_threads(os.cpu_count() or 1)
    res = {"host": {"cpu_model": cpu_model(), "nproc": os.cpu_count(), "torch_threads": torch.get_num_threads(), "torch": torch.__version__},
           "note": "reference Python on this container's CPU cores; no physics in either leg (Isaac Gym / PhysX absent): upper bounds on the "
                   "reference's CPU path, measured with 3 warm-up calls / one warm-up round and the repeats listed",
           "B_env_side": [leg_b(64), leg_b(4096)], "C_learner": [leg_c(64), leg_c(4096)]}
    out = os.path.join(ROOT, "profiles", "ref_cpu_partial.json")
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))
