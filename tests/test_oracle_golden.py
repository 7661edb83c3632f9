"""Pin the CPU oracle (oracle/lsim_oracle.c) against golden vectors captured from the reference's own
LeggedRobot.step() torch code (tools/gen_golden.py).  Runs on CPU."""
import numpy as np
import pytest

import golden_replay as GR
from helpers import LC, aliengo


def make_oracle_from_fixture(name, big=False):
    from oracle import oracle
    fx = GR.load_big(name) if big else GR.load(name)
    cfg = GR.big_scenario_cfg(name) if big else GR.scenario_cfg(name)
    N = int(fx["num_envs"])
    model = aliengo.build_model()
    ter = GR.FixtureTerrain(fx)
    lc = LC.make_lsim_config(cfg, num_envs=N, terrain=ter, model=model, seed=int(fx["seed"]))
    sim = oracle.OracleSim(lc, model, ter.heightsamples, ter.env_origins)
    return fx, sim


@pytest.mark.parametrize("name", GR.SCENARIOS)
def test_oracle_matches_reference_step(name):
    fx, sim = make_oracle_from_fixture(name)

    def get(n):
        return np.array(sim.buf[n])

    def put(n, a):
        sim.buf[n][...] = a
    worst = {}
    nsteps = 0
    for t, ref in GR.replay(fx, sim, get, put):
        errs = GR.compare_step(t, ref, get, sim.stats_row)
        for k, v in errs.items():
            worst[k] = max(worst.get(k, 0.0), v)
        nsteps += 1
    assert nsteps == fx["in_actions"].shape[0]
    print(name, "max abs err:", {k: f"{v:.2e}" for k, v in worst.items()})
    if "fin_ids" in fx.files:     # the reference's reset_idx(env_ids) called by hand after the last step (LR:290): orc_reset_envs
        mask = GR.replay_final_reset(fx, sim, get, put)
        GR.compare_final_reset(fx, mask, get, sim.stats_row)


@pytest.mark.parametrize("name", GR.BIG_SCENARIOS)
def test_oracle_matches_reference_step_at_baseline_size(name):
    """N = 4096 (BASELINE cfg 2-4): the index-dependent logic -- high-velocity commands of the first 20 % of the envs (LR:649), terrain columns
    (LR:1234), the stumble slices (LR:72-90, 1597-1607) -- around the command-curriculum step 999 -> 1000, against the reference's own outputs."""
    fx, sim = make_oracle_from_fixture(name, big=True)

    def get(n):
        return np.array(sim.buf[n])

    def put(n, a):
        sim.buf[n][...] = a
    nsteps = 0
    for t, ref in GR.replay(fx, sim, get, put):
        GR.compare_step(t, ref, get, sim.stats_row)
        nsteps += 1
    assert nsteps == len(fx["in_counter_before"]) == 3
