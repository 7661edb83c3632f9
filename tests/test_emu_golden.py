"""CPU lane-emulation of the HIP kernel sources (isaacgymloco_amd/csrc/ls_*.h) against the reference golden
vectors and against the oracle's physics.  Development aid for the GPU kernels' lane orchestration; the
authoritative parity tests are the -m gpu ones that call the HIP library through the C-ABI."""
import numpy as np
import pytest

import golden_replay as GR
from helpers import LC, aliengo, make_oracle, quiet_cfg, abi


def make_emu_from_fixture(name, big=False):
    import emu_binding
    fx = GR.load_big(name) if big else GR.load(name)
    cfg = GR.big_scenario_cfg(name) if big else GR.scenario_cfg(name)
    N = int(fx["num_envs"])
    model = aliengo.build_model()
    ter = GR.FixtureTerrain(fx)
    lc = LC.make_lsim_config(cfg, num_envs=N, terrain=ter, model=model, seed=int(fx["seed"]))
    return fx, emu_binding.EmuSim(lc, model, ter.heightsamples, ter.env_origins)


@pytest.mark.parametrize("name", GR.SCENARIOS)
def test_emu_matches_reference_step(name):
    fx, sim = make_emu_from_fixture(name)

    def get(n):
        return np.array(sim.buf[n])

    def put(n, a):
        sim.buf[n][...] = a
    for t, ref in GR.replay(fx, sim, get, put):
        GR.compare_step(t, ref, get, sim.stats_row)
    if "fin_ids" in fx.files:     # the reference's by-hand reset_idx(env_ids): kernel B's masked mode (lsim_reset_envs)
        mask = GR.replay_final_reset(fx, sim, get, put)
        GR.compare_final_reset(fx, mask, get, sim.stats_row)


@pytest.mark.parametrize("name", GR.BIG_SCENARIOS)
def test_emu_matches_reference_step_at_baseline_size(name):
    fx, sim = make_emu_from_fixture(name, big=True)

    def get(n):
        return np.array(sim.buf[n])

    def put(n, a):
        sim.buf[n][...] = a
    for t, ref in GR.replay(fx, sim, get, put):
        GR.compare_step(t, ref, get, sim.stats_row)
