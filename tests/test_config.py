"""Task configs restate the reference's config classes leaf by leaf (golden JSON captured with helpers.class_to_dict)."""
import json
import os

import pytest

from helpers import C, ROOT


@pytest.mark.parametrize("task", ["aliengo", "aliengo_stairs", "aliengo_amp", "aliengo_recover"])
def test_config_matches_reference(task):
    ref = json.load(open(os.path.join(ROOT, "tests", "golden", f"ref_cfg_{task}.json")))
    env, ppo = C.TASKS[task]
    mine_env, mine_ppo = env().to_dict(), ppo().to_dict()
    ref["ppo"]["runner"].pop("load_run", None); mine_ppo["runner"].pop("load_run", None)          # a host path in the reference
    ref["ppo"]["runner"].pop("amp_motion_files", None); mine_ppo["runner"].pop("amp_motion_files", None)  # absolute paths there
    assert mine_env == ref["env"]
    assert mine_ppo == ref["ppo"]


def test_active_reward_sets():
    from helpers import abi, LC, T
    c = C.aliengo_cfg(); c.terrain.terrain_proportions = [1.0, 0, 0, 0]
    lc = LC.make_lsim_config(c, num_envs=64, terrain=T.Terrain(c.terrain, 64))
    active = [abi.REWARD_NAMES[i] for i in range(abi.NUM_REWARD_TERMS) if lc.reward_scales[i] != 0]
    assert len(active) == 21 and "feet_mirror" in active and "collision" not in active
    assert lc.max_episode_length == 1000 and lc.push_interval == 800 and lc.resampling_steps == 500
    assert (lc.dof_init_vel_range[0], lc.dof_init_vel_range[1]) == (-1.0, 1.0)     # LR:708 reads an absent key
