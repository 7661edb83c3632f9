"""Row (b) of SURVEY.md section 8, the sentence that was untested through round 3: the REFERENCE's own runner class
(rsl_rl/rsl_rl/runners/him_on_policy_runner.py:44-157, imported unchanged from /root/reference) drives the product's LeggedRobot surface.
Container only (the reference never travels to the GPU box, and the container has no GPU): the simulator behind the surface is the CPU lane
emulator of the kernel sources (tests/emu_env.py), everything above the C-ABI is the product's Python."""
import os
import sys

import pytest
import torch

from helpers import C

REF = "/root/reference/rsl_rl"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not present")


def _reference_runner_class():
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
    import refenv
    refenv.install()                         # dummy tensorboard writer; puts /root/reference/rsl_rl on the path
    from rsl_rl.runners import HIMOnPolicyRunner
    return HIMOnPolicyRunner


@pytest.mark.parametrize("task", ["aliengo", "aliengo_stairs"])
def test_reference_himonpolicyrunner_drives_the_env_surface(task, tmp_path):
    from emu_env import EmuLeggedRobot
    Runner = _reference_runner_class()
    assert Runner.__module__.startswith("rsl_rl.") and "/root/reference" in sys.modules[Runner.__module__].__file__
    cfg, tcfg = C.TASKS[task][0](), C.TASKS[task][1]()
    cfg.env.num_envs = 16
    cfg.env.episode_length_s = 0.6            # 30 steps: time-out resets inside the two iterations
    env = EmuLeggedRobot(cfg, seed=3)
    train_cfg = {k: tcfg.to_dict()[k] for k in ("runner", "algorithm", "policy")}
    train_cfg["runner"]["num_steps_per_env"] = 24
    torch.manual_seed(0)
    runner = Runner(env, train_cfg, log_dir=str(tmp_path), device="cpu")     # HIMR:84: env.reset() inside
    seen = {"steps": 0, "resets": 0, "episode": 0}
    step = env.step

    def checked_step(actions):               # what the reference's loop receives from env.step (HIMR:115), checked on every call
        out = step(actions)
        obs, priv, rew, dones, infos, ids, term_priv = out
        assert obs.shape == (16, 270) and priv.shape == (16, 238) and rew.shape == (16,) and dones.shape == (16,)
        assert obs.dtype == priv.dtype == rew.dtype == torch.float32 and dones.dtype == torch.bool and ids.dtype == torch.int64
        assert term_priv.shape == (len(ids), 238) and torch.equal(ids, dones.nonzero(as_tuple=False).flatten())
        assert infos["time_outs"].shape == (16,) and infos["time_outs"].dtype == torch.bool
        assert torch.isfinite(obs).all() and torch.isfinite(rew).all()
        seen["steps"] += 1
        seen["resets"] += len(ids)
        seen["episode"] += int("episode" in infos and len(ids) > 0)
        return out
    env.step = checked_step
    before = {k: v.clone() for k, v in runner.alg.actor_critic.state_dict().items()}
    runner.learn(2, init_at_random_ep_len=True)                               # HIMR:86-157, unchanged
    after = runner.alg.actor_critic.state_dict()
    assert seen["steps"] == 2 * 24 and seen["resets"] > 0 and seen["episode"] > 0
    assert any(not torch.equal(before[k], after[k]) for k in before) and all(torch.isfinite(v).all() for v in after.values())
    st = runner.alg.storage                                                   # the reference's HIMRolloutStorage, filled from our tuples
    assert st.observations.shape == (24, 16, 270) and st.privileged_observations.shape == (24, 16, 238)
    assert st.next_privileged_observations.shape == (24, 16, 238) and st.dones.dtype == torch.uint8
    assert os.path.exists(os.path.join(str(tmp_path), "model_2.pt"))          # HIMR:157 save()
    # the checkpoint the reference's runner wrote loads into the build's runner class (same keys, HIMR:233-240)
    from isaacgymloco_amd.learn.modules import HIMActorCritic
    ck = torch.load(os.path.join(str(tmp_path), "model_2.pt"), weights_only=False)
    mine = HIMActorCritic(270, 238, 45, 12, **train_cfg["policy"])
    mine.load_state_dict(ck["model_state_dict"])
    env.close()


def test_reference_hybridpolicyrunner_drives_the_amp_env_surface(tmp_path):
    """BASELINE config 4's runner: the reference's HybridPolicyRunner (rsl_rl/runners/hybrid_runner.py:54-366, imported unchanged) with its own AMPLoader,
    AMPDiscriminator, Normalizer and ReplayBuffer on the product's AMP surface -- the 8-tuple of LR:173-176 with terminal AMP states, get_amp_observations(),
    dof_pos_limits (HYBR:118-120), update_reward_curriculum (HYBR:162)."""
    import glob
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
    import refenv
    refenv.install()
    from rsl_rl.runners import HybridPolicyRunner
    assert "/root/reference" in sys.modules[HybridPolicyRunner.__module__].__file__
    from emu_env import EmuLeggedRobot
    cfg, tcfg = C.TASKS["aliengo_amp"][0](), C.TASKS["aliengo_amp"][1]()
    cfg.env.num_envs = 16
    cfg.env.episode_length_s = 0.6
    env = EmuLeggedRobot(cfg, seed=3, using_amp=True)
    train_cfg = {k: tcfg.to_dict()[k] for k in ("runner", "algorithm", "policy")}
    train_cfg["runner"]["num_steps_per_env"] = 24
    train_cfg["runner"]["amp_num_preload_transitions"] = 2000
    train_cfg["runner"]["amp_motion_files"] = sorted(glob.glob("/root/reference/datasets/mocap_motions_aliengo/*"))[:7]
    train_cfg["algorithm"]["amp_replay_buffer_size"] = 4000
    assert train_cfg["runner"]["amp_motion_files"], "the reference's Aliengo clips"
    torch.manual_seed(0)
    import numpy as np
    np.random.seed(0)
    runner = HybridPolicyRunner(env, train_cfg, log_dir=str(tmp_path), device="cpu")
    seen = {"steps": 0, "resets": 0}
    step = env.step

    def checked_step(actions):               # HYBR:183: the 8-tuple
        out = step(actions)
        obs, priv, rew, dones, infos, ids, term_priv, term_amp = out
        assert obs.shape == (16, 270) and priv.shape == (16, 238) and term_priv.shape == (len(ids), 238) and term_amp.shape == (len(ids), 30)
        assert env.get_amp_observations().shape == (16, 30) and torch.isfinite(term_amp).all()
        seen["steps"] += 1
        seen["resets"] += len(ids)
        return out
    env.step = checked_step
    before = {k: v.clone() for k, v in runner.alg.discriminator.state_dict().items()}
    runner.learn(2, init_at_random_ep_len=True)
    assert seen["steps"] == 2 * 24 and seen["resets"] > 0
    after = runner.alg.discriminator.state_dict()
    assert any(not torch.equal(before[k], after[k]) for k in before) and all(torch.isfinite(v).all() for v in after.values())
    assert runner.alg.amp_storage.num_samples == 2 * 24 * 16                  # every transition pair went into the reference's replay buffer
    env.close()
