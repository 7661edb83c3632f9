"""GPU: the wall path of the terrain contact query spread over the wave (ls_physics.h: wc_wall_contacts -- seven points per round, one lane
per cell of a point's 3 x 3 block, the owner picks the first strict minimum) against the serial walk it replaces (every point's own lane
walks its nine cells; the form the CPU lane emulator runs), as two builds of the library stepping the same robots on the stairs task:
the product build and a -DLS_SERIAL_WALLS build compiled on the spot.  Same candidates, same order, same arithmetic: the states agree
bit for bit, step after step from identical states."""
import os

import numpy as np
import pytest

from helpers import C, ROOT

pytestmark = pytest.mark.gpu

VARIANT = os.path.join(ROOT, "isaacgymloco_amd", "csrc", "variants", "liblsim_serialwalls.so")


def test_wall_path_over_the_wave_matches_the_serial_walk():
    import torch
    from isaacgymloco_amd import abi, lib
    from isaacgymloco_amd.csrc import build as B
    from isaacgymloco_amd.envs.legged_robot import LeggedRobot
    src_time = max(os.path.getmtime(f) for f in B.all_sources())
    if not os.path.exists(VARIANT) or os.path.getmtime(VARIANT) < src_time:
        B.build_variant(VARIANT, ["-DLS_SERIAL_WALLS"])
    serial = lib.load_path(VARIANT)

    class SerialWalls(LeggedRobot):
        def _load_library(self):
            return serial

    N = 2048
    envs = []
    for cls in (LeggedRobot, SerialWalls):
        cfg = C.TASKS["aliengo_stairs"][0]()
        cfg.env.num_envs = N
        cfg.terrain.curriculum = False          # robots spread over all levels: tall steps from the first step on
        envs.append(cls(cfg, sim_device="cuda:0", seed=5))
    ea, eb = envs
    ea.reset(); eb.reset()
    g = torch.Generator(device="cuda:0").manual_seed(2)
    wall_contacts = 0
    for t in range(60):
        for name in abi.BUFFER_IDS:              # identical states before every step
            eb.buf[name].copy_(ea.buf[name])
        act = torch.randn(N, 12, device="cuda:0", generator=g)
        ea.step_device(act); eb.step_device(act)
        torch.cuda.synchronize()
        for name in ("root_states", "dof_state", "contact_forces", "rigid_body_states", "rew", "obs"):       # measured: bit for bit
            assert torch.equal(ea.buf[name], eb.buf[name]), f"step {t}: {name} differs between the two forms of the wall path"
        assert torch.equal(ea.reset_buf, eb.reset_buf), f"step {t}: different robots reset"
        feet = ea.buf["contact_forces"][:, [4, 8, 12, 16], :]
        wall_contacts += int(((feet[..., :2].norm(dim=-1) > 2.0 * feet[..., 2].abs()) & (feet.norm(dim=-1) > 5.0)).sum())
    print(f"mostly-horizontal foot forces seen {wall_contacts}")
    assert wall_contacts > 50, "the window must contain contacts against risers"
