"""GPU: the element-parallel kinematics (wc_kinematics: one rotation-matrix element per lane, DPP quad broadcasts / row rotations) against
the per-leg form it replaces (ph_kinematics, the one the CPU lane emulator runs) on random states: every array of the phase -- R, p, S,
V, Ab of all 17 bodies -- to 1e-4.  Uses the diagnostics build (-DLS_DEBUG_KIN, tools/kin_check.py), compiled on the spot if absent."""
import ctypes
import os
import sys

import numpy as np
import pytest

from helpers import C, ROOT

pytestmark = pytest.mark.gpu


def test_element_parallel_kinematics_matches_per_leg_form():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import kin_check
    from isaacgymloco_amd.csrc import build as B
    src_time = max(os.path.getmtime(f) for f in B.all_sources())
    if not os.path.exists(kin_check.OUT) or os.path.getmtime(kin_check.OUT) < src_time:
        kin_check.build()
    from isaacgymloco_amd.envs.legged_robot import LeggedRobot
    L = ctypes.CDLL(kin_check.OUT)
    L.lsim_debug_kinematics.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    for task in ("aliengo", "go1"):
        cfg = C.TASKS[task][0]()
        cfg.env.num_envs = 8
        env = LeggedRobot(cfg, sim_device="cuda:0", seed=1)     # product library: only supplies a handle with the model table on the device
        rs = np.random.RandomState(0)
        for t in range(8):
            q4 = rs.normal(0, 1, 4); q4 /= np.linalg.norm(q4)
            st = np.concatenate([rs.normal(0, 1, 3), q4, rs.normal(0, 1, 6), rs.uniform(-1.5, 1.5, 12), rs.normal(0, 5, 12)]).astype(np.float32)
            o = [np.zeros(480, np.float32) for _ in range(2)]
            for v in range(2):
                assert L.lsim_debug_kinematics(env._h, st.ctypes.data, o[v].ctypes.data, v) == 0
            np.testing.assert_allclose(o[1], o[0], rtol=1e-4, atol=1e-4, err_msg=f"{task} trial {t}")
            assert np.abs(o[0]).max() > 1.0
        env.close()
