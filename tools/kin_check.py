"""Diagnostics (GPU): the element-parallel kinematics (wc_kinematics, DPP) against the per-leg form (ph_kinematics) on random states.
Builds isaacgymloco_amd/csrc/variants/liblsim_kindebug.so with -DLS_DEBUG_KIN.   python tools/kin_check.py [--build-only]"""
import ctypes
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "isaacgymloco_amd", "csrc", "variants", "liblsim_kindebug.so")


def build():
    from isaacgymloco_amd.csrc import build as B
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    B.build_variant(OUT, ["-DLS_DEBUG_KIN"])


if __name__ == "__main__":
    if "--build-only" in sys.argv:
        build(); sys.exit(0)
    from isaacgymloco_amd.csrc.build import variant_is_stale
    if variant_is_stale(OUT):
        build()
    os.environ["LSIM_LIB"] = OUT
    from isaacgymloco_amd.envs import config as C
    from isaacgymloco_amd.envs.legged_robot import LeggedRobot
    cfg = C.TASKS["aliengo"][0]()
    cfg.env.num_envs = 8
    env = LeggedRobot(cfg, sim_device="cuda:0", seed=1)
    L = ctypes.CDLL(OUT)
    L.lsim_debug_kinematics.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    rs = np.random.RandomState(0)
    names = [("R", 0, 153), ("p", 153, 204), ("S", 204, 276), ("V", 276, 378), ("Ab", 378, 480)]
    worst = 0.0
    for t in range(5):
        q4 = rs.normal(0, 1, 4); q4 /= np.linalg.norm(q4)
        st = np.concatenate([rs.normal(0, 1, 3), q4, rs.normal(0, 1, 6), rs.uniform(-1, 1, 12), rs.normal(0, 3, 12)]).astype(np.float32)
        o = [np.zeros(480, np.float32) for _ in range(2)]
        for v in range(2):
            assert L.lsim_debug_kinematics(env._h, st.ctypes.data, o[v].ctypes.data, v) == 0
        for nm, a, b in names:
            d = np.abs(o[0][a:b] - o[1][a:b])
            worst = max(worst, d.max())
            if d.max() > 1e-4:
                idx = np.argmax(d)
                print(f"trial {t} {nm}: max diff {d.max():.3e} at {idx}  old {o[0][a + idx]:.5f} new {o[1][a + idx]:.5f}; first rows old {o[0][a:a+9].round(4)} new {o[1][a:a+9].round(4)}")
                bad = np.nonzero(d > 1e-4)[0]
                print("   bad indices:", bad[:40])
    print("worst abs diff", worst)
