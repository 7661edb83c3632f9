"""A robot whose simulated state holds a NaN / infinity is a REPORTED quantity (include/lsim.h: LSIM_BUF_NONFINITE, LSIM_STATS_NONFINITE), not a timing
anomaly ("kernel A at 0.19 ms means non-finite robot states" was how rounds 4-5 found it).  CPU: oracle and lane emulator (the kernel
sources) agree on the count; GPU: the HIP library through the product binding; the C-ABI's failure paths leave the handle's counters unchanged."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

from helpers import ROOT, abi, make_oracle, quiet_cfg


def _poison_and_count(put, get, step, stats_row_of, N):
    """NaN joint angle in env 1, infinite joint velocity in env 3: counted in the step that simulated them; on a terrain with a border their NaN position
    fails in_terrain_range (TER:220-227: every comparison with NaN is false), so check_termination resets them in that same step (LR:255-259) and the next
    step is clean again.  Healthy envs are never counted."""
    act = np.zeros((N, 12), np.float32)
    step(act)
    assert get("nonfinite")[0] == 0
    dof = get("dof_state").copy()
    dof[1, 4, 0] = np.nan
    dof[3, 0, 1] = np.inf
    put("dof_state", dof)
    step(act)
    assert get("nonfinite")[0] == 2 and get("stats")[stats_row_of(), abi.STATS["nonfinite"]] == 2.0
    assert bool(get("reset")[1]) and bool(get("reset")[3])
    step(act)
    assert get("nonfinite")[0] == 2 and get("stats")[stats_row_of(), abi.STATS["nonfinite"]] == 0.0       # per-step word: this step's; the buffer: cumulative
    assert get("nonfinite")[1] == 2                                                                          # the step counter of the latest such step


def test_oracle_and_emulator_count_nonfinite_robots():
    import emu_binding
    N = 6
    cfg = quiet_cfg()
    orc, lc, model, ter = make_oracle(cfg, N, seed=2)
    emu = emu_binding.EmuSim(lc, model, ter.heightsamples, ter.env_origins)
    for be in (orc, emu):
        be.reset_all()
        def put(name, arr, be=be):
            be.buf[name][...] = arr
        _poison_and_count(put, lambda k, be=be: np.array(be.buf[k]), be.step, lambda be=be: be.stats_row, N)
    orc.close()


def test_config_validation_rejects_unbounded_limit_passes():
    """tgs_limit_passes / lin_vel_at_com came out of reserved words and were only range-checked in Python"""
    from helpers import C, T, LC
    from isaacgymloco_amd import lib
    from oracle import oracle
    L = lib.load()
    c = C.aliengo_cfg()
    c.terrain.terrain_proportions = [1.0, 0, 0, 0]
    ter = T.Terrain(c.terrain, 16)
    n = ctypes.c_size_t()
    for field, bad in (("tgs_limit_passes", 1000), ("tgs_limit_passes", -1), ("lin_vel_at_com", 2)):
        lc = LC.make_lsim_config(c, num_envs=16, terrain=ter)
        assert L.lsim_query_arena(ctypes.byref(lc), ctypes.byref(n)) == 0
        setattr(lc, field, bad)
        assert L.lsim_query_arena(ctypes.byref(lc), ctypes.byref(n)) == abi.E_INVALID, (field, bad)
        from isaacgymloco_amd.envs.legged_robot import build_robot_model
        with pytest.raises(Exception):
            oracle.OracleSim(lc, build_robot_model(c.asset), ter.heightsamples, ter.env_origins)


@pytest.mark.gpu
def test_hip_counts_nonfinite_robots_and_reports_them():
    import torch
    from hip_backend import HipBackend
    N = 6
    be = HipBackend(quiet_cfg(), N, None, seed=2)
    be.reset_all()
    _poison_and_count(be.put, be.get, be.step, lambda: be.stats_row, N)
    assert int(be.env.nonfinite_envs) == 2 and int(be.env.extras["nonfinite_envs"]) == 2                    # the live tensor LeggedRobot.extras carries
    be.env.buf["nonfinite"].zero_()                                                                          # "the caller may zero it"
    st = be.env.state_dict()
    assert st["conventions"]["lin_vel_at_com"] in (0, 1) and st["conventions"]["abi_version"] == abi.ABI_VERSION
    with pytest.warns(UserWarning):
        be.env.load_state_dict({k: v for k, v in st.items() if k != "conventions"})                          # a checkpoint from before round 6
    torch.cuda.synchronize()


_FAILED_LAUNCH = r"""
import ctypes, sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import torch
from helpers import quiet_cfg
from hip_backend import HipBackend
be = HipBackend(quiet_cfg(), 4, None, seed=2)
be.reset_all()
hip = ctypes.CDLL("libamdhip64.so")
s = ctypes.c_void_p()
assert hip.hipStreamCreate(ctypes.byref(s)) == 0
assert hip.hipStreamDestroy(s) == 0                      # a stale handle: every launch on it is refused
L, h = be.env._L, be.env._h
c0, r0 = be.step_counter, be.stats_row
a = torch.zeros(4, 12, device="cuda:0")
rc = L.lsim_step_ex(h, a.data_ptr(), 0, s)
print("rc", rc, "counter", be.step_counter - c0, "row", be.stats_row - r0, "err", L.lsim_last_error(h).decode())
rc2 = L.lsim_step_ex(h, a.data_ptr(), 0, None)          # the handle still works
torch.cuda.synchronize()
print("rc2", rc2, "counter", be.step_counter - c0)
"""


@pytest.mark.gpu
def test_failed_launch_leaves_the_handle_where_it_was():
    """lsim_step_ex on a destroyed stream: the launch is refused, step counter and stats row do not advance (in a child process: a runtime that
    crashed on the stale handle instead of refusing it must not take the test session with it)"""
    code = _FAILED_LAUNCH % (ROOT, os.path.join(ROOT, "tests"))
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    out = p.stdout
    if p.returncode != 0 and "rc " not in out:
        pytest.skip(f"the HIP runtime does not refuse a stale stream handle gracefully on this box: {p.stderr[-300:]}")
    line = [l for l in out.splitlines() if l.startswith("rc ")][0].split()
    rc, counter, row = int(line[1]), int(line[3]), int(line[5])
    if rc == 0:
        pytest.skip("the HIP runtime accepted a launch on a destroyed stream: no failure to observe")
    assert rc == abi.E_HIP and counter == 0 and row == 0, out
    line2 = [l for l in out.splitlines() if l.startswith("rc2 ")][0].split()
    assert int(line2[1]) == 0 and int(line2[3]) == 1, out
