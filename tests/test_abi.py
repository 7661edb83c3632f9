"""CPU checks of the C-ABI: the HIP library loads and exports every symbol include/lsim.h declares (no compute
calls here: there is no GPU), struct mirrors agree, argument validation works on the host side."""
import ctypes
import os

import pytest

from helpers import ROOT, abi


def test_header_parses_and_mirrors():
    assert ctypes.sizeof(abi.LsimConfig) > 1000
    assert abi.NUM_REWARD_TERMS == 51
    assert abi.REWARD_NAMES == sorted(abi.REWARD_NAMES), "reward ids must be in the reference's alphabetical order"
    assert abi.BUFFER_IDS["obs"] == 0


def test_library_exports_every_declared_symbol():
    from isaacgymloco_amd.csrc import build
    path = build.build()
    L = ctypes.CDLL(path)
    for fn in abi.declared_functions():
        assert hasattr(L, fn), f"liblsim.so does not export {fn}"
    abi.check_abi(L, "lsim")
    L.lsim_reward_name.restype = ctypes.c_char_p
    L.lsim_buffer_name.restype = ctypes.c_char_p
    assert [L.lsim_reward_name(i).decode() for i in range(abi.NUM_REWARD_TERMS)] == abi.REWARD_NAMES
    names = sorted(abi.BUFFER_IDS, key=abi.BUFFER_IDS.get)
    assert [L.lsim_buffer_name(i).decode() for i in range(abi.NUM_BUFFERS)] == names


def test_query_arena_validates_config():
    from isaacgymloco_amd.csrc import build
    L = ctypes.CDLL(build.build())
    cfg = abi.LsimConfig()
    n = ctypes.c_size_t()
    assert L.lsim_query_arena(ctypes.byref(cfg), ctypes.byref(n)) == abi.DEFINES["LSIM_E_ABI"] - (1 << 32) or \
        L.lsim_query_arena(ctypes.byref(cfg), ctypes.byref(n)) == -5
    from helpers import C, T, LC, aliengo
    c = C.aliengo_cfg()
    c.terrain.terrain_proportions = [1.0, 0, 0, 0]
    ter = T.Terrain(c.terrain, 64)
    lc = LC.make_lsim_config(c, num_envs=64, terrain=ter)
    assert L.lsim_query_arena(ctypes.byref(lc), ctypes.byref(n)) == 0
    assert n.value > 64 * (270 + 238) * 4


def test_env_refuses_to_run_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from isaacgymloco_amd import lib
    from isaacgymloco_amd.envs.legged_robot import LeggedRobot
    from helpers import C
    with pytest.raises(lib.LsimError):
        LeggedRobot(C.aliengo_cfg(), sim_device="cuda:0")
