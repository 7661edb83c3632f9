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


def test_integration_document_names_every_entry_point():
    """INTEGRATION.md's table maps the reference's calls to the C-ABI: an entry point added to include/lsim.h without a row there is undocumented"""
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    missing = [fn for fn in abi.declared_functions() if fn not in doc]
    assert not missing, missing


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


def test_learner_workspace_queries_validate_shapes():
    """the host-side planners of the learner entry points (no launches): sizes grow with the batch, unsupported shapes are refused"""
    from isaacgymloco_amd import lib
    L = lib.load()             # loading needs no GPU; only the launches do
    n, parts = ctypes.c_size_t(), ctypes.c_int()
    B = 102400
    # weight gradient: every layer of the three networks has a plan; a 1024 x 1024 layer (discriminator-sized) is left to BLAS
    for k_in, n_out in ((270, 128), (128, 64), (64, 19), (64, 512), (512, 256), (256, 128), (128, 12), (238, 512), (128, 1), (45, 128), (64, 16)):
        assert L.lsim_linear_wgrad_workspace(B, k_in, n_out, ctypes.byref(n), ctypes.byref(parts)) == 0, (k_in, n_out)
        assert parts.value >= 1 and n.value >= parts.value * (k_in * n_out + n_out) * 4
    assert L.lsim_linear_wgrad_workspace(B, 1024, 1024, ctypes.byref(n), ctypes.byref(parts)) == abi.E_UNSUPPORTED
    assert L.lsim_linear_wgrad_workspace(0, 64, 64, ctypes.byref(n), ctypes.byref(parts)) == abi.E_UNSUPPORTED
    # Sinkhorn / estimator loss head: K <= 64 prototypes, latent <= 32
    assert L.lsim_sinkhorn_workspace(B, 32, ctypes.byref(n)) == 0 and n.value >= B * 32 * 4
    assert L.lsim_sinkhorn_workspace(B, 65, ctypes.byref(n)) == abi.E_INVALID
    small = ctypes.c_size_t()
    assert L.lsim_estimator_loss_workspace(4096, 16, 32, ctypes.byref(small)) == 0
    assert L.lsim_estimator_loss_workspace(B, 16, 32, ctypes.byref(n)) == 0 and n.value > small.value
    assert n.value >= (2 * B * 16 + 2 * B * 32) * 4          # z and the scores of both matrices at least (E = exp(S / eps) is formed again where it is used)
    assert L.lsim_estimator_loss_workspace(B, 33, 32, ctypes.byref(n)) == abi.E_INVALID
    assert L.lsim_estimator_loss_workspace(B, 16, 65, ctypes.byref(n)) == abi.E_INVALID
    assert L.lsim_estimator_loss_workspace(B, 16, 32, None) == abi.E_INVALID
    assert L.lsim_ppo_loss_workspace(B, ctypes.byref(n)) == 0 and n.value > 0
