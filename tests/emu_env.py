"""TEST INFRASTRUCTURE -- the product's LeggedRobot Python surface (isaacgymloco_amd/envs/legged_robot.py: the class a user of the reference
switches to) with the CPU lane emulator of the kernel sources (tests/emu, tests/emu_binding.py) in place of the HIP library, so that code
which only exists in the build container -- the REFERENCE's own runner classes -- can drive that surface end to end without a GPU.
Everything above the C-ABI is the product's code unchanged: constructor, buffer binding, step() / reset() / reset_idx(), extras, the
attribute names runners read.  Nothing under isaacgymloco_amd/ imports this module."""
import ctypes

import torch

import emu_binding
from isaacgymloco_amd import abi
from isaacgymloco_amd.envs.legged_robot import LeggedRobot


class _EmuApi:
    """the C-ABI entry points LeggedRobot calls, bound to the emulator's symbols (emu_* = the same ls_api_impl.h compiled by g++)"""

    def __init__(self):
        L = emu_binding.lib()
        vp, i32, u32, i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_uint32, ctypes.c_int64
        sig = {"query_arena": [ctypes.POINTER(abi.LsimConfig), ctypes.POINTER(ctypes.c_size_t)],
               "create": [ctypes.POINTER(abi.LsimConfig), ctypes.POINTER(abi.LsimRobotModel), vp, vp, vp, i32, ctypes.POINTER(vp)],
               "get_buffer": [vp, i32, ctypes.POINTER(vp), ctypes.POINTER(i64), ctypes.POINTER(i32), ctypes.POINTER(i32)],
               "reset_all": [vp, vp], "reset_envs": [vp, vp, vp], "step": [vp, vp, vp], "step_ex": [vp, vp, u32, vp],
               "get_step_counter": [vp, ctypes.POINTER(i64)], "set_step_counter": [vp, i64], "get_stats_row": [vp, ctypes.POINTER(i32)],
               "get_reset_calls": [vp, ctypes.POINTER(u32)], "set_reset_calls": [vp, u32],
               "destroy": [vp]}
        for name, argtypes in sig.items():
            fn = getattr(L, "emu_" + name)
            fn.argtypes = argtypes
            if name == "destroy":
                fn.restype = None
            setattr(self, "lsim_" + name, fn)


class EmuLeggedRobot(LeggedRobot):
    def __init__(self, cfg, sim_params=None, physics_engine=None, sim_device="cpu", headless=True, **kw):
        super().__init__(cfg, sim_params, physics_engine, "cpu", headless, **kw)

    def _load_library(self):
        return _EmuApi()

    def _sync(self):
        pass

    def _stream(self):
        return None
