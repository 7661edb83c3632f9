"""TEST INFRASTRUCTURE -- builds and binds tests/emu (CPU lane emulator of the HIP kernel sources)."""
import ctypes
import os
import subprocess

import numpy as np

from helpers import ROOT, abi

SRC = os.path.join(ROOT, "tests", "emu", "emu_lsim.cpp")
OUT = os.path.join(ROOT, "tests", "_build", "liblsim_emu.so")
_NP = {abi.DT_F32: np.float32, abi.DT_I64: np.int64, abi.DT_U8: np.uint8, abi.DT_I32: np.int32, abi.DT_I16: np.int16}
_lib = None


def build():
    csrc = os.path.join(ROOT, "isaacgymloco_amd", "csrc")
    deps = [SRC] + [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith(".h")] + \
           [os.path.join(ROOT, "include", f) for f in ("lsim.h", "lsim_layout.h")]
    if not os.path.exists(OUT) or any(os.path.getmtime(d) > os.path.getmtime(OUT) for d in deps):
        os.makedirs(os.path.dirname(OUT), exist_ok=True)
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-Wno-unknown-pragmas", "-o", OUT, SRC])
    return OUT


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        abi.check_abi(_lib, prefix="emu")
    return _lib


class EmuSim:
    def __init__(self, cfg, model, height_grid=None, terrain_origins=None):
        L = lib()
        self.cfg = cfg
        self._h = ctypes.c_void_p()
        gp = op = None
        if height_grid is not None:
            self._g = np.ascontiguousarray(height_grid, np.int16)
            self._o = np.ascontiguousarray(terrain_origins, np.float32)
            gp, op = self._g.ctypes.data_as(ctypes.c_void_p), self._o.ctypes.data_as(ctypes.c_void_p)
        rc = L.emu_create(ctypes.byref(cfg), ctypes.byref(model), gp, op, None, 0, ctypes.byref(self._h))
        assert rc == 0, rc
        self.buf = {}
        for name, bid in abi.BUFFER_IDS.items():
            ptr, shape, nd, dt = ctypes.c_void_p(), (ctypes.c_int64 * 4)(), ctypes.c_int(), ctypes.c_int()
            assert L.emu_get_buffer(self._h, bid, ctypes.byref(ptr), shape, ctypes.byref(nd), ctypes.byref(dt)) == 0
            shp = tuple(shape[i] for i in range(nd.value))
            npdt = np.dtype(_NP[dt.value])
            raw = np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ctypes.c_uint8)), shape=(int(np.prod(shp)) * npdt.itemsize,))
            self.buf[name] = raw.view(npdt).reshape(shp)

    def step(self, actions, flags=0):
        a = np.ascontiguousarray(actions, np.float32)
        assert lib().emu_step_ex(self._h, a.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint32(flags), None) == 0

    def reset_all(self):
        assert lib().emu_reset_all(self._h, None) == 0

    def reset_envs(self, mask):
        m = np.ascontiguousarray(mask, dtype=np.uint8)
        assert lib().emu_reset_envs(self._h, m.ctypes.data_as(ctypes.c_void_p), None) == 0

    @property
    def stats_row(self):
        v = ctypes.c_int()
        lib().emu_get_stats_row(self._h, ctypes.byref(v))
        return v.value

    @property
    def step_counter(self):
        v = ctypes.c_int64()
        lib().emu_get_step_counter(self._h, ctypes.byref(v))
        return v.value

    @step_counter.setter
    def step_counter(self, v):
        lib().emu_set_step_counter(self._h, ctypes.c_int64(v))
