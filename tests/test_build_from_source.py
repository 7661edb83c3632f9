"""build(): the HIP library compiles from its sources alone.  `csrc/build.py` skips hipcc when an up-to-date .so is present (the artefact
travels with the tree but is git-ignored), so a fresh clone has never been exercised by the other tests: this one cross-compiles
the library for gfx950 into a scratch directory (no GPU needed, ~40 s) and checks that every symbol include/lsim.h declares is there."""
import ctypes
import os
import subprocess

import pytest

from helpers import ROOT, abi


def test_from_source_compile_exports_the_abi(tmp_path):
    from isaacgymloco_amd.csrc import build as B
    out = os.path.join(tmp_path, "liblsim_fresh.so")
    B.build_variant(out, workdir=str(tmp_path))      # both translation units from source, with the product's flag sets
    assert os.path.getsize(out) > 100_000
    L = ctypes.CDLL(out)                          # loading needs no GPU; no compute call is made
    missing = [f for f in abi.declared_functions() if not hasattr(L, f)]
    assert not missing, missing
    abi.check_abi(L, prefix="lsim")
    L.lsim_abi_version.restype = ctypes.c_int
    assert L.lsim_abi_version() == abi.ABI_VERSION


def test_build_rebuilds_when_the_artefact_is_missing(tmp_path, monkeypatch):
    """stale(): no .so -> must compile; here only the decision is checked (the compile itself is the test above)"""
    from isaacgymloco_amd.csrc import build as B
    monkeypatch.setattr(B, "LIB", os.path.join(tmp_path, "absent.so"))
    assert B.stale()


_DIGEST_SNIPPET = r"""
import hashlib, sys, numpy as np, torch
sys.path.insert(0, {root!r}); sys.path.insert(0, {root!r} + "/tests")
from helpers import C
from hip_backend import HipBackend
from isaacgymloco_amd.envs import terrain as T
cfg = C.aliengo_cfg(); cfg.terrain.terrain_proportions = [0.5, 0.0, 0.0, 0.0, 0.25, 0.25]
be = HipBackend(cfg, 64, T.Terrain(cfg.terrain, 64, seed=1), seed=7)
be.reset_all()
rs = np.random.RandomState(0)
h = hashlib.sha256()
for t in range(4):
    be.step(rs.normal(0, 1, (64, 12)).astype(np.float32))
    for k in ("obs", "priv_obs", "rew", "reset", "root_states", "dof_state", "contact_forces"):
        h.update(be.get(k).tobytes())
print("DIGEST", h.hexdigest())
"""


@pytest.mark.gpu
def test_library_compiled_on_this_machine_reproduces_the_shipped_one(tmp_path):
    """the GPU tests load a liblsim.so that travelled with the tree.  Here the library is compiled FROM SOURCE on the GPU box
    (both translation units, the product's flags), loaded through LSIM_LIB in a fresh process, and four full-physics steps at N = 64 must
    reproduce the shipped library's outputs bit for bit (same compiler, same flags: same code)."""
    import subprocess
    import sys
    from isaacgymloco_amd.csrc import build as B
    fresh = os.path.join(tmp_path, "liblsim_fresh.so")
    B.build_variant(fresh, workdir=str(tmp_path))
    code = _DIGEST_SNIPPET.format(root=ROOT)

    def digest(env_extra):
        env = dict(os.environ)
        env.pop("LSIM_LIB", None)
        env.update(env_extra)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        return [l for l in r.stdout.splitlines() if l.startswith("DIGEST")][-1]
    assert digest({"LSIM_LIB": fresh}) == digest({})


def test_wavefront_fences_only_in_one_wave_kernels():
    """LS_WAVE_SYNC (ls_math.h) replaces __syncthreads() by wavefront-scope fences: correct only where the workgroup IS one wavefront.  Every kernel
    that runs the wave drivers must be compiled for, and launched with, 64 threads; kernels of more than one wave keep __syncthreads()."""
    import re
    from isaacgymloco_amd.csrc import build as B
    src = open(os.path.join(B.HERE, "lsim_hip.hip")).read()
    # kernels (or kernel macros) whose body reaches a wave driver
    bodies = re.findall(r"__global__\s+__launch_bounds__\((\d+)\)[^\n]*\n((?:[^\n]*\n){1,12}?)\}", src)
    wave = [(int(n), body) for n, body in bodies if re.search(r"ls_wave_step_[ab]|ls_wave_debug|LS_PHASE|LS_WAVE_SYNC|wc_", body)]
    assert wave, "no wave-driver kernel found: the pattern of this test is out of date"
    assert all(n == 64 for n, _ in wave), [n for n, _ in wave]
    for name in ("lsim_k_step_a_tgs", "lsim_k_step_a_pgs", "lsim_k_step_b"):
        launches = re.findall(r"hipLaunchKernelGGL\(\s*" + name + r"\s*,\s*dim3\(.*?\)\s*,\s*dim3\((\d+)\)", src)
        assert launches and all(int(t) == 64 for t in launches), (name, launches)
    multi = [body for n, body in bodies if int(n) > 64]
    assert multi and not any("LS_WAVE_SYNC" in body or "LS_PHASE" in body for body in multi)
    # the learn translation unit (multi-wave blocks everywhere) never uses the macro
    for f in set(B.LEARN_HEADERS + ["lsim_learn.hip"]) - set(B.SIM_HEADERS):       # (ls_math.h, where the macro is defined, belongs to both)
        assert "LS_WAVE_SYNC" not in open(os.path.join(B.HERE, f)).read(), f
