"""AddressSanitizer + UndefinedBehaviorSanitizer over the CPU builds of the shared phase code ("no CPU sanitizer build of oracle/ /
tests/emu").  GPU sanitizers are not available on the pool; the lane emulator compiles the SAME phase functions the HIP kernels run
(isaacgymloco_amd/csrc/ls_*.h), so an out-of-bounds LDS index, a read past a buffer row or signed overflow in them shows up here.
One process (tests/san/san_driver.cpp) runs emulator and oracle side by side on blobs of the config / model / terrain structs."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from helpers import C, LC, ROOT, T

BUILD = os.path.join(ROOT, "tests", "_build")
SAN = os.path.join(BUILD, "san_driver")


def _build():
    os.makedirs(BUILD, exist_ok=True)
    src = [os.path.join(ROOT, "tests", "san", "san_driver.cpp"), os.path.join(ROOT, "tests", "emu", "emu_lsim.cpp")]
    csrc = os.path.join(ROOT, "isaacgymloco_amd", "csrc")
    deps = src + [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith(".h")] + \
        [os.path.join(ROOT, "oracle", f) for f in ("lsim_oracle.c", "orc_physics.c", "orc_internal.h", "orc_philox.h")]
    if os.path.exists(SAN) and all(os.path.getmtime(d) <= os.path.getmtime(SAN) for d in deps):
        return
    flags = ["-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-ffp-contract=off"]
    objs = []
    for f in ("lsim_oracle.c", "orc_physics.c"):
        o = os.path.join(BUILD, "san_" + f.replace(".c", ".o"))
        subprocess.check_call(["gcc", "-std=gnu11", "-c"] + flags + [os.path.join(ROOT, "oracle", f), "-o", o])
        objs.append(o)
    subprocess.check_call(["g++", "-std=c++17", "-Wno-unknown-pragmas"] + flags + src + objs + ["-lm", "-o", SAN])


@pytest.mark.parametrize("task,using_amp", [("aliengo", False), ("aliengo_stairs", False), ("aliengo_amp", True), ("aliengo_allterms", False), ("go1", False)])
def test_emulator_and_oracle_run_clean_under_asan_and_ubsan(task, using_amp, tmp_path):
    _build()
    from helpers import abi
    from isaacgymloco_amd.envs.legged_robot import build_robot_model
    allterms = task == "aliengo_allterms"
    cfg = C.TASKS["aliengo_stairs" if allterms else task][0]()
    if allterms:                                       # every one of the 51 reward terms switched on: more parts than the item table holds
        for name in abi.REWARD_IDS:
            setattr(cfg.rewards.scales, name, -0.01)
    elif task != "aliengo_stairs":
        cfg.rewards.scales.termination = -1.0          # one more active term: touches the termination-reward path as well
    N, steps = 24, 60
    ter = T.Terrain(cfg.terrain, N, seed=1)
    model = build_robot_model(cfg.asset)
    lc = LC.make_lsim_config(cfg, num_envs=N, terrain=ter, model=model, seed=1, using_amp=using_amp)
    (tmp_path / "cfg.bin").write_bytes(bytes(lc))
    (tmp_path / "model.bin").write_bytes(bytes(model))
    np.ascontiguousarray(ter.heightsamples, np.int16).tofile(tmp_path / "grid.bin")
    np.ascontiguousarray(ter.env_origins, np.float32).tofile(tmp_path / "org.bin")
    rs = np.random.RandomState(3)
    acts = rs.normal(0, 1, (steps, N, 12)).astype(np.float32)
    acts[40:] *= 4.0                                       # saturating targets: joint-limit rows, falls, resets
    acts.tofile(tmp_path / "act.bin")
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([SAN, str(tmp_path / "cfg.bin"), str(tmp_path / "model.bin"), str(tmp_path / "grid.bin"), str(tmp_path / "org.bin"),
                        str(steps), str(tmp_path / "act.bin")], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    assert "clean" in r.stdout and "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]
