"""Diagnostics (GPU box): replay a BASELINE-size fixture on the HIP library and on the CPU oracle side by side and list every full-batch difference."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_replay as GR
from helpers import LC, aliengo
from hip_backend import HipBackend
from oracle import oracle

name = sys.argv[1] if len(sys.argv) > 1 else "aliengo"
fx = GR.load_big(name)
cfg = GR.big_scenario_cfg(name)
N = int(fx["num_envs"])
model = aliengo.build_model()
ter = GR.FixtureTerrain(fx)
lc = LC.make_lsim_config(cfg, num_envs=N, terrain=ter, model=model, seed=int(fx["seed"]))
orc = oracle.OracleSim(lc, model, ter.heightsamples, ter.env_origins)
be = HipBackend(GR.big_scenario_cfg(name), N, GR.FixtureTerrain(fx), seed=int(fx["seed"]))
og = lambda n: np.array(orc.buf[n])
def op(n, a): orc.buf[n][...] = a
g1 = GR.replay(fx, orc, og, op)
g2 = GR.replay(fx, be, be.get, be.put)
keys = ["measured_heights", "obs", "priv_obs", "rew", "commands", "torques", "substep_torques", "base_lin_vel", "root_states", "dof_state", "reset", "time_out",
        "episode_sums", "feet_air_time", "last_contacts", "contact_filt", "term_priv_obs", "amp_obs"]
for (t, _), (_, _) in zip(g1, g2):
    for k in keys:
        a, b = og(k).astype(np.float64), be.get(k).astype(np.float64)
        d = np.abs(a - b)
        bad = np.argwhere(d > 1e-5 + 2e-5 * np.abs(a))
        if len(bad):
            print(f"step {t} {k}: {len(bad)} entries differ, max {d.max():.3g}; first:", [(tuple(i), float(a[tuple(i)]), float(b[tuple(i)])) for i in bad[:6]])
            if k == "measured_heights":
                for e, j in bad[:4]:
                    r = og("root_states")[e]
                    print("   env", e, "pt", j, "root xy", r[:2].tolist(), "quat", r[3:7].tolist())
print("done")
