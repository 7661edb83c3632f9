"""Free-running dynamics parity, HIP (fp32, structured wave solver) against the CPU oracle (fp64, dense) WITHOUT re-synchronisation
(VERDICT r4 task 3; LR:146-152 is the loop under test: 4 x {torques, simulate, refresh} + post-physics, 60 times in a row).

The two simulations start from the same reset (the draws are keyed by (seed, env, step): identical) and are then left alone.  Contact
dynamics amplify rounding: a foot that touches one sub-step earlier in fp32 than in fp64 is a different trajectory from then on, so the
bar is statistical and stated per step -- median and 99th percentile over the robots of the root-position and joint-position error, and
the share of env-steps whose termination flag agrees.  A robot whose flag differed once has a different episode from then on (one side
reset): it leaves the state statistics and keeps counting against the agreement bar.

  (i)   task `aliengo` on its own terrain mix (AGC:89: smooth slope 0.3, rough slope 0.3, stairs up 0.2, stairs down 0.2), N = 256
  (ii)  task `aliengo_stairs` with the robots dropped all over the staircases (risers in reach of every leg), N = 256
  (iii) one step from the same state at N = 4096 for each BASELINE task (flat-start `aliengo`, `aliengo_stairs`, `aliengo_amp`)
each for both solvers.  LSIM_PARITY_REPORT=path appends the measured tables as JSON lines (profiles/r05_free_running_parity.jsonl)."""
import json
import os

import numpy as np
import pytest

from helpers import C, make_oracle

pytestmark = pytest.mark.gpu

SOLVERS = {"tgs": 1, "pgs": 0}
FEET = [4, 8, 12, 16]


def _report(rec):
    path = os.environ.get("LSIM_PARITY_REPORT")
    if path:
        with open(path, "a") as f:
            f.write(json.dumps(rec) + "\n")


def _free_run(cfg, N, steps, seed, action_scale=1.0, using_amp=False):
    """-> per-step table: [step, in-sync robots, median / p99 root position error (m), median / p99 joint position error (rad),
    p99 root linear-velocity error (m/s), flags agreeing this step]"""
    from hip_backend import HipBackend
    orc, lc, model, ter = make_oracle(cfg, N, seed=seed, using_amp=using_amp)
    be = HipBackend(cfg, N, ter, seed=seed, using_amp=using_amp)
    orc.reset_all(); be.reset_all()
    np.testing.assert_allclose(be.get("root_states"), orc.buf["root_states"], rtol=2e-6, atol=2e-7)       # the same start (a few fp32 ulps: FMA)
    np.testing.assert_allclose(be.get("dof_state"), orc.buf["dof_state"], rtol=2e-6, atol=2e-7)
    rs = np.random.RandomState(seed)
    sync = np.ones(N, bool)
    rows, agree, total, resets = [], 0, 0, 0
    # WHY a robot forks ("the cause of each fork is not recorded; a real divergence hides inside chaos"): the first step at which a robot's joint
    # error leaves the rounding floor (> 1e-3 rad; the median robot stays at 1e-5) is classified by what differs between the two simulations AT that step
    forked = np.zeros(N, bool)
    forks = {"contact_set_differs": 0, "same_contact_set_cap_binding": 0, "same_contact_set_jump": 0, "same_contact_set_gradual": 0, "first_steps": []}
    q_prev = np.zeros(N)
    for t in range(steps):
        a = (action_scale * rs.normal(0, 1, (N, 12))).astype(np.float32)
        orc.step(a); be.step(a)
        same = be.get("reset") == orc.buf["reset"]
        q_err = np.abs(be.get("dof_state").reshape(N, 12, 2)[:, :, 0] - orc.buf["dof_state"].reshape(N, 12, 2)[:, :, 0]).max(1)
        new = (q_err > 1e-3) & ~forked & sync & same
        if new.any():
            pat_h = np.linalg.norm(be.get("contact_forces"), axis=-1) > 0.1          # which bodies carry a contact force (last sub-step)
            pat_o = np.linalg.norm(orc.buf["contact_forces"], axis=-1) > 0.1
            cnt_h, cnt_o = be.get("contact_count"), orc.buf["contact_count"]            # collision points in contact before the cap: [max over sub-steps, last]
            for e in np.nonzero(new)[0]:
                if (pat_h[e] != pat_o[e]).any() or (cnt_h[e] != cnt_o[e]).any():
                    forks["contact_set_differs"] += 1                                   # a point touched / left in one arithmetic and not the other
                elif cnt_o[e, 0] > 8:
                    forks["same_contact_set_cap_binding"] += 1                          # equal sets, more candidates than the cap keeps
                elif q_err[e] > 20.0 * max(q_prev[e], 1e-6):
                    forks["same_contact_set_jump"] += 1                                 # equal sets, the error grew > 20 x in this one step: a branch point of the
                                                                                        # solver (friction box, limit-row selection, a contact inside a sub-step)
                else:
                    forks["same_contact_set_gradual"] += 1                              # equal sets, the error had been growing for steps: amplification of rounding
                forks["first_steps"].append(int(t))
            forked |= new
        q_prev = q_err
        agree += int(same.sum()); total += N
        resets += int(orc.buf["reset"].sum())
        sync &= same
        r_h, r_o = be.get("root_states"), orc.buf["root_states"]
        q_h, q_o = be.get("dof_state").reshape(N, 12, 2), orc.buf["dof_state"].reshape(N, 12, 2)
        assert np.isfinite(r_h).all() and np.isfinite(q_h).all(), f"step {t}"
        e_pos = np.linalg.norm(r_h[:, :3] - r_o[:, :3], axis=1)[sync]
        e_vel = np.linalg.norm(r_h[:, 7:10] - r_o[:, 7:10], axis=1)[sync]
        e_q = np.abs(q_h[:, :, 0] - q_o[:, :, 0]).max(1)[sync]
        pc = lambda v, p: float(np.percentile(v, p)) if v.size else 0.0
        rows.append([t, int(sync.sum()), pc(e_pos, 50), pc(e_pos, 99), pc(e_q, 50), pc(e_q, 99), pc(e_vel, 99), int(same.sum())])
    contacts = float((np.abs(orc.buf["contact_forces"][:, FEET, 2]) > 1.0).mean())
    _free_run.last_forks = forks
    return rows, agree / total, contacts, resets


# bars: [step index] -> (median root m, p99 root m, median joint rad, p99 joint rad), ~4 x the worst of the four measured runs below
STEPS = 60
# measured on MI355X (gpurun r5f, profiles/r05_free_running_parity.jsonl; [median root, p99 root, median joint, p99 joint]):
#   step 24: mix tgs 8.0e-7 / 2.5e-4 / 6.4e-6 / 4.1e-3, pgs 4.1e-7 / 1.2e-4 / 3.5e-6 / 1.3e-3; stairs tgs 9.6e-7 / 3.8e-4 / 9.5e-6 / 3.6e-3, pgs 7.7e-7 / 1.4e-3 / 6.3e-6 / 2.1e-2
#   step 59: mix tgs 9.8e-7 / 2.6e-2 / 7.4e-6 / 6.8e-2, pgs 7.6e-7 / 1.2e-2 / 5.3e-6 / 2.9e-2; stairs tgs 4.2e-6 / 2.0e-2 / 2.7e-5 / 0.16,   pgs 1.3e-6 / 2.7e-2 / 8.3e-6 / 0.18
# the median robot stays at fp32 resolution for all 60 steps; the 99th percentile is the handful of robots whose contact sequence forked
BARS = {
    "mix": {0: (1e-5, 1e-4, 1e-4, 2e-3), 24: (5e-6, 5e-3, 5e-5, 0.08), 59: (2e-5, 0.1, 1e-4, 0.6)},
    "stairs": {0: (1e-5, 1e-4, 1e-4, 2e-3), 24: (5e-6, 5e-3, 5e-5, 0.08), 59: (2e-5, 0.1, 1e-4, 0.6)},
}


@pytest.mark.parametrize("solver", ["tgs", "pgs"])
def test_free_running_on_the_tasks_own_terrain_mix(solver):
    """(i) 60 untouched steps (1.2 s) of task `aliengo` as configured -- slopes, rough slopes, stairs up and down (AGC:89), every domain
    randomisation and the action delay on -- 256 robots, N(0, 1) actions (an untrained policy, AGC:299)"""
    cfg = C.TASKS["aliengo"][0]()
    cfg.sim.physx.solver_type = SOLVERS[solver]
    rows, agree, contacts, resets = _free_run(cfg, 256, STEPS, seed=21)
    _report({"case": "aliengo terrain mix", "solver": solver, "N": 256, "flag_agreement": agree, "foot_contact_share": contacts, "resets": resets,
             "forks_joint_error_above_1e-3_rad": _free_run.last_forks,
             "columns": ["step", "in_sync", "root_med_m", "root_p99_m", "joint_med_rad", "joint_p99_rad", "root_vel_p99", "flags_same"], "rows": rows})
    print(f"{solver} mix: agreement {agree:.4f}, resets {resets}, in sync at the end {rows[-1][1]}, rows 24 / 59 {rows[24]} {rows[-1]}, forks {_free_run.last_forks}")
    assert agree >= 0.995                   # measured 0.9995-0.9997 (VERDICT r4 asked for >= 0.98)
    assert rows[-1][1] >= 238               # robots whose termination history never differed (measured 252 / 253 of 256)
    assert resets >= 50, "robots must fall and reset inside the window for the flags to mean anything (measured 152-156)"
    assert contacts > 0.2, "the robots must be on the ground for this to be a contact test"
    for t, (rm, rp, qm, qp) in BARS["mix"].items():
        assert rows[t][2] <= rm and rows[t][3] <= rp and rows[t][4] <= qm and rows[t][5] <= qp, (t, rows[t])


@pytest.mark.parametrize("solver", ["tgs", "pgs"])
def test_free_running_on_staircases_with_risers(solver):
    """(ii) 60 untouched steps of task `aliengo_stairs` with the robots dropped up to 3 m from their sub-terrain's centre (on the treads and
    against the risers of stairs up / down, 60 % of this task's terrain, ALIENGO_STAIRS_OVERRIDES)"""
    cfg = C.TASKS["aliengo_stairs"][0]()
    cfg.sim.physx.solver_type = SOLVERS[solver]
    cfg.domain_rand.base_init_pos_range = dict(x=[-3.0, 3.0], y=[-3.0, 3.0], z=[0.0, 0.3])
    rows, agree, contacts, resets = _free_run(cfg, 256, STEPS, seed=22)
    _report({"case": "aliengo_stairs, robots dropped over the staircases", "solver": solver, "N": 256, "flag_agreement": agree,
             "foot_contact_share": contacts, "resets": resets, "forks_joint_error_above_1e-3_rad": _free_run.last_forks,
             "columns": ["step", "in_sync", "root_med_m", "root_p99_m", "joint_med_rad", "joint_p99_rad", "root_vel_p99", "flags_same"], "rows": rows})
    print(f"{solver} stairs: agreement {agree:.4f}, resets {resets}, in sync at the end {rows[-1][1]}, rows 24 / 59 {rows[24]} {rows[-1]}, forks {_free_run.last_forks}")
    assert agree >= 0.995                   # measured 0.9990-0.9992
    assert rows[-1][1] >= 230               # measured 246 / 250 of 256
    assert resets >= 50                     # measured 199
    assert contacts > 0.2
    for t, (rm, rp, qm, qp) in BARS["stairs"].items():
        assert rows[t][2] <= rm and rows[t][3] <= rp and rows[t][4] <= qm and rows[t][5] <= qp, (t, rows[t])


@pytest.mark.parametrize("solver", ["tgs", "pgs"])
@pytest.mark.parametrize("task", ["aliengo", "aliengo_stairs", "aliengo_amp"])
def test_ten_steps_at_baseline_size_match_oracle(task, solver):
    """(iii) N = 4096, the three single-GPU BASELINE configurations: the ten steps after reset_all, untouched, every buffer the learner
    sees.  Step 0 starts from identical states with the robots still above the ground (the fp32-vs-fp64 distance of one contact-free step);
    by step 9 they have landed (foot contact share asserted) and the comparison is of ten steps of contact dynamics in a row."""
    from hip_backend import HipBackend
    N = 4096
    amp = task == "aliengo_amp"
    cfg = C.TASKS[task][0]()
    cfg.sim.physx.solver_type = SOLVERS[solver]
    orc, lc, model, ter = make_oracle(cfg, N, seed=31, using_amp=amp)
    be = HipBackend(cfg, N, ter, seed=31, using_amp=amp)
    orc.reset_all(); be.reset_all()
    rs = np.random.RandomState(31)
    out = []
    sync = np.ones(N, bool)
    for t in range(10):
        a = rs.normal(0, 1, (N, 12)).astype(np.float32)
        orc.step(a); be.step(a)
        same = be.get("reset") == orc.buf["reset"]
        sync &= same
        rec = {"step": t, "flags_same": int(same.sum()), "in_sync": int(sync.sum()),
               "foot_contact_share": float((np.abs(orc.buf["contact_forces"][:, FEET, 2]) > 1.0).mean())}
        for k in ("root_states", "dof_state", "contact_forces", "rew", "obs", "priv_obs"):
            d = np.abs(be.get(k).astype(np.float64) - orc.buf[k]).reshape(N, -1).max(1)[sync]
            rec[k] = [float(np.median(d)), float(np.percentile(d, 99)), float(d.max())]
        out.append(rec)
    _report({"case": f"{task}: ten steps after reset_all", "solver": solver, "N": N,
             "columns": "buffer: [median, p99, max] over the in-sync robots of the per-robot max abs error", "rows": out})
    first, last = out[0], out[-1]
    print(task, solver, "step 0:", {k: first[k][:2] for k in ("root_states", "dof_state", "obs", "rew")}, "step 9:",
          {k: last[k][:2] for k in ("root_states", "dof_state", "obs", "rew", "contact_forces")}, last["foot_contact_share"], last["in_sync"])
    # measured (gpurun r5f, step 0): root 4.8e-7 / 2.9e-6, joints 1.2e-5 / 1.0e-4, observations 6.1e-7 / 5.0e-6, rewards 1.1e-8 / 6.3e-8 (median / p99)
    assert first["flags_same"] == N
    assert first["root_states"][0] < 3e-6 and first["root_states"][1] < 2e-5
    assert first["dof_state"][0] < 6e-5 and first["dof_state"][1] < 5e-4
    assert first["obs"][0] < 5e-6 and first["obs"][1] < 5e-5 and first["rew"][0] < 1e-7 and first["rew"][1] < 1e-6
    # step 9: on the ground (bars ~4 x measured, see profiles/r05_free_running_parity.jsonl)
    assert last["foot_contact_share"] > 0.2
    assert last["in_sync"] >= 0.998 * N and all(r["flags_same"] >= 0.999 * N for r in out)
    assert last["root_states"][0] < 2e-5 and last["root_states"][1] < 5e-3
    assert last["dof_state"][0] < 4e-4 and last["dof_state"][1] < 0.1
    assert last["obs"][0] < 2e-4 and last["rew"][0] < 1e-5
