"""TEST / MEASUREMENT INFRASTRUCTURE -- times the CPU oracle (the build's scalar-C twin of LeggedRobot.step, fp64 physics) on the
host cores: BASELINE.md section 3 plan A, the "build CPU baseline" of SURVEY.md 8(d).  kind = "port": this is NOT the reference's
PhysX CPU path, which cannot run here.  Started by bench.py as a child process (never imported by the product); touches no GPU.

Legs (one oracle instance each; the timing loop is orc_run_steps in C -- actions, step, clock -- so no Python runs inside the timed region;
OpenMP over envs in contiguous blocks, threads pinned, every per-env buffer first touched by the thread that owns the block):
    N = 64   on 1 thread, N(0,1) actions     -- the scalar port (BASELINE config 1's size)
    N = 4096 on all cores, N(0,1) actions    -- the headline `value`: the size bench.py runs on the GPU
    N = 4096 on all cores, zero actions      -- standing robots
    N = 4096 on all cores, closed loop       -- a randomly initialised HIMActorCritic (torch.manual_seed(1)) evaluated in C, mean + N(0,1) noise
(the three action sources of SURVEY.md 8d)

    python oracle/cpu_bench.py --task aliengo --seconds 24   ->  one JSON line
"""
import argparse
import ctypes
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path[:] = [p for p in sys.path if os.path.abspath(p or ".") != HERE]   # `oracle` must resolve to the package, not oracle/oracle.py
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_quota():
    """CPUs' worth of time this process's cgroup grants (cgroup v2 cpu.max, v1 cfs quota), or None: a container may SHOW every hardware thread of the
    host in its affinity mask and still be throttled to a few CPUs of run time -- 256 busy-waiting OpenMP threads then spend the quota spinning
    (measured on the GPU box in round 6: 256 pinned, actively waiting threads ran at 0.12 x ONE core; rounds 1-5's "256 threads = 7 x" was the same cap)"""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / per
    except (OSError, ValueError):
        return None


class _Layer(ctypes.Structure):
    _fields_ = [("w", ctypes.c_void_p), ("b", ctypes.c_void_p), ("n_in", ctypes.c_int), ("n_out", ctypes.c_int), ("elu", ctypes.c_int)]


class _Policy(ctypes.Structure):
    _fields_ = [("enc", _Layer * 3), ("act", _Layer * 4)]


_POLICY_CHILD = r"""
import sys, numpy as np, torch, torch.nn as nn
sys.path.insert(0, sys.argv[1])
from isaacgymloco_amd.learn.bench_train import train_cfg_dict
from isaacgymloco_amd.learn.modules import HIMActorCritic
torch.manual_seed(1)
ac = HIMActorCritic(270, 238, 45, 12, **train_cfg_dict("aliengo")["policy"])
out = {}
for name, seq in (("enc", ac.estimator.encoder), ("act", ac.actor)):
    for i, l in enumerate([m for m in seq if isinstance(m, nn.Linear)]):
        out[f"{name}{i}_w"], out[f"{name}{i}_b"] = l.weight.detach().numpy(), l.bias.detach().numpy()
np.savez(sys.argv[2], **out)
"""


def random_policy():
    """HIMActorCritic as the runner creates it, torch.manual_seed(1): the C struct of its encoder / actor weights (+ the arrays that keep them alive).
    The weights are drawn in a CHILD process: importing torch here would bring a second OpenMP runtime (the wheel's own libgomp) into this process, the
    oracle's parallel regions would bind to that one and run with whatever thread count / places IT derived -- measured: one thread (256 = 1 x one core)."""
    import subprocess
    import tempfile
    import numpy as np
    path = os.path.join(tempfile.mkdtemp(prefix="lsim_cpu_bench_"), "policy.npz")
    env = {k: v for k, v in os.environ.items() if not k.startswith("OMP_")}
    subprocess.check_call([sys.executable, "-c", _POLICY_CHILD, ROOT, path], env=env, stdout=subprocess.DEVNULL, timeout=400)   # (a fresh box's first torch import: minutes)
    z = np.load(path)
    P, keep = _Policy(), []
    for dst, name, n in ((P.enc, "enc", 3), (P.act, "act", 4)):
        for i in range(n):
            w, b = np.ascontiguousarray(z[f"{name}{i}_w"], np.float32), np.ascontiguousarray(z[f"{name}{i}_b"], np.float32)
            keep += [w, b]
            dst[i].w, dst[i].b, dst[i].n_in, dst[i].n_out, dst[i].elu = w.ctypes.data, b.ctypes.data, w.shape[1], w.shape[0], int(i + 1 < n)
    return P, keep


def leg(task, envs, threads, seconds, gomp, source="normal", policy=None):
    import numpy as np
    from helpers import C, make_oracle
    gomp.omp_set_num_threads(threads)
    cfg = C.TASKS[task][0]()
    orc, lc, model, ter = make_oracle(cfg, envs, seed=1)      # (orc_create: parallel first touch by the pinned threads)
    L = orc._L
    L.orc_run_steps.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_double)]
    orc.reset_all()
    rs = np.random.RandomState(1)
    table = np.ascontiguousarray(rs.normal(0, 1, (8, envs, 12)).astype(np.float32))
    mode = {"zeros": 0, "normal": 1, "policy": 2}[source]
    pol = ctypes.byref(policy[0]) if mode == 2 else None
    sec = ctypes.c_double()

    def run(n):
        rc = L.orc_run_steps(orc._h, n, mode, table.ctypes.data_as(ctypes.c_void_p), 8, pol, ctypes.byref(sec))
        assert rc == 0, rc
        return sec.value
    used = L.orc_parallel_threads()                             # what a parallel region of the oracle really runs with
    assert used == threads, f"asked OpenMP for {threads} threads, the oracle's parallel regions run with {used} (a second OpenMP runtime in this process?)"
    run(2)                                                      # warm-up (page faults of the scratch, thread pool)
    probe = max(run(3) / 3, 1e-6)
    n = max(int(seconds / probe), 2)
    dt = run(n)
    bad = int(np.array(orc.buf["nonfinite"])[0])
    orc.close()
    return {"envs": envs, "threads": threads, "actions": source, "steps": n, "seconds": round(dt, 3), "env_steps_per_s": envs * n / dt, "nonfinite_env_steps": bad,
            "threads_in_parallel_region": used}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--task", default="aliengo")
    ap.add_argument("--seconds", type=float, default=24.0, help="total budget over the three legs")
    ap.add_argument("--threads", type=int, default=0, help="0 = all cores this process may run on")
    a = ap.parse_args()
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = cpu_quota()
    threads = a.threads if a.threads > 0 else (avail if quota is None else max(1, min(avail, int(quota))))
    # threads pinned (one place per hardware thread): with schedule(static) every thread then owns the same env block in every loop of every step, the
    # block's pages were first touched by it (orc_create) and stay in its caches.  Must be in the environment before libgomp initialises.
    os.environ.setdefault("OMP_PROC_BIND", "true")
    os.environ.setdefault("OMP_PLACES", "threads")
    os.environ.setdefault("OMP_WAIT_POLICY", "passive" if quota is not None else "active")     # under a CPU-time quota a spinning thread burns the others' time
    try:
        pol = random_policy()                # (a child process: no torch, i.e. no second OpenMP runtime, in this one)
    except Exception as e:                   # the closed-loop leg is one of three action sources: without it the baseline still stands
        pol = None
        print(f"cpu_bench: no random policy ({type(e).__name__}: {e}); the closed-loop leg is skipped", file=sys.stderr)
    gomp = ctypes.CDLL("libgomp.so.1")
    assert "torch" not in sys.modules
    legs = [leg(a.task, 64, 1, a.seconds * 0.15, gomp), leg(a.task, 4096, threads, a.seconds * 0.35, gomp),
            leg(a.task, 4096, threads, a.seconds * 0.2, gomp, "zeros")]
    legs.append(leg(a.task, 4096, threads, a.seconds * 0.3, gomp, "policy", pol) if pol is not None else
                {"envs": 4096, "threads": threads, "actions": "policy", "steps": 0, "seconds": 0.0, "env_steps_per_s": None, "skipped": "no policy weights"})
    scaling = legs[1]["env_steps_per_s"] / max(legs[0]["env_steps_per_s"], 1e-9)
    print(json.dumps({
        "value": legs[1]["env_steps_per_s"], "unit": "env-steps/s", "cores": threads, "kind": "port",
        "parallel_speedup_over_one_core": scaling,
        "value_1core": legs[0]["env_steps_per_s"], "value_zero_actions": legs[2]["env_steps_per_s"], "value_closed_loop_random_policy": legs[3]["env_steps_per_s"],
        "nproc": os.cpu_count(), "cores_available": avail, "cgroup_cpu_quota": quota, "cpu_model": cpu_model(), "legs": legs,
        "omp": {k: os.environ.get(k) for k in ("OMP_PROC_BIND", "OMP_PLACES", "OMP_WAIT_POLICY")},
        "sample": f"task {a.task}: {legs[1]['steps']} steps x 4096 envs, N(0,1) actions, on {threads} pinned OpenMP threads (value); {legs[2]['steps']} steps with zero "
                  f"actions; {legs[3]['steps']} steps closed loop with a randomly initialised HIMActorCritic evaluated in C; {legs[0]['steps']} steps x 64 envs on 1 thread "
                  f"(value_1core).  Timing loop in C (orc_run_steps), static env blocks, first-touch placement: {threads} threads = {scaling:.1f} x one core"
                  + (f" (the box shows {avail} hardware threads but its cgroup grants {quota:g} CPUs of run time: the thread count follows the grant)" if quota is not None else "") + ".  "
                  "CPU oracle = the build's scalar C restatement with fp64 physics -- not the reference's PhysX CPU path (closed binary, absent here)"}))
