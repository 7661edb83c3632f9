"""TEST / MEASUREMENT INFRASTRUCTURE -- times the CPU oracle (the build's scalar-C twin of LeggedRobot.step, fp64 physics) on the
host cores: `procs` independent worker processes, each stepping its own `envs`-robot oracle instance for `seconds`; the
aggregate is the "build CPU baseline" of SURVEY.md 8(d) (kind = "port": this is NOT the reference's PhysX CPU path, which
cannot run here).  Started by bench.py as a child process (never imported by the product); touches no GPU.

    python oracle/cpu_bench.py --task aliengo --envs 64 --seconds 10 --procs 16   ->  one JSON line
"""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path[:] = [p for p in sys.path if os.path.abspath(p or ".") != HERE]   # `oracle` must resolve to the package, not oracle/oracle.py
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _worker(task, envs, seconds, seed, start, q):
    try:
        _work(task, envs, seconds, seed, start, q)
    except BaseException as e:      # report instead of leaving the parent waiting
        q.put(("error", repr(e)))
        raise


def _work(task, envs, seconds, seed, start, q):
    import numpy as np
    from helpers import C, make_oracle
    cfg = C.TASKS[task][0]()
    orc, lc, model, ter = make_oracle(cfg, envs, seed=seed)
    orc.reset_all()
    acts = np.random.RandomState(seed).normal(0, 1, (envs, 12)).astype(np.float32)
    orc.step(acts)
    start.wait()
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < seconds:
        orc.step(acts)
        n += 1
    q.put((n, time.perf_counter() - t0))


def run(task, envs, seconds, procs):
    ctx = mp.get_context("fork")
    q, start = ctx.Queue(), ctx.Event()
    ps = [ctx.Process(target=_worker, args=(task, envs, seconds, 1 + i, start, q)) for i in range(procs)]
    for p in ps:
        p.start()
    time.sleep(0.5 + 0.02 * procs)      # let every worker finish building its terrain before the clock starts
    start.set()
    res = [q.get(timeout=seconds * 6 + 120) for _ in ps]
    for p in ps:
        p.join()
    bad = [r for r in res if r[0] == "error"]
    if bad:
        raise RuntimeError(bad[0][1])
    return sum(envs * n / dt for n, dt in res), sum(n for n, _ in res)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--task", default="aliengo")
    ap.add_argument("--envs", type=int, default=64)
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--procs", type=int, default=0, help="0 = all cores this process may run on")
    a = ap.parse_args()
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    procs = a.procs if a.procs > 0 else avail
    one, n1 = run(a.task, a.envs, a.seconds * 0.4, 1)
    allc, nall = (one, n1) if procs == 1 else run(a.task, a.envs, a.seconds, procs)
    print(json.dumps({"value": allc, "unit": "env-steps/s", "cores": procs, "kind": "port", "value_1core": one,
                      "sample": f"{nall} steps x {a.envs} envs of task {a.task} over {procs} worker processes (one oracle instance each, "
                                f"{avail} cores available, {os.cpu_count()} present), {a.seconds:.0f} s; CPU oracle = scalar C, fp64 physics"}))
