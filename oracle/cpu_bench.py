"""TEST / MEASUREMENT INFRASTRUCTURE -- times the CPU oracle (the build's scalar-C twin of LeggedRobot.step, fp64 physics) on the
host cores: BASELINE.md section 3 plan A, the "build CPU baseline" of SURVEY.md 8(d).  kind = "port": this is NOT the reference's
PhysX CPU path, which cannot run here.  Started by bench.py as a child process (never imported by the product); touches no GPU.

Legs (one oracle instance each, OpenMP over envs inside liborc.so, N(0,1) actions, seed 1):
    N = 64   on 1 thread     -- the scalar port (BASELINE config 1's size)
    N = 64   on all cores
    N = 4096 on all cores    -- the headline `value`: the size bench.py runs on the GPU

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


def leg(task, envs, threads, seconds, gomp):
    import numpy as np
    from helpers import C, make_oracle
    gomp.omp_set_num_threads(threads)
    cfg = C.TASKS[task][0]()
    orc, lc, model, ter = make_oracle(cfg, envs, seed=1)
    orc.reset_all()
    rs = np.random.RandomState(1)
    acts = [rs.normal(0, 1, (envs, 12)).astype(np.float32) for _ in range(8)]
    orc.step(acts[0])
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < seconds or n < 2:
        orc.step(acts[n % 8])
        n += 1
    dt = time.perf_counter() - t0
    orc.close()
    return {"envs": envs, "threads": threads, "steps": n, "seconds": round(dt, 3), "env_steps_per_s": envs * n / dt}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--task", default="aliengo")
    ap.add_argument("--seconds", type=float, default=24.0, help="total budget over the three legs")
    ap.add_argument("--threads", type=int, default=0, help="0 = all cores this process may run on")
    a = ap.parse_args()
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    threads = a.threads if a.threads > 0 else avail
    os.environ.setdefault("OMP_PROC_BIND", "false")
    gomp = ctypes.CDLL("libgomp.so.1")
    legs = [leg(a.task, 64, 1, a.seconds * 0.2, gomp), leg(a.task, 64, threads, a.seconds * 0.2, gomp),
            leg(a.task, 4096, threads, a.seconds * 0.6, gomp)]
    scaling = legs[2]["env_steps_per_s"] / max(legs[0]["env_steps_per_s"], 1e-9)
    print(json.dumps({
        "value": legs[2]["env_steps_per_s"], "unit": "env-steps/s", "cores": threads, "kind": "port",
        "parallel_speedup_over_one_core": scaling,
        "value_1core": legs[0]["env_steps_per_s"], "value_n64_all_cores": legs[1]["env_steps_per_s"],
        "nproc": os.cpu_count(), "cores_available": avail, "cpu_model": cpu_model(), "legs": legs,
        "sample": f"task {a.task}, N(0,1) actions: {legs[2]['steps']} steps x 4096 envs on {threads} OpenMP threads (value); "
                  f"{legs[1]['steps']} steps x 64 envs on {threads} threads; {legs[0]['steps']} steps x 64 envs on 1 thread (value_1core); "
                  f"CPU oracle = the build's scalar C restatement, fp64 physics -- not the reference's PhysX CPU path.  A WEAK baseline: {threads} "
                  f"threads give only {scaling:.1f} x one core (the per-env loops are OpenMP-parallel, reset_idx's cross-env part and the Python "
                  f"driver are serial; N = 64 on all threads is about one core), so a GPU / CPU ratio from this line says nothing about either"}))
