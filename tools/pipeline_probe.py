"""What would pipelining two half-batches through the rollout buy?  A proxy that needs no new kernels: ONE runner
with N robots against TWO independent runners with N / 2 robots each, stepped alternately on two HIP streams -- the policy launch and
kernel B of one half can then run under the other half's kernel A.  Same total env-steps; only the collection loop is timed.
   python tools/pipeline_probe.py [N] [iterations]"""
import json
import sys
import time

import torch

sys.path.insert(0, ".")
from isaacgymloco_amd.envs import config as C  # noqa: E402
from isaacgymloco_amd.envs.legged_robot import LeggedRobot  # noqa: E402
from isaacgymloco_amd.learn.bench_train import train_cfg_dict  # noqa: E402
from isaacgymloco_amd.learn.runner import HIMOnPolicyRunner  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ITERS = int(sys.argv[2]) if len(sys.argv) > 2 else 6
dev = torch.device("cuda", 0)


def make(n, seed):
    cfg = C.TASKS["aliengo"][0]()
    cfg.env.num_envs = n
    env = LeggedRobot(cfg, sim_device="cuda:0", seed=seed, rank=0)
    torch.manual_seed(1)
    runner = HIMOnPolicyRunner(env, train_cfg_dict("aliengo"), log_dir=None, device="cuda:0")
    assert runner.enable_graphs()
    env.episode_length_buf = torch.randint_like(env.episode_length_buf, high=int(env.max_episode_length))
    return env, runner


def rollout(runners, streams, T):
    for _ in range(T):
        for r, s in zip(runners, streams):
            with torch.cuda.stream(s):
                r.graphs.step()


def end(runners, streams):
    for r, s in zip(runners, streams):
        with torch.cuda.stream(s):
            r.graphs.end_iteration()
            r.alg.storage.clear()


def timed(runners, streams):
    T = runners[0].num_steps_per_env
    out = []
    for it in range(ITERS + 2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rollout(runners, streams, T)
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        t = time.perf_counter() - t0
        end(runners, streams)
        if it >= 2:
            out.append((t, t_host))
    out.sort()
    return {"collection_ms_median": round(out[len(out) // 2][0] * 1e3, 3), "collection_ms_min": round(out[0][0] * 1e3, 3),
            "host_issue_ms_median": round(sorted(h for _, h in out)[len(out) // 2] * 1e3, 3)}


res = {"N": N, "steps_per_rollout": 100}
env1, r1 = make(N, 1)
res["one_runner"] = timed([r1], [torch.cuda.current_stream()])
print(res, flush=True)
del env1, r1
envA, rA = make(N // 2, 1)
envB, rB = make(N // 2, 2)
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
res["two_halves_one_stream"] = timed([rA, rB], [torch.cuda.current_stream(), torch.cuda.current_stream()])
print(res, flush=True)
res["two_halves_two_streams"] = timed([rA, rB], [sA, sB])
res["speedup_two_streams_vs_one_runner"] = round(res["one_runner"]["collection_ms_median"] / res["two_halves_two_streams"]["collection_ms_median"], 3)
print(json.dumps(res))
