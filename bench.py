#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the Aliengo hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--envs 4096] [--task aliengo] [--mode env|train]

One "step" = one LeggedRobot.step() over the whole batch of envs of a rank (4 physics sub-steps + post-physics +
reset + observations), driven the way the reference's runner drives it (HIMR:105-157).  Prints ONE JSON line from rank 0.
Multi-GPU: one process per GPU (torchrun), environments sharded with no data-path collective (weak scaling); in
train mode the PPO gradients are all-reduced over RCCL.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

ALGO_BYTES_PER_ENV_STEP = 6900.0   # 3.07 KB read + 3.79 KB written per env-step (fused design)
HBM_PEAK_GBS = 8000.0              # MI355X_MICROARCH.md: 8 TB/s spec


def cpu_baseline(task, sample_envs=64, budget_s=12.0):
    """Time the CPU oracle (build's scalar C twin of the same step; kind='port') on the host cores of this box: a child process
    (oracle/cpu_bench.py, no GPU, no torch) runs one oracle instance per core for a bounded sample and reports the aggregate."""
    import subprocess
    procs = min(len(os.sched_getaffinity(0)), 64)          # bounded: 64 workers x 64 envs is plenty to show the per-core rate
    out = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "cpu_bench.py"), "--task", task, "--envs", str(sample_envs),
                          "--seconds", str(budget_s), "--procs", str(procs)], capture_output=True, text=True, timeout=600)
    if out.returncode != 0:
        raise RuntimeError(out.stderr[-400:])
    return json.loads(out.stdout.strip().splitlines()[-1])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--envs", type=int, default=4096, help="envs per GPU")
    ap.add_argument("--task", default="aliengo")
    ap.add_argument("--mode", default="auto", choices=["auto", "env", "train"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--actions", default="normal", choices=["normal", "zeros"],
                    help="env mode action source (SURVEY.md 8d): N(0,1) = an untrained policy (init_noise_std 1), or zeros = standing robots")
    ap.add_argument("--mixed-robots", action="store_true",
                    help="BASELINE config 5: the upper half of the ranks simulate Go1 instead of --task's robot (one shared policy; not reference-comparable)")
    args = ap.parse_args()

    # fp32 GEMMs of the learner: use the hipBLASLt/rocBLAS solutions pre-selected by PyTorch TunableOp on gfx950
    # (isaacgymloco_amd/learn/tunableop_gfx950.csv; LSIM_TUNE=1 re-tunes and rewrites it).  Must be set before torch loads.
    tuned = os.path.join(ROOT, "isaacgymloco_amd", "learn", "tunableop_gfx950.csv")
    if os.environ.get("LSIM_TUNE") == "1" or os.path.exists(tuned):
        import shutil, tempfile
        lr_ = int(os.environ.get("LOCAL_RANK", "0"))
        tdir = tempfile.mkdtemp(prefix="lsim_tunableop_")
        if os.path.exists(tuned):
            shutil.copy(tuned, os.path.join(tdir, f"tuned{lr_}.csv"))   # TunableOp appends the device ordinal to the name
        os.environ.setdefault("PYTORCH_TUNABLEOP_ENABLED", "1")
        os.environ.setdefault("PYTORCH_TUNABLEOP_FILENAME", os.path.join(tdir, "tuned.csv"))
        os.environ.setdefault("PYTORCH_TUNABLEOP_TUNING", "1" if os.environ.get("LSIM_TUNE") == "1" else "0")
        os.environ.setdefault("PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS", "100")
        os.environ.setdefault("PYTORCH_TUNABLEOP_VERBOSE", "0")
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("LSIM_DEBUG_SINGLE_DEVICE") == "1":         # debugging aid: exercise the N > 1 code path on a 1-GPU box (all ranks on
        local_rank = 0                                            # cuda:0, gloo instead of RCCL); never set by the driver
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if os.environ.get("LSIM_DEBUG_SINGLE_DEVICE") == "1":
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=dev)   # RCCL; the rank's GPU is bound before the first collective

    from isaacgymloco_amd.envs import config as C
    from isaacgymloco_amd.envs.legged_robot import LeggedRobot
    mode = args.mode
    try:
        from isaacgymloco_amd.learn.runner import HIMOnPolicyRunner  # noqa: F401
        have_learner = True
    except Exception:
        have_learner = False
    if mode == "auto":
        mode = "train" if have_learner else "env"

    if args.mixed_robots and world > 1 and rank >= world // 2:
        args.task = "go1"           # same observation / action layout, different model table and gains (envs/config.py GO1_OVERRIDES)
    cfg = C.TASKS[args.task][0]()
    cfg.env.num_envs = args.envs
    env = LeggedRobot(cfg, sim_device=f"cuda:{local_rank}", seed=1, rank=rank, using_amp=(args.task == "aliengo_amp"))
    N, K, W = args.envs, args.steps, args.warmup

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    extra = {}
    if mode == "env":
        env.reset()
        # HIMR:90-91 init_at_random_ep_len=True: spread the resets
        env.episode_length_buf = torch.randint_like(env.episode_length_buf, high=int(env.max_episode_length))
        g = torch.Generator(device=dev).manual_seed(1 + rank)
        acts = [torch.randn(N, 12, device=dev, generator=g) for _ in range(16)]   # untrained policy: N(0,1) (init_noise_std=1, AGC:299)
        if args.actions == "zeros":
            acts = [torch.zeros(N, 12, device=dev)] * 16
        for i in range(W):
            env.step_device(acts[i % 16])
        env._L.lsim_set_profiling(env._h, K)
        barrier()
        t0 = time.perf_counter()
        for i in range(K):
            env.step_device(acts[i % 16])
        barrier()
        elapsed = time.perf_counter() - t0
        ms_a = (ctypes.c_float * K)()
        ms_b = (ctypes.c_float * K)()
        n = ctypes.c_int(K)
        env._L.lsim_read_profile(env._h, ms_a, ms_b, ctypes.byref(n))
        ka = sum(ms_a[i] for i in range(n.value)) / max(n.value, 1)
        kb = sum(ms_b[i] for i in range(n.value)) / max(n.value, 1)
        extra = {"kernel_a_ms": ka, "kernel_b_ms": kb}
        workload = (f"{args.task}: LeggedRobot.step() back-to-back, {'N(0,1)' if args.actions == 'normal' else 'zero'} actions, {N} envs/GPU "
                    "(no policy/learner in the loop)")
    else:
        from isaacgymloco_amd.learn.bench_train import run_train_bench
        elapsed, extra, workload = run_train_bench(env, cfg, args, dev, rank, world, barrier)
        ka = extra.get("kernel_a_ms", float("nan"))

    t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    if rank == 0:
        value = world * N * K / elapsed
        achieved = ALGO_BYTES_PER_ENV_STEP * N / (ka * 1e-3) / 1e9 if ka == ka and ka > 0 else None
        # HBM traffic per launch of kernel A comes from separate rocprofv3 --pmc passes of this same command (PMC counters cannot be
        # read in-process); tools/pmc_summary.py writes the corrected figure, valid for the workload it was collected on
        traffic, valu_frac, tpath = None, None, os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):
            tj = json.load(open(tpath))
            if tj.get("task") == args.task and tj.get("envs_per_gpu") == N:
                traffic = tj["traffic_bytes_per_launch"]
                if tj.get("valu_wave_insts_per_launch") and achieved:
                    # the kernel's real limiter: wave64 VALU instructions issue over 4 cycles on each of 1024 SIMDs (256 CUs x 4) at 2.4 GHz
                    valu_frac = tj["valu_wave_insts_per_launch"] * 4.0 / (1024 * 2.4e9 * ka * 1e-3)
        out = {
            "metric": "env-steps/sec (whole node)", "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": 1e3 * elapsed / K, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload, "task": args.task, "envs_per_gpu": N, "mode": mode, "parallelism": f"dp{world}"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": (achieved / HBM_PEAK_GBS) if achieved else None, "traffic": traffic,
                         "kernel": "lsim_k_step_a", "algorithmic_bytes_per_launch": ALGO_BYTES_PER_ENV_STEP * N,
                         "kernel_avg_ms": ka,
                         "valu_issue_frac": valu_frac,
                         "note": "VALU-issue bound, not HBM bound: 6.9 KB and ~18 k VALU wave instructions per env-step (DESIGN.md, kernel A); "
                                 "traffic and valu_issue_frac come from separate rocprofv3 --pmc passes (profiles/pmc_traffic.json)"},
        }
        # measured device-memory copy rate on this box (quote the datasheet peak AND a measurement): 1 GiB fp32 copy
        try:
            xs = torch.empty(1 << 28, device=dev); ys = torch.empty_like(xs)
            ys.copy_(xs); torch.cuda.synchronize(dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                ys.copy_(xs)
            e1.record(); torch.cuda.synchronize(dev)
            out["roofline"]["measured_copy_gbs"] = 5 * 2 * xs.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9
            del xs, ys
        except Exception:
            out["roofline"]["measured_copy_gbs"] = None
        out.update({k: v for k, v in extra.items() if k not in out})
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(args.task if args.task in C.TASKS else "aliengo")
            except Exception as e:  # the baseline is a reported extra, never the thing measured
                out["cpu_baseline"] = {"value": None, "error": str(e)}
        print(json.dumps(out))
    if os.environ.get("LSIM_TUNE") == "1" and rank == 0:   # keep the freshly tuned table (written at interpreter exit)
        import atexit, shutil
        src = os.environ["PYTORCH_TUNABLEOP_FILENAME"].replace(".csv", f"{local_rank}.csv")
        atexit.register(lambda: os.path.exists(src) and shutil.copy(src, os.path.join(ROOT, "gpurun_out", "tunableop_new.csv")))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
