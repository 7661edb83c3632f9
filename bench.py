#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the Aliengo hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--envs 4096] [--task aliengo] [--mode train|env]

One "step" = one LeggedRobot.step() over the whole batch of envs of a rank (4 physics sub-steps + post-physics +
reset + observations), driven the way the reference's runner drives it (HIMR:105-157).  Prints ONE JSON line from rank 0.

--mode train (default) is the reference's `Perf/total_fps` (HIMR:179): value = N * T * iterations / (collection + learn).  The timed
region always consists of WHOLE PPO iterations -- max(5, ceil(K / T)) of them, T = num_steps_per_env = 100 -- each one = T x {policy
inference + step + storage} + compute_returns + update(); `steps` echoes the request, `timed_env_steps` says what was timed.
--mode env times step() back-to-back with pre-generated actions (exactly K steps), no learner.

Multi-GPU: one process per GPU, environments sharded with no data-path collective (weak scaling); in train mode the PPO
gradients are all-reduced over RCCL.  `python bench.py --gpus N` (no torchrun) starts the N rank processes itself, before
anything touches a GPU, and relays rank 0's line; under `python -m torch.distributed.run ... bench.py --gpus N` it is one of the ranks.
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES_PER_ENV_STEP = 6900.0   # 3.07 KB read + 3.79 KB written per env-step (fused design)
HBM_PEAK_GBS = 8000.0              # MI355X_MICROARCH.md: 8 TB/s spec
N_SIMD = 1024                      # 256 CUs x 4 SIMD-32
VALU_CYCLES_PER_WAVE_INST = 2.0    # MI355X_MICROARCH.md "Wave scheduling": a wave64 VALU instruction issues over 2 cycles on a SIMD-32: the floor.
                                   # tools/micro/valu_peak.hip measures 2.25-2.5 sustained (profiles/r03_valu_peak.json, wall-rate derived),
                                   # so frac priced at 2.0 is conservative
MAX_CLOCK_HZ = 2.4e9
WORLD1_HW_QUEUES = None            # GPU_MAX_HW_QUEUES of a single-rank run when the environment does not set it (None: the runtime's default, 4);
                                   # chosen by measurement, see DESIGN.md section 7


def cpu_baseline(task, budget_s=24.0):
    """Time the CPU oracle (the build's scalar-C twin of the same step, fp64 physics; kind='port') on the host cores of this box,
    BASELINE.md section 3 plan A: OpenMP over envs on the CPUs the cgroup grants at N = 4096 with three action sources, plus the 1-core figure at N = 64.  A child process
    (oracle/cpu_bench.py: no GPU, no torch), started before this process touches the GPU."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "cpu_bench.py"), "--task", task, "--seconds", str(budget_s)],
                         capture_output=True, text=True, timeout=900)
    if out.returncode != 0:
        raise RuntimeError(out.stderr[-400:])
    return json.loads(out.stdout.strip().splitlines()[-1])


def _parse_cpulist(text):
    out = []
    for part in text.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        out += list(range(int(a), int(b or a) + 1))
    return out


def host_cpu_plan(local_world, sysfs="/sys", allowed=None):
    """CPU set of every local rank of an N-rank run on one node -> ([cpus of rank 0, cpus of rank 1, ...], how it was derived).

    The PPO update needs 39-47 ms of host time per 71 ms (~2 200 launches, ~800 autograd nodes per update), so eight
    rank processes are eight Python loops each within 1.6 x of host-bound: where they run decides weak scaling before xGMI does (VERDICT r4
    task 6).  Every rank gets a DISJOINT set of hardware threads on the NUMA node its GPU hangs off -- its interpreter thread, autograd's
    device thread, RCCL's proxy and watchdog threads and the HIP runtime's signal thread stay next to the GPU's PCIe root and never migrate
    across sockets or on top of another rank -- and OMP_NUM_THREADS = 1 (torch's intra-op pool has nothing to do here; eight pools of 256
    threads would only fight for the same cores).
    GPU i's NUMA node: KFD topology node order (the HIP device order when no *_VISIBLE_DEVICES reorders it) -> PCI address -> numa_node.
    Anything unreadable: the allowed CPUs in NUMA-node order cut into local_world equal contiguous pieces (GPU i on socket i // (N / 2) is the
    usual 8-GPU board).  `allowed`: the CPUs this process may use (default: its current affinity mask)."""
    allowed = sorted(os.sched_getaffinity(0)) if allowed is None else sorted(allowed)
    allowed_set = set(allowed)
    nodes = {}
    try:
        base = os.path.join(sysfs, "devices", "system", "node")
        for d in sorted(os.listdir(base)):
            if d.startswith("node") and d[4:].isdigit():
                cpus = [c for c in _parse_cpulist(open(os.path.join(base, d, "cpulist")).read()) if c in allowed_set]
                if cpus:
                    nodes[int(d[4:])] = cpus
    except OSError:
        pass
    if not nodes:
        nodes = {0: allowed}
    ordered = [c for n in sorted(nodes) for c in nodes[n]]

    def even(cpus, parts):
        k, r = divmod(len(cpus), parts)
        out, at = [], 0
        for i in range(parts):
            n = k + (1 if i < r else 0)
            out.append(cpus[at:at + n] if n else list(cpus))       # fewer CPUs than ranks: share them all
            at += n
        return out
    gpu_node = []
    # KFD node order is the HIP device order only while nothing re-maps devices: under HIP_ / ROCR_ / CUDA_VISIBLE_DEVICES (or the single-device debug
    # mode, where every rank shares device 0) rank i's GPU is not the i-th KFD node, and pinning by it would put ranks next to the WRONG GPU's NUMA node
    # while the line still said "kfd topology" (ADVICE r5): fall back to the even split and say why
    remap = [v for v in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "LSIM_DEBUG_SINGLE_DEVICE") if os.environ.get(v)]
    if remap:
        return even(ordered, local_world), "even split of the allowed CPUs in NUMA-node order (" + ", ".join(remap) + " set: KFD order is not the device order)"
    try:
        top = os.path.join(sysfs, "class", "kfd", "kfd", "topology", "nodes")
        for d in sorted(os.listdir(top), key=int):
            props = dict(line.split()[:2] for line in open(os.path.join(top, d, "properties")) if len(line.split()) >= 2)
            if int(props.get("simd_count", "0")) <= 0:
                continue                                            # a CPU node
            loc, dom = int(props["location_id"]), int(props.get("domain", "0"))
            addr = f"{dom:04x}:{(loc >> 8) & 0xff:02x}:{(loc >> 3) & 0x1f:02x}.{loc & 7}"
            nn = int(open(os.path.join(sysfs, "bus", "pci", "devices", addr, "numa_node")).read())
            gpu_node.append(nn if nn in nodes else None)
    except (OSError, ValueError, KeyError):
        gpu_node = []
    if len(gpu_node) >= local_world and all(n is not None for n in gpu_node[:local_world]):
        plan = [None] * local_world
        for n in sorted(set(gpu_node[:local_world])):
            ranks = [r for r in range(local_world) if gpu_node[r] == n]
            for r, cpus in zip(ranks, even(nodes[n], len(ranks))):
                plan[r] = cpus
        return plan, "kfd topology -> pci numa_node"
    return even(ordered, local_world), "even split of the allowed CPUs in NUMA-node order"


def _cpu_ranges(cpus):
    """[0, 1, 2, 3, 8, 9] -> '0-3,8-9'"""
    out, cpus = [], sorted(cpus)
    i = 0
    while i < len(cpus):
        j = i
        while j + 1 < len(cpus) and cpus[j + 1] == cpus[j] + 1:
            j += 1
        out.append(str(cpus[i]) if i == j else f"{cpus[i]}-{cpus[j]}")
        i = j + 1
    return ",".join(out)


def pin_rank(local_rank, local_world):
    """bind this rank process to its CPU set and keep torch's CPU pools small -- BEFORE torch or HIP is loaded (threads created later inherit
    the mask).  LSIM_PIN_RANKS=0 leaves the process alone.  -> what the JSON line reports"""
    info = {"pinned": False, "omp_num_threads": os.environ.get("OMP_NUM_THREADS")}
    if os.environ.get("LSIM_PIN_RANKS", "1") == "0" or local_world <= 1:
        return info
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    os.environ.setdefault("MKL_NUM_THREADS", "1")
    info["omp_num_threads"] = os.environ["OMP_NUM_THREADS"]
    try:
        plan, how = host_cpu_plan(local_world)
        os.sched_setaffinity(0, plan[local_rank])
        info.update(pinned=True, source=how, cpu_affinity_by_local_rank=[_cpu_ranges(c) for c in plan])
    except (OSError, ValueError, IndexError) as e:
        info["error"] = f"{type(e).__name__}: {e}"
    return info


def task_of_rank(task, rank, world, mixed_robots):
    """BASELINE config 5's mapping (--mixed-robots): the upper half of the ranks simulate Go2 (same observation / action layout, its own model
    table and gains: envs/config.py GO2_OVERRIDES), the lower half --task's robot; one asset per process as in the reference (LR:1133-1135)"""
    return "go2" if (mixed_robots and world > 1 and rank >= world // 2) else task


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes (this parent never imports torch or touches a device),
    relay rank 0's JSON line, exit with the worst return code."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else sys.stderr))
    rc, pending = 0, set(range(n))
    while pending:
        for r in list(pending):
            c = procs[r].poll()
            if c is None:
                continue
            pending.discard(r)
            if c != 0:
                rc = rc or c
                for o in pending:            # a dead rank leaves the others waiting in a collective: end exactly the processes started here
                    procs[o].terminate()
        time.sleep(0.05)
    sys.exit(rc)


class ClockSampler:
    """mean shader clock of this rank's GPU over a stretch of the run, read from sysfs (pp_dpm_sclk: the level marked '*') by a background
    thread every 25 ms -- a file read, no driver call; None when the file is not there or not readable (the line then says so)"""

    def __init__(self, pci_address):
        """pci_address: '0000:xx:yy.z' of the GPU this rank computes on (a box shows every GPU of the node in sysfs, whichever one the
        process was given: card numbers say nothing)"""
        import glob
        self.path = None
        for p in glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"):
            if os.path.basename(os.path.realpath(os.path.dirname(p))).lower() == str(pci_address).lower():
                self.path = p
        self.samples, self.times, self._stop, self._thread = [], [], False, None
        self.error = None if self.path else f"no /sys/class/drm/card*/device -> {pci_address} with a pp_dpm_sclk"

    def _read(self):
        for line in open(self.path):
            if "*" in line:
                return float(line.split(":")[1].strip().split("Mhz")[0].split("MHz")[0])
        return None

    def start(self):
        if self.path is None:
            return self
        import threading
        try:
            self._read()
        except Exception as e:
            self.error = f"{type(e).__name__}: {e}"
            return self

        def loop():
            while not self._stop:
                try:
                    v = self._read()
                    if v:
                        self.samples.append(v)
                        self.times.append(time.perf_counter())
                except Exception:
                    pass
                time.sleep(0.025)
        self._thread = threading.Thread(target=loop, daemon=True)
        self._thread.start()
        return self

    def stop(self, phases=None):
        """phases: {name: [(t0, t1), ...]} in time.perf_counter() seconds -> also the mean clock over the samples that fall inside each phase
        (the rollout's short kernels and the update's GEMMs need not run at the same clock: one box of round 5 ran kernel A 1.7 x slower
        than every other while its update was the fastest)"""
        self._stop = True
        if self._thread is not None:
            self._thread.join(timeout=1.0)
        if not self.samples:
            return {"mean_mhz": None, "samples": 0, "source": self.path, "error": self.error or "no sample"}
        out = {"mean_mhz": sum(self.samples) / len(self.samples), "min_mhz": min(self.samples), "max_mhz": max(self.samples),
               "samples": len(self.samples), "source": self.path}
        for name, spans in (phases or {}).items():
            vals = [v for v, t in zip(self.samples, self.times) if any(a <= t < b for a, b in spans)]
            out[f"mean_mhz_{name}"] = sum(vals) / len(vals) if vals else None
            out[f"samples_{name}"] = len(vals)
        return out


def gemm_probe(torch, dev, seconds=0.02):
    """fp32 GEMM rate of this box right now: 4096^3 torch.mm (137 GFLOP each) for ~20 ms -- the same matrix pipe and library the update's
    GEMMs use; a box whose probe is low explains a slow update without any change of the tree"""
    try:
        a = torch.randn(4096, 4096, device=dev)
        b = torch.randn(4096, 4096, device=dev)
        for _ in range(3):
            torch.mm(a, b)
        torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 0
        t0 = time.perf_counter()
        e0.record()
        while True:
            for _ in range(4):
                torch.mm(a, b)
            n += 4
            if time.perf_counter() - t0 > seconds or n >= 64:
                break
        e1.record(); torch.cuda.synchronize(dev)
        ms = e0.elapsed_time(e1)
        return {"tflops": n * 2 * 4096 ** 3 / (ms * 1e-3) / 1e12, "shape": "4096x4096x4096 fp32 torch.mm", "launches": n, "ms": ms}
    except Exception as e:
        return {"tflops": None, "error": f"{type(e).__name__}: {e}"}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--envs", type=int, default=4096, help="envs per GPU")
    ap.add_argument("--task", default="aliengo")
    ap.add_argument("--mode", default="train", choices=["env", "train"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--actions", default="normal", choices=["normal", "zeros"],
                    help="env mode action source (SURVEY.md 8d): N(0,1) = an untrained policy (init_noise_std 1), or zeros = standing robots")
    ap.add_argument("--mixed-robots", action="store_true",
                    help="BASELINE config 5: the upper half of the ranks simulate Go2 instead of --task's robot (one shared policy; not reference-comparable)")
    return ap.parse_args()


def pmc_for(task, n_envs, mode, actions, solver="tgs"):
    """HBM traffic / instruction counts per launch of kernel A from separate rocprofv3 --pmc passes of this same command (PMC counters cannot
    be read in-process); tools/pmc_summary.py writes them.  Only valid for the workload they were collected on: task, size, mode and action
    source must all match, otherwise None."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(path):
        return None
    tj = json.load(open(path))
    for rec in (tj.get("runs") or [tj]):
        if (rec.get("task"), rec.get("envs_per_gpu"), rec.get("mode"), rec.get("actions"), rec.get("solver", "pgs")) == (task, n_envs, mode, actions, solver):
            return rec         # (records from before round 4 carry no solver: they were collected on the PGS kernel)
    return None


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args.gpus)             # never returns

    # The contract is ONE JSON line on stdout.  RCCL prints a five-line banner (version, host, library path) to the C stdout of every process that
    # opens a communicator, flushed at exit -- i.e. AFTER the JSON line.  Keep a private handle on the real stdout for the result and point file
    # descriptor 1 at stderr for everything else (libraries included).
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch one rank per GPU (or plain `python bench.py --gpus N`)")

    # host side of a multi-rank run: this rank's CPU set next to its GPU, small CPU thread pools -- before torch / HIP create their threads
    host = pin_rank(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", world)))

    # CPU baseline first: a child process on the host cores, while this process has not initialised the GPU yet (rank 0 at N = 1 only)
    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        try:
            cpu = cpu_baseline(args.task)
        except Exception as e:  # the baseline is a reported extra, never the thing measured
            cpu = {"value": None, "error": str(e)}

    # fp32 GEMMs of the learner: use the hipBLASLt/rocBLAS solutions pre-selected by PyTorch TunableOp on gfx950
    # (isaacgymloco_amd/learn/tunableop_gfx950.csv; LSIM_TUNE=1 re-tunes and rewrites it).  Must be set before torch loads.
    tuned = os.path.join(ROOT, "isaacgymloco_amd", "learn", "tunableop_gfx950.csv")
    if os.environ.get("LSIM_TUNE") == "1" or os.path.exists(tuned):
        import shutil, tempfile
        tdir = tempfile.mkdtemp(prefix="lsim_tunableop_")
        if os.path.exists(tuned) and os.environ.get("LSIM_TUNE_FRESH") != "1":          # LSIM_TUNE_FRESH=1: tune every shape again from scratch
            mine = os.path.join(tdir, f"tuned{local_rank}.csv")                # TunableOp appends the device ordinal to the name
            shutil.copy(tuned, mine)
            if os.environ.get("LSIM_DEBUG_STALE_TUNE_TABLE") == "1":           # test hook: a table from "another build" (rejected at load)
                text = open(mine).read().replace("Validator,PT_VERSION,", "Validator,PT_VERSION,0.")
                open(mine, "w").write(text)
            os.environ["LSIM_TUNABLEOP_TABLE"] = mine                          # the file this process really reads (bench_train reports on it)
        os.environ.setdefault("PYTORCH_TUNABLEOP_ENABLED", "1")
        os.environ.setdefault("PYTORCH_TUNABLEOP_FILENAME", os.path.join(tdir, "tuned.csv"))
        os.environ.setdefault("PYTORCH_TUNABLEOP_TUNING", "1" if os.environ.get("LSIM_TUNE") == "1" else "0")
        os.environ.setdefault("PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS", "100")
        os.environ.setdefault("PYTORCH_TUNABLEOP_VERBOSE", "0")
    # Data-parallel ranks: two hardware queues per process.  The ROCm runtime spreads HIP streams over GPU_MAX_HW_QUEUES (default 4) hardware
    # queues; with the default, torch's RCCL stream lands on a queue of its own and every hand-off compute stream -> RCCL stream -> compute
    # stream crosses hardware queues (an inter-queue barrier / signal round trip on each side of every collective).  Measured on one MI355X with
    # a 1-rank RCCL group and every collective issued (profiles/r04_collective_overhead.json): update 79.1 ms with 4 queues, 77.1 with 8,
    # 75.7 with 2 -- against 75.4 without collectives.  Must be set before the HIP runtime starts; an explicit setting wins.
    if world > 1 or (world == 1 and os.environ.get("LSIM_DEBUG_FORCE_COLLECTIVES") == "1"):
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "2")
    elif WORLD1_HW_QUEUES is not None:
        os.environ.setdefault("GPU_MAX_HW_QUEUES", WORLD1_HW_QUEUES)
    import torch
    import torch.distributed as dist
    single_dev = os.environ.get("LSIM_DEBUG_SINGLE_DEVICE") == "1"   # debugging aid: exercise the N > 1 code path on a 1-GPU box (all ranks on
    if single_dev:                                                   # cuda:0, gloo instead of RCCL); never set by the driver
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    forced = world == 1 and os.environ.get("LSIM_DEBUG_FORCE_COLLECTIVES") == "1"   # debugging aid: a 1-rank RCCL group, every collective issued
    if forced:
        for k, v in (("RANK", "0"), ("WORLD_SIZE", "1"), ("MASTER_PORT", "29533")):
            os.environ.setdefault(k, v)
    if world > 1 or forced:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if single_dev:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=dev)   # RCCL; the rank's GPU is bound before the first collective

    from isaacgymloco_amd.envs import config as C
    from isaacgymloco_amd.envs.legged_robot import LeggedRobot
    mode = args.mode
    base_task = args.task
    args.task = task_of_rank(base_task, rank, world, args.mixed_robots)     # (Go2: kinematics fitted to the reference's Go2 mocap clips, inertial
                                                                             # values nominal; rounds 1-3 used Go1 here)
    cfg = C.TASKS[args.task][0]()
    cfg.env.num_envs = args.envs
    env = LeggedRobot(cfg, sim_device=f"cuda:{local_rank}", seed=1, rank=rank, using_amp=(args.task == "aliengo_amp"))
    N, K, W = args.envs, args.steps, args.warmup

    def barrier():
        if world > 1 or forced:
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
        nf0 = int(env.nonfinite_envs)            # (a host read: outside the timed region)
        t0 = time.perf_counter()
        for i in range(K):
            env.step_device(acts[i % 16])
        barrier()
        elapsed = time.perf_counter() - t0
        nonfinite = int(env.nonfinite_envs) - nf0
        timed_steps = K
        ms_a = (ctypes.c_float * K)()
        ms_b = (ctypes.c_float * K)()
        n = ctypes.c_int(K)
        env._L.lsim_read_profile(env._h, ms_a, ms_b, ctypes.byref(n))
        ka = sum(ms_a[i] for i in range(n.value)) / max(n.value, 1)
        kb = sum(ms_b[i] for i in range(n.value)) / max(n.value, 1)
        extra = {"kernel_a_ms": ka, "kernel_b_ms": kb, "timed_env_steps": K, "nonfinite_envs": nonfinite}
        actions_src = args.actions
        workload = (f"{args.task}: LeggedRobot.step() back-to-back, {'N(0,1)' if args.actions == 'normal' else 'zero'} actions, {N} envs/GPU "
                    "(no policy/learner in the loop)")
    else:
        from isaacgymloco_amd.learn.bench_train import run_train_bench   # raises if the learner is broken: no silent change of workload
        pr = torch.cuda.get_device_properties(dev)
        try:
            pci = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
        except AttributeError:
            pci = "unknown"
        clocks = ClockSampler(pci)
        elapsed, extra, workload = run_train_bench(env, cfg, args, dev, rank, world, barrier, clocks=clocks)
        extra["gemm_probe_after_timed_region"] = gemm_probe(torch, dev)
        extra["gpu_max_hw_queues"] = os.environ.get("GPU_MAX_HW_QUEUES", "unset (runtime default 4)")
        extra["host"] = host
        extra["device"] = {"name": pr.name, "compute_units": pr.multi_processor_count, "pci": pci, "arch": getattr(pr, "gcnArchName", None)}
        ka = extra["kernel_a_ms"]
        timed_steps = extra["timed_env_steps"]
        actions_src = "policy"

    t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    nf = torch.tensor([float(extra.get("nonfinite_envs", 0))], device=dev, dtype=torch.float64)
    if world > 1 or forced:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(nf, op=dist.ReduceOp.SUM)
    elapsed = float(t.item())
    extra["nonfinite_envs"] = int(nf.item())
    if extra["nonfinite_envs"] != 0:
        # env-steps of the timed region in which a robot's simulated state held a NaN / infinity (LSIM_BUF_NONFINITE): the number that would be printed
        # is then not a measurement of the workload (non-finite robots also run slower: until round 6 this was only noticed as a slow kernel A)
        raise SystemExit(f"bench.py: {extra['nonfinite_envs']} env-steps of the timed region had a non-finite robot state (solver blow-up): no line printed")

    if rank == 0:
        value = world * N * timed_steps / elapsed
        achieved = ALGO_BYTES_PER_ENV_STEP * N / (ka * 1e-3) / 1e9 if ka == ka and ka > 0 else None
        pmc = pmc_for(args.task, N, mode, actions_src, "tgs" if int(env.lcfg.solver_type) == 1 else "pgs")
        traffic = pmc["traffic_bytes_per_launch"] if pmc else None
        valu_frac = None
        if pmc and pmc.get("valu_wave_insts_per_launch") and achieved:
            # share of the chip's VALU issue slots kernel A uses: wave64 VALU instructions x 2 cycles each / (1024 SIMDs x clock x duration).
            # The clock is the measured effective clock of the PMC pass (GRBM_GUI_ACTIVE / duration) when it was collected, else the 2.4 GHz maximum.
            clk = pmc.get("effective_clock_hz") or MAX_CLOCK_HZ
            valu_frac = pmc["valu_wave_insts_per_launch"] * VALU_CYCLES_PER_WAVE_INST / (N_SIMD * clk * ka * 1e-3)
        out = {
            "metric": "env-steps/sec (whole node)", "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": 1e3 * elapsed / timed_steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload, "task": base_task, "envs_per_gpu": N, "mode": mode, "parallelism": f"dp{world}",
                       "mixed_robots": bool(args.mixed_robots and world > 1)},
            "roofline": None,
        }
        # Roofline of the dominant kernel (kernel A).  The task's schema offers hbm | mfma; kernel A is neither: it moves 6.9 KB per env-step
        # (3 % of HBM) and has no matrix work worth MFMA (tools/micro/delassus_mfma).  Its real limiter is VALU issue /
        # dependent-instruction latency, so when wave-instruction counts from a PMC pass of THIS workload are available the line says
        # bound = "valu" with frac = share of the chip's VALU issue slots used; the HBM figures the schema asks for stay beside it.
        hbm = {"achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": (achieved / HBM_PEAK_GBS) if achieved else None}
        kernel = "lsim_k_step_a_tgs" if int(env.lcfg.solver_type) == 1 else "lsim_k_step_a_pgs"
        common = {"traffic": traffic, "kernel": kernel, "solver": "tgs" if int(env.lcfg.solver_type) == 1 else "pgs",
                  "algorithmic_bytes_per_launch": ALGO_BYTES_PER_ENV_STEP * N, "kernel_avg_ms": ka,
                  "hbm": hbm, "hbm_frac": hbm["frac"], "hbm_achieved_gbs": achieved,      # flat copies: parsers that drop nested objects keep these
                  "valu_issue_frac": valu_frac, "valu_cycles_per_wave_inst": VALU_CYCLES_PER_WAVE_INST,
                  "pmc_source": ({k: pmc.get(k) for k in ("task", "envs_per_gpu", "mode", "actions", "effective_clock_hz", "file")} if pmc else None)}
        # Two clocks can be quoted for kernel A: `effective_clock_hz` of the PMC pass = GRBM_GUI_ACTIVE / 8 XCDs / kernel duration, which counts
        # dispatch time outside the kernel and therefore EXCEEDS the part's 2.4 GHz maximum (2.54 GHz), and the shader clock sampled from sysfs during this
        # run's collection phase (2.35-2.39 GHz).  The sampled clock is the believed one; `frac` keeps the counter-derived clock (the lower, conservative
        # fraction, comparable with earlier rounds) and `valu_issue_frac_at_sampled_sclk` says what the sampled clock gives.
        sclk_mhz = ((extra.get("sclk_during_timed_region") or {}).get("mean_mhz_collection") or (extra.get("sclk_during_timed_region") or {}).get("mean_mhz"))
        if valu_frac is not None and sclk_mhz:
            common["valu_issue_frac_at_sampled_sclk"] = pmc["valu_wave_insts_per_launch"] * VALU_CYCLES_PER_WAVE_INST / (N_SIMD * sclk_mhz * 1e6 * ka * 1e-3)
            common["clock_note"] = ("frac uses the PMC pass's GRBM_GUI_ACTIVE-derived clock (includes dispatch time outside the kernel, hence above the 2.4 GHz maximum: "
                                    "conservative); the sysfs-sampled shader clock of this run is the believed one")
        if valu_frac is not None:
            clk = pmc.get("effective_clock_hz") or MAX_CLOCK_HZ
            peak_rate = N_SIMD * clk / VALU_CYCLES_PER_WAVE_INST / 1e9                       # G wave64 VALU instructions per second, whole chip
            out["roofline"] = dict(common, bound="valu", achieved=pmc["valu_wave_insts_per_launch"] / (ka * 1e-3) / 1e9, peak=peak_rate,
                                   unit="G wave-inst/s", frac=valu_frac,
                                   note="bound = VALU issue (1024 SIMDs x clock / 2 cycles per wave64 instruction); `hbm` holds the schema's HBM figures: "
                                        "algorithmic bytes / kernel time against 8 TB/s; `traffic` = HBM bytes per launch from the PMC passes of "
                                        "profiles/pmc_traffic.json (same task / size / mode / action source)")
        else:
            out["roofline"] = dict(common, bound="hbm", achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s", frac=hbm["frac"],
                                   note="no PMC pass matches this workload (task / size / mode / actions): only the HBM figures of the schema; the "
                                        "kernel's real limiter is VALU issue, not HBM (DESIGN.md section 6)")
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
        if cpu is not None:
            out["cpu_baseline"] = cpu
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    if os.environ.get("LSIM_TUNE") == "1" and rank == 0:   # keep the freshly tuned table (written at interpreter exit)
        import atexit, shutil
        src = os.environ["PYTORCH_TUNABLEOP_FILENAME"].replace(".csv", f"{local_rank}.csv")
        atexit.register(lambda: os.path.exists(src) and shutil.copy(src, os.path.join(ROOT, "gpurun_out", "tunableop_new.csv")))
    if world > 1 or forced:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
