"""Write profiles/INDEX.md: every file under profiles/ -> what it pins -> which section of DESIGN.md (current round) or HISTORY.md (rounds 1-5) cites it.
Descriptions come from the naming scheme of tools/publish_profiles.py and the one-off measurements of each round.  usage: python tools/profiles_index.py"""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
CUR = "r06"
RULES = [   # (regex on the name without its round prefix, what it is, section)
    (r"^bench_default_train(_repeat)?\.json$", "bench.py default line: aliengo, 4096 envs, HIMOnPolicyRunner loop (the driver's command)", "8"),
    (r"^bench_driver_command_run\d\.json$", "the driver's exact command (`--gpus 1 --steps 20 --warmup 5`, CPU baseline included) three times on one lease: run-to-run spread on a shared host", "8"),
    (r"^bench_driver_args.*\.json$", "the same with the driver's `--gpus 1 --steps 20 --warmup 5`", "8"),
    (r"^bench_default_train_pgs_solver\.json$", "default line with the PGS solver", "4"),
    (r"^bench_env_only\.json$", "env-only line: LeggedRobot.step() back to back, N(0,1) actions (kernel A + finish)", "6, 8"),
    (r"^bench_env_only_N\d+\.json$", "env-only line at another batch size (N262144: streaming regime)", "6"),
    (r"^bench_env_only_zero_actions\.json$", "env-only line, zero actions", "8"),
    (r"^bench_env_only_aliengo_stairs.*\.json$", "env-only line, aliengo_stairs (cfg3's physics)", "6, 8"),
    (r"^bench_env_only_.*\.json$", "env-only line under a stated switch (solver, priorities, conventions)", "4, 6"),
    (r"^bench_aliengo_amp_torch_rollout_step_and_update\.json$", "cfg4 with round 5's torch rollout step and update (LSIM_AMP_FUSED_STEP=0 LSIM_AMP_FUSED_UPDATE=0): the A/B of round 6", "7.2, 7.4, 10"),
    (r"^bench_aliengo_amp\.json$", "train line, aliengo_amp (cfg4)", "10"),
    (r"^bench_aliengo_stairs\.json$", "train line, aliengo_stairs (cfg3)", "8"),
    (r"^bench_(go1|go2|aliengo_recover)\.json$", "train line on another robot / task", "8"),
    (r"^bench_2ranks.*\.json$", "two rank processes on ONE GPU (gloo, debug mode; mixed: rank 1 on the Go2 table): the N > 1 code path and its per-rank keys", "9"),
    (r"^bench_rccl_1rank.*\.json$", "1-rank RCCL group with every collective issued (LSIM_DEBUG_FORCE_COLLECTIVES)", "9"),
    (r"^kernel_stats_env_only(_aliengo_stairs)?\.csv$", "rocprofv3 --kernel-trace --stats of the env-only command (kernel A's average duration)", "6"),
    (r"^kernel_stats_train\.csv$", "rocprofv3 kernel statistics of the default train command", "7.3"),
    (r"^kernel_stats_amp\.csv$", "rocprofv3 kernel statistics of the cfg4 train command (ATen share, GEMM share)", "7.4"),
    (r"^pmc(_stairs)?_N4096\.csv$", "PMC counters of kernel A folded from separate rocprofv3 --pmc passes (tools/pmc_summary.py)", "6"),
    (r"^policy_time\.txt$", "fused policy kernel alone (tools/policy_time.py)", "7.1"),
    (r"^amp_step_time\.txt$", "lsim_amp_step alone against the torch statements (tools/amp_step_time.py)", "7.2"),
    (r"^amp_update_kernel_sequence\.txt$", "kernel sequence of one cfg4 minibatch before the fusion (round 5 tree's path): what the elementwise passes were", "7.4"),
    (r"^trained_policy_physics_.*\.json$", "closed-loop replay of the trained 1000-iteration policy through the CPU oracle's variants (shipped / cap 32 / true shapes / PGS); `_calf6`: with round 6's calf points", "4"),
    (r"^policy_.*\.pt$", "the trained actor-critic + simulator curriculum state after 1000 iterations (tools/train_probe.py ... checkpoint.pt): input of tools/trained_policy_physics.py", "4"),
    (r"^train_curve_.*\.json$", "learning curve (tools/train_probe.py): per-iteration reward, episode length, terrain level, losses", "4"),
    (r"^wgrad_knockouts\.txt$", "the shipped weight-gradient kernels alone: product / no operand loads in the loop / no MFMAs (-DLS_WG_KNOCK builds): where the time of VERDICT r5 task 3 is", "7.3"),
    (r"^padded_shuffle_ab\.txt$", "default train line with 16-byte-aligned shuffled observation rows + padded-weight fused first layers against contiguous rows (no gain: opt-in)", "7.3"),
    (r"^cpu_scaling_probe\.txt$", "the GPU box's host: cgroup CPU grant and oracle/cpu_bench.py at 8 .. 256 threads", "8"),
    (r"^kernel_a_ifetch_pmc\.txt$", "kernel A: instruction-cache and fetch counters (93 KB of code against a 64 KB cache: hit rate 99.76 %, not the limiter)", "6"),
    (r"^amp_host_enqueue\.txt$", "host time to enqueue one update against its wall time, default and AMP, before / after the sample indices stopped draining the pipeline (learn/amp.py: _IndexUploader)", "10"),
    (r"^update_host_profile_.*\.txt$", "cProfile of the host side of one update (tools/update_host_profile.py): where the enqueue time goes", "10"),
    (r"^kernel_a_sched_flags\.txt$", "kernel A built with the compiler's other instruction-scheduling strategies (max-ilp, max-memory-clause, latency bias, iterative ilp): all within 0.6 % (negative result)", "6"),
    (r"^kernel_a_ab\.txt$", "kernel A, product against build variants, interleaved on one lease (tools/gpu_ab_kernel_a.sh)", "6"),
    (r"^free_running_parity\.jsonl$", "HIP vs fp64 oracle free-running: per-case flags agreement and error percentiles (tests/test_gpu_free_running.py)", "4"),
    (r"^gpu_tests\.log$", "tail of `pytest -m gpu` on the MI355X", "8"),
    (r"^valu_peak\.json$", "tools/micro/valu_peak: wave64 VALU issue rate vs waves per SIMD (basis of valu_issue_frac)", "6"),
    (r"^phase_profile_.*\.txt$", "per-phase shader-clock profile of kernels A / B (-DLS_PHASE_TIMING build)", "6"),
    (r"^wave_(times|phases)_.*\.txt$", "per-wave start / end / checkpoints of a kernel-A launch (-DLS_WAVE_TIMES build)", "6"),
    (r"^trace_idle.*\.txt$", "device idle inside the update from a kernel trace (tools/trace_idle.py)", "9"),
    (r"^delassus_mfma.*$", "tools/micro/delassus_mfma: Delassus build, FMA rows vs MFMA tiles", "6"),
    (r"^linear_elu_forward\.txt$", "lsim_linear_elu_forward per layer shape against BLAS + ELU, alone and in the loop", "7.3"),
    (r"^wgrad.*$", "weight-gradient kernels: per-shape times, counters, the bf16-split form", "7.3"),
    (r"^split_bf16.*$", "fp32 GEMM work on the bf16 matrix pipe: register-tile rates and accuracy", "7.3"),
    (r"^collective_overhead\.json$", "update time with a 1-rank RCCL group at 2 / 4 / 8 hardware queues", "9"),
    (r"^train_line.*$", "consecutive train lines on one lease (run-to-run stability)", "8"),
    (r"^pmc_env_only_N4096\.csv$", "PMC counters of kernel A, env-only workload (round 1's form)", "6"),
    (r"^elu_probe\.json$", "the update's ELU passes in isolation; MLP forward layer by layer vs depth-first row chunks (negative result)", "7.3"),
    (r"^gemm_tn_lds_prototype\.json$", "LDS-tiled weight-gradient prototype (tools/micro/gemm_tn.hip; not adopted)", "7.3"),
    (r"^pipeline_probe\.json$", "one runner of N robots vs two of N / 2 on two streams (negative result)", "7.3"),
    (r"^bench_default_train_wgrad_split_bf16\.json$", "default train line with the bf16-split weight gradient switched on", "7.3"),
    (r"^soak.*\.txt$", "tools/soak.py: long random-action runs on every task, finiteness and state bounds", "4"),
    (r"^box_probe\.json$", "is this lease's GPU normal: kernel A at 4096 / 4160 / 8192 robots", "6"),
    (r"^estimator_loss.*\.txt$", "the estimator's loss head alone and in the training loop (DPP-quad form)", "7.3"),
    (r"^gather_time.*\.txt$", "the storage shuffle alone (lsim_gather_rows vs advanced indexing)", "7.3"),
    (r"^gpu_tests_mid_round\.log$", "tail of `pytest -m gpu` in the middle of that round", "8"),
    (r"^hw_queues_world1\.json$", "GPU_MAX_HW_QUEUES 2 / 4 / 8 at one rank (equal)", "9"),
    (r"^kernel_regs_learn\.txt$", "register / scratch / LDS budget of every learner kernel from the compiler's metadata (tools/kernel_regs.py)", "7.3"),
    (r"^linear_elu_forward_pmc\.txt$", "MFMA-busy / LDS / wait counters of lsim_linear_elu_forward", "7.3"),
    (r"^update_graphs_ab\.json$", "the update replayed from captured HIP graphs vs eager, idle and loaded host (negative result)", "7.3"),
]
STANDALONE = {"pmc_traffic.json": ("the PMC record bench.py matches (task, envs, mode, actions, solver) before quoting `roofline.traffic` / `valu_issue_frac`; rewritten by every round's script", "6, 8"),
              "ref_cpu_partial.json": ("the reference's own torch post-physics stack and learner timed in the build container (no physics): SURVEY 8d legs B / C", "8"),
              "INDEX.md": ("this file", "")}


def describe(name):
    if name in STANDALONE:
        return ("—",) + STANDALONE[name] + ("DESIGN",)
    m = re.match(r"^(r\d\d)_(.*)$", name)
    if not m:
        return "—", "(unclassified)", "", "DESIGN"
    rnd, rest = m.groups()
    doc = "DESIGN" if rnd == CUR else "HISTORY"
    for pat, what, sec in RULES:
        if re.match(pat, rest):
            return rnd, what, sec, doc
    return rnd, "one-off measurement of that round (see the section that cites it)", "", doc


files = sorted(f for f in os.listdir(P))
# a published profile must be a measurement: round 5 published the error text of a tool that had loaded a stale diagnostics build
broken = [f for f in files if f.endswith((".txt", ".log", ".json", ".jsonl", ".csv")) and b"Traceback (most recent call last)" in open(os.path.join(P, f), "rb").read()
          and not f.endswith("gpu_tests.log")]
if broken:
    raise SystemExit("profiles/ holds error text instead of a measurement: " + ", ".join(broken))
lines = ["# profiles/ — index", "",
         f"Files of the current round ({CUR}) are cited by `DESIGN.md`; files of earlier rounds by `HISTORY.md` (same section numbers as listed: HISTORY keeps the",
         "numbering of the rounds 1-5 document: 4 physics, 6 kernel A, 7 measurement, 8 multi-GPU, 9 AMP, 11 open items).  Generated by `tools/profiles_index.py`.", "",
         "| file | round | what it pins | cited in |", "|---|---|---|---|"]
for f in files:
    rnd, what, sec, doc = describe(f)
    if doc == "HISTORY":      # the old document's numbering: measurement was 7, multi-GPU 8, AMP 9
        sec = {"8": "7", "9": "8", "10": "9", "7.1": "7", "7.2": "9", "7.3": "7.1, 7.2", "7.4": "9", "6, 8": "6, 7", "4, 6": "4, 6", "7.2, 7.4, 10": "9"}.get(sec, sec)
    lines.append(f"| `{f}` | {rnd} | {what} | {doc + ' §' + sec if sec else doc} |")
open(os.path.join(P, "INDEX.md"), "w").write("\n".join(lines) + "\n")
print(len(files), "files indexed;", sum(1 for f in files if describe(f)[1].startswith("one-off") or describe(f)[1].startswith("(uncl")), "unclassified")
