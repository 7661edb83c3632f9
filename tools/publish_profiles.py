"""Copy the judged summaries of one tools/gpu_round6.sh (or an earlier round's script, tools/archive/) run (gpurun_out/<tag>/) into profiles/ under round-prefixed names.
usage: python tools/publish_profiles.py r02a r02"""
import csv
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, rnd = sys.argv[1], sys.argv[2]
O, P = os.path.join(ROOT, "gpurun_out", tag), os.path.join(ROOT, "profiles")
for src, dst in (("bench_default", "bench_default_train"), ("bench_driver_args", "bench_driver_args_steps20_warmup5"), ("bench_env", "bench_env_only"),
                 ("bench_aliengo_stairs", "bench_aliengo_stairs"), ("bench_aliengo_amp", "bench_aliengo_amp"), ("bench_go1", "bench_go1"),
                 ("bench_env_N262144", "bench_env_only_N262144"), ("bench_env_N64", "bench_env_only_N64"),
                 ("bench_env_zero_actions", "bench_env_only_zero_actions"), ("bench_2ranks_debug", "bench_2ranks_one_gpu_debug"),
                 ("bench_2ranks_mixed_debug", "bench_2ranks_mixed_robots_one_gpu_debug"), ("valu_peak", "valu_peak"),
                 ("bench_rccl_1rank", "bench_rccl_1rank_forced_collectives"), ("delassus_mfma", "delassus_mfma"),
                 ("bench_env_pgs", "bench_env_only_pgs_solver"), ("bench_default_pgs", "bench_default_train_pgs_solver"), ("bench_go2", "bench_go2"),
                 ("bench_env_aliengo_stairs", "bench_env_only_aliengo_stairs"), ("bench_env_aliengo_stairs_pgs", "bench_env_only_aliengo_stairs_pgs_solver"),
                 ("bench_rccl_1rank_4queues", "bench_rccl_1rank_forced_collectives_4_hw_queues"), ("bench_plain_again", "bench_default_train_repeat"),
                 ("bench_env_flat_priority", "bench_env_only_flat_wave_priority"), ("bench_env_aliengo_stairs_flat_priority", "bench_env_only_aliengo_stairs_flat_wave_priority"),
                 ("bench_env_r4_conventions", "bench_env_only_link_origin_velocities_no_limit_pass"), ("bench_driver_args_again", "bench_driver_args_steps20_warmup5_repeat"),
                 ("bench_aliengo_amp_round5_path", "bench_aliengo_amp_torch_rollout_step_and_update")):
    f = os.path.join(O, src + ".json")
    if os.path.exists(f) and open(f).read().lstrip().startswith("{"):
        shutil.copy(f, os.path.join(P, f"{rnd}_{dst}.json"))
    else:
        print("missing / invalid:", f)
if os.path.exists(os.path.join(O, "pmc_N4096.csv")):
    shutil.copy(os.path.join(O, "pmc_N4096.csv"), os.path.join(P, f"{rnd}_pmc_N4096.csv"))
    shutil.copy(os.path.join(O, "pmc_traffic.json"), os.path.join(P, "pmc_traffic.json"))
for src, dst in (("prof_env/env_kernel_stats.csv", "kernel_stats_env_only"), ("prof_train/train_kernel_stats.csv", "kernel_stats_train"),
                 ("prof_amp/amp_kernel_stats.csv", "kernel_stats_amp")):
    path = os.path.join(O, src)
    if not os.path.exists(path):
        print("missing:", path)
        continue
    rows = list(csv.reader(open(path)))
    with open(os.path.join(P, f"{rnd}_{dst}.csv"), "w", newline="") as fh:
        w = csv.writer(fh)
        for r in rows:
            r[0] = r[0][:160]     # torch's templated kernel names run to kilobytes
            w.writerow(r)
for name in ("pmc_stairs_N4096.csv", "phase_profile_aliengo.txt", "phase_profile_aliengo_stairs.txt", "trace_idle_rccl.txt", "free_running_parity.jsonl", "policy_time.txt", "amp_step_time.txt",
             "wave_times_aliengo.txt", "wave_times_aliengo_stairs.txt", "wave_times_aliengo_N256.txt", "wave_phases_aliengo.txt", "wave_phases_aliengo_N256.txt"):
    f = os.path.join(O, name)
    if os.path.exists(f):
        shutil.copy(f, os.path.join(P, f"{rnd}_{name}"))
for src, dst in (("prof_env_stairs/env_stairs_kernel_stats.csv", "kernel_stats_env_only_aliengo_stairs"),):
    path = os.path.join(O, src)
    if os.path.exists(path):
        shutil.copy(path, os.path.join(P, f"{rnd}_{dst}.csv"))
for name in ("gpu_tests.log",):
    f = os.path.join(O, name)
    if os.path.exists(f):
        open(os.path.join(P, f"{rnd}_{name}"), "w").write("".join(open(f).readlines()[-6:]))
print("published", tag, "->", P)
