"""bench.py's command line: the N > 1 launcher, the rank/world check, and that a train-mode line always contains whole PPO
iterations (update included) whatever --steps / --warmup the driver passes."""
import json
import os
import subprocess
import sys

import pytest

from helpers import ROOT

BENCH = os.path.join(ROOT, "bench.py")


def _run(argv, env=None, timeout=900):
    e = dict(os.environ)
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH] + argv, env=e, capture_output=True, text=True, timeout=timeout)


def test_world_size_mismatch_fails_loudly():
    """a rank started with the wrong world size must not report dp1 quietly; this exits before torch is imported"""
    r = _run(["--gpus", "2", "--no-cpu-baseline"], env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"}, timeout=120)
    assert r.returncode != 0
    assert "WORLD_SIZE=1" in r.stderr and "--gpus 2" in r.stderr
    assert r.stdout.strip() == ""


def test_launcher_relays_failure_of_ranks():
    """`bench.py --gpus 2` starts two rank processes itself; here (no GPU) both fail, and the parent must return non-zero
    without printing a JSON line of its own"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("CPU-container check")
    r = _run(["--gpus", "2", "--no-cpu-baseline", "--envs", "16", "--steps", "1", "--warmup", "1"], timeout=600)
    assert r.returncode != 0
    assert r.stdout.strip() == ""


@pytest.mark.gpu
def test_driver_arguments_time_a_full_ppo_iteration():
    """the driver's `--steps 20 --warmup 5`: one whole iteration (100 steps + GAE + update) must be inside the timed region"""
    r = _run(["--gpus", "1", "--steps", "20", "--warmup", "5", "--envs", "512", "--no-cpu-baseline"])
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads(r.stdout.strip().splitlines()[-1])
    assert j["n_gpus"] == 1 and j["steps"] == 20 and j["warmup"] == 5
    assert j["ppo_updates_timed"] >= 5 and j["timed_env_steps"] == 100 * j["ppo_updates_timed"]     # never a single-iteration sample
    lo, med, hi = j["iteration_wall_s_min_median_max"]
    assert 0 < lo <= med <= hi and len(j["collection_learn_s_by_iteration"]) == j["ppo_updates_timed"]
    assert j["learn_s_per_update"] > 0 and j["collection_s_per_iteration"] > 0 and j["ppo_iteration_wall_s"] > 0
    # value = N * T * iterations / (collection + learn), HIMR:179
    expect = 512 * j["timed_env_steps"] / (j["ppo_iteration_wall_s"] * j["ppo_updates_timed"])
    assert abs(j["value"] - expect) / expect < 0.02
    assert "update" in j["config"]["workload"] and j["config"]["mode"] == "train"
    assert j["roofline"]["kernel_avg_ms"] > 0


@pytest.mark.gpu
def test_the_line_explains_its_update_and_announces_a_rejected_gemm_table():
    """VERDICT r4 task 1: the train line says what the update ran on (streams, TunableOp entries and validators, hardware queues, a GEMM-rate
    probe, the shader clock) -- and a GEMM table from another build (rejected at load: LSIM_DEBUG_STALE_TUNE_TABLE) is announced on stderr
    and in the line instead of silently costing the update its second stream"""
    r = _run(["--gpus", "1", "--steps", "20", "--warmup", "5", "--no-cpu-baseline"], env={"LSIM_DEBUG_STALE_TUNE_TABLE": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    assert "REJECTED the GEMM table" in r.stderr and "PT_VERSION" in r.stderr and "ONE stream" in r.stderr
    j = json.loads(r.stdout.strip().splitlines()[-1])
    t = j["tunableop"]
    assert t["enabled"] and t["validators_match"] is False and "PT_VERSION" in t["validator_mismatches"]
    assert t["explicit_solutions_loaded"] == 0 and j["update_two_streams"] is False
    assert j["ppo_updates_timed"] >= 10 and j["iteration_spread_frac"] < 0.2
    assert j["gemm_probe_after_timed_region"]["tflops"] > 20 and "gpu_max_hw_queues" in j and j["linear_elu_forward"] == "aligned"
    s = j["sclk_during_timed_region"]
    assert s["mean_mhz"] is None or 300 < s["mean_mhz"] < 3000
    # the update is enqueued without a read-back: the host finishes enqueueing before the device finishes computing
    assert 0.0 < j["update_host_enqueue_s"] < j["learn_s_per_update"]


@pytest.mark.gpu
def test_two_ranks_on_the_fused_gpu_path_stay_in_lockstep():
    """N > 1 path on the fused GPU kernels: two rank processes (both on cuda:0, gloo -- RCCL refuses two ranks on one device) shard
    the envs, all-reduce the gradients and must end with bit-identical weights; rank 0 prints n_gpus = 2"""
    r = _run(["--gpus", "2", "--steps", "100", "--warmup", "100", "--envs", "256", "--no-cpu-baseline"],
             env={"LSIM_DEBUG_SINGLE_DEVICE": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["config"]["parallelism"] == "dp2"
    assert j["ppo_updates_timed"] >= 1
    assert len(j["weights_digest_by_rank"]) == 2 and j["ranks_in_lockstep"] is True
    assert abs(j["value"] - 2 * 256 * j["timed_env_steps"] / (j["ppo_iteration_wall_s"] * j["ppo_updates_timed"])) / j["value"] < 0.02
    # a multi-rank line explains itself (VERDICT r5 task 6): per-rank halves of the iteration, time blocked in the gradient collectives, rank skew
    pr = j["per_rank"]
    assert pr["collective_timing"] is True and all(len(pr[k]) == 2 for k in ("collection_s", "learn_s", "collective_blocked_s", "update_two_streams"))
    assert all(0.0 <= b <= l for b, l in zip(pr["collective_blocked_s"], pr["learn_s"])) and all(c > 0 for c in pr["collection_s"])
    assert len(set(pr["update_two_streams"])) == 1                     # the side-stream decision is taken jointly
    assert len(j["iteration_skew_s_max_mean"]) == 2 and 0.0 <= j["iteration_skew_s_max_mean"][1] <= j["iteration_skew_s_max_mean"][0]
    assert j["nonfinite_envs"] == 0


@pytest.mark.gpu
def test_mixed_robots_two_ranks_share_one_policy():
    """BASELINE config 5's mapping (`--mixed-robots`: the upper half of the ranks simulate the second robot, one asset per process as in the
    reference, LR:1135) on the fused GPU path: rank 0 on the Aliengo table, rank 1 REALLY on the Go2 table (total model mass as created),
    gradients all-reduced, bit-identical weights on both ranks.  Two ranks on one device, gloo (the 8-GPU RCCL run is the driver's)."""
    r = _run(["--gpus", "2", "--steps", "100", "--warmup", "100", "--envs", "256", "--no-cpu-baseline", "--mixed-robots"],
             env={"LSIM_DEBUG_SINGLE_DEVICE": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["config"]["mixed_robots"] is True and j["config"]["parallelism"] == "dp2"
    m0, m1 = j["robot_mass_kg_by_rank"]
    from isaacgymloco_amd.robots import aliengo, urdf
    want0 = sum(b.mass for b in aliengo.build_model().bodies)
    want1 = sum(b.mass for b in urdf.build_model_from_table("go2")[0].bodies)
    assert abs(m0 - want0) < 1e-2 and abs(m1 - want1) < 1e-2 and abs(want0 - want1) > 5.0, (m0, m1, want0, want1)
    assert len(j["weights_digest_by_rank"]) == 2 and j["ranks_in_lockstep"] is True
    assert j["ppo_updates_timed"] >= 5


@pytest.mark.gpu
@pytest.mark.parametrize("task", ["aliengo", "aliengo_amp"])
def test_rccl_collectives_of_the_multi_rank_path_run_on_one_gpu(task):
    """the RCCL calls of the N > 1 path (gradient buckets, KL mean, advantage / normaliser moments, parameter broadcast, barrier, timing
    max) on real hardware: a 1-rank `nccl` group with every collective issued (LSIM_DEBUG_FORCE_COLLECTIVES); the 2-rank test above can
    only use gloo because RCCL refuses two ranks on one device"""
    r = _run(["--gpus", "1", "--steps", "20", "--warmup", "5", "--envs", "256", "--task", task, "--no-cpu-baseline"],
             env={"LSIM_DEBUG_FORCE_COLLECTIVES": "1", "MASTER_PORT": "29541"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 1 and j["ppo_updates_timed"] >= 1 and j["ranks_in_lockstep"] is True
