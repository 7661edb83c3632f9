"""N>1 path on CPU: world_size-2 gloo processes must take identical optimiser steps (gradient all-reduce, synchronised
KL / advantage statistics), and match a single process that sees the concatenated batch for the gradient average."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import ROOT


class FakeEnv:
    """Deterministic stand-in for the simulator (CPU): per-rank different data, same shapes as the real env."""
    num_obs, num_privileged_obs, num_one_step_obs, num_actions, max_episode_length, dt = 270, 238, 45, 12, 1000.0, 0.02

    def __init__(self, num_envs, seed):
        self.num_envs = num_envs
        self.g = torch.Generator().manual_seed(seed)
        self.episode_length_buf = torch.zeros(num_envs, dtype=torch.long)
        self.extras = {}

    def _r(self, *s):
        return torch.randn(*s, generator=self.g)

    def reset(self):
        return self.get_observations(), self.get_privileged_observations()

    def get_observations(self):
        return self._r(self.num_envs, 270)

    def get_privileged_observations(self):
        return self._r(self.num_envs, 238)

    def step(self, a):
        d = torch.rand(self.num_envs, generator=self.g) < 0.1
        ids = d.nonzero().flatten()
        self.extras["time_outs"] = torch.zeros(self.num_envs, dtype=torch.bool)
        return self._r(self.num_envs, 270), self._r(self.num_envs, 238), self._r(self.num_envs), d, self.extras, ids, self._r(len(ids), 238)


def _train_cfg():
    from isaacgymloco_amd.learn.bench_train import train_cfg_dict
    tc = train_cfg_dict("aliengo")
    tc["runner"]["num_steps_per_env"] = 6
    tc["algorithm"]["num_learning_epochs"] = 2
    tc["algorithm"]["num_mini_batches"] = 2
    return tc


def _worker(rank, world, port, out_dir, same_data=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from isaacgymloco_amd.learn.runner import HIMOnPolicyRunner
    r = 0 if same_data else rank
    torch.manual_seed(100 + r)             # different initial weights per rank: the broadcast must fix that
    runner = HIMOnPolicyRunner(FakeEnv(8, seed=7 + r), _train_cfg(), log_dir=None, device="cpu")
    torch.manual_seed(5 + r)
    c0 = runner.dist_ctx.collectives
    runner.learn(2, init_at_random_ep_len=False)
    sd = {k: v.clone() for k, v in runner.alg.actor_critic.state_dict().items()}
    torch.save({"sd": sd, "lr": runner.alg.learning_rate, "collectives": runner.dist_ctx.collectives - c0}, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.destroy_process_group()


def test_two_ranks_stay_in_lockstep(tmp_path):
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a = torch.load(os.path.join(tmp_path, "rank0.pt"))
    b = torch.load(os.path.join(tmp_path, "rank1.pt"))
    assert a["lr"] == b["lr"]
    for k in a["sd"]:
        torch.testing.assert_close(a["sd"][k], b["sd"][k], rtol=0, atol=0, msg=k)
    # one collective per minibatch (every gradient of both optimisers + the KL estimate in one bucket) + the advantage statistics per iteration
    # (21 per iteration with the reference's 5 epochs x 4 minibatches; here 2 x 2 minibatches, 2 iterations)
    assert a["collectives"] == b["collectives"] == 2 * (2 * 2 + 1)


def test_data_parallel_order_equals_the_single_rank_order(tmp_path):
    """the N > 1 path moves the estimator's step behind the PPO backward (one all-reduce then carries every gradient); with the SAME data on both
    ranks the averaged gradients equal the local ones exactly ((g + g) / 2), so the two ranks must end where a single process running the
    reference's order (lr rule -> estimator step -> PPO backward -> PPO step) ends, bit for bit -- given the same advantage statistics, i.e. the
    single process normalises with the doubled sums the two ranks see (a batch that holds every sample twice)"""
    port = 33500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path), True), nprocs=2, join=True)
    from isaacgymloco_amd.learn.runner import HIMOnPolicyRunner
    torch.manual_seed(100)
    runner = HIMOnPolicyRunner(FakeEnv(8, seed=7), _train_cfg(), log_dir=None, device="cpu")
    assert not runner.dist_ctx.enabled
    runner.alg.storage.advantage_sync = lambda s1, s2, n: (2.0 * s1, 2.0 * s2, 2.0 * n)
    torch.manual_seed(5)
    runner.learn(2, init_at_random_ep_len=False)
    single = runner.alg.actor_critic.state_dict()
    for r in range(2):
        d = torch.load(os.path.join(tmp_path, f"rank{r}.pt"))
        assert d["lr"] == runner.alg.learning_rate
        for k in single:
            torch.testing.assert_close(d["sd"][k], single[k], rtol=0, atol=0, msg=k)


def _grad_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from isaacgymloco_amd.learn.him_ppo import DistCtx
    ctx = DistCtx()
    torch.manual_seed(0)
    lin = torch.nn.Linear(5, 3)
    x = torch.arange(20, dtype=torch.float32).reshape(4, 5)[2 * rank:2 * rank + 2] / 10.0
    lin(x).pow(2).mean().backward()
    ctx.average_grads(list(lin.parameters()))
    s1, s2, n = ctx.sum_triple(x.sum(), (x * x).sum(), torch.tensor(float(x.numel())))
    torch.save({"g": lin.weight.grad.clone(), "stats": torch.stack((s1, s2, n))}, os.path.join(out_dir, f"g{rank}.pt"))
    dist.destroy_process_group()


def test_gradient_average_equals_full_batch(tmp_path):
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_grad_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    torch.manual_seed(0)
    lin = torch.nn.Linear(5, 3)
    x = torch.arange(20, dtype=torch.float32).reshape(4, 5) / 10.0
    lin(x).pow(2).mean().backward()
    for r in range(2):
        d = torch.load(os.path.join(tmp_path, f"g{r}.pt"))
        torch.testing.assert_close(d["g"], lin.weight.grad, rtol=1e-6, atol=1e-7)
        torch.testing.assert_close(d["stats"], torch.stack((x.sum(), (x * x).sum(), torch.tensor(20.0))), rtol=1e-6, atol=1e-6)


# ---- HybridPPO (AMP) on the data-parallel path: the discriminator travels in the same bucket as the PPO group (`more_params`), is
#      averaged, and is NOT clipped (HYBP:270 clips the actor-critic only)
def _hybrid_alg(dist_ctx):
    import test_amp_golden as TA
    from isaacgymloco_amd.learn import amp
    from isaacgymloco_amd.learn.hybrid import HybridPPO
    from isaacgymloco_amd.learn.modules import HIMActorCritic
    fx = np.load(TA.FX)
    ld = TA._loader()
    torch.manual_seed(0)
    ac = HIMActorCritic(270, 238, 45, 12, actor_hidden_dims=[512, 256, 128], critic_hidden_dims=[512, 256, 128], activation="elu", init_noise_std=1.0)
    disc = amp.AMPDiscriminator(60, 0.5 * 0.02, [1024, 512], "cpu", 0.3)
    nz = amp.Normalizer(30)
    alg = HybridPPO(ac, disc, ld, nz, device="cpu", min_std=torch.tensor([0.05, 0.02, 0.05] * 4) * 1.5, dist_ctx=dist_ctx, **TA.ALG)
    N, T = 8, 6
    alg.init_storage(N, T, [270], [238], [12])
    return alg, fx, N, T


def _hybrid_run(alg, fx, N, T):
    obs, crit, ampo = (torch.from_numpy(fx[k]) for k in ("hy_obs", "hy_crit", "hy_amp"))
    rew, done = torch.from_numpy(fx["hy_rew"]), torch.from_numpy(fx["hy_done"])
    torch.manual_seed(1)
    np.random.seed(7)
    with torch.inference_mode():
        for t in range(T):
            alg.act(obs[t], crit[t], ampo[t])
            r = alg.discriminator.predict_amp_reward(ampo[t], ampo[t + 1], rew[t], normalizer=alg.amp_normalizer)[0]
            alg.process_env_step(r, done[t], {"time_outs": done[t] & False}, ampo[t + 1], crit[t + 1])
        alg.compute_returns(crit[T])
    torch.manual_seed(2)
    alg.update()
    return {"ac": {k: v.clone() for k, v in alg.actor_critic.state_dict().items()}, "disc": {k: v.clone() for k, v in alg.discriminator.state_dict().items()},
            "lr": alg.learning_rate, "nz": (alg.amp_normalizer._mean.clone(), alg.amp_normalizer._var.clone(), alg.amp_normalizer._count.clone())}


def _hybrid_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from isaacgymloco_amd.learn.him_ppo import DistCtx
    ctx = DistCtx()
    alg, fx, N, T = _hybrid_alg(ctx)
    c0 = ctx.collectives
    out = _hybrid_run(alg, fx, N, T)
    out["collectives"] = ctx.collectives - c0
    torch.save(out, os.path.join(out_dir, f"hy{rank}.pt"))
    dist.destroy_process_group()


def test_hybrid_ppo_data_parallel_order_equals_the_single_rank_order(tmp_path):
    """two gloo ranks fed the SAME AMP fixture data end, bit for bit, where one process running HybridPPO's single-rank order ends (given the
    doubled advantage / normaliser sums a batch that holds every sample twice has): actor-critic, DISCRIMINATOR, learning rate and the
    running normaliser.  That pins the `more_params` bucket (the discriminator's gradients are averaged with the PPO group's, and only the
    actor-critic is clipped) and the normaliser's moment all-reduce."""
    port = 35500 + (os.getpid() % 2000)
    mp.spawn(_hybrid_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    alg, fx, N, T = _hybrid_alg(None)
    alg.storage.advantage_sync = lambda s1, s2, n: (2.0 * s1, 2.0 * s2, 2.0 * n)
    alg.amp_normalizer.moment_sync = lambda a, b, c: (2.0 * a, 2.0 * b, 2.0 * c)
    single = _hybrid_run(alg, fx, N, T)
    for r in range(2):
        d = torch.load(os.path.join(tmp_path, f"hy{r}.pt"), weights_only=False)
        assert d["lr"] == single["lr"]
        # per minibatch one gradient bucket + two normaliser moment reductions; per update the advantage statistics
        assert d["collectives"] == 2 * 2 * (1 + 2) + 1, d["collectives"]
        for part in ("ac", "disc"):
            for k in single[part]:
                torch.testing.assert_close(d[part][k], single[part][k], rtol=0, atol=0, msg=f"{part}.{k}")
        for a, b in zip(d["nz"], single["nz"]):
            torch.testing.assert_close(a, b, rtol=0, atol=0)


def _worker8(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), OMP_NUM_THREADS="1")
    sys.path.insert(0, ROOT)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from isaacgymloco_amd.learn.bench_train import train_cfg_dict, weights_digest
    from isaacgymloco_amd.learn.runner import HIMOnPolicyRunner
    tc = train_cfg_dict("aliengo")                       # the reference's update: 5 epochs x 4 minibatches (AGC:316-317)
    tc["runner"]["num_steps_per_env"] = 4
    torch.manual_seed(100 + rank)                         # different initial weights per rank: the broadcast must fix that
    runner = HIMOnPolicyRunner(FakeEnv(4, seed=7 + rank), tc, log_dir=None, device="cpu")
    torch.manual_seed(5 + rank)
    per_iter, blocked = [], []
    runner.dist_ctx.timing = True                         # what bench.py's multi-rank line reports per rank (per_rank.collective_blocked_s)
    for _ in range(2):
        c0 = runner.dist_ctx.collectives
        runner.learn(1, init_at_random_ep_len=False)
        per_iter.append(runner.dist_ctx.collectives - c0)
        blocked.append(runner.dist_ctx.take_blocked_seconds())
    torch.save({"digest": weights_digest(runner.alg.actor_critic), "lr": runner.alg.learning_rate, "collectives": per_iter, "world": runner.dist_ctx.world,
                "blocked": blocked},
               os.path.join(out_dir, f"rank{rank}.pt"))
    dist.destroy_process_group()


def test_eight_ranks_stay_in_lockstep_with_21_collectives_per_iteration(tmp_path):
    """SURVEY.md 8(e) / BASELINE config 5 at its real width (VERDICT r4 task 6): eight rank processes (gloo, CPU) with the reference's
    5 x 4 minibatches take identical optimiser steps -- bit-identical weight digests and learning rate on all eight -- and issue exactly
    21 collectives per PPO iteration each (20 merged gradient + KL buckets, one advantage-statistics sum); nothing else crosses ranks."""
    port = 35500 + (os.getpid() % 2000)
    mp.spawn(_worker8, args=(8, port, str(tmp_path)), nprocs=8, join=True)
    outs = [torch.load(os.path.join(tmp_path, f"rank{r}.pt")) for r in range(8)]
    assert all(o["world"] == 8 for o in outs)
    assert all(o["collectives"] == [21, 21] for o in outs), [o["collectives"] for o in outs]
    assert all(o["digest"] == outs[0]["digest"] and o["lr"] == outs[0]["lr"] for o in outs), [o["digest"] for o in outs]
    # every rank timed its waits on the gradient collectives (host clock on gloo): positive, and far below the iteration's wall time
    assert all(len(o["blocked"]) == 2 and all(0.0 < b < 60.0 for b in o["blocked"]) for o in outs), [o["blocked"] for o in outs]


def test_mixed_robot_mapping_puts_ranks_4_to_7_on_the_go2_table():
    """bench.py --mixed-robots at world size 8: ranks 0-3 simulate --task's robot, ranks 4-7 Go2 -- a different model table (total mass), the
    same observation / action layout -- one asset per process as in the reference (LR:1133-1135)"""
    import bench
    from isaacgymloco_amd.envs import config as C
    from isaacgymloco_amd.envs.legged_robot import build_robot_model
    tasks = [bench.task_of_rank("aliengo", r, 8, True) for r in range(8)]
    assert tasks == ["aliengo"] * 4 + ["go2"] * 4
    assert [bench.task_of_rank("aliengo", r, 8, False) for r in range(8)] == ["aliengo"] * 8 and bench.task_of_rank("aliengo", 0, 1, True) == "aliengo"
    mass = {}
    for t in ("aliengo", "go2"):
        cfg = C.TASKS[t][0]()
        m = build_robot_model(cfg.asset)
        mass[t] = sum(m.bodies[i].mass for i in range(17))
        assert (cfg.env.num_observations, cfg.env.num_privileged_obs, cfg.env.num_actions) == (270, 238, 12)
    assert abs(mass["aliengo"] - mass["go2"]) > 5.0, mass


def _fake_sysfs(root, numa_cpus, gpu_numa):
    """a node with len(numa_cpus) NUMA nodes and len(gpu_numa) GPUs behind a KFD topology (CPU nodes first, as the driver lists them)"""
    for n, cl in enumerate(numa_cpus):
        d = os.path.join(root, "devices", "system", "node", f"node{n}")
        os.makedirs(d)
        open(os.path.join(d, "cpulist"), "w").write(cl + "\n")
    top = os.path.join(root, "class", "kfd", "kfd", "topology", "nodes")
    for n in range(len(numa_cpus)):
        os.makedirs(os.path.join(top, str(n)))
        open(os.path.join(top, str(n), "properties"), "w").write("cpu_cores_count 64\nsimd_count 0\nlocation_id 0\ndomain 0\n")
    for i, nn in enumerate(gpu_numa):
        d = os.path.join(top, str(len(numa_cpus) + i))
        os.makedirs(d)
        bus = 0x05 + 0x10 * i
        open(os.path.join(d, "properties"), "w").write(f"cpu_cores_count 0\nsimd_count 1024\nlocation_id {bus << 8}\ndomain 0\n")
        pd = os.path.join(root, "bus", "pci", "devices", f"0000:{bus:02x}:00.0")
        os.makedirs(pd)
        open(os.path.join(pd, "numa_node"), "w").write(f"{nn}\n")


def test_host_cpu_plan_gives_every_rank_its_own_cpus_next_to_its_gpu(tmp_path):
    """bench.host_cpu_plan (the CPU set a rank process binds to before torch / HIP start): disjoint sets, on the GPU's NUMA node when the KFD
    topology says which one that is, otherwise an even split in NUMA order; eight ranks on a 2 x 64-core, 256-thread host"""
    import bench
    root = str(tmp_path / "sys")
    # sockets interleave their SMT siblings: node0 = 0-63,128-191; GPUs 0-3 on socket 1 (!), 4-7 on socket 0: the plan must follow the topology
    _fake_sysfs(root, ["0-63,128-191", "64-127,192-255"], [1, 1, 1, 1, 0, 0, 0, 0])
    plan, how = bench.host_cpu_plan(8, sysfs=root, allowed=range(256))
    assert how.startswith("kfd topology")
    node0, node1 = set(bench._parse_cpulist("0-63,128-191")), set(bench._parse_cpulist("64-127,192-255"))
    assert all(len(p) == 32 for p in plan) and len(set().union(*map(set, plan))) == 256          # disjoint, everything used
    assert all(set(plan[r]) <= node1 for r in range(4)) and all(set(plan[r]) <= node0 for r in range(4, 8))
    # a container that only allows 64 CPUs of socket 0: the GPUs of socket 1 cannot get local CPUs -> even split of what is allowed
    plan, how = bench.host_cpu_plan(8, sysfs=root, allowed=range(64))
    assert how.startswith("even split") and [len(p) for p in plan] == [8] * 8 and sorted(sum(plan, [])) == list(range(64))
    # no topology at all (this container): even split; more ranks than CPUs: everybody shares
    plan, how = bench.host_cpu_plan(2, sysfs=str(tmp_path / "nothing"), allowed=[3, 4, 5, 9])
    assert plan == [[3, 4], [5, 9]] and how.startswith("even split")
    plan, _ = bench.host_cpu_plan(4, sysfs=str(tmp_path / "nothing"), allowed=[0, 1])
    assert plan[0] == [0] and plan[1] == [1] and plan[2] == [0, 1] and plan[3] == [0, 1]
    assert bench._cpu_ranges([0, 1, 2, 3, 8, 9, 11]) == "0-3,8-9,11"
