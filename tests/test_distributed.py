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
