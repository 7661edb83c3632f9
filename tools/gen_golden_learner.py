"""Container-only: golden vectors for the learner side from the reference rsl_rl.
Runs rsl_rl.algorithms.HIMPPO (HIMP:38-198) + HIMRolloutStorage (HST) + HIMActorCritic/HIMEstimator (HAC/HES) on a tiny
synthetic rollout with fixed seeds and stores inputs + outputs (returns, advantages, losses, parameter checksums)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
import refenv  # noqa: E402

refenv.install()
from rsl_rl.algorithms import HIMPPO  # noqa: E402
from rsl_rl.modules import HIMActorCritic  # noqa: E402
from rsl_rl.modules.him_estimator import sinkhorn  # noqa: E402

N, T = 8, 6
N_LARGE, T_LARGE = 128, 64      # 8192 transitions in 2 minibatches of 4096: the size at which the build's MFMA weight-gradient kernels engage
ALG = dict(value_loss_coef=1.0, use_clipped_value_loss=True, clip_param=0.2, entropy_coef=0.01, num_learning_epochs=2,
           num_mini_batches=2, learning_rate=1e-3, schedule="adaptive", gamma=0.99, lam=0.95, desired_kl=0.01, max_grad_norm=1.0)


def checksums(module):
    return {k: np.array([float(v.double().sum()), float(v.double().abs().sum())]) for k, v in module.state_dict().items()}


def synth_inputs(n, t, seed):
    """the synthetic rollout both sides consume; regenerated from the seed by the tests (torch's CPU generator is deterministic)"""
    g = torch.Generator().manual_seed(seed)
    obs_seq = torch.randn(t + 1, n, 270, generator=g)
    crit_seq = torch.randn(t + 1, n, 238, generator=g)
    rew_seq = torch.randn(t, n, generator=g)
    done_seq = torch.rand(t, n, generator=g) < 0.2
    tout_seq = done_seq & (torch.rand(t, n, generator=g) < 0.5)
    return obs_seq, crit_seq, rew_seq, done_seq, tout_seq


def main_large():
    """same procedure at N = 128, T = 64; inputs are NOT stored (9 MB): only the seed, their checksums, the sampled actions and the outputs"""
    n, t = N_LARGE, T_LARGE
    torch.manual_seed(0)
    ac = HIMActorCritic(270, 238, 45, 12, actor_hidden_dims=[512, 256, 128], critic_hidden_dims=[512, 256, 128], activation="elu", init_noise_std=1.0)
    alg = HIMPPO(ac, device="cpu", **ALG)
    alg.init_storage(n, t, [270], [238], [12])
    obs_seq, crit_seq, rew_seq, done_seq, tout_seq = synth_inputs(n, t, 321)
    torch.manual_seed(1)
    actions = []
    with torch.inference_mode():
        for k in range(t):
            a = alg.act(obs_seq[k], crit_seq[k])
            actions.append(a.clone())
            alg.process_env_step(rew_seq[k], done_seq[k], {"time_outs": tout_seq[k]}, crit_seq[k + 1])
        alg.compute_returns(crit_seq[t])
    out = dict(n=np.array(n), t=np.array(t), input_seed=np.array(321),
               input_checksums=np.array([float(x.double().sum()) for x in (obs_seq, crit_seq, rew_seq, done_seq.float(), tout_seq.float())]),
               actions=torch.stack(actions).numpy(), returns=alg.storage.returns.clone().numpy(), advantages=alg.storage.advantages.clone().numpy(),
               values=alg.storage.values.clone().numpy())
    torch.manual_seed(2)
    out["perm"] = torch.randperm(n * t).numpy()       # what mini_batch_generator is about to draw (HST:140)
    torch.manual_seed(2)
    losses = alg.update()
    out["losses"] = np.array(losses, dtype=np.float64)
    out["final_lr"] = np.float64(alg.learning_rate)
    for k, v in checksums(ac).items():
        out["final/" + k] = v
    path = os.path.join(ROOT, "tests", "golden", "learner_himppo_large.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) / 1e3, "KB; losses", losses, "lr", alg.learning_rate)


def main():
    torch.manual_seed(0)
    ac = HIMActorCritic(270, 238, 45, 12, actor_hidden_dims=[512, 256, 128], critic_hidden_dims=[512, 256, 128], activation="elu", init_noise_std=1.0)
    alg = HIMPPO(ac, device="cpu", **ALG)
    alg.init_storage(N, T, [270], [238], [12])
    init_ck = checksums(ac)
    g = torch.Generator().manual_seed(123)
    obs_seq = torch.randn(T + 1, N, 270, generator=g)
    crit_seq = torch.randn(T + 1, N, 238, generator=g)
    rew_seq = torch.randn(T, N, generator=g)
    done_seq = torch.rand(T, N, generator=g) < 0.2
    tout_seq = done_seq & (torch.rand(T, N, generator=g) < 0.5)
    torch.manual_seed(1)
    actions = []
    with torch.inference_mode():
        for t in range(T):
            a = alg.act(obs_seq[t], crit_seq[t])
            actions.append(a.clone())
            alg.process_env_step(rew_seq[t], done_seq[t], {"time_outs": tout_seq[t]}, crit_seq[t + 1])
        alg.compute_returns(crit_seq[T])
    returns = alg.storage.returns.clone().numpy()
    advantages = alg.storage.advantages.clone().numpy()
    values = alg.storage.values.clone().numpy()
    torch.manual_seed(2)
    losses = alg.update()
    final_ck = checksums(ac)
    sk_in = torch.randn(16, 32, generator=g)
    sk_out = sinkhorn(sk_in.clone()).numpy()
    out = dict(obs_seq=obs_seq.numpy(), crit_seq=crit_seq.numpy(), rew_seq=rew_seq.numpy(), done_seq=done_seq.numpy(), tout_seq=tout_seq.numpy(),
               actions=torch.stack(actions).numpy(), returns=returns, advantages=advantages, values=values, losses=np.array(losses, dtype=np.float64),
               final_lr=np.float64(alg.learning_rate), sinkhorn_in=sk_in.numpy(), sinkhorn_out=sk_out)
    for k, v in init_ck.items():
        out["init/" + k] = v
    for k, v in final_ck.items():
        out["final/" + k] = v
    path = os.path.join(ROOT, "tests", "golden", "learner_himppo.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) / 1e3, "KB; losses", losses, "lr", alg.learning_rate)


if __name__ == "__main__":
    main()
    main_large()
