"""HIMPPO: clipped PPO with adaptive-KL learning rate plus the HIM estimator update.
Same surface and update rule as rsl_rl.algorithms.HIMPPO (HIMP:38-198): act / process_env_step / compute_returns /
update, value-clip, entropy bonus, KL-adaptive lr x/1.5 (HIMP:144-156), grad-clip, estimator stepped first with the PPO lr.

Data parallel (not in the reference): `dist_ctx` averages gradients over ranks so that every rank takes identical optimiser
steps.  ONE collective per minibatch, 21 per PPO iteration (20 minibatches + the advantage statistics; round 3: 41, round 2: 61):
  * both optimisers' gradients -- the estimator's 0.24 MB and the PPO group's 2.2 MB -- and the KL estimate of the adaptive-lr rule live in ONE
    persistent flat buffer (fused_linear.GradArena: the weight-gradient kernels write into it, the loss kernel writes its statistics into
    its tail), which is all-reduced in place (RCCL over xGMI, ReduceOp.AVG: a latency-bound 2.4 MB payload) right behind the PPO backward;
  * then the lr rule, the estimator's step and the clipped PPO step (HIMP:183: clip AFTER the reduce) follow in the single-rank order.  The
    estimator's step does not feed the PPO backward (the actor's input features are detached and were computed before it, HAC:136-141),
    so running it behind that backward changes nothing.
Round 3 reduced the estimator's bucket separately, in flight under the PPO backward; that hid 0.24 MB of transfer but paid a second
torch.distributed call and two more stream hand-offs per minibatch in a loop that is within 1.6 x of launch-bound: a one-rank group with
every collective issued cost 4.5 % (0.0997 vs 0.0954 s per iteration), of which the device shows 40 x ~35 us of idle around the hand-offs and
the rest is host time (profiles/r04_collective_overhead.json, profiles/r04_trace_idle_rccl.txt).  CPU tensors (gloo tests) keep the concatenated form of the same bucket.
"""
import os
import time

import torch
import torch.nn as nn

from .storage import HIMRolloutStorage


_two_stream_memo = {}


def _loaded_tunableop_solutions(device):
    """{shape signature: solution} of the GEMM entries TunableOp ACTUALLY holds in this process.  The table file is read lazily, at the first
    tunable GEMM, and is rejected as a whole when its Validator lines (torch / HIP / hipBLASLt / rocBLAS versions, GPU architecture) do not
    match this build -- every GEMM then runs hipBLASLt's default heuristics although the file on disk lists explicit solutions.
    So: run one small GEMM, then ask TunableOp what it loaded."""
    tun = torch.cuda.tunable
    a = torch.zeros(8, 8, device=device)
    torch.mm(a, a)
    table = {}
    for rec in tun.get_results():
        if len(rec) >= 3 and str(rec[0]).startswith("Gemm"):
            table[str(rec[1]).split("_ld_")[0]] = str(rec[2])
    return table


def tunableop_status(device, table_path=None):
    """what TunableOp really does in this process (bench.py prints it; VERDICT r4: a table that is rejected at load silently costs the
    update its second stream): enabled / tuning flags, the GEMM entries LOADED, and -- when `table_path` names the shipped table -- its
    Validator lines against this build's (torch / HIP / hipBLASLt / rocBLAS versions, GPU architecture string): one mismatch rejects the
    whole file."""
    tun = getattr(torch.cuda, "tunable", None)
    out = {"enabled": bool(tun is not None and tun.is_enabled()), "tuning": bool(tun is not None and tun.tuning_is_enabled()),
           "entries_loaded": 0, "explicit_solutions_loaded": 0, "entries_in_table": None, "validators_match": None, "validator_mismatches": None}
    if not out["enabled"]:
        return out
    try:
        table = _loaded_tunableop_solutions(device)
        out["entries_loaded"] = len(table)
        out["explicit_solutions_loaded"] = sum(1 for v in table.values() if v != "Default")
    except Exception as e:
        out["error"] = f"{type(e).__name__}: {e}"
    if table_path and os.path.exists(table_path):
        want, n = {}, 0
        for line in open(table_path):
            f = line.rstrip("\n").split(",")
            if f[0] == "Validator" and len(f) >= 3:
                want[f[1]] = ",".join(f[2:])
            elif len(f) >= 3:
                n += 1
        out["entries_in_table"] = n
        try:
            have = {str(k): str(v) for k, v in tun.get_validators()}
            bad = {k: {"table": v, "this_build": have.get(k)} for k, v in want.items() if have.get(k) != v}
            out["validators_match"] = not bad
            out["validator_mismatches"] = bad or None
        except Exception as e:
            out["error"] = f"{type(e).__name__}: {e}"
    return out


def _two_streams_allowed(critic, rows, multi_rank=False, single_device_ranks=False):
    """May the critic chain of the update run on a side stream, concurrently with the actor / estimator chain?  (LSIM_UPDATE_STREAMS = 0 / 1
    forces it off / on.)  Two library GEMMs in flight at once are only safe when neither is a kernel whose workgroups wait for each other
    (hipBLASLt's default heuristics pick such stream-K style kernels for some of these shapes: with TunableOp off, two concurrent GEMM streams
    hung the device at N = 4096, round 3).  So the automatic answer is yes only when TunableOp is enabled in look-up mode AND it has LOADED an
    explicit (non-"Default") solution for every BLAS GEMM of the critic chain at this minibatch size: forward (tn_), input gradients (nn_)
    and the weight gradients (nt_) of the layers the library's own weight-gradient kernel leaves to BLAS.  That holds for the shipped gfx950
    table on the build it was tuned with, at the BASELINE minibatch of 102 400 rows (the configuration the overlap was measured on:
    -1.6 % update time); on any other build the table is rejected at load, nothing is found here, and the answer is no.
    Several ranks: every rank process owns its GPU, so the kernels that share a device are exactly those of the one-rank case plus RCCL's
    (which wait for the peer GPU, not for a workgroup of another kernel on this one); explicit non-cooperative GEMM solutions cannot form a
    wait cycle with them, and a 1-rank RCCL group with every collective issued runs both streams (tests/test_bench_cli.py).  What does hang is
    the DEBUG mode with several rank processes on ONE GPU (LSIM_DEBUG_SINGLE_DEVICE, gloo): four GEMM streams of two processes time-sliced on
    one device -- `single_device_ranks` keeps the side stream off there."""
    mode = os.environ.get("LSIM_UPDATE_STREAMS", "auto")
    if mode in ("0", "1"):
        return mode == "1"
    if multi_rank and (single_device_ranks or os.environ.get("LSIM_DEBUG_SINGLE_DEVICE") == "1"):
        return False
    tun = getattr(torch.cuda, "tunable", None)
    if tun is None or not tun.is_enabled() or tun.tuning_is_enabled():
        return False
    dims = tuple((m.in_features, m.out_features) for m in critic if isinstance(m, nn.Linear))
    key = (rows, dims, tun.get_filename())
    if key not in _two_stream_memo:
        ok = False
        try:
            from .fused_linear import _eligible
            table = _loaded_tunableop_solutions(next(critic.parameters()).device)
            need = []
            for i, (k, n) in enumerate(dims):
                if n > 1:
                    need.append(f"tn_{n}_{rows}_{k}")                  # forward  y = x W^T (+ bias); a single output column is a GEMV
                if i > 0:
                    need.append(f"nn_{k}_{rows}_{n}")                  # input gradient  g W
                if not _eligible(rows, k, n):
                    need.append(f"nt_{k}_{n}_{rows}")                  # weight gradient  g^T x through BLAS
            ok = bool(table) and all(table.get(sig, "Default") != "Default" for sig in need)
        except Exception:
            ok = False
        _two_stream_memo[key] = ok
    return _two_stream_memo[key]


class DistCtx:
    """Thin helper around torch.distributed for gradient / scalar averaging (world_size 1 => no-ops)."""

    def __init__(self):
        import torch.distributed as dist
        self.dist = dist
        # LSIM_DEBUG_FORCE_COLLECTIVES=1: issue every collective even in a group of one rank (bench.py then opens a 1-rank RCCL group), so that
        # the RCCL calls of the N > 1 path can be exercised on a box with a single GPU
        forced = os.environ.get("LSIM_DEBUG_FORCE_COLLECTIVES") == "1"
        self.enabled = dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or forced)
        self.world = dist.get_world_size() if self.enabled else 1
        self.collectives = 0            # number of collectives issued so far (tests / DESIGN.md section 8 count them per iteration)
        # time the rank spends BLOCKED in gradient collectives (bench.py's multi-rank line, VERDICT r5 task 6): off by default.  On a GPU the wait of
        # finish_bucket() is a stream dependency, so the blocked time is device time -- an event pair on the compute stream around the wait: elapsed =
        # how long that stream stood still for the collective (~ 0 when it had finished under the kernels issued in between); on the CPU (gloo) the
        # wait blocks the host and is timed with the host clock
        self.timing = False
        self._ev_pool, self._ev_live, self._blocked_host_s = [], [], 0.0

    def average_grads(self, params):
        """one flattened all-reduce per optimiser step; the parameters' .grad are (or become) views of the reduced buffer"""
        if not self.enabled:
            return
        self.finish_bucket(self.reduce_bucket_async([p for p in params if p.grad is not None]))

    def reduce_bucket_async(self, params, extra=None, key=None, n_extra=None, extra_at=0):
        """start the all-reduce of one flattened bucket [gradients of `params`..., extra] WITHOUT waiting for it: RCCL works on its own stream
        (ordered behind what the compute stream has issued so far), kernels issued after this call overlap with it.  finish_bucket() waits.
        = prepare_bucket + start_reduce."""
        return self.start_reduce(self.prepare_bucket(params, extra, key, n_extra, extra_at))

    def prepare_bucket(self, params, extra=None, key=None, n_extra=None, extra_at=0):
        """the flat bucket of one reduce, filled: -> (flat, pieces, n_extra).  With a gradient arena (fused_linear.GradArena: CUDA parameters)
        the bucket is a PERSISTENT flat buffer that the gradients already live in -- the weight-gradient kernels wrote them there -- so there
        is no concatenation and nothing to re-view afterwards; the few gradients that autograd's own kernels produced are copied into their
        slices by one multi-tensor launch (Bucket.adopt).  Device work only."""
        from . import fused_linear as FL
        if n_extra is None:
            n_extra = extra.numel() if extra is not None else 0
            extra_at = 0
        if FL._arena is not None and params[0].is_cuda:
            # n_extra / extra_at: size of the bucket's tail and where `extra` goes in it (the PPO loss kernel's 5 statistics: slot 3 is the KL
            # estimate; when the kernel wrote them there itself, extra is None and nothing is copied)
            b = FL._arena.bucket(key if key is not None else tuple(id(p) for p in params), params, n_extra)
            b.adopt()
            if extra is not None:
                b.extra_view[extra_at:extra_at + extra.numel()].copy_(extra.detach().reshape(-1))
            return b.flat, None, n_extra
        parts = [p.grad.reshape(-1) for p in params]
        n_extra = extra.numel() if extra is not None else 0
        if extra is not None:
            parts.append(extra.detach().reshape(-1).to(parts[0].dtype))
        return torch.cat(parts), params, n_extra

    def start_reduce(self, bucket):
        flat, pieces, n_extra = bucket
        self.collectives += 1
        avg = self._avg_op(flat)
        work = self._all_reduce_async(flat, avg if avg is not None else self.dist.ReduceOp.SUM)
        return flat, work, pieces, n_extra, avg is not None

    def _all_reduce_async(self, flat, op):
        """dist.all_reduce(flat, op, async_op=True) on the default group through the ProcessGroup object itself: the Python wrapper's argument
        checks and group look-ups are a third of the host time of a call, in a loop whose host time shows"""
        pg = getattr(self, "_pg", None)
        if pg is None:
            try:
                pg = self._pg = self.dist.distributed_c10d._get_default_group()
            except Exception:
                pg = self._pg = False
        if pg:
            try:
                opts = self.dist.AllreduceOptions()
                opts.reduceOp = op
                return pg.allreduce([flat], opts)
            except Exception:
                self._pg = False
        return self.dist.all_reduce(flat, op=op, async_op=True)

    def _avg_op(self, t):
        """ReduceOp.AVG where the backend has it (RCCL / NCCL: the division happens inside the collective); gloo sums and we scale"""
        return self.dist.ReduceOp.AVG if t.is_cuda and self.dist.get_backend() == "nccl" else None

    def finish_bucket(self, handle):
        """wait; every parameter's .grad becomes a view of its slice of the reduced buffer; returns the averaged `extra` values (or None)"""
        flat, work, params, n_extra, averaged = handle
        if not self.timing:
            work.wait()
        elif flat.is_cuda:
            e0, e1 = self._ev_pool.pop() if self._ev_pool else (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            e0.record()
            work.wait()
            e1.record()
            self._ev_live.append((e0, e1))
        else:
            t0 = time.perf_counter()
            work.wait()
            self._blocked_host_s += time.perf_counter() - t0
        if not averaged:
            flat.div_(self.world)
        if params is None:                  # arena bucket: the gradients ARE the buffer
            return flat[flat.numel() - n_extra:] if n_extra else None
        sizes = [p.grad.numel() for p in params]
        pieces = flat.split(sizes + ([n_extra] if n_extra else []))
        for p, v in zip(params, pieces):
            p.grad = v.view_as(p.grad)
        return pieces[-1] if n_extra else None

    def take_blocked_seconds(self):
        """seconds blocked in gradient collectives since the last call (timing = True); call after a device synchronisation"""
        s = self._blocked_host_s
        self._blocked_host_s = 0.0
        for e0, e1 in self._ev_live:
            s += e0.elapsed_time(e1) * 1e-3
        self._ev_pool += self._ev_live
        self._ev_live = []
        return s

    def agree(self, flag, device):
        """True iff `flag` is true on EVERY rank (one tiny MIN all-reduce; used once per configuration, not per iteration, and not counted in
        `collectives`): per-rank decisions that change the launch pattern -- the update's side stream -- are taken jointly, so that all ranks
        run the same schedule (a rank whose TunableOp table was rejected would otherwise run one stream beside ranks that run two)"""
        if not self.enabled:
            return bool(flag)
        t = torch.tensor([1.0 if flag else 0.0], device=device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN)
        return bool(t.item() > 0.5)

    def average_scalar(self, t):
        if not self.enabled:
            return t
        t = t.clone()
        self.collectives += 1
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return t / self.world

    def sum_triple(self, a, b, c):
        if not self.enabled:
            return a, b, c
        v = torch.stack((a, b, c)).to(torch.float64)
        self.collectives += 1
        self.dist.all_reduce(v, op=self.dist.ReduceOp.SUM)
        v = v.to(torch.float32)
        return v[0], v[1], v[2]

    def sum_triple_vec(self, a, b, c):
        """element-wise sums over ranks of two vectors and a scalar (running-normaliser moments)"""
        if not self.enabled:
            return a, b, c
        v = torch.cat((a.reshape(-1), b.reshape(-1), c.reshape(1))).to(torch.float64)
        self.collectives += 1
        self.dist.all_reduce(v, op=self.dist.ReduceOp.SUM)
        n = a.numel()
        return v[:n].view_as(a), v[n:2 * n].view_as(b), v[2 * n]

    def broadcast_module(self, module):
        if not self.enabled:
            return
        for t in list(module.parameters()) + list(module.buffers()):
            self.dist.broadcast(t.data, src=0)


class HIMPPO:
    def __init__(self, actor_critic, num_learning_epochs=1, num_mini_batches=1, clip_param=0.2, gamma=0.998, lam=0.95,
                 value_loss_coef=1.0, entropy_coef=0.0, learning_rate=1e-3, max_grad_norm=1.0, use_clipped_value_loss=True,
                 schedule="fixed", desired_kl=0.01, device="cpu", dist_ctx=None):
        self.device = device
        self.desired_kl, self.schedule, self.learning_rate = desired_kl, schedule, learning_rate
        self.actor_critic = actor_critic.to(device)
        self.storage = None
        self.optimizer = torch.optim.Adam(self.actor_critic.parameters(), lr=learning_rate)
        self.transition = HIMRolloutStorage.Transition()
        self.clip_param, self.num_learning_epochs, self.num_mini_batches = clip_param, num_learning_epochs, num_mini_batches
        self.value_loss_coef, self.entropy_coef, self.gamma, self.lam = value_loss_coef, entropy_coef, gamma, lam
        self.max_grad_norm, self.use_clipped_value_loss = max_grad_norm, use_clipped_value_loss
        self.dist_ctx = dist_ctx
        self._lr_t = None      # device-resident learning rate (enable_device_lr)
        if dist_ctx is not None and dist_ctx.enabled:
            dist_ctx.broadcast_module(self.actor_critic)
            self.actor_critic.estimator.grad_sync = dist_ctx.average_grads

    def enable_device_lr(self):
        """GPU fast path of the optimiser side (same update rule, HIMP:144-156 / HIMP:181-184): the learning rate becomes a device scalar that
        the adaptive-KL rule rewrites with one tiny kernel (lsim_adaptive_lr) instead of a host read-back per minibatch, and both Adam
        optimisers become torch's single-kernel (`fused=True`) implementation reading that scalar.  Together 10 % of the update."""
        dev = next(self.actor_critic.parameters()).device
        if dev.type != "cuda" or self._lr_t is not None:
            return self._lr_t is not None
        self._lr_t = torch.tensor(float(self.learning_rate), device=dev, dtype=torch.float32)

        def rebuild(opt):
            keep = ("params", "weight_decay", "betas", "eps", "amsgrad", "maximize", "name")
            new = torch.optim.Adam([{k: v for k, v in g.items() if k in keep} for g in opt.param_groups], lr=self._lr_t, fused=True)
            if opt.state:
                new.load_state_dict(opt.state_dict())
            for g in new.param_groups:
                g["lr"] = self._lr_t
            return new
        self.optimizer = rebuild(self.optimizer)
        est = self.actor_critic.estimator
        est.optimizer = rebuild(est.optimizer)
        est.fused_step = True
        return True

    def _grad_arena(self, more_params=()):
        """persistent gradient buckets on the GPU (fused_linear.GradArena), created at the first update: the estimator's parameters + one slot
        for the KL estimate, and everything else the PPO optimiser steps (+ `more_params`: HybridPPO's discriminator)"""
        from . import fused_linear as FL
        ac = self.actor_critic
        if getattr(self, "_arena", None) is None:
            if not next(ac.parameters()).is_cuda or os.environ.get("LSIM_GRAD_ARENA", "1") == "0":
                return None
            self._arena = FL.GradArena()
        if FL._arena is not self._arena:
            FL.set_grad_arena(self._arena)
        return self._arena

    def _two_streams(self, critic, rows, multi_rank):
        """_two_streams_allowed, decided once per minibatch size -- and, with several ranks, decided JOINTLY (all ranks or none).  The side
        stream never overlaps a collective: the one all-reduce of a minibatch is issued after both streams have joined behind the backward
        pass, and the next minibatch's critic forward is ordered behind the optimiser step that waited for it."""
        memo = self.__dict__.setdefault("_two_stream_decision", {})
        key = (rows, bool(multi_rank), os.environ.get("LSIM_UPDATE_STREAMS", "auto"))
        if key not in memo:
            ok = _two_streams_allowed(critic, rows, multi_rank)
            if multi_rank:
                ok = self.dist_ctx.agree(ok, next(critic.parameters()).device)
            memo[key] = ok
        return memo[key]

    def _side_stream(self, device):
        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream(device=device)
        return self._side

    def _clip_and_step(self, optimizer, params, max_grad_norm):
        """clip_grad_norm_ + optimizer.step() (HIMP:183-184); on the device-lr fast path one C-ABI call of two launches"""
        params = list(params)
        if self._lr_t is not None:
            from .fused_linear import adam_clip_step_hip
            if adam_clip_step_hip(optimizer, max_grad_norm, clip_params=params):
                return
        nn.utils.clip_grad_norm_(params, max_grad_norm)
        optimizer.step()

    def _relink_lr(self):
        """optimizer.load_state_dict() restores plain float learning rates: point the groups at the device scalar again"""
        if self._lr_t is not None:
            for opt in (self.optimizer, self.actor_critic.estimator.optimizer):
                for g in opt.param_groups:
                    g["lr"] = self._lr_t

    def _set_learning_rate(self, lr):
        self.learning_rate = float(lr)
        if self._lr_t is not None:
            self._lr_t.fill_(float(lr))

    def init_storage(self, num_envs, num_transitions_per_env, actor_obs_shape, critic_obs_shape, action_shape):
        self.storage = HIMRolloutStorage(num_envs, num_transitions_per_env, actor_obs_shape, critic_obs_shape, action_shape, self.device)
        if self.dist_ctx is not None and self.dist_ctx.enabled:
            self.storage.advantage_sync = self.dist_ctx.sum_triple

    def test_mode(self):
        self.actor_critic.eval()

    def train_mode(self):
        self.actor_critic.train()

    def act(self, obs, critic_obs):
        t, ac = self.transition, self.actor_critic
        t.actions = ac.act(obs).detach()
        t.values = ac.evaluate(critic_obs).detach()
        t.actions_log_prob = ac.get_actions_log_prob(t.actions).detach()
        t.action_mean, t.action_sigma = ac.action_mean.detach(), ac.action_std.detach()
        t.observations, t.critic_observations = obs, critic_obs   # recorded before env.step()
        return t.actions

    def process_env_step(self, rewards, dones, infos, next_critic_obs):
        t = self.transition
        t.next_critic_observations = next_critic_obs.clone()
        t.rewards = rewards.clone()
        t.dones = dones
        if "time_outs" in infos:   # bootstrap on time-outs (HIMP:110-111)
            t.rewards += self.gamma * torch.squeeze(t.values * infos["time_outs"].unsqueeze(1).to(self.device), 1)
        self.storage.add_transitions(t)
        t.clear()
        self.actor_critic.reset(dones)

    def compute_returns(self, last_critic_obs):
        last_values = self.actor_critic.evaluate(last_critic_obs).detach()
        self.storage.compute_returns(last_values, self.gamma, self.lam)

    def _local_kl(self, mu, sigma, old_mu, old_sigma):
        with torch.inference_mode():
            kl = torch.sum(torch.log(sigma / old_sigma + 1.0e-5) + (torch.square(old_sigma) + torch.square(old_mu - mu)) / (2.0 * torch.square(sigma)) - 0.5, dim=-1)
            return torch.mean(kl)

    def _adapt_lr(self, mu, sigma, old_mu, old_sigma, kl_mean=None, already_global=False):
        if self._lr_t is not None and kl_mean is not None:        # device path: no host round trip
            from .. import lib
            if self.dist_ctx is not None and self.dist_ctx.enabled and not already_global:
                kl_mean = self.dist_ctx.average_scalar(kl_mean)
            kl_mean = kl_mean.detach().reshape(1).contiguous()
            lib.check(lib.load().lsim_adaptive_lr(kl_mean.data_ptr(), float(self.desired_kl), 1e-5, 1e-2, 1.5, self._lr_t.data_ptr(),
                                                  torch.cuda.current_stream(self._lr_t.device).cuda_stream), what="lsim_adaptive_lr")
            return
        with torch.inference_mode():
            if kl_mean is None:
                kl_mean = self._local_kl(mu, sigma, old_mu, old_sigma)
            if self.dist_ctx is not None and not already_global:
                kl_mean = self.dist_ctx.average_scalar(kl_mean)
            kl_mean = kl_mean.item()
        if kl_mean > self.desired_kl * 2.0:
            self.learning_rate = max(1e-5, self.learning_rate / 1.5)
        elif self.desired_kl / 2.0 > kl_mean > 0.0:
            self.learning_rate = min(1e-2, self.learning_rate * 1.5)
        for g in self.optimizer.param_groups:
            g["lr"] = self.learning_rate

    def _ppo_loss(self, ac, mu, sigma, value, actions, old_logp, advantages, returns, target_values, old_mu, old_sigma):
        """(total loss, surrogate, value loss, KL mean or None) of HIMP:136-176.  On the GPU one HIP kernel (lsim_ppo_loss: forward, backward
        and the KL estimate of the adaptive-lr rule in a single pass); elsewhere the reference's torch statement."""
        if mu.is_cuda and mu.dtype == torch.float32:
            from .fused_linear import ppo_loss_hip
            loss, st = ppo_loss_hip(mu, sigma, value, actions, old_logp, advantages, returns, target_values, old_mu, old_sigma, self.clip_param,
                                    self.value_loss_coef, self.entropy_coef, self.use_clipped_value_loss, out=self._stats_slot())
            return loss, st[0], st[1], st[3]
        logp = ac.get_actions_log_prob(actions)
        entropy = ac.entropy
        adv = torch.squeeze(advantages)
        ratio = torch.exp(logp - torch.squeeze(old_logp))
        surrogate_loss = torch.max(-adv * ratio, -adv * torch.clamp(ratio, 1.0 - self.clip_param, 1.0 + self.clip_param)).mean()
        if self.use_clipped_value_loss:
            clipped = target_values + (value - target_values).clamp(-self.clip_param, self.clip_param)
            value_loss = torch.max((value - returns).pow(2), (clipped - returns).pow(2)).mean()
        else:
            value_loss = (returns - value).pow(2).mean()
        return surrogate_loss + self.value_loss_coef * value_loss - self.entropy_coef * entropy.mean(), surrogate_loss, value_loss, None

    def _stats_slot(self):
        """the 5-float tail of the gradient bucket, where the loss kernel may write its statistics directly (None until the bucket exists)"""
        ctx = self.dist_ctx
        if ctx is None or not ctx.enabled or getattr(self, "_arena", None) is None:
            return None
        b = self._arena.buckets.get("all")
        return b.extra_view if b is not None and b.extra == 5 else None

    def _step_minibatch_data_parallel(self, ac, obs, next_critic_obs, loss, mu, sigma, old_mu, old_sigma, kl_mean, adaptive, more_params=(),
                                      est_losses=None):
        """the optimiser half of one minibatch in the data-parallel order: both backwards, ONE all-reduce of every gradient + the KL estimate,
        then the same two optimiser steps as the single-rank order (lr rule -> estimator step -> PPO step, HIMP:144-184).  Also the order of
        the single-rank two-stream path (no collectives).  Returns the estimator's (estimation, swap) losses.  Three pieces: device work
        up to the gradients, the collective, device work of the two optimiser steps."""
        st = self._mb_backward(ac, obs, next_critic_obs, loss, mu, sigma, old_mu, old_sigma, kl_mean, adaptive, more_params, est_losses)
        self._mb_reduce(st)
        self._mb_optim(ac, st)
        return st["est"], st["swap"]

    def _mb_backward(self, ac, obs, next_critic_obs, loss, mu, sigma, old_mu, old_sigma, kl_mean, adaptive, more_params=(), est_losses=None):
        """piece 1: both backward passes; every gradient and the KL estimate end up in the arena bucket (GPU) / a flat tensor (CPU)"""
        from . import fused_linear as FL
        ctx, est_mod = self.dist_ctx, ac.estimator
        if ctx is not None and not ctx.enabled:
            ctx = None                                               # single rank: same order, no collectives (the two-stream path of update())
        est_params = list(est_mod.parameters())
        self.optimizer.zero_grad()                                   # every parameter of the optimiser, the estimator's included
        FL.grad_cycle()
        # est_losses: the estimator's losses when update() already formed them (before the critic's stream was joined)
        est, swap, total = est_losses if est_losses is not None else est_mod.losses(obs, next_critic_obs)
        est_mod._primed = None
        with FL.deferred_wgrad_reduce():                             # one summing launch for the partial results of all ~15 layers
            FL.backward_losses(total, loss)                          # estimator; actor / critic / std gradients
        extra = None
        if adaptive:
            extra = kl_mean if kl_mean is not None else self._local_kl(mu, sigma, old_mu, old_sigma)
        # `more_params`: parameters outside the actor-critic that the same optimiser steps (HybridPPO: the discriminator) -- reduced in the same
        # bucket, not clipped (HYBP:270 clips the actor-critic only)
        est_live = [p for p in est_params if p.grad is not None]
        est_ids = {id(p) for p in est_params}
        ppo_params = [p for p in ac.parameters() if p.grad is not None and id(p) not in est_ids]
        more = [p for p in more_params if p.grad is not None]
        st = dict(est=est.detach(), swap=swap.detach(), est_params=est_params, ppo_params=ppo_params, adaptive=adaptive, kl=extra,
                  dist=(mu, sigma, old_mu, old_sigma) if self._lr_t is None else None, bucket=None)
        if ctx is not None:
            slot = self._stats_slot()
            in_place = slot is not None and extra is not None and extra.data_ptr() == slot[3:4].data_ptr()     # the loss kernel already wrote it there
            st["bucket"] = ctx.prepare_bucket(est_live + ppo_params + more, extra=None if in_place else extra, key="all",
                                              n_extra=5 if (extra is not None and extra.is_cuda) else None, extra_at=3)
        elif self._grad_arena() is not None:                         # single rank: the same bucket, so that the optimisers' pointer tables repeat
            self._arena.bucket("all", est_live + ppo_params + more, 0).adopt()
        return st

    def _mb_reduce(self, st):
        """piece 2: the minibatch's ONE collective (clip AFTER the all-reduce, HIMP:183)"""
        if st["bucket"] is None:
            return
        ctx = self.dist_ctx
        got = ctx.finish_bucket(ctx.start_reduce(st["bucket"]))
        if st["kl"] is not None:
            st["kl"] = got[3:4] if got.numel() == 5 else got

    def _mb_optim(self, ac, st):
        """piece 3: learning-rate rule, the estimator's clipped step, the PPO group's clipped step"""
        from . import fused_linear as FL
        est_mod = ac.estimator
        if st["adaptive"]:
            mu, sigma, old_mu, old_sigma = st["dist"] if st["dist"] is not None else (None,) * 4
            self._adapt_lr(mu, sigma, old_mu, old_sigma, st["kl"].reshape(()), already_global=True)
        if self._lr_t is None:                                       # host learning rate: the estimator steps with the PPO rate (HIMP:158)
            est_mod.learning_rate = self.learning_rate
            for g in est_mod.optimizer.param_groups:
                g["lr"] = self.learning_rate
        stepped = False
        if est_mod.fused_step:
            stepped = FL.adam_clip_step_hip(est_mod.optimizer, est_mod.max_grad_norm)
        if not stepped:
            nn.utils.clip_grad_norm_(st["est_params"], est_mod.max_grad_norm)
            est_mod.optimizer.step()
        for p in st["est_params"]:                                   # the PPO optimiser also holds these parameters: as in the reference
            p.grad = None                                            # (zero_grad before the PPO backward) it must not step them
        self._clip_and_step(self.optimizer, st["ppo_params"], self.max_grad_norm)

    def _mb_forward(self, ac, batch, two_streams):
        """forward of one minibatch up to the two losses (HIMP:136-176 and the estimator's loss head) -> dict"""
        (obs, critic_obs, actions, next_critic_obs, target_values, advantages, returns, old_logp, old_mu, old_sigma) = batch
        if two_streams:
            # The critic chain (forward here, backward inside loss.backward(): autograd runs a node on the stream of its forward) goes to a
            # side stream; encoder + actor + estimator stay on the main one.  The two halves carry about the same matrix work (858 k vs
            # 771 k multiply-adds per sample over forward + backward), and their kernels interleave on the device: one chain's
            # bandwidth-bound passes (ELU, stores) and tile-quantisation tails run under the other chain's MFMA-bound GEMMs.
            cur = torch.cuda.current_stream(obs.device)
            side = self._side_stream(obs.device)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                value = ac.evaluate(critic_obs)
        ac.estimator.prime(obs)        # one encoder forward serves the policy features and the estimator loss below
        early_est = None
        if two_streams and os.environ.get("LSIM_EARLY_EST_LOSS", "1") != "0":
            # The estimator's loss head -- target encoder, prototype scores, three Sinkhorn rounds, log-softmax, losses and their gradients:
            # ~15 launches that each leave most of the device idle -- depends on the encoder output alone.  Formed HERE, while the critic's
            # GEMMs run on the side stream, it fills what they leave; behind the join it ran by itself (0.27 ms per minibatch).  Same values.
            early_est = ac.estimator.losses(obs, next_critic_obs)
        # the reference calls act() here (HIMP:141) and throws the sample away; torch.normal(mean, std) validates std >= 0 with a
        # host read-back, i.e. one pipeline drain per minibatch on the GPU: only the distribution is needed
        std_direct = obs.is_cuda and obs.dtype == torch.float32 and ac.std.dim() == 1 and ac.std.numel() <= 60 and \
            os.environ.get("LSIM_PPO_STD_DIRECT", "1") != "0"
        if std_direct:
            # the policy's std is one value per action (HAC:93): the loss kernel takes it as it is (lsim_ppo_loss_std) instead of the
            # broadcast mean * 0 + std the distribution object forms (HAC:147) -- no [B, A] sigma, no backward of the broadcast, no column sum
            mu_direct = ac.actor(ac._actor_input(obs))
        elif obs.is_cuda:
            ac.update_distribution(obs)
        else:
            ac.act(obs)
        if two_streams:
            cur.wait_stream(side)
            value.record_stream(cur)
        else:
            value = ac.evaluate(critic_obs)
        mu, sigma = (mu_direct, ac.std) if std_direct else (ac.action_mean, ac.action_std)
        loss, surrogate_loss, value_loss, kl_mean = self._ppo_loss(ac, mu, sigma, value, actions, old_logp, advantages, returns, target_values,
                                                                   old_mu, old_sigma)
        return dict(obs=obs, next_critic_obs=next_critic_obs, loss=loss, mu=mu, sigma=sigma, old_mu=old_mu, old_sigma=old_sigma, kl_mean=kl_mean,
                    surrogate_loss=surrogate_loss, value_loss=value_loss, early_est=early_est)

    def update(self):
        ac = self.actor_critic
        t_enqueue = time.perf_counter()
        self._grad_arena()
        sums = torch.zeros(4, device=self.device)
        last_est = last_swap = None
        adaptive = self.desired_kl is not None and self.schedule == "adaptive"
        for batch in self.storage.mini_batch_generator(self.num_mini_batches, self.num_learning_epochs):
            obs = batch[0]
            multi_rank = self.dist_ctx is not None and self.dist_ctx.enabled and self.dist_ctx.world > 1
            two_streams = obs.is_cuda and self._lr_t is not None and self._two_streams(ac.critic, obs.shape[0], multi_rank)
            f = self._mb_forward(ac, batch, two_streams)
            if (self.dist_ctx is not None and self.dist_ctx.enabled) or two_streams:
                est, swap = self._step_minibatch_data_parallel(ac, obs, f["next_critic_obs"], f["loss"], f["mu"], f["sigma"], f["old_mu"], f["old_sigma"],
                                                               f["kl_mean"], adaptive, est_losses=f["early_est"])
            else:
                if adaptive:
                    self._adapt_lr(f["mu"], f["sigma"], f["old_mu"], f["old_sigma"], f["kl_mean"])
                est, swap = ac.estimator.update(obs, f["next_critic_obs"], lr=None if self._lr_t is not None else self.learning_rate)
                self.optimizer.zero_grad()
                from . import fused_linear as FL
                FL.grad_cycle()
                with FL.deferred_wgrad_reduce():
                    FL.backward_losses(f["loss"])
                if FL._arena is not None:
                    FL._arena.bucket("ppo", [p for p in ac.parameters() if p.grad is not None]).adopt()
                self._clip_and_step(self.optimizer, ac.parameters(), self.max_grad_norm)
            sums += torch.stack((f["value_loss"].detach(), f["surrogate_loss"].detach(), est, swap))
            last_est, last_swap = est, swap
        n = self.num_learning_epochs * self.num_mini_batches
        # host time to ENQUEUE the update (no read-back before this point): against the update's wall time it says whether the device or the
        # host's launch rate bounds it (bench line: update_host_enqueue_s)
        self.update_enqueue_s = time.perf_counter() - t_enqueue
        if self._lr_t is not None:
            self.learning_rate = float(self._lr_t)       # one read-back per update (logging, checkpoints)
        sums = (sums / n).tolist()
        self.storage.clear()
        # the reference returns the LAST minibatch's estimator losses in slots 3 and 4 (HIMP:198)
        return sums[0], sums[1], float(last_est), float(last_swap)
