"""Adversarial-motion-prior pieces of the AMP configuration (BASELINE config 4), PyTorch-ROCm.

Restates, with the same public surface and numerics:
  AMPLoader        rsl_rl/datasets/motion_loader.py:15-349  (mocap clips -> 30-dim expert transition pairs)
  Normalizer       rsl_rl/utils/utils.py:78-130             (running mean/var, float64)
  ReplayBuffer     rsl_rl/storage/replay_buffer.py:36-74    (ring buffer of policy transition pairs)
  AMPDiscriminator rsl_rl/algorithms/amp_discriminator.py:9-72 (LSGAN discriminator, style reward, gradient penalty)
Differences in mechanism (not in results): only the 30 feature columns of each pre-sampled frame are kept (the reference
keeps all 49 and slices per minibatch, ML:315-330), blending is one vectorised gather per clip set, and the normaliser
moments are accumulated on the device in float64 instead of round-tripping every minibatch through host numpy (HYBP:279-281).
Host RNG draws (np.random) are made in the reference's order so that a seeded run samples the same transitions.
"""
import json
import os

import numpy as np
import torch
import torch.nn as nn
from torch import autograd

# 61-column mocap frame layout (ML:17-48)
POS, ROT, JP, TOE, LV, AV, JV, TOEV = (0, 3), (3, 7), (7, 19), (19, 31), (31, 34), (34, 37), (37, 49), (49, 61)
FEATURE_COLS = list(range(*JP)) + list(range(LV[0], JV[1]))      # joint pos (12) + base lin/ang vel (6) + joint vel (12) = 30


def _reorder_pybullet_to_isaac(frames):
    """Leg order FR,FL,RR,RL -> FL,FR,RL,RR for the four per-leg blocks (ML:134-164)."""
    out = frames.copy()
    for lo, hi in (JP, TOE, JV, TOEV):
        fr, fl, rr, rl = np.split(frames[:, lo:hi], 4, axis=1)
        out[:, lo:hi] = np.hstack([fl, fr, rl, rr])
    return out


def load_clip(path):
    with open(path, "r") as f:
        j = json.load(f)
    return dict(frames=np.array(j["Frames"], dtype=np.float64), weight=float(j["MotionWeight"]), frame_duration=float(j["FrameDuration"]))


def load_clip_bundle(npz_path):
    """Clips stored as one .npz (frames_<i>, weight_<i>, frame_duration_<i>): the data fixture shipped with the tests."""
    z = np.load(npz_path)
    n = int(z["num_clips"])
    return [dict(frames=z[f"frames_{i}"].astype(np.float64), weight=float(z[f"weight_{i}"]), frame_duration=float(z[f"frame_duration_{i}"])) for i in range(n)]


class _IndexUploader:
    """Host-drawn index vectors (np.random.choice: the reference's draws, ML:329, RB:72 -- the numpy stream is part of the pinned behaviour) to the
    device WITHOUT draining the pipeline.  torch.from_numpy(idx).to(device) is a pageable host-to-device copy: the call returns when the stream has
    reached and finished it, i.e. the host waits for everything enqueued before -- twice per minibatch the update ran in lockstep with the host
    (round 6: HybridPPO.update took 0.272 s to ENQUEUE of 0.284 s wall, and every hiccup of a shared host went straight into the line: 1.25-1.35 M
    env-steps/s between leases).  Here the indices go through page-locked staging buffers with an asynchronous copy; a buffer is reused only after
    the event behind its last copy has completed (64 slots = more than the 40 draws of one update: the wait never stalls in practice)."""

    def __init__(self, slots=64):
        self._slots = [None] * slots
        self._next = 0

    def __call__(self, idx, device):
        t = torch.from_numpy(idx)
        if torch.device(device).type != "cuda":
            return t.to(device)
        i = self._next
        self._next = (i + 1) % len(self._slots)
        slot = self._slots[i]
        if slot is None or slot[0].numel() < t.numel() or slot[0].dtype != t.dtype:
            try:
                buf = torch.empty(t.numel(), dtype=t.dtype).pin_memory()
            except RuntimeError:               # no page-locked memory to be had (memlock limit): the pageable copy still works, it only drains the pipeline
                return t.to(device)
            slot = [buf, None, buf.numpy()]
            self._slots[i] = slot
        buf, ev, host = slot
        if ev is not None:
            ev.synchronize()                   # the copy that last read this buffer is done (normally long ago)
        # numpy, not Tensor.copy_: a CPU tensor copy of this size opens an OpenMP region over every hardware thread the container SHOWS, and under a
        # cgroup CPU quota (the GPU box: 16 of 256) their spinning gets the process throttled -- measured 23 ms per call, the update 3.4 x slower
        np.copyto(host[:idx.shape[0]], idx)
        out = buf[:t.numel()].to(device, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(out.device))
        slot[1] = ev
        return out


_upload_indices = _IndexUploader()


class AMPLoader:
    def __init__(self, device, time_between_frames, data_dir="", preload_transitions=False, num_preload_transitions=1000000,
                 motion_files=(), clips=None):
        self.device = device
        self.time_between_frames = time_between_frames
        if clips is None:
            clips = []
            for mf in motion_files:
                if str(mf).endswith(".npz"):
                    clips += load_clip_bundle(mf)
                else:
                    clips.append(load_clip(mf))
        if not clips:
            raise ValueError("AMPLoader needs at least one motion clip")
        self.trajectories, weights, fdur, lens, nfr = [], [], [], [], []
        for c in clips:
            data = _reorder_pybullet_to_isaac(c["frames"])
            self.trajectories.append(torch.tensor(data[:, FEATURE_COLS], dtype=torch.float32, device=device))
            weights.append(c["weight"]); fdur.append(c["frame_duration"])
            lens.append((data.shape[0] - 1) * c["frame_duration"]); nfr.append(float(data.shape[0]))
        self.trajectory_idxs = list(range(len(clips)))
        self.trajectory_weights = np.array(weights) / np.sum(weights)
        self.trajectory_frame_durations, self.trajectory_lens, self.trajectory_num_frames = np.array(fdur), np.array(lens), np.array(nfr)
        self.preload_transitions = preload_transitions
        if preload_transitions:
            idxs = self.weighted_traj_idx_sample_batch(num_preload_transitions)
            times = self.traj_time_sample_batch(idxs)
            self.preloaded_s = self.get_frame_at_time_batch(idxs, times)
            self.preloaded_s_next = self.get_frame_at_time_batch(idxs, times + self.time_between_frames)

    @property
    def observation_dim(self):
        return 30

    @property
    def num_motions(self):
        return len(self.trajectories)

    def weighted_traj_idx_sample_batch(self, size):          # ML:171-175
        return np.random.choice(self.trajectory_idxs, size=size, p=self.trajectory_weights, replace=True)

    def traj_time_sample_batch(self, traj_idxs):             # ML:183-187
        subst = self.time_between_frames + self.trajectory_frame_durations[traj_idxs]
        t = self.trajectory_lens[traj_idxs] * np.random.uniform(size=len(traj_idxs)) - subst
        return np.maximum(np.zeros_like(t), t)

    def get_frame_at_time_batch(self, traj_idxs, times):     # ML:231-255 (feature columns only)
        p = times / self.trajectory_lens[traj_idxs]
        n = self.trajectory_num_frames[traj_idxs]
        lo, hi = np.floor(p * n).astype(np.int64), np.ceil(p * n).astype(np.int64)
        start = torch.zeros(len(traj_idxs), 30, device=self.device)
        end = torch.zeros(len(traj_idxs), 30, device=self.device)
        for k in np.unique(traj_idxs):
            m = torch.from_numpy(traj_idxs == k).to(self.device)
            sel = traj_idxs == k
            traj = self.trajectories[k]
            start[m] = traj[torch.from_numpy(lo[sel]).to(self.device)]
            end[m] = traj[torch.from_numpy(hi[sel]).to(self.device)]
        blend = torch.tensor(p * n - lo, device=self.device, dtype=torch.float32).unsqueeze(-1)
        return (1.0 - blend) * start + blend * end

    def feed_forward_generator(self, num_mini_batch, mini_batch_size):   # ML:315-343
        for _ in range(num_mini_batch):
            if self.preload_transitions:
                idxs = _upload_indices(np.random.choice(self.preloaded_s.shape[0], size=mini_batch_size), self.device)
                yield self.preloaded_s[idxs], self.preloaded_s_next[idxs]
            else:
                ti = self.weighted_traj_idx_sample_batch(mini_batch_size)
                t = self.traj_time_sample_batch(ti)
                yield self.get_frame_at_time_batch(ti, t), self.get_frame_at_time_batch(ti, t + self.time_between_frames)


def _fused_update_wanted():
    """LSIM_AMP_FUSED_UPDATE=0: the discriminator update's torch statements (A/B switch)"""
    return os.environ.get("LSIM_AMP_FUSED_UPDATE") != "0"


class Normalizer:
    """Running mean / variance (parallel algorithm), float64, clip +-clip_obs (UT:78-130)."""

    def __init__(self, input_dim, epsilon=1e-4, clip_obs=10.0, device="cpu"):
        self._mean = torch.zeros(input_dim, dtype=torch.float64, device=device)
        self._var = torch.ones(input_dim, dtype=torch.float64, device=device)
        self._count = torch.tensor(1e-4, dtype=torch.float64, device=device)   # RunningMeanStd default epsilon (UT:79)
        self.epsilon, self.clip_obs = epsilon, clip_obs
        self.moment_sync = None     # data-parallel hook: callable(sum, sumsq, count) -> global triple

    mean = property(lambda self: self._mean.cpu().numpy())
    var = property(lambda self: self._var.cpu().numpy())
    count = property(lambda self: float(self._count.reshape(-1)[0]))

    def to(self, device):
        self._mean, self._var, self._count = self._mean.to(device), self._var.to(device), self._count.to(device)
        return self

    def normalize_torch(self, x, device=None):
        mean = self._mean.to(dtype=torch.float32, device=x.device)
        std = torch.sqrt((self._var + self.epsilon).to(dtype=torch.float32, device=x.device))
        return torch.clamp((x - mean) / std, -self.clip_obs, self.clip_obs)

    def normalize(self, x):
        return np.clip((x - self.mean) / np.sqrt(self.var + self.epsilon), -self.clip_obs, self.clip_obs)

    def update(self, arr):
        a = torch.as_tensor(arr).to(device=self._mean.device)
        if self.moment_sync is None and a.is_cuda and a.dim() == 2 and a.dtype == torch.float32 and a.stride(1) == 1 and _fused_update_wanted():
            # lsim_running_moments_update: two launches, the running state updated IN PLACE (no host round trip: the torch statements below
            # start with a pageable host -> device copy of the row count, i.e. one pipeline drain per call, twice per minibatch)
            import ctypes
            from .. import abi, lib
            L = lib.load()
            if getattr(self, "_ws", None) is None or self._ws.device != a.device:
                need = ctypes.c_size_t()
                lib.check(L.lsim_running_moments_workspace(ctypes.byref(need)), what="lsim_running_moments_workspace")
                self._ws = torch.empty(need.value // 8, dtype=torch.float64, device=a.device)
            if self._count.dim() == 0:
                self._count = self._count.reshape(1)
            rc = L.lsim_running_moments_update(a.data_ptr(), a.stride(0), a.shape[0], a.shape[1], self._mean.data_ptr(), self._var.data_ptr(),
                                               self._count.data_ptr(), self._ws.data_ptr(), self._ws.numel() * 8, torch.cuda.current_stream(a.device).cuda_stream)
            if rc == 0:
                return
            if rc != abi.E_UNSUPPORTED:
                lib.check(rc, what="lsim_running_moments_update")
        n = torch.tensor(float(a.shape[0]), dtype=torch.float64, device=a.device)
        if self.moment_sync is None:
            # the reference feeds float32 arrays to np.mean / np.var (HYBP:280-281): batch moments are formed in the input
            # precision, only the running state is float64
            bmean, bvar = a.mean(dim=0).to(torch.float64), a.var(dim=0, unbiased=False).to(torch.float64)
        else:
            a = a.to(torch.float64)
            s1, s2, n = self.moment_sync(a.sum(dim=0), (a * a).sum(dim=0), n)
            bmean = s1 / n
            bvar = s2 / n - bmean * bmean
        delta = bmean - self._mean
        tot = self._count + n
        new_mean = self._mean + delta * n / tot
        m2 = self._var * self._count + bvar * n + torch.square(delta) * self._count * n / (self._count + n)
        self._mean, self._var, self._count = new_mean, m2 / (self._count + n), n + self._count


class ReplayBuffer:
    def __init__(self, obs_dim, buffer_size, device):
        self.states = torch.zeros(buffer_size, obs_dim, device=device)
        self.next_states = torch.zeros(buffer_size, obs_dim, device=device)
        self.buffer_size, self.device = buffer_size, device
        self.step = 0
        self.num_samples = 0

    def insert(self, states, next_states):                    # RB:52-68
        n = states.shape[0]
        end = self.step + n
        if end > self.buffer_size:
            head = self.buffer_size - self.step
            self.states[self.step:] = states[:head]; self.next_states[self.step:] = next_states[:head]
            self.states[:end - self.buffer_size] = states[head:]; self.next_states[:end - self.buffer_size] = next_states[head:]
        else:
            self.states[self.step:end] = states; self.next_states[self.step:end] = next_states
        self.num_samples = min(self.buffer_size, max(end, self.num_samples))
        self.step = (self.step + n) % self.buffer_size

    def reserve(self, n):
        """the bookkeeping of insert() for n rows written by someone else (lsim_amp_step fills ring rows (cursor + i) % size): returns the cursor"""
        if n > self.buffer_size:
            raise ValueError("more rows than the ring holds")
        cur = self.step
        self.num_samples = min(self.buffer_size, max(cur + n, self.num_samples))
        self.step = (cur + n) % self.buffer_size
        return cur

    def feed_forward_generator(self, num_mini_batch, mini_batch_size):   # RB:70-74
        for _ in range(num_mini_batch):
            idx = _upload_indices(np.random.choice(self.num_samples, size=mini_batch_size), self.device)
            yield self.states[idx], self.next_states[idx]


def _linear_relu(bias, x, weight):
    """relu(x W^T + b): bias + ReLU in the GEMM epilogue through the PRIVATE torch._addmm_activation where this torch has it, else the two
    public statements (same values; ADVICE r3: a torch without the private op must keep working)"""
    if hasattr(torch, "_addmm_activation"):
        return torch._addmm_activation(bias, x, weight.t(), use_gelu=False)
    return torch.addmm(bias, x, weight.t()).relu_()


class _GradPenFn(autograd.Function):
    """lambda * mean_b || dD/dx (x_b) ||^2 for D = w3 . relu(W2 relu(W1 x + b1) + b2) + b3 (DISC:36-53), forward and backward in closed form.

    dD/dx = W1^T (m1 * (W2^T (m2 * w3))) with m1, m2 the ReLU masks; relu'' = 0, so the penalty's gradient reaches W1, W2 and w3 only
    through these products.  autograd's double backward gets the same numbers but also pushes the (identically zero) mask gradients back
    through both layers -- 8 large GEMMs here instead of ~14, no retained graph.  Used on the GPU; the CPU path keeps the reference's autograd
    statement and tests/test_gpu_learner_golden.py compares the two through the reference fixture."""

    @staticmethod
    def forward(ctx, x, W1, b1, W2, b2, w3, lambda_):
        # every mask is applied by aten's threshold_backward(grad, pre, 0) = grad where pre > 0 else 0 -- the kernel relu's own backward runs --
        # on the fp32 activation itself: one pass per mask instead of compare + cast / not + multiply / fill
        tb = torch.ops.aten.threshold_backward
        a1 = _linear_relu(b1, x, W1)                       # relu(z1) straight from the GEMM epilogue; z1 > 0  <=>  a1 > 0
        z2 = _linear_relu(b2, a1, W2)                      # relu(z2): only its sign pattern is used below (z2 > 0 <=> relu(z2) > 0)
        u2 = tb(w3.expand_as(z2), z2, 0.0)                  # (B, H2): m2 * w3
        u1 = tb(u2 @ W2, a1, 0.0)                           # (B, H1): m1 * (W2^T u2)
        g = u1 @ W1                                         # (B, D): dD/dx
        ctx.save_for_backward(W1, W2, a1, z2, u2, u1, g)
        ctx.scale = 2.0 * lambda_ / x.shape[0]
        return lambda_ * g.pow(2).sum(dim=1).mean()

    @staticmethod
    def backward(ctx, grad_out):
        tb = torch.ops.aten.threshold_backward
        W1, W2, a1, z2, u2, u1, g = ctx.saved_tensors
        dg = g * (ctx.scale * grad_out)                     # d penalty / d g
        B, n1 = a1.shape
        n2 = z2.shape[1]
        fused = _fused_update_wanted() and B >= 16384 and n1 % 4 == 0 and dg.shape[1] % 4 == 0
        if fused:
            # lsim_linear_wgrad: u1^T dg with K = B as MFMA tiles (was a split-K BLAS GEMM + its post-sum); lsim_linear_masked_forward: dg W1^T with the
            # mask of a1 applied to the accumulators (was a GEMM + a mask pass over (B, H1)); lsim_masked_colsum: mask of z2 + column sum in one pass
            import ctypes
            from .. import abi, lib
            from . import fused_linear as FL
            L = lib.load()
            st = torch.cuda.current_stream(a1.device).cuda_stream
            dW1 = FL.linear_wgrad(dg, u1, want_bias=False)[0] if FL._eligible(B, dg.shape[1], n1) else u1.t() @ dg
            du1 = torch.empty_like(a1)
            rc = L.lsim_linear_masked_forward(dg.data_ptr(), dg.stride(0), W1.data_ptr(), a1.data_ptr(), a1.stride(0), B, dg.shape[1], n1, du1.data_ptr(), du1.stride(0), st)
            if rc == abi.E_UNSUPPORTED:
                du1 = tb(dg @ W1.t(), a1, 0.0)
            else:
                lib.check(rc, what="lsim_linear_masked_forward")
            dW2 = u2.t() @ du1
            t = du1 @ W2.t()
            ws = _cols_ws(B, n2, a1.device)
            dw3 = torch.empty(1, n2, device=a1.device)
            if ws is None or L.lsim_masked_colsum(t.data_ptr(), t.stride(0), z2.data_ptr(), z2.stride(0), B, n2, dw3.data_ptr(), ws.data_ptr(), ws.numel(), st) != 0:
                dw3 = tb(t, z2, 0.0).sum(dim=0, keepdim=True)
            return None, dW1, None, dW2, None, dw3, None
        dW1 = u1.t() @ dg                                   # g = u1 W1
        du1 = tb(dg @ W1.t(), a1, 0.0)                      # through the mask m1 (a constant)
        dW2 = u2.t() @ du1                                  # u1 = m1 * (u2 W2)
        dw3 = tb(du1 @ W2.t(), z2, 0.0).sum(dim=0, keepdim=True)   # u2 = m2 * w3
        return None, dW1, None, dW2, None, dw3, None


def _cols_ws(batch, n, device):
    import ctypes
    from .. import lib
    from . import fused_linear as FL
    need = ctypes.c_size_t()
    if lib.load().lsim_relu_cols_workspace(batch, n, ctypes.byref(need)) != 0:
        return None
    return FL._workspace("relu_cols", device, need.value)


def _relu_wgrad(x, g, a, weight, bias):
    """(dW, db) of relu(x W^T + b) from the gradient g of its OUTPUT a: the mask is applied to g as the MFMA operand is formed (lsim_linear_relu_wgrad);
    the masked gradient is never written.  Falls back to the torch statements for shapes the library leaves to BLAS."""
    import ctypes
    from .. import lib
    from . import fused_linear as FL
    L = lib.load()
    batch, k_in = x.shape
    n_out = weight.shape[0]
    need, parts = ctypes.c_size_t(), ctypes.c_int()
    if FL._eligible_fused_elu(batch, k_in, n_out) and L.lsim_linear_wgrad_workspace(batch, k_in, n_out, ctypes.byref(need), ctypes.byref(parts)) == 0:
        # plain output tensors, summed at once: the discriminator's parameters receive a second contribution from the gradient penalty in the same
        # backward pass, so their gradient-arena slices (whose deferred sums assume a single contribution, fused_linear._wgrad_call) are not used here
        ws = FL._workspace("wgrad_relu", x.device, need.value, floor=1 << 20)
        dw, db = torch.empty(n_out, k_in, device=x.device), torch.empty(n_out, device=x.device)
        lib.check(L.lsim_linear_relu_wgrad(x.data_ptr(), x.stride(0), g.data_ptr(), g.stride(0), a.data_ptr(), a.stride(0), batch, k_in, n_out, dw.data_ptr(),
                                           db.data_ptr(), None, ws.data_ptr(), ws.numel(), torch.cuda.current_stream(x.device).cuda_stream),
                  what="lsim_linear_relu_wgrad")
        return dw, db
    gy = torch.ops.aten.threshold_backward(g, a, 0.0)
    return gy.t() @ x, gy.sum(dim=0)


def _fused_disc_ok(x, trunk, head):
    if not (x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and x.shape[0] >= 16384 and _fused_update_wanted() and len(trunk) == 4):
        return False
    if not (isinstance(trunk[0], nn.Linear) and isinstance(trunk[1], nn.ReLU) and isinstance(trunk[2], nn.Linear) and isinstance(trunk[3], nn.ReLU)):
        return False
    n2 = trunk[2].out_features
    return (trunk[0].bias is not None and trunk[2].bias is not None and head.bias is not None and head.out_features == 1 and n2 % 4 == 0 and n2 <= 1024
            and 256 % (n2 // 4) == 0 and trunk[0].out_features % 4 == 0)


class _LsganFn(autograd.Function):
    """sum_i mean_b (D(x_i)[b] - target_i)^2 for row blocks x_i stacked in x (HYBP:258-261: expert block with target +1, policy block with -1), D = the
    two-layer ReLU trunk + linear head; forward and backward in closed form.  Backward is five GEMMs (hipBLASLt) and three passes of the build's kernels:
      lsim_relu_head_backward   gd x w3 masked by a2 -> g2, with db2, d w3 and d b3 summed in the same pass (was an outer-product "GEMM", a mask pass, a GEMV and two sums)
      lsim_linear_relu_wgrad    dW1, db1 from (g2 W2, a1, x): the mask of a1 applied as the MFMA operand is formed (was a mask pass, a split-K GEMM and a column sum)
    Returns (the loss, mean D per block [detached])."""
    _targets = {}

    @staticmethod
    def forward(ctx, x, W1, b1, W2, b2, w3, b3, targets, block):
        a1 = _linear_relu(b1, x, W1)
        a2 = _linear_relu(b2, a1, W2)
        d = torch.addmm(b3, a2, w3.t()).view(-1)            # (the GEMM path: rocBLAS's gemv reads the 420 MB at 2 TB/s)
        nb = x.shape[0] // block
        key = (x.device, tuple(targets), block)
        t = _LsganFn._targets.get(key)
        if t is None:                                       # built once with fills (a host list -> device tensor would be a pageable copy: one pipeline drain per minibatch)
            t = torch.empty(nb, block, device=x.device)
            for i, v in enumerate(targets):
                t[i].fill_(float(v))
            t = _LsganFn._targets[key] = t.view(-1)
        err = d - t
        ctx.save_for_backward(x, W1, b1, W2, b2, w3, a1, a2, err)
        ctx.block = block
        means = d.view(nb, block).mean(dim=1)
        ctx.mark_non_differentiable(means)
        return err.pow(2).sum() / block, means

    @staticmethod
    def backward(ctx, g_loss, _g_means):
        import ctypes
        from .. import lib
        x, W1, b1, W2, b2, w3, a1, a2, err = ctx.saved_tensors
        L = lib.load()
        B, n2 = a2.shape
        gd = err * (g_loss * (2.0 / ctx.block))
        g2 = torch.empty_like(a2)
        db2 = torch.empty(n2, device=x.device)
        dhead = torch.empty(n2 + 4, device=x.device)
        ws = _cols_ws(B, n2, x.device)
        lib.check(L.lsim_relu_head_backward(a2.data_ptr(), a2.stride(0), gd.data_ptr(), w3.data_ptr(), B, n2, g2.data_ptr(), db2.data_ptr(), dhead.data_ptr(),
                                            ws.data_ptr(), ws.numel(), torch.cuda.current_stream(x.device).cuda_stream), what="lsim_relu_head_backward")
        # block by block: K = one minibatch is the shape the shipped GEMM table is tuned for (the stacked K ran an untuned kernel at 79 TFLOP/s)
        blk = ctx.block
        dW2 = g2[:blk].t() @ a1[:blk]
        for i in range(blk, B, blk):
            dW2.addmm_(g2[i:i + blk].t(), a1[i:i + blk])
        g1 = g2 @ W2                                        # gradient of a1 (the mask of a1 is applied inside the weight-gradient kernel)
        dW1, db1 = _relu_wgrad(x, g1, a1, W1, b1)
        return None, dW1, db1, dW2, db2, dhead[:n2].view(1, n2), dhead[n2:n2 + 1], None, None


class _LinearReluFn(autograd.Function):
    """relu(x W^T + b) with the bias + ReLU in the GEMM's epilogue (torch._addmm_activation -> hipBLASLt RELU_BIAS): the separate relu pass
    over [B, N] disappears from the forward (102 400 x 1024: 418 -> 254 us, x 512: 915 -> 806 us, tools/relu_epilogue_probe.py).  The op has no
    autograd derivative, hence this Function; backward = the statements autograd would run (mask from the saved OUTPUT, three GEMMs / sums)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        z = _linear_relu(bias, x, weight)
        ctx.save_for_backward(x, weight, z)
        return z

    @staticmethod
    def backward(ctx, g):
        x, weight, z = ctx.saved_tensors
        gy = torch.ops.aten.threshold_backward(g, z, 0.0)      # relu's own backward kernel on the saved OUTPUT (z > 0 <=> pre-activation > 0)
        gx = gy @ weight if ctx.needs_input_grad[0] else None
        gw = gy.t() @ x if ctx.needs_input_grad[1] else None   # a frozen discriminator pays for neither GEMM nor column sum
        gb = gy.sum(dim=0) if ctx.needs_input_grad[2] else None
        return gx, gw, gb


def _trunk_fused(trunk, x):
    """the discriminator trunk (Linear, ReLU, Linear, ReLU, ...) through _LinearReluFn; None when the layout / device does not qualify"""
    # only for the update's tall minibatches: at the rollout's 4096 rows the TunableOp-selected plain GEMM + relu is faster than the
    # default-heuristic epilogue GEMM (collection 0.034 -> 0.037 s per iteration when it was used there too)
    if (not (x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and len(trunk) % 2 == 0) or x.shape[0] < 16384
            or os.environ.get("LSIM_AMP_RELU_EPILOGUE") == "0"):
        return None
    mods = list(trunk)
    for i in range(0, len(mods), 2):
        if not (isinstance(mods[i], nn.Linear) and mods[i].bias is not None and isinstance(mods[i + 1], nn.ReLU)):
            return None
    for i in range(0, len(mods), 2):
        x = _LinearReluFn.apply(x, mods[i].weight, mods[i].bias)
    return x


class AMPDiscriminator(nn.Module):
    def __init__(self, input_dim, amp_reward_coef, hidden_layer_sizes, device, task_reward_lerp=0.0):
        super().__init__()
        self.device, self.input_dim, self.amp_reward_coef, self.task_reward_lerp = device, input_dim, amp_reward_coef, task_reward_lerp
        layers, d = [], input_dim
        for h in hidden_layer_sizes:
            layers += [nn.Linear(d, h), nn.ReLU()]
            d = h
        self.trunk = nn.Sequential(*layers).to(device)
        self.amp_linear = nn.Linear(hidden_layer_sizes[-1], 1).to(device)
        self.trunk.train(); self.amp_linear.train()

    def forward(self, x):
        h = _trunk_fused(self.trunk, x)
        return self.amp_linear(h if h is not None else self.trunk(x))

    def pair_inputs(self, exp_s, exp_ns, pol_s, pol_ns, normalizer):
        """(expert_in, policy_in, expert_raw, expert_state_n, policy_state_n): the [B, 2 D] rows the update feeds the discriminator -- normalised (state,
        next state) pairs of the expert and the policy block (HYBP:247-251 + DISC:57), the un-normalised expert pair of the gradient penalty (HYBP:262-263,
        DISC:37) -- and the normalised states by themselves ([B, D]; views of the rows on the GPU), which the reference feeds its normaliser (HYBP:279-281).  On the GPU three
        launches of lsim_amp_pair_rows (normalise + cat in one pass; the two normalised blocks land in ONE [2 B, 2 D] buffer, so lsgan_loss needs no
        second cat) instead of 28 elementwise launches and three cats; the reference's statements otherwise."""
        B, D = exp_s.shape
        fused = (_fused_update_wanted() and exp_s.is_cuda and exp_s.dtype == torch.float32 and pol_s.shape == exp_s.shape and B >= 16384
                 and all(t.dim() == 2 and t.stride(1) == 1 and t.dtype == torch.float32 for t in (exp_s, exp_ns, pol_s, pol_ns))
                 and (normalizer is None or (normalizer._mean.is_cuda and normalizer._mean.dtype == torch.float64)))
        if not fused:
            raw = torch.cat([exp_s, exp_ns], dim=-1)
            if normalizer is not None:
                with torch.no_grad():
                    nz = normalizer.normalize_torch
                    exp_s, exp_ns, pol_s, pol_ns = nz(exp_s, self.device), nz(exp_ns, self.device), nz(pol_s, self.device), nz(pol_ns, self.device)
            return torch.cat([exp_s, exp_ns], dim=-1), torch.cat([pol_s, pol_ns], dim=-1), raw, exp_s, pol_s
        from .. import lib
        L = lib.load()
        st = torch.cuda.current_stream(exp_s.device).cuda_stream
        stacked = torch.empty(2 * B, 2 * D, device=exp_s.device)
        raw = torch.empty(B, 2 * D, device=exp_s.device)
        mean = normalizer._mean.data_ptr() if normalizer is not None else None
        var = normalizer._var.data_ptr() if normalizer is not None else None
        eps, clip = (float(normalizer.epsilon), float(normalizer.clip_obs)) if normalizer is not None else (0.0, 0.0)
        for s_, ns_, out, m, v in ((exp_s, exp_ns, stacked[:B], mean, var), (pol_s, pol_ns, stacked[B:], mean, var), (exp_s, exp_ns, raw, None, None)):
            lib.check(L.lsim_amp_pair_rows(s_.data_ptr(), s_.stride(0), ns_.data_ptr(), ns_.stride(0), m, v, eps, clip, B, D, out.data_ptr(), out.stride(0), st),
                      what="lsim_amp_pair_rows")
        return stacked[:B], stacked[B:], raw, stacked[:B, :D], stacked[B:, :D]

    def lsgan_loss(self, expert_in, policy_in):
        """0.5 * (mse(D(expert), 1) + mse(D(policy), -1)) (HYBP:258-261) -> (loss, mean D(policy), mean D(expert)); both blocks through ONE closed-form
        forward / backward (_LsganFn) on the GPU, else the reference's statements"""
        if expert_in.shape == policy_in.shape and _fused_disc_ok(expert_in, self.trunk, self.amp_linear):
            l1, l2, h = self.trunk[0], self.trunk[2], self.amp_linear
            B = expert_in.shape[0]
            if (expert_in.is_contiguous() and policy_in.is_contiguous() and expert_in.untyped_storage().data_ptr() == policy_in.untyped_storage().data_ptr()
                    and policy_in.data_ptr() == expert_in.data_ptr() + expert_in.numel() * 4 and expert_in.storage_offset() == 0
                    and expert_in.untyped_storage().nbytes() >= 2 * expert_in.numel() * 4):
                both = torch.as_strided(expert_in, (2 * B, expert_in.shape[1]), (expert_in.shape[1], 1))     # pair_inputs' stacked buffer: no second cat
            else:
                both = torch.cat([expert_in, policy_in], dim=0)
            loss, means = _LsganFn.apply(both, l1.weight, l1.bias, l2.weight, l2.bias, h.weight, h.bias, (1.0, -1.0), B)
            return 0.5 * loss, means[1], means[0]
        policy_d, expert_d = self(policy_in), self(expert_in)
        loss = 0.5 * (torch.nn.functional.mse_loss(expert_d, torch.ones_like(expert_d)) + torch.nn.functional.mse_loss(policy_d, -torch.ones_like(policy_d)))
        return loss, policy_d.mean().detach(), expert_d.mean().detach()

    def compute_grad_pen(self, expert_state, expert_next_state, lambda_=10, pair=None):   # DISC:36-53; pair: the two already concatenated (pair_inputs)
        data = pair if pair is not None else torch.cat([expert_state, expert_next_state], dim=-1)
        if (data.is_cuda and len(self.trunk) == 4 and isinstance(self.trunk[0], nn.Linear) and isinstance(self.trunk[1], nn.ReLU)
                and isinstance(self.trunk[2], nn.Linear) and isinstance(self.trunk[3], nn.ReLU)):
            l1, l2 = self.trunk[0], self.trunk[2]
            return _GradPenFn.apply(data, l1.weight, l1.bias, l2.weight, l2.bias, self.amp_linear.weight, float(lambda_))
        data.requires_grad = True
        disc = self.amp_linear(self.trunk(data))
        grad = autograd.grad(outputs=disc, inputs=data, grad_outputs=torch.ones(disc.size(), device=disc.device),
                             create_graph=True, retain_graph=True, only_inputs=True)[0]
        return lambda_ * (grad.norm(2, dim=1) - 0).pow(2).mean()

    def predict_amp_reward(self, state, next_state, task_reward, normalizer=None):   # DISC:55-72
        with torch.no_grad():
            self.eval()
            if normalizer is not None:
                state, next_state = normalizer.normalize_torch(state, self.device), normalizer.normalize_torch(next_state, self.device)
            d = self.forward(torch.cat([state, next_state], dim=-1))
            reward = self.amp_reward_coef * torch.clamp(1 - (1 / 4) * torch.square(d - 1), min=0)
            if self.task_reward_lerp > 0:
                reward = (1.0 - self.task_reward_lerp) * reward + self.task_reward_lerp * task_reward.unsqueeze(-1)
            self.train()
        return reward.squeeze(), d


def default_motion_files():
    """The Aliengo clip bundle shipped as a data fixture (7 clips selected by AGA:34-36)."""
    here = os.path.dirname(os.path.abspath(__file__))
    return [os.path.join(os.path.dirname(here), "data", "mocap_aliengo.npz")]
