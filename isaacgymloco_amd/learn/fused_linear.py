"""nn.Linear whose backward uses the library's tall-skinny weight-gradient kernel (include/lsim.h, lsim_linear_wgrad) for the
learner's narrow layers on the GPU.

Forward and grad_input are the usual BLAS calls.  grad_weight = g^T x and grad_bias = g.sum(0) over a 102 400-row minibatch are
K = 102 400 reductions into a small output; rocBLAS/hipBLASLt run them at 1-55 % of the fp32 MFMA peak (and torch's column sum
needs 260 us for 19 columns); the MFMA kernels in csrc/ls_learn.h read x and g once per 64 x 128 output tile.  Same parameters,
same state_dict keys, same initialisation as nn.Linear (it IS an nn.Linear); results differ from BLAS by fp32 summation order.
"""
import ctypes
import os
import weakref

import torch
import torch.nn as nn
import torch.nn.functional as F

_MIN_BATCH = 4096
_workspaces = {}
_plans = {}
_adam_tables = {}


def _workspace(kind, device, nbytes, floor=0):
    """scratch buffer of one kernel family, PER STREAM: the update runs the actor / estimator chain and the critic chain on two streams
    (him_ppo.py), and two weight-gradient kernels in flight at once must not share their partial-result buffers"""
    key = (kind, device, torch.cuda.current_stream(device).cuda_stream)
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(int(nbytes), int(floor)), dtype=torch.uint8, device=device)
        _workspaces[key] = ws
    return ws


class GradArena:
    """Persistent flat fp32 gradient buffers ("buckets") whose slices ARE the parameters' gradients.

    The weight-gradient kernels of this module write a layer's dW / db straight into the parameter's slice and hand that view to autograd,
    which adopts it as `.grad` without a copy (a leaf without a gradient takes ownership of the incoming tensor).  Data-parallel training
    then all-reduces the bucket in place -- no torch.cat of ~40 tensors per bucket and minibatch, no re-viewing afterwards --
    and the gradient pointers of the fused clip + Adam step stay the same from minibatch to minibatch (its pointer tables are cached).
    A slice is handed out at most once per cycle (new_cycle(): call it wherever the gradients are reset) and only while the parameter has no
    gradient: a second contribution within the cycle goes through an ordinary tensor and autograd's accumulation, as without the arena.
    Gradients that reach a parameter some other way (autograd's own kernels: the action std, the prototypes, layers the kernels leave to
    BLAS) are copied into their slices by Bucket.adopt()."""

    class Bucket:
        def __init__(self, params, extra, device):
            self.params = list(params)
            self.sizes = [p.numel() for p in self.params]
            self.extra = int(extra)
            self.flat = torch.zeros(sum(self.sizes) + self.extra, dtype=torch.float32, device=device)
            pieces = self.flat.split(self.sizes + ([self.extra] if self.extra else []))
            self.views = [v.view_as(p) for v, p in zip(pieces, self.params)]
            self.extra_view = pieces[-1] if self.extra else None
            self.ptrs = [v.data_ptr() for v in self.views]

        def adopt(self):
            """make every parameter's .grad its slice of the flat buffer (copying gradients that were produced elsewhere); returns False when a
            parameter of the bucket has no gradient (the caller rebuilds the bucket: membership follows `p.grad is not None`)"""
            src, dst = [], []
            for p, v, ptr in zip(self.params, self.views, self.ptrs):
                g = p.grad
                if g is None:
                    return False
                if g.data_ptr() != ptr:
                    src.append(g); dst.append(v)
                    p.grad = v
            if src:
                torch._foreach_copy_(dst, src)
            return True

    def __init__(self):
        self.buckets = {}
        self._slot = {}          # parameter data_ptr -> (parameter, view)
        self._taken = set()

    def bucket(self, key, params, extra=0):
        """the bucket registered under `key` for exactly these parameters (created or re-created on demand)"""
        params = list(params)
        b = self.buckets.get(key)
        if b is None or len(b.params) != len(params) or any(x is not y for x, y in zip(b.params, params)) or b.extra != extra:
            if b is not None:
                for q in b.params:
                    self._slot.pop(q.data_ptr(), None)
            b = GradArena.Bucket(params, extra, params[0].device)
            self.buckets[key] = b
            for q, v in zip(b.params, b.views):
                self._slot[q.data_ptr()] = (q, v)
        return b

    def take(self, param_ptr, shape):
        """the gradient slice of the parameter stored at `param_ptr`, or None (unknown parameter, already handed out in this cycle, the
        parameter already has a gradient, or a shape mismatch)"""
        ent = self._slot.get(param_ptr)
        if ent is None or param_ptr in self._taken:
            return None
        q, v = ent
        if q.grad is not None or tuple(v.shape) != tuple(shape):
            return None
        self._taken.add(param_ptr)
        return v.view(v.shape)       # a FRESH alias: autograd adopts an incoming gradient without a copy only if nobody else holds that tensor object

    def new_cycle(self):
        self._taken.clear()


_arena = None


def set_grad_arena(arena):
    global _arena
    _arena = arena


def grad_cycle():
    """call where gradients are reset (optimizer.zero_grad()): the arena's slices may be handed out again"""
    if _arena is not None:
        _arena.new_cycle()


def _grad_out(param_ptr, shape, device, hit=None):
    """output tensor for a parameter's gradient: its arena slice when there is one to hand out (then hit[0] stays as it is), else a fresh
    tensor (hit[0] = False)"""
    if _arena is not None and param_ptr is not None:
        v = _arena.take(param_ptr, shape)
        if v is not None:
            return v
    if hit is not None:
        hit[0] = False
    return torch.empty(shape, device=device, dtype=torch.float32)


# ---- deferred sums of the weight-gradient partial results: inside `with deferred_wgrad_reduce():` the backward functions below run their
# MFMA kernels as usual but leave the final fixed-order sum of the partial results to ONE launch at the end of the block (15 launches of a few
# microseconds per minibatch become one; same arithmetic, same bits).  Each layer then needs its own partial-result buffer until that launch.
_defer_without_arena = False   # tests only: defer for plain output tensors too (no autograd behind the call)
_pending = None          # None: sums are launched at once; else [record, ...] with the tensors they refer to kept alive
_pending_keep = []
_pending_params = set()


class deferred_wgrad_reduce:
    """LSIM_DEFER_WGRAD_REDUCE=0 makes it a no-op (A/B: every layer sums its partial results at once)"""

    def __enter__(self):
        global _pending
        self._outer = _pending is not None or os.environ.get("LSIM_DEFER_WGRAD_REDUCE", "1") == "0"
        if not self._outer:
            _pending = []
        return self

    def __exit__(self, *exc):
        global _pending
        if not self._outer:
            try:
                flush_wgrad_reduces()
            finally:
                _pending = None
        return False


def flush_wgrad_reduces(mid_backward=False):
    """sum every pending set of partial results (on the current stream: call it where the backward passes that produced them have
    returned -- autograd has then ordered this stream behind the streams their kernels ran on).  mid_backward: called from inside a
    backward node instead; nothing has joined the streams yet, so this stream first waits for every other stream a pending kernel ran on"""
    if not _pending:
        return
    from .. import abi, lib
    arr = (abi.LsimWgradPending * len(_pending))(*_pending)
    dev = _pending_keep[0][0].device
    if mid_backward:
        cur = torch.cuda.current_stream(dev)
        for st in {k[-1] for k in _pending_keep}:
            if st != cur:
                cur.wait_stream(st)
    lib.check(lib.load().lsim_wgrad_reduce_batch(arr, len(_pending), torch.cuda.current_stream(dev).cuda_stream), what="lsim_wgrad_reduce_batch")
    if _arena is not None:       # the sums went to the arena slices: they are the gradients only if autograd adopted those slices
        for wptr in _pending_params:
            ent = _arena._slot.get(wptr)
            if ent is not None and ent[0].grad is not None and ent[0].grad.data_ptr() != ent[1].data_ptr():
                raise RuntimeError("deferred weight-gradient sum: autograd copied a gradient instead of adopting the arena slice")
    _pending.clear(); _pending_keep.clear(); _pending_params.clear()


def _wgrad_call(L, fn_now, fn_deferred, args, weight_ptr, ws, keep, in_arena):
    """run one weight-gradient call, deferring its final sum when a deferred block is open AND the results go to gradient-arena slices: the
    sum is written after autograd has taken the returned tensor as `.grad`, which is only the same memory if autograd adopted it without a copy
    -- the arena hands out a fresh alias for exactly that, and flush_wgrad_reduces() checks it.  (`keep` must not hold dW / db: a second
    owner makes autograd clone the gradient, and the clone would be of memory that holds nothing yet.)  A second contribution to the same
    parameter inside one block first flushes, whichever way it is computed."""
    if _pending is not None and weight_ptr in _pending_params:
        # a second contribution to a parameter whose first one is still a pending sum (a module applied twice inside one block): it cannot
        # be in the arena (take() hands a slice out once per cycle), so autograd will ADD it to the slice in place -- the pending sum must be
        # in the slice before that, not overwrite it afterwards
        flush_wgrad_reduces(mid_backward=True)
    if _pending is None or not in_arena:
        return getattr(L, fn_now)(*args)
    from .. import abi
    rec = abi.LsimWgradPending()
    rc = getattr(L, fn_deferred)(*args, ctypes.byref(rec))
    if rc == 0:
        _pending.append(rec); _pending_params.add(weight_ptr)
        _pending_keep.append((ws,) + tuple(keep) + (torch.cuda.current_stream(ws.device),))       # last entry: the stream the kernel runs on
    return rc


def _eligible(batch, k_in, n_out):
    """the library decides (lsim_linear_wgrad_workspace returns LSIM_E_UNSUPPORTED for shapes it leaves to BLAS)"""
    if batch < _MIN_BATCH:
        return False
    key = (batch, k_in, n_out)
    ok = _plans.get(key)
    if ok is None:
        from .. import lib
        need, parts = ctypes.c_size_t(), ctypes.c_int()
        ok = lib.load().lsim_linear_wgrad_workspace(batch, k_in, n_out, ctypes.byref(need), ctypes.byref(parts)) == 0
        _plans[key] = ok
    return ok


def linear_wgrad(x, g, want_bias=True, weight_ptr=None, bias_ptr=None):
    """(g^T x, g.sum(0)) for 2-D fp32 CUDA tensors through lsim_linear_wgrad; weight_ptr / bias_ptr: data_ptr() of the parameters the results
    are gradients of (GradArena: they are written into the parameters' gradient slices)"""
    from .. import lib
    L = lib.load()
    if x.stride(1) != 1:
        x = x.contiguous()
    if g.stride(1) != 1:
        g = g.contiguous()
    batch, k_in = x.shape
    n_out = g.shape[1]
    need, waves = ctypes.c_size_t(), ctypes.c_int()
    lib.check(L.lsim_linear_wgrad_workspace(batch, k_in, n_out, ctypes.byref(need), ctypes.byref(waves)), what="lsim_linear_wgrad_workspace")
    # inside a deferred block every layer keeps its partial results until the block's one summing launch: a buffer per parameter
    ws = _workspace("wgrad" if _pending is None else ("wgrad", weight_ptr), x.device, need.value, floor=1 << 20 if _pending is None else 0)
    hit = [True]
    dw = _grad_out(weight_ptr, (n_out, k_in), x.device, hit)
    db = _grad_out(bias_ptr, (n_out,), x.device, hit) if want_bias else None
    lib.check(_wgrad_call(L, "lsim_linear_wgrad", "lsim_linear_wgrad_deferred",
                          (x.data_ptr(), x.stride(0), g.data_ptr(), g.stride(0), batch, k_in, n_out, dw.data_ptr(),
                           db.data_ptr() if want_bias else None, ws.data_ptr(), ws.numel(), torch.cuda.current_stream(x.device).cuda_stream),
                          weight_ptr, ws, (x, g), hit[0] or _defer_without_arena), what="lsim_linear_wgrad")
    return dw, db


class _SkinnyLinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        ctx.bias_ptr = bias.data_ptr() if bias is not None else None
        return F.linear(x, weight, bias)

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        gx = g @ weight if ctx.needs_input_grad[0] else None
        dw = db = None
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            dw, db = linear_wgrad(x, g, want_bias=ctx.has_bias, weight_ptr=weight.data_ptr(), bias_ptr=ctx.bias_ptr)
        return gx, dw, db


def _eligible_fused_elu(batch, k_in, n_out):
    """Linear + ELU pairs whose weight gradient runs in the library's tiled kernel (more than 4096 outputs, see ls_learn.h)"""
    return _eligible(batch, k_in, n_out) and k_in * n_out > 4096


def _fused_forward_wanted(x, k_in):
    """Default: the library kernel for the hidden layers whose rows are 16-byte aligned -- k_in % 4 == 0 (64 -> 512, 512 -> 256, 256 -> 128, 128 -> 64) or,
    since round 6, rows an aligned pitch apart (the shuffled 238- / 270-wide observation fields, learn/storage.py: the weights are then zero-padded to the
    pitch's width, see linear_elu_forward).  Unaligned rows stay on BLAS + torch ELU: their 8- and 4-byte loads double and quadruple the kernel's fetch
    instructions, and the fetch is what bounds it (238 -> 512 took 415 us against ~325 for BLAS + ELU in round 5's loop).
    LSIM_ELU_FORWARD=0 / all: A/B switches"""
    mode = os.environ.get("LSIM_ELU_FORWARD", "aligned")
    if mode == "aligned":
        return x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0 and (k_in % 4 == 0 or x.stride(0) >= (k_in + 3) // 4 * 4)
    return mode != "0"


_padded_weights = {}


def _padded_weight(weight, k4):
    """[n_out, k4] copy of `weight` [n_out, k_in] with zero columns k_in .. k4 - 1, refreshed on every call (the weights move with every optimiser step;
    0.5 MB for 238 -> 512).  With it x's columns beyond k_in -- the zero padding of a shuffled observation row, or whatever finite values follow in a wider
    row -- meet zero weights: the product is the unpadded one exactly."""
    key = (weight.data_ptr(), k4, torch.cuda.current_stream(weight.device).cuda_stream)
    wp = _padded_weights.get(key)
    if wp is None or wp.shape[0] != weight.shape[0]:
        wp = _padded_weights[key] = torch.zeros(weight.shape[0], k4, device=weight.device, dtype=torch.float32)
    wp[:, :weight.shape[1]].copy_(weight.detach())
    return wp


def linear_elu_forward(x, weight, bias):
    """elu(x W^T + b): lsim_linear_elu_forward (fp32 MFMA, the activation applied to the accumulators, one write of the output) where it is
    the faster form, else BLAS + torch's elementwise ELU"""
    # (x wider than k_in would be silently truncated by the kernel where F.linear raises: shapes, bias dtype and layout are checked here)
    if (x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and weight.dtype == torch.float32 and weight.is_contiguous() and weight.shape[0] % 4 == 0
            and x.shape[1] == weight.shape[1] and (bias is None or (bias.dtype == torch.float32 and bias.is_contiguous() and bias.numel() == weight.shape[0]))
            and x.stride(1) == 1 and _fused_forward_wanted(x, weight.shape[1])):
        from .. import abi, lib
        k_in = weight.shape[1]
        w = weight
        if k_in % 4 != 0:            # aligned pitch, unaligned width: the kernel runs over the padded width against zero-padded weights
            k_in = (k_in + 3) // 4 * 4
            w = _padded_weight(weight, k_in)
        z = torch.empty(x.shape[0], weight.shape[0], device=x.device, dtype=torch.float32)
        b = bias.detach() if bias is not None else None
        rc = lib.load().lsim_linear_elu_forward(x.data_ptr(), x.stride(0), w.data_ptr(), b.data_ptr() if b is not None else None, x.shape[0],
                                                k_in, weight.shape[0], z.data_ptr(), z.stride(0), torch.cuda.current_stream(x.device).cuda_stream)
        if rc == 0:
            return z
        if rc != abi.E_UNSUPPORTED:
            lib.check(rc, what="lsim_linear_elu_forward")
    return F.elu(F.linear(x, weight, bias))


class _LinearEluFn(torch.autograd.Function):
    """z = elu(x W^T + b) whose backward runs ONE fused pass (lsim_linear_elu_wgrad): grad_pre = g * elu'(z) formed on the fly as the
    MFMA operand of the weight-gradient kernel (and written once), then the usual BLAS input gradient grad_pre @ W."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        z = linear_elu_forward(x, weight, bias)
        ctx.save_for_backward(x, weight, z)
        ctx.has_bias = bias is not None
        ctx.bias_ptr = bias.data_ptr() if bias is not None else None
        return z

    @staticmethod
    def backward(ctx, g):
        from .. import lib
        x, weight, z = ctx.saved_tensors
        L = lib.load()
        if x.stride(1) != 1:
            x = x.contiguous()
        g = g.contiguous()
        batch, k_in = x.shape
        n_out = weight.shape[0]
        need, parts = ctypes.c_size_t(), ctypes.c_int()
        lib.check(L.lsim_linear_wgrad_workspace(batch, k_in, n_out, ctypes.byref(need), ctypes.byref(parts)), what="lsim_linear_wgrad_workspace")
        wptr = weight.data_ptr()
        ws = _workspace("wgrad" if _pending is None else ("wgrad", wptr), x.device, need.value, floor=1 << 20 if _pending is None else 0)
        hit = [True]
        dw = _grad_out(wptr, (n_out, k_in), x.device, hit)
        db = _grad_out(ctx.bias_ptr, (n_out,), x.device, hit) if ctx.has_bias else None
        # the gradient of the pre-activation is written out only where an input gradient follows (not for a network's first layer: 210 MB per call)
        g_pre = torch.empty(batch, n_out, device=x.device, dtype=torch.float32) if ctx.needs_input_grad[0] else None
        lib.check(_wgrad_call(L, "lsim_linear_elu_wgrad", "lsim_linear_elu_wgrad_deferred",
                              (x.data_ptr(), x.stride(0), g.data_ptr(), g.stride(0), z.data_ptr(), z.stride(0), batch, k_in, n_out,
                               dw.data_ptr(), db.data_ptr() if db is not None else None, g_pre.data_ptr() if g_pre is not None else None,
                               ws.data_ptr(), ws.numel(), torch.cuda.current_stream(x.device).cuda_stream),
                              wptr, ws, (x, g, z), hit[0]), what="lsim_linear_elu_wgrad")
        gx = g_pre @ weight if g_pre is not None else None
        return gx, dw, db


class HimMLP(nn.Sequential):
    """nn.Sequential of Linear / ELU modules (same children, same state_dict keys) that runs eligible (Linear, ELU) pairs through
    _LinearEluFn on the GPU when gradients are needed; everything else exactly as nn.Sequential."""

    def forward(self, x):
        mods = list(self)
        i = 0
        fuse = x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and torch.is_grad_enabled()
        while i < len(mods):
            m = mods[i]
            nxt = mods[i + 1] if i + 1 < len(mods) else None
            if (fuse and isinstance(m, nn.Linear) and m.weight.requires_grad and isinstance(nxt, nn.ELU) and nxt.alpha == 1.0 and not nxt.inplace
                    and _eligible_fused_elu(x.shape[0], m.in_features, m.out_features)):
                x = _LinearEluFn.apply(x, m.weight, m.bias)
                i += 2
            else:
                x = m(x)
                i += 1
        return x


class SkinnyLinear(nn.Linear):
    def forward(self, x):
        if (x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and torch.is_grad_enabled() and self.weight.requires_grad
                and _eligible(x.shape[0], self.in_features, self.out_features)):
            return _SkinnyLinearFn.apply(x, self.weight, self.bias)
        return F.linear(x, self.weight, self.bias)


def sinkhorn_hip(scores, eps, iters):
    """HIMEstimator's Sinkhorn-Knopp assignment through lsim_sinkhorn (same arithmetic as modules.sinkhorn, factored as E * u[k] * v[b])"""
    from .. import lib
    L = lib.load()
    if scores.stride(1) != 1:
        scores = scores.contiguous()
    B, K = scores.shape
    need = ctypes.c_size_t()
    lib.check(L.lsim_sinkhorn_workspace(B, K, ctypes.byref(need)), what="lsim_sinkhorn_workspace")
    ws = _workspace("sinkhorn", scores.device, need.value)
    out = torch.empty(B, K, device=scores.device, dtype=torch.float32)
    lib.check(L.lsim_sinkhorn(scores.data_ptr(), scores.stride(0), B, K, float(eps), int(iters), out.data_ptr(), ws.data_ptr(), ws.numel(),
                              torch.cuda.current_stream(scores.device).cuda_stream), what="lsim_sinkhorn")
    return out


class _PpoLossFn(torch.autograd.Function):
    """clipped-PPO loss through lsim_ppo_loss: forward and the three input gradients come from the same pass.  `sigma` is either the
    per-sample [B, A] tensor or the policy's state-independent std [A] itself (lsim_ppo_loss_std: the broadcast mean * 0 + std of HAC:147, its
    backward and the column sum of its gradient are then never formed)"""

    @staticmethod
    def forward(ctx, mu, sigma, value, actions, old_logp, adv, returns, target_values, old_mu, old_sigma, clip, vcoef, ecoef, clipped, out=None):
        from .. import lib
        L = lib.load()
        c = lambda t: t.detach().contiguous()
        mu_, sg_, v_ = c(mu), c(sigma), c(value).reshape(-1)
        B, A = mu_.shape
        std_mode = sg_.dim() == 1
        need = ctypes.c_size_t()
        if std_mode:
            lib.check(L.lsim_ppo_loss_std_workspace(B, A, ctypes.byref(need)), what="lsim_ppo_loss_std_workspace")
        else:
            lib.check(L.lsim_ppo_loss_workspace(B, ctypes.byref(need)), what="lsim_ppo_loss_workspace")
        ws = torch.empty(need.value, dtype=torch.uint8, device=mu.device)
        if out is None:       # [surrogate, value loss, entropy, kl, total]; the data-parallel step passes the tail of its gradient bucket
            out = torch.empty(5, device=mu.device)
        g_mu, g_sg, g_v = torch.empty_like(mu_), torch.empty_like(sg_), torch.empty_like(v_)
        args = [c(t).reshape(B, -1) if t is not None else None for t in (actions, old_logp, adv, returns, target_values, old_mu, old_sigma)]
        ptr = lambda t: t.data_ptr() if t is not None else None
        fn = L.lsim_ppo_loss_std if std_mode else L.lsim_ppo_loss
        lib.check(fn(mu_.data_ptr(), sg_.data_ptr(), v_.data_ptr(), ptr(args[0]), ptr(args[1]), ptr(args[2]), ptr(args[3]), ptr(args[4]),
                     ptr(args[5]), ptr(args[6]), B, A, float(clip), float(vcoef), float(ecoef), int(bool(clipped)), out.data_ptr(),
                     g_mu.data_ptr(), g_sg.data_ptr(), g_v.data_ptr(), ws.data_ptr(), ws.numel(),
                     torch.cuda.current_stream(mu.device).cuda_stream), what="lsim_ppo_loss")
        ctx.save_for_backward(g_mu, g_sg, g_v)
        ctx.value_shape = value.shape
        _PpoLossFn.last_grads = (g_mu, g_sg, g_v.reshape(value.shape))
        stats = out[:4]
        ctx.mark_non_differentiable(stats)
        return out[4], stats

    @staticmethod
    def backward(ctx, g_loss, _g_stats):
        g_mu, g_sg, g_v = ctx.saved_tensors
        return (g_loss * g_mu, g_loss * g_sg, (g_loss * g_v).reshape(ctx.value_shape)) + (None,) * 12


def ppo_loss_hip(mu, sigma, value, actions, old_logp, adv, returns, target_values, old_mu, old_sigma, clip, vcoef, ecoef, clipped, out=None):
    """-> (total loss with autograd to mu / sigma / value, stats = [surrogate, value loss, entropy, kl] means, detached).  `out`: optional
    5-float CUDA tensor the kernel writes [stats, total] into (the tail of a gradient bucket: the KL estimate then travels with the gradients)"""
    loss, stats = _PpoLossFn.apply(mu, sigma, value, actions, old_logp, adv, returns, target_values, old_mu, old_sigma, clip, vcoef, ecoef, clipped, out)
    loss._lsim_direct = ((mu, sigma, value), _PpoLossFn.last_grads)          # see backward_losses()
    _PpoLossFn.last_grads = None
    return loss, stats


class _EstimatorLossFn(torch.autograd.Function):
    """loss head of HIMEstimator.update through lsim_estimator_loss: the loss and its three input gradients come from one call"""

    @staticmethod
    def forward(ctx, enc_out, tgt_out, proto, vel, temperature, eps, iters):
        from .. import lib
        L = lib.load()
        row = lambda t: t.detach() if t.stride(-1) == 1 else t.detach().contiguous()
        enc_, tgt_, vel_, proto_ = row(enc_out), row(tgt_out), row(vel), proto.detach().contiguous()
        B, D = tgt_.shape
        K = proto_.shape[0]
        need = ctypes.c_size_t()
        lib.check(L.lsim_estimator_loss_workspace(B, D, K, ctypes.byref(need)), what="lsim_estimator_loss_workspace")
        ws = _workspace("estimator_loss", enc_.device, need.value)
        out = torch.empty(3, device=enc_.device)
        g_enc, g_tgt, g_proto = torch.empty(B, 3 + D, device=enc_.device), torch.empty(B, D, device=enc_.device), torch.empty_like(proto_)
        lib.check(L.lsim_estimator_loss(enc_.data_ptr(), enc_.stride(0), tgt_.data_ptr(), tgt_.stride(0), proto_.data_ptr(), vel_.data_ptr(),
                                        vel_.stride(0), B, D, K, float(temperature), float(eps), int(iters), out.data_ptr(), g_enc.data_ptr(),
                                        g_tgt.data_ptr(), g_proto.data_ptr(), ws.data_ptr(), ws.numel(),
                                        torch.cuda.current_stream(enc_.device).cuda_stream), what="lsim_estimator_loss")
        ctx.save_for_backward(g_enc, g_tgt, g_proto)
        _EstimatorLossFn.last_grads = (g_enc, g_tgt, g_proto)
        parts = out[:2]
        ctx.mark_non_differentiable(parts)
        return out[2], parts

    @staticmethod
    def backward(ctx, g_total, _g_parts):
        grads = list(ctx.saved_tensors)
        torch._foreach_mul_(grads, g_total)       # private buffers of this call: scaled in place, one launch
        return grads[0], grads[1], grads[2], None, None, None, None


def estimator_loss_hip(enc_out, tgt_out, proto, vel, temperature, eps=0.05, iters=3):
    """-> (est + swap with autograd to enc_out / tgt_out / proto, [est, swap] detached); see include/lsim.h lsim_estimator_loss"""
    total, parts = _EstimatorLossFn.apply(enc_out, tgt_out, proto, vel, temperature, eps, iters)
    total._lsim_direct = ((enc_out, tgt_out, proto), _EstimatorLossFn.last_grads)      # see backward_losses()
    _EstimatorLossFn.last_grads = None
    return total, parts


def backward_losses(*losses):
    """loss.backward() for each of `losses`, except that a loss that came out of one of the fused loss kernels above is back-propagated from
    the gradients that kernel already produced: torch.autograd.backward(inputs, d loss / d inputs) instead of a root gradient of ones times
    those gradients -- the ones_like fill and the three multiplications by 1.0 per loss disappear (14 MB per minibatch for the estimator's);
    multiplying by one changes no bit.  All such losses go through ONE engine run.  Losses without the recipe (torch statements, sums of
    losses) take the ordinary path."""
    tensors, grads, rest = [], [], []
    direct = os.environ.get("LSIM_DIRECT_LOSS_BACKWARD", "1") != "0"        # A/B hook
    for l in losses:
        d = getattr(l, "_lsim_direct", None) if direct else None
        if d is None or not all(t.requires_grad for t in d[0]):
            rest.append(l)
        else:
            tensors += list(d[0]); grads += list(d[1])
    if tensors:
        torch.autograd.backward(tensors, grads)
    for l in rest:
        l.backward()


def estimator_loss_supported(latent, K):
    return latent <= 32 and K <= 64


def adam_clip_step_hip(optimizer, max_grad_norm, clip_params=None):
    """clip_grad_norm_(clip_params or all params, max_grad_norm) + optimizer.step() for a plain torch.optim.Adam through lsim_adam_clip_step_ex
    (2 launches instead of ~12): works on the optimizer's own parameter / state tensors, so state_dict() and checkpoints stay torch's.
    Several parameter groups are fine as long as they share lr / betas / eps (HybridPPO: three groups that differ in weight decay only);
    `clip_params` restricts the clipped norm to a subset (HybridPPO clips the actor-critic only, HYBP:270).  Returns False when the
    optimizer is not eligible (the caller then runs the torch statements): amsgrad / maximize, non-fp32 or non-CUDA parameters, more than
    48 tensors, groups with different hyper-parameters, or a parameter that has a gradient but no state yet (first step: torch
    initialises it)."""
    from .. import lib
    groups = optimizer.param_groups
    g0 = groups[0]
    for grp in groups:
        if grp.get("amsgrad") or grp.get("maximize") or grp.get("differentiable") or grp.get("weight_decay", 0) < 0:
            return False
        same_lr = grp["lr"] is g0["lr"] or (not torch.is_tensor(grp["lr"]) and not torch.is_tensor(g0["lr"]) and grp["lr"] == g0["lr"])
        if grp["betas"] != g0["betas"] or grp["eps"] != g0["eps"] or not same_lr:      # (no tensor comparison: that would be a host sync)
            return False
    clip_ids = None if clip_params is None else {id(p) for p in clip_params}
    entries = [(p, float(grp.get("weight_decay", 0))) for grp in groups for p in grp["params"] if p.grad is not None]
    if not entries or len(entries) > 48:
        return False
    if clip_ids is not None:       # the clipped tensors first
        entries = [e for e in entries if id(e[0]) in clip_ids] + [e for e in entries if id(e[0]) not in clip_ids]
    n_clip = len(entries) if clip_ids is None else sum(1 for e in entries if id(e[0]) in clip_ids)
    n = len(entries)
    L = lib.load()
    dev = entries[0][0].device
    # the C tables of one call are cached: with the gradients in a GradArena every pointer repeats from minibatch to minibatch and the ~40 state
    # look-ups / 5 x 40 ctypes conversions of a call disappear (the loop is within 1.6 x of launch-bound).  The key holds EVERY
    # pointer and value the tables hold -- parameters, gradients, both moments, the step counters, the weight decays -- and the entry is
    # dropped with the optimizer (weak reference; id() alone can be reused): optimizer.load_state_dict() / state.clear() replace the state
    # tensors, and a table that still pointed at the old ones would step freed memory and leave the loaded moments untouched
    states = [optimizer.state.get(p) for p, _ in entries]
    for (p, _), st in zip(entries, states):
        if (not st or not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous() or not p.grad.is_contiguous()
                or not torch.is_tensor(st.get("step")) or not st["step"].is_cuda or st["step"].dtype != torch.float32):
            return False
    key = (n_clip, tuple(p.data_ptr() for p, _ in entries), tuple(p.grad.data_ptr() for p, _ in entries),
           tuple(st["exp_avg"].data_ptr() for st in states), tuple(st["exp_avg_sq"].data_ptr() for st in states),
           tuple(st["step"].data_ptr() for st in states), tuple(w for _, w in entries))
    cached = _adam_tables.get(id(optimizer))
    if cached is not None and cached[0] == key and cached[2]() is optimizer:
        tables = cached[1]
    else:
        tabs = ([], [], [], [], [])
        for (p, _), st in zip(entries, states):
            for tab, t in zip(tabs, (p, p.grad, st["exp_avg"], st["exp_avg_sq"], st["step"])):
                tab.append(t.data_ptr())
        need = ctypes.c_size_t()
        lib.check(L.lsim_adam_clip_step_workspace(n, ctypes.byref(need)), what="lsim_adam_clip_step_workspace")
        arr = lambda v: (ctypes.c_void_p * n)(*v)
        tables = (arr(tabs[0]), arr(tabs[1]), arr(tabs[2]), arr(tabs[3]), arr(tabs[4]), (ctypes.c_int64 * n)(*[p.numel() for p, _ in entries]),
                  (ctypes.c_float * n)(*[w for _, w in entries]), need.value)
        oid = id(optimizer)
        _adam_tables[oid] = (key, tables, weakref.ref(optimizer, lambda _r, oid=oid: _adam_tables.pop(oid, None)))
    ws = _workspace(("adam", id(optimizer)), dev, tables[7])
    lr = g0["lr"]
    lr_dev = lr.data_ptr() if torch.is_tensor(lr) and lr.is_cuda else None
    lr_host = 0.0 if lr_dev is not None else float(lr)
    b1, b2 = g0["betas"]
    lib.check(L.lsim_adam_clip_step_ex(n, tables[5], tables[0], tables[1], tables[2], tables[3], tables[4], tables[6], n_clip, lr_dev, lr_host,
                                       float(b1), float(b2), float(g0["eps"]), float(max_grad_norm), None, ws.data_ptr(), ws.numel(),
                                       torch.cuda.current_stream(dev).cuda_stream), what="lsim_adam_clip_step_ex")
    return True
