"""nn.Linear whose backward uses the library's tall-skinny weight-gradient kernel (include/lsim.h, lsim_linear_wgrad) for the
learner's narrow layers on the GPU.

Forward and grad_input are the usual BLAS calls.  grad_weight = g^T x and grad_bias = g.sum(0) over a 102 400-row minibatch are
K = 102 400 reductions into a small output; rocBLAS/hipBLASLt run them at 1-55 % of the fp32 MFMA peak (and torch's column sum
needs 260 us for 19 columns); the MFMA kernels in csrc/ls_learn.h read x and g once per 64 x 128 output tile.  Same parameters,
same state_dict keys, same initialisation as nn.Linear (it IS an nn.Linear); results differ from BLAS by fp32 summation order.
"""
import ctypes

import torch
import torch.nn as nn
import torch.nn.functional as F

_MIN_BATCH = 4096
_workspaces = {}
_plans = {}


def _workspace(kind, device, nbytes, floor=0):
    """scratch buffer of one kernel family, PER STREAM: the update runs the actor / estimator chain and the critic chain on two streams
    (him_ppo.py), and two weight-gradient kernels in flight at once must not share their partial-result buffers"""
    key = (kind, device, torch.cuda.current_stream(device).cuda_stream)
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(int(nbytes), int(floor)), dtype=torch.uint8, device=device)
        _workspaces[key] = ws
    return ws


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


def linear_wgrad(x, g, want_bias=True):
    """(g^T x, g.sum(0)) for 2-D fp32 CUDA tensors through lsim_linear_wgrad"""
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
    ws = _workspace("wgrad", x.device, need.value, floor=1 << 20)
    dw = torch.empty(n_out, k_in, device=x.device, dtype=torch.float32)
    db = torch.empty(n_out, device=x.device, dtype=torch.float32) if want_bias else None
    lib.check(L.lsim_linear_wgrad(x.data_ptr(), x.stride(0), g.data_ptr(), g.stride(0), batch, k_in, n_out, dw.data_ptr(),
                                  db.data_ptr() if want_bias else None, ws.data_ptr(), ws.numel(),
                                  torch.cuda.current_stream(x.device).cuda_stream), what="lsim_linear_wgrad")
    return dw, db


class _SkinnyLinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return F.linear(x, weight, bias)

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        gx = g @ weight if ctx.needs_input_grad[0] else None
        dw = db = None
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            dw, db = linear_wgrad(x, g, want_bias=ctx.has_bias)
        return gx, dw, db


def _eligible_fused_elu(batch, k_in, n_out):
    """Linear + ELU pairs whose weight gradient runs in the library's tiled kernel (more than 4096 outputs, see ls_learn.h)"""
    return _eligible(batch, k_in, n_out) and k_in * n_out > 4096


class _LinearEluFn(torch.autograd.Function):
    """z = elu(x W^T + b) whose backward runs ONE fused pass (lsim_linear_elu_wgrad): grad_pre = g * elu'(z) formed on the fly as the
    MFMA operand of the weight-gradient kernel (and written once), then the usual BLAS input gradient grad_pre @ W."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        z = F.elu(F.linear(x, weight, bias))
        ctx.save_for_backward(x, weight, z)
        ctx.has_bias = bias is not None
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
        ws = _workspace("wgrad", x.device, need.value, floor=1 << 20)
        dw = torch.empty(n_out, k_in, device=x.device, dtype=torch.float32)
        db = torch.empty(n_out, device=x.device, dtype=torch.float32) if ctx.has_bias else None
        # the gradient of the pre-activation is written out only where an input gradient follows (not for a network's first layer: 210 MB per call)
        g_pre = torch.empty(batch, n_out, device=x.device, dtype=torch.float32) if ctx.needs_input_grad[0] else None
        lib.check(L.lsim_linear_elu_wgrad(x.data_ptr(), x.stride(0), g.data_ptr(), g.stride(0), z.data_ptr(), z.stride(0), batch, k_in, n_out,
                                          dw.data_ptr(), db.data_ptr() if db is not None else None, g_pre.data_ptr() if g_pre is not None else None,
                                          ws.data_ptr(), ws.numel(), torch.cuda.current_stream(x.device).cuda_stream), what="lsim_linear_elu_wgrad")
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
    """clipped-PPO loss through lsim_ppo_loss: forward and the three input gradients come from the same pass"""

    @staticmethod
    def forward(ctx, mu, sigma, value, actions, old_logp, adv, returns, target_values, old_mu, old_sigma, clip, vcoef, ecoef, clipped):
        from .. import lib
        L = lib.load()
        c = lambda t: t.detach().contiguous()
        mu_, sg_, v_ = c(mu), c(sigma), c(value).reshape(-1)
        B, A = mu_.shape
        need = ctypes.c_size_t()
        lib.check(L.lsim_ppo_loss_workspace(B, ctypes.byref(need)), what="lsim_ppo_loss_workspace")
        ws = torch.empty(need.value, dtype=torch.uint8, device=mu.device)
        out = torch.empty(5, device=mu.device)
        g_mu, g_sg, g_v = torch.empty_like(mu_), torch.empty_like(sg_), torch.empty_like(v_)
        args = [c(t).reshape(B, -1) if t is not None else None for t in (actions, old_logp, adv, returns, target_values, old_mu, old_sigma)]
        ptr = lambda t: t.data_ptr() if t is not None else None
        lib.check(L.lsim_ppo_loss(mu_.data_ptr(), sg_.data_ptr(), v_.data_ptr(), ptr(args[0]), ptr(args[1]), ptr(args[2]), ptr(args[3]), ptr(args[4]),
                                  ptr(args[5]), ptr(args[6]), B, A, float(clip), float(vcoef), float(ecoef), int(bool(clipped)), out.data_ptr(),
                                  g_mu.data_ptr(), g_sg.data_ptr(), g_v.data_ptr(), ws.data_ptr(), ws.numel(),
                                  torch.cuda.current_stream(mu.device).cuda_stream), what="lsim_ppo_loss")
        ctx.save_for_backward(g_mu, g_sg, g_v)
        ctx.value_shape = value.shape
        stats = out[:4]
        ctx.mark_non_differentiable(stats)
        return out[4], stats

    @staticmethod
    def backward(ctx, g_loss, _g_stats):
        g_mu, g_sg, g_v = ctx.saved_tensors
        return (g_loss * g_mu, g_loss * g_sg, (g_loss * g_v).reshape(ctx.value_shape)) + (None,) * 11


def ppo_loss_hip(mu, sigma, value, actions, old_logp, adv, returns, target_values, old_mu, old_sigma, clip, vcoef, ecoef, clipped):
    """-> (total loss with autograd to mu / sigma / value, stats = [surrogate, value loss, entropy, kl] means, detached)"""
    return _PpoLossFn.apply(mu, sigma, value, actions, old_logp, adv, returns, target_values, old_mu, old_sigma, clip, vcoef, ecoef, clipped)


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
    return _EstimatorLossFn.apply(enc_out, tgt_out, proto, vel, temperature, eps, iters)


def estimator_loss_supported(latent, K):
    return latent <= 32 and K <= 64


def adam_clip_step_hip(optimizer, max_grad_norm, clip_params=None):
    """clip_grad_norm_(clip_params or all params, max_grad_norm) + optimizer.step() for a plain torch.optim.Adam through lsim_adam_clip_step_ex
    (3 launches instead of ~12): works on the optimizer's own parameter / state tensors, so state_dict() and checkpoints stay torch's.
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
    tabs = ([], [], [], [], [])
    for p, _ in entries:
        st = optimizer.state.get(p)
        if (not st or not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous() or not p.grad.is_contiguous()
                or not torch.is_tensor(st.get("step")) or not st["step"].is_cuda or st["step"].dtype != torch.float32):
            return False
        for tab, t in zip(tabs, (p, p.grad, st["exp_avg"], st["exp_avg_sq"], st["step"])):
            tab.append(t.data_ptr())
    n = len(entries)
    L = lib.load()
    dev = entries[0][0].device
    need = ctypes.c_size_t()
    lib.check(L.lsim_adam_clip_step_workspace(n, ctypes.byref(need)), what="lsim_adam_clip_step_workspace")
    ws = _workspace(("adam", id(optimizer)), dev, need.value)
    arr = lambda v: (ctypes.c_void_p * n)(*v)
    numel = (ctypes.c_int64 * n)(*[p.numel() for p, _ in entries])
    wd = (ctypes.c_float * n)(*[w for _, w in entries])
    lr = g0["lr"]
    lr_dev = lr.data_ptr() if torch.is_tensor(lr) and lr.is_cuda else None
    lr_host = 0.0 if lr_dev is not None else float(lr)
    b1, b2 = g0["betas"]
    lib.check(L.lsim_adam_clip_step_ex(n, numel, arr(tabs[0]), arr(tabs[1]), arr(tabs[2]), arr(tabs[3]), arr(tabs[4]), wd, n_clip, lr_dev, lr_host,
                                       float(b1), float(b2), float(g0["eps"]), float(max_grad_norm), None, ws.data_ptr(), ws.numel(),
                                       torch.cuda.current_stream(dev).cuda_stream), what="lsim_adam_clip_step_ex")
    return True
