"""GPU: the kernels of the discriminator UPDATE (csrc/ls_amp.h second half, ls_learn.h lsim_linear_relu_wgrad, ls_gemm.h lsim_linear_masked_forward)
against the torch statements they replace, and the closed-form LSGAN / gradient-penalty functions built on them (learn/amp.py) against autograd on the
reference's formulation (rsl_rl/algorithms/hybrid_ppo.py:252-263, amp_discriminator.py:36-53).  fp32; tolerances state the summation-order
difference of K = batch sums (1e-4 relative on gradients of 20 000-row batches)."""
import ctypes
import os

import numpy as np
import pytest
import torch

from helpers import ROOT

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _stream():
    return torch.cuda.current_stream().cuda_stream


def test_relu_head_backward_and_masked_colsum_match_torch():
    from isaacgymloco_amd import lib
    L = lib.load()
    g = torch.Generator().manual_seed(0)
    for B, n in ((20000, 512), (777, 256), (4100, 1024)):
        a2 = torch.relu(torch.randn(B, n, generator=g)).to(DEV)
        gd, w3 = torch.randn(B, generator=g).to(DEV), torch.randn(n, generator=g).to(DEV)
        need = ctypes.c_size_t()
        assert L.lsim_relu_cols_workspace(B, n, ctypes.byref(need)) == 0
        ws = torch.empty(need.value, dtype=torch.uint8, device=DEV)
        g2, db2, dh = torch.empty(B, n, device=DEV), torch.empty(n, device=DEV), torch.empty(n + 4, device=DEV)
        lib.check(L.lsim_relu_head_backward(a2.data_ptr(), n, gd.data_ptr(), w3.data_ptr(), B, n, g2.data_ptr(), db2.data_ptr(), dh.data_ptr(), ws.data_ptr(),
                                            ws.numel(), _stream()))
        ref = torch.ops.aten.threshold_backward(gd[:, None] * w3[None, :], a2, 0.0)
        torch.testing.assert_close(g2, ref, rtol=0, atol=0)                                     # one product per element: exact
        torch.testing.assert_close(db2, ref.double().sum(0).float(), rtol=2e-5, atol=2e-4)
        torch.testing.assert_close(dh[:n], (a2.double().t() @ gd.double()).float(), rtol=2e-5, atol=2e-4)
        torch.testing.assert_close(dh[n], gd.double().sum().float(), rtol=2e-5, atol=2e-4)
        v = torch.randn(B, n, generator=g).to(DEV)
        out = torch.empty(n, device=DEV)
        lib.check(L.lsim_masked_colsum(v.data_ptr(), n, a2.data_ptr(), n, B, n, out.data_ptr(), ws.data_ptr(), ws.numel(), _stream()))
        torch.testing.assert_close(out, torch.where(a2 > 0, v, torch.zeros_like(v)).double().sum(0).float(), rtol=2e-5, atol=2e-4)
    assert L.lsim_relu_cols_workspace(100, 30, ctypes.byref(need)) == -4                       # LSIM_E_UNSUPPORTED: n % 4


def test_linear_masked_forward_and_relu_wgrad_match_torch():
    from isaacgymloco_amd import lib
    from isaacgymloco_amd.learn import amp
    L = lib.load()
    g = torch.Generator().manual_seed(1)
    B, k, n = 20000, 60, 1024
    x, W = torch.randn(B, k, generator=g).to(DEV), (torch.randn(n, k, generator=g) * 0.1).to(DEV)
    a = torch.relu(torch.randn(B, n, generator=g)).to(DEV)
    out = torch.empty(B, n, device=DEV)
    lib.check(L.lsim_linear_masked_forward(x.data_ptr(), k, W.data_ptr(), a.data_ptr(), n, B, k, n, out.data_ptr(), n, _stream()))
    ref = torch.ops.aten.threshold_backward((x.double() @ W.double().t()).float(), a, 0.0)
    torch.testing.assert_close(out, ref, rtol=1e-5, atol=1e-5)
    assert bool(((out == 0) == (a <= 0)).all()) or bool((out[a > 0] == 0).sum() < 10)         # the mask is the activation's, exactly
    # dW, db of relu(x W^T + b) from the gradient of its output: the mask rides in the weight-gradient kernel
    go = torch.randn(B, n, generator=g).to(DEV)
    bias = torch.zeros(n, device=DEV)
    dw, db = amp._relu_wgrad(x, go, a, W, bias)
    from isaacgymloco_amd.learn import fused_linear as FL
    FL.flush_wgrad_reduces()
    gy = torch.where(a > 0, go, torch.zeros_like(go)).double()
    torch.testing.assert_close(dw, (gy.t() @ x.double()).float(), rtol=1e-4, atol=1e-3)
    torch.testing.assert_close(db, gy.sum(0).float(), rtol=1e-4, atol=1e-3)


def test_running_moments_update_matches_the_reference_fixture_and_torch():
    from isaacgymloco_amd.learn import amp
    fx = np.load(os.path.join(ROOT, "tests", "golden", "learner_amp.npz"))
    nz = amp.Normalizer(30, device=DEV)
    nz.update(torch.from_numpy(fx["nz_x1"]).to(DEV)); nz.update(torch.from_numpy(fx["nz_x2"]).to(DEV))
    assert getattr(nz, "_ws", None) is not None, "the fused update did not run"
    np.testing.assert_allclose(nz.mean, fx["nz_mean"], rtol=2e-6, atol=1e-7)                   # the REFERENCE's Normalizer after the same two updates
    np.testing.assert_allclose(nz.var, fx["nz_var"], rtol=2e-6, atol=1e-7)
    assert abs(nz.count - float(fx["nz_count"])) < 1e-9
    np.testing.assert_allclose(nz.normalize_torch(torch.from_numpy(fx["nz_probe"]).to(DEV)).cpu().numpy(), fx["nz_probe_out"], rtol=1e-6, atol=1e-6)
    # a tall batch against float64 torch
    g = torch.Generator().manual_seed(3)
    x = (torch.randn(102400, 30, generator=g) * 2.5 + 0.7).to(DEV)
    m0, v0, c0 = nz._mean.clone(), nz._var.clone(), nz._count.clone().reshape(())
    nz.update(x)
    xd = x.double()
    bm, bv, n = xd.mean(0), xd.var(0, unbiased=False), 102400.0
    delta, tot = bm - m0, c0 + n
    torch.testing.assert_close(nz._mean, m0 + delta * n / tot, rtol=1e-12, atol=1e-12)
    torch.testing.assert_close(nz._var, (v0 * c0 + bv * n + delta * delta * c0 * n / tot) / tot, rtol=1e-10, atol=1e-12)
    assert abs(nz.count - float(tot)) < 1e-6


def _reference_losses(disc, exp_in, pol_in, raw_s, raw_ns):
    """the reference's statements through autograd (HYBP:252-263, DISC:36-53)"""
    pd, ed = disc.amp_linear(disc.trunk(pol_in)), disc.amp_linear(disc.trunk(exp_in))
    amp_loss = 0.5 * (torch.nn.functional.mse_loss(ed, torch.ones_like(ed)) + torch.nn.functional.mse_loss(pd, -torch.ones_like(pd)))
    data = torch.cat([raw_s, raw_ns], dim=-1).clone().requires_grad_(True)
    out = disc.amp_linear(disc.trunk(data))
    grad = torch.autograd.grad(outputs=out, inputs=data, grad_outputs=torch.ones_like(out), create_graph=True, retain_graph=True, only_inputs=True)[0]
    return amp_loss, 10 * (grad.norm(2, dim=1) - 0).pow(2).mean(), pd.mean().detach(), ed.mean().detach()


def test_fused_lsgan_loss_and_gradient_penalty_match_autograd():
    from isaacgymloco_amd.learn import amp
    from isaacgymloco_amd.learn import fused_linear as FL
    torch.manual_seed(3)
    disc = amp.AMPDiscriminator(60, 0.01, [1024, 512], DEV, 0.3).to(DEV)
    g = torch.Generator().manual_seed(7)
    B = 20000
    exp_in, pol_in = torch.randn(B, 60, generator=g).to(DEV), (torch.randn(B, 60, generator=g) * 1.5).to(DEV)
    raw_s, raw_ns = torch.randn(B, 30, generator=g).to(DEV), torch.randn(B, 30, generator=g).to(DEV)
    al, gp, pm, em = _reference_losses(disc, exp_in, pol_in, raw_s, raw_ns)
    (al + gp).backward()
    ref = {k: p.grad.clone() for k, p in disc.named_parameters()}
    disc.zero_grad()
    assert amp._fused_disc_ok(exp_in, disc.trunk, disc.amp_linear)
    al2, pm2, em2 = disc.lsgan_loss(exp_in, pol_in)
    gp2 = disc.compute_grad_pen(raw_s, raw_ns, lambda_=10)
    with FL.deferred_wgrad_reduce():
        (al2 + gp2).backward()
    torch.testing.assert_close(al2, al, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(gp2, gp, rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(pm2, pm, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(em2, em, rtol=1e-4, atol=1e-5)
    for k, p in disc.named_parameters():
        scale = float(ref[k].abs().max())
        torch.testing.assert_close(p.grad, ref[k], rtol=2e-4, atol=2e-5 * max(scale, 1e-3), msg=lambda m, k=k: f"{k}: {m}")


def _hybrid_update(fused, monkeypatch):
    """one HybridPPO.update() on synthetic rollout data at minibatches of 16 384 rows (where the fused discriminator path is eligible), inside the real
    optimiser flow: gradient arena, deferred weight-gradient sums, fused clip + Adam"""
    from isaacgymloco_amd.learn import amp
    from isaacgymloco_amd.learn.hybrid import HybridPPO
    from isaacgymloco_amd.learn.modules import HIMActorCritic
    from test_amp_golden import ALG as AMP_ALG, BUNDLE
    from test_gpu_learner_golden import _replay_rollout
    monkeypatch.setenv("LSIM_AMP_FUSED_UPDATE", "1" if fused else "0")
    N, T = 1024, 64
    np.random.seed(1)
    ld = amp.AMPLoader(DEV, time_between_frames=0.02, preload_transitions=True, num_preload_transitions=50000, motion_files=[BUNDLE])
    torch.manual_seed(0)
    ac = HIMActorCritic(270, 238, 45, 12, actor_hidden_dims=[512, 256, 128], critic_hidden_dims=[512, 256, 128], activation="elu", init_noise_std=1.0)
    disc = amp.AMPDiscriminator(60, 0.5 * 0.02, [1024, 512], "cpu", 0.3)
    nz = amp.Normalizer(30, device=DEV)
    alg = HybridPPO(ac, disc, ld, nz, device=DEV, min_std=(torch.tensor([0.05, 0.02, 0.05] * 4) * 1.5).to(DEV), **dict(AMP_ALG, amp_replay_buffer_size=N * T))
    alg.discriminator.device = DEV
    alg.init_storage(N, T, [270], [238], [12])
    g = torch.Generator().manual_seed(5)
    obs, crit = torch.randn(T + 1, N, 270, generator=g).to(DEV), torch.randn(T + 1, N, 238, generator=g).to(DEV)
    ampo = (torch.randn(T + 1, N, 30, generator=g) * 0.5).to(DEV)
    rew, done = torch.randn(T, N, generator=g).to(DEV), (torch.rand(T, N, generator=g) < 0.05).to(DEV)
    actions = torch.randn(T, N, 12, generator=g).to(DEV)
    np.random.seed(7)
    _replay_rollout(alg, obs, crit, rew, done, done & False, actions, amp=ampo)
    orig = torch.randperm
    monkeypatch.setattr(torch, "randperm", lambda n, **kw: orig(n, generator=torch.Generator().manual_seed(11)).to(kw.get("device", "cpu")))
    res = alg.update()
    torch.cuda.synchronize()
    return np.array(res), {k: v.detach().clone() for k, v in list(ac.named_parameters()) + [("disc." + k, p) for k, p in disc.named_parameters()]}, nz


def test_hybrid_update_with_fused_discriminator_path_matches_torch_statements(monkeypatch):
    res_f, par_f, nz_f = _hybrid_update(True, monkeypatch)
    res_t, par_t, nz_t = _hybrid_update(False, monkeypatch)
    assert getattr(nz_f, "_ws", None) is not None and getattr(nz_t, "_ws", None) is None        # the switch did switch
    np.testing.assert_allclose(res_f, res_t, rtol=2e-3, atol=2e-5)
    np.testing.assert_allclose(nz_f.mean, nz_t.mean, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(nz_f.var, nz_t.var, rtol=1e-5, atol=1e-6)
    for k in par_t:                                       # four Adam steps from identical states: parameters agree far inside one step's size (lr 1e-3)
        torch.testing.assert_close(par_f[k], par_t[k], rtol=0, atol=3e-4, msg=lambda m, k=k: f"{k}: {m}")


@pytest.mark.gpu
def test_index_upload_without_a_pipeline_drain_delivers_every_vector():
    """learn/amp.py: _IndexUploader -- host-drawn indices through a ring of page-locked buffers with asynchronous copies: 200 uploads (the ring of 64
    wraps three times) behind a long-running kernel queue, checked only at the end: a buffer overwritten before its copy ran would show"""
    from isaacgymloco_amd.learn.amp import _IndexUploader
    up = _IndexUploader(slots=64)
    rs = np.random.RandomState(0)
    a = torch.randn(4096, 4096, device="cuda:0")
    host, dev = [], []
    for i in range(200):
        if i % 10 == 0:
            a = a @ a * 1e-3                  # keep the stream busy: the host runs ahead of the copies
        idx = rs.choice(1 << 20, size=102400 if i % 3 else 1000)
        host.append(idx.copy())
        dev.append(up(idx, "cuda:0"))
        idx[:] = -1                           # the caller's array is free to change once the call has returned
    torch.cuda.synchronize()
    for h, d in zip(host, dev):
        assert d.dtype == torch.int64 and d.device.type == "cuda" and np.array_equal(d.cpu().numpy(), h)
    assert up(np.arange(5), "cpu").tolist() == [0, 1, 2, 3, 4]


@pytest.mark.gpu
def test_pair_inputs_match_the_reference_statements():
    """lsim_amp_pair_rows through AMPDiscriminator.pair_inputs: normalise_torch on the four sampled blocks + torch.cat (HYBP:247-251, DISC:57, DISC:37),
    bit for bit (the same float32 operations per element); the two normalised blocks are adjacent rows of one buffer and lsgan_loss takes them without a copy"""
    from isaacgymloco_amd.learn.amp import AMPDiscriminator, Normalizer
    torch.manual_seed(3)
    B, D = 20000, 30
    disc = AMPDiscriminator(2 * D, 2.0, [1024, 512], DEV, 0.3).to(DEV)
    nz = Normalizer(D, device=DEV)
    nz.update(torch.randn(4096, D, device=DEV) * 3.0 + 1.0)
    big = torch.randn(4, B, D + 2, device=DEV) * 4.0                       # strided rows (pitch D + 2)
    big[:, ::7, 3] *= 50.0                                                  # ... and values beyond the clip
    es, ens, ps, pns = (big[i, :, :D] for i in range(4))
    e_in, p_in, raw, e_n, p_n = disc.pair_inputs(es, ens, ps, pns, nz)
    f = nz.normalize_torch
    assert torch.equal(e_in, torch.cat([f(es), f(ens)], dim=-1)) and torch.equal(p_in, torch.cat([f(ps), f(pns)], dim=-1))
    assert torch.equal(raw, torch.cat([es, ens], dim=-1)) and torch.equal(e_n, f(es)) and torch.equal(p_n, f(ps))
    assert float(e_in.abs().max()) == nz.clip_obs                           # the clamp is exercised
    assert p_in.data_ptr() == e_in.data_ptr() + e_in.numel() * 4            # one stacked buffer
    a = disc.lsgan_loss(e_in, p_in)
    b = disc.lsgan_loss(e_in.clone(), p_in.clone())                         # separate tensors: the cat path
    assert all(torch.equal(x, y) for x, y in zip(a, b))
    # without a normaliser: plain concatenation
    e2, p2, raw2, e2n, p2n = disc.pair_inputs(es, ens, ps, pns, None)
    assert torch.equal(e2, torch.cat([es, ens], dim=-1)) and torch.equal(p2, torch.cat([ps, pns], dim=-1)) and torch.equal(e2n, es)
    # small batches keep the torch statements (same values)
    s_in = disc.pair_inputs(es[:100], ens[:100], ps[:100], pns[:100], nz)
    assert torch.equal(s_in[0], e_in[:100]) and torch.equal(s_in[1], p_in[:100])
