"""GPU: lsim_amp_step (csrc/ls_amp.h) -- the AMP rollout step in one launch -- against the statements it replaces:
AMPDiscriminator.predict_amp_reward (rsl_rl/algorithms/amp_discriminator.py:55-72) with the running-moment normaliser
(rsl_rl/utils/utils.py:124-130), the terminal-state patch of HybridPolicyRunner (rsl_rl/runners/hybrid_runner.py:191-196) and
ReplayBuffer.insert (rsl_rl/storage/replay_buffer.py:52-68).

Pinned to the reference twice: `learner_amp.npz` holds predict_amp_reward's outputs and a replay ring filled by the reference's own
classes (tools/gen_golden_amp.py); the larger cases compare with the build's torch restatement of those classes, which
tests/test_amp_golden.py pins to the same fixture on the CPU.
Tolerance: fp32 with a different summation order (MFMA tiles over k chunks of 16 vs BLAS) -- d to 2e-5 relative / 2e-5 absolute at |d| ~ 1,
rewards to 1e-5 absolute (coef 0.01: |d reward / d d| <= 0.005); replay rows and the carry are copies: bit-exact.
"""
import os

import numpy as np
import pytest
import torch

from helpers import ROOT

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _disc(seed=3, hidden=(1024, 512), lerp=0.3, coef=0.5 * 0.02):
    from isaacgymloco_amd.learn import amp
    torch.manual_seed(seed)
    d = amp.AMPDiscriminator(60, coef, list(hidden), "cpu", lerp).to(DEV)
    d.device = DEV
    return d


def _normalizer(fx=None, seed=0):
    from isaacgymloco_amd.learn import amp
    nz = amp.Normalizer(30, device=DEV)
    if fx is not None:
        nz.update(torch.from_numpy(fx["nz_x1"]).to(DEV)); nz.update(torch.from_numpy(fx["nz_x2"]).to(DEV))
    else:
        g = torch.Generator().manual_seed(seed)
        nz.update((torch.randn(500, 30, generator=g) * 1.5 + 0.3).to(DEV))
    return nz


def test_amp_step_reproduces_the_reference_fixture():
    """predict_amp_reward of the REFERENCE on 12 pairs (normaliser after two updates, lerp 0.3) and its ReplayBuffer after four wrapped inserts"""
    from isaacgymloco_amd.learn import amp
    from isaacgymloco_amd.learn.fused_amp import PackedAmpDisc
    fx = np.load(os.path.join(ROOT, "tests", "golden", "learner_amp.npz"))
    disc, nz = _disc(), _normalizer(fx)
    assert PackedAmpDisc.supported(disc)
    pk = PackedAmpDisc(disc, nz, 12)
    s, ns, task = (torch.from_numpy(fx[k]).to(DEV) for k in ("disc_s", "disc_ns", "disc_task"))
    rew, d = torch.zeros(12, device=DEV), torch.zeros(12, device=DEV)
    pk.step(s, ns, None, None, task, rew, disc_out=d)
    torch.cuda.synchronize()
    np.testing.assert_allclose(d.cpu().numpy(), fx["disc_d"][:, 0], rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(rew.cpu().numpy(), fx["disc_reward"], rtol=1e-5, atol=1e-6)
    # the replay ring: four chunks of 8 rows into 20 slots (two wraps), as the reference inserted them
    rb = amp.ReplayBuffer(30, 20, DEV)
    pk8 = PackedAmpDisc(disc, nz, 8)
    for c in torch.from_numpy(fx["rb_chunks"]).to(DEV):
        pk8.step(c, c + 1, None, None, torch.zeros(8, device=DEV), torch.zeros(8, device=DEV), replay=rb)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(rb.states.cpu().numpy(), fx["rb_states"])
    np.testing.assert_array_equal(rb.next_states.cpu().numpy(), fx["rb_next"])
    assert rb.step == int(fx["rb_step"]) and rb.num_samples == int(fx["rb_num"])


@pytest.mark.parametrize("n,lerp,use_nz", [(4096, 0.3, True), (37, 0.3, True), (4096, 0.0, False), (9000, 0.3, True)])
def test_amp_step_matches_torch_statements(n, lerp, use_nz):
    """N = 4096: two blocks per row group (the cross-block sum); 9000: one block per group; 37: a ragged last group"""
    from isaacgymloco_amd.learn import amp
    from isaacgymloco_amd.learn.fused_amp import PackedAmpDisc
    disc = _disc(seed=11, lerp=lerp)
    nz = _normalizer(seed=4) if use_nz else None
    g = torch.Generator().manual_seed(n)
    s, nxt, term = (torch.randn(n, 30, generator=g).to(DEV) * 1.2 for _ in range(3))
    s[:, 3] = 40.0                                        # a column that hits the normaliser's clip
    task = torch.randn(n, generator=g).to(DEV)
    dones = (torch.rand(n, generator=g) < 0.2).to(DEV)
    cap = n + 100
    rb_t, rb_k = amp.ReplayBuffer(30, cap, DEV), amp.ReplayBuffer(30, cap, DEV)
    rb_t.step = rb_k.step = cap - 50                      # the insert wraps
    pk = PackedAmpDisc(disc, nz, n)
    rew, d, carry = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV), torch.zeros(n, 30, device=DEV)
    for rep in range(3):                                  # the workspace's counters must come back to zero: repeated launches agree
        rb_k.step, rb_k.num_samples = cap - 50, 0
        pk.step(s, nxt, dones, term, task, rew, disc_out=d, carry=carry, replay=rb_k)
    with_term = torch.where(dones.unsqueeze(1), term, nxt)
    r_ref, d_ref = disc.predict_amp_reward(s, with_term, task, normalizer=nz)
    rb_t.insert(s, with_term)
    torch.cuda.synchronize()
    torch.testing.assert_close(d, d_ref[:, 0], rtol=2e-5, atol=2e-5)
    torch.testing.assert_close(rew, r_ref, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(carry, nxt, rtol=0, atol=0)
    torch.testing.assert_close(rb_k.states, rb_t.states, rtol=0, atol=0)
    torch.testing.assert_close(rb_k.next_states, rb_t.next_states, rtol=0, atol=0)
    assert (rb_k.step, rb_k.num_samples) == (rb_t.step, rb_t.num_samples)
    assert int(pk.workspace[: (n + 31) // 32].abs().sum()) == 0


def test_amp_step_sees_refreshed_weights_and_rejects_bad_arguments():
    import ctypes
    from isaacgymloco_amd import abi, lib
    from isaacgymloco_amd.learn.fused_amp import PackedAmpDisc
    disc, nz = _disc(seed=5), _normalizer(seed=2)
    pk = PackedAmpDisc(disc, nz, 64)
    g = torch.Generator().manual_seed(1)
    s, nxt, task = torch.randn(64, 30, generator=g).to(DEV), torch.randn(64, 30, generator=g).to(DEV), torch.randn(64, generator=g).to(DEV)
    rew = torch.zeros(64, device=DEV)
    with torch.no_grad():
        for p in disc.parameters():
            p.mul_(1.3)
    nz.update(s * 2.0)                                     # re-binds the moment tensors: the kernel must read the live ones
    pk.refresh()
    pk.step(s, nxt, None, None, task, rew)
    torch.testing.assert_close(rew, disc.predict_amp_reward(s, nxt, task, normalizer=nz)[0], rtol=1e-5, atol=1e-6)
    L = lib.load()
    bad = abi.LsimAmpDisc.from_buffer_copy(pk._D)
    bad.amp_dim = 33
    args = (s.data_ptr(), nxt.data_ptr(), None, None, task.data_ptr(), 64, rew.data_ptr(), None, None, None, None, 0, 0, pk.workspace.data_ptr(),
            pk.workspace.numel() * 4, None)
    assert L.lsim_amp_step(ctypes.byref(bad), *args) == abi.E_UNSUPPORTED
    assert L.lsim_amp_step(ctypes.byref(pk._D), *args[:13], None, 0, None) == abi.E_INVALID        # no workspace
    with pytest.raises(ValueError):
        pk.step(torch.zeros(65, 30, device=DEV), torch.zeros(65, 30, device=DEV), None, None, None, torch.zeros(65, device=DEV))
