"""Philox4x32-10: Random123 known-answer vectors for the numpy spec, and oracle C == numpy on the draw function."""
import ctypes

import numpy as np

import philox_np


def test_known_answer_vectors():
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, out in kat:
        got = philox_np.philox4x32_10(*ctr, *key)
        assert tuple(int(x) for x in got) == out


def test_oracle_draws_equal_numpy_spec():
    from helpers import make_oracle, quiet_cfg, abi
    cfg = quiet_cfg()
    cfg.domain_rand.delay = True
    sim, lc, _, _ = make_oracle(cfg, 32, seed=11)
    sim.reset_all()
    sim.step(np.zeros((32, 12), np.float32))
    want = (philox_np.u01(11, 0, np.arange(32), 1, abi.RNG_TAGS["delay"], 0) * 4).astype(np.int32)
    np.testing.assert_array_equal(sim.buf["delay_steps"], want)
    u = philox_np.u01(11, 0, np.arange(32)[:, None], 0xFFFFFFFF, abi.RNG_TAGS["init"], np.arange(12)[None, :])
    assert u.min() >= 0.0 and u.max() < 1.0
