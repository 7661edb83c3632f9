"""numpy Philox4x32-10 (test infrastructure): the RNG spec of include/lsim.h, vectorised.
Used by tools/gen_golden.py to inject the simulator's uniforms into the reference, and by tests."""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint32).copy() for c in np.broadcast_arrays(c0, c1, c2, c3))
    k0 = np.uint32(k0)
    k1 = np.uint32(k1)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = M0 * c0.astype(np.uint64)
            p1 = M1 * c2.astype(np.uint64)
            hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), p0.astype(np.uint32)
            hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), p1.astype(np.uint32)
            c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
            k0 = np.uint32((int(k0) + int(W0)) & 0xFFFFFFFF)
            k1 = np.uint32((int(k1) + int(W1)) & 0xFFFFFFFF)
    return c0, c1, c2, c3


def u01(seed, rank, env, step, tag, idx):
    """float32 uniforms in [0,1) for broadcastable (env, idx) arrays."""
    env, idx = np.broadcast_arrays(np.asarray(env, dtype=np.uint32), np.asarray(idx, dtype=np.uint32))
    out = philox4x32_10(env, np.uint32(step & 0xFFFFFFFF), np.uint32(tag), idx >> np.uint32(2), seed, rank)
    sel = np.choose(idx & np.uint32(3), out)
    return (sel >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)
