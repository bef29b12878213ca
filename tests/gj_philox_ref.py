"""numpy restatement of the library's noise stream (csrc/gj_device.h), test infrastructure only:
Philox4x32-10 (Salmon et al., SC'11; Random123), counter = (agent >> 1, step), key = seed; the even agent of a pair
takes words 0-1, the odd one words 2-3; u01 = ((x >> 9) + 0.5) * 2^-23 (exact in fp32, never 0 or 1);
Exponential(1) draw = -log(u01)."""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised over uint32 arrays; returns the four output words."""
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint32).copy() for c in (c0, c1, c2, c3))
    k0, k1 = np.uint32(k0), np.uint32(k1)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = M0 * c0.astype(np.uint64)
            p1 = M1 * c2.astype(np.uint64)
            hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), p0.astype(np.uint32)
            hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), p1.astype(np.uint32)
            c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
            k0, k1 = np.uint32(k0 + W0), np.uint32(k1 + W1)
    return c0, c1, c2, c3


def u01(x):
    return ((x >> np.uint32(9)).astype(np.float32) + np.float32(0.5)) * np.float32(1.1920928955078125e-7)


def exp_pair(seed: int, step: int, agents):
    """(e0, e1) float32 of the global agent ids ``agents`` at (seed, step): exp_pair() of gj_device.h."""
    agents = np.asarray(agents, dtype=np.uint64)
    ctr = agents >> np.uint64(1)
    r = philox4x32_10(ctr.astype(np.uint32), (ctr >> np.uint64(32)).astype(np.uint32),
                      np.full(len(agents), step & 0xFFFFFFFF, np.uint32), np.full(len(agents), step >> 32, np.uint32),
                      seed & 0xFFFFFFFF, seed >> 32)
    odd = (agents & np.uint64(1)).astype(bool)
    w0 = np.where(odd, r[2], r[0])
    w1 = np.where(odd, r[3], r[1])
    return -np.log(u01(w0)), -np.log(u01(w1))
