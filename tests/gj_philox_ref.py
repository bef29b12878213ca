"""numpy restatement of the library's noise stream (csrc/gj_device.h), test infrastructure only:
Philox4x32-10 (Salmon et al., SC'11; Random123), key = seed; u01 = ((x >> 9) + 0.5) * 2^-23 (exact in fp32, never 0
or 1).  Forward: theta = u01(word a & 3 of block (a >> 2, step)), infected iff p < theta.  Backward: the pair of
Exponential(1) draws (theta * s, (1 - theta) * s), s ~ Gamma(2, 1) from block (a >> 1, step | 2^62)."""
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


def _block(seed: int, step: int, ctr):
    ctr = np.asarray(ctr, dtype=np.uint64)
    n = len(ctr)
    return philox4x32_10(ctr.astype(np.uint32), (ctr >> np.uint64(32)).astype(np.uint32),
                         np.full(n, step & 0xFFFFFFFF, np.uint32), np.full(n, (step >> 32) & 0xFFFFFFFF, np.uint32),
                         seed & 0xFFFFFFFF, seed >> 32)


def infection_uniform(seed: int, step: int, agents):
    """theta of the global agent ids ``agents`` at (seed, step): one block serves agents 4k .. 4k+3, agent a takes word
    a & 3 (infection_uniform() of gj_device.h).  The forward decision is p < theta."""
    agents = np.asarray(agents, dtype=np.uint64)
    r = np.stack(_block(seed, step, agents >> np.uint64(2)))
    return u01(r[(agents & np.uint64(3)).astype(np.int64), np.arange(len(agents))])


def exp_pair(seed: int, step: int, agents):
    """(e0, e1) float32: e0 = theta * s, e1 = (1 - theta) * s with s = -log(u1) - log(u2) ~ Gamma(2, 1) from a second
    block (counter = agent >> 1, stream bit 62 set): two iid Exponential(1) draws in law, with e0 / (e0 + e1) == theta
    (exp_pair() of gj_device.h: what the backward pass feeds the softmax derivative)."""
    agents = np.asarray(agents, dtype=np.uint64)
    theta = infection_uniform(seed, step, agents)
    r = _block(seed, step | (1 << 62), agents >> np.uint64(1))
    odd = (agents & np.uint64(1)).astype(bool)
    u1, u2 = u01(np.where(odd, r[2], r[0])), u01(np.where(odd, r[3], r[1]))
    s = -np.log(u1) - np.log(u2)
    return theta * s, (np.float32(1.0) - theta) * s
