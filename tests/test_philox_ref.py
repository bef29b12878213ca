"""CPU: the numpy restatement of the noise stream against the published Philox4x32-10 known answers
(Random123 kat_vectors: philox4x32 10)."""
import numpy as np

import gj_philox_ref as P


def _kat(ctr, key):
    out = P.philox4x32_10(*[np.array([c], dtype=np.uint32) for c in ctr], key[0], key[1])
    return [int(x[0]) for x in out]


def test_random123_known_answers():
    assert _kat((0, 0, 0, 0), (0, 0)) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert _kat((0xffffffff,) * 4, (0xffffffff, 0xffffffff)) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert _kat((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0)) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_uniforms_and_exponentials():
    x = np.array([0, 511, 512, 0xFFFFFFFF], dtype=np.uint32)
    u = P.u01(x)
    assert u.dtype == np.float32 and (u > 0).all() and (u < 1).all() and u[0] == u[1] and u[2] > u[1]
    assert u[3] == np.float32(1.0) - np.float32(2.0 ** -24) and np.isfinite(-np.log(u)).all() and (-np.log(u) > 0).all()
    # every value is exact: (k + 0.5) * 2^-23 needs 24 significant bits at most
    k = np.array([0, 1, 2 ** 22, 2 ** 23 - 1], dtype=np.uint32)
    assert np.array_equal(P.u01(k << np.uint32(9)).astype(np.float64), (k.astype(np.float64) + 0.5) * 2.0 ** -23)
    e0, e1 = P.exp_pair(7, 3, np.arange(200_000))
    assert e0.dtype == np.float32 and abs(e0.mean() - 1.0) < 0.01 and abs(e1.mean() - 1.0) < 0.01
    assert abs(np.corrcoef(e0, e1)[0, 1]) < 0.01                  # two iid Exponential(1) draws, in law
    assert abs(e0.var() - 1.0) < 0.03 and abs(e1.var() - 1.0) < 0.03
    theta = P.infection_uniform(7, 3, np.arange(200_000))
    assert (theta > 0).all() and (theta < 1).all() and abs(theta.mean() - 0.5) < 0.005
    assert np.allclose(e0 / (e0 + e1), theta, rtol=3e-7)          # the backward's draws give the forward's decision
    a, b = P.exp_pair(7, 3, np.array([8, 9, 10, 11]))            # one block serves four agents: different words
    assert len(set(P.infection_uniform(7, 3, np.array([8, 9, 10, 11])).tolist())) == 4 and len(set(a.tolist())) == 4
    assert not np.array_equal(P.infection_uniform(7, 4, np.arange(100)), theta[:100])      # the step is part of the counter
