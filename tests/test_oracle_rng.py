"""Oracle RNG layer: Philox known answers (Random123 kat_vectors) and the distribution of every
primitive standing in for the reference's absent RNG library, against scipy."""
import ctypes as C

import numpy as np
from scipy import stats

N = 40000


def test_philox_known_answers(oracle):
    # Random123 kat_vectors, philox4x32 10 rounds
    assert oracle.philox([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert oracle.philox([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert oracle.philox([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_stream_layout(oracle):
    L = oracle.lib()
    r = oracle.rng(seed=0x1122334455667788, idx=0x00ABCDEF01234567, domain=2, epoch=9)
    assert list(r.key) == [0x55667788, 0x11223344]
    assert list(r.ctr) == [0x01234567, 0x00ABCDEF | (2 << 24), 9, 0]
    u0 = L.bl_unif(C.byref(r))
    u1 = L.bl_unif(C.byref(r))
    w = oracle.philox([0x01234567, 0x00ABCDEF | (2 << 24), 9, 0], [0x55667788, 0x11223344])
    assert u0 == (((w[0] << 32 | w[1]) >> 12) + 0.5) * 2.0 ** -52
    assert u1 == (((w[2] << 32 | w[3]) >> 12) + 0.5) * 2.0 ** -52
    L.bl_unif(C.byref(r))
    assert r.ctr[3] == 2 and r.nunif == 3


def _draw(oracle, fn, *args, n=N, seed=5):
    L = oracle.lib()
    f = getattr(L, fn)
    r = oracle.rng(seed)
    return np.array([f(C.byref(r), *args) for _ in range(n)])


def _ks(x, cdf, alpha=1e-3):
    p = stats.kstest(x, cdf).pvalue
    assert p > alpha, p


def test_unif_expon_norm_flat(oracle):
    u = _draw(oracle, "bl_unif")
    assert u.min() > 0 and u.max() < 1
    _ks(u, stats.uniform.cdf)
    _ks(_draw(oracle, "bl_expon_rate", 2.5), stats.expon(scale=1 / 2.5).cdf)
    _ks(_draw(oracle, "bl_norm", 1.0, 3.0), stats.norm(1.0, 3.0).cdf)
    _ks(_draw(oracle, "bl_flat", -2.0, 5.0), stats.uniform(-2.0, 7.0).cdf)


def test_gamma(oracle):
    for a in (0.3, 1.0, 2.5, 40.0):
        _ks(_draw(oracle, "bl_gamma_scale", a, 2.0, seed=int(a * 10)), stats.gamma(a, scale=2.0).cdf)


def test_igauss(oracle):
    for mu, lam in ((1.0, 1.0), (0.3, 4.0), (2.0, 16.0)):
        _ks(_draw(oracle, "bl_igauss", mu, lam), stats.invgauss(mu / lam, scale=lam).cdf)


def test_ltgamma(oracle):
    for a, rate, t in ((1.0, 2.0, 0.64), (2.5, 1.3, 1.0), (4.0, 2.0, 4.13), (20.0, 25.0, 1.1)):
        g = stats.gamma(a, scale=1 / rate)
        x = _draw(oracle, "bl_ltgamma", a, rate, t)
        assert x.min() >= t
        _ks(x, lambda v: (g.cdf(v) - g.cdf(t)) / g.sf(t))


def test_rtinvchi2(oracle):
    for scale, t in ((1.0, 0.64), (16.0, 4.13), (20.0, 1.1)):
        x = _draw(oracle, "bl_rtinvchi2", scale, t)
        assert x.max() <= t and x.min() > 0
        # X = scale / chi2_1 | X <= t  <=>  chi2_1 >= scale/t
        c = stats.chi2(1)
        _ks(x, lambda v: c.sf(scale / np.maximum(v, 1e-300)) / c.sf(scale / t))


def test_tnorm(oracle):
    inf = float("inf")
    cases = [(-inf, inf), (0.5, inf), (3.0, inf), (-inf, -1.0), (-0.3, 0.4), (-1.0, 3.0), (2.0, 2.3), (2.0, 6.0),
             (-5.0, -4.5), (-8.0, -2.0), (-inf, 1.0)]
    for lo, hi in cases:
        x = _draw(oracle, "bl_tnorm", lo, hi, n=20000)
        assert x.min() >= lo and x.max() <= hi
        a, b = stats.norm.cdf(lo), stats.norm.cdf(hi)
        if b - a > 1e-12:
            _ks(x, lambda v: (stats.norm.cdf(v) - a) / (b - a))
    # a call always consumes exactly nine uniforms, whatever the bounds (stream never desynchronises)
    for lo, hi in cases + [(0.7, 0.7), (0.7, 0.7 + 1e-15), (1.0, 0.5), (40.0, inf)]:
        r = oracle.rng(1)
        x = oracle.lib().bl_tnorm(C.byref(r), lo, hi)
        assert r.nunif == 9
        if hi <= lo:
            assert x == lo
    x = _draw(oracle, "bl_tnorm", 40.0, inf, n=2000)
    assert x.min() >= 40.0 and abs(x.mean() - 40.0 - 1 / 40.0) < 2e-3


def test_qnorm_as241(oracle):
    L = oracle.lib()
    L.bl_qnorm.restype = C.c_double
    L.bl_qnorm.argtypes = [C.c_double]
    ps = np.concatenate([10.0 ** np.linspace(-300, -1.2, 500), np.linspace(0.07, 0.93, 500)])
    q = np.array([L.bl_qnorm(p) for p in ps])
    assert np.allclose(q, stats.norm.ppf(ps), rtol=3e-15, atol=1e-15)


def test_call_sequences_never_share_a_stream(oracle):
    """set_seed(s); gibbs(); rpg_devroye(); ...: the chain started by the k-th gibbs() call reads streams
    (chain_key(s, k), DOM_OMEGA / DOM_BETA, sweep), the j-th rpg_* call (s, DOM_DRAW, j).  No (key, counter) pair of
    one is a pair of the other: the domains differ, and the derived key is a Philox output, not the seed."""
    s = 0x42A7E5105EEDB00F
    keys = [oracle.chain_key(s, k) for k in range(6)]
    assert len(set(keys + [s])) == 7
    L = oracle.lib()
    seen = set()
    for key, dom in [(s, 0)] + [(k, d) for k in keys for d in (1, 3)] + [(s, 3), (s, 1)]:
        for epoch in range(4):
            r = oracle.rng(key, 5, dom, epoch)
            u = L.bl_unif(C.byref(r))
            assert u not in seen
            seen.add(u)
    # chain_key is the documented Philox block: words (0, 1) of Philox(ctr = (call, DOM_KEY << 24, 0, 0), key = seed)
    o = oracle.philox([3, 4 << 24, 0, 0], [s & 0xFFFFFFFF, s >> 32])
    assert oracle.chain_key(s, 3) == (o[1] << 32) | o[0]
