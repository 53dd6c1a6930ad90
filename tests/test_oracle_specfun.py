"""Oracle special functions vs scipy known answers (they stand in for R's pnorm/pgamma, Appendix B)."""
import numpy as np
from scipy import special, stats


def test_p_norm(oracle):
    L = oracle.lib()
    for x in np.concatenate([np.linspace(-60, 10, 281), [-36.9999, -37.0, -37.0001, 0.0]]):
        assert np.isclose(L.bl_p_norm(x, 1), stats.norm.logcdf(x), rtol=2e-14, atol=1e-300)
        assert np.isclose(L.bl_p_norm(x, 0), stats.norm.cdf(x), rtol=1e-13, atol=1e-300)


def test_p_gamma_rate(oracle):
    L = oracle.lib()
    for a in (0.5, 1.0, 2.5, 4.0, 14.0, 50.0, 170.0):
        for x in (1e-3, 0.1, 0.64, 1.0, 4.13, 10.0, 100.0, 400.0):
            for rate in (0.5, 1.2337, 7.0):
                ref = special.gammainc(a, rate * x)
                got = L.bl_p_gamma_rate(x, a, rate)
                assert np.isclose(got, ref, rtol=1e-12, atol=1e-15), (a, x, rate, got, ref)
    assert L.bl_p_gamma_rate(0.0, 2.0, 1.0) == 0.0


def test_p_igauss(oracle):
    L = oracle.lib()
    for mu, lam in ((1.0, 1.0), (0.2, 14.0), (3.0, 50.0), (0.05, 170.0)):
        d = stats.invgauss(mu / lam, scale=lam)
        for x in (0.01, 0.1, 0.64, 1.1, 3.0):
            assert np.isclose(L.bl_p_igauss(x, mu, lam), d.cdf(x), rtol=1e-9, atol=1e-14), (mu, lam, x)
