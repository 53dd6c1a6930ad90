"""Oracle PG samplers against what the reference's own tests and data hold for this path
(SURVEY.md 8c): closed-form moments pg_m1/pg_m2 next to sample moments
(test_pgomp.cpp:56-62, test_hybrid_par.cpp:55-59, LogitTest.R:89-104), the p/q constants of
lambda-jacobi.R:7-8, the exact PG(1,z) CDF of PG.R:320-348, and cross-sampler agreement."""
import numpy as np
import pytest
from scipy import stats

import pgmath


def test_mass_texpon_known_values(oracle):
    L = oracle.lib()
    p, q = 0.57810262346829443, 0.422599094            # Code/R/lambda-jacobi.R:7-8
    assert abs(L.bl_pg_mass_texpon(0.0) - p / (p + q)) < 1e-8
    assert abs(L.bl_pg_mass_texpon(1.0) - 0.4605903) < 1e-7
    assert abs(L.bl_pg_mass_texpon(2.0) - 0.2305365) < 1e-7
    assert L.bl_pg_mass_texpon(100.0) == 0.0                      # right piece vanishes, no NaN


def test_series_coefficients(oracle):
    L = oracle.lib()
    # the two forms of a_n (PolyaGamma.cpp:46-51) are the SAME function summed over n:
    # sum_n (-1)^n a_n(x) is the J*(1,0) density, continuous at x = 0.64
    for x in (0.05, 0.3, 0.64, 0.6400001, 1.0, 3.0):
        s = sum((-1) ** n * L.bl_pg_a(n, x) for n in range(40))
        assert np.isclose(s, pgmath.jstar_density(np.array([x]))[0], rtol=1e-9)
    assert L.bl_pg_a(0, 0.0) == 0.0 and L.bl_pg_a(3, -1.0) == 0.0
    # a_coef(n, x, h=1) of the alt sampler equals a(n, x) of the Devroye sampler (left form)
    for n in range(5):
        for x in (0.1, 0.4, 0.64):
            assert np.isclose(L.bl_alt_a_coef(n, x, 1.0), L.bl_pg_a(n, x), rtol=1e-12)


def test_closed_form_moments(oracle):
    L = oracle.lib()
    for b in (0.5, 1.0, 3.7, 50.0, 171.0):
        for z in (0.0, 1e-13, 0.3, 2.0, 10.0, -2.0):
            m1, m2 = L.bl_pg_m1(b, z), L.bl_pg_m2(b, z)
            assert np.isclose(m1, pgmath.pg_mean(b, z), rtol=1e-12)
            if abs(z) > 1e-3 or z == 0.0:
                assert np.isclose(m2 - m1 * m1, pgmath.pg_var(b, z), rtol=1e-6)


def test_C1_rpg_1e3_ks(oracle):
    """BASELINE config C1: rpg(num=1e3, h=1, z=0): KS vs the exact PG(1,0) CDF, mean 1/4, var 1/24."""
    x = oracle.rpg_hybrid(1000, 1.0, 0.0, seed=2024)
    assert stats.kstest(x, lambda w: pgmath.pg1_cdf(w, 0.0)).pvalue > 0.01
    assert abs(x.mean() - 0.25) < 4 * np.sqrt(1 / 24 / 1000)
    assert abs(x.var() - 1 / 24) < 4 * np.sqrt(2 * (1 / 24) ** 2 * 3 / 1000)


@pytest.mark.parametrize("z", [0.0, 0.5, 1.37, 3.0, 3.2, 8.0, -4.0, 40.0])
def test_devroye_exact_cdf(oracle, z):
    x = oracle.rpg_devroye(40000, 1, z, seed=int(abs(z) * 100) + 1)
    assert stats.kstest(x, lambda w: pgmath.pg1_cdf(w, z)).pvalue > 1e-3
    assert np.all(x > 0)


@pytest.mark.parametrize("z", [0.0, 0.5, 1.37, 3.0, 3.2, 8.0, -4.0, 40.0])
def test_literal_reference_loops_exact_cdf(oracle, z):
    """The call-for-call restatement of PolyaGamma.cpp:82-115,151-202 against the exact CDF."""
    x = oracle.rpg_devroye(40000, 1, z, seed=int(abs(z) * 100) + 11, literal=True)
    assert stats.kstest(x, lambda w: pgmath.pg1_cdf(w, z)).pvalue > 1e-3
    assert np.all(x > 0)


@pytest.mark.parametrize("z", [0.0, 1.0, 2.5, 3.1, 3.13, 4.5, 12.0])
def test_attempt_form_equals_literal_form_in_distribution(oracle, z):
    """bl_pg1_attempt (two uniforms per attempt, recycled: what the HIP kernels run) and the literal
    reference loops: two-sample KS at 4e5 draws each (resolves CDF differences of ~3e-3), equal first two
    moments within sampling error, and both within sampling error of the closed-form mean."""
    n = 400000
    a = oracle.rpg_devroye(n, 1, z, seed=1000 + int(z * 10))
    b = oracle.rpg_devroye(n, 1, z, seed=2000 + int(z * 10), literal=True)
    assert stats.ks_2samp(a, b).pvalue > 1e-3
    var = pgmath.pg_var(1, z)
    se = np.sqrt(var / n)
    assert abs(a.mean() - pgmath.pg_mean(1, z)) < 5 * se and abs(b.mean() - pgmath.pg_mean(1, z)) < 5 * se
    assert abs(a.var() - b.var()) < 5 * np.sqrt(2 * 12 * var * var / n)
    # n > 1 sums use the same attempts back to back
    a3 = oracle.rpg_devroye(100000, 3, z, seed=3000 + int(z * 10))
    b3 = oracle.rpg_devroye(100000, 3, z, seed=4000 + int(z * 10), literal=True)
    assert stats.ks_2samp(a3, b3).pvalue > 1e-3


MOMENT_GRID = [(1, 0.0), (1, 2.0), (2, 1.0), (2, 6.0), (1.01, 0.7), (3.0, 0.0), (3.5, 1.0), (4.0, 2.0), (4.5, 0.5),
               (5.0, 3.0), (7.5, 0.5), (9.0, 1.0), (12.99, 4.0), (13.0, 0.1), (13.5, 1.0), (14.0, 1.0), (20.0, 0.0),
               (50.0, 2.0), (100.0, 5.0), (170.0, 0.1), (171.0, 1.0), (400.0, 3.0), (0.5, 1.0), (0.1, 0.0)]


@pytest.mark.parametrize("literal", [True, False])
@pytest.mark.parametrize("b,z", MOMENT_GRID)
def test_hybrid_sample_moments(oracle, b, z, literal):
    """The reference's own criterion: sample m1, m2 next to pg_m1, pg_m2 (test_hybrid_par.cpp:55-59), for the
    reference's loops (literal) and for the attempt forms the HIP path computes."""
    n = 3000 if b < 1 else 60000
    x = oracle.rpg_hybrid(n, b, z, seed=int(b * 1000 + z * 10) + 7, literal=literal)
    L = oracle.lib()
    m1, m2 = L.bl_pg_m1(b, z), L.bl_pg_m2(b, z)
    var = m2 - m1 * m1
    assert abs(x.mean() - m1) < 5 * np.sqrt(var / n)
    # variance of the sample variance ~ (kurt-1) var^2 / n; PG kurtosis < 12
    assert abs(x.var() - var) < 5 * np.sqrt(12 * var * var / n) + 2e-3 * var   # T=200 truncation bias for b<1
    assert x.min() > 0


def test_methods_agree_in_distribution(oracle):
    """LogitTest.R:89-104 compares rpg / rpg.devroye / rpg.R moments; here two-sample KS."""
    n = 20000
    z = 1.3
    d2 = oracle.rpg_devroye(n, 2, z, seed=1)
    a2 = oracle.rpg_alt(n, 2.0, z, seed=2, literal=True)
    assert stats.ks_2samp(d2, a2).pvalue > 1e-3
    d1 = oracle.rpg_devroye(n, 1, z, seed=3)
    a1 = oracle.rpg_alt(n, 1.0, z, seed=4, literal=True)
    assert stats.ks_2samp(d1, a1).pvalue > 1e-3
    a14 = oracle.rpg_alt(n, 14.0, z, seed=5, literal=True)
    s14, it = oracle.rpg_sp(n, 14.0, z, seed=6, literal=True)
    assert stats.ks_2samp(a14, s14).pvalue > 1e-3
    assert it.min() >= 1 and it.max() <= 200 and it.mean() < 2.0
    g3 = oracle.rpg_gamma(4000, 3.0, z, seed=7)
    a3 = oracle.rpg_alt(n, 3.0, z, seed=8, literal=True)
    assert stats.ks_2samp(g3, a3).pvalue > 1e-3


def test_inverty(oracle):
    L = oracle.lib()
    for y in np.concatenate([2.0 ** np.linspace(-5, 5, 101), [1.0, 0.0625, 16.0]]):
        v = L.bl_v_eval(y)
        if 0.0625 <= y <= 16.0:
            assert abs(L.bl_y_eval(v) - y) < 5e-6 * max(1.0, y)     # bracket table has 7 digits; Newton is clamped to it
        elif y < 0.0625:
            assert v == -1.0 / (y * y)                               # InvertY.cpp:63
        else:
            assert np.isclose(v, np.arctan(0.5 * np.pi * y) ** 2)    # InvertY.cpp:65-66
    assert L.bl_v_eval(1.0) == 0.0
    # H5: the series branch evaluates to exactly 1 (integer-division literals are 0)
    assert L.bl_y_eval(1e-9) == 1.0 and L.bl_sp_y_func(-5e-7) == 1.0


def test_edge_cases(oracle):
    # n = 0 / h = 0 give 0 (LogitWrapper.cpp:74-77,95-98); n < 1 clamps to 1 in the NTHROW build
    assert oracle.rpg_devroye(3, [0, 0, 0], 1.0, 1).tolist() == [0, 0, 0]
    for lit in (True, False):
        assert oracle.rpg_alt(2, 0.0, 1.0, 1, literal=lit).tolist() == [0, 0]
        assert oracle.rpg_hybrid(2, -1.0, 1.0, 1, literal=lit).tolist() == [0, 0]
        assert oracle.rpg_alt(1, 0.5, 1.0, 1, literal=lit)[0] == 0.0   # h < 1 -> 0, PolyaGammaAlt.cpp:207-210
    # sign of z is irrelevant, per-observation streams make results order-independent
    a = oracle.rpg_devroye(100, 1, 2.5, seed=9)
    b = oracle.rpg_devroye(100, 1, -2.5, seed=9)
    assert np.array_equal(a, b)
    for lit in (True, False):
        full = oracle.rpg_hybrid(1000, 3.0, 1.0, seed=4, literal=lit)
        part = oracle.rpg_hybrid(400, 3.0, 1.0, seed=4, idx0=600, literal=lit)
        assert np.array_equal(full[600:], part)
    omp = oracle.rpg_hybrid(1000, 3.0, 1.0, seed=4, threads=4, literal=True)
    assert np.array_equal(full, oracle.rpg_hybrid(1000, 3.0, 1.0, seed=4)) and np.array_equal(
        omp, oracle.rpg_hybrid(1000, 3.0, 1.0, seed=4, literal=True))
