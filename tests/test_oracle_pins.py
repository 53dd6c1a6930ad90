"""Numbers the reference itself holds, used to pin the oracle's literal restatement (the
reference has no test fixtures, so these are taken from its notes and data files).

  * Notes/notes.tex:1000-1027 -- "Distribution of inner-loop iterations": at z = 1.37 the probabilities that
    the alternating-series test of a proposal ends at term 1, 2, 3 are 0.9991977085, 0.0008022898,
    0.0000000017 (integrals of a_{j-1} - a_j over the normalising constant of a_0), and the expected number
    of terms per draw is 1.0016 (1.0014 empirically).
  * Code/R/d1to4.txt (loaded next to t1to4.txt at Code/R/Ch.R:7-10, 301 values on the grid h = 1, 1.01 .. 4):
    the gap between the envelope and the density at the truncation point t(h), relative to the envelope:
    d(h) = 1 - f(t | h)/a_0(t, h) = sum_{n>=1} (-1)^(n+1) a_n(t, h)/a_0(t, h), f = the alternating series.
"""
import os

import numpy as np
from scipy import integrate

HERE = os.path.dirname(os.path.abspath(__file__))
NOTES_P = (0.9991977085, 0.0008022898, 0.0000000017)
NOTES_Z = 1.37          # the J*(1, z) parameter, i.e. |z_user| / 2 (PolyaGamma.cpp:154)


def _integral_a(L, n, Z):
    f = lambda x: L.bl_pg_a(n, x) * np.exp(-0.5 * Z * Z * x)
    a, _ = integrate.quad(f, 0, 0.64, epsabs=0, epsrel=1e-13, limit=400)
    b, _ = integrate.quad(f, 0.64, np.inf, epsabs=0, epsrel=1e-13, limit=400)
    return a + b


def test_series_stop_probabilities_of_the_notes(oracle):
    """a(n, x) (PolyaGamma.cpp:41-55 restated) integrates to the notes' ten-decimal values."""
    L = oracle.lib()
    c, i1, i2, i3 = (_integral_a(L, n, NOTES_Z) for n in range(4))
    got = ((c - i1) / c, (i1 - i2) / c, (i2 - i3) / c)
    for g, want in zip(got, NOTES_P):
        assert abs(g - want) < 6e-11, (g, want)
    # expected series terms per draw: terms per proposal / acceptance probability = 1.0016 (notes.tex:1003)
    per_prop = 1 * got[0] + 2 * got[1] + 3 * got[2]
    assert abs(per_prop / got[0] - 1.0016) < 5e-5


def test_literal_loop_stops_where_the_notes_say(oracle):
    """The literal draw_like_devroye loop (PolyaGamma.cpp:167-200 restated), run: where its series test stops."""
    counts, nprop = oracle.devroye_census(2.0 * NOTES_Z, 3_000_000, seed=1370)
    p1, p2 = counts[1] / nprop, counts[2] / nprop
    assert abs(p1 - NOTES_P[0]) < 5 * np.sqrt(NOTES_P[1] / nprop)
    assert abs(p2 - NOTES_P[1]) < 5 * np.sqrt(NOTES_P[1] / nprop)
    assert counts[0] == 0 and counts[3] <= 3                       # 1.7e-9 per proposal
    assert abs(nprop / 3_000_000 - 1 / NOTES_P[0]) < 1e-4          # proposals per draw
    terms_per_draw = (counts[1] + 2 * counts[2] + 3 * counts[3]) / 3_000_000
    assert abs(terms_per_draw - 1.0016) < 1e-4                     # "1.0014 empirical / 1.0016 analytic"


def test_d1to4_is_the_series_gap_at_the_truncation_point(oracle):
    """Pins a_n(x, h) (PolyaGammaAlt.cpp:26-35 restated) on all 301 (h, t(h)) of the reference's tables, and the
    ratio recurrence the attempt-form series runs on (oracle/pg_attempt.c alt_series, bl_alt_sm.hpp)."""
    L = oracle.lib()
    h = np.loadtxt(os.path.join(HERE, "golden", "h1to4.txt"))
    t = np.loadtxt(os.path.join(HERE, "golden", "t1to4.txt"))
    d = np.loadtxt(os.path.join(HERE, "golden", "d1to4.txt"))
    assert d.shape == (301,)
    for hh, tt, dd in zip(h, t, d):
        a0 = L.bl_alt_a_coef(0, tt, hh)
        lit = sum((-1) ** (n + 1) * L.bl_alt_a_coef(n, tt, hh) for n in range(1, 40)) / a0
        assert abs(lit / dd - 1) < 2e-9, (hh, lit, dd)
        e, q2, a, gap = np.exp(-2.0 * (hh + 1.0) / tt), np.exp(-4.0 / tt), 1.0, 0.0
        for n in range(1, 40):
            a *= (n + hh - 1.0) * (2.0 * n + hh) / (n * (2.0 * n + hh - 2.0)) * e
            gap += (-1) ** (n + 1) * a
            e *= q2
        assert abs(gap / dd - 1) < 2e-9


def test_H11_normal_approximation_variance_is_ill_conditioned_at_small_z(oracle):
    """Hazard H11 (found by a randomised soak in round 3): above b = 170 rpg_hybrid draws N(pg_m1, pg_m2 - pg_m1^2)
    (LogitWrapper.cpp:143-146), and jj_m2's (tanh z - z)/z^3 (PolyaGamma.cpp:221-231) cancels for 1e-12 < z <~ 1e-3 -- the
    series branch only starts at z <= 1e-12.  The literal formula, which the oracle restates and the kernels run, loses digits
    like 1e-16/z^2: at |z| = 2e-6 the variance is off by 4e-5 relative, and WHICH way depends on the last bit of tanh, so two
    correct libms (glibc here, ocml on the GPU) give draws that differ by 1e-5 there.  Not a parity failure: the reference's own
    output is rounding noise in that window.  GPU-vs-oracle comparisons of the b > 170 class keep |z| >= 0.01."""
    import mpmath as mp
    mp.mp.dps = 50
    L = oracle.lib()
    b = 400.0
    err = {}
    for z in (2e-6, 2e-4, 2e-2, 0.2):
        m1, m2 = L.bl_pg_m1(b, z), L.bl_pg_m2(b, z)
        Z = mp.mpf(z) / 2
        j1 = b * mp.tanh(Z) / Z
        j2 = (b + 1) * b * (mp.tanh(Z) / Z) ** 2 + b * ((mp.tanh(Z) - Z) / Z ** 3)
        exact = j2 / 16 - (j1 / 4) ** 2
        err[z] = float(abs((m2 - m1 * m1) - exact) / exact)
    assert err[0.2] < 1e-12 and err[2e-2] < 1e-10              # well conditioned where the samplers' data live
    assert 1e-7 < err[2e-6] < 1e-2 and err[2e-4] > 1e-10        # the window: digits lost like 1e-16 / z^2
