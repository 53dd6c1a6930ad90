"""The alternating-series and saddle-point samplers in the attempt form the HIP kernels execute
(oracle/pg_attempt.c) against the literal restatement of the reference's loops (oracle/pg_alt.c, pg_sp.c):
same distribution, and every quantity the attempt form evaluates by a different formula compared value for
value with the literal one."""
import ctypes as C

import numpy as np
import pytest
from scipy import stats

ALT_GRID = [(1.0, 0.0), (1.0, 3.5), (1.37, 0.7), (2.0, 2.0), (2.5, 6.0), (3.0, 0.0), (3.99, 1.0), (4.0, 2.0),
            (4.0, 9.0), (5.0, 1.0), (8.6, 0.3), (13.0, 2.5), (41.3, 1.0)]
SP_GRID = [(14.0, 0.0), (14.0, 3.0), (20.0, 1.0), (33.3, 7.0), (50.0, 2.0), (100.0, 0.5), (170.0, 5.0), (1.0, 1.0),
           (3.5, 0.2), (60.0, 9.0)]


@pytest.mark.parametrize("h,z", ALT_GRID)
def test_alt_attempt_form_has_the_literal_distribution(oracle, h, z):
    n = 60000
    a = oracle.rpg_alt(n, h, z, seed=int(100 * h + z) + 1)
    l = oracle.rpg_alt(n, h, z, seed=int(100 * h + z) + 2, literal=True)
    assert stats.ks_2samp(a, l).pvalue > 1e-3
    L = oracle.lib()
    m1 = L.bl_pg_m1(h, z)
    var = L.bl_pg_m2(h, z) - m1 * m1
    assert abs(a.mean() - m1) < 5 * np.sqrt(var / n)
    assert abs(a.var() - var) < 5 * np.sqrt(12 * var * var / n)


@pytest.mark.parametrize("h,z", SP_GRID)
def test_sp_attempt_form_has_the_literal_distribution(oracle, h, z):
    n = 60000
    a, ia = oracle.rpg_sp(n, h, z, seed=int(100 * h + z) + 3)
    l, il = oracle.rpg_sp(n, h, z, seed=int(100 * h + z) + 4, literal=True)
    assert stats.ks_2samp(a, l).pvalue > 1e-3
    # the iteration count (rpg.sp's track.iter, LogitWrapper.cpp:117) has the same law
    assert abs(ia.mean() - il.mean()) < 5 * np.sqrt((ia.var() + il.var()) / n) + 1e-9
    assert ia.min() >= 1 and ia.max() <= 200


def test_sp_right_piece_where_the_literal_weight_cancels(oracle):
    """Where Q(n, n rr md) < 1e-16 (large n with large |z|: here n = 170, z = 12, Q ~ 1e-20) the reference's
    `1.0 - p_gamma_rate(...)` (PolyaGammaSP.cpp:222) cancels to 0, wr = 0, and its sampler never proposes right of
    md = 1.1 x the mode, although the saddle-point density has ~1 % of its envelope there.  The attempt form
    evaluates Gamma(n) Q by a continued fraction and keeps the piece (hazard H9)."""
    n, z, N = 170.0, 12.0, 200000
    Z = 0.5 * z
    md = 1.1 * np.tanh(Z) / Z
    lit, il = oracle.rpg_sp(N, n, z, seed=5, literal=True)
    att, ia = oracle.rpg_sp(N, n, z, seed=6)
    assert (lit > n * 0.25 * md).sum() == 0
    frac = (att > n * 0.25 * md).mean()
    assert 1e-3 < frac < 2e-2
    # left of md the two agree in distribution
    assert stats.ks_2samp(att[att <= n * 0.25 * md], lit).pvalue > 1e-3


def test_blocks_per_draw(oracle):
    """Work per draw in Philox blocks (quoted in DESIGN.md)."""
    n = 20000
    _, nb = oracle.rpg_alt(n, 4.0, 1.0, 5, blocks=True)
    assert 1.5 < nb.mean() < 2.4
    _, _, nb = oracle.rpg_sp(n, 30.0, 1.0, 5, blocks=True)
    assert 1.2 < nb.mean() < 1.8


def test_alt_mixture_weight_matches_literal(oracle):
    """prob_right of the attempt form's set-up against w_right/(w_right + w_left) of the literal functions
    (PolyaGammaAlt.cpp:60-75, :129-131)."""
    L = oracle.lib()

    class AltPar(C.Structure):
        _fields_ = [(k, C.c_double) for k in ("h", "Z", "t", "fz", "p", "ip", "iq", "R", "b", "ic0", "omc", "log_m", "cR")] + [("small", C.c_int)]
    L.bl_alt_par_of.argtypes = [C.POINTER(AltPar), C.c_double, C.c_double]
    L.bl_alt_par_of.restype = None
    for h in (1.0, 1.5, 2.0, 2.75, 3.0, 4.0):
        for z in (0.0, 0.3, 1.0, 2.5, 6.0, 15.0):
            p = AltPar()
            L.bl_alt_par_of(C.byref(p), h, z)
            wl, wr = L.bl_alt_w_left(p.t, h, p.Z), L.bl_alt_w_right(p.t, h, p.Z)
            lit = wr / (wr + wl)
            assert abs(p.p - lit) <= 2e-11 * lit + 1e-13, (h, z, p.p, lit)     # the literal 1 - P (P to 1e-16) bounds the agreement


def test_fitted_v_against_newton_and_identity(oracle):
    """v(x), -log cos_rt(v), log K2 from the fitted table against InvertY.cpp's Newton solve (tolerance 1e-9,
    :57-99) and against the defining identities at full precision."""
    L = oracle.lib()
    for x in np.concatenate([2.0 ** np.linspace(-3.99, 3.99, 801), [0.999999, 1.000001, 1.0]]):
        v, Lc, lK2 = oracle.sp_vlk(float(x))
        assert abs(v - L.bl_v_eval(x)) <= 3e-9 * max(1.0, abs(v)) + 2e-9
        if abs(v) > 1e-6:
            r = np.sqrt(abs(v))
            y = np.tan(r) / r if v > 0 else np.tanh(r) / r
            assert abs(y - x) < 3e-13 * x * max(1.0, abs(v))                     # x = tan(sqrt v)/sqrt v
            assert abs(Lc + np.log(np.cos(r) if v > 0 else np.cosh(r))) < 2e-13 * max(1.0, abs(Lc))
            assert abs(lK2 - np.log(x * x + (1 - x) / v)) < 3e-10 / min(1.0, abs(v))   # (1 - x)/v: 0/0 near x = 1
        else:
            assert lK2 == 2.0 * np.log(x)                                        # H5: K2 = x^2 there
    # outside [2^-4, 2^4]: the reference's asymptotic forms
    for x in (0.01, 0.06, 17.0, 300.0):
        v, _, _ = oracle.sp_vlk(x)
        assert v == L.bl_v_eval(x)


def test_sp_setup_matches_literal(oracle):
    """The attempt form's tangent lines and mixture probability against the literal set-up of
    PolyaGammaSP.cpp:171-229 (whose Newton tolerance and 1 - P cancellation bound the agreement)."""
    L = oracle.lib()

    class SpPar(C.Structure):
        _fields_ = [(k, C.c_double) for k in ("n", "Z2", "md", "logmd", "lcZ", "lhal", "lhar", "rl", "il", "rr", "ir",
                                              "mu", "pl", "ipl", "iql", "b", "ic0", "omc", "log_m")]
    L.bl_sp_par_of.argtypes = [C.POINTER(SpPar), C.c_double, C.c_double]
    L.bl_sp_par_of.restype = None
    L.tgamma = None
    from math import gamma, log, sqrt, exp, pi
    for n in (14.0, 20.0, 35.5, 60.0):
        for z in (0.0, 0.001, 0.4, 1.0, 3.0, 8.0):
            p = SpPar()
            L.bl_sp_par_of(C.byref(p), n, z)
            Z = 0.5 * abs(z)
            xl = L.bl_sp_y_func(-Z * Z)
            md, xr = 1.1 * xl, 1.2 * xl
            sl, il, sr, ir = (C.c_double() for _ in range(4))
            L.bl_sp_tangent_to_eta(xl, Z, md, C.byref(sl), C.byref(il))
            L.bl_sp_tangent_to_eta(xr, Z, md, C.byref(sr), C.byref(ir))
            assert abs(p.rl + sl.value) < 2e-8 * abs(sl.value) and abs(p.il - il.value) < 2e-8 * max(1, abs(il.value))
            assert abs(p.rr + sr.value) < 2e-8 * abs(sr.value) and abs(p.ir - ir.value) < 2e-8 * max(1, abs(ir.value))
            vmd = L.bl_v_eval(md)
            K2 = md * md + (1 - md) / vmd
            al, ar = md ** 3 / K2, md * md / K2
            assert abs(p.lhal - 0.5 * log(al)) < 1e-8 and abs(p.lhar - 0.5 * log(ar)) < 1e-8
            rl, rr = -sl.value, -sr.value
            wl = exp(0.5 * log(al) - n * sqrt(2 * rl) + n * il.value + 0.5 * n / md) * L.bl_p_igauss(md, 1 / sqrt(2 * rl), n)
            wr = exp(0.5 * log(ar) + 0.5 * log(0.5 * n / pi) - n * log(n * rr) + n * ir.value - n * log(md)) * gamma(n) * (
                1.0 - L.bl_p_gamma_rate(md, n, n * rr))
            lit = wl / (wl + wr)
            assert abs(p.pl - lit) < 3e-6 * n, (n, z, p.pl, lit)
