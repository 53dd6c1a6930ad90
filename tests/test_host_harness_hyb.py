"""CPU-side unit tests of the kernels' alternating-series and saddle-point bodies: the very headers the HIP
kernels inline (bayeslogit_amd/csrc/bl_alt_sm.hpp, bl_sp_sm.hpp; host+device portable) compiled as plain C++
and driven task by task must reproduce the oracle's attempt forms draw for draw on the same Philox streams, and
their set-up algebra (erfcx forms of the mixture weights, continued fraction, fitted v(x)) the literal formulas.
(Scaffolding: the host object never ships.)"""
import ctypes as C
import glob
import math
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "host_harness", "hyb_sm_host.cpp")
LIB = os.path.join(HERE, "host_harness", "libhyb_sm_host.so")


def _build_hh(rebuild=True):
    """The host build of the kernels' headers.  rebuild=False (gpu-marked tests: a process that has touched the GPU
    starts no compiler): load what the CPU suite / the snapshot left, or None."""
    hdrs = glob.glob(os.path.join(HERE, "..", "bayeslogit_amd", "csrc", "*.hpp"))
    if rebuild and (not os.path.exists(LIB) or any(os.path.getmtime(f) > os.path.getmtime(LIB) for f in hdrs + [SRC])):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wno-unknown-pragmas", "-o", LIB, SRC, "-lm"])
    if not os.path.exists(LIB):
        return None
    H = C.CDLL(LIB)
    H.hh_cf.restype = C.c_double
    H.hh_cf.argtypes = [C.c_double, C.c_double]
    H.hh_alt_par.argtypes = [C.c_double, C.c_double, C.POINTER(C.c_double)]
    H.hh_sp_par.argtypes = [C.c_double, C.c_double, C.POINTER(C.c_double)]
    H.hh_sp_vlk.argtypes = [C.c_double, C.POINTER(C.c_double)]
    return H


@pytest.fixture(scope="module")
def hh():
    return _build_hh()


def _run_alt(H, O, h, z, seed, epoch=0, idx0=0):
    n = len(h)
    x = np.zeros(n)
    st = C.c_int(0)
    H.hh_rpg_alt(O.dp(x), O.dp(h), O.dp(z), C.c_long(n), C.c_ulonglong(seed), C.c_uint(epoch), C.c_ulonglong(idx0), C.byref(st))
    return x, st.value


def _run_sp(H, O, h, z, seed, epoch=0, idx0=0):
    n = len(h)
    x = np.zeros(n)
    it = np.full(n, -7, dtype=np.int32)
    st = C.c_int(0)
    H.hh_rpg_sp(O.dp(x), O.dp(h), O.dp(z), C.c_long(n), O.ip(it), C.c_ulonglong(seed), C.c_uint(epoch), C.c_ulonglong(idx0),
                C.byref(st))
    return x, it, st.value


def _rel(x, xo):
    r = np.abs(x - xo) / np.maximum(np.abs(xo), 1e-300)
    r[xo == 0] = np.abs(x[xo == 0])
    return r


def test_alt_body_equals_oracle_draw_for_draw(hh, oracle):
    rng = np.random.default_rng(3)
    N = 120000
    h = rng.integers(1, 14, N).astype(float)
    h[::5] = rng.uniform(1, 13, len(h[::5]))
    h[::97] = rng.uniform(13, 60, len(h[::97]))
    z = rng.normal(0, 2 ** 0.5, N)
    z[::17] = rng.uniform(-14, 14, len(z[::17]))
    z[3] = 0.0
    h[5], h[6] = 0.0, 0.5
    x, st = _run_alt(hh, oracle, h, z, 99, 2, 5)
    xo = oracle.rpg_alt(N, h, z, 99, 2, 5)
    assert st == 2 and x[5] == 0.0 and x[6] == 0.0          # h = 0 -> 0; h < 1 refused and flagged
    r = _rel(x, xo)
    assert (r > 1e-10).sum() <= 1, r.max()


def test_sp_body_equals_oracle_draw_for_draw(hh, oracle):
    rng = np.random.default_rng(4)
    N = 120000
    h = rng.integers(14, 51, N).astype(float)
    h[::5] = rng.uniform(13.01, 170, len(h[::5]))
    h[::101] = rng.uniform(1, 13, len(h[::101]))
    z = rng.normal(0, 2 ** 0.5, N)
    z[::17] = rng.uniform(-14, 14, len(z[::17]))
    z[::1001] = rng.uniform(30, 80, len(z[::1001]))          # proposals below 2^-4: the asymptotic branch of v(x)
    z[3] = 0.0
    h[5] = 0.0
    x, it, st = _run_sp(hh, oracle, h, z, 98, 1, 7)
    xo, ito = oracle.rpg_sp(N, h, z, 98, 1, 7)
    assert st == 0 and x[5] == 0.0 and it[5] == -7           # h = 0: x = 0, iter untouched (LogitWrapper.cpp:116-120)
    ito[5] = -7
    r = _rel(x, xo)
    assert (r > 1e-10).sum() <= 1, r.max()
    assert (it != ito).sum() <= 1


def test_continued_fraction(hh, oracle):
    """Gamma(a, x) exp(x) x^-a: the forward-recurrence form against modified Lentz and against scipy."""
    from scipy import special
    L = oracle.lib()
    for a in (1.0, 1.5, 2.5, 3.9, 14.0, 30.0, 50.0, 100.0, 170.0, 400.0):
        # where the samplers call it: x = fz t >= 0.79 for shapes <= 4 (PolyaGammaAlt.cpp:73), x >= 1.18 n for the
        # saddle point's n (PolyaGammaSP.cpp:222); the fraction is ill-conditioned for x << a and not used there
        for lam in ((0.3, 0.9, 1.0, 1.18, 1.3, 1.6, 3.0, 10.0) if a <= 4 else (1.0, 1.18, 1.3, 1.6, 3.0, 10.0)):
            x = a * lam
            got = hh.hh_cf(a, x)
            assert abs(got / L.bl_upper_gamma_cf(a, x) - 1) < 5e-14
            lq = np.log(special.gammaincc(a, x)) if special.gammaincc(a, x) > 1e-290 else None
            if lq is not None:
                assert abs(got / np.exp(lq + special.gammaln(a) + x - a * np.log(x)) - 1) < 2e-12, (a, x)


def test_alt_setup_matches_literal(hh, oracle):
    L = oracle.lib()
    out = (C.c_double * 11)()
    for h in (1.0, 1.5, 2.0, 2.75, 3.0, 3.99, 4.0):
        for z in (0.0, 0.3, 1.0, 2.5, 6.0, 9.0, 15.0, 40.0):
            hh.hh_alt_par(h, z, out)
            Z, t, p = out[1], out[2], out[5]
            wl, wr = L.bl_alt_w_left(t, h, Z), L.bl_alt_w_right(t, h, Z)
            lit = wr / (wr + wl)
            assert abs(p - lit) <= 2e-11 * lit + 1e-13, (h, z, p, lit)   # the literal 1 - P carries ~1e-16 of absolute noise into a weight that small
            assert abs(out[10] - (h * np.log(4 / np.pi) + math.lgamma(h + 1) - 0.5 * np.log(2 * np.pi))) < 1e-13   # cR


def test_sp_setup_equals_oracle(hh, oracle):
    L = oracle.lib()

    class SpPar(C.Structure):
        _fields_ = [(k, C.c_double) for k in ("n", "Z2", "md", "logmd", "lcZ", "lhal", "lhar", "rl", "il", "rr", "ir",
                                              "mu", "pl", "ipl", "iql", "b", "ic0", "omc", "log_m")]
    L.bl_sp_par_of.argtypes = [C.POINTER(SpPar), C.c_double, C.c_double]
    L.bl_sp_par_of.restype = None
    out = (C.c_double * 15)()
    names = ("n", "Z2h", "md", "mu", "pl", "b", "mdb", "lmdb", "ic0", "omc", "log_m", "cL0", "cL1", "cR0", "cR1")
    for n in (1.0, 2.5, 14.0, 20.0, 35.5, 60.0, 170.0):
        for z in (0.0, 0.001, 0.4, 1.0, 3.0, 8.0, 30.0):
            hh.hh_sp_par(n, z, out)
            p = SpPar()
            L.bl_sp_par_of(C.byref(p), n, z)
            got = dict(zip(names, out))
            # the kernel's folded constants (bl_sp_sm.hpp, SpPar) from the oracle's unfolded ones
            want = {k: getattr(p, k) for k in ("n", "md", "mu", "pl", "b", "ic0", "omc", "log_m")}
            want["Z2h"] = 0.5 * p.Z2
            want["mdb"] = p.md / p.b
            want["lmdb"] = np.log(p.md / p.b)
            want["cL0"] = p.lhal + n * (p.il + 0.5 / p.md - p.lcZ)
            want["cL1"] = n * p.rl
            want["cR0"] = p.lhar + n * (p.ir - p.logmd - p.lcZ)
            want["cR1"] = n * p.rr
            for k in names:
                a, b = got[k], want[k]
                assert abs(a - b) <= 1e-11 * max(1.0, abs(b)), (n, z, k, a, b)


def test_vlk_equals_oracle(hh, oracle):
    out = (C.c_double * 3)()
    for x in np.concatenate([2.0 ** np.linspace(-6, 6, 601), [1.0, 0.9999999, 1.0000001]]):
        hh.hh_sp_vlk(float(x), out)
        ref = oracle.sp_vlk(float(x))
        for a, b in zip(out, ref):
            assert abs(a - b) <= 1e-13 * max(1.0, abs(b))


def test_alt_block_cap_is_per_draw(hh, oracle):
    """PolyaGammaAlt::draw sums floor((h-1)/4) abridged draws at shape 4 plus the remainder (PolyaGammaAlt.cpp:205-225).
    The block cap counts from the start of each abridged draw: built with a cap of 64 blocks, a group of 2500 draws
    (h = 10001.5: ~4000 blocks) is not cut short, raises no flag and equals the default build on the same stream."""
    lib = os.path.join(HERE, "host_harness", "libhyb_sm_host_cap64.so")
    hdrs = glob.glob(os.path.join(HERE, "..", "bayeslogit_amd", "csrc", "*.hpp"))
    if not os.path.exists(lib) or any(os.path.getmtime(f) > os.path.getmtime(lib) for f in hdrs + [SRC]):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wno-unknown-pragmas", "-DBL_ALT_BLK_CAP=64u",
                               "-o", lib, SRC, "-lm"])
    Hc = C.CDLL(lib)
    h = np.array([10001.5, 4003.0, 13.0])
    z = np.array([0.7, 0.0, 2.5])
    xc, stc = _run_alt(Hc, oracle, h, z, 31)
    xd, std = _run_alt(hh, oracle, h, z, 31)
    assert stc == 0 and std == 0 and np.array_equal(xc, xd)
    m1 = h * np.tanh(np.maximum(z, 1e-12) / 2) / (2 * np.maximum(z, 1e-12))
    assert np.all(np.abs(xc[:2] / m1[:2] - 1) < 0.05)           # E PG(h, z); sd/mean ~ 1 % at h = 4000
