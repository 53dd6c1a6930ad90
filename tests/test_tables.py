"""Data tables vs the reference's own data files / identities (SURVEY.md 8c 'embedded constant tables')."""
import os
import re

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _c_table(path, name):
    src = open(path).read()
    m = re.search(name + r"\[\d+\]\s*=\s*\{(.*?)\};", src, re.S)
    return np.array([float(x) for x in m.group(1).replace("\n", " ").split(",") if x.strip()])


def test_trunc_schedule_is_t1to4():
    t = np.loadtxt(os.path.join(HERE, "golden", "t1to4.txt"))
    h = np.loadtxt(os.path.join(HERE, "golden", "h1to4.txt"))
    assert t.shape == (301,) and h.shape == (301,)
    assert np.allclose(h, 1.0 + 0.01 * np.arange(301))          # grid of PolyaGammaAlt.cpp:124-125
    assert np.array_equal(_c_table(os.path.join(ROOT, "oracle", "trunc_schedule.c"), "bl_trunc_schedule"), t)
    assert np.array_equal(_c_table(os.path.join(ROOT, "bayeslogit_amd", "csrc", "bl_tables.hpp"), "kTruncSchedule"), t)
    assert t[0] == 0.64 and t[300] == 4.13


def test_inverty_grid_identity():
    g = np.loadtxt(os.path.join(HERE, "golden", "inverty_grid.txt"))
    y, v = g[:, 0], g[:, 1]
    assert y.shape == (81,)
    assert np.allclose(y, 2.0 ** (-4 + 0.1 * np.arange(81)), rtol=1e-6)      # InvertY.hpp:20, 7 printed digits
    r = np.sqrt(np.abs(v))
    with np.errstate(invalid="ignore", divide="ignore"):
        yy = np.where(v > 0, np.tan(r) / r, np.where(v < 0, np.tanh(r) / r, 1.0))
    assert np.allclose(yy, y, rtol=5e-6)                                        # y = tan(sqrt v)/sqrt v
    for path, ny, nv in ((os.path.join(ROOT, "oracle", "inverty_grid.c"), "bl_ygrid", "bl_vgrid"),
                         (os.path.join(ROOT, "bayeslogit_amd", "csrc", "bl_tables.hpp"), "kYGrid", "kVGrid")):
        assert np.array_equal(_c_table(path, ny), y)
        assert np.array_equal(_c_table(path, nv), v)


def test_oracle_tables_loaded(oracle):
    import ctypes as C
    L = oracle.lib()
    t = (C.c_double * 301).in_dll(L, "bl_trunc_schedule")
    assert t[0] == 0.64 and t[300] == 4.13
