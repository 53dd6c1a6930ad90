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
    path = os.path.join(ROOT, "oracle", "inverty_grid.c")
    assert np.array_equal(_c_table(path, "bl_ygrid"), y)
    assert np.array_equal(_c_table(path, "bl_vgrid"), v)


def test_fitted_inversion_table_reproduces_reference_grid(oracle):
    """The product's and the oracle's fitted table of v(x) (bl_vtab.hpp / oracle/vtab.c, scripts/gen_vtab.py)
    passes through the reference's own (ygrid, vgrid) points (InvertY.hpp:20-57) to their 7 printed digits, and
    the two copies hold the same numbers."""
    g = np.loadtxt(os.path.join(HERE, "golden", "inverty_grid.txt"))
    for k, (y, v) in enumerate(g):
        vv, _, _ = oracle.sp_vlk(2.0 ** (-4 + 0.1 * k))      # the grid's exact abscissa (y is printed to 7 digits)
        assert abs(vv - v) <= 6e-7 * max(1.0, abs(v)), (y, v, vv)
    src = open(os.path.join(ROOT, "bayeslogit_amd", "csrc", "bl_vtab.hpp")).read()
    osrc = open(os.path.join(ROOT, "oracle", "vtab.c")).read()
    nums = lambda t: re.findall(r"-?\d\.\d+(?:e[-+]\d+)?|-?\d+\.\d+(?:e[-+]\d+)?", t[t.index("= {"):])
    assert nums(src) == nums(osrc) and len(nums(src)) == 3 * 16 * 11


def test_oracle_tables_loaded(oracle):
    import ctypes as C
    L = oracle.lib()
    t = (C.c_double * 301).in_dll(L, "bl_trunc_schedule")
    assert t[0] == 0.64 and t[300] == 4.13
