"""Closed forms used as known answers: exact PG(1,z) CDF from the
alternating series of Code/R/PG.R:320-348 / PolyaGamma.cpp:41-55, and PG(b,z) moments."""
import numpy as np

_T = 0.64


def jstar_density(x):
    """Density of J*(1,0) at x > 0: sum_n (-1)^n a_n(x), both series forms (PolyaGamma.cpp:41-55)."""
    x = np.asarray(x, dtype=float)
    out = np.zeros_like(x)
    n = np.arange(0, 60)[:, None]
    K = (n + 0.5) * np.pi
    sign = (-1.0) ** n
    with np.errstate(divide="ignore", over="ignore", invalid="ignore", under="ignore"):
        left = (2.0 / (np.pi * x ** 3)) ** 0.5 * np.sum(sign * (2 * n + 1) * np.exp(-2 * (n + 0.5) ** 2 / x), axis=0)
        right = np.sum(sign * K * np.exp(-0.5 * K * K * x), axis=0)
    out = np.where(x <= _T, left, right)
    return np.where(x > 0, out, 0.0)


_GRID = None


def pg1_cdf(w, z):
    """CDF of PG(1, z) at w: tilt exp(-z^2 w/2) cosh(z/2) of the PG(1,0) density, integrated numerically."""
    global _GRID
    if _GRID is None:
        g = np.concatenate([np.linspace(1e-6, 0.5, 200001), np.linspace(0.5, 12.0, 100001)[1:]])
        _GRID = (g, 4.0 * jstar_density(4.0 * g))
    g, f0 = _GRID
    f = f0 * np.exp(-0.5 * z * z * g) * np.cosh(0.5 * z)
    c = np.concatenate([[0.0], np.cumsum(0.5 * (f[1:] + f[:-1]) * np.diff(g))])
    return np.interp(w, g, c / c[-1] if abs(c[-1] - 1) < 1e-6 else c)


def pg_mean(b, z):
    z = np.abs(np.asarray(z, dtype=float))
    with np.errstate(invalid="ignore", divide="ignore"):
        m = np.where(z > 1e-8, b * np.tanh(z / 2) / (2 * z), b / 4.0)
    return m


def pg_var(b, z):
    z = np.abs(np.asarray(z, dtype=float))
    with np.errstate(invalid="ignore", divide="ignore"):
        v = np.where(z > 1e-4, b * (np.sinh(z) - z) / (4 * z ** 3 * np.cosh(z / 2) ** 2), b / 24.0)
    return v
