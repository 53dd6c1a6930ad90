#!/usr/bin/env python3
"""Golden vectors of the reference's own InvertY.cpp.

    make -C oracle ref && python tests/golden/make_inverty_ref.py

oracle/_ref/libinverty_ref.so is /root/reference/Code/C/InvertY.cpp compiled UNCHANGED (g++ -std=c++11 -O2,
no stand-in header, not the reference's build system; recipe: oracle/Makefile).  Its functions have C++
linkage, so they are bound here by their mangled names.  This script runs only where /root/reference exists
(the build container); what travels is its output, tests/golden/inverty_ref.json: inputs and outputs as C99
hex floats (bit-exact), nothing of the reference's text.

Grids: v_eval on y = 2^-5 .. 2^5 (both asymptotic branches, the 81 table abscissae 2^(-4+0.1k), y = 1, the
ends of the table, a log-uniform random set); y_eval / ydy_eval / fdf_eval on v in +-[1e-12, 60] including the
|v| < 1e-8 series branch (whose (1/3), (2/15), (17/315) are integer divisions = 0: hazard H5) and v = 0."""
import ctypes as C
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
LIB = os.path.join(ROOT, "oracle", "_ref", "libinverty_ref.so")


def load():
    L = C.CDLL(LIB)
    f = {}
    f["y_eval"] = getattr(L, "_Z6y_evald")                       # double y_eval(double)
    f["y_eval"].restype, f["y_eval"].argtypes = C.c_double, [C.c_double]
    f["ydy_eval"] = getattr(L, "_Z8ydy_evaldPdS_")               # void ydy_eval(double, double*, double*)
    f["ydy_eval"].restype, f["ydy_eval"].argtypes = None, [C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    f["fdf_eval"] = getattr(L, "_Z8fdf_evaldPvPdS0_")            # void fdf_eval(double, void*, double*, double*)
    f["fdf_eval"].restype = None
    f["fdf_eval"].argtypes = [C.c_double, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    f["v_eval"] = getattr(L, "_Z6v_evalddi")                     # double v_eval(double y, double tol, int max_iter)
    f["v_eval"].restype, f["v_eval"].argtypes = C.c_double, [C.c_double, C.c_double, C.c_int]
    return f


def grids():
    rng = np.random.default_rng(20240013)
    ys = [2.0 ** e for e in np.arange(-5.0, 5.0001, 0.125)]
    ys += [2.0 ** (-4 + 0.1 * k) for k in range(81)]                       # the table's abscissae (InvertY.hpp:20)
    ys += [1.0, np.nextafter(1.0, 0.0), np.nextafter(1.0, 2.0), 0.0625, 16.0, 0.0624, 16.01, 0.0626, 15.99]
    ys += list(2.0 ** rng.uniform(-5.0, 5.0, 600))
    ys += list(rng.uniform(0.9, 1.1, 100))                                  # around the mode of the sampler's x
    mags = list(10.0 ** np.arange(-12.0, 1.51, 0.25)) + [1e-8, 0.99e-8, 1.01e-8, 2.0, 2.4, 2.46, 30.0, 60.0]
    vs = [0.0] + mags + [-m for m in mags] + list(rng.uniform(-40.0, 2.45, 300)) + list(rng.uniform(-2e-8, 2e-8, 40))
    return [float(y) for y in ys], [float(v) for v in vs]


def main():
    f = load()
    ys, vs = grids()
    hx = float.hex
    out = {"source": "Code/C/InvertY.cpp:10-99 compiled unchanged (oracle/Makefile target `ref`); C99 hex floats",
           "v_eval": [], "y_eval": [], "ydy_eval": [], "fdf_eval": []}
    for y in ys:
        out["v_eval"].append([hx(y), hx(f["v_eval"](y, 1e-9, 1000))])     # the header's default arguments
    a, b = C.c_double(), C.c_double()
    for v in vs:
        out["y_eval"].append([hx(v), hx(f["y_eval"](v))])
        f["ydy_eval"](v, C.byref(a), C.byref(b))
        out["ydy_eval"].append([hx(v), hx(a.value), hx(b.value)])
    for v in vs[::7]:
        for y in (0.3, 1.0, 1.7, 9.0):
            yy = C.c_double(y)
            f["fdf_eval"](v, C.cast(C.byref(yy), C.c_void_p), C.byref(a), C.byref(b))
            out["fdf_eval"].append([hx(v), hx(y), hx(a.value), hx(b.value)])
    with open(os.path.join(HERE, "inverty_ref.json"), "w") as fh:
        json.dump(out, fh, indent=0)
    print({k: len(v) for k, v in out.items() if isinstance(v, list)})


if __name__ == "__main__":
    main()
