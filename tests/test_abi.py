"""The C-ABI shared library loads and exports every symbol include/*.h declares (bayeslogit_hip.h: the drop-in
boundary and its device-resident form; bayeslogit_hip_diag.h: diagnostics).  No compute call is made here (this
file runs without a GPU)."""
import os
import re
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
HEADERS = [os.path.join(ROOT, "include", f) for f in ("bayeslogit_hip.h", "bayeslogit_hip_diag.h")]


def declared_symbols(headers=HEADERS):
    src = "\n".join(open(h).read() for h in headers)
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    names = re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\(", src)
    skip = {"defined", "sizeof"}
    return sorted({n for n in names if n not in skip})


def test_header_declares_reference_table():
    """The drop-in table is exactly Code/C/LogitWrapper.h:23-64 (10 entry points)."""
    ref = {"rpg_gamma", "rpg_devroye", "rpg_alt", "rpg_sp", "rpg_hybrid", "gibbs", "EM", "combine", "mult_gibbs",
           "mult_combine"}
    assert ref <= set(declared_symbols(HEADERS[:1]))
    # diagnostics live in their own header: the boundary header declares none
    assert not [n for n in declared_symbols(HEADERS[:1]) if n.startswith("bl_diag_")]


def test_library_exports_every_declared_symbol(hiplib):
    from bayeslogit_amd import _lib
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH], text=True)
    exported = {line.split()[-1] for line in out.splitlines() if " T " in line}
    missing = [s for s in declared_symbols() if s not in exported]
    assert not missing, missing
    # and the ctypes table binds each of them (signature table in _lib.py covers the header)
    assert set(_lib.SIGNATURES) == set(declared_symbols())


def test_reference_signatures_are_void_pointer_style(hiplib):
    """`.C` hands pointers only; the ten reference symbols return void (LogitWrapper.h:23-64)."""
    from bayeslogit_amd import _lib
    nargs = {"rpg_gamma": 5, "rpg_devroye": 4, "rpg_alt": 4, "rpg_sp": 5, "rpg_hybrid": 4, "gibbs": 11, "EM": 8,
             "combine": 5, "mult_gibbs": 12, "mult_combine": 6}
    for name, k in nargs.items():
        res, args = _lib.SIGNATURES[name]
        assert res is None and len(args) == k


def test_no_gpu_fails_loudly(hiplib):
    """Without a device the product path must refuse to compute (no CPU fallback)."""
    import bayeslogit_amd as bl
    if hiplib.bl_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(bl.BayesLogitError):
        bl.rpg(10, 1.0, 0.0)
    with pytest.raises(bl.BayesLogitError):
        bl.logit([0.0, 1.0], [[1.0], [2.0]], samp=2, burn=0)
    assert hiplib.bl_rpg_devroye_dev(None, None, 1, None, 0, 0, 0, 0, None) == 1   # BL_ERR_NO_DEVICE
    assert b"no HIP device" in hiplib.bl_last_error()


def test_product_never_touches_oracle():
    """Nothing under bayeslogit_amd/ may import, link or include the oracle."""
    pkg = os.path.join(ROOT, "bayeslogit_amd")
    for base, _, files in os.walk(pkg):
        if "_obj" in base:
            continue
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                txt = open(os.path.join(base, f)).read()
                assert "oracle" not in txt.lower().replace("# oracle-free", ""), os.path.join(base, f)
