"""The committed golden vectors (tests/golden/pg_golden_v2.json, made by make_golden.py from the oracle)
are reproduced bit-for-bit by the oracle.  The HIP path is checked against the same file under -m gpu."""
import ctypes as C
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def load():
    return json.load(open(os.path.join(HERE, "golden", "pg_golden_v2.json")))


def test_oracle_reproduces_golden(oracle):
    g = load()
    L = oracle.lib()
    u = g["uniforms"]
    r = oracle.rng(u["seed"], u["idx"], u["domain"], u["epoch"])
    assert [L.bl_unif(C.byref(r)) for _ in u["u"]] == u["u"]
    for Z, v in g["helpers"]["mass_texpon"]:
        assert L.bl_pg_mass_texpon(Z) == v
    for n, x, v in g["helpers"]["a"]:
        assert L.bl_pg_a(int(n), x) == v
    for y, v in g["helpers"]["v_eval"]:
        assert L.bl_v_eval(y) == v
    for x, n, z, v in g["helpers"]["sp_approx"]:
        assert L.bl_sp_approx(x, n, z) == v
    for d in g["hybrid_draws"]:
        for key, lit in (("x", False), ("x_literal", True)):
            x = oracle.rpg_hybrid(len(d[key]), d["b"], d["z"], d["seed"], d["epoch"], d["idx0"], literal=lit)
            assert x.tolist() == d[key], d["b"]
    d = g["devroye_n"]
    assert oracle.rpg_devroye(16, d["n"], np.array(d["z"]), d["seed"]).tolist() == d["x"]
    d = g["sp_iter"]
    xs, it = oracle.rpg_sp(6, d["h"], d["z"], d["seed"])
    assert xs.tolist() == d["x"] and it.tolist() == d["iter"]
    xs, it = oracle.rpg_sp(6, d["h"], d["z"], d["seed"], literal=True)
    assert xs.tolist() == d["x_literal"] and it.tolist() == d["iter_literal"]
    d = g["alt"]
    assert oracle.rpg_alt(10, d["h"], d["z"], d["seed"]).tolist() == d["x"]
    assert oracle.rpg_alt(10, d["h"], d["z"], d["seed"], literal=True).tolist() == d["x_literal"]


def test_oracle_reproduces_golden_gibbs(oracle):
    from golden.make_golden import gibbs_problem
    g = load()["gibbs"]
    X, y, n = gibbs_problem()
    P = X.shape[1]
    for con in (0, 1):
        assert oracle.chain_key(g["seed"], 0) == g["chain_key"]
        w, beta = oracle.gibbs(y, X, n, np.zeros(P), np.eye(P) * 0.25, g["samp"], g["burn"], g["chain_key"], con)
        assert beta.tolist() == g[f"beta_constrain{con}"]
        assert w[-1, :8].tolist() == g[f"w_last_head_constrain{con}"]
    be, it = oracle.em(y, X, n)
    assert be.tolist() == g["em_beta"] and it == g["em_iter"]
