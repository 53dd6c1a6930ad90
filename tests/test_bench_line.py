"""The bench.py output contract, checked on the committed line of the last GPU session
(profiles/rNN_bench_line.json is bench.py's own stdout, copied by scripts/gpu_round.sh) and on bench.py's
argument defaults.  No GPU."""
import glob
import importlib.util
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _line():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_line.json")))
    assert files, "no committed bench line"
    return json.loads(open(files[-1]).read())


def test_contract_keys_and_types():
    d = _line()
    for k, typ in [("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                   ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str),
                   ("config", dict)]:
        assert k in d and isinstance(d[k], typ), k
    # vs_baseline: value / the CPU baseline of the same run (the oracle on the box's host cores; the reference publishes no
    # number at this size: BASELINE.md); the ratio to its one published PG(1,0) figure is a separate key
    assert abs(d["vs_baseline"] - d["value"] / d["cpu_baseline"]["value"]) < 1e-9 * d["vs_baseline"]
    assert abs(d["vs_published_pg10_serial"] - d["value"] / 2.4) < 1e-9 * d["value"] and "oracle" in d["vs_baseline_note"]
    assert d["metric"].startswith("PG draws/sec") and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["dtype"] == "f64" and d["data"] == "synthetic" and "workload" in d["config"]
    assert "model" not in d["config"]
    # value is draws per second over the timed steps
    draws = d["config"]["draws_per_gpu_per_step"] * d["n_gpus"]
    assert abs(d["value"] * 1e6 - draws / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"] * 1e6


def test_roofline_and_cpu_baseline_objects():
    d = _line()
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert r["traffic"] is None or r["traffic"] >= r["algorithmic_bytes_per_launch"]   # measured >= algorithmic
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0
    g = d["gibbs"]
    assert g["unit"] == "sweeps/s" and g["value"] == g["constrained"]["sweeps_per_s"]
    gr = g["roofline"]
    assert abs(gr["frac"] - gr["achieved"] / gr["peak"]) < 1e-12
    ch = g["chain"]
    assert ch["burn"] == 100 and ch["samp"] == 1000 and ch["max_abs_z"] < 5.0       # posterior means near the truth
    # the exchange on its own, attempts per draw, the fp64-VALU rate, per-branch rates of C3, the C5 shard
    ex = g["exchange"]
    assert ex["bytes"] == 8 * 64 * 64 and ex["world_size"] == d["n_gpus"] and "allreduce_us" in ex
    wk = r["work"]
    assert 1.0 < wk["attempts_per_draw"] < 2.0 and 0.0 < wk["fp64_valu"]["frac"] < 1.0
    assert abs(wk["fp64_valu"]["frac"] - wk["fp64_valu"]["achieved_tflops"] / wk["fp64_valu"]["peak_tflops"]) < 1e-12
    br = d["mixed"]["branches"]
    shares = sum(br[k]["share"] for k in ("saddle_point", "alternating_series", "devroye", "normal_approximation", "sum_of_gammas"))
    assert abs(shares - 1.0) < 1e-6 and br["saddle_point"]["attempts_per_draw"] >= 1.0
    mt = d["mixed"]["roofline"]
    assert mt["traffic"] is None or mt["traffic"] >= mt["algorithmic_bytes_per_launch"]
    c5 = d["gibbs_c5"]
    assert c5["unit"] == "sweeps/s" and c5["roofline"]["bound"] == "mfma" and "12500000" in c5["workload"]
    assert c5["roofline"]["traffic"] is None or c5["roofline"]["traffic"] < 1.1 * c5["roofline"]["algorithmic_bytes_per_launch"]
    assert g["mlogit"]["sweeps_per_s_incl_host_transfers"] > 0 and g["combine"]["rows_out"] == g["combine"]["rows_out_expected"]


def _bench_module(args):
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    argv = sys.argv
    try:
        sys.argv = ["bench.py"] + list(args)
        spec.loader.exec_module(m)
        a = m.parse()
    finally:
        sys.argv = argv
    return m, a


def test_launcher_starts_the_ranks_as_children_before_any_gpu_call(monkeypatch):
    """`python bench.py --gpus N` (N > 1, no WORLD_SIZE): main() must hand over to torch.distributed.run -- the
    driver's own launch line, bench.py's arguments passed through -- as a CHILD process, before this process initialises
    the GPU, and exit with the child's code."""
    import torch
    args = ["--gpus", "2", "--steps", "3", "--warmup", "1", "--backend", "gloo", "--no-cpu", "--draws", "1000"]
    m, a = _bench_module(args)
    cmd = m.launcher_cmd(args, 2, 4711)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=2" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "4711"
    k = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[k + 1:] == args                                   # everything after the script is bench.py's own line
    seen = {}
    before = torch.cuda.is_initialized()                         # True only when an earlier GPU test of this process did it

    def fake_call(c, env=None):
        seen["cmd"], seen["env"] = c, env
        seen["gpu_initialised"] = torch.cuda.is_initialized()
        return 7

    monkeypatch.setattr(m.subprocess, "call", fake_call)
    monkeypatch.setattr(m.sys, "argv", ["bench.py"] + args)
    for k_ in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k_, raising=False)
    try:
        m.main()
        raise AssertionError("main() returned instead of exiting with the children's code")
    except SystemExit as e:
        assert e.code == 7                                       # a failing child fails the parent
    assert seen["gpu_initialised"] is before                     # main() itself touched no GPU before handing over
    assert seen["cmd"][seen["cmd"].index(os.path.join(ROOT, "bench.py")) + 1:] == args
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and "WORLD_SIZE" not in seen["env"]
    # RCCL needs one GPU per rank: without them the launcher refuses (rc 2) instead of starting ranks that cannot run
    monkeypatch.setattr(m.sys, "argv", ["bench.py", "--gpus", "64"])
    try:
        m.main()
        raise AssertionError
    except SystemExit as e:
        assert e.code == 2


def test_bench_defaults_are_one_gpu_and_short():
    m, a = _bench_module([])
    assert a.gpus == 1 and a.steps <= 50 and a.warmup <= 10
    assert a.draws == 100_000_000 and a.gibbs_n == 10_000_000 and a.gibbs_p == 64      # BASELINE configs C2, C4
    assert a.backend == "nccl"
    assert not a.no_c5 and a.c5_rows == 12_500_000 and not a.no_mixed     # C3 and the C5 shard are in the default line
