#!/usr/bin/env python3
"""bench.py -- headline benchmark of the Polya-Gamma hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W
  N > 1: one rank per GPU over RCCL.  Either the caller starts the ranks
  (python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N: WORLD_SIZE is then set)
  or bench.py does: a plain `python bench.py --gpus N` starts that same command as a CHILD process before this
  process has made any GPU call, passes its output through and exits with its code (launch_ranks below).

Primary metric (BASELINE.json): PG draws/sec (millions) on config C2 --
1e8 PG(1, z) draws per GPU, z ~ Unif(0,4) generated on the device.  A "step" is one
launch of the draw kernel over the whole resident (z) vector.  PG draws shard by index
range with no collective ("weak": every rank draws its own 1e8).
The same JSON line carries the Gibbs metric (sweeps/sec at N=1e7, P=64, rows sharded over
the ranks, one P*P all-reduce per sweep), the roofline object of the dominant kernel of
each, and the CPU baseline (the oracle, timed on this box's host cores, rank 0, N=1 only).
"""
import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: 8 TB/s spec
FP64_VALU_PEAK_TFLOPS = 78.6    # 256 CU x 4 SIMD x 16 FMA lanes/clk x 2 x 2.4 GHz
FP64_MFMA_PEAK_TFLOPS = 78.6    # AMD's MI355X figure for fp64 matrix (= the vector rate); MI355X_MICROARCH.md has no fp64 row
BYTES_PER_DRAW = 16             # 8 B z in + 8 B x out (scalar shape)
BYTES_PER_DRAW_VEC = 24         # 8 B h + 8 B z in + 8 B x out through the vector-shape rpg_hybrid signature


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--draws", type=int, default=100_000_000, help="PG(1,z) draws per GPU per step (C2)")
    ap.add_argument("--gibbs-n", type=int, default=10_000_000, help="total rows of the Gibbs problem (C4)")
    ap.add_argument("--gibbs-p", type=int, default=64)
    ap.add_argument("--gibbs-sweeps", type=int, default=30)
    ap.add_argument("--no-gibbs", action="store_true")
    ap.add_argument("--gibbs-chain", type=int, default=1000,
                    help="C4 only: also run a chain of 100 burn-in + this many sampling sweeps (SURVEY 8d: posterior "
                         "mean and sd of beta, sweeps/s over the sampling phase); 0 = skip")
    ap.add_argument("--c5", action="store_true", help="(default now; kept so that older command lines still parse)")
    ap.add_argument("--no-c5", action="store_true", help="skip config C5 (N = 1e8, P = 256 over 8 GPUs: 12.5e6 x 256 rows, "
                                                         "25.6 GB, per rank -- the full problem at --gpus 8)")
    ap.add_argument("--c5-rows", type=int, default=12_500_000, help="rows per rank of the C5 leg")
    ap.add_argument("--c5-sweeps", type=int, default=5)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-mixed", action="store_true", help="skip config C3 (mixed shapes b in 1..50 through rpg_hybrid)")
    ap.add_argument("--mixed", action="store_true", help="(default now; kept so that older command lines still parse)")
    ap.add_argument("--mixed-steps", type=int, default=5)
    ap.add_argument("--post-n", type=int, default=10000,
                    help="rows of the problem on which a GPU chain and a CPU (oracle) chain are run on the same (X, y) and "
                         "their posterior mean/sd compared (0 = skip)")
    ap.add_argument("--post-samp", type=int, default=2000)
    ap.add_argument("--mlogit-n", type=int, default=1_000_000, help="rows of the mlogit timing (P=32, J=5; 0 = skip) and of the combine timing (P=64)")
    ap.add_argument("--cpu-gibbs-n", type=int, default=1_000_000, help="rows of the timed CPU Gibbs sample (10 sweeps)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for rehearsing the N>1 path on a 1-GPU box)")
    return ap.parse_args()


def launcher_cmd(argv, nproc, port):
    """The command a plain `python bench.py --gpus N` (N > 1, no WORLD_SIZE) starts: the driver's own launch line
    (one rank per GPU, rendezvous on 127.0.0.1) with bench.py's arguments passed through unchanged."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def launch_ranks(a, argv):
    """N > 1 without a launcher: start the ranks as fresh children and return their exit code.  This process never
    touches the GPU (counting devices does not initialise it on this image; a process that HAS initialised the GPU must not
    start other programs on this pool), it only relays: rank 0 of the children prints the JSON line."""
    ndev = torch.cuda.device_count()
    if a.backend == "nccl" and ndev < a.gpus:
        print(f"bench.py: --gpus {a.gpus} over RCCL needs {a.gpus} GPUs, {ndev} visible "
              "(--backend gloo rehearses the N > 1 path with the ranks sharing the GPUs there are)", file=sys.stderr)
        return 2
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return subprocess.call(launcher_cmd(argv, a.gpus, port), env=env)


def barrier_sync(world):
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()


def max_over_ranks(x, world, dev):
    if world == 1:
        return x
    t = torch.tensor([x], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return t.item()


def timed_steps(fn, steps, warmup, world, dev):
    """W untimed steps, then exactly K steps between barrier+sync; per-step HIP-event durations."""
    for _ in range(warmup):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    barrier_sync(world)
    t0 = time.perf_counter()
    for a, b in ev:
        a.record()
        fn()
        b.record()
    barrier_sync(world)
    wall = time.perf_counter() - t0
    wall = max_over_ranks(wall, world, dev)
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    return wall, kern_ms


def pmc_entry(kernel):
    """The newest committed PMC summary's entry for `kernel` ({} if none)."""
    import glob
    files = sorted(glob.glob(os.path.join(HERE, "profiles", "r*_pmc_summary.json")))
    if not files:
        return {}
    try:
        with open(files[-1]) as f:
            return json.load(f).get(kernel, {})
    except (OSError, ValueError):
        return {}


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the newest committed rocprofv3 --pmc summary
    (profiles/rNN_pmc_summary.json: 2 x FETCH_SIZE + WRITE_SIZE, separate passes, gfx950 correction),
    or None.  bench.py cannot collect counters on itself; the summary is of this same command."""
    import glob
    files = sorted(glob.glob(os.path.join(HERE, "profiles", "r*_pmc_summary.json")))
    if not files:
        return None, None
    try:
        with open(files[-1]) as f:
            d = json.load(f)
        e = d.get(kernel, {})
        return e.get("hbm_bytes_per_launch"), os.path.basename(files[-1])
    except (OSError, ValueError):
        return None, None


def flops_table():
    """profiles/flops_table.json (scripts/count_flops.py: VALU instructions and fp64 flops per call, counted in the gfx950 ISA)."""
    try:
        with open(os.path.join(HERE, "profiles", "flops_table.json")) as f:
            return json.load(f)["per_call"]
    except (OSError, ValueError, KeyError):
        return None


def devroye_work(D, z, seed, epoch, idx0, draws_per_s):
    """SURVEY 8d for the PG(1,z) kernel: mean attempts (Philox blocks) per draw on THIS input -- an exact replay of the
    streams the timed kernel drew from -- and, through the flops-per-call table, the fp64 VALU op rate the measured
    draws/s amount to, against the 78.6 TFLOP/s vector peak."""
    n = min(z.numel(), 20_000_000)
    c = D.count_blocks(None, z[:n], seed=seed, epoch=epoch, idx0=idx0)
    dv = c["devroye"]
    wide = c.get("devroye_wide_z", {"observations": 0, "draws": 0, "blocks": 0})
    nar = {k: dv[k] - wide[k] for k in dv}
    out = {"replayed_draws": dv["draws"], "attempts_per_draw": dv["blocks"] / dv["draws"],
           "attempts_per_draw_small_z": nar["blocks"] / max(nar["draws"], 1),
           "attempts_per_draw_wide_z": (wide["blocks"] / wide["draws"]) if wide["draws"] else None,
           "share_wide_z": wide["draws"] / dv["draws"],
           "note": "attempt = one Philox4x32-10 block = one pass of the attempt body (bl_pg1_sm.hpp); small_z / wide_z: "
                   "|z|/2 < 1/t resp. >= 1/t, the two left-piece samplers of PolyaGamma.cpp:87-113; the reference's own "
                   "count is 1.0007 proposals x 1.0016 series terms per draw (Notes/notes.tex:1000-1027) with the inner "
                   "rejection loops of rtigauss counted inside a proposal -- here each inner retry is an attempt"}
    ft = flops_table()
    if ft:
        f1 = ft["attempt_class1"]["fp64_flops"] + ft["philox"]["fp64_flops"]
        f2 = ft["attempt_class2"]["fp64_flops"] + ft["philox"]["fp64_flops"]
        s1 = ft["mass_small"]["fp64_flops"] + 2 * ft["div"]["fp64_flops"] + 3      # mass, the two reciprocals, fz
        s2 = ft["mass_general"]["fp64_flops"] + 2 * ft["div"]["fp64_flops"] + 3
        v1 = ft["attempt_class1"]["valu_instructions"] + ft["philox"]["valu_instructions"]
        v2 = ft["attempt_class2"]["valu_instructions"] + ft["philox"]["valu_instructions"]
        nd = dv["draws"]
        flops = (nar["blocks"] * f1 + wide["blocks"] * f2 + nar["draws"] * s1 + wide["draws"] * s2) / nd
        valu = (nar["blocks"] * v1 + wide["blocks"] * v2 + nar["draws"] * (ft["mass_small"]["valu_instructions"] + 16)
                + wide["draws"] * (ft["mass_general"]["valu_instructions"] + 16)) / nd
        tf = flops * draws_per_s / 1e12
        out["fp64_valu"] = {"flops_per_draw": flops, "useful_valu_lane_instructions_per_draw": valu,
                            "achieved_tflops": tf, "peak_tflops": FP64_VALU_PEAK_TFLOPS, "frac": tf / FP64_VALU_PEAK_TFLOPS,
                            "flops_per_call": {k: ft[k]["fp64_flops"] for k in ("log", "exp", "div", "sqrt", "philox", "mass_small",
                                                                               "mass_general", "attempt_class1", "attempt_class2")},
                            "valu_instructions_per_call": {k: ft[k]["valu_instructions"] for k in ("log", "exp", "div", "philox",
                                                                                                   "attempt_class1", "attempt_class2")},
                            "note": "flops per call counted in the gfx950 ISA (scripts/count_flops.py -> profiles/flops_table.json: "
                                    "v_fma_f64 = 2, v_mul/v_add_f64 = 1, conversions, seeds, compares, selects and the Philox integer "
                                    "multiplies 0) x the replayed attempt counts; the draw kernels are bound by VALU ISSUE SLOTS "
                                    "(valu.busy_frac_of_cu_cycles), of which fp64 FMAs are about half"}
    return out


def time_allreduce(numel, world, dev, reps=50):
    """The sweep's one exchange on its own: all-reduce(sum) of `numel` doubles (P*P), microseconds per call, max over
    ranks (barrier + synchronize on both sides; 5 untimed calls first)."""
    if world == 1:
        return 0.0
    t = torch.zeros(numel, dtype=torch.float64, device=dev)
    for _ in range(5):
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    barrier_sync(world)
    t0 = time.perf_counter()
    for _ in range(reps):
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return max_over_ranks(dt, world, dev) / reps * 1e6


def cpu_baseline(draws_sample, ncores):
    """The oracle's literal restatement of the reference loops (PolyaGamma.cpp:151-202, call for
    call) on this box's host cores."""
    sys.path.insert(0, os.path.join(HERE, "tests"))
    import oracle_lib as O
    rng = np.random.default_rng(20240001)
    n1 = min(draws_sample, 4_000_000)
    z = rng.uniform(0.0, 4.0, draws_sample)
    t0 = time.perf_counter()
    O.rpg_devroye(n1, 1, z[:n1], 20240002, literal=True)
    one = n1 / (time.perf_counter() - t0)
    t0 = time.perf_counter()
    O.rpg_devroye(draws_sample, 1, z, 20240002, threads=ncores, literal=True)
    allc = draws_sample / (time.perf_counter() - t0)
    return one, allc


def synth_logit(D, dev, N, P, lo=0):
    """The C4 generator: x_ij ~ N(0,1)/sqrt(P), last column 1; beta_true = |N(0,1)|, intercept -0.5;
    y ~ Bernoulli(sigma(x.beta)); rows lo .. lo+N of the global problem."""
    X = torch.empty((N, P), dtype=torch.float64, device=dev)
    D.fill_norm(X, 0.0, 1.0 / P ** 0.5, 20240003, idx0=lo * P)
    X[:, -1] = 1.0
    bt = torch.empty(P, dtype=torch.float64, device=dev)
    D.fill_norm(bt, 0.0, 1.0, 20240003, epoch=1)
    bt.abs_()
    bt[-1] = -0.5
    y = torch.empty(N, dtype=torch.float64, device=dev)
    D.fill_logit_y(y, X, bt, 20240003, epoch=2, idx0=lo)
    return X, y, bt


def batch_mcse(chain, nb=20):
    """Monte-Carlo standard error of the column means of a (samp, P) chain by batch means."""
    m = (chain.shape[0] // nb) * nb
    bm = chain[:m].reshape(nb, m // nb, -1).mean(1)
    return bm.std(0, ddof=1) / np.sqrt(nb)


def cpu_gibbs(D, dev, n_time, P, post_n, post_samp):
    """The Gibbs half of the CPU baseline: the oracle's restatement of Logit::gibbs (Logit.hpp:402-481: serial, as the
    reference is) (a) timed for 10 sweeps on an n_time x P sample of the C4 problem, (b) run as a whole chain on a
    post_n x P problem next to the GPU chain on the SAME (X, y, seed): posterior mean and sd of beta compared."""
    sys.path.insert(0, os.path.join(HERE, "tests"))
    import oracle_lib as O
    out = {}
    m0, P0 = np.zeros(P), np.eye(P) * 0.01
    X, y, _ = synth_logit(D, dev, n_time, P)
    Xh, yh = X.cpu().numpy(), y.cpu().numpy()
    del X, y
    t0 = time.perf_counter()
    O.gibbs(yh, Xh, np.ones(n_time), m0, P0, 10, 0, 20240004, 1, store_w=False)
    dt = time.perf_counter() - t0
    out["timed"] = {"rows": n_time, "P": P, "sweeps": 10, "seconds": dt, "sweeps_per_s": 10 / dt, "cores": 1,
                    "note": "oracle bl_o_gibbs (C restatement of Logit.hpp:402-481, constrained draw, serial like the "
                            "reference: Notes/benchmarks.tex:103-104); per-sweep cost is linear in the rows"}
    if post_n > 0:
        burn, samp, seed = 200, post_samp, 20240005
        X, y, bt = synth_logit(D, dev, post_n, P)
        nn = torch.ones(post_n, dtype=torch.float64, device=dev)
        shard = D.GibbsShard(X, y, nn, seed=seed)
        shard.set_prior(m0, P0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        bg = shard.run_stream(samp, burn, constrain=1)["beta"]
        tg = time.perf_counter() - t0
        shard.close()
        Xh, yh = X.cpu().numpy(), y.cpu().numpy()
        t0 = time.perf_counter()
        _, bc = O.gibbs(yh, Xh, np.ones(post_n), m0, P0, samp, burn, seed, 1, store_w=False)
        tc = time.perf_counter() - t0
        same = np.all(np.abs(bg - bc) <= 1e-8 * (1.0 + np.abs(bc)), axis=1)
        lock = int(np.argmin(same)) if not same.all() else samp
        se = np.sqrt(batch_mcse(bg) ** 2 + batch_mcse(bc) ** 2)
        zmax = float(np.max(np.abs(bg.mean(0) - bc.mean(0)) / np.maximum(se, 1e-300)))
        sdr = bg.std(0, ddof=1) / bc.std(0, ddof=1)
        out["posterior"] = {
            "problem": f"C4 generator at N={post_n}, P={P}, prior N(0, 100 I), beta_0 = 0, burn {burn} + samp {samp}, "
                       f"constrained draw (the reference's active draw), chain seed {seed} on both",
            "gpu_seconds": tg, "cpu_seconds": tc, "cpu_cores": 1,
            "sweeps_identical_to_1e-8_before_the_chains_part": lock,
            "max_abs_diff_of_posterior_means_in_mcse": zmax,
            "sd_ratio_gpu_over_cpu_min_max": [float(sdr.min()), float(sdr.max())],
            "tolerance": "posterior means within 5 combined batch-means MCSE, posterior sds within 25 % "
                         "(chains of 2000 correlated draws); fp64 both sides",
            "within_tolerance": bool(zmax < 5.0 and sdr.min() > 0.75 and sdr.max() < 1.25),
            "beta_mean_gpu_head": [float(v) for v in bg.mean(0)[:3]], "beta_mean_cpu_head": [float(v) for v in bc.mean(0)[:3]],
            "beta_sd_gpu_head": [float(v) for v in bg.std(0, ddof=1)[:3]], "beta_sd_cpu_head": [float(v) for v in bc.std(0, ddof=1)[:3]],
        }
    return out


def cpu_hybrid(ncores, sample=2_000_000):
    """C3's CPU figure: the oracle's literal restatement of rpg_hybrid's loop (LogitWrapper.cpp:129-167 over
    PolyaGamma*.cpp) on a bounded sample, OpenMP schedule(dynamic) over the host cores."""
    sys.path.insert(0, os.path.join(HERE, "tests"))
    import oracle_lib as O
    rng = np.random.default_rng(20240001)
    h = rng.integers(1, 51, sample).astype(float)
    z = rng.normal(0.0, 2 ** 0.5, sample)
    t0 = time.perf_counter()
    O.rpg_hybrid(sample, h, z, 20240002, threads=ncores, literal=True)
    return sample / (time.perf_counter() - t0)


def mlogit_bench(N, P=32, J=5, samp=20, burn=5):
    """mult_gibbs as the R wrapper's .C call reaches it (LogitWrapper.R:395; MultLogit.hpp:261-372: J-1 category sweeps per
    sweep): host buffers in (X, y, n uploaded through pinned staging), omega of every kept sweep copied back through a
    device ring while the next sweep runs.  The caller's buffers are allocated and touched BEFORE the timed call, as .C's
    are (R copies every argument): a fresh numpy.zeros would put the host's first-touch page faults into the number."""
    import bayeslogit_amd as bl
    from bayeslogit_amd import _lib
    rng = np.random.default_rng(20240006)
    X = rng.normal(size=(N, P)) / P ** 0.5
    X[:, -1] = 1.0
    B = rng.normal(size=(P, J - 1)) * 0.5
    eta = np.concatenate([X @ B, np.zeros((N, 1))], axis=1)
    cum = (np.exp(eta) / np.exp(eta).sum(1, keepdims=True)).cumsum(1)
    k = (rng.uniform(size=N)[:, None] > cum).sum(1)
    y = np.zeros((N, J - 1))
    for j in range(J - 1):
        y[k == j, j] = 1.0
    del eta, cum
    n = np.ones(N)
    m0 = np.zeros((P, J - 1), order="F")
    P0 = np.asfortranarray(np.repeat((np.eye(P) * 0.01)[:, :, None], J - 1, axis=2))
    w = np.empty((samp, J - 1, N))
    w.fill(0.0)
    beta = np.zeros((samp, J - 1, P))
    bl.set_seed(20240007)
    c_i = lambda v: ctypes.byref(ctypes.c_int(int(v)))
    dp = lambda a: a.ctypes.data_as(_lib.c_dp)
    t0 = time.perf_counter()
    _lib.lib().mult_gibbs(dp(w), dp(beta), dp(y), dp(X), dp(n), dp(m0), dp(P0), c_i(N), c_i(P), c_i(J), c_i(samp), c_i(burn))
    dt = time.perf_counter() - t0
    bytes_in = 8.0 * (N * P + N * (J - 1) + N)
    bytes_out = 8.0 * N * (J - 1) * samp
    return {"workload": f"mult_gibbs N={N}, P={P}, J={J}, burn {burn} + samp {samp} through the .C boundary (host X, y, n in: "
                        f"{bytes_in / 1e9:.2f} GB; omega of every kept sweep out: {bytes_out / 1e9:.2f} GB)",
            "seconds": dt, "sweeps_per_s_incl_host_transfers": (samp + burn) / dt,
            "category_sweeps_per_s": (samp + burn) * (J - 1) / dt,
            "beta_err_max_last_sweep": float(np.abs(beta[-1].T - B).max()),
            "omega_mean": float(w.mean())}


def combine_bench(N=1_000_000, P=64):
    """combine() (LogitWrapper.cpp:279-310 / Logit::compress, Logit.hpp:192-270: an O(N^2 P) list walk in the reference)
    through the .C symbol at a size the replacement exists for: N rows of which a third repeat an earlier row."""
    from bayeslogit_amd import _lib
    rng = np.random.default_rng(20240008)
    uniq = 2 * N // 3
    base = rng.normal(size=(uniq, P))
    pick = np.concatenate([np.arange(uniq), rng.integers(0, uniq // 4, N - uniq)])
    rng.shuffle(pick)
    X = np.ascontiguousarray(base[pick])
    y = rng.uniform(size=N)
    n = rng.integers(1, 5, N).astype(np.float64)
    Nc = ctypes.c_int(N)
    n_sum = float(n.sum())
    dp = lambda a: a.ctypes.data_as(_lib.c_dp)
    t0 = time.perf_counter()
    _lib.lib().combine(dp(y), dp(X), dp(n), ctypes.byref(Nc), ctypes.byref(ctypes.c_int(P)))
    dt = time.perf_counter() - t0
    return {"workload": f"combine N={N}, P={P}, {N - uniq} rows repeat an earlier one; .C boundary (host buffers in and out)",
            "rows_out": Nc.value, "rows_out_expected": uniq, "seconds": dt, "M_rows_per_s": N / dt / 1e6,
            "n_total_preserved": bool(abs(float(n[:Nc.value].sum()) - n_sum) < 1e-9 * n_sum)}


def gibbs_bench(N, P, sweeps, tag, rank, world, dev, D, DistGibbs, shard_range, chain=0):
    """Gibbs sweeps/s on an N x P problem with rows sharded over the ranks: per sweep one pass over this
    rank's rows (psi, omega, X' Omega X), one P*P all-reduce, the replicated P x P stage."""
    lo, hi = shard_range(N, rank, world)
    nl = hi - lo
    X, y, bt = synth_logit(D, dev, nl, P, lo)
    nn = torch.ones(nl, dtype=torch.float64, device=dev)
    shard = D.GibbsShard(X, y, nn, seed=20240004, idx0=lo)
    drv = DistGibbs(shard)
    drv.setup(np.zeros(P), np.eye(P) * 0.01, np.zeros(P))
    res = {}
    for name, con in (("constrained", 1), ("unconstrained", 0)):
        sw = [0]
        ks, ka, kb = [], [], []

        def gstep():
            e0, e1, e2, e3 = (torch.cuda.Event(enable_timing=True) for _ in range(4))
            e0.record()
            shard.sweep_local(sw[0], None)
            e1.record()
            drv._all_reduce(shard.pp())
            e2.record()
            shard.draw_beta(sw[0], con)
            e3.record()
            ks.append((e0, e1))
            ka.append((e1, e2))
            kb.append((e2, e3))
            sw[0] += 1

        shard.set_beta(np.zeros(P))
        for _ in range(3):
            gstep()
        ks.clear(), ka.clear(), kb.clear()
        barrier_sync(world)
        t0 = time.perf_counter()
        for _ in range(sweeps):
            gstep()
        barrier_sync(world)
        gw = max_over_ranks(time.perf_counter() - t0, world, dev)
        sweep_ms = float(np.mean([p.elapsed_time(q) for p, q in ks]))
        ar_ms = float(np.mean([p.elapsed_time(q) for p, q in ka]))
        beta_ms = float(np.mean([p.elapsed_time(q) for p, q in kb]))
        res[name] = {"sweeps_per_s": sweeps / gw, "ms_per_sweep": gw / sweeps * 1e3,
                     "sweep_kernel_ms": sweep_ms, "allreduce_in_sweep_ms": ar_ms, "beta_stage_ms": beta_ms,
                     "allreduce_plus_beta_ms": ar_ms + beta_ms}
    exchange = {"allreduce_us": time_allreduce(P * P, world, dev), "bytes": 8 * P * P, "world_size": world,
                "backend": (dist.get_backend() if world > 1 else None),
                "note": "the sweep's only exchange: all-reduce(sum) of the P x P fp64 partial X'Omega X, timed alone (50 calls "
                        "between barriers, max over ranks); allreduce_in_sweep_ms is the same call inside the sweep, events "
                        "on the compute stream, so it includes waiting for the slowest rank's sweep kernels"}
    D.sync_status()
    chain_out = None
    if chain > 0:
        # the reference's run shape (Logit.hpp:460-481): beta = 0, 100 burn-in sweeps, `chain` sampling sweeps with
        # the fork's active (constrained) draw; omega is not stored; moments of beta accumulated on the device
        burn = 100
        shard.set_beta(np.zeros(P))
        bsum = torch.zeros(P, dtype=torch.float64, device=dev)
        bsq = torch.zeros(P, dtype=torch.float64, device=dev)
        for s_ in range(burn):
            drv.sweep(s_, 1)
        barrier_sync(world)
        t0 = time.perf_counter()
        for s_ in range(burn, burn + chain):
            drv.sweep(s_, 1)
            b = shard.beta()
            bsum += b
            bsq += b * b
        barrier_sync(world)
        cw = max_over_ranks(time.perf_counter() - t0, world, dev)
        D.sync_status()
        mean = (bsum / chain).cpu().numpy()
        sd = np.sqrt(np.maximum((bsq / chain).cpu().numpy() - mean * mean, 0.0) * chain / max(chain - 1, 1))
        bt_h = bt.cpu().numpy()
        chain_out = {"burn": burn, "samp": chain, "sweeps_per_s_sampling_phase": chain / cw,
                     "beta_post_mean_head": [float(v) for v in mean[:4]], "beta_post_sd_head": [float(v) for v in sd[:4]],
                     "beta_post_mean_intercept": float(mean[-1]), "beta_true_head": [float(v) for v in bt_h[:4]],
                     "beta_true_intercept": float(bt_h[-1]),
                     "max_abs_z": float(np.max(np.abs(mean - bt_h) / np.maximum(sd, 1e-300)))}
    sk = res["constrained"]["sweep_kernel_ms"]
    gb = 8.0 * nl * P / (sk * 1e-3) / 1e9
    if P == 64:
        # kernels_sweep1.hip: psi, the draws and X' Omega X in ONE pass over X (rows that leave the fast path: second kernel)
        roof = {"kernel": "k_sweep_once64 + k_sweep_deferred64 + k_reduce_q4 (one sweep: ONE pass over X)", "bound": "hbm",
                "achieved": gb, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gb / HBM_PEAK_GBS,
                "algorithmic_bytes_per_launch": 8 * nl * P,
                "note": "algorithmic bytes = X once (SURVEY 8d), which is what the sweep reads; what bounds it is the "
                        "instruction total (545 vector + 144 small matrix instructions per 16 rows; fp64 vector and "
                        "matrix instructions do not overlap on gfx950), DESIGN.md 4.3",
                }
        D.sweep_deferred_rows()                    # (reset) rows of ONE sweep at the chain's current beta that left the fast path
        shard.sweep_local(1 << 20, None)
        roof["deferred_rows_per_sweep"] = D.sweep_deferred_rows()
        t1, src = pmc_traffic("k_sweep_once64")
        t2, _ = pmc_traffic("k_sweep_deferred64")
        roof["traffic"] = (t1 + (t2 or 0.0)) if (t1 and N == 10_000_000 and world == 1) else None
        roof["traffic_source"] = src
        roof["traffic_measured_in_this_run"] = False
    elif P < 64:
        nbk = (P + 15) // 16
        kernels = f"k_psi_omega_nb<{nbk},0> + k_xwx_mfma<{nbk}> + k_reduce_fused<{nbk}>"
        roof = {"kernel": kernels + " (one sweep: two passes over X)", "bound": "hbm", "achieved": gb,
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gb / HBM_PEAK_GBS,
                "algorithmic_bytes_per_launch": 8 * nl * P,
                "note": "algorithmic bytes = X once (SURVEY 8d); the sweep reads X twice (DESIGN.md 4.3)"}
        exact = "true" if P == 16 * nbk else "false"
        t1, src = pmc_traffic(f"k_psi_omega_nb<{nbk}, 0, {exact}>")
        t2, _ = pmc_traffic(f"k_xwx_mfma<{nbk}, {exact}>")
        roof["traffic"] = (t1 + t2) if (t1 and t2 and N == 10_000_000 and world == 1) else None
        roof["traffic_source"] = src
    else:
        nbk = P // 16
        tri = nbk * (nbk + 1) // 2
        flops = 2.0 * nl * tri * 256                       # upper-triangle 16x16 blocks of the rank-N update
        tf = flops / (sk * 1e-3) / 1e12
        xk = ("k_xwx_q4_blk16 + k_reduce_q4_blk16" if P == 256 else "k_xwx_q4_big<8> + k_reduce_q4_big<8>" if P == 128
              else f"k_xwx_mfma_big<{nbk},8> + k_reduce_big<{nbk}>")
        if P == 256:
            # kernels_sweep256.hip: psi, the draws and X' Omega X in ONE pass over X (rows that leave the fast path: second kernel)
            t1, src = pmc_traffic("k_sweep_once256")
            t2, _ = pmc_traffic("k_sweep_deferred256")
            t3, _ = pmc_traffic("k_reduce_256")
            kname = "k_sweep_once256 + k_sweep_deferred256 + k_reduce_256 (one sweep: ONE pass over X)"
            traffic = (t1 + (t2 or 0.0) + (t3 or 0.0)) if (t1 and nl == 12_500_000) else None
        else:
            t1, src = pmc_traffic(f"k_psi_omega_nb<{nbk}, 0, true>")
            t2, _ = pmc_traffic("k_xwx_q4_big<8>")
            kname = f"k_psi_omega_nb<{nbk},0> + " + xk + " (one sweep: two passes over X)"
            traffic = None
        roof = {"kernel": kname, "bound": "mfma",
                "achieved": tf, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / FP64_MFMA_PEAK_TFLOPS,
                "traffic": traffic, "traffic_source": src,
                "traffic_measured_in_this_run": False, "algorithmic_bytes_per_launch": 8 * nl * P,
                "algorithmic_flops_per_launch": flops,
                "hbm_view": {"achieved_GBs": gb, "peak_GBs": HBM_PEAK_GBS, "frac": gb / HBM_PEAK_GBS},
                "note": "whole sweep time against the fp64 MFMA peak: upper-triangle flops of the rank-N update "
                        "(2 x 136 cells x 256 per row); the small matrix instruction sustains 75 of the nominal 78.6 TFLOP/s "
                        "(mfma_f64 below)"}
        if P == 256:
            D.sweep_deferred_rows()
            shard.sweep_local(1 << 20, None)
            roof["deferred_rows_per_sweep"] = D.sweep_deferred_rows()
    roof["kernel_ms"] = sk
    # the fp64 matrix pipe as measured on this GPU (register-only MFMA loop, 2 waves/SIMD): the X'Omega X pass
    # cannot take less than its flops at that rate
    tri_b = (P // 16) * (P // 16 + 1) // 2 if P % 16 == 0 else ((P + 15) // 16) * ((P + 15) // 16 + 1) // 2
    xwx_flops = 2.0 * nl * tri_b * 256
    sustained = D.mfma_f64_sustained_tflops(2)
    sustained_small = D.mfma_f64_sustained_tflops(2, small=True)
    roof["mfma_f64"] = {"sustained_tflops_measured": sustained, "nominal_tflops": FP64_MFMA_PEAK_TFLOPS,
                        "small_instruction_tflops_measured": sustained_small,
                        "xwx_flops_per_sweep": xwx_flops, "xwx_ms_at_sustained": xwx_flops / (sustained * 1e12) * 1e3,
                        "xwx_ms_at_small_instruction_rate": xwx_flops / (sustained_small * 1e12) * 1e3,
                        "note": "register-only loops, 2 waves/SIMD: v_mfma_f64_16x16x4_f64 (sustained_tflops_measured) and "
                                "v_mfma_f64_4x4x4_4b_f64 (small_instruction_tflops_measured), which X' Omega X runs on for "
                                "P = 64, 128, 256 (P < 64 and padded P: the big instruction)"}
    if roof["bound"] == "mfma":
        roof["frac_of_sustained"] = roof["achieved"] / sustained
    outg = {
        "metric": "Gibbs sweeps/sec", "workload": f"{tag}: logit Gibbs N={N}, P={P}, fp64, omega not stored, "
        f"rows sharded over {world} GPU(s), one P*P all-reduce per sweep",
        "scaling": "strong" if tag == "C4" else "weak", "sweeps_timed": sweeps,
        "value": res["constrained"]["sweeps_per_s"], "unit": "sweeps/s",
        "beta_draw": "constrained = the reference's active draw (Logit.hpp:322-400); "
                     "unconstrained = Logit.hpp:291-320",
        **res,
        "exchange": exchange,
        "roofline": roof,
        "chain": chain_out,
    }
    shard.close()
    del X, y, nn
    return outg


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(a, sys.argv[1:]))          # before anything below touches the GPU
    # stdout carries ONE JSON line and nothing else: from here on file descriptor 1 is stderr (library chatter -- gloo prints
    # its rank connections on stdout -- and the .C entry points' C-level messages go there), and the line is written to the
    # saved descriptor at the end
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    ndev = torch.cuda.device_count()
    devidx = (local % ndev) if world > 1 else 0       # local == device on a real multi-GPU node
    dev = torch.device("cuda", devidx)
    torch.cuda.set_device(dev)
    if world > 1:
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(a.backend)
    if world != a.gpus:
        print(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        sys.exit(2)

    from bayeslogit_amd import _lib
    from bayeslogit_amd import device as D
    from bayeslogit_amd.dist import DistGibbs, shard_range
    _lib.require_gpu()

    # ---------------------------------------------------------------- C2: PG(1,z) draws
    n = a.draws
    idx0 = rank * n
    z = torch.empty(n, dtype=torch.float64, device=dev)
    x = torch.empty(n, dtype=torch.float64, device=dev)
    D.fill_unif(z, 0.0, 4.0, 20240001, idx0=idx0)
    step_no = [0]

    def step():
        D.rpg_devroye(z, 1, seed=20240002, epoch=step_no[0], idx0=idx0, out=x)
        step_no[0] += 1

    wall, kern_ms = timed_steps(step, a.steps, a.warmup, world, dev)
    D.sync_status()
    draws_per_s = n * world * a.steps / wall
    mean_x = x.mean().item()
    work2 = devroye_work(D, z, 20240002, 0, idx0, n / (kern_ms * 1e-3)) if rank == 0 else None
    ach_gbs = BYTES_PER_DRAW * n / (kern_ms * 1e-3) / 1e9
    t_a, c2_traffic_src = pmc_traffic("k_rpg_devroye")
    # the committed counters are of the default workload
    c2_traffic = t_a if (t_a and n == 100_000_000) else None
    out = {
        "metric": "PG draws/sec (millions)",
        "value": draws_per_s / 1e6,
        "unit": "M draws/s",
        "n_gpus": world,
        "steps": a.steps,
        "warmup": a.warmup,
        "ms_per_step": wall / a.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": ("C2: " if n == 100_000_000 else "C2 at reduced size: ") +
                               f"N={n:.3g} PG(1,z) draws per GPU, z~Unif(0,4) (Devroye lane path)",
                   "draws_per_gpu_per_step": n, "rng": "philox4x32-10, one stream per observation",
                   "sample_mean": mean_x},
        "roofline": {
            "kernel": "k_rpg_devroye (both left-piece sampler classes in one launch: z read once)",
            # scalar fp64 transcendental work: neither HBM nor MFMA binds.  The object is the HBM
            # view the contract asks for (achieved / peak in GB/s); `limiter` and `valu` say what actually
            # bounds the kernel
            "bound": "hbm",
            "limiter": "valu (scalar fp64 transcendental work; see the valu object)",
            "achieved": ach_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach_gbs / HBM_PEAK_GBS,
            "traffic": c2_traffic, "traffic_source": c2_traffic_src, "traffic_measured_in_this_run": False,
            "work": work2,
            # the binding resource (same PMC summary, a committed profile of this command; per launch of the default 1e8-draw workload)
            "valu": {k: {"wave_insts_per_launch": pmc_entry(k).get("valu_insts_per_launch"),
                         "busy_frac_of_cu_cycles": pmc_entry(k).get("valu_busy_frac"),
                         "issue_slot_frac_lower_bound": pmc_entry(k).get("valu_issue_frac_min"),
                         "wait_frac_of_wave_cycles": pmc_entry(k).get("wait_any_frac"),
                         "issue_stall_frac_of_wave_cycles": pmc_entry(k).get("wait_inst_frac"),
                         "salu_insts_per_valu_inst": pmc_entry(k).get("salu_per_valu"),
                         "lds_bank_conflict_frac_of_lds_cycles": pmc_entry(k).get("lds_conflict_frac"),
                         "active_lane_frac": pmc_entry(k).get("active_lane_frac"),
                         "source": "committed profile " + str(c2_traffic_src) + " (rocprofv3 --pmc passes of this command)"}
                     for k in ("k_rpg_devroye",)},
            "kernel_ms": kern_ms,
            "algorithmic_bytes_per_launch": BYTES_PER_DRAW * n,
            "draws_per_s_kernel": n / (kern_ms * 1e-3),
        },
    }
    del z, x

    if not a.no_mixed:
        zz = torch.empty(n, dtype=torch.float64, device=dev)
        hh = torch.empty(n, dtype=torch.float64, device=dev)
        xx = torch.empty(n, dtype=torch.float64, device=dev)
        D.fill_norm(zz, 0.0, 2.0 ** 0.5, 20240001, idx0=idx0)
        D.fill_shape(hh, 50, 20240001, epoch=1, idx0=idx0)
        w3, k3 = timed_steps(lambda: D.rpg_hybrid(hh, zz, seed=20240002, idx0=idx0, out=xx), a.mixed_steps, 1, world, dev)
        D.sync_status()
        gb3 = BYTES_PER_DRAW_VEC * n / (k3 * 1e-3) / 1e9
        # every launch of one rpg_hybrid call, in launch order (profiles/ kernel names)
        c3_kernels = ("bl::k_rpg_tasks<bl::SpPolicy>", "bl::k_rpg_tasks<bl::AltPolicy>", "k_rpg_hybrid_class<2>",
                      "k_rpg_hybrid_class<5>", "k_rpg_hybrid_class<1>")
        tq = {k: pmc_entry(k) for k in c3_kernels}
        per_kernel_traffic = {k: v.get("hbm_bytes_per_launch") for k, v in tq.items()}
        have_all = n == 100_000_000 and per_kernel_traffic[c3_kernels[0]] and per_kernel_traffic[c3_kernels[1]]
        # a class without members returns at once inside rpg_hybrid (C3: no b > 170, no b < 1); the committed per-kernel
        # averages of those two kernels include the stand-alone passes of the branch table below, which do scan
        c3_class_of = {"k_rpg_hybrid_class<5>": "normal_approximation", "k_rpg_hybrid_class<1>": "sum_of_gammas"}
        # per-branch rates: every class pass alone, event-timed, next to the replayed attempt counts
        branches = None
        if rank == 0:
            cnt = D.count_blocks(hh[:min(n, 20_000_000)], zz[:min(n, 20_000_000)], seed=20240002, idx0=idx0)
            scale = n / min(n, 20_000_000)
            branches = {}
            for cls, name in ((4, "saddle_point"), (3, "alternating_series"), (2, "devroye"), (5, "normal_approximation"),
                              (1, "sum_of_gammas")):
                ms = []
                for rep in range(3):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    D.rpg_hybrid_class(hh, zz, cls, seed=20240002, idx0=idx0, out=xx)
                    e1.record()
                    e1.synchronize()
                    if rep:
                        ms.append(e0.elapsed_time(e1))
                c = cnt.get(name, {"observations": 0, "draws": 0, "blocks": 0})
                obs = c["observations"] * scale
                branches[name] = {"observations": int(obs), "share": obs / n, "kernel_ms_alone": float(np.mean(ms)),
                                  "M_observations_per_s": (obs / (np.mean(ms) * 1e-3) / 1e6) if obs else None,
                                  "pg_draws_per_observation": (c["draws"] / c["observations"]) if c["observations"] else None,
                                  "attempts_per_draw": (c["blocks"] / c["draws"]) if c["draws"] and c["blocks"] else None}
            D.sync_status()
            branches["note"] = ("each class pass of rpg_hybrid launched alone over the whole vector (bl_diag_rpg_hybrid_class_dev: "
                                "it scans every shape and draws its own class), HIP events, mean of 2 after a warm-up; an empty "
                                "class's pass alone still scans (inside rpg_hybrid it returns at once: the first pass counts "
                                "the classes); attempts = Philox blocks from an exact replay of the streams of the first "
                                f"{min(n, 20_000_000)} observations (alternating series: per abridged draw of "
                                "PolyaGammaAlt::draw's sum; saddle point: per proposal attempt, the reference's `iter` counts "
                                "iterations of its outer loop only)")
        out["mixed"] = {
            "workload": ("C3: " if n == 100_000_000 else "C3 at reduced size: ") +
                        f"N={n:.3g} draws per GPU, b in {{1..50}} (4 % Devroye, 22 % alternating series, 74 % saddle point), "
                        "z~N(0,sd^2=2), through rpg_hybrid",
            "value": n * world * a.mixed_steps / w3 / 1e6, "unit": "M draws/s", "M_draws_per_s": n * world * a.mixed_steps / w3 / 1e6,
            "steps": a.mixed_steps, "ms_per_step": w3 / a.mixed_steps * 1e3, "sample_mean": xx.mean().item(),
            "branches": branches,
            "roofline": {
                "kernel": "k_rpg_tasks<SpPolicy> (first pass: also counts the classes and writes the b <= 0 zeros) + "
                          "k_rpg_tasks<AltPolicy> + k_rpg_hybrid_class<2,5,1> (one launch per sampler class; a class without "
                          "members returns at once; no zeroing launch)",
                "bound": "hbm", "limiter": "valu (scalar fp64 transcendental work; see the valu object)",
                "achieved": gb3, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gb3 / HBM_PEAK_GBS,
                # the sum over ALL launches of one call (round 2 summed the two task kernels only)
                "traffic": (sum((v or 0.0) for k, v in per_kernel_traffic.items()
                                if k not in c3_class_of or not branches or branches[c3_class_of[k]]["observations"] > 0)
                            if have_all else None),
                "traffic_per_kernel": per_kernel_traffic if have_all else None,
                "traffic_source": c2_traffic_src, "traffic_measured_in_this_run": False,
                "kernel_ms": k3, "algorithmic_bytes_per_launch": BYTES_PER_DRAW_VEC * n,
                "valu": {k: {"wave_insts_per_launch": v.get("valu_insts_per_launch"),
                             "busy_frac_of_cu_cycles": v.get("valu_busy_frac"),
                             "issue_slot_frac_lower_bound": v.get("valu_issue_frac_min"),
                             "wait_frac_of_wave_cycles": v.get("wait_any_frac"),
                             "issue_stall_frac_of_wave_cycles": v.get("wait_inst_frac"),
                             "active_lane_frac": v.get("active_lane_frac")} for k, v in tq.items() if v},
            },
        }
        del zz, hh, xx

    # ---------------------------------------------------------------- C4: Gibbs sweeps
    if not a.no_gibbs:
        out["gibbs"] = gibbs_bench(a.gibbs_n, a.gibbs_p, a.gibbs_sweeps, "C4", rank, world, dev, D, DistGibbs,
                                   shard_range, chain=a.gibbs_chain)
    # mlogit through the .C boundary (host buffers in and out, omega of every sweep stored as the reference does)
    if not a.no_gibbs and rank == 0 and world == 1 and a.mlogit_n > 0:
        out["gibbs"]["mlogit"] = mlogit_bench(a.mlogit_n)
        out["gibbs"]["combine"] = combine_bench(a.mlogit_n)
    # C5: N = 1e8, P = 256 over 8 GPUs = 12.5e6 rows (25.6 GB) per GPU; run here with that shard per rank
    if not a.no_c5:
        tag5 = "C5" if (world == 8 and a.c5_rows == 12_500_000) else \
               f"C5 shard ({a.c5_rows} rows per rank; the full N = 1e8 problem is 8 ranks x 12.5e6)"
        out["gibbs_c5"] = gibbs_bench(a.c5_rows * world, 256, a.c5_sweeps, tag5, rank, world, dev, D,
                                      DistGibbs, shard_range)

    # ---------------------------------------------------------------- CPU baseline
    if rank == 0 and world == 1 and not a.no_cpu:
        ncores = min(len(os.sched_getaffinity(0)), 16)   # a 1-GPU box's CPU share is 16 cores
        sample = 3_000_000 * max(1, min(ncores, 16))
        one, allc = cpu_baseline(sample, ncores)
        out["vs_baseline"] = out["value"] / (allc / 1e6)
        out["vs_baseline_note"] = (f"value / cpu_baseline.value: the oracle (this repository's C restatement of the reference's "
                                   f"loops; parity to the reference's own binary unpinned, DESIGN.md 1) on {ncores} host cores of this "
                                   "box, same run.  BASELINE.md's only published figure for this metric is 2.3-2.5 M PG(1,0) draws/s "
                                   "on one core of a 2012 desktop (Code/C/test_pgpar.cpp:74-76): see vs_published_pg10_serial")
        out["vs_published_pg10_serial"] = out["value"] / 2.4
        out["cpu_baseline"] = {
            "value": allc / 1e6, "unit": "M draws/s", "cores": ncores, "kind": "port",
            "sample": f"{sample} PG(1,z) draws, z~Unif(0,4), oracle (C restatement of PolyaGamma.cpp:151-202), "
                      f"OpenMP schedule(dynamic) as PolyaGammaOMP.h:61-71; 1-core rate on {min(sample, 4_000_000)} draws",
            "one_core_M_draws_per_s": one / 1e6,
        }
        if not a.no_mixed:
            out["cpu_baseline"]["mixed_M_draws_per_s"] = cpu_hybrid(ncores) / 1e6
            out["cpu_baseline"]["mixed_sample"] = ("2000000 draws of the C3 mix, oracle's literal restatement of rpg_hybrid "
                                                  f"(LogitWrapper.cpp:129-167), OpenMP on {ncores} cores")
        if not a.no_gibbs:
            cg = cpu_gibbs(D, dev, a.cpu_gibbs_n, a.gibbs_p, a.post_n, a.post_samp)
            tm = cg["timed"]
            tm["sweeps_per_s_extrapolated_to_gibbs_n"] = tm["sweeps_per_s"] * tm["rows"] / a.gibbs_n
            tm["extrapolation"] = f"linear in rows: x {tm['rows']}/{a.gibbs_n}"
            out["cpu_baseline"]["gibbs"] = tm
            out["gibbs"]["vs_cpu_1_core_extrapolated"] = out["gibbs"]["value"] / tm["sweeps_per_s_extrapolated_to_gibbs_n"]
            out["gibbs"]["vs_cpu_note"] = ("CPU side = the oracle (repo restatement of Logit.hpp:402-481; parity to the reference "
                                           "binary unpinned), one core, timed at cpu_baseline.gibbs.rows rows and extrapolated "
                                           "linearly in the rows")
            if "posterior" in cg:
                cg["posterior"]["cpu_side"] = "oracle (repo restatement), parity to the reference binary unpinned"
                out["gibbs"]["posterior_vs_cpu"] = cg["posterior"]

    if world > 1:
        dist.destroy_process_group()
    sys.stdout.flush()
    ctypes.CDLL(None).fflush(None)
    if rank == 0:
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    os.close(real_stdout)


if __name__ == "__main__":
    main()
