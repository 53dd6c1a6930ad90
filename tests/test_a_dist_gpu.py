"""The N > 1 Gibbs path on the real shard: two ranks (child processes, gloo backend, both on the one GPU of the
test box) drive bayeslogit_amd.dist.DistGibbs over the HIP GibbsShard -- rows split, X'kappa all-reduced once, one
P*P all-reduce per sweep, beta drawn redundantly.  Checked: both ranks hold bit-identical beta histories without a
broadcast; the chain is the one-rank chain (omega keyed by the global row index; the two partial X'Omega X sums
add in another order than one rank's, so the comparison is to 1e-9 over a short horizon, not bit for bit).

This module sorts first on purpose: the children are started before this process has touched the GPU (on this pool
a process that has initialised the GPU must not start other programs), so it uses no `gpu` fixture before the spawn."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` without a launcher (no WORLD_SIZE): bench.py starts its two ranks as children (gloo here: both
    on the box's one GPU) and rank 0 prints ONE JSON line with n_gpus = 2, the Gibbs rows split over the ranks, the exchange
    timed alone.  Reduced sizes: the plumbing is what is tested.  First in this module: the child processes are started before
    this process has touched the GPU."""
    import json
    root = os.path.dirname(HERE)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "2", "--warmup", "1",
           "--draws", "200000", "--mixed-steps", "1", "--gibbs-n", "40000", "--gibbs-sweeps", "2", "--gibbs-chain", "0",
           "--no-cpu", "--mlogit-n", "0", "--c5-rows", "8192", "--c5-sweeps", "1"]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines[:3]                      # stdout is the one JSON line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["metric"].startswith("PG draws/sec") and d["value"] > 0
    g = d["gibbs"]
    assert "rows sharded over 2 GPU(s)" in g["workload"] and g["exchange"]["world_size"] == 2
    assert g["exchange"]["backend"] == "gloo" and g["exchange"]["allreduce_us"] > 0
    assert d["gibbs_c5"]["exchange"]["bytes"] == 8 * 256 * 256 and "cpu_baseline" not in d


def test_two_ranks_on_the_real_shard(tmp_path):
    import torch
    if torch.cuda.device_count() < 1:                  # counting devices does not initialise the GPU
        pytest.fail("gpu-marked test on a machine without a HIP device")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "dist_rank_gpu.py"), str(r), "2", str(port), str(tmp_path)],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    r0, r1 = (np.load(os.path.join(tmp_path, f"rank{r}.npz")) for r in range(2))
    assert (int(r0["lo"]), int(r0["hi"]), int(r1["lo"]), int(r1["hi"])) == (0, 15006, 15006, 30011)
    # one rank, in this process (the GPU is touched from here on)
    sys.path.insert(0, HERE)
    from dist_rank_gpu import problem, run_chain
    X, y, n, m0, P0 = problem()
    for con in (0, 1):
        assert np.array_equal(r0[f"hist{con}"], r1[f"hist{con}"]) and np.array_equal(r0[f"bp{con}"], r1[f"bp{con}"])
        hist, bp, w = run_chain(X, y, n, m0, P0, 0, 4, 2, con)
        assert np.allclose(r0[f"bp{con}"], bp, rtol=1e-12, atol=1e-13)
        assert np.allclose(r0[f"hist{con}"], hist, rtol=1e-9, atol=1e-10), np.abs(r0[f"hist{con}"] - hist).max()
        # omega of the sweep after the last beta: rows of both ranks concatenated = the one-rank pass (same streams)
        wcat = np.concatenate([r0[f"w{con}"], r1[f"w{con}"]])
        assert np.mean(np.isclose(wcat, w, rtol=1e-8, atol=0)) > 1 - 1e-4
