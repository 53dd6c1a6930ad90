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
