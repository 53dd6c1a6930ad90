"""N > 1 path on CPU: world_size-2 gloo run of bayeslogit_amd.dist.DistGibbs with an oracle-backed stand-in
shard (the oracle is the checker here; the product shard is the HIP GibbsShard, tested under -m gpu).
Checks that sharding rows over ranks + one all-reduce of P*P per sweep reproduces the unsharded chain."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from bayeslogit_amd.dist import DistGibbs, shard_range


class OracleShard:
    """GibbsShard interface on CPU tensors, arithmetic by the oracle."""

    def __init__(self, X, y, n, seed, idx0):
        import oracle_lib as O
        self.O = O
        self.X, self.y, self.n, self.seed, self.idx0 = X, y, n, seed, idx0
        self.P = X.shape[1]
        self._pp = torch.zeros(self.P * self.P, dtype=torch.float64)
        self._bp = torch.zeros(self.P, dtype=torch.float64)
        self._beta = torch.zeros(self.P, dtype=torch.float64)

    def set_prior(self, m0, P0):
        self.m0, self.P0 = np.asarray(m0, float), np.asarray(P0, float)

    def set_beta(self, b):
        self._beta[:] = torch.as_tensor(np.asarray(b, float))

    def set_bp_local(self):
        z = np.zeros(self.P)
        self._bp[:] = torch.as_tensor(self.O.set_bP(self.y, self.X, self.n, z, np.zeros((self.P, self.P))))

    def finish_bp(self):
        self._bp += torch.as_tensor(self.P0 @ self.m0)

    def bp(self):
        return self._bp

    def pp(self):
        return self._pp

    def beta(self):
        return self._beta

    def sweep_local(self, sweep, w_out=None):
        PP, w = self.O.sweep_partial(self.X, self.n, self._beta.numpy(), self.seed, sweep, self.idx0)
        self._pp[:] = torch.as_tensor(PP.reshape(-1))
        self.last_w = w

    def draw_beta(self, sweep, constrain):
        PP = self._pp.numpy().reshape(self.P, self.P) + self.P0
        b = self.O.draw_beta(PP, self._bp.numpy(), self._beta.numpy(), self.seed, sweep, constrain)
        self._beta[:] = torch.as_tensor(b)


def problem():
    rng = np.random.default_rng(21)
    N, P = 501, 5
    X = rng.normal(size=(N, P)) / np.sqrt(P)
    X[:, -1] = 1.0
    bt = np.abs(rng.normal(size=P))
    y = (rng.uniform(size=N) < 1 / (1 + np.exp(-X @ bt))).astype(float)
    n = rng.integers(1, 3, N).astype(float)
    return X, y, n, np.linspace(0, 0.2, P), np.eye(P) * 0.3


def worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    X, y, n, m0, P0 = problem()
    lo, hi = shard_range(len(y), rank, world)
    out = {}
    for con in (0, 1):
        sh = OracleShard(X[lo:hi], y[lo:hi], n[lo:hi], seed=33, idx0=lo)
        drv = DistGibbs(sh)
        drv.setup(m0, P0, np.zeros(X.shape[1]))
        hist = drv.run(samp=4, burn=2, constrain=con)
        out[con] = (hist.numpy(), sh.bp().numpy().copy(), sh.last_w.copy())
    q.put((rank, lo, hi, out))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_matches_unsharded():
    import oracle_lib as O
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    X, y, n, m0, P0 = problem()
    assert (res[0][1], res[0][2], res[1][1], res[1][2]) == (0, 251, 251, 501)
    for con in (0, 1):
        h0, bp0, w0 = res[0][3][con]
        h1, bp1, w1 = res[1][3][con]
        # every rank holds the same bP and draws the same beta without a broadcast
        assert np.array_equal(h0, h1) and np.array_equal(bp0, bp1)
        assert np.allclose(bp0, O.set_bP(y, X, n, m0, P0), rtol=1e-12)
        # same chain as the single-process oracle (omega keyed by global row index)
        _, ref = O.gibbs(y, X, n, m0, P0, samp=4, burn=2, seed=33, constrain=con, store_w=False)
        assert np.allclose(h0, ref, rtol=1e-9, atol=1e-11), np.abs(h0 - ref).max()
    # the last sweep's omega, concatenated over ranks, is the unsharded omega of that sweep
    wfull, _ = None, None
    hist = res[0][3][0][0]
    PP, wref = O.sweep_partial(X, n, hist[-2], 33, 5, 0)
    assert np.allclose(np.concatenate([res[0][3][0][2], res[1][3][0][2]]), wref, rtol=1e-9)
