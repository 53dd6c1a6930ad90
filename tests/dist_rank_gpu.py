"""One rank of the two-rank run of tests/test_a_dist_gpu.py (not a test module): DistGibbs over the HIP GibbsShard,
gloo backend, every rank on cuda:0.
    python tests/dist_rank_gpu.py RANK WORLD PORT OUTDIR"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))


def problem():
    rng = np.random.default_rng(2121)
    N, P = 30011, 64
    X = rng.normal(size=(N, P)) / np.sqrt(P)
    X[:, -1] = 1.0
    bt = np.abs(rng.normal(size=P))
    bt[-1] = -0.5
    n = rng.integers(1, 3, N).astype(float)
    y = rng.binomial(n.astype(int), 1 / (1 + np.exp(-X @ bt))) / n
    return X, y, n, np.zeros(P), np.eye(P) * 0.1


def run_chain(X, y, n, m0, P0, lo, samp, burn, constrain):
    import torch
    from bayeslogit_amd import device as D
    from bayeslogit_amd.dist import DistGibbs
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64, device="cuda:0")
    sh = D.GibbsShard(t(X), t(y), t(n), seed=4242, idx0=lo)
    drv = DistGibbs(sh)
    drv.setup(m0, P0, np.zeros(X.shape[1]))
    hist = drv.run(samp=samp, burn=burn, constrain=constrain).numpy()
    w = torch.zeros(X.shape[0], dtype=torch.float64, device="cuda:0")
    sh.sweep_local(burn + samp, w)                 # one more pass: this rank's omega for the last beta
    D.sync_status()
    out = hist, sh.bp().cpu().numpy().copy(), w.cpu().numpy()
    sh.close()
    return out


def main():
    rank, world, port, outdir = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = port
    import torch.distributed as dist
    from bayeslogit_amd.dist import shard_range
    dist.init_process_group("gloo", rank=rank, world_size=world)
    X, y, n, m0, P0 = problem()
    lo, hi = shard_range(len(y), rank, world)
    res = {}
    for con in (0, 1):
        hist, bp, w = run_chain(X[lo:hi], y[lo:hi], n[lo:hi], m0, P0, lo, 4, 2, con)
        res[f"hist{con}"], res[f"bp{con}"], res[f"w{con}"] = hist, bp, w
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), lo=lo, hi=hi, **res)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
