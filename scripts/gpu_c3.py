"""C3 (mixed shapes) timing on the GPU box: the whole rpg_hybrid launch sequence and its two large classes alone.
    python scripts/gpu_c3.py [N]"""
import sys
import time

import torch

sys.path.insert(0, ".")
from bayeslogit_amd import device as D

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
dev = torch.device("cuda:0")
z = torch.empty(n, dtype=torch.float64, device=dev)
h = torch.empty(n, dtype=torch.float64, device=dev)
x = torch.empty(n, dtype=torch.float64, device=dev)
D.fill_norm(z, 0.0, 2 ** 0.5, 20240001)
D.fill_shape(h, 50, 20240001, epoch=1)


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


ms = timed(lambda: D.rpg_hybrid(h, z, seed=20240002, out=x))
D.sync_status()
print(f"C3 rpg_hybrid N={n}: {ms:.2f} ms  {n / ms / 1e6:.2f} G draws/s  mean {x.mean().item():.6f}")
hs = h.clone()
hs[hs <= 13] = 0.0
ms = timed(lambda: D.rpg_sp(hs, z, seed=20240002, out=x))
D.sync_status()
k = int((hs > 0).sum())
print(f"  saddle-point members only ({k}): {ms:.2f} ms  {k / ms / 1e6:.2f} G draws/s")
ha = h.clone()
ha[(ha > 13) | (ha < 3)] = 0.0
ms = timed(lambda: D.rpg_alt(ha, z, seed=20240002, out=x))
D.sync_status()
k = int((ha > 0).sum())
print(f"  alternating-series members only ({k}): {ms:.2f} ms  {k / ms / 1e6:.2f} G obs/s")
