"""C2 (PG(1,z), z ~ U(0,4)) timing on the GPU box, with the mean as a sanity value.
    python scripts/gpu_c2.py [N]"""
import sys
import time

import torch

sys.path.insert(0, ".")
from bayeslogit_amd import device as D

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
dev = torch.device("cuda:0")
z = torch.empty(n, dtype=torch.float64, device=dev)
x = torch.empty(n, dtype=torch.float64, device=dev)
D.fill_unif(z, 0.0, 4.0, 20240001)
for rep in range(3):
    D.rpg_devroye(z, 1, seed=20240002, epoch=rep, out=x)
torch.cuda.synchronize()
t0 = time.perf_counter()
K = 20
for rep in range(K):
    D.rpg_devroye(z, 1, seed=20240002, epoch=rep, out=x)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / K * 1e3
D.sync_status()
print(f"C2 rpg_devroye N={n}: {ms:.3f} ms  {n / ms / 1e6:.2f} G draws/s  mean {x.mean().item():.8f}")
for lo, hi in ((0.0, 3.0), (3.2, 4.0)):
    D.fill_unif(z, lo, hi, 20240001)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for rep in range(5):
        D.rpg_devroye(z, 1, seed=20240002, epoch=rep, out=x)
    torch.cuda.synchronize()
    print(f"  z in ({lo},{hi}): {(time.perf_counter() - t0) / 5 * 1e3:.3f} ms")
