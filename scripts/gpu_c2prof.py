"""C2 kernels only: 6 launches at z ~ U(0,4), then 6 at z = 0 (read the per-dispatch trace in order)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from bayeslogit_amd import device as D
dev = torch.device("cuda:0")
n = 100_000_000
z = torch.empty(n, dtype=torch.float64, device=dev); D.fill_unif(z, 0.0, 4.0, 20240001)
x = torch.empty_like(z)
for e in range(6):
    D.rpg_devroye(z, 1, seed=20240002, epoch=e, out=x)
torch.cuda.synchronize()
z.zero_()
for e in range(6):
    D.rpg_devroye(z, 1, seed=20240002, epoch=e, out=x)
torch.cuda.synchronize()
D.sync_status()
