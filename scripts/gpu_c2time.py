import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from bayeslogit_amd import device as D
dev = torch.device("cuda:0")
n = 100_000_000
z = torch.empty(n, dtype=torch.float64, device=dev); D.fill_unif(z, 0.0, 4.0, 20240001)
x = torch.empty_like(z)
def tm(zz):
    D.rpg_devroye(zz, 1, seed=20240002, out=x); torch.cuda.synchronize(); ts = []
    for _ in range(4):
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record(); D.rpg_devroye(zz, 1, seed=20240002, out=x); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return float(np.median(ts))
t1 = tm(z); m = x.mean().item(); z.zero_(); t0 = tm(z)
print(f"V={os.environ.get('BL_VARIANT')} W={os.environ.get('BL_WAVES')}: U(0,4) {t1:.2f} ms  z=0 {t0:.2f} ms  mean {m:.6f}")
