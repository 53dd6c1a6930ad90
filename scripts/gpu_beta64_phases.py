"""Phase timing of the P = 64 constrained beta stage (BL_BETA_DEBUG=1: k_beta64's wall-clock stamps) on a C4-shaped chain.
    BL_BETA_DEBUG=1 python scripts/gpu_beta64_phases.py"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
import torch

from bayeslogit_amd import device as D

sys.argv = ['x']
import bench

dev = torch.device('cuda:0')
N, P = int(os.environ.get('BL_N', '2000000')), 64
X, y, bt = bench.synth_logit(D, dev, N, P)
nn = torch.ones(N, dtype=torch.float64, device=dev)
sh = D.GibbsShard(X, y, nn, seed=20240004)
sh.set_prior(np.zeros(P), np.eye(P) * 0.01)
sh.set_bp_local()
sh.finish_bp()
sh.set_beta(np.zeros(P))
for s in range(12):
    sh.sweep_local(s, None)
    sh.draw_beta(s, 1)
D.sync_status()
