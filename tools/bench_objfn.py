#!/usr/bin/env python3
"""Bandwidth of the stored-matrix objective-function kernel (smart_objfn_matrix): HBM-bound, one pass over [R][N]."""
import sys, os, json
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smartpy_amd import engine

N, R = int(sys.argv[1]) if len(sys.argv) > 1 else 100000, 3653
g = torch.Generator(device='cuda').manual_seed(0)
sim = torch.rand((R, N), dtype=torch.float64, device='cuda', generator=g) * 5
obs = np.abs(np.random.default_rng(0).normal(2, 1, R))
obs[np.random.default_rng(1).random(R) < 0.12] = np.nan
gw = torch.rand(N, dtype=torch.float64, device='cuda', generator=g)
for _ in range(2):
    out = engine.objective_functions(sim, obs, gw, 0.12667)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 10
e0.record()
for _ in range(reps):
    out = engine.objective_functions(sim, obs, gw, 0.12667)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
valid = int(np.sum(~np.isnan(obs)))
alg = 8 * R * N                  # the matrix is read once
print(json.dumps({'kernel': 'smart_objfn_matrix', 'N': N, 'R': R, 'ms': ms, 'algorithmic_GB': alg / 1e9,
                  'GBps': alg / ms / 1e6, 'frac_of_8TBps': alg / ms / 1e6 / 8000}))
