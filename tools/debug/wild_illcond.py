"""Which ill-conditioned rows with wild parameters differ between the fast mode and the literal kernel (the set-up of
tests/test_gpu_parity.py::test_ill_conditioned_rows_with_wild_parameters_match_the_literal_kernel)."""
import os
import sys
sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
sys.path.insert(0, 'oracle')
import numpy as np
from smartpy_amd import engine
import lhs_oracle
rng = np.random.default_rng(77)
n = 192
params = lhs_oracle.lhs_params(n, seed=5)
params[:, 9] = rng.uniform(1.0, 8.0, n)
wild = [(0, np.nan), (1, np.nan), (1, np.inf), (1, -0.5), (1, 0.0), (1, -0.0), (2, np.nan), (2, 5.0), (3, np.nan),
        (3, -1.0), (4, np.nan), (4, 0.9), (4, 0.0), (5, 1e-3), (6, 0.5), (7, np.inf), (8, 1e300), (0, 0.0), (0, -1.0)]
rows = {}
for k, (col, val) in enumerate(wild):
    params[(k * 10 + 3) % n, col] = val
    rows[(k * 10 + 3) % n] = (col, val)
T, W = 900, 120
rain = rng.gamma(0.6, 5.0, T) * (rng.random(T) < 0.6)
peva = np.maximum(0.0, rng.normal(1.5, 1.0, T))
f = np.stack([rain, peva], axis=1)
extra = {'aar': 1200, 'r-o_ratio': 0.45, 'r-o_split': (0.10, 0.15, 0.15, 0.30, 0.30)}
kw = dict(extra=extra, want_final=True)
fast = engine.run_ensemble(params, f, 175.46e6, 86400.0, W, 1, **kw)
lit = engine.run_ensemble(params, f, 175.46e6, 86400.0, W, 1, math_mode='literal', **kw)
print(fast._prepared.describe())
a, b = fast.discharge.cpu().numpy(), lit.discharge.cpu().numpy()
same = (a.view(np.uint64) == b.view(np.uint64)) | (np.isnan(a) & np.isnan(b))
print('discharge shape', a.shape)
bad = np.nonzero(~same.all(axis=1 if a.shape[0] == n else 0))[0]
for r in bad:
    col = a[r] if a.shape[0] == n else a[:, r]
    colb = b[r] if a.shape[0] == n else b[:, r]
    print('row', r, 'wild', rows.get(int(r)), 'fast', col[:3], 'literal', colb[:3])
