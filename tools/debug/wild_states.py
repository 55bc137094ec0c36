"""The fast mode against the literal kernel on wild INITIAL states (a NaN, an infinity, a negative volume, a layer far
above its capacity) and wild area / extra values."""
import sys
sys.path.insert(0, '.')
sys.path.insert(0, 'oracle')
import numpy as np
import bench
from smartpy_amd import engine
import lhs_oracle
n = 192
params = lhs_oracle.lhs_params(n, seed=6)
f = bench.synthetic_forcing(0, True)[0][:24 * 100]
rng = np.random.default_rng(2)
init = np.abs(rng.normal(1e5, 5e4, (n, 12)))
wild = [(0, np.nan), (3, np.nan), (5, np.nan), (8, np.nan), (11, np.nan), (0, np.inf), (6, np.inf), (11, np.inf), (2, -1e4),
        (5, -1e3), (11, -1e5), (7, 1e12), (10, 0.0), (11, 0.0), (4, -0.0)]
rows = {}
for k, (col, val) in enumerate(wild):
    init[k * 12 + 1, col] = val
    rows[k * 12 + 1] = (col, val)
with np.errstate(all='ignore'):
    fast = engine.run_ensemble(params, f, bench.AREA, 3600.0, 24 * 10, 24, initial=init)
    lit = engine.run_ensemble(params, f, bench.AREA, 3600.0, 24 * 10, 24, initial=init, math_mode='literal')
print(fast._prepared.describe())
a, b = fast.discharge.cpu().numpy(), lit.discharge.cpu().numpy()
for r in range(n):
    na, nb = np.isnan(a[r]), np.isnan(b[r])
    ok = ~na & ~nb
    rel = np.max(np.abs(a[r][ok] - b[r][ok]) / np.maximum(np.abs(b[r][ok]), 1e-300)) if ok.any() else 0.0
    if not np.array_equal(na, nb) or rel > 1e-9:
        print('row', r, 'wild', rows.get(r), 'NaN fast %d literal %d of %d, rel on finite %.2e' % (na.sum(), nb.sum(), len(na), rel), a[r][:2], b[r][:2])
print('done')
