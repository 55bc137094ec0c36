"""The fast mode against the literal kernel on forcing with negative values (a sensor offset, a correction gone wrong):
finite, so the fast kernels must agree (<= 1e-9); for forcing constant over the day, 6-hourly and varying."""
import sys
sys.path.insert(0, '.')
sys.path.insert(0, 'oracle')
import numpy as np
import bench
from smartpy_amd import engine
import lhs_oracle
params = lhs_oracle.lhs_params(300, seed=4)
base = bench.synthetic_forcing(0, True)[0][:24 * 150]
rng = np.random.default_rng(5)
for name, f0, k in (('daily', base, 24), ('six-hourly', bench.six_hourly_forcing(base), 6), ('varying', bench.hourly_varying_forcing(base), 1)):
    for col in (0, 1):
        f = f0.copy()
        days = rng.choice(len(f) // 24, 12, replace=False)
        for d in days:
            t0 = d * 24 + (rng.integers(0, 24) // k) * k
            f[t0:t0 + k, col] = -abs(f[t0, col]) - 0.01 * (d % 3)
        f[24 * 7:24 * 7 + k, col] = -0.0
        fast = engine.run_ensemble(params, f, bench.AREA, 3600.0, 24 * 20, 24, extra=bench.EXTRA)
        lit = engine.run_ensemble(params, f, bench.AREA, 3600.0, 24 * 20, 24, extra=bench.EXTRA, math_mode='literal')
        a, b = fast.discharge.cpu().numpy(), lit.discharge.cpu().numpy()
        rel = np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300))
        g = np.max(np.abs(fast.gw.cpu().numpy() - lit.gw.cpu().numpy()))
        print('%-10s negative %s: %s | max rel discharge %.2e  max abs gw %.2e' % (name, 'rain' if col == 0 else 'peva', fast._prepared.describe()[:36], rel, g))
