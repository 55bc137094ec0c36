"""What the fast mode makes of a NaN (a missing value) in the forcing, next to the literal kernel: NaN patterns of the
discharge, for forcing constant over the day, 6-hourly and varying; a NaN in the rain, in the evaporation."""
import sys
sys.path.insert(0, '.')
sys.path.insert(0, 'oracle')
import numpy as np
import bench
from smartpy_amd import engine
import lhs_oracle
params = lhs_oracle.lhs_params(200, seed=3)
base = bench.synthetic_forcing(0, True)[0][:24 * 120]
extra = bench.EXTRA
for name, f0 in (('daily', base), ('six-hourly', bench.six_hourly_forcing(base)), ('varying', bench.hourly_varying_forcing(base))):
    for col in (0, 1):
        f = f0.copy()
        f[24 * 50 + 7, col] = np.nan
        if name != 'varying':                       # keep the run constant
            k = 24 if name == 'daily' else 6
            t0 = (24 * 50 + 7) // k * k
            f[t0:t0 + k, col] = np.nan
        with np.errstate(all='ignore'):
            fast = engine.run_ensemble(params, f, bench.AREA, 3600.0, 24 * 20, 24, extra=extra)
            lit = engine.run_ensemble(params, f, bench.AREA, 3600.0, 24 * 20, 24, extra=extra, math_mode='literal')
        a, b = fast.discharge.cpu().numpy(), lit.discharge.cpu().numpy()
        same_nan = np.array_equal(np.isnan(a), np.isnan(b))
        ok = ~np.isnan(b)
        rel = np.max(np.abs(a[ok & ~np.isnan(a)] - b[ok & ~np.isnan(a)]) / np.maximum(np.abs(b[ok & ~np.isnan(a)]), 1e-300)) if ok.any() else 0
        print('%-10s NaN in %s: %s | literal NaN share %.3f fast NaN share %.3f | same NaN pattern %s | rel on finite %.2e' % (
            name, 'rain' if col == 0 else 'peva', fast._prepared.describe()[:40], np.isnan(b).mean(), np.isnan(a).mean(), same_nan, rel))
