"""How far does the fast (STIFF) arithmetic drift from the reference on rows with dt / k > 2, as a function of dt / k?
Needs a library built with -DSMART_NO_ILLCOND (no row sent to the literal model):

    python tools/build_variants.py noill=-DSMART_NO_ILLCOND
    SMART_AMD_LIB=$PWD/smartpy_amd/csrc/libsmart_amd_noill.so python tools/debug/illcond_err.py [n_rows]
"""
import sys
sys.path.insert(0, '.')
import numpy as np
from smartpy_amd import engine as eng
from oracle import smart_oracle as so, lhs_oracle
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
rng = np.random.default_rng(12345)
days = 3653
rain = (rng.random(days) < 0.80) * rng.gamma(0.70, 4.57, days)
peva = np.maximum(0.0, 1.47 * (1 + 0.85 * np.sin(2 * np.pi * ((np.arange(days) % 365.25) - 110) / 365.25)))
params = lhs_oracle.lhs_params(n, seed=2718)
# widen the routing constants downwards so that dt / k reaches 100 on each of them
r2 = np.random.default_rng(7)
for col in (6, 7, 8, 9):
    pick = r2.random(n) < (0.6 if col == 9 else 0.15)
    params[pick, col] = np.exp(r2.uniform(np.log(0.2), np.log(24.0), pick.sum()))
extra = {'aar': 1200, 'r-o_ratio': 0.45, 'r-o_split': (0.10, 0.15, 0.15, 0.30, 0.30)}
area, dt = 175.46e6, 86400.0
f = np.stack([rain, peva], axis=1)
out = eng.run_ensemble(params, f, area, dt, 365, 1, extra=extra, group_variants=False)   # the library classifies
print(out._prepared.describe())
dis, gw, _ = so.run_batch(area, dt, 3653, 365, rain, peva, params, extra, 1, 1)
got = out.discharge.cpu().numpy()
top = np.abs(dis).max(axis=1, keepdims=True)
err = (np.abs(got - dis) / (np.abs(dis) + 1e-6 * top)).max(axis=1)      # relative, floored at 1e-6 of the row's peak
ratio = dt / (params[:, 6:10] * 3600.0)
names = ['SK', 'FK', 'GK', 'RK']
edges = [0, 1, 2, 4, 8, 12, 16, 20, 24, 32, 48, 200]
for j, name in enumerate(names):
    others_ok = np.delete(ratio, j, axis=1).max(axis=1) <= 2.0            # this constant alone is beyond 2
    print('dt/%s alone beyond 2 (the other three <= 2): max error per bin of dt/%s' % (name, name))
    for lo, hi in zip(edges[:-1], edges[1:]):
        m = others_ok & (ratio[:, j] > lo) & (ratio[:, j] <= hi)
        if m.any():
            print('   (%3d, %3d]  rows %5d  max %.2e  median %.2e' % (lo, hi, m.sum(), err[m].max(), np.median(err[m])))
fine = [1, 1.5, 2, 2.5, 3, 3.5, 4, 5, 6, 7, 8, 10, 12, 14, 16, 18, 19, 20, 21, 22, 24, 28, 32, 40, 48, 64, 100, 200]
print('dt/RK in fine bins, all rows (whatever the other three are):')
for lo, hi in zip(fine[:-1], fine[1:]):
    m = (ratio[:, 3] > lo) & (ratio[:, 3] <= hi)
    if m.any():
        print('   (%5.1f, %5.1f]  rows %5d  max %.2e  99%% %.2e  median %.2e' % (lo, hi, m.sum(), err[m].max(), np.quantile(err[m], 0.99), np.median(err[m])))
worst = np.argsort(err)[-5:]
for w in worst:
    print('row', w, 'err %.2e' % err[w], 'dt/k', np.round(ratio[w], 2))
