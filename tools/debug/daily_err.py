import sys, numpy as np
sys.path.insert(0, '.')
from smartpy_amd import engine as eng
from oracle import smart_oracle as so, lhs_oracle
rng = np.random.default_rng(12345)
days = 3653
rain = (rng.random(days) < 0.80) * rng.gamma(0.70, 4.57, days)
peva = np.maximum(0.0, 1.47 * (1 + 0.85 * np.sin(2 * np.pi * ((np.arange(days) % 365.25) - 110) / 365.25)))
params = lhs_oracle.lhs_params(10000, seed=2718)
extra = {'aar': 1200, 'r-o_ratio': 0.45, 'r-o_split': (0.10, 0.15, 0.15, 0.30, 0.30)}
area = 175.46e6
f = np.stack([rain, peva], axis=1)
out = eng.run_ensemble(params, f, area, 86400.0, 365, 1, extra=extra)
dis, gw, _ = so.run_batch(area, 86400.0, 3653, 365, rain, peva, params, extra, 1, 1)
got = out.discharge.cpu().numpy()
err = np.abs(got - dis) / np.maximum(np.abs(dis), 1e-300)
bad = np.argwhere(err > 1e-9)
print('n bad values', len(bad), 'bad samples', np.unique(bad[:, 0])[:20], 'max', err.max())
n, t = np.unravel_index(np.argmax(err), err.shape)
print('worst sample', n, 'step', t, 'params', params[n])
first = bad[bad[:, 0] == n][:, 1].min()
print('first bad step of that sample', first)
sl = slice(max(first - 3, 0), first + 6)
print('gpu   ', got[n, sl]); print('oracle', dis[n, sl]); print('relerr', err[n, sl])
print('rain', rain[sl], 'peva', peva[sl])
# per-sample breakdown: which parameter correlates
bs = np.unique(bad[:, 0])
print('RK of bad samples', np.sort(params[bs, 9])[:20], 'SK', np.sort(params[bs, 6])[:10])
lit = eng.run_ensemble(params[bs], f, area, 86400.0, 365, 1, extra=extra, math_mode='literal').discharge.cpu().numpy()
print('literal vs oracle(libm) max rel on the bad samples', (np.abs(lit - dis[bs]) / np.maximum(np.abs(dis[bs]), 1e-300)).max())
