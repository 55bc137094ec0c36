import sys, numpy as np
sys.path.insert(0, '.')
from smartpy_amd import engine as eng
from oracle import smart_oracle as so, lhs_oracle
g = np.load('tests/golden/forcing_example.npz')
rain = np.repeat(g['rain_daily'] / 24, 24); peva = np.repeat(g['peva_daily'] / 24, 24)
extra = {'aar': 1200, 'r-o_ratio': 0.45, 'r-o_split': (0.10, 0.15, 0.15, 0.30, 0.30)}
params = lhs_oracle.lhs_params(512, seed=4)
out = eng.run_ensemble(params, np.stack([rain, peva], 1), float(g['area']), 3600.0, 8760, 24, extra=extra)
dis, gw, _ = so.run_batch(float(g['area']), 3600.0, 87672, 8760, rain, peva, params, extra, 1, 24)
e = np.abs(out.gw.cpu().numpy() - gw) / np.abs(gw)
d = np.abs(out.discharge.cpu().numpy() - dis) / np.maximum(np.abs(dis), 1e-300)
print('512 LHS rows, hourly 10 yr: max rel err gw %.3e (median %.1e), discharge max %.3e' % (e.max(), np.median(e), d.max()))
