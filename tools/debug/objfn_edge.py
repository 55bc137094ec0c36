import sys
sys.path.insert(0, '.')
import numpy as np, torch, warnings
import bench
from smartpy_amd import engine
from oracle import smart_oracle as so, objfn_oracle, lhs_oracle
warnings.simplefilter('ignore')
f = bench.synthetic_forcing(0, True)[0][:24 * 100]
T, W = f.shape[0], 24 * 10
p = lhs_oracle.lhs_params(70, seed=3)
R = T // 24
cases = {'all NaN': np.full(R, np.nan), 'constant': np.full(R, 2.0), 'single value': np.where(np.arange(R) == 17, 1.5, np.nan),
         'two values': np.where(np.arange(R) % 50 == 7, 1.5 + np.arange(R) / 100.0, np.nan), 'zeros': np.zeros(R)}
dis, gw, _ = so.run_batch(bench.AREA, 3600.0, T, W, f[:, 0].copy(), f[:, 1].copy(), p, bench.EXTRA, so.REPORT_SUMMARY, 24)
np.set_printoptions(precision=4, linewidth=200)
for name, obs in cases.items():
    out = engine.run_ensemble(p, f, bench.AREA, 3600.0, W, 24, extra=bench.EXTRA, obs=obs, gw_obs=0.12667)
    got = out.objfn.cpu().numpy()
    try:
        want = objfn_oracle.objective_matrix(dis, obs, gw, 0.12667)
    except Exception as e:
        want = [repr(e)]
    two = engine.objective_functions(out.discharge_report_major, obs, out.gw, 0.12667).cpu().numpy()
    print(name); print('  fused ', got[0]); print('  matrix', two[0]); print('  numpy ', want[0])
