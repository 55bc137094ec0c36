"""Hourly reports of an hourly run (report gap 1: smart_fast_plain) and raw reports at gap 24, 1e5 samples: launch time
next to the headline's (summary, gap 24: the interval engine)."""
import sys
sys.path.insert(0, '.')
sys.path.insert(0, 'oracle')
import numpy as np, torch
import bench
from smartpy_amd import engine
import lhs_oracle
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
params = torch.from_numpy(lhs_oracle.lhs_params(N, seed=2718)).cuda()
f = torch.from_numpy(bench.synthetic_forcing(0, True)[0]).cuda()
T = f.shape[0]
def timeit(prep, n=4):
    prep.launch(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): prep.launch()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
rng = np.random.default_rng(1)
for name, gap, report in (('summary gap 24 (headline)', 24, 'summary'), ('raw gap 24', 24, 'raw'), ('summary gap 1 (hourly reports)', 1, 'summary')):
    R = T // gap
    obs = np.abs(rng.normal(2.0, 1.0, R)); obs[rng.random(R) < 0.12] = np.nan
    prep = engine.prepare_ensemble(params, f, bench.AREA, 3600.0, 8760, gap, report=report, extra=bench.EXTRA, obs=obs, gw_obs=0.12667, want_discharge=False)
    print('%-34s %8.3f ms  %s' % (name, timeit(prep), prep.describe()))
