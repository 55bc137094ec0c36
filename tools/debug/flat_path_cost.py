"""Launch time of the flat step loop (forcing that varies inside the report interval) next to the interval engine."""
import os, sys, time
sys.path.insert(0, '.')
import numpy as np
import torch
import bench
from smartpy_amd import engine, sampling
from smartpy_amd.parameters import Parameters
dev = torch.device('cuda:0')
base = bench.synthetic_forcing(0, True)[0]
T, W = base.shape[0], 8760
rng = np.random.default_rng(3)
vary = base.copy()
# same daily totals, distributed unevenly over the hours of the day (rain in 6 random hours, PE on a daytime sine)
wts = rng.random((T // 24, 24)) * (rng.random((T // 24, 24)) < 0.25)
wts[wts.sum(1) == 0, 0] = 1.0
wts /= wts.sum(1, keepdims=True)
vary[:, 0] = (base[::24, 0][:, None] * 24 * wts).ravel()
day = np.maximum(0.0, np.sin(np.pi * (np.arange(24) - 5) / 14)); day /= day.sum()
vary[:, 1] = (base[::24, 1][:, None] * 24 * day[None, :]).ravel()
extra = {'aar': 1200, 'r-o_ratio': 0.45, 'r-o_split': (0.10, 0.15, 0.15, 0.30, 0.30)}
obs = torch.rand(T // 24, dtype=torch.float64, device=dev) + 0.5
for n in (100000, 1000000):
    params = torch.as_tensor(sampling.latin_hypercube(n, Parameters().ranges, seed=2718), device=dev)
    for name, f in (('daily values / 24 (interval engine)', base), ('hourly-varying (flat loop)', vary)):
        ft = torch.as_tensor(f, device=dev)
        for final in (False, True):
            ts = []
            for rep in range(3):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                engine.run_ensemble(params, ft, 175.46e6, 3600.0, W, 24, extra=extra, obs=obs, gw_obs=0.12667, want_discharge=False, want_final=final)
                torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
            print('N=%7d  %-38s final states %-5s %8.3f ms  %.3g steps/s' % (n, name, final, min(ts[1:]) * 1e3, n * (T + W) / min(ts[1:])), flush=True)
