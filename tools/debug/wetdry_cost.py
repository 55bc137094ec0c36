"""Cycles per wave-step of the fast kernel on forcing that is wet everywhere / dry everywhere / the bench's mix,
at 1 and 2 waves per SIMD (N = 65536 / 131072).  Objectives fused, discharge stored, hourly steps, gap 24."""
import os, sys, time
sys.path.insert(0, '.')
import numpy as np
import torch
import bench
from smartpy_amd import engine, sampling

dev = torch.device('cuda:0')
base = bench.synthetic_forcing(0, True)[0]
T = base.shape[0]
W = 8760
days = T // 24
cases = {}
cases['bench mix'] = base
wet = np.empty_like(base); wet[:, 0] = 4.0 / 24; wet[:, 1] = 1.0 / 24          # 3 mm/d excess at T=1: deep fill always
cases['all wet (4 mm/d rain, 1 mm/d pe)'] = wet
lw = np.empty_like(base); lw[:, 0] = 1.3 / 24; lw[:, 1] = 1.0 / 24             # small excess
cases['all wet (1.3 mm/d rain, 1 mm/d pe)'] = lw
dry = np.empty_like(base); dry[:, 0] = 0.0; dry[:, 1] = 1.0 / 24
cases['all dry (pe 1 mm/d)'] = dry
alt = np.empty_like(base)
d = np.arange(T) // 24
alt[:, 0] = np.where(d % 2 == 0, 4.0 / 24, 0.0); alt[:, 1] = 1.0 / 24
cases['alternating wet/dry days'] = alt
extra = {'aar': 1200, 'r-o_ratio': 0.45, 'r-o_split': (0.10, 0.15, 0.15, 0.30, 0.30)}
for n in (65536, 131072):
    np.random.seed(2718)
    params = torch.as_tensor(sampling.latin_hypercube(n, bench.ranges if hasattr(bench, "ranges") else __import__("smartpy_amd.parameters", fromlist=["x"]).Parameters().ranges, seed=2718), device=dev)
    obs = None if os.environ.get('NO_OBS') else torch.rand(days, dtype=torch.float64, device=dev) + 0.5
    for name, f in cases.items():
        ft = torch.as_tensor(f, device=dev)
        ts = []
        for rep in range(4):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            engine.run_ensemble(params, ft, 175.46e6, 3600.0, W, 24, extra=extra, obs=obs, gw_obs=0.12667 if obs is not None else None, want_discharge=not os.environ.get('NO_DIS'))
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        t = min(ts[1:])
        waves_per_simd = n / 65536
        cyc = t * 2.4e9 / (T + W) / waves_per_simd
        print('N=%6d  %-36s %7.3f ms   %6.1f cycles per wave-step (at 2.4 GHz, per wave)' % (n, name, t * 1e3, cyc), flush=True)
