"""Time-sliced launch against the plain one: bit-identical results, and the launch time for a few slice counts."""
import os, sys, time
sys.path.insert(0, '.')
import numpy as np
import torch
import bench
from smartpy_amd import engine, sampling
from smartpy_amd.parameters import Parameters

dev = torch.device('cuda:0')
forcing = bench.synthetic_forcing(0, True)[0]
T, W = forcing.shape[0], 8760
extra = {'aar': 1200, 'r-o_ratio': 0.45, 'r-o_split': (0.10, 0.15, 0.15, 0.30, 0.30)}
ft = torch.as_tensor(forcing, device=dev)
obs = torch.rand(T // 24, dtype=torch.float64, device=dev) + 0.5
obs[::9] = float('nan')
for n in [int(a) for a in sys.argv[1:]] or [100000]:
    params = torch.as_tensor(sampling.latin_hypercube(n, Parameters().ranges, seed=2718), device=dev)
    ref = None
    for k in (0, -1, 4, 8, 16, 32, 64):
        if k < 0:
            os.environ.pop('SMART_TIME_SLICES', None)
        else:
            os.environ['SMART_TIME_SLICES'] = str(k)
        ts = []
        for rep in range(4):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            out = engine.run_ensemble(params, ft, 175.46e6, 3600.0, W, 24, extra=extra, obs=obs, gw_obs=0.12667)
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        res = (out.discharge.clone(), out.gw.clone(), out.objfn.clone())
        if ref is None:
            ref = res
        same = all(torch.equal(a.view(torch.int64), b.view(torch.int64)) for a, b in zip(ref, res))
        print('N=%d  SMART_TIME_SLICES=%-8s %8.3f ms   identical to unsliced: %s' % (n, 'default' if k < 0 else k, min(ts[1:]) * 1e3, same), flush=True)
