"""Launch time, plain against time-sliced (forced 16 slices), over the sample count."""
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
for n in [int(a) for a in sys.argv[1:]]:
    params = torch.as_tensor(sampling.latin_hypercube(n, Parameters().ranges, seed=11), device=dev)
    row = []
    for k in ('0', '16', None):
        if k is None:
            os.environ.pop('SMART_TIME_SLICES', None)
        else:
            os.environ['SMART_TIME_SLICES'] = k
        ts = []
        for rep in range(4):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            engine.run_ensemble(params, ft, 175.46e6, 3600.0, W, 24, extra=extra, obs=obs, gw_obs=0.12667, want_discharge=False)
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        row.append(min(ts[1:]) * 1e3)
    print('N=%7d  blocks/SIMD %.2f   plain %7.3f ms   16 slices %7.3f ms   default %7.3f ms   -> %.3g steps/s' % (n, (n + 63) // 64 / 1024, row[0], row[1], row[2], n * (T + W) / row[2] * 1e3), flush=True)
