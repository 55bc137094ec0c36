"""Repeat the time-sliced launch and compare every result with the unsliced one bit for bit (the output buffers are
poisoned before every launch: a row that was not written cannot pass for the previous launch's)."""
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
n, reps = int(sys.argv[1]), int(sys.argv[2])
params = torch.as_tensor(sampling.latin_hypercube(n, Parameters().ranges, seed=11), device=dev)
os.environ['SMART_TIME_SLICES'] = '0'
ref = engine.run_ensemble(params, ft, 175.46e6, 3600.0, W, 24, extra=extra, obs=obs, gw_obs=0.12667, want_discharge=False)
ref = (ref.gw.clone(), ref.objfn.clone())
t0 = time.perf_counter()
torch.cuda.synchronize()
t0 = time.perf_counter()
engine.run_ensemble(params, ft, 175.46e6, 3600.0, W, 24, extra=extra, obs=obs, gw_obs=0.12667, want_discharge=False)
torch.cuda.synchronize()
t_plain = time.perf_counter() - t0
os.environ.pop('SMART_TIME_SLICES')
bad = 0
t0 = time.perf_counter()
for i in range(reps):
    poison = torch.full((n * 12,), float('nan'), dtype=torch.float64, device=dev)      # what torch.empty hands out next
    del poison
    out = engine.run_ensemble(params, ft, 175.46e6, 3600.0, W, 24, extra=extra, obs=obs, gw_obs=0.12667, want_discharge=False)
    if not (torch.equal(out.gw.view(torch.int64), ref[0].view(torch.int64)) and torch.equal(out.objfn.view(torch.int64), ref[1].view(torch.int64))):
        bad += 1
torch.cuda.synchronize()
print('N=%d: plain %.2f ms; %d default launches, %.2f ms each incl. compare, %d differ' % (n, t_plain * 1e3, reps, (time.perf_counter() - t0) / reps * 1e3, bad))
