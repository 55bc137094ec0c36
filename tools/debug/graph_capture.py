"""Does an ensemble launch (incl. the time-sliced one with its stream-ordered scratch) capture into a HIP graph?"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import torch
import bench
from smartpy_amd import engine, sampling
from smartpy_amd.parameters import Parameters
dev = torch.device('cuda:0')
forcing = torch.as_tensor(bench.synthetic_forcing(0, True)[0], device=dev)
T, W = forcing.shape[0], 8760
extra = {'aar': 1200, 'r-o_ratio': 0.45, 'r-o_split': (0.10, 0.15, 0.15, 0.30, 0.30)}
obs = torch.rand(T // 24, dtype=torch.float64, device=dev) + 0.5
area = torch.tensor([175.46e6], dtype=torch.float64, device=dev)
extra = torch.tensor([[1200, 0.45, 0.10, 0.15, 0.15, 0.30, 0.30]], dtype=torch.float64, device=dev)
gwo = torch.tensor([0.2], dtype=torch.float64, device=dev)
for n in (2000, 100000):
    params = torch.as_tensor(sampling.latin_hypercube(n, Parameters().ranges, seed=3), device=dev)
    ref = engine.run_ensemble(params, forcing, area, 3600.0, W, 24, extra=extra, obs=obs, gw_obs=gwo, want_discharge=False)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        engine.run_ensemble(params, forcing, area, 3600.0, W, 24, extra=extra, obs=obs, gw_obs=gwo, want_discharge=False)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            out = engine.run_ensemble(params, forcing, area, 3600.0, W, 24, extra=extra, obs=obs, gw_obs=gwo, want_discharge=False)
    out.objfn.zero_()
    g.replay(); torch.cuda.synchronize()
    same = torch.equal(out.objfn, ref.objfn) and torch.equal(out.gw, ref.gw)
    t0 = time.perf_counter()
    for _ in range(10):
        g.replay()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(10):
        engine.run_ensemble(params, forcing, area, 3600.0, W, 24, extra=extra, obs=obs, gw_obs=gwo, want_discharge=False)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('N=%d: graph replay identical: %s; replay %.3f ms, direct call %.3f ms per launch' % (n, same, (t1 - t0) * 100, (t2 - t1) * 100))
