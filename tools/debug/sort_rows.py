import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
import bench
from smartpy_amd import engine
from smartpy_amd.sampling import latin_hypercube
from smartpy_amd.parameters import Parameters
dev = torch.device('cuda:0')
f = torch.as_tensor(bench.synthetic_forcing(0, True)[0], device=dev)
T, W = f.shape[0], 8760
obs = torch.rand(T // 24, dtype=torch.float64, device=dev) + 0.5
for n in (100000, 125000, 1000000):
    p = torch.as_tensor(latin_hypercube(n, Parameters().ranges, seed=2718), device=dev)
    variants = {'as drawn': p, 'sorted by T': p[torch.argsort(p[:, 0])].contiguous(),
                'sorted by Z': p[torch.argsort(p[:, 5])].contiguous(), 'sorted by T*Z': p[torch.argsort(p[:, 0] * 1000 + p[:, 5])].contiguous()}
    for rep in range(2):
        for name, q in variants.items():
            prep = engine.prepare_ensemble(q, f, bench.AREA, 3600.0, W, 24, extra=bench.EXTRA, obs=obs, gw_obs=0.12667, want_discharge=False)
            prep.launch(); torch.cuda.synchronize()
            ts = []
            for _ in range(4):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); prep.launch(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
            print('N=%d %-14s %.3f ms' % (n, name, min(ts)), flush=True)
            del prep
