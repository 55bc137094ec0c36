"""The flat-forcing workload (1e5 samples x hourly 10 yr + 1 yr warm-up, forcing that varies inside the day) under other
report gaps: launch times with and without the pair blocks (SMART_PAIR_BLOCKS=0).  usage: gap_sweep.py [gaps ...]"""
import os
import sys
sys.path.insert(0, '.')
import torch
import bench
from smartpy_amd import engine
from smartpy_amd.parameters import Parameters
from smartpy_amd.sampling import latin_hypercube

gaps = [int(x) for x in sys.argv[1:]] or [4, 8, 12, 24, 6]
dev = torch.device('cuda:0')
base = bench.synthetic_forcing(0, True)[0]
vary = bench.hourly_varying_forcing(base)
T, W = base.shape[0], 8760
params = torch.as_tensor(latin_hypercube(100000, Parameters().ranges, seed=2718), device=dev)
for gap in gaps:
    obs = torch.rand(T // gap, dtype=torch.float64, device=dev) + 0.5
    row = []
    for pairs in ('1', '0'):
        os.environ['SMART_PAIR_BLOCKS'] = pairs
        p = engine.prepare_ensemble(params, vary, bench.AREA, 3600.0, W, gap, extra=bench.EXTRA, obs=obs, gw_obs=0.12667,
                                    want_discharge=False)
        ts = []
        for _ in range(6):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); p.launch(); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        p.verify()
        row.append(min(ts))
        kernel = p.describe().split('[')[0]
        del p
    print('gap %3d  %-22s pair blocks %.3f ms   step by step %.3f ms' % (gap, kernel, row[0], row[1]))
