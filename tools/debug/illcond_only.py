"""Config 2's rows one class at a time and all together: what each of the side-by-side kernels of a daily ensemble takes
when it has the chip to itself (round 5: smart_fast_illcond with one sample per DPP row)."""
import sys
sys.path.insert(0, '.')
import numpy as np
import torch
import bench
from smartpy_amd import engine
from smartpy_amd.parameters import Parameters
from smartpy_amd.sampling import latin_hypercube

forcing, rng = bench.synthetic_forcing(0, hourly=False)
T = forcing.shape[0]
params = latin_hypercube(10000, Parameters().ranges, seed=2718)
dev = torch.device('cuda', 0)
p = torch.from_numpy(params).to(dev)
cls = engine.variant_classes(p, 86400.0).cpu().numpy()
obs = np.abs(np.sin(np.arange(T))) + 1.0


def timed(rows, label):
    prep = engine.prepare_ensemble(torch.from_numpy(np.ascontiguousarray(params[rows])).to(dev), forcing, bench.AREA,
                                   86400.0, 365, 1, obs=obs, gw_obs=bench.GW_OBS, extra=bench.EXTRA, want_discharge=True)
    for _ in range(3):
        prep.launch()
    torch.cuda.synchronize()
    ts = []
    for _ in range(10):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        prep.launch()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    prep.verify()
    print('%-22s %6d rows  %.3f ms (best of 10, median %.3f)  %s' % (label, len(rows), min(ts), float(np.median(ts)), prep.describe()))


for c in range(4):
    rows = np.nonzero(cls == c)[0]
    if len(rows):
        timed(rows, 'class %d alone' % c)
timed(np.arange(len(params)), 'all rows')
