"""The raw_gap24 / gap1 legs of bench.py on their own (for rocprofv3 passes of smart_fast_intervals_raw,
smart_fast_steps_raw and smart_fast_steps_every).

    python tools/debug/reports_only.py raw|raw_flat|every [n_samples] [launches]
"""
import sys
sys.path.insert(0, '.')

import numpy as np
import torch
import bench
from smartpy_amd import engine
from smartpy_amd.parameters import Parameters
from smartpy_amd.sampling import latin_hypercube
mode = sys.argv[1] if len(sys.argv) > 1 else 'raw'
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dev = torch.device('cuda:0')
f = bench.synthetic_forcing(0, True)[0]
if mode == 'raw_flat':
    f = bench.hourly_varying_forcing(f)
T, W = f.shape[0], 8760
gap = 1 if mode == 'every' else 24
rng = np.random.default_rng(1)
obs = np.abs(rng.normal(2.0, 1.0, T // gap))
obs[rng.random(T // gap) < 0.12] = np.nan
params = torch.as_tensor(latin_hypercube(n, Parameters().ranges, seed=2718), device=dev)
p = engine.prepare_ensemble(params, f, bench.AREA, 3600.0, W, gap, extra=bench.EXTRA, obs=obs, gw_obs=0.12667,
                            report='summary' if mode == 'every' else 'raw', want_discharge=False)
print(p.describe())
for _ in range(reps):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); p.launch(); e1.record(); torch.cuda.synchronize()
    print('%.3f ms' % e0.elapsed_time(e1))
p.verify()
