"""A daily ensemble on its own (for the rocprofv3 passes of tools/gpu_profile_r06.sh): n LHS samples x ten years of daily
steps, a report every step, objective functions fused, the matrix stored up to 2e5 samples -- with the literal rows in
the form given (rows | lanes | auto).  `python tools/debug/daily_only.py <n> <reps> <form>`.  The kernel trace of the
run gives every class's kernel its own time (they run side by side on forked streams)."""
import sys
sys.path.insert(0, '.')

import numpy as np
import torch
import bench
from smartpy_amd import engine
from smartpy_amd.parameters import Parameters
from smartpy_amd.sampling import latin_hypercube
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
form = sys.argv[3] if len(sys.argv) > 3 else 'auto'
dev = torch.device('cuda:0')
forcing, _ = bench.synthetic_forcing(0, hourly=False)
T = forcing.shape[0]
obs = np.abs(np.sin(np.arange(T))) + 1.0
params = torch.as_tensor(latin_hypercube(n, Parameters().ranges, seed=2718 + n), device=dev)
p = engine.prepare_ensemble(params, forcing, bench.AREA, 86400.0, 365, 1, extra=bench.EXTRA, obs=obs, gw_obs=bench.GW_OBS,
                            want_discharge=n <= 200000, literal_form=form)
print(p.describe())
for _ in range(reps):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); p.enqueue(); e1.record(); torch.cuda.synchronize()
    print('%.3f ms' % e0.elapsed_time(e1))
p.verify()
