"""The flat-forcing leg of bench.py on its own (for rocprofv3 --pmc runs of the step-loop kernel)."""
import sys
sys.path.insert(0, '.')

import torch
import bench
from smartpy_amd import engine
from smartpy_amd.parameters import Parameters
from smartpy_amd.sampling import latin_hypercube
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device('cuda:0')
base = bench.synthetic_forcing(0, True)[0]
vary = bench.hourly_varying_forcing(base)
T, W = base.shape[0], 8760
obs = torch.rand(T // 24, dtype=torch.float64, device=dev) + 0.5
params = torch.as_tensor(latin_hypercube(n, Parameters().ranges, seed=2718), device=dev)
store = not (len(sys.argv) > 3 and sys.argv[3] == 'objectives')      # third argument 'objectives': no discharge matrix
p = engine.prepare_ensemble(params, vary, bench.AREA, 3600.0, W, 24, extra=bench.EXTRA, obs=obs, gw_obs=0.12667,
                            want_discharge=store)
print(p.describe())
for _ in range(reps):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); p.launch(); e1.record(); torch.cuda.synchronize()
    print('%.3f ms' % e0.elapsed_time(e1))
p.verify()
