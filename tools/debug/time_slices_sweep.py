"""Launch time over the sample count: unsliced, forced 16 slices, and what the library chooses by itself.
Prepared launches timed with HIP events (min of 3 after a warm-up); the default column also says what was launched.

    python tools/debug/time_slices_sweep.py <n_samples> ...
"""
import sys
sys.path.insert(0, '.')
import torch
import bench
from smartpy_amd import engine, sampling
from smartpy_amd.parameters import Parameters
dev = torch.device('cuda:0')
forcing = bench.synthetic_forcing(0, True)[0]
T, W = forcing.shape[0], 8760
ft = torch.as_tensor(forcing, device=dev)
obs = torch.rand(T // 24, dtype=torch.float64, device=dev) + 0.5
print('# launch time vs samples per GPU (tools/debug/time_slices_sweep.py), one MI355X, hourly 10 yr + 1 yr warm-up')
print('# (96,432 steps), objectives fused, discharge not stored, rows ordered.  blocks/SIMD = ceil(N / 64) / 1024')


def timed(prep):
    prep.launch()
    best = 1e30
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        prep.launch()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    assert prep.status() == 0
    return best


for n in [int(float(a)) for a in sys.argv[1:]]:
    params = sampling.latin_hypercube_device(n, Parameters().ranges, seed=11, device=dev)
    row = []
    for k in (1, 16, 0):
        if k == 16 and n > 4000000:         # (18 GB of hand-over states at 1e8 samples: not a set-up anyone would force)
            row.append(float('nan'))
            continue
        prep = engine.prepare_ensemble(params, ft, bench.AREA, 3600.0, W, 24, extra=bench.EXTRA, obs=obs,
                                       gw_obs=bench.GW_OBS, want_discharge=False, time_slices=k)
        row.append(timed(prep))
        what = prep.describe()
        del prep
    print('N=%9d  blocks/SIMD %8.2f   unsliced %9.3f ms   16 slices %9.3f ms   default %9.3f ms   -> %.3g steps/s   %s'
          % (n, (n + 63) // 64 / 1024, row[0], row[1], row[2], n * (T + W) / row[2] * 1e3, what), flush=True)
    del params
    torch.cuda.empty_cache()
