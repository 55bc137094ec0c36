"""Round 6: daily ensembles of 1e4 ... 1e6 samples with the literal rows (class 3: dt / RK > 2, 11.6 % of the default
LHS space at daily steps) in BOTH forms -- smart_fast_illcond (one sample per DPP row, sixteen wavefronts per block of
64 samples) and smart_fast_illcond_lanes (one per lane) -- whole ensemble and class 3 alone, and what the library picks
by itself.  `python tools/debug/daily_forms.py [sizes...] [--classes]`."""
import sys
sys.path.insert(0, '.')
import numpy as np
import torch
import bench
from smartpy_amd import engine
from smartpy_amd.parameters import Parameters
from smartpy_amd.sampling import latin_hypercube

args = [a for a in sys.argv[1:] if not a.startswith('--')]
sizes = [int(float(a)) for a in args] or [10000, 30000, 50000, 70000, 100000, 200000, 1000000]
forcing, rng = bench.synthetic_forcing(0, hourly=False)
T = forcing.shape[0]
dev = torch.device('cuda', 0)
obs = np.abs(np.sin(np.arange(T))) + 1.0
STORE_MAX = 200000      # a stored matrix above this is 1e6 x 3653 x 8 B = 29 GB: objectives only there


def timed(params, form, reps=8, store=True):
    prep = engine.prepare_ensemble(torch.from_numpy(np.ascontiguousarray(params)).to(dev), forcing, bench.AREA,
                                   86400.0, 365, 1, obs=obs, gw_obs=bench.GW_OBS, extra=bench.EXTRA,
                                   want_discharge=store, literal_form=form)
    for _ in range(2):
        prep.launch()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        prep.enqueue()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    prep.verify()
    res = prep.result()
    return min(ts), float(np.median(ts)), prep.describe(), res


print('daily 10 yr (%d steps + 365 warm-up), a report every step, objective functions fused; matrix stored up to %d '
      'samples' % (T, STORE_MAX))
for n in sizes:
    params = latin_hypercube(n, Parameters().ranges, seed=2718 + n)
    cls = engine.variant_classes(torch.from_numpy(params), 86400.0).numpy()
    n3 = int((cls == 3).sum())
    store = n <= STORE_MAX
    print('\nN = %d: class 0 / 1 / 2 / 3 = %s rows; class-3 blocks %d (x 16 = %d row-form wavefronts)' % (
        n, [int((cls == c).sum()) for c in range(4)], -(-n3 // 64), 16 * -(-n3 // 64)))
    keep = {}
    for form in ('rows', 'lanes', 'auto'):
        best, med, text, res = timed(params, form, store=store)
        keep[form] = (res.gw.cpu().numpy().copy(), res.objfn.cpu().numpy().copy())
        units = n * (T + 365)
        print('  whole ensemble  %-5s  %8.3f ms (median %8.3f)  %.3e sample-steps/s   %s' % (
            form, best, med, units / best * 1e3, text))
    same = np.array_equal(keep['rows'][0].view(np.int64), keep['lanes'][0].view(np.int64))
    a, b = keep['rows'][1][:, :7], keep['lanes'][1][:, :7]
    print('  groundwater ratios of the two forms: %s; objective functions (one-pass moments, the report\'s arithmetic): '
          '%.1e relative' % ('the same bits' if same else 'DIFFER',
                             float(np.max(np.abs(a - b) / np.maximum(np.maximum(np.abs(a), np.abs(b)), 1e-12)))))
    if '--classes' in sys.argv:
        for c in range(4):
            rows = np.nonzero(cls == c)[0]
            if not len(rows):
                continue
            for form in (('rows', 'lanes') if c == 3 else ('auto',)):
                best, med, text, _ = timed(params[rows], form, store=store)
                print('  class %d alone   %-5s  %8.3f ms (median %8.3f)  %6d rows  %s' % (c, form, best, med, len(rows), text))
