"""The bench legs that walk the pair blocks / the every-step stream, at full size (1e5 samples x hourly 10 yr + 1 yr warm-up),
against the same library with SMART_PAIR_BLOCKS=0: every output bit for bit."""
import os
import sys
sys.path.insert(0, '.')
import numpy as np
import torch
import bench
from smartpy_amd import engine
from smartpy_amd.parameters import Parameters
from smartpy_amd.sampling import latin_hypercube

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
dev = torch.device('cuda:0')
base = bench.synthetic_forcing(0, True)[0]
vary = bench.hourly_varying_forcing(base)
T, W = base.shape[0], 8760
rng = np.random.default_rng(5)
params = torch.as_tensor(latin_hypercube(n, Parameters().ranges, seed=2718), device=dev)
for name, gap, report, store in (('flat_forcing', 24, 'summary', True), ('raw_gap24_flat', 24, 'raw', False),
                                 ('gap1', 1, 'summary', False)):
    obs = rng.random(T // gap) + 0.5
    obs[rng.random(T // gap) < 0.12] = np.nan
    outs = []
    for pairs in ('1', '0'):
        os.environ['SMART_PAIR_BLOCKS'] = pairs
        r = engine.run_ensemble(params, vary, bench.AREA, 3600.0, W, gap, extra=bench.EXTRA, obs=obs, gw_obs=0.12667,
                                report=report, want_discharge=store)
        outs.append([x.cpu().numpy().copy() for x in ((r.discharge,) if store else ()) + (r.gw, r.objfn)])
        kernel = r._prepared.describe()
        del r
        torch.cuda.empty_cache()
    same = all(np.array_equal(a.view(np.int64), b.view(np.int64)) for a, b in zip(*outs))
    print('%-15s %s: %d arrays, %d values, %s' % (name, kernel, len(outs[0]), sum(a.size for a in outs[0]),
                                                  'bit-identical' if same else 'DIFFER'))
