"""Config 2's ill-conditioned rows under variations of the call: which part of it costs what."""
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
cls = engine.variant_classes(torch.from_numpy(params).to(dev), 86400.0).cpu().numpy()
rows = np.nonzero(cls == 3)[0]
p = torch.from_numpy(np.ascontiguousarray(params[rows])).to(dev)


def timed(label, T_use=T, gap=1, warm=365, **kw):
    f = forcing[:T_use]
    R = T_use // gap
    obs = np.abs(np.sin(np.arange(R))) + 1.0
    args = dict(obs=obs, gw_obs=bench.GW_OBS, extra=bench.EXTRA, want_discharge=True)
    args.update(kw)
    if args.get('obs', 1) is None:
        args.pop('gw_obs')
    prep = engine.prepare_ensemble(p, f, bench.AREA, 86400.0, warm, gap, **args)
    for _ in range(3):
        prep.launch()
    torch.cuda.synchronize()
    ts = []
    for _ in range(8):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        prep.launch()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    n = T_use + warm
    print('%-34s %.3f ms = %.1f ns per step  %s' % (label, min(ts), min(ts) * 1e6 / n, prep.describe()))


timed('as the bench calls it')
timed('no objective functions', obs=None)
timed('no discharge matrix', want_discharge=False)
timed('neither', obs=None, want_discharge=False)
timed('no educated guess (extra=None)', extra=None)
timed('gap 7 (3647 steps)', T_use=3647, gap=7, warm=364)
timed('gap 7, neither', T_use=3647, gap=7, warm=364, obs=None, want_discharge=False)
timed('no warm-up', warm=0)
timed('literal mode', math_mode='literal')
