"""Repeated launches, direct and through a captured HIP graph, outputs poisoned before every launch."""
import os, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
from conftest import load_golden
from oracle import lhs_oracle
from smartpy_amd import engine as eng
g = load_golden('forcing_example.npz')
dev = torch.device('cuda:0')
T, W = 24 * 200, 24 * 20
rain = np.repeat(g['rain_daily'] / 24, 24); peva = np.repeat(g['peva_daily'] / 24, 24)
f = torch.as_tensor(np.stack([rain[:T], peva[:T]], axis=1), device=dev)
obs = torch.as_tensor(g['flow_obs'][:T // 24], device=dev)
area = torch.tensor([float(g['area'])], dtype=torch.float64, device=dev)
extra = torch.tensor([list(g['extra'])], dtype=torch.float64, device=dev)
gwo = torch.tensor([0.12667], dtype=torch.float64, device=dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 70000
params = torch.as_tensor(lhs_oracle.lhs_params(n, seed=3), device=dev)
kw = dict(extra=extra, obs=obs, gw_obs=gwo, want_discharge=False)
os.environ['SMART_TIME_SLICES'] = '0'
ref = eng.run_ensemble(params, f, area, 3600.0, W, 24, **kw)
ref = (ref.objfn.clone(), ref.gw.clone())
for mode in sys.argv[2:] or ['0', 'default']:
    if mode == 'default':
        os.environ.pop('SMART_TIME_SLICES', None)
    else:
        os.environ['SMART_TIME_SLICES'] = mode
    bad = []
    for k in range(30):
        poison = torch.full((n * 12,), float('nan'), dtype=torch.float64, device=dev)
        del poison
        out = eng.run_ensemble(params, f, area, 3600.0, W, 24, **kw)
        torch.cuda.synchronize()
        d = ~((out.objfn == ref[0]).all(dim=-1)) | (out.gw != ref[1])
        bad.append(int(d.sum()))
        print('.', end='', flush=True)
    print(flush=True, *('slices=%s direct : rows differing per launch' % mode, bad))
    graph, side = torch.cuda.CUDAGraph(), torch.cuda.Stream()
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            out = eng.run_ensemble(params, f, area, 3600.0, W, 24, **kw)
    bad = []
    for k in range(30):
        out.objfn.fill_(float('nan')); out.gw.fill_(float('nan'))
        graph.replay(); torch.cuda.synchronize()
        d = ~((out.objfn == ref[0]).all(dim=-1)) | (out.gw != ref[1])
        bad.append(int(d.sum()))
    print(flush=True, *('slices=%s graph  : rows differing per replay' % mode, bad))
