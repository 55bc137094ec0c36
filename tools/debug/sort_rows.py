"""Launch time against the order of the rows (no discharge stored): as drawn, by T, by Z, by T bins then Z, ...
The engine's own reordering (engine._variant_grouping, sort_rows) is switched off here so that the order is the tool's."""
import sys
sys.path.insert(0, '.')
import torch
import bench
from smartpy_amd import engine
from smartpy_amd.sampling import latin_hypercube
from smartpy_amd.parameters import Parameters
dev = torch.device('cuda:0')
f = torch.as_tensor(bench.synthetic_forcing(0, True)[0], device=dev)
T, W = f.shape[0], 8760
obs = torch.rand(T // 24, dtype=torch.float64, device=dev) + 0.5
engine_sort = engine._variant_grouping
engine._variant_grouping = lambda params, dt, sort_rows=False: engine_sort(params, dt, False)


def binned(p, bins, second):
    b = torch.clamp(((p[:, 0] - p[:, 0].min()) / (p[:, 0].max() - p[:, 0].min()) * bins).floor(), 0, bins - 1)
    v = second if isinstance(second, torch.Tensor) else p[:, second]
    return b * 2.0 + (v - v.min()) / (v.max() - v.min())


for n in [int(a) for a in sys.argv[1:]] or [100000, 125000, 1000000]:
    p = torch.as_tensor(latin_hypercube(n, Parameters().ranges, seed=2718), device=dev)
    sz = p[:, 4] * p[:, 5]
    span = lambda v: (v - v.min()) / (v.max() - v.min())        # noqa: E731
    keys = {'as drawn': None, 'T in 64 bins, then S*Z': binned(p, 64, sz), 'T in 32 bins, then S*Z': binned(p, 32, sz),
            'S*Z in 64 bins, then T': torch.clamp((span(sz) * 64).floor(), max=63) * 2.0 + span(p[:, 0]),
            'S*Z in 256 bins, then T': torch.clamp((span(sz) * 256).floor(), max=255) * 2.0 + span(p[:, 0]),
            'T in 64 bins, then S*Z*(1+C)': binned(p, 64, sz * (1 + p[:, 1])),
            'T in 64 bins, then S*sqrt(Z)': binned(p, 64, p[:, 4] * p[:, 5].sqrt()),
            'T in 64 bins, then S*Z^1.5': binned(p, 64, p[:, 4] * p[:, 5] ** 1.5),
            'T 16 x S*Z 16 bins, then H': (torch.clamp((span(p[:, 0]) * 16).floor(), max=15) * 64.0
                                           + torch.clamp((span(sz) * 16).floor(), max=15) * 2.0 + span(p[:, 2]))}
    for rep in range(2):
        for name, key in keys.items():
            q = p if key is None else p[torch.argsort(key)].contiguous()
            prep = engine.prepare_ensemble(q, f, bench.AREA, 3600.0, W, 24, extra=bench.EXTRA, obs=obs, gw_obs=0.12667,
                                           want_discharge=False)
            prep.launch()
            torch.cuda.synchronize()
            ts = []
            for _ in range(4):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                prep.launch()
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            print('N=%d %-22s %.3f ms' % (n, name, min(ts)), flush=True)
            del prep
