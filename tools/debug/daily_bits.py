"""Daily ensembles under the library in SMART_AMD_LIB (or the tree's): timings of the whole launch and of every class
alone, and the outputs -- groundwater ratios, objective functions, the stored matrix up to 1e5 samples -- dumped for a
bit-for-bit comparison between two builds:   python tools/debug/daily_bits.py dump a.npz ; ... dump b.npz ; compare a.npz b.npz"""
import sys
sys.path.insert(0, '.')
import numpy as np


def dump(path):
    import torch
    import bench
    from smartpy_amd import engine
    from smartpy_amd.parameters import Parameters
    from smartpy_amd.sampling import latin_hypercube
    forcing, _ = bench.synthetic_forcing(0, hourly=False)
    T = forcing.shape[0]
    obs = np.abs(np.sin(np.arange(T))) + 1.0
    obs[::13] = np.nan
    dev = torch.device('cuda', 0)
    out = {}
    for n in (10000, 100000, 1000000):
        params = latin_hypercube(n, Parameters().ranges, seed=2718 + n)
        cls = engine.variant_classes(torch.from_numpy(params), 86400.0).numpy()
        for label, rows in [('all', np.arange(n))] + [('class%d' % c, np.nonzero(cls == c)[0]) for c in (0, 1, 3)]:
            p = engine.prepare_ensemble(torch.from_numpy(np.ascontiguousarray(params[rows])).to(dev), forcing, bench.AREA,
                                        86400.0, 365, 1, obs=obs, gw_obs=bench.GW_OBS, extra=bench.EXTRA,
                                        want_discharge=n <= 100000, want_final=False)
            for _ in range(2):
                p.launch()
            torch.cuda.synchronize()
            ts = []
            for _ in range(8):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); p.enqueue(); b.record(); torch.cuda.synchronize()
                ts.append(a.elapsed_time(b))
            res = p.verify()
            print('N = %7d %-7s %7d rows  %8.3f ms (median %8.3f)  %s' % (n, label, len(rows), min(ts), float(np.median(ts)),
                                                                        p.describe()), flush=True)
            if label == 'all':
                out['gw_%d' % n] = res.gw.cpu().numpy()
                out['objfn_%d' % n] = res.objfn.cpu().numpy()
                if res.discharge is not None:
                    out['dis_%d' % n] = res.discharge.cpu().numpy()[::7]
    np.savez(path, **out)


def compare(a, b):
    x, y = np.load(a), np.load(b)
    bad = 0
    for k in x.files:
        same = x[k].shape == y[k].shape and np.array_equal(x[k].view(np.int64), y[k].view(np.int64))
        bad += not same
        print('%-16s %s' % (k, 'the same bits' if same else 'DIFFER'))
    print('%d arrays compared, %d differ' % (len(x.files), bad))


if __name__ == '__main__':
    dump(sys.argv[2]) if sys.argv[1] == 'dump' else compare(sys.argv[2], sys.argv[3])
