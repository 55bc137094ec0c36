"""Outputs of the merged summary kernels (step loop, interval engine, run engine) on a fixed set of seeded set-ups, for
comparing two builds of the library bit for bit:

    SMART_AMD_LIB=.../libsmart_amd_oldsteps.so python tools/debug/steps_bits.py dump a.npz
    python tools/debug/steps_bits.py dump b.npz
    python tools/debug/steps_bits.py compare a.npz b.npz

`oldsteps` = the same sources with -DSMART_STEP_ARMS=0 (tools/build_variants.py): the compiled step_lazy() loop of
round 2.  The asm arms of round 3 perform the same operations in the same order, so every output must agree in every
bit: discharge, groundwater ratio, objective functions, final rows; whole and time-sliced; quick and not."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

import numpy as np


def cases():
    import bench
    from smartpy_amd.parameters import Parameters
    from smartpy_amd.sampling import latin_hypercube
    rng = np.random.default_rng(31337)
    ranges = Parameters().ranges
    base = bench.synthetic_forcing(0, True)[0]
    vary = bench.hourly_varying_forcing(base)
    out = []
    # the bench's own sub-daily forcing, two years, ragged sample count, whole / sliced, final row or not
    for n, days, slices, final in [(200, 400, '', False), (4096 + 37, 730, '5', False), (300, 500, '3', True),
                                   (129, 365, '', True)]:
        out.append(dict(name='bench_%d_%d_%s_%d' % (n, days, slices or 'w', final), forcing=vary[:days * 24].copy(),
                        params=latin_hypercube(n, ranges, seed=n), dt=3600.0, W=120 * 24, gap=24, slices=slices,
                        final=final, initial=None))
    # the interval engine and the run engine (forcing constant over the report interval / over runs of 6 and 3 steps):
    # their wet intervals are an asm loop as well (SMART_WET_ASM)
    six = bench.six_hourly_forcing(base)
    three = np.repeat(bench.hourly_varying_forcing(base)[::3], 3, axis=0)[:base.shape[0]]
    for tag, forcing in (('iv', base), ('runs6', six), ('runs3', three)):
        for n, days, slices, final, exits in [(257, 400, '', False, '0'), (1000, 600, '4', False, '0'),
                                              (300, 365, '', False, '1'), (200, 365, '3', True, '')]:
            out.append(dict(name='%s_%d_%d_%s_%d_%s' % (tag, n, days, slices or 'w', final, exits or 'd'),
                            forcing=forcing[:days * 24].copy(), params=latin_hypercube(n, ranges, seed=n + 1),
                            dt=3600.0, W=96 * 24, gap=24, slices=slices, final=final, initial=None, exits=exits))
    # random set-ups: gaps that are and are not multiples of the chunk, storms, droughts, calm steps, exact zeros
    for case in range(24):
        gap = int(rng.choice([2, 3, 4, 6, 8, 24, 48, 5]))
        n_rep = int(rng.integers(64, 160))
        dt = float(rng.choice([900.0, 3600.0, 10800.0]))
        scale = dt / 86400.0 * gap
        rain_iv = rng.gamma(0.4, 8.0, n_rep) * (rng.random(n_rep) < rng.uniform(0.2, 0.9)) * rng.choice([1.0, 15.0])
        peva_iv = np.maximum(0.0, rng.normal(1.5, 1.0, n_rep))
        w_r = rng.random((n_rep, gap)) * (rng.random((n_rep, gap)) < 0.3)
        w_r[w_r.sum(1) == 0, 0] = 1.0
        w_e = rng.random((n_rep, gap)) * (rng.random((n_rep, gap)) < 0.5)
        w_e[w_e.sum(1) == 0, -1] = 1.0
        rain = (rain_iv[:, None] * scale * w_r / w_r.sum(1, keepdims=True)).ravel()
        peva = (peva_iv[:, None] * scale * w_e / w_e.sum(1, keepdims=True)).ravel()
        n = int(rng.integers(1, 400))
        params = latin_hypercube(max(n, 2), ranges, seed=1000 + case)[:n]
        kind = case % 6
        initial = None
        if kind == 3:       # a layer above its capacity: no calm shortcut for the wave (zero_ok false)
            initial = np.zeros((n, 12))
            initial[:, 5:11] = (params[:, 5:6] / 6.0) * rng.uniform(0.2, 1.6, (n, 6)) / 1e3 * 175.46e6
            initial[:, :5] = rng.uniform(0, 1e5, (n, 5))
            initial[:, 11] = rng.uniform(0, 1e5, n)
        if kind == 4:       # a negative evaporation value: the forcing is not sane, every step takes the general arm
            peva[rng.integers(0, peva.size, 3)] *= -1.0
        if kind == 5:       # -0.0 rain
            rain[np.flatnonzero(rain == 0)[:5]] = -0.0
        W = int(rng.integers(0, n_rep // 2 + 1)) * gap if rng.random() < 0.7 else 0
        out.append(dict(name='rnd%02d_gap%d' % (case, gap), forcing=np.stack([rain, peva], axis=1), params=params,
                        dt=dt, W=W, gap=gap, slices=str(rng.choice(['', '0', '3', '9'])),
                        final=bool(rng.random() < 0.5), initial=initial))
    if os.environ.get('STEPS_BITS_QUICK'):     # the test suite's share: every kind of set-up, the lighter ones
        out = [c for c in out if c['params'].shape[0] <= 400 and c['forcing'].shape[0] <= 12000]
    return out


def dump(path):
    import torch
    import bench
    from smartpy_amd import engine
    res = {}
    for c in cases():
        for var, val in (('SMART_TIME_SLICES', c['slices']), ('SMART_EXITS', c.get('exits', ''))):
            if val:
                os.environ[var] = val
            else:
                os.environ.pop(var, None)
        n_rep = c['forcing'].shape[0] // c['gap']
        obs = np.random.default_rng(7).random(n_rep) * 3
        obs[::7] = np.nan
        r = engine.run_ensemble(c['params'], c['forcing'], bench.AREA, c['dt'], c['W'], c['gap'],
                                extra=None if c['initial'] is not None else bench.EXTRA, initial=c['initial'], obs=obs,
                                gw_obs=0.2, want_final=c['final'])
        torch.cuda.synchronize()
        kern = r._prepared.describe()
        if 'smart_fast_steps' not in kern and c['name'].startswith(('bench', 'rnd')):
            continue                            # (a lone ill-conditioned row: nothing of the step loop to compare)
        res[c['name'] + '/kernel'] = np.array(kern)
        res[c['name'] + '/discharge'] = r.discharge.cpu().numpy()
        res[c['name'] + '/gw'] = r.gw.cpu().numpy()
        res[c['name'] + '/objfn'] = r.objfn.cpu().numpy()
        if c['final']:
            res[c['name'] + '/final'] = r.final_vars.cpu().numpy()
        print(c['name'], kern, float(np.nanmax(res[c['name'] + '/discharge'])))
    np.savez_compressed(path, **res)


def compare(a, b):
    A, B = np.load(a), np.load(b)
    bad = 0
    for k in A.files:
        if k.endswith('/kernel'):
            continue
        x, y = A[k], B[k]
        same = x.shape == y.shape and np.array_equal(x.view(np.int64), y.view(np.int64))
        if not same:
            bad += 1
            d = np.abs(x - y)
            print('DIFFERS', k, 'max abs %.3e' % np.nanmax(d), 'n', int((x.view(np.int64) != y.view(np.int64)).sum()))
    print('%d arrays compared, %d differ' % (sum(1 for k in A.files if not k.endswith('/kernel')), bad))
    return bad


if __name__ == '__main__':
    if sys.argv[1] == 'dump':
        dump(sys.argv[2])
    else:
        sys.exit(1 if compare(sys.argv[2], sys.argv[3]) else 0)
