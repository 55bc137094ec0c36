"""The ill-conditioned rows (dt / RK > 2) of the fast mode, whose divisions by per-sample constants go through cached
reciprocals and Markstein's correction step (smart_literal_model.h: LiteralModelT<true>), against the literal kernel
(true IEEE divisions): every bit of discharge, groundwater ratio and final row, over rows drawn to stress the division:
daily and 6-hourly steps, RK down to 0.05 h, routing constants near the step length, reservoirs that drain towards
the subnormal range, layers that run empty, areas from 1e4 to 1e10 m2."""
import sys
sys.path.insert(0, '.')
import numpy as np
import torch
from smartpy_amd import engine
from smartpy_amd.parameters import Parameters
from smartpy_amd.sampling import latin_hypercube

rng = np.random.default_rng(2024)
total = bad = 0
for case in range(24):
    dt = float(rng.choice([86400.0, 21600.0, 3600.0]))
    days = int(rng.integers(300, 2500))
    T = int(days * 86400 / dt)
    gap = int(rng.choice([1, 2, 4])) if dt < 86400 else 1
    T -= T % gap
    wet = rng.random(T) < rng.uniform(0.02, 0.8)
    rain = wet * rng.gamma(0.7, 4.57, T) * dt / 86400.0 * rng.choice([1.0, 1e-3, 50.0])
    peva = np.maximum(0.0, rng.normal(1.5, 1.0, T)) * dt / 86400.0
    if case % 4 == 3:
        rain[T // 3:] = 0.0                 # a drought: reservoirs drain for years
    n = int(rng.integers(64, 700))
    p = latin_hypercube(n, Parameters().ranges, seed=case)
    hours = dt / 3600.0
    p[:, 9] = rng.uniform(0.02, 0.49, n) * hours                    # RK: dt / RK in (2, 50)
    p[:, 6] = np.where(rng.random(n) < 0.5, rng.uniform(0.3, 3.0, n) * hours, p[:, 6])   # SK around the step
    p[:, 8] = np.where(rng.random(n) < 0.2, rng.uniform(1.01, 1.2, n) * hours, p[:, 8])  # GK just above it
    area = float(np.exp(rng.uniform(np.log(1e4), np.log(1e10))))
    extra = {'aar': 1200, 'r-o_ratio': 0.45, 'r-o_split': (0.10, 0.15, 0.15, 0.30, 0.30)} if case % 3 else None
    f = np.stack([rain, peva], axis=1)
    W = (T // 4) - (T // 4) % gap
    fast = engine.run_ensemble(p, f, area, dt, W, gap, extra=extra, want_final=True)
    lit = engine.run_ensemble(p, f, area, dt, W, gap, extra=extra, want_final=True, math_mode='literal')
    assert 'smart_fast_illcond' in fast._prepared.describe(), fast._prepared.describe()
    for name in ('discharge', 'gw', 'final_vars'):
        a, b = getattr(fast, name).cpu().numpy(), getattr(lit, name).cpu().numpy()
        same = a.view(np.int64) == b.view(np.int64)
        total += same.size
        bad += int((~same).sum())
        if not same.all():
            i = np.argwhere(~same)[0]
            print('case %d %s: %d of %d values differ, first at %s: %r vs %r' % (case, name, (~same).sum(), same.size,
                                                                                 tuple(i), a[tuple(i)], b[tuple(i)]))
    small = float(np.min(np.where(lit.final_vars.cpu().numpy()[:, 7:] > 0, lit.final_vars.cpu().numpy()[:, 7:], np.inf)))
    print('case %2d: dt %6.0f gap %d T %6d n %3d area %.1e  smallest positive final state %.3e' % (case, dt, gap, T, n, area, small))
print('%d values compared, %d differ' % (total, bad))
