#!/usr/bin/env python3
"""How often the rain excess of a wet step fits into the TOP soil layer for every lane of a wavefront -- the path
frequencies behind SMART_WET_MODES / SMART_RAIN_FILL_EXIT (smart_fast_arms.h), which tools/isa_model.py weighs the
instruction counts with.  A numpy walk of the soil layers alone (structure.py:339-419 as restated in SURVEY.md
App. A: filling, the three leak passes, the evaporation cascade) over the bench workloads, for the first 2,048 rows
(32 wavefronts of 64 consecutive rows) of the 1e5-row LHS matrix the bench draws.  Depends on the workload only, not on
the kernels:

    python tools/fill_paths.py profiles/r03_fill_paths.json        (about three minutes on one core per workload)
"""
import json
import os
import sys
from concurrent.futures import ProcessPoolExecutor

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
N_ROWS = 2048


def walk(which):
    import bench
    from smartpy_amd.parameters import Parameters
    from smartpy_amd.sampling import latin_hypercube
    base = bench.synthetic_forcing(0, True)[0]
    f, gap = {'headline': (base, 24), 'runs_of_6': (bench.six_hourly_forcing(base), 6),
              'flat_forcing': (bench.hourly_varying_forcing(base), 1)}[which]
    f = np.concatenate([f[:bench.WARM_DAYS * 24], f])
    P = latin_hypercube(100000, Parameters().ranges, seed=2718)[:N_ROWS]
    T, C, H, S, Z = P[:, 0], P[:, 1], P[:, 2], P[:, 4], P[:, 5]
    N, W = N_ROWS, N_ROWS // 64
    z = Z / 6
    lv = np.tile((Z / 12)[:, None], (1, 6)).copy()
    n_wet = n_absorbed = n_prefix = n_switch = n_iv = 0
    mode = np.ones(W, bool)
    for t in range(len(f)):
        rain, pe = f[t]
        ex = rain * T - pe
        wet = ex >= 0
        tot = lv.sum(1)
        x = np.where(wet, ex - H * (tot / Z) * ex, 0.0)
        lw = lv.copy()
        past_top = None
        for i in range(6):                                   # filling, top down (:367-374)
            put = np.minimum(x, z - lw[:, i])
            lw[:, i] += put
            x = x - put
            if i == 0:
                past_top = (x > 0) & wet
        s1 = S * (tot / Z)
        for i in range(6):
            lw[:, i] -= lw[:, i] * s1 ** (i + 1)             # (:381-385)
        for i in range(6):
            lw[:, i] -= lw[:, i] * s1 / (i + 1)              # (:387-392)
        for i in range(6):
            lw[:, i] -= lw[:, i] * s1 ** (6 - i)             # (:394-399)
        ld = lv.copy()
        d = np.where(wet, 0.0, -ex)
        for i in range(6):                                   # evaporation cascade (:409-419)
            take = np.minimum(ld[:, i], d)
            ld[:, i] -= take
            d = np.where(ld[:, i] > 0, 0.0, C * (d - take))
        lv = np.where(wet[:, None], lw, ld)
        if rain > 0:
            any_wet = wet.reshape(W, 64).any(1)
            over = past_top.reshape(W, 64).any(1)
            if t % gap == 0:
                mode[:] = True
                n_iv += int(any_wet.sum())
            n_wet += int(any_wet.sum())
            n_absorbed += int((any_wet & ~over).sum())
            n_switch += int((mode & any_wet & over).sum())
            mode &= ~over
            n_prefix += int((mode & any_wet).sum())
    return which, {'rows': N_ROWS, 'steps': len(f), 'run_length': gap,
                   'rainy_wave_steps_with_a_wet_lane': n_wet, 'per_wave_step': n_wet / (W * len(f)),
                   'absorbed_by_the_top_layer': n_absorbed / n_wet,
                   'absorbed_prefix_of_the_run': n_prefix / n_wet, 'mode_switches_per_wet_run': n_switch / max(n_iv, 1)}


if __name__ == '__main__':
    with ProcessPoolExecutor(3) as pool:
        res = dict(pool.map(walk, ['headline', 'runs_of_6', 'flat_forcing']))
    print(json.dumps(res, indent=1))
    if len(sys.argv) > 1:
        with open(sys.argv[1], 'w') as fh:
            json.dump(res, fh, indent=1)
