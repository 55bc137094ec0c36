"""Latency of one SMART.simulate() (N = 1) on the shipped example, hourly 10 yr: the per-sample protocol of the reference."""
import os, shutil, sys, time, tempfile
sys.path.insert(0, '.')
from datetime import datetime, timedelta
import numpy as np
import torch
import smartpy_amd
root = os.path.join(tempfile.mkdtemp(), 'data')
shutil.copytree('tests/golden/data/in', os.path.join(root, 'in'))
for delta in (timedelta(hours=1), timedelta(days=1)):
    sm = smartpy_amd.SMART('Catchment', 175.46e6, datetime(2007, 1, 1, 9), datetime(2016, 12, 31, 9), delta, timedelta(days=1), 365,
                           'csv', 'csv', root, gauged_area_m2=175.97e6)
    sm.extra = {'aar': 1200, 'r-o_ratio': 0.45, 'r-o_split': (0.10, 0.15, 0.15, 0.30, 0.30)}
    sm.parameters.set_parameters_with_file(''.join([sm.in_f, sm.catchment, '.parameters']))
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        d, g = sm.simulate(sm.parameters.values)
        torch.cuda.synchronize(); t1 = time.perf_counter()
    print('simulate(), simulation step %s: %.2f ms' % (delta, (t1 - t0) * 1e3))
    rows = np.array([[sm.parameters.values[n] for n in sm.parameters.names]])
    for mode in ('fast', 'literal'):
        for rep in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            ens = sm.simulate_ensemble(rows, math_mode=mode)
            torch.cuda.synchronize(); t1 = time.perf_counter()
        print('   simulate_ensemble(1 row, %s): %.2f ms' % (mode, (t1 - t0) * 1e3))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(20):
    sm.simulate(sm.parameters.values)
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(14)
