"""smartcpp.allsteps (the reference's hook: one sample per call) on ten years of hourly forcing that varies inside the day:
literal / SMART_ALLSTEPS_MATH=fast, with and without the pair blocks."""
import os, sys, time
sys.path.insert(0, '.')
import numpy as np
import bench
from smartpy_amd import smartcpp
base = bench.synthetic_forcing(0, True)[0]
vary = bench.hourly_varying_forcing(base)
T = base.shape[0]
rain, peva = np.ascontiguousarray(vary[:, 0]), np.ascontiguousarray(vary[:, 1])
from smartpy_amd.parameters import Parameters
params = np.array([0.5 * (lo + hi) for lo, hi in Parameters().ranges.values()])       # the middle of the sampling ranges
initial = np.zeros(19)
for mode in ('fast', 'literal'):
    os.environ['SMART_ALLSTEPS_MATH'] = mode
    for pairs in ('1', '0'):
        os.environ['SMART_PAIR_BLOCKS'] = pairs
        ts = []
        for _ in range(4):
            t0 = time.perf_counter()
            out = smartcpp.allsteps(bench.AREA, 3600.0, T, rain, peva, params, initial, 1, 24)   # REPORT_SUMMARY, daily means
            ts.append(time.perf_counter() - t0)
        print('%-8s pair blocks %s: %.2f ms per call (best of 4), discharge[100] %.17g' % (mode, pairs, min(ts) * 1e3, np.asarray(out[0])[100]))
