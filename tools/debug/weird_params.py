"""Fast mode against the literal arithmetic through the hook (one sample per call) for a parameter vector far outside every
sampling range, and for each of its values alone in an otherwise ordinary vector: which combination leaves the fast
arithmetic (round 4: D = 300 with a soil of half a millimetre -- now a row of the literal class, wave_class)."""
import os, sys
sys.path.insert(0, '.')
import numpy as np
import bench
from smartpy_amd import smartcpp
from smartpy_amd.parameters import Parameters
base = bench.synthetic_forcing(0, True)[0]
vary = bench.hourly_varying_forcing(base)[:24 * 400]
T = vary.shape[0]
rain, peva = np.ascontiguousarray(vary[:, 0]), np.ascontiguousarray(vary[:, 1])
mid = np.array([0.5 * (lo + hi) for lo, hi in Parameters().ranges.values()])
weird = np.array([1.0, 0.2, 0.2, 300.0, 0.3, 0.5, 2000.0, 200.0, 20000.0, 20.0])
names = list(Parameters().ranges)
initial = np.zeros(19)
def both(p):
    out = []
    for mode in ('fast', 'literal'):
        os.environ['SMART_ALLSTEPS_MATH'] = mode
        out.append(np.asarray(smartcpp.allsteps(bench.AREA, 3600.0, T, rain, peva, p, initial, 1, 24)[0]))
    a, b = out
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-30)))
print('middle of the ranges: fast against literal %.3g' % both(mid))
print('the odd vector      : %.3g' % both(weird))
for i in range(10):
    p = mid.copy(); p[i] = weird[i]
    print('  only %-2s = %-8g: %.3g' % (names[i], weird[i], both(p)))
