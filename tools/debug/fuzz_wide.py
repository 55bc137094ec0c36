"""More seeds of tests/test_gpu_parity.py::run_wide_cases (parameters far outside the default ranges, every arithmetic
class, final rows) and ::run_interval_cases (interval engine and step loop, slices and exits at random; round 4: also
under report='raw' and under a report every step) than the test suite runs.  usage: python tools/debug/fuzz_wide.py <first seed> <n seeds> [cases]"""
import os
import sys
import traceback
sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
from smartpy_amd import engine
import test_gpu_parity as t
first, count = int(sys.argv[1]), int(sys.argv[2])
cases = int(sys.argv[3]) if len(sys.argv) > 3 else 10
bad = 0
for seed in range(first, first + count):
    try:
        t.run_wide_cases(engine, seed, cases)
        t.run_batch_cases(engine, seed, cases)
        t.run_batch_cases(engine, seed + 1000003, cases, stress_initial=True)   # round 4: starts far above capacity, large H
        setenv = lambda k, v: os.environ.__setitem__(k, str(v)) if v else os.environ.pop(k, None)     # noqa: E731
        t.run_interval_cases(engine, setenv, seed, cases)
        t.run_interval_cases(engine, setenv, seed, cases, mode='raw')         # round 4: the raw / every-step kernels
        t.run_interval_cases(engine, setenv, seed, max(cases // 2, 1), mode='every')
    except (AssertionError, Exception):
        bad += 1
        tb = traceback.format_exc().splitlines()
        where = [ln.strip() for ln in tb if ln.strip().startswith('assert')]
        print('seed', seed, 'FAILED:', tb[-1][:300], '|', where[-1][:160] if where else '', flush=True)
print('%d seeds x %d cases: %d failed' % (count, cases, bad))
