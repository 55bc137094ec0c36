"""More seeds of tests/test_gpu_parity.py::run_wide_cases than the test suite runs (parameters far outside the default
ranges, every arithmetic class, final rows).  usage: python tools/debug/fuzz_wide.py <first seed> <n seeds> [cases]"""
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
    except AssertionError:
        bad += 1
        print('seed', seed, 'FAILED:', traceback.format_exc().splitlines()[-1][:300], flush=True)
print('%d seeds x %d cases: %d failed' % (count, cases, bad))
