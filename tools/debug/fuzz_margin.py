"""The margins (excess(): 1.0 = the tolerance) of given fuzz seeds of tests/test_gpu_parity.py's randomized families, with the
gate lifted to the tolerance itself.  usage: python tools/debug/fuzz_margin.py <seed> [<seed> ...]   (every family of
tools/debug/fuzz_wide.py, ten cases each; prints the largest margins per seed)"""
import os, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from smartpy_amd import engine
import test_gpu_parity as t
t.EXCESS_GATE = t.FUZZ_GATE = 1.0        # (the randomized families read FUZZ_GATE: round 5)
setenv = lambda k, v: os.environ.__setitem__(k, str(v)) if v else os.environ.pop(k, None)     # noqa: E731
for seed in (int(a) for a in sys.argv[1:]):
    t.MARGINS.clear()
    t.run_wide_cases(engine, seed, 10)
    t.run_batch_cases(engine, seed, 10)
    t.run_batch_cases(engine, seed + 1000003, 10, stress_initial=True)
    t.run_interval_cases(engine, setenv, seed, 10)
    t.run_interval_cases(engine, setenv, seed, 10, mode='raw')
    t.run_interval_cases(engine, setenv, seed, 5, mode='every')
    print(seed, ['%s %.3f' % kv for kv in sorted(t.MARGINS.items(), key=lambda kv: -kv[1])[:3]])
