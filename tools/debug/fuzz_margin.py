"""The margins (excess(): 1.0 = the tolerance) of given fuzz seeds of tests/test_gpu_parity.py::run_interval_cases, with the
gate lifted to the tolerance itself.  usage: python tools/debug/fuzz_margin.py <mode: summary|raw|every> <seed> [<seed> ...]"""
import os, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from smartpy_amd import engine
import test_gpu_parity as t
t.EXCESS_GATE = 1.0
setenv = lambda k, v: os.environ.__setitem__(k, str(v)) if v else os.environ.pop(k, None)     # noqa: E731
mode = sys.argv[1]
for seed in (int(a) for a in sys.argv[2:]):
    t.MARGINS.clear()
    if mode == 'summary':
        t.run_interval_cases(engine, setenv, seed, 10)
    else:
        t.run_interval_cases(engine, setenv, seed, 10 if mode == 'raw' else 5, mode=mode)
    print(seed, mode, sorted(t.MARGINS.items(), key=lambda kv: -kv[1])[:4])
