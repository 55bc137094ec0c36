import os, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from smartpy_amd import engine
import test_gpu_parity as t
t.EXCESS_GATE = 1.0
setenv = lambda k, v: os.environ.__setitem__(k, str(v)) if v else os.environ.pop(k, None)
for seed in (30008, 30264):
    t.MARGINS.clear()
    t.run_interval_cases(engine, setenv, seed, 5, mode='every')
    print(seed, sorted(t.MARGINS.items(), key=lambda kv: -kv[1])[:4])
