"""End-to-end timing of montecarlo.LHS(...).run() at 1e5 samples on the example catchment (hourly 10 yr)."""
import os, shutil, sys, time, tempfile
sys.path.insert(0, '.')
import numpy as np
import torch
t0 = time.perf_counter()
from smartpy_amd.montecarlo import LHS
root = os.path.join(tempfile.mkdtemp(), 'data')
shutil.copytree('tests/golden/data/in', os.path.join(root, 'in'))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
t1 = time.perf_counter()
np.random.seed(2718)
lhs = LHS('Catchment', root, 'csv', 'csv', n, save_sim=False)
lhs.model.extra = {'aar': 1200, 'r-o_ratio': 0.45, 'r-o_split': (0.10, 0.15, 0.15, 0.30, 0.30)}
t2 = time.perf_counter()
lhs.run()
torch.cuda.synchronize()
t3 = time.perf_counter()
lhs.run()
t4 = time.perf_counter()
print('import %.2f s | LHS() incl. file parsing + sampling %.2f s | first run() %.2f s | second run() %.2f s' % (t1 - t0, t2 - t1, t3 - t2, t4 - t3))
print('db size %.1f MB, best NSE %.4f' % (os.path.getsize(lhs.db_file) / 1e6, np.nanmax(lhs.obj_fns[:, 0])))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); lhs.run(); pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(8)
