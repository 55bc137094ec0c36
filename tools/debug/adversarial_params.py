"""Parameter rows far outside every sampling range, several values at a time (the wide family of the test suite varies
them within what a hydrologist could mean): fast mode against the literal arithmetic of the same library, row by row.
Rows the fast arithmetic is not made for must be of a class that takes the literal model (wave_class); this looks for
rows that are not and differ.  usage: adversarial_params.py [seed] [rows]"""
import sys
sys.path.insert(0, '.')
import numpy as np
import torch
from smartpy_amd import engine

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
rng = np.random.default_rng(seed)
T, gap = 24 * 200, 24
rain = rng.gamma(0.5, 3.0, T) * (rng.random(T) < 0.35)
peva = np.where(rain > 0, 0.0, rng.uniform(0.0, 0.3, T) * (rng.random(T) < 0.6))
forcing = np.stack([rain, peva], axis=1)
lo = np.array([0.9, 0.0, 0.0, 0.0, 0.0, 15.0, 1.0, 48.0, 1200.0, 1.0])
hi = np.array([1.1, 1.0, 0.3, 1.0, 0.013, 150.0, 240.0, 1440.0, 4800.0, 96.0])
p = lo + (hi - lo) * rng.random((n, 10))
# every row: one to four values thrown far out (either way), the others ordinary
for i in range(n):
    for j in rng.choice(10, size=int(rng.integers(1, 5)), replace=False):
        kind = rng.integers(0, 4)
        if kind == 0:
            p[i, j] = hi[j] * 10.0 ** rng.uniform(0.0, 3.0)
        elif kind == 1:
            p[i, j] = max(lo[j], 1e-3) * 10.0 ** -rng.uniform(0.0, 4.0)
        elif kind == 2:
            p[i, j] = -abs(p[i, j]) * 10.0 ** rng.uniform(-2.0, 1.0)
        else:
            p[i, j] = 0.0
fast = engine.run_ensemble(p, forcing, 2.0e8, 3600.0, 0, gap)
lit = engine.run_ensemble(p, forcing, 2.0e8, 3600.0, 0, gap, math_mode='literal')
a, b = fast.discharge.cpu().numpy(), lit.discharge.cpu().numpy()
cls = engine.variant_classes(torch.as_tensor(p), 3600.0).numpy()
both_nan = np.isnan(a) & np.isnan(b)
with np.errstate(invalid='ignore', divide='ignore'):
    top = np.nanmax(np.abs(b), axis=1, keepdims=True)
    err = np.where(both_nan, 0.0, np.abs(a - b) / (1e-9 * np.abs(b) + 1e-12 * np.where(np.isfinite(top), top, 1.0) + 1e-300))
err = np.where(np.isnan(err), np.inf, err)          # a NaN on one side only
worst = err.max(axis=1)
bad = np.flatnonzero(worst > 1.0)
c3 = cls == 3
print('  (class 3 = the literal model in both launches: %d rows differ by more than the sums\' order: 1e-12 relative)' % int(((np.where(both_nan, 0.0, np.abs(a - b) / (1e-12 * np.abs(b) + 1e-300)) > 1.0).any(axis=1) & c3).sum()))
with np.errstate(invalid='ignore', divide='ignore'):
    d3 = np.where(both_nan | (a == b), 0.0, np.abs(a - b) / (1e-12 * np.abs(b) + 1e-300))
for i in np.flatnonzero((np.where(np.isnan(d3), np.inf, d3) > 1.0).any(axis=1) & c3)[:6]:
    k = int(np.nanargmax(np.where(np.isnan(d3[i]), np.inf, d3[i])))
    print('  class-3 row %d: report %d fast %r literal %r (row peak %.3g)  params %s' % (
        i, k, a[i, k], b[i, k], top[i, 0], np.array2string(p[i], precision=6)))
bad = np.flatnonzero((worst > 1.0) & ~c3)
print('seed %d: %d rows, classes %s; rows beyond 1e-9 relative / 1e-12 of the row\'s peak: %d' % (
    seed, n, np.bincount(cls, minlength=4).tolist(), len(bad)))
for i in bad[:12]:
    print('  row %d class %d excess %.3g  params %s' % (i, cls[i], worst[i], np.array2string(p[i], precision=4)))
