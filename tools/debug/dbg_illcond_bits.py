"""Replays one set-up of tests/test_gpu_parity.py::run_wide_cases and compares the ill-conditioned rows (class 3: the
literal model inside the fast mode, divisions through reciprocals) with the literal kernel bit for bit: which outputs of
which rows differ, by how many ulps.  usage: python tools/debug/dbg_illcond_bits.py <seed> <case>"""
import sys
sys.path.insert(0, '.')
import numpy as np
from smartpy_amd import engine as eng
from oracle import smart_oracle as so
seed, want_case = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
for case in range(want_case + 1):
    dt = float(rng.choice([900.0, 3600.0, 86400.0]))
    gap = int(rng.choice([1, 4, 24]))
    n_rep = int(rng.integers(20, 80))
    T = n_rep * gap
    W = int(rng.integers(0, n_rep // 2 + 1)) * gap if rng.random() < 0.6 else 0
    scale = dt / 86400.0
    rain = rng.gamma(0.4, 8.0, T) * (rng.random(T) < rng.uniform(0.2, 0.9)) * scale * rng.choice([1.0, 10.0])
    peva = np.maximum(0.0, rng.normal(1.5, 1.0, T)) * scale
    peva[rng.random(T) < 0.1] = 0.0
    area = float(np.exp(rng.uniform(np.log(5e6), np.log(5e9))))
    n = int(rng.integers(65, 400))
    params = np.column_stack([
        rng.uniform(0.7, 1.3, n), rng.uniform(-0.2, 1.2, n), rng.uniform(0.0, 0.9, n), rng.uniform(0.0, 1.0, n),
        rng.uniform(0.0, 0.9, n) * (rng.random(n) < 0.5) + rng.uniform(0.0, 0.013, n), rng.uniform(5.0, 300.0, n),
        np.exp(rng.uniform(np.log(0.2), np.log(500.0), n)), np.exp(rng.uniform(np.log(1.0), np.log(3000.0), n)),
        np.exp(rng.uniform(np.log(10.0), np.log(20000.0), n)), np.exp(rng.uniform(np.log(0.2), np.log(300.0), n))])
    extra = {'aar': float(rng.uniform(600, 2500)), 'r-o_ratio': float(rng.uniform(0.2, 0.7)),
             'r-o_split': tuple(rng.dirichlet(np.ones(5)))} if rng.random() < 0.7 else None
    report, rtype = ('summary', so.REPORT_SUMMARY) if rng.random() < 0.7 else ('raw', so.REPORT_RAW)
f = np.stack([rain, peva], 1)
print('seed', seed, 'case', want_case, 'dt', dt, 'gap', gap, 'T', T, 'W', W, 'n', n, report, 'extra', extra is not None, 'area %.4g' % area)
fast = eng.run_ensemble(params, f, area, dt, W, gap, report=report, extra=extra, want_final=True)
print(fast._prepared.describe())
bad = params[:, 9] * 3600.0 < 0.5 * dt
lit = eng.run_ensemble(params[bad], f, area, dt, W, gap, report=report, extra=extra, math_mode='literal', want_final=True)
np.set_printoptions(precision=17, linewidth=220)
for name in ('final_vars', 'discharge', 'gw'):
    a = np.ascontiguousarray(getattr(fast, name).cpu().numpy()[bad]).reshape(int(bad.sum()), -1)
    b = np.ascontiguousarray(getattr(lit, name).cpu().numpy()).reshape(int(bad.sum()), -1)
    ne = a.view(np.int64) != b.view(np.int64)
    print(name, 'rows', a.shape[0], 'differing values', int(ne.sum()), 'in rows', np.nonzero(ne.any(1))[0][:10])
    for r, c in zip(*np.nonzero(ne)):
        if r in np.nonzero(ne.any(1))[0][:2]:
            print('   row %d col %d fast %r literal %r ulps %d' % (r, c, a[r, c], b[r, c], int(a[r, c:c + 1].view(np.int64)[0] - b[r, c:c + 1].view(np.int64)[0])))
rows = np.nonzero(bad)[0][np.nonzero((np.ascontiguousarray(fast.final_vars.cpu().numpy()[bad]).view(np.int64) != np.ascontiguousarray(lit.final_vars.cpu().numpy()).view(np.int64)).any(1))[0]]
for r in rows[:3]:
    print('params of row', r, params[r], 'dt/k', dt / (params[r, 6:10] * 3600))

# ---- the first differing row on its own, step by step (raw reports every step, no warm-up): where do the two leave each other?
if len(rows):
    r = int(rows[0])
    p1 = np.repeat(params[r:r + 1], 64, axis=0)            # a whole wavefront of the row (the reciprocal path is wave-uniform)
    fa = eng.run_ensemble(p1, f, area, dt, 0, 1, report='raw', extra=extra, want_final=True, group_variants=False)
    li = eng.run_ensemble(p1, f, area, dt, 0, 1, report='raw', extra=extra, math_mode='literal', want_final=True)
    a, b = fa.discharge.cpu().numpy()[0], li.discharge.cpu().numpy()[0]
    ne = np.nonzero(a.view(np.int64) != b.view(np.int64))[0]
    print('row', r, 'alone, gap 1:', fa._prepared.describe(), '| steps whose outflow differs:', len(ne), 'first', ne[:5])
    if len(ne):
        t0 = int(ne[0])
        for t in range(max(t0 - 3, 0), t0 + 1):             # the outflow of step t is the river's state ahead of it: look one back
            ff = eng.run_ensemble(p1, f[:t + 1], area, dt, 0, 1, report='raw', extra=extra, want_final=True,
                                  group_variants=False).final_vars.cpu().numpy()[0]
            fl = eng.run_ensemble(p1, f[:t + 1], area, dt, 0, 1, report='raw', extra=extra, math_mode='literal',
                                  want_final=True).final_vars.cpu().numpy()[0]
            d = np.nonzero(ff.view(np.int64) != fl.view(np.int64))[0]
            print('after step', t, 'rain %r peva %r ex %r' % (rain[t], peva[t], rain[t] * params[r, 0] - peva[t]), 'differing vars', d)
            print('   fast   ', ff)
            print('   literal', fl)
