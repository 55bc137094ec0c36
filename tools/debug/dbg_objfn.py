"""Replays the seeds of tests/test_gpu_parity.py::run_interval_cases (or, with --wide / --batch, ::run_wide_cases /
::run_batch_cases; --raw / --every: run_interval_cases under those reports) given on the command line and prints, for every comparison that is off by more than its tolerance
(or 1e-9 for the purely relative ones), where the largest difference sits and how large the values are there.
--plain leaves SMART_TIME_SLICES / SMART_EXITS alone."""
import os
import sys
sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
import numpy as np
from smartpy_amd import engine
import test_gpu_parity as t
orig_rel, orig_excess = t.rel, t.excess
np.set_printoptions(precision=12, linewidth=200)


def show(kind, r, e, a, b):
    idx = np.unravel_index(np.nanargmax(e), e.shape)
    top = np.abs(b[idx[0]]).max() if b.ndim > 1 else np.abs(b).max()
    print('%s %.3g at %s of shape %s: got %r want %r | abs diff %.3g | largest |value| of that row %.3g' % (
        kind, r, idx, a.shape, a[idx], b[idx], abs(a[idx] - b[idx]), top))


def spy_rel(a, b, floor=0.0):
    r = orig_rel(a, b, floor)
    a, b = np.asarray(a, float), np.asarray(b, float)
    if r > 1e-9:
        m = np.maximum(np.abs(a), np.abs(b))
        show('rel', r, np.where(m > floor, np.abs(a - b) / np.maximum(m, 1e-300), 0), a, b)
    return r


def spy_excess(got, want, rtol, top=None, top_frac=1e-13, tiny=1e-40):
    r = orig_excess(got, want, rtol, top, top_frac, tiny)
    if r > 1.0:
        got, want = np.asarray(got, float), np.asarray(want, float)
        tp = top if top is not None else (np.abs(want).max(axis=-1, keepdims=True) if want.ndim > 1 else np.abs(want).max())
        show('excess', r, np.abs(got - want) / (rtol * np.abs(want) + top_frac * tp + tiny), got, want)
        if want.ndim > 1:
            i = np.unravel_index(np.nanargmax(np.abs(got - want) / (rtol * np.abs(want) + top_frac * tp + tiny)), want.shape)[0]
            print('  got ', got[i][:24])
            print('  want', want[i][:24])
    return r


t.rel, t.excess = spy_rel, spy_excess
plain = '--plain' in sys.argv          # leave SMART_TIME_SLICES / SMART_EXITS alone: the default launch of each case


def setenv(k, v):
    if plain:
        return
    if v:
        os.environ[k] = str(v)
    else:
        os.environ.pop(k, None)


for seed in [int(x) for x in sys.argv[1:] if not x.startswith('--')]:
    try:
        if '--wide' in sys.argv:
            t.run_wide_cases(engine, seed, 10)
        elif '--batch' in sys.argv:
            t.run_batch_cases(engine, seed, 10)
        elif '--every' in sys.argv:
            t.run_interval_cases(engine, setenv, seed, 5, mode='every')      # (fuzz_wide.py runs cases // 2 of these)
        elif '--raw' in sys.argv:
            t.run_interval_cases(engine, setenv, seed, 10, mode='raw')
        else:
            t.run_interval_cases(engine, setenv, seed, 10)
        print('seed', seed, 'passed')
    except AssertionError as e:
        print('seed', seed, 'failed:', str(e)[:150])
