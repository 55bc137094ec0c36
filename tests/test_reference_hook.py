"""The plug-in contract, exercised with the REAL reference where it is available (build container only):
register smartpy_amd.smartcpp as `smartcpp`, import the unmodified reference, and check that its structure.run
routes the warm-up and the run through our allsteps() with the arguments we expect.  No GPU here, so the launch
behind allsteps is replaced by the CPU oracle; on the GPU box the same call path ends in smart_allsteps_hip
(tests/test_gpu_api.py::test_smartcpp_module_contract)."""
import importlib
import os
import shutil
import sys
from datetime import datetime, timedelta

import numpy as np
import pytest

from conftest import load_golden

REF = '/root/reference'

pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, 'smartpy')),
                                reason='the reference is only mounted in the build container')


def test_unmodified_reference_runs_through_our_smartcpp(tmp_path, monkeypatch, example):
    from oracle import smart_oracle as so
    from smartpy_amd import engine
    calls = []

    def allsteps_on_cpu(area, dt, length, rain, peva, params, initial, report_type, gap):
        calls.append((length, len(rain), np.asarray(initial).shape, report_type, gap))
        return so.all_steps(area, dt, length, rain, peva, params, initial, report_type, gap)

    monkeypatch.setattr(engine, 'allsteps', allsteps_on_cpu)
    for name in [m for m in sys.modules if m == 'smartpy' or m.startswith('smartpy.')] + ['smartcpp']:
        monkeypatch.delitem(sys.modules, name, raising=False)
    import smartpy_amd.smartcpp as shim
    importlib.reload(shim)                       # re-binds allsteps to the patched engine function
    monkeypatch.setitem(sys.modules, 'smartcpp', shim)
    monkeypatch.syspath_prepend(REF)
    monkeypatch.setattr(sys, 'dont_write_bytecode', True)
    smartpy = importlib.import_module('smartpy')
    try:
        from smartpy import structure
        assert structure.smart_in_cpp and structure.smartcpp is shim          # structure.py:22-24 took the hook
        data = tmp_path / 'data'
        shutil.copytree(os.path.join(REF, 'tests', 'data'), data)
        for dirpath, _, files in os.walk(data):
            os.chmod(dirpath, 0o755)
        sm = smartpy.SMART('Catchment', 175.46e6, datetime(2007, 1, 1, 9), datetime(2016, 12, 31, 9),
                           timedelta(hours=1), timedelta(days=1), 365, 'csv', 'csv', str(data) + os.sep,
                           gauged_area_m2=175.97e6)
        sm.extra = example['extra']
        sm.parameters.set_parameters_with_file(sm.in_f + 'Catchment.parameters')
        discharge, gw = sm.simulate(sm.parameters.values)
    finally:
        for name in [m for m in sys.modules if m == 'smartpy' or m.startswith('smartpy.')]:
            sys.modules.pop(name, None)
    # warm-up call: full-length forcing with the shorter length, then the run (structure.py:118-121,143-146)
    assert calls == [(8760, 87672, (19,), 1, 24), (87672, 87672, (19,), 1, 24)]
    g1 = load_golden('g1_reference_test.npz')
    for i, v in zip(g1['report_index'], g1['expected']):
        assert '%.6e' % discharge[i] == '%.6e' % v
    assert np.array_equal(discharge, load_golden('kat1_hourly.npz')['discharge_summary'])


def _write_series(path, header, stamps, values, fmt='%.6e'):
    with open(path, 'w') as f:
        f.write('DateTime,%s\n' % header)
        for s, v in zip(stamps, values):
            f.write('%s,%s\n' % (s.strftime('%Y-%m-%d %H:%M:%S'), v if isinstance(v, str) else fmt % v))


@pytest.mark.parametrize('seed', range(16))
def test_input_pipeline_against_the_reference_on_random_files(tmp_path, monkeypatch, seed):
    """Random rain / PE files (hourly, 3-hourly or daily stamps at random offsets) and irregular flow files (missing
    days, '', -99, different clock time), random simulation and report steps: the arrays our readers + resamplers
    produce must equal the reference's (smart.py:130-143), exceptions included."""
    rng = np.random.default_rng(1000 + seed)
    d_data = timedelta(hours=int(rng.choice([1, 3, 24])))
    d_simu = timedelta(hours=int(rng.choice([1, 2, 3, 6, 12, 24])))
    d_save = timedelta(hours=24)
    if d_save.total_seconds() % d_simu.total_seconds():
        d_simu = timedelta(hours=6)
    t0 = datetime(2001, 3, 1, int(rng.integers(0, 24)))
    n_data = int(rng.integers(24 * 40, 24 * 60) * 3600 // d_data.total_seconds())
    stamps = [t0 + k * d_data for k in range(n_data)]
    root = tmp_path / 'data'
    cat = root / 'in' / 'C'
    os.makedirs(cat)
    _write_series(cat / 'C.rain', 'rain', stamps, rng.gamma(0.5, 3.0, n_data) * (rng.random(n_data) < 0.6))
    _write_series(cat / 'C.peva', 'peva', stamps, rng.random(n_data) * 0.3)
    # daily mean flows at some clock time, with holes
    f0 = datetime(2001, 2, 20, int(rng.integers(0, 24)))
    fst, fval = [], []
    for k in range(90):
        u = rng.random()
        if u < 0.08:
            continue                                    # missing row
        fst.append(f0 + timedelta(days=k))
        fval.append('' if u < 0.12 else ('-99' if u < 0.16 else '%.3f' % (rng.random() * 20)))
    _write_series(cat / 'C.flow', 'flow', fst, fval)
    start = datetime(2001, 3, 8, int(rng.integers(0, 24))) + timedelta(days=int(rng.integers(0, 5)))
    end = start + timedelta(days=int(rng.integers(5, 25)))

    for name in [m for m in sys.modules if m == 'smartpy' or m.startswith('smartpy.')] + ['smartcpp']:
        monkeypatch.delitem(sys.modules, name, raising=False)
    monkeypatch.syspath_prepend(REF)
    monkeypatch.setattr(sys, 'dont_write_bytecode', True)
    smartpy = importlib.import_module('smartpy')
    import smartpy_amd
    args = ('C', 50e6, start, end, d_simu, d_save, 0, 'csv', 'csv', str(root) + os.sep)
    try:
        try:
            ref = smartpy.SMART(*args, gauged_area_m2=47e6)
            ref_err = None
        except Exception as e:                          # e.g. data not sufficient, time deltas not multiples
            ref, ref_err = None, e
        try:
            ours = smartpy_amd.SMART(*args, gauged_area_m2=47e6)
            our_err = None
        except Exception as e:
            ours, our_err = None, e
    finally:
        for name in [m for m in sys.modules if m == 'smartpy' or m.startswith('smartpy.')]:
            sys.modules.pop(name, None)
    assert (ref_err is None) == (our_err is None), (ref_err, our_err)
    if ref_err is not None:
        assert str(ref_err) == str(our_err) or type(ref_err) is type(our_err)
        print('both raise:', ref_err)
        return
    assert ours.timeseries == ref.timeseries and ours.timeseries_report == ref.timeseries_report
    assert np.array_equal(ours.nd_rain, ref.nd_rain) and np.array_equal(ours.nd_peva, ref.nd_peva)
    assert np.array_equal(ours.nd_flow, ref.nd_flow, equal_nan=True)
