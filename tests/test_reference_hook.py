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
