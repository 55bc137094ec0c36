"""The reference's Python surface on top of the HIP engine, end to end on the GPU: SMART from the example
files, the Monte-Carlo classes and their database, the smartcpp-compatible module."""
import os
import shutil
from datetime import datetime, timedelta

import numpy as np
import pytest

from conftest import load_golden, GOLDEN
from oracle import smart_oracle as so
from oracle import objfn_oracle

pytestmark = pytest.mark.gpu

EXTRA = {'aar': 1200, 'r-o_ratio': 0.45, 'r-o_split': (0.10, 0.15, 0.15, 0.30, 0.30)}


@pytest.fixture()
def root(tmp_path):
    r = str(tmp_path / 'data')
    shutil.copytree(os.path.join(GOLDEN, 'data', 'in'), os.path.join(r, 'in'))
    return r


def test_smart_from_files_reproduces_the_reference_unit_test(root):
    """The scenario of the reference's tests/test_run_daily_to_hourly.py, same constructor call, same checks."""
    import smartpy_amd
    sm = smartpy_amd.SMART(
        catchment='Catchment', catchment_area_m2=175.46 * 1E6,
        start=datetime.strptime('01/01/2007 09:00:00', '%d/%m/%Y %H:%M:%S'),
        end=datetime.strptime('31/12/2016 09:00:00', '%d/%m/%Y %H:%M:%S'),
        time_delta_simu=timedelta(hours=1), time_delta_save=timedelta(days=1), warm_up_days=365,
        in_format='csv', out_format='csv', root=root, gauged_area_m2=175.97 * 1E6)
    sm.extra = EXTRA
    sm.parameters.set_parameters_with_file(''.join([sm.in_f, sm.catchment, '.parameters']))
    discharge, gw = sm.simulate(sm.parameters.values)
    assert sm.nd_discharge is discharge and sm.gw_contribution == gw and sm.get_simulation_array() is discharge
    g1 = load_golden('g1_reference_test.npz')
    for i, v in zip(g1['report_index'], g1['expected']):
        assert '%.6e' % sm.nd_discharge[i] == '%.6e' % v
    assert abs(gw - 0.0529870) < 1e-6
    # the output file equals the example's committed ExampleDaily.mod.flow (G2) after CRLF -> LF
    sm.write_output_files(which='both')
    g2 = load_golden('g2_g3_example_flows.npz')
    lines = open(os.path.join(sm.out_f, 'Catchment.mod.flow'), newline='').read().split('\r\n')
    assert lines[0] == 'DateTime,flow' and len(lines) == 3655
    assert [ln.split(',')[1] for ln in lines[1:-1]] == ['%e' % v for v in g2['mod_flow']]
    assert lines[1].split(',')[0] == '2007-01-01 09:00:00'
    raw, gw_raw = sm.simulate(sm.parameters.values, report='raw')
    k1 = load_golden('kat1_hourly.npz')
    assert np.allclose(raw, k1['discharge_raw'], rtol=1e-9, atol=0) and abs(gw_raw - float(k1['gw_raw'])) < 1e-10
    with pytest.raises(Exception, match="Reporting type 'weekly' unknown"):
        sm.simulate(sm.parameters.values, report='weekly')
    # batched entry: rows of a matrix, objective functions fused
    rows = np.array([[sm.parameters.values[n] for n in sm.parameters.names]] * 3)
    rows[1, 0], rows[2, 9] = 0.95, 40.0
    ens = sm.simulate_ensemble(rows, objective_functions=True, gw_constraint=0.12667)
    assert ens.discharge.shape == (3, 3653) and ens.objfn.shape == (3, 8)
    assert np.allclose(ens.discharge[0].cpu().numpy(), discharge, rtol=1e-12)
    assert abs(float(ens.objfn[0, 0]) - 0.390445) < 1e-6           # NSE of the example (notebook cell 40)


def test_from_arrays_and_warmup_error(example):
    import smartpy_amd
    sm = smartpy_amd.SMART.from_arrays(example['area'], datetime(2007, 1, 1, 9), datetime(2007, 3, 1, 9),
                                       timedelta(days=1), timedelta(days=1), 10, example['rain_daily'][:60],
                                       example['peva_daily'][:60])
    sm.extra = EXTRA
    d, g = sm.simulate(dict(zip(sm.parameters.names, example['params'])))
    want, gw, _ = so.run(example['area'], 86400.0, 60, 10, example['rain_daily'], example['peva_daily'],
                         example['params'], EXTRA, so.REPORT_SUMMARY, 1)
    assert np.allclose(d, want, rtol=1e-10, atol=0) and abs(g - gw) < 1e-10
    sm.warm_up = 61
    with pytest.raises(Exception, match='warm-up duration'):
        sm.simulate(dict(zip(sm.parameters.names, example['params'])))
    with pytest.raises(Exception, match='observation array does not exist'):
        sm.get_evaluation_array()


def test_a_repeated_simulate_uploads_its_ten_parameters_and_nothing_else(example):
    """SMART.simulate() keeps the forcing, the output buffers and the plan on the device (engine.SingleRun): the second
    call moves 80 bytes to the device -- no forcing, no planning -- and a row of another class (a stiff one, an
    ill-conditioned one) gets its own kernel; every call equals the one-shot path bit for bit."""
    import smartpy_amd
    from smartpy_amd import engine, structure
    sm = smartpy_amd.SMART.from_arrays(example['area'], datetime(2007, 1, 1, 9), datetime(2007, 12, 31, 9),
                                       timedelta(days=1), timedelta(days=1), 30, example['rain_daily'][:365],
                                       example['peva_daily'][:365])
    sm.extra = EXTRA
    names = sm.parameters.names
    base = np.array(example['params'], dtype=float)
    stiff, wild = base.copy(), base.copy()
    stiff[6], wild[9] = 10.0, 3.0            # SK < 24 h: clamps reachable; RK < 12 h: the river is ill-conditioned
    first = sm.simulate(dict(zip(names, base)))
    for row in (stiff, wild, base, stiff):
        before = engine.h2d_bytes
        d, g = sm.simulate(dict(zip(names, row)))
        assert engine.h2d_bytes - before == 80
        want = structure.run(sm.area, sm.delta_simu, sm.nd_rain, sm.nd_peva, row, sm.extra, sm.timeseries,
                             sm.timeseries_report, 'summary', warm_up=sm.warm_up)
        assert bits_equal(d, want[0]) and g == want[1]
    assert bits_equal(sm.simulate(dict(zip(names, base)))[0], first[0])
    # another report type is another kept run; the series are COMPARED with what is on the device on every call: the same
    # numbers in another array cost nothing, a value written into the user's own array is noticed
    raw = sm.simulate(dict(zip(names, base)), report='raw')
    assert raw[0].shape == first[0].shape and len(sm._single) == 2
    sm.nd_rain = sm.nd_rain * 1.0
    before = engine.h2d_bytes
    again = sm.simulate(dict(zip(names, base)))
    assert engine.h2d_bytes - before == 80 and bits_equal(again[0], first[0])
    sm.nd_rain[100] += 5.0                                   # in place: same array object, other numbers
    before = engine.h2d_bytes
    wetter = sm.simulate(dict(zip(names, base)))
    assert engine.h2d_bytes - before > 80 and not bits_equal(wetter[0], first[0]) and len(sm._single) == 1
    want = structure.run(sm.area, sm.delta_simu, sm.nd_rain, sm.nd_peva, base, sm.extra, sm.timeseries,
                         sm.timeseries_report, 'summary', warm_up=sm.warm_up)
    assert bits_equal(wetter[0], want[0])
    sm.nd_rain[100] -= 5.0
    assert bits_equal(sm.simulate(dict(zip(names, base)))[0], first[0])


def bits_equal(a, b):
    a, b = np.ascontiguousarray(a, dtype=np.float64), np.ascontiguousarray(b, dtype=np.float64)
    return a.shape == b.shape and np.array_equal(a.view(np.int64), b.view(np.int64))


def _settings(root, name, start, end, warm):
    with open(os.path.join(root, 'in', 'Catchment', name), 'w') as f:
        f.write('ARGUMENT,VALUE\ncatchment_area_km2,175.46\ngauged_area_km2,175.97\nstart_datetime,%s 09:00:00\n'
                'end_datetime,%s 09:00:00\nsimu_timedelta_min,60\nreport_timedelta_min,1440\nwarm_up_days,%d\n'
                'gw_constraint,0.12667\n' % (start, end, warm))


def test_monte_carlo_pipeline_lhs_glue_best_total(root, example):
    """LHS sampling on one period, then GLUE / Best / Total on another: files, formats and numbers."""
    from smartpy_amd.montecarlo import LHS, GLUE, Best, Total
    _settings(root, 'Catchment.sampling.sttngs', '01/01/2007', '31/12/2008', 365)
    _settings(root, 'Catchment.evaluating.sttngs', '01/01/2009', '31/12/2009', 365)
    np.random.seed(7)
    lhs = LHS('Catchment', root, 'csv', 'csv', 300, save_sim=True, settings_filename='Catchment.sampling.sttngs')
    lhs.model.extra = EXTRA
    assert lhs.lhs_params.shape == (300, 10) and len(lhs.p_map) == 300 and lhs.params[9].name == 'RK'
    lhs.run()
    db = os.path.join(root, 'out', 'Catchment', 'Catchment.SMART.lhs')
    table = np.loadtxt(db, delimiter=',', skiprows=1)
    assert table.shape == (300, 8 + 10 + 731)
    # against the oracle on the same rows (float32 '%.6e' text: 1e-6 relative)
    T = 731 * 24
    dis, gw, _ = so.run_batch(example['area'], 3600.0, T, 8760, example['rain_hourly'], example['peva_hourly'],
                              lhs.lhs_params, EXTRA, so.REPORT_SUMMARY, 24)
    want = objfn_oracle.objective_matrix(dis, example['flow_obs'][:731], gw, 0.12667)
    assert np.allclose(table[:, :7], want[:, :7], rtol=2e-6, atol=1e-9) and np.array_equal(table[:, 7], want[:, 7])
    assert np.allclose(table[:, 8:18], lhs.lhs_params, rtol=1e-6) and np.allclose(table[:, 18:], dis, rtol=2e-6, atol=1e-12)
    assert np.allclose(lhs.obj_fns[:, :7], want[:, :7], rtol=1e-9, atol=1e-12)

    glue = GLUE('Catchment', root, 'csv', 'csv', conditioning={'KGE': ('min', (0.3,)), 'PBias': ('inside', (-30.0, 30.0))},
                settings_filename='Catchment.evaluating.sttngs')
    glue.model.extra = EXTRA
    keep = (table[:, 1].astype(np.float32) >= 0.3) & (np.abs(table[:, 5]) <= 30.0)
    assert 0 < keep.sum() < 300 and glue.behavioural_params.shape == (int(keep.sum()), 10)
    assert np.array_equal(glue.behavioural_params, table[keep, 8:18].astype(np.float32))
    glue.run(compression=True)
    assert os.path.exists(os.path.join(root, 'out', 'Catchment', 'Catchment.SMART.glue.gz'))

    best = Best('Catchment', root, 'csv', 'csv', target='NSE', nb_best=5, constraining={'GW': ('equal', (1.0,))},
                settings_filename='Catchment.evaluating.sttngs')
    best.model.extra = EXTRA
    ok = table[:, 7] == 1.0
    top = table[ok][np.argsort(table[ok, 0].astype(np.float32))][-5:]
    assert np.array_equal(best.best_params, top[:, 8:18].astype(np.float32))
    best.run()
    t5 = np.loadtxt(os.path.join(root, 'out', 'Catchment', 'Catchment.SMART.5best'), delimiter=',', skiprows=1)
    assert t5.shape == (5, 18)
    # the evaluation period really is simulated: compare with the oracle on 2009 with the float32 parameters
    s0 = (datetime(2009, 1, 1) - datetime(2007, 1, 1)).days
    d9, g9, _ = so.run_batch(example['area'], 3600.0, 365 * 24, 8760, example['rain_hourly'][s0 * 24:],
                             example['peva_hourly'][s0 * 24:], best.best_params.astype(np.float64), EXTRA,
                             so.REPORT_SUMMARY, 24)
    w9 = objfn_oracle.objective_matrix(d9, example['flow_obs'][s0:s0 + 365], g9, 0.12667)
    assert np.allclose(t5[:, :7], w9[:, :7], rtol=2e-6, atol=1e-9)

    total = Total('Catchment', root, 'csv', 'csv', settings_filename='Catchment.evaluating.sttngs')
    total.model.extra = EXTRA
    total.run()
    tt = np.loadtxt(os.path.join(root, 'out', 'Catchment', 'Catchment.SMART.total'), delimiter=',', skiprows=1)
    assert tt.shape == (300, 18) and np.allclose(tt[:, 8:], table[:, 8:18], rtol=1e-6)

    # the per-sample protocol of the reference still works (spotpy setup methods)
    vec = lhs.lhs_params[0]
    sim = lhs.simulation(vec)
    like = lhs.objectivefunction(sim, lhs.evaluation())
    assert np.allclose(like[:7], want[0, :7], rtol=1e-9, atol=1e-12) and like[7] == want[0, 7]


def test_second_stages_on_the_device_pick_the_references_rows(root):
    """KAT-12 through the constructors: a finished sampling run whose 48 parameter rows and objective functions are
    the fixture's matrices, handed to GLUE / Best as `sampling=` (mask, count and top-n on the GPU) and, through its
    database file, to the file-based constructors.  Expected rows: what the REFERENCE's _get_behavioural_sets /
    _get_best_sets returned for the same matrices (tests/golden/make_golden.py).  Where equal keys leave the order
    to numpy's unstable sort, both constructors must still agree with each other row for row."""
    import json
    import torch
    from smartpy_amd.montecarlo import LHS, GLUE, Best
    z = load_golden('kat12_selection.npz')
    with open(os.path.join(GOLDEN, 'kat12_selection.json')) as fh:
        cases = json.load(fh)
    params, fns = z['params'], z['obj_fns']
    _settings(root, 'Catchment.sampling.sttngs', '01/01/2007', '31/12/2007', 180)
    lhs = LHS('Catchment', root, 'csv', 'csv', 48, settings_filename='Catchment.sampling.sttngs')
    assert lhs.obj_fn_names[-1] == 'GW'
    lhs._set_sample(params.astype(np.float64))
    lhs.obj_fns = fns.astype(np.float64)
    lhs.device_obj_fns = torch.from_numpy(lhs.obj_fns).cuda()
    db = lhs._open_database()                       # the file the file-based constructors read
    db.write_table(lhs.obj_fns, lhs._sample)
    lhs._finish_database(db, None)
    names = lhs.obj_fn_names
    kw = dict(settings_filename='Catchment.sampling.sttngs')
    for c in cases['glue']:
        cond = {names[col]: (kind, tuple(val)) for col, kind, val in zip(c['columns'], c['kinds'], c['values'])}
        dev = GLUE('Catchment', root, 'csv', 'csv', conditioning=cond, sampling=lhs, **kw)
        fil = GLUE('Catchment', root, 'csv', 'csv', conditioning=cond, **kw)
        assert dev._device_obj_fns.is_cuda
        assert [int(v) for v in dev.behavioural_params[:, 0]] == c['rows'], c
        assert np.array_equal(dev.behavioural_params, fil.behavioural_params) and \
            dev.behavioural_params.dtype == np.float32
    for c in cases['best']:
        con = {names[col]: (kind, tuple(val)) for col, kind, val in zip(c['columns'], c['kinds'], c['values'])}
        args = dict(target=names[c['target']], nb_best=c['nb_best'], constraining=con, **kw)
        dev = Best('Catchment', root, 'csv', 'csv', sampling=lhs, **args)
        fil = Best('Catchment', root, 'csv', 'csv', **args)
        assert np.array_equal(dev.best_params, fil.best_params), c           # ties included
        rows = [int(v) for v in dev.best_params[:, 0]]
        if not c['ambiguous']:
            assert rows == c['rows'], c
        else:
            assert [repr(float(fns[r, c['target']])) for r in rows] == c['keys'] and len(set(rows)) == len(rows), c


def test_second_stages_from_the_device_and_device_sampling(root, example):
    """(f1) GLUE / Best handed the finished LHS run itself (`sampling=`): the behavioural mask, its count and the
    top-n rows are evaluated on the GPU over the objective functions the launch left there, and select exactly the
    rows -- with the database's float32 '%.6e' rounding -- that the file-based constructors read back.
    (f3) LHS(device_sampling=True): the Latin hypercube is drawn on the GPU and stays there for the launch."""
    import torch
    from smartpy_amd.montecarlo import LHS, GLUE, Best, Total
    _settings(root, 'Catchment.sampling.sttngs', '01/01/2007', '31/12/2007', 180)
    _settings(root, 'Catchment.evaluating.sttngs', '01/01/2008', '30/06/2008', 90)
    np.random.seed(11)
    lhs = LHS('Catchment', root, 'csv', 'csv', 1000, settings_filename='Catchment.sampling.sttngs')
    lhs.model.extra = EXTRA
    lhs.run()
    assert lhs.device_obj_fns.is_cuda and lhs.device_obj_fns.shape == (1000, 8) and lhs.device_gw.shape == (1000,)
    assert np.array_equal(lhs.device_obj_fns.cpu().numpy(), lhs.obj_fns)
    cond = {'KGE': ('min', (0.2,)), 'PBias': ('inside', (-40.0, 40.0)), 'GW': ('equal', (1.0,))}
    from_file = GLUE('Catchment', root, 'csv', 'csv', conditioning=cond, settings_filename='Catchment.evaluating.sttngs')
    on_device = GLUE('Catchment', root, 'csv', 'csv', conditioning=cond, settings_filename='Catchment.evaluating.sttngs',
                     sampling=lhs)
    assert 0 < from_file.behavioural_params.shape[0] < 1000
    assert on_device.behavioural_params.dtype == np.float32
    assert np.array_equal(on_device.behavioural_params, from_file.behavioural_params)
    assert np.array_equal(on_device.sampled_params, from_file.sampled_params)
    assert np.array_equal(on_device.sampled_obj_fns, from_file.sampled_obj_fns)
    for target in ('NSE', 'RMSE'):
        kw = dict(target=target, nb_best=25, constraining={'KGEa': ('max', (1.2,))},
                  settings_filename='Catchment.evaluating.sttngs')
        a = Best('Catchment', root, 'csv', 'csv', **kw)
        b = Best('Catchment', root, 'csv', 'csv', sampling=lhs, **kw)
        assert a.best_params.shape == (25, 10) and np.array_equal(a.best_params, b.best_params)
    with pytest.raises(Exception, match='higher than the restrained sample size'):
        Best('Catchment', root, 'csv', 'csv', target='NSE', nb_best=900, constraining={'NSE': ('min', (0.3,))},
             settings_filename='Catchment.evaluating.sttngs', sampling=lhs)
    tot = Total('Catchment', root, 'csv', 'csv', settings_filename='Catchment.evaluating.sttngs', sampling=lhs)
    assert np.array_equal(tot._sample, from_file.sampled_params.astype(np.float64))
    on_device.model.extra = from_file.model.extra = EXTRA
    on_device.run()
    first = np.array(on_device.obj_fns)
    from_file.run()
    assert np.array_equal(first, from_file.obj_fns)                 # same rows, same evaluation-period results
    with pytest.raises(Exception, match='has not been run yet'):
        GLUE('Catchment', root, 'csv', 'csv', conditioning=cond, settings_filename='Catchment.evaluating.sttngs',
             sampling=LHS('Catchment', root, 'csv', 'csv', 5, settings_filename='Catchment.sampling.sttngs'))

    # ---- (f3) the sample drawn on the device
    dev = LHS('Catchment', root, 'csv', 'csv', 5000, settings_filename='Catchment.sampling.sttngs',
              device_sampling=True, seed=4)
    assert dev.device_sample.is_cuda and dev.lhs_params.shape == (5000, 10)
    assert np.array_equal(dev.device_sample.cpu().numpy(), dev.lhs_params)
    ranges = dev.model.parameters.ranges
    for j, name in enumerate(dev.param_names):                      # a Latin hypercube: one value per stratum
        lo, hi = ranges[name]
        strata = np.floor((dev.lhs_params[:, j] - lo) / (hi - lo) * 5000).astype(int)
        assert sorted(np.clip(strata, 0, 4999)) == list(range(5000))
    again = LHS('Catchment', root, 'csv', 'csv', 5000, settings_filename='Catchment.sampling.sttngs',
                device_sampling=True, seed=4)
    assert np.array_equal(again.lhs_params, dev.lhs_params)
    dev.model.extra = EXTRA
    dev.run()
    rows = [0, 1777, 4999]
    dis, gw, _ = so.run_batch(example['area'], 3600.0, 365 * 24, 180 * 24, example['rain_hourly'],
                              example['peva_hourly'], dev.lhs_params[rows], EXTRA, so.REPORT_SUMMARY, 24)
    want = objfn_oracle.objective_matrix(dis, example['flow_obs'][:365], gw, 0.12667)
    assert np.allclose(dev.obj_fns[rows, :7], want[:, :7], rtol=1e-9, atol=1e-12)


def test_smartcpp_module_contract(example):
    """What the reference's hook calls: smartcpp.allsteps(...)[2] for the warm-up, [0:2] for the run
    (structure.py:118-121,143-146); smartcpp.onestep per step for old versions (structure.py:171-187)."""
    from smartpy_amd import smartcpp, structure
    k1 = load_golden('kat1_hourly.npz')
    assert tuple(int(x) for x in smartcpp.__version__.split('.')) >= (0, 2, 0)
    init = smartcpp.allsteps(example['area'], 3600.0, 8760, example['rain_hourly'], example['peva_hourly'],
                             example['params'], k1['initial_warmup'], 1, 24)[2]
    assert np.allclose(init[7:], k1['initial_run'][7:], rtol=1e-12, atol=0)
    dis, gw = smartcpp.allsteps(example['area'], 3600.0, 24 * 90, example['rain_hourly'], example['peva_hourly'],
                                example['params'], init, 1, 24)[0:2]
    assert np.allclose(dis, k1['discharge_summary'][:90], rtol=1e-12, atol=0)
    row = np.zeros(19)
    row[:] = smartcpp.onestep(example['area'], 3600.0, example['rain_hourly'][0], example['peva_hourly'][0],
                              *example['params'], *k1['initial_run'][7:])
    assert np.allclose(row, k1['first48'][1], rtol=1e-13, atol=0)
    assert structure.run_one_step(example['area'], 3600.0, example['rain_hourly'][0], example['peva_hourly'][0],
                                  *example['params'], *k1['initial_run'][7:]) == tuple(row)


@pytest.mark.parametrize('in_fmt,out_fmt', [('csv', 'csv'), ('netcdf', 'csv'), ('csv', 'netcdf')])
def test_run_lhs_terminates_like_the_reference_smoke_test(root, in_fmt, out_fmt):
    """Mirror of the reference's tests/test_run_mc_lhs.py: sample_size=5, save_sim=True, compression=True, for the
    input / output formats.  Without the optional netCDF4 package the NetCDF variants raise the reference's own
    message (inout.py:146-147, montecarlo.py:119-121) instead of running."""
    from smartpy_amd.montecarlo import LHS
    try:
        import netCDF4  # noqa: F401
        have_netcdf = True
    except ImportError:
        have_netcdf = False
    if 'netcdf' in (in_fmt, out_fmt) and not have_netcdf:
        with pytest.raises(Exception, match="requires the package 'netCDF4'"):
            LHS(catchment='Catchment', root_f=root, in_format=in_fmt, out_format=out_fmt, sample_size=5, parallel='seq',
                save_sim=True).run(compression=True)
        return
    lhs = LHS(catchment='Catchment', root_f=root, in_format=in_fmt, out_format=out_fmt, sample_size=5, parallel='seq',
              save_sim=True)
    lhs.run(compression=True)
    assert os.path.exists(os.path.join(root, 'out', 'Catchment', 'Catchment.SMART.lhs.gz'))
    assert lhs.obj_fns.shape == (5, 8) and np.all(np.isfinite(lhs.obj_fns))


def test_netcdf_in_and_out_through_the_dataset_double(root, monkeypatch):
    """The fourth variant of the reference's smoke test (NetCDF in, NetCDF out) plus a second stage reading the .nc
    database, with tests/fake_netcdf.FakeDataset standing in for the absent netCDF4 package: the .nc inputs are made
    from the example's CSV files, and every number must equal the CSV-in / CSV-out run of the same sample."""
    import csv
    from fake_netcdf import FakeDataset
    import smartpy_amd.inout as inout
    import smartpy_amd.montecarlo.database as database
    from smartpy_amd.montecarlo import LHS, GLUE
    monkeypatch.setattr(inout, 'Dataset', FakeDataset)
    monkeypatch.setattr(database, 'Dataset', FakeDataset)
    folder = os.path.join(root, 'in', 'Catchment')
    for kind in ('rain', 'peva', 'flow'):
        with open(os.path.join(folder, 'Catchment.' + kind)) as f:
            rows = list(csv.DictReader(f))
        stamps = [(datetime.strptime(r['DateTime'], '%Y-%m-%d %H:%M:%S') - datetime(1970, 1, 1)).total_seconds()
                  for r in rows]
        values = [float(r[kind]) if r[kind] not in ('', '-99') else np.nan for r in rows]
        with FakeDataset(os.path.join(folder, 'Catchment.%s.nc' % kind), 'w') as nc:
            nc.createDimension('DateTime', len(rows))
            nc.createVariable('DateTime', np.float64, ('DateTime',))
            nc.createVariable(kind, np.float64, ('DateTime',))
            nc.variables['DateTime'][0:len(rows)] = stamps
            nc.variables[kind][0:len(rows)] = values
    _settings(root, 'Catchment.short.sttngs', '01/01/2007', '31/12/2007', 100)
    np.random.seed(3)
    a = LHS('Catchment', root, 'csv', 'csv', 40, save_sim=True, settings_filename='Catchment.short.sttngs')
    a.model.extra = EXTRA
    a.run()
    np.random.seed(3)
    b = LHS('Catchment', root, 'netcdf', 'netcdf', 40, save_sim=True, settings_filename='Catchment.short.sttngs')
    b.model.extra = EXTRA
    assert np.array_equal(b.model.nd_rain, a.model.nd_rain) and np.array_equal(b.model.nd_peva, a.model.nd_peva)
    assert np.array_equal(b.model.nd_flow, a.model.nd_flow, equal_nan=True)
    b.run(compression=True)
    assert np.array_equal(b.obj_fns, a.obj_fns) and b.db_file.endswith('Catchment.SMART.lhs.nc')
    nc = FakeDataset(b.db_file, 'r')
    assert nc.variables['Parameters'].filters == {'zlib': True, 'complevel': 6}
    assert np.array_equal(nc.variables['ObjFunctions'][:, :], a.obj_fns.astype(np.float32))
    assert nc.variables['Simulations'][:, :].shape == (40, 365)
    assert os.path.exists(os.path.join(root, 'out', 'Catchment', 'Catchment.obs.flow.nc'))
    glue = GLUE('Catchment', root, 'netcdf', 'netcdf', conditioning={'KGE': ('min', (0.0,))},
                settings_filename='Catchment.short.sttngs')
    keep = a.obj_fns[:, 1].astype(np.float32) >= 0.0
    assert 0 < keep.sum() and np.array_equal(glue.behavioural_params, a.lhs_params[keep].astype(np.float32))
    # the per-sample protocol writes NetCDF rows at the index of their parameter set (montecarlo.py:215-224)
    b._init_db()
    like = [0.5] * 8
    b.save(like, b.lhs_params[7], [np.arange(365.0)])
    b.database.close()
    assert np.array_equal(FakeDataset(b.db_file, 'r').variables['ObjFunctions'][7], np.float32(like))


def test_the_c_abi_from_a_c_program(tmp_path, example):
    """examples/ensemble_from_c.c -- plain C, no Python, no torch in the process -- plans, launches (time-sliced: 70,000
    samples) and reads back through include/smart_amd.h; its numbers are those of the Python binding, bit for bit."""
    import subprocess
    from test_capi_symbols import build_c_example
    from smartpy_amd import engine
    from oracle import lhs_oracle
    exe = build_c_example(tmp_path)
    N, T, W, gap = 70000, 24 * 120, 24 * 20, 24
    params = lhs_oracle.lhs_params(N, seed=21)
    forcing = np.stack([example['rain_hourly'][:T], example['peva_hourly'][:T]], axis=1)
    obs = example['flow_obs'][:T // gap]
    for name, a in (('params', params), ('forcing', forcing), ('obs', obs)):
        np.ascontiguousarray(a, dtype='<f8').tofile(str(tmp_path / (name + '.bin')))
    r = subprocess.run([exe, str(tmp_path / 'params.bin'), str(tmp_path / 'forcing.bin'), str(tmp_path / 'obs.bin'),
                        str(N), str(T), str(W), str(gap), repr(example['area']), str(tmp_path / 'out.bin')],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert 'smart_fast_intervals[2 slices x 1094 blocks' in r.stdout and 'plan 0x111' in r.stdout
    out = np.fromfile(str(tmp_path / 'out.bin'), dtype='<f8')
    objfn, gw, dis = out[:N * 8].reshape(N, 8), out[N * 8:N * 9], out[N * 9:].reshape(T // gap, N)
    ref = engine.run_ensemble(params, forcing, example['area'], 3600.0, W, gap, extra=EXTRA, obs=obs, gw_obs=0.12667)
    assert np.array_equal(objfn, ref.objfn.cpu().numpy()) and np.array_equal(gw, ref.gw.cpu().numpy())
    assert np.array_equal(dis, ref.discharge_report_major.cpu().numpy())


def test_init_makes_an_rccl_subgroup_beside_the_gloo_default_group_and_probes_it():
    """Round 6's distributed.init(): a gloo default group (host objects, agreements) + an RCCL subgroup for the device
    tensors whose first collective runs at once, in a thread, with a bound.  A one-GPU box cannot hold two RCCL ranks,
    but a group of ONE runs every call of that path: new_group(backend='nccl'), the polled first all-reduce, and the
    package's collectives through the subgroup (gather, broadcast, point-to-point collection, barrier) -- then
    finish()."""
    import subprocess
    import sys
    code = r'''
import os, sys
sys.path.insert(0, %r)
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29641', RANK='0', WORLD_SIZE='1', LOCAL_RANK='0',
                  SMART_DIST_SINGLE='1')
os.environ.pop('SMART_DIST_BACKEND', None)
import numpy as np, torch, torch.distributed as dist
from smartpy_amd import distributed as sd
rank, world, device = sd.init()
assert (rank, world) == (0, 1) and device.type == 'cuda' and dist.is_initialized()
assert str(dist.get_backend()) == 'gloo' and sd._DATA_GROUP is not None and sd._STAGED is False
assert sd.rccl_failure is None and not sd._ABANDONED and str(dist.get_backend(sd._DATA_GROUP)) == 'nccl'
sd.is_distributed = lambda: True                      # a world of one is "not distributed" for the product
assert sd.data_backend() == 'nccl'
x = torch.arange(27, dtype=torch.float64, device='cuda').reshape(3, 9)
out = sd.gather_rows(x, 3)
assert out.is_cuda and torch.equal(out, x)
assert sd.max_over_ranks(2.5) == 2.5 and sd.sum_over_ranks(1.5) == 1.5      # (host scalars: the gloo default group)
sd.barrier()
m = np.arange(30, dtype=np.float64).reshape(3, 10) / 7
assert np.array_equal(sd.broadcast_matrix(m), m)
got = sd.collect_rows(x.to(torch.float32), 3, dst=0)
assert got.dtype == np.float32 and np.array_equal(got, x.cpu().numpy().astype(np.float32))
sd.agree_or_raise(None)
# a first collective that raises: every rank (here: the one) falls back to host staging and says why
os.environ['SMART_DIST_PROBE_TIMEOUT'] = '2'
def boom(device, group):
    raise RuntimeError('NCCL error: unhandled cuda error')
import warnings
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter('always')
    assert sd.rccl_answers(device, group=sd._DATA_GROUP, probe=boom) is False
assert 'unhandled cuda error' in sd.rccl_failure and sd._ABANDONED and len(w) == 1
print('subgroup ok')
sd.finish(0)
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and 'subgroup ok' in r.stdout, r.stdout + r.stderr


def test_rccl_code_path_with_a_single_rank_group():
    """A box with one GPU cannot hold a two-rank RCCL group, but a one-rank group runs the very calls the multi-GPU
    path makes on device tensors (all_gather_into_tensor, all_reduce, broadcast, barrier with device_ids): the
    backend-specific half of smartpy_amd/distributed.py that the gloo tests on CPU cannot reach -- including the
    sharded launches of bench.py --config 4 (rows) and --config 5 (catchments) with their gather over RCCL."""
    import subprocess
    import sys
    code = r'''
import os, sys
sys.path.insert(0, %r)
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29631', RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
import torch, torch.distributed as dist
from smartpy_amd import distributed as sd
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1)
sd.is_distributed = lambda: True                      # a world of one is "not distributed" for the product
x = torch.arange(27, dtype=torch.float64, device='cuda').reshape(3, 9)
out = sd.gather_rows(x, 3)
assert out.is_cuda and torch.equal(out, x)
assert sd.max_over_ranks(2.5, x.device) == 2.5 and sd.sum_over_ranks(1.5, x.device) == 1.5
sd.barrier()
# the sample broadcast of MonteCarlo.run() and the two sharded launches of bench.py --config 4 / 5, over RCCL
import numpy as np
import bench
from smartpy_amd import engine
from smartpy_amd.sampling import latin_hypercube
from smartpy_amd.parameters import Parameters
m = np.arange(30, dtype=np.float64).reshape(3, 10) / 7
assert np.array_equal(sd.broadcast_matrix(m), m)
f = bench.synthetic_forcing(0, True)[0][:24 * 200]
p = latin_hypercube(500, Parameters().ranges, seed=3)
obs = np.abs(np.random.default_rng(0).normal(2, 1, 200))
rows = sd.ShardedEnsemble(torch.from_numpy(p).cuda(), f, bench.AREA, 3600.0, 24 * 30, 24, axis='samples', obs=obs,
                          gw_obs=0.12667, extra=bench.EXTRA, want_discharge=False)
got = rows.step()
one = engine.run_ensemble(p, f, bench.AREA, 3600.0, 24 * 30, 24, obs=obs, gw_obs=0.12667, extra=bench.EXTRA)
assert got.shape == (500, 9) and torch.equal(got[:, :8], one.objfn) and torch.equal(got[:, 8], one.gw)
rows.verify()
cats = sd.ShardedEnsemble(torch.from_numpy(p[:100]).cuda(), lambda c: bench.synthetic_forcing(c, True)[0][:24 * 200],
                          np.array([1e8, 2e8, 3e8]), 3600.0, 24 * 30, 24, axis='catchments', n_catchments=3,
                          obs=np.tile(obs, (3, 1)), gw_obs=0.12667, extra=bench.EXTRA, want_discharge=False)
got = cats.step()
assert got.shape == (3, 100, 9)
for c in range(3):
    one = engine.run_ensemble(p[:100], bench.synthetic_forcing(c, True)[0][:24 * 200], (c + 1) * 1e8, 3600.0, 24 * 30, 24,
                              obs=obs, gw_obs=0.12667, extra=bench.EXTRA)
    assert torch.equal(got[c, :, :8], one.objfn) and torch.equal(got[c, :, 8], one.gw)
dist.destroy_process_group()
print('rccl ok')
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and 'rccl ok' in r.stdout, r.stdout + r.stderr
