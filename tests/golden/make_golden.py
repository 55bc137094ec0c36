"""Generate the golden vectors under tests/golden/ by IMPORTING THE REFERENCE (build container only).

Run from the repo root:   PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

The reference (ThibHlln/smartpy v0.2.2, read-only at /root/reference) is pure Python; it is imported
here, never copied, and never travels to the GPU box.  What is committed are numbers only: inputs and
the outputs the reference produced for them, plus the values the reference's own tests/examples hold
(G1..G4).  spotpy / netCDF4 are not installed, so a stub `spotpy` module (parameter.List /
parameter.generate only) is injected to let the reference's montecarlo.LHS construct and sample; none
of spotpy's arithmetic is involved in any vector stored here.
"""
import importlib
import os
import shutil
import sys
import tempfile
import types
from datetime import datetime, timedelta

from collections import OrderedDict

import numpy as np

REF = '/root/reference'
OUT = os.path.dirname(os.path.abspath(__file__))

sys.dont_write_bytecode = True
sys.path.insert(0, REF)

# ---- stub spotpy so that smartpy.montecarlo imports (lhs.py:23-26, montecarlo.py:27-30) ---------------------
spotpy = types.ModuleType('spotpy')
spotpy.parameter = types.ModuleType('spotpy.parameter')


class _List(object):
    def __init__(self, name, values):
        self.name, self.values = name, values


spotpy.parameter.List = _List
spotpy.parameter.generate = lambda params: None
spotpy.algorithms = types.ModuleType('spotpy.algorithms')
spotpy.objectivefunctions = types.ModuleType('spotpy.objectivefunctions')
sys.modules['spotpy'] = spotpy
sys.modules['spotpy.parameter'] = spotpy.parameter

import smartpy  # noqa: E402  (the reference)
from smartpy import structure  # noqa: E402

assert not structure.smart_in_cpp, "the oracle must be the pure-Python path"

EXTRA = {'aar': 1200, 'r-o_ratio': 0.45, 'r-o_split': (0.10, 0.15, 0.15, 0.30, 0.30)}
AREA = 175.46 * 1e6
NAMES = ['T', 'C', 'H', 'D', 'S', 'Z', 'SK', 'FK', 'GK', 'RK']


def save(name, **arrays):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **arrays)
    print('%-28s %8.1f kB' % (name, os.path.getsize(path) / 1e3))


def make_model(root, delta_simu, start='01/01/2007 09:00:00', end='31/12/2016 09:00:00', warm_up=365,
               gauged=175.97 * 1e6):
    return smartpy.SMART(
        catchment='Catchment', catchment_area_m2=AREA,
        start=datetime.strptime(start, '%d/%m/%Y %H:%M:%S'), end=datetime.strptime(end, '%d/%m/%Y %H:%M:%S'),
        time_delta_simu=delta_simu, time_delta_save=timedelta(days=1), warm_up_days=warm_up,
        in_format='csv', out_format='csv', root=root, gauged_area_m2=gauged)


class Capture(object):
    """Record what structure.run_all_steps returns (run() drops the final state, structure.py:143-146)."""

    def __init__(self):
        self.calls = []
        self.orig = structure.run_all_steps

    def __enter__(self):
        def wrapped(*a):
            r = self.orig(*a)
            self.calls.append((a, r))
            return r
        structure.run_all_steps = wrapped
        return self

    def __exit__(self, *exc):
        structure.run_all_steps = self.orig


def storage_table(area, dt, L, rain, peva, p, initial):
    """The (L+1) x 19 table of structure.py:177-187, rebuilt by calling the reference's run_one_step."""
    tab = np.zeros((L + 1, 19))
    tab[0] = initial
    for i in range(1, L + 1):
        tab[i] = structure.run_one_step(area, dt, rain[i - 1], peva[i - 1], *p, *tab[i - 1, 7:])
    return tab


def resampling_vectors(root):
    """KAT-10: the input pipeline on axes that need re-aggregation (timeframe.py:158-233): 6-hourly steps on the
    daily 09:00 data (resolution 6 h) and the same shifted to 12:00 (resolution 3 h, two portions per step)."""
    out = {}
    for tag, hh in (('aligned', 9), ('shifted', 12)):
        sm = smartpy.SMART('Catchment', AREA, datetime(2007, 1, 1, hh), datetime(2007, 12, 31, hh),
                           timedelta(hours=6), timedelta(days=1), 0, 'csv', 'csv', root, gauged_area_m2=175.97 * 1e6)
        out['rain_' + tag], out['peva_' + tag], out['flow_' + tag] = sm.nd_rain, sm.nd_peva, sm.nd_flow
    save('kat10_resampling.npz', **out)


def dict_helper_vectors(root):
    """The reference's dictionary-keyed helpers called directly (timeframe.py:167-309, inout.py:35-78): results as
    (seconds since 1970, value) arrays."""
    from smartpy import timeframe as tf, inout as io
    epoch = datetime(1970, 1, 1)

    def unpack(d):
        keys = list(d)
        return (np.array([(k - epoch).total_seconds() for k in keys]), np.array([d[k] for k in keys], dtype=np.float64))

    rng = np.random.default_rng(11)
    out = {}
    # regular cumulative data: 40 daily values at 09:00 -> 3-hourly simulation stamps from 12:00 (common grid 3 h)
    start = datetime(2010, 3, 1, 9)
    daily = {start + timedelta(days=k): float(v) for k, v in enumerate(rng.gamma(0.7, 4.0, 40))}
    out['reg_in_t'], out['reg_in_v'] = unpack(daily)
    res = tf.get_required_resolution(start, datetime(2010, 3, 2, 12), timedelta(days=1), timedelta(hours=3))
    out['reg_res_sec'] = res.total_seconds()
    inc = tf.increase_time_resolution_of_regular_cumulative_data(daily, start, start + timedelta(days=39),
                                                                 timedelta(days=1), res)
    out['reg_inc_t'], out['reg_inc_v'] = unpack(inc)
    dec = tf.decrease_time_resolution_of_regular_cumulative_data(inc, datetime(2010, 3, 2, 12), datetime(2010, 4, 5, 12),
                                                                 timedelta(hours=6), res)
    out['reg_dec_t'], out['reg_dec_v'] = unpack(dec)
    resc = tf.rescale_time_resolution_of_regular_cumulative_data(daily, start, start + timedelta(days=39),
                                                                 timedelta(days=1), res, datetime(2010, 3, 2, 12),
                                                                 datetime(2010, 4, 5, 12), timedelta(hours=3))
    out['reg_resc_t'], out['reg_resc_v'] = unpack(resc)
    # irregular mean data: daily means with a 3-day gap and a missing first neighbour
    days = [0, 1, 2, 3, 7, 8, 9, 10, 11, 13, 14]
    flows = OrderedDict((datetime(2010, 3, 1) + timedelta(days=k), float(v)) for k, v in zip(days, rng.random(len(days)) * 5))
    out['irr_in_t'], out['irr_in_v'] = unpack(flows)
    inc = tf.increase_time_resolution_of_irregular_mean_data(flows, timedelta(days=1), timedelta(hours=1))
    out['irr_inc_t'], out['irr_inc_v'] = unpack(inc)
    resc = tf.rescale_time_resolution_of_irregular_mean_data(flows, datetime(2010, 3, 1, 9), datetime(2010, 3, 15, 9),
                                                             timedelta(days=1), timedelta(hours=1))
    out['irr_resc_t'], out['irr_resc_v'] = unpack(resc)
    # the readers on the shipped example files
    loc = os.path.join(root, 'in', 'Catchment', 'Catchment')
    rain = io.get_dict_rain_series_simu(loc + '.rain', 'csv', datetime(2007, 1, 1, 12), datetime(2007, 2, 1, 9),
                                        timedelta(hours=3))
    out['rain_3h_t'], out['rain_3h_v'] = unpack(rain)
    flow = io.get_dict_discharge_series(loc + '.flow', 'csv', datetime(2007, 1, 1, 9), datetime(2007, 3, 1, 9),
                                        175.46e6, 175.97e6)
    out['flow_t'], out['flow_v'] = unpack(flow)
    raw, first, last, step = io.read_peva_file(loc + '.peva', 'csv')
    out['peva_raw_n'] = len(raw)
    out['peva_raw_first'] = (first - epoch).total_seconds()
    out['peva_raw_last'] = (last - epoch).total_seconds()
    out['peva_raw_step'] = step.total_seconds()
    out['flow_raw_n'] = len(io.read_flow_file(loc + '.flow', 'csv'))
    save('kat11_dict_helpers.npz', **out)


def api_surface():
    """Public names of the reference's modules with the parameter names of every function / method: the inventory
    tests/test_host_logic.py::test_public_surface_has_every_name_of_the_reference checks the package against."""
    import inspect
    import json
    mods = ['smartpy', 'smartpy.smart', 'smartpy.structure', 'smartpy.inout', 'smartpy.timeframe', 'smartpy.parameters',
            'smartpy.objfunctions', 'smartpy.montecarlo', 'smartpy.montecarlo.montecarlo', 'smartpy.montecarlo.lhs',
            'smartpy.montecarlo.glue', 'smartpy.montecarlo.best', 'smartpy.montecarlo.total', 'smartpy.version']
    out = {}
    for name in mods:
        mod = importlib.import_module(name)
        entry = {}
        for n, o in vars(mod).items():
            if n.startswith('_'):
                continue
            if inspect.isfunction(o) and o.__module__ == mod.__name__:
                entry[n] = list(inspect.signature(o).parameters)
            elif inspect.isclass(o) and o.__module__ == mod.__name__:
                entry[n] = {k: (list(inspect.signature(v).parameters) if inspect.isfunction(v) else None)
                            for k, v in vars(o).items() if not k.startswith('__') and callable(v)}
        out[name] = entry
    with open(os.path.join(OUT, 'api_surface.json'), 'w') as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
    print('api_surface.json', sum(len(v) for v in out.values()), 'names')


def montecarlo_vectors():
    """KAT-12: the conditioning rules of the second-stage classes, straight from the reference's static methods
    GLUE._get_behavioural_sets (glue.py:222-289) and Best._get_best_sets (best.py:221-287) on a float32 matrix with
    ties, values that sit exactly on thresholds and a NaN row.  KAT-13: the bytes of a sampling database written by
    the reference's own MonteCarlo._init_db + save (montecarlo.py:122-127, 211-231), with and without the simulated
    series, and what its _get_sampled_sets_from_file (:233-262) reads -- from its own file and from a file written
    by this repository's database.SamplingCsv."""
    import json
    from smartpy.montecarlo.glue import GLUE
    from smartpy.montecarlo.best import Best
    from smartpy.montecarlo.montecarlo import MonteCarlo

    rng = np.random.default_rng(1212)
    n = 48
    params = rng.uniform(0.0, 100.0, (n, 10)).astype(np.float32)
    params[:, 0] = np.arange(n, dtype=np.float32)                 # column 0 names the row: outputs -> indices
    f32 = np.float32
    fns = np.empty((n, 8), dtype=np.float32)
    fns[:, 0] = rng.choice(np.array([0.1, 0.5, 0.7, np.nextafter(f32(0.7), f32(1)), 0.9], dtype=np.float32), n)  # NSE
    fns[:, 1] = rng.uniform(-0.5, 1.0, n)                         # KGE, continuous
    fns[:, 2] = rng.uniform(0.0, 1.0, n)
    fns[:, 3] = rng.uniform(0.5, 1.5, n)
    fns[:, 4] = np.round(rng.uniform(0.5, 1.5, n), 1)             # KGEb on a 0.1 grid: ties
    fns[:, 5] = rng.uniform(-40.0, 40.0, n)                       # PBias
    fns[:, 6] = rng.gamma(2.0, 2.0, n)                            # RMSE
    fns[:, 7] = rng.integers(0, 2, n)                             # GW flag: two values only
    fns[7, :] = np.nan                                            # a row no comparison accepts
    fns[11, 5] = f32(10.0)                                        # exactly on the thresholds used below
    fns[12, 5] = f32(-10.0)

    # both matrices as a sampling database hands them over: float32 -> '%.6e' -> float32 (montecarlo.py:225-231, :262),
    # so that the same vectors also serve the constructors that take their sample from a finished run
    def stored(m):
        return np.array([[f32('%.6e' % v) for v in row] for row in m], dtype=np.float32)
    fns[fns[:, 0] == np.nextafter(f32(0.7), f32(1)), 0] = f32(0.700001)      # a neighbour of 0.7 that survives the text
    params, fns = stored(params), stored(fns)
    assert np.array_equal(stored(params), params) and np.array_equal(stored(fns), fns, equal_nan=True)

    def rows_of(out):
        return [int(v) for v in out[:, 0]]

    glue_cases = []
    for cols, kinds, vals in [
        ([0], ['min'], [(0.7,)]),
        ([0], ['equal'], [(0.7,)]),                               # float32(0.7) == 0.7 under the array's dtype
        ([0], ['max'], [(0.5,)]),
        ([5], ['inside'], [(-10.0, 10.0)]),
        ([5], ['outside'], [(-10.0, 10.0)]),                      # glue.py:277 as written: no value passes
        ([0, 5, 7], ['min', 'inside', 'equal'], [(0.5,), (-25.0, 25.0), (1.0,)]),
        ([1, 6], ['min', 'max'], [(0.3,), (3.5,)]),
        ([1], ['min'], [(5.0,)]),                                 # nobody: an empty [0, 10] matrix
        ([7], ['equal'], [(0.0,)]),
    ]:
        out = GLUE._get_behavioural_sets(params, fns[:, cols], vals, kinds)
        glue_cases.append({'columns': cols, 'kinds': kinds, 'values': [list(v) for v in vals], 'rows': rows_of(out),
                           'shape': list(out.shape), 'dtype': str(out.dtype)})

    best_cases = []
    for cols, kinds, vals, target, nb in [
        ([], [], [], 1, 5),                                       # continuous target: no ties anywhere
        ([], [], [], 1, 48),                                      # all of them (the NaN row sorts last)
        ([5], ['inside'], [(-25.0, 25.0)], 1, 7),
        ([7], ['equal'], [(1.0,)], 6, 4),                         # RMSE as target: the LARGEST four, as written
        ([], [], [], 0, 6),                                       # NSE on five levels: ties inside the selection
        ([], [], [], 7, 10),                                      # GW flag as target: ties across the cut
        ([0], ['min'], [(0.7,)], 4, 3),                           # 0.1 grid
    ]:
        out = Best._get_best_sets(params, fns[:, cols], vals, kinds, fns[:, [target]], nb)
        rows = rows_of(out)
        # is the answer independent of how a sort orders equal keys?  (the reference calls numpy's default argsort,
        # an unstable introsort / SIMD sort whose tie order is an implementation detail of the numpy build and CPU)
        mask = np.ones(n, dtype=bool)
        for c, k, v in zip(cols, kinds, vals):
            col = fns[:, c]
            mask &= {'min': lambda: col >= v[0], 'max': lambda: col <= v[0], 'equal': lambda: col == v[0],
                     'inside': lambda: (col >= v[0]) & (col <= v[1])}[k]()
        keys = fns[mask, target]
        srt = np.sort(keys)                                       # NaN last, like argsort
        tail = srt[-nb:]
        ambiguous = bool(len(np.unique(tail[~np.isnan(tail)])) + int(np.isnan(tail).sum()) < len(tail)
                         or (len(srt) > nb and (srt[-nb - 1] == tail[0] or
                                                (np.isnan(srt[-nb - 1]) and np.isnan(tail[0])))))
        best_cases.append({'columns': cols, 'kinds': kinds, 'values': [list(v) for v in vals], 'target': target,
                           'nb_best': nb, 'rows': rows, 'keys': [repr(float(fns[r, target])) for r in rows],
                           'ambiguous': ambiguous,
                           'rows_if_stable': [int(v) for v in
                                              params[mask][np.argsort(keys, kind='stable')][-nb:][:, 0]]})

    errors = []
    for label, call in [
        ('glue_equal_two', lambda: GLUE._get_behavioural_sets(params, fns[:, [0]], [(0.1, 0.2)], ['equal'])),
        ('glue_inside_order', lambda: GLUE._get_behavioural_sets(params, fns[:, [0]], [(0.9, 0.2)], ['inside'])),
        ('glue_kind', lambda: GLUE._get_behavioural_sets(params, fns[:, [0]], [(0.9,)], ['above'])),
        ('glue_dims', lambda: GLUE._get_behavioural_sets(params, fns[:, [0, 1]], [(0.9,)], ['min'])),
        ('glue_1d', lambda: GLUE._get_behavioural_sets(params, fns[:, 0], [(0.9,)], ['min'])),
        ('best_too_many', lambda: Best._get_best_sets(params, fns[:, []], [], [], fns[:, [1]], 49)),
        ('best_too_many_constrained', lambda: Best._get_best_sets(params, fns[:, [0]], [(0.9,)], ['min'],
                                                                  fns[:, [1]], 40)),
        ('best_sizes', lambda: Best._get_best_sets(params, fns[:, []], [], [], fns[:20, [1]], 4)),
    ]:
        try:
            call()
            errors.append({'case': label, 'message': None})
        except Exception as e:
            errors.append({'case': label, 'message': str(e)})

    save('kat12_selection.npz', params=params, obj_fns=fns)
    with open(os.path.join(OUT, 'kat12_selection.json'), 'w') as fh:
        json.dump({'numpy': np.__version__, 'glue': glue_cases, 'best': best_cases, 'errors': errors}, fh, indent=1)

    # ---- KAT-13 ---------------------------------------------------------------------------------------------
    names_obj = ['NSE', 'KGE', 'KGEc', 'KGEa', 'KGEb', 'PBias', 'RMSE', 'GW']
    stamps = [datetime(2008, 1, 1, 9) + timedelta(days=k) for k in range(6)]
    rng = np.random.default_rng(1313)
    obj = rng.uniform(-1.0, 1.0, (5, 8))
    obj[0] = [1.0 / 3.0, 1e-10, -2.5e8, 0.0, 123456.789, 9.9999995e-1, 5e-324, 1.0]    # ties of '%.6e', tiny, huge
    obj[1, :3] = [np.nan, np.inf, -np.inf]
    obj[2, 0] = 0.30000001192092896                                                    # a float32 exactly
    par = rng.uniform(0.0, 1000.0, (5, 10))
    par[3] = [1.0, 0.0, 0.25, 0.5, 1e-3, 105.25734595830215, 46.81961454361724, 315.5490902162102,
              1066.7332319333473, 10.640277777777778]
    sim = rng.gamma(2.0, 3.0, (5, 6))
    out13 = {'obj_fns': obj, 'params': par, 'sims': sim,
             'stamps': np.array([s.strftime('%Y-%m-%d %H:%M:%S') for s in stamps])}
    tmp = tempfile.mkdtemp(prefix='smart_db_')

    class _Model(object):
        flow = OrderedDict((s, 0.0) for s in stamps)

    for tag, save_sim in (('nosim', False), ('sim', True)):
        mc = MonteCarlo.__new__(MonteCarlo)
        mc.out_format, mc.save_sim, mc.model = 'csv', save_sim, _Model()
        mc.obj_fn_names, mc.param_names = names_obj, NAMES
        mc.db_file = os.path.join(tmp, 'ref_%s.SMART.lhs' % tag)
        mc._init_db()
        for k in range(5):
            mc.save(obj[k].tolist(), par[k], [sim[k]])
        mc.database.close()
        with open(mc.db_file, 'rb') as fh:
            out13['bytes_' + tag] = np.frombuffer(fh.read(), dtype=np.uint8)
        p_ref, o_ref = mc._get_sampled_sets_from_file(mc.db_file, NAMES, names_obj, False)
        out13['ref_reads_ref_params_' + tag], out13['ref_reads_ref_objfns_' + tag] = p_ref, o_ref
        # ... and from a file written by this repository's writer (host-only code of the library: no GPU involved)
        sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))
        from smartpy_amd.montecarlo.database import SamplingCsv
        ours = SamplingCsv(os.path.join(tmp, 'ours_%s.SMART.lhs' % tag), names_obj, NAMES)
        ours.create(5, stamps if save_sim else None)
        ours.write_table(obj, par, sim if save_sim else None)
        ours.close()
        p_o, o_o = mc._get_sampled_sets_from_file(ours.path, NAMES, names_obj, False)
        out13['ref_reads_ours_params_' + tag], out13['ref_reads_ours_objfns_' + tag] = p_o, o_o
    shutil.rmtree(tmp)
    save('kat13_database.npz', **out13)


def main():
    if '--only-montecarlo' in sys.argv:
        montecarlo_vectors()
        return
    scratch = tempfile.mkdtemp(prefix='smart_golden_')
    shutil.copytree(os.path.join(REF, 'tests', 'data'), os.path.join(scratch, 'data'))
    root = os.path.join(scratch, 'data') + os.sep
    for dirpath, _, files in os.walk(scratch):
        os.chmod(dirpath, 0o755)
        for f in files:
            os.chmod(os.path.join(dirpath, f), 0o644)

    resampling_vectors(root)
    dict_helper_vectors(root)
    api_surface()
    montecarlo_vectors()
    if '--only-resampling' in sys.argv:
        shutil.rmtree(scratch)
        return

    # ---------------------------------------------------------------------------------------------------
    # forcing + observations of the shipped example, as the reference's input pipeline delivers them
    # ---------------------------------------------------------------------------------------------------
    sm_h = make_model(root, timedelta(hours=1))
    sm_d = make_model(root, timedelta(days=1))
    sm_h.extra = EXTRA
    sm_d.extra = EXTRA
    sm_h.parameters.set_parameters_with_file(sm_h.in_f + 'Catchment.parameters')
    p_ex = np.array([sm_h.parameters.values[n] for n in NAMES])
    rain_d, peva_d = sm_d.nd_rain.copy(), sm_d.nd_peva.copy()
    # the hourly series is the daily one split equally over 24 steps (timeframe.py:167-233)
    assert np.array_equal(sm_h.nd_rain, np.repeat(rain_d / 24, 24))
    assert np.array_equal(sm_h.nd_peva, np.repeat(peva_d / 24, 24))
    assert np.array_equal(sm_h.nd_flow, sm_d.nd_flow, equal_nan=True)
    save('forcing_example.npz', rain_daily=rain_d, peva_daily=peva_d, flow_obs=sm_h.nd_flow, area=AREA,
         params=p_ex, extra=np.array([EXTRA['aar'], EXTRA['r-o_ratio']] + list(EXTRA['r-o_split'])))

    # ---------------------------------------------------------------------------------------------------
    # KAT-1 hourly, W=365 d, extra, shipped parameters, summary and raw (structure.py:189-195)
    # ---------------------------------------------------------------------------------------------------
    with Capture() as cap:
        dis_s, gw_s = sm_h.simulate(sm_h.parameters.values, report='summary')
    (a_wu, r_wu), (a_run, r_run) = cap.calls
    init_wu, init_run, final_s = a_wu[6].copy(), a_run[6].copy(), r_run[2].copy()
    with Capture() as cap:
        dis_r, gw_r = sm_h.simulate(sm_h.parameters.values, report='raw')
    final_r = cap.calls[1][1][2].copy()
    first = storage_table(AREA, 3600.0, 48, sm_h.nd_rain, sm_h.nd_peva, p_ex, init_run)
    save('kat1_hourly.npz', discharge_summary=dis_s, gw_summary=gw_s, final_summary=final_s,
         discharge_raw=dis_r, gw_raw=gw_r, final_raw=final_r, initial_warmup=init_wu, initial_run=init_run,
         first48=first, n_steps=len(sm_h.nd_rain), n_warm=365 * 24, gap=24)

    # G2: the committed example output must be what the reference regenerates (examples/out/.../*.mod.flow)
    g2 = np.loadtxt(os.path.join(REF, 'examples/out/ExampleDaily/ExampleDaily.mod.flow'), delimiter=',',
                    skiprows=1, usecols=1)
    assert [('%e' % v) for v in dis_s] == [('%e' % v) for v in g2]
    g3 = np.genfromtxt(os.path.join(REF, 'examples/out/ExampleDaily/ExampleDaily.obs.flow'), delimiter=',',
                       skip_header=1, usecols=1)
    save('g2_g3_example_flows.npz', mod_flow=g2, obs_flow=g3)

    # G1: the 91 dated values hard-coded in the reference's own test (tests/test_run_daily_to_hourly.py:30-122)
    sys.path.insert(0, os.path.join(REF, 'tests'))
    cwd = os.getcwd()
    os.chdir(scratch)  # the test's setUp builds SMART(root="data/")
    try:
        tmod = importlib.import_module('test_run_daily_to_hourly')
        tc = tmod.TestRunDaily2Hourly('test_compare_discharge_series')
        tc.setUp()
        idx = np.array([tc.sm.timeseries_report[1:].index(dt) for dt in tc.expected_outcome])
        val = np.array([tc.expected_outcome[dt] for dt in tc.expected_outcome])
    finally:
        os.chdir(cwd)
    assert all(('%.6e' % dis_s[i]) == ('%.6e' % v) for i, v in zip(idx, val))
    save('g1_reference_test.npz', report_index=idx, expected=val)

    # ---------------------------------------------------------------------------------------------------
    # KAT-2 daily (gap = 1), KAT-3 no warm-up with / without extra
    # ---------------------------------------------------------------------------------------------------
    with Capture() as cap:
        dis_d, gw_d = sm_d.simulate(sm_h.parameters.values)
    save('kat2_daily.npz', discharge=dis_d, gw=gw_d, final=cap.calls[1][1][2], n_steps=len(rain_d), n_warm=365,
         gap=1)

    out3 = {}
    for tag, extra in (('extra', EXTRA), ('noextra', None)):
        for res, delta in (('daily', timedelta(days=1)), ('hourly', timedelta(hours=1))):
            sm = make_model(root, delta, end='31/12/2008 09:00:00', warm_up=0)
            sm.extra = extra
            with Capture() as cap:
                d, g = sm.simulate(sm_h.parameters.values)
            out3['discharge_%s_%s' % (res, tag)] = d
            out3['gw_%s_%s' % (res, tag)] = g
            out3['final_%s_%s' % (res, tag)] = cap.calls[0][1][2]
            out3['initial_%s_%s' % (res, tag)] = cap.calls[0][0][6]
    save('kat3_nowarm.npz', n_days=731, **out3)

    # ---------------------------------------------------------------------------------------------------
    # KAT-7 LHS sampler (lhs.py:133-167) through the reference's own class, and KAT-4 batch on 32 rows
    # ---------------------------------------------------------------------------------------------------
    from smartpy.montecarlo.lhs import LHS
    lhs = LHS.__new__(LHS)            # only _get_params_from_lh is exercised: needs model.parameters + names
    lhs.model = sm_h
    lhs.param_names = sm_h.parameters.names
    out7 = {}
    for seed in (0, 42, 2718):
        for n in (5, 64, 1000):
            np.random.seed(seed)
            out7['seed%d_n%d' % (seed, n)] = lhs._get_params_from_lh(n)
    save('kat7_lhs.npz', **out7)

    np.random.seed(2718)
    p32 = lhs._get_params_from_lh(32)
    sm2 = make_model(root, timedelta(hours=1), end='31/12/2008 09:00:00', warm_up=365)
    sm2.extra = EXTRA
    dis_h2, gw_h2, dis_d10, gw_d10 = [], [], [], []
    for row in p32:
        par = dict(zip(NAMES, row))
        d, g = sm2.simulate(par)
        dis_h2.append(d.copy()); gw_h2.append(g)
        d, g = sm_d.simulate(par)
        dis_d10.append(d.copy()); gw_d10.append(g)
    save('kat4_batch.npz', params=p32, discharge_hourly_2yr=np.array(dis_h2), gw_hourly_2yr=np.array(gw_h2),
         discharge_daily_10yr=np.array(dis_d10), gw_daily_10yr=np.array(gw_d10), n_days_hourly=731)

    # ---------------------------------------------------------------------------------------------------
    # KAT-5 river "95 % rule" and zero clamps: daily steps with RK*3600 < dt (structure.py:429-450,492-496)
    # ---------------------------------------------------------------------------------------------------
    rows, tabs = [], []
    for rk in (1.0, 2.0, 6.0, 12.0, 23.9):
        for skv in (1.0, 5.0):
            p = p_ex.copy()
            p[9], p[6] = rk, skv
            init = np.zeros(19)
            init[7:12] = [3e4, 2e4, 5e5, 4e6, 6e6]
            init[12:18] = (p[5] / 12) / 1000 * AREA
            init[18] = 1e5
            tabs.append(storage_table(AREA, 86400.0, 400, rain_d, peva_d, p, init))
            rows.append(p)
    save('kat5_river.npz', params=np.array(rows), tables=np.array(tabs), n_steps=400)

    # raw report with a length that is not a multiple of the gap (structure.py:192-195)
    init = first[0]
    d_raw, g_raw, f_raw = structure.run_all_steps(AREA, 3600.0, 1000, sm_h.nd_rain, sm_h.nd_peva, p_ex, init, 2, 24)
    d_big, g_big, f_big = structure.run_all_steps(AREA, 86400.0, 3653, rain_d, peva_d, p_ex, init, 2, 1)
    save('kat9_raw_ragged.npz', initial=init, discharge=d_raw, gw=g_raw, final=f_raw, n_steps=1000, gap=24,
         discharge_gap1=d_big, gw_gap1=g_big)

    # ---------------------------------------------------------------------------------------------------
    # KAT-6 single steps: every branch of structure.py:267-503, hand-made + random
    # ---------------------------------------------------------------------------------------------------
    rng = np.random.default_rng(20261002)
    lo = np.array([0.9, 0.0, 0.0, 0.0, 0.0, 15.0, 1.0, 48.0, 1200.0, 1.0])
    hi = np.array([1.1, 1.0, 0.3, 1.0, 0.013, 150.0, 240.0, 1440.0, 4800.0, 96.0])
    cases = []

    def add(area, dt, rain, peva, p, st):
        cases.append((area, dt, rain, peva, np.array(p, float), np.array(st, float)))

    def soil(p, frac):
        return list(np.asarray(frac, float) * (p[5] / 6) / 1000 * AREA)

    res5 = [3e4, 2e4, 5e5, 4e6, 6e6]
    for dt in (3600.0, 86400.0):
        p = p_ex.copy()
        add(AREA, dt, 0.0, 0.0, p, res5 + soil(p, [0.5] * 6) + [1e5])              # ex == 0 exactly, wet path
        add(AREA, dt, 1.5, 1.5, p, res5 + soil(p, [0.5] * 6) + [1e5])              # rain*T == peva (T = 1)
        add(AREA, dt, 30.0, 0.1, p, res5 + soil(p, [1.0] * 6) + [1e5])             # saturated soil
        add(AREA, dt, 30.0, 0.1, p, res5 + soil(p, [0.99, 1, 1, 0.5, 0, 0]) + [1e5])
        add(AREA, dt, 200.0, 0.0, p, res5 + soil(p, [0.0] * 6) + [1e5])            # fills every layer, excess left
        for c in (0.0, 0.5, 1.0):                                                  # empty soil, big deficit
            q = p.copy(); q[1] = c
            add(AREA, dt, 0.0, 5.0, q, res5 + soil(q, [0.0] * 6) + [1e5])
            add(AREA, dt, 0.1, 9.0, q, res5 + soil(q, [0.01, 0.02, 0.0, 0.5, 0.0, 1.0]) + [1e5])
        for k, v in ((4, 0.0), (2, 0.0), (3, 0.0), (3, 1.0)):                      # S = 0, H = 0, D in {0, 1}
            q = p.copy(); q[k] = v
            add(AREA, dt, 12.0, 0.3, q, res5 + soil(q, [0.9, 0.8, 1.0, 0.2, 0.0, 0.6]) + [1e5])
        add(AREA, dt, 0.0, 0.0, p, [0.0] * 5 + soil(p, [0.0] * 6) + [0.0])         # everything empty (gw = 0/0)
        q = p.copy(); q[9] = 1.0; q[6] = 1.0                                       # river rule + reservoir clamps
        add(AREA, dt, 0.0, 2.0, q, res5 + soil(q, [0.5] * 6) + [1e5])
        add(AREA, dt, 0.0, 2.0, q, res5 + soil(q, [0.5] * 6) + [0.0])
    for _ in range(600):
        p = lo + rng.random(10) * (hi - lo)
        dt = float(rng.choice([3600.0, 86400.0, 900.0]))
        area = float(np.exp(rng.uniform(np.log(20e6), np.log(2000e6))))
        frac = rng.random(6) * rng.choice([0.0, 1.0, 1.0], 6)
        frac[rng.random(6) < 0.2] = 1.0
        st = list(rng.random(5) * [1e5, 1e5, 1e6, 1e7, 1e7] * rng.choice([0.0, 1.0, 1.0], 5)) + \
            list(frac * (p[5] / 6) / 1000 * area) + [float(rng.random() * 1e6 * rng.choice([0.0, 1.0]))]
        scale = dt / 86400.0
        rain = float(rng.choice([0.0, 1.0, 1.0]) * rng.gamma(0.7, 4.57) * scale * rng.choice([1.0, 1.0, 20.0]))
        peva = float(rng.choice([0.0, 1.0, 1.0]) * rng.random() * 3.0 * scale)
        add(area, dt, rain, peva, p, st)
    outs = np.array([structure.run_one_step(a, dt, r, e, *p, *st) for (a, dt, r, e, p, st) in cases])
    save('kat6_steps.npz', area=np.array([c[0] for c in cases]), dt=np.array([c[1] for c in cases]),
         rain=np.array([c[2] for c in cases]), peva=np.array([c[3] for c in cases]),
         params=np.array([c[4] for c in cases]), states=np.array([c[5] for c in cases]), out=outs)

    # ---------------------------------------------------------------------------------------------------
    # G4: the committed sampling database of the example (float32 text), examples/out/.../*.SMART.lhs
    # ---------------------------------------------------------------------------------------------------
    with open(os.path.join(REF, 'examples/out/ExampleDaily/ExampleDaily.SMART.lhs')) as f:
        header = f.readline().strip().split(',')
    g4 = np.loadtxt(os.path.join(REF, 'examples/out/ExampleDaily/ExampleDaily.SMART.lhs'), delimiter=',', skiprows=1)
    cols = {h: g4[:, i] for i, h in enumerate(header)}
    # the run that produced it: ExampleDaily.sttngs (hourly simu, daily report, W = 365) + the notebook's extra
    dis4, gw4 = [], []
    for r in range(g4.shape[0]):
        par = {n: float(cols[n][r]) for n in NAMES}
        d, g = sm_h.simulate(par)
        dis4.append(d.copy()); gw4.append(g)
    save('g4_example_lhs.npz', params=np.array([cols[n] for n in NAMES]).T,
         objfns=np.array([cols[n] for n in ['NSE', 'KGE', 'KGEc', 'KGEa', 'KGEb', 'PBias', 'RMSE', 'GW']]).T,
         discharge=np.array(dis4), gw=np.array(gw4), gw_constraint=0.12667)

    shutil.rmtree(scratch)


if __name__ == '__main__':
    main()
