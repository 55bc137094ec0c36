"""The oracle (oracle/) against the reference's golden vectors.  CPU only.

Bit equality is asserted where the oracle runs in its reference-exact configuration
(pow through libm, numpy's pairwise reduction order): the vectors were produced by the reference
itself in this container (tests/golden/make_golden.py).
"""
import numpy as np
import pytest

from conftest import load_golden
from oracle import smart_oracle as so
from oracle import objfn_oracle, lhs_oracle


def rel(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    d = np.abs(a - b)
    m = np.maximum(np.abs(a), np.abs(b))
    with np.errstate(invalid='ignore', divide='ignore'):
        r = np.where(m > 0, d / m, 0.0)
    return float(np.max(r)) if r.size else 0.0


def test_kat6_single_steps_bit_exact():
    """Every branch of run_one_step (structure.py:200-503): 630 hand-made and random vectors."""
    g = load_golden('kat6_steps.npz')
    n = len(g['area'])
    assert n > 600
    wet = dry = river_rule = clamp = 0
    for k in range(n):
        out = so.one_step(g['area'][k], g['dt'][k], g['rain'][k], g['peva'][k], g['params'][k], g['states'][k])
        assert np.array_equal(out, g['out'][k], equal_nan=True), k
        ex = g['rain'][k] * g['params'][k][0] - g['peva'][k]
        wet += ex >= 0
        dry += ex < 0
        rk = g['params'][k][9] * 3600
        river_rule += g['states'][k][11] * (1 - g['dt'][k] / rk) + out[1:6].sum() * g['dt'][k] < 0
        clamp += np.any((g['out'][k][7:12] == 0) & (g['states'][k][0:5] > 0))
    assert wet > 100 and dry > 100 and river_rule > 5 and clamp > 5    # the vectors reach every branch


def test_kat6_pow_by_multiplication_is_rounding_level():
    g = load_golden('kat6_steps.npz')
    worst = 0.0
    for k in range(len(g['area'])):
        out = so.one_step(g['area'][k], g['dt'][k], g['rain'][k], g['peva'][k], g['params'][k], g['states'][k],
                          pow_mode=so.POW_MUL)
        worst = max(worst, rel(out, g['out'][k]))
    assert worst < 1e-12


def test_kat1_hourly_summary_and_raw_bit_exact(example):
    g = load_golden('kat1_hourly.npz')
    L, W, gap = int(g['n_steps']), int(g['n_warm']), int(g['gap'])
    assert (L, W, gap) == (87672, 8760, 24)
    init = so.initial(example['area'], example['params'], example['extra'])
    assert np.array_equal(init, g['initial_warmup'])
    for tag, rtype in (('summary', so.REPORT_SUMMARY), ('raw', so.REPORT_RAW)):
        dis, gw, fin = so.run(example['area'], 3600.0, L, W, example['rain_hourly'], example['peva_hourly'],
                              example['params'], example['extra'], rtype, gap)
        assert np.array_equal(dis, g['discharge_' + tag])
        assert gw == float(g['gw_' + tag])
        assert np.array_equal(fin, g['final_' + tag])
    # per-step table of the first 48 steps after warm-up
    _, _, _, sto = so.all_steps(example['area'], 3600.0, 48, example['rain_hourly'], example['peva_hourly'],
                                example['params'], g['initial_run'], so.REPORT_SUMMARY, 24, want_storage=True)
    assert np.array_equal(sto, g['first48'])


def test_g1_g2_reference_own_goldens(example):
    """G1: tests/test_run_daily_to_hourly.py:30-122 (91 values, '%.6e'); G2: examples/out/.../ExampleDaily.mod.flow."""
    dis, gw, _ = so.run(example['area'], 3600.0, 87672, 8760, example['rain_hourly'], example['peva_hourly'],
                        example['params'], example['extra'], so.REPORT_SUMMARY, 24)
    g1 = load_golden('g1_reference_test.npz')
    assert len(g1['expected']) == 91
    for i, v in zip(g1['report_index'], g1['expected']):
        assert '%.6e' % dis[i] == '%.6e' % v
    g2 = load_golden('g2_g3_example_flows.npz')
    assert ['%e' % v for v in dis] == ['%e' % v for v in g2['mod_flow']]
    # G3 pins the observation series used by the objective functions
    obs = example['flow_obs']
    assert np.array_equal(np.isnan(obs), np.isnan(g2['obs_flow']))
    ok = ~np.isnan(obs)
    assert ['%e' % v for v in obs[ok]] == ['%e' % v for v in g2['obs_flow'][ok]]
    assert abs(gw - 0.0529870) < 1e-6


def test_kat2_daily_bit_exact(example):
    g = load_golden('kat2_daily.npz')
    dis, gw, fin = so.run(example['area'], 86400.0, int(g['n_steps']), int(g['n_warm']), example['rain_daily'],
                          example['peva_daily'], example['params'], example['extra'], so.REPORT_SUMMARY, 1)
    assert np.array_equal(dis, g['discharge']) and gw == float(g['gw']) and np.array_equal(fin, g['final'])


@pytest.mark.parametrize('tag', ['extra', 'noextra'])
@pytest.mark.parametrize('res', ['daily', 'hourly'])
def test_kat3_no_warmup_initial_conditions(example, tag, res):
    g = load_golden('kat3_nowarm.npz')
    nd = int(g['n_days'])
    extra = example['extra'] if tag == 'extra' else None
    if res == 'daily':
        args = (86400.0, nd, 0, example['rain_daily'], example['peva_daily'])
        gap = 1
    else:
        args = (3600.0, nd * 24, 0, example['rain_hourly'], example['peva_hourly'])
        gap = 24
    init = so.initial(example['area'], example['params'], extra)
    assert np.array_equal(init, g['initial_%s_%s' % (res, tag)])
    dis, gw, fin = so.run(example['area'], *args, example['params'], extra, so.REPORT_SUMMARY, gap)
    assert np.array_equal(dis, g['discharge_%s_%s' % (res, tag)])
    assert gw == float(g['gw_%s_%s' % (res, tag)])
    assert np.array_equal(fin, g['final_%s_%s' % (res, tag)])


def test_kat4_batch_of_32_lhs_rows(example):
    g = load_golden('kat4_batch.npz')
    nd = int(g['n_days_hourly'])
    dis, gw, _ = so.run_batch(example['area'], 3600.0, nd * 24, 8760, example['rain_hourly'], example['peva_hourly'],
                              g['params'], example['extra'], so.REPORT_SUMMARY, 24)
    assert np.array_equal(dis, g['discharge_hourly_2yr']) and np.array_equal(gw, g['gw_hourly_2yr'])
    dis, gw, _ = so.run_batch(example['area'], 86400.0, 3653, 365, example['rain_daily'], example['peva_daily'],
                              g['params'], example['extra'], so.REPORT_SUMMARY, 1)
    assert np.array_equal(dis, g['discharge_daily_10yr']) and np.array_equal(gw, g['gw_daily_10yr'])
    # the GPU-friendly configuration (product chain for s'**i, sequential sums) stays at rounding level
    dis2, gw2, _ = so.run_batch(example['area'], 86400.0, 3653, 365, example['rain_daily'], example['peva_daily'],
                                g['params'], example['extra'], so.REPORT_SUMMARY, 1, pow_mode=so.POW_MUL,
                                sum_mode=so.SUM_SEQ)
    assert rel(dis2, dis) < 1e-11 and rel(gw2, gw) < 1e-12


def test_kat5_river_rule_and_clamps(example):
    g = load_golden('kat5_river.npz')
    L = int(g['n_steps'])
    hits = 0
    for p, tab in zip(g['params'], g['tables']):
        _, _, _, sto = so.all_steps(example['area'], 86400.0, L, example['rain_daily'], example['peva_daily'], p,
                                    tab[0], so.REPORT_SUMMARY, 1, want_storage=True)
        assert np.array_equal(sto, tab)
        q, v = tab[1:, 6], tab[:-1, 18]
        hits += int(np.sum(np.abs(q - v / (p[9] * 3600)) > 1e-9 * np.abs(q)))      # steps where the 95 % rule fired
    assert hits > 50


def test_kat9_raw_report_ragged_length(example):
    g = load_golden('kat9_raw_ragged.npz')
    dis, gw, fin = so.all_steps(example['area'], 3600.0, int(g['n_steps']), example['rain_hourly'],
                                example['peva_hourly'], example['params'], g['initial'], so.REPORT_RAW, int(g['gap']))
    assert len(dis) == 42 and np.array_equal(dis, g['discharge']) and np.array_equal(fin, g['final'])
    assert abs(gw - float(g['gw'])) <= 2e-16 * abs(gw)
    dis, gw, _ = so.all_steps(example['area'], 86400.0, 3653, example['rain_daily'], example['peva_daily'],
                              example['params'], g['initial'], so.REPORT_RAW, 1)
    assert np.array_equal(dis, g['discharge_gap1']) and gw == float(g['gw_gap1'])
    with pytest.raises(Exception):      # summary needs length % gap == 0 (np.reshape, structure.py:190)
        so.all_steps(example['area'], 3600.0, 1000, example['rain_hourly'], example['peva_hourly'],
                     example['params'], g['initial'], so.REPORT_SUMMARY, 24)


def test_kat7_lhs_sampler_bit_exact():
    g = load_golden('kat7_lhs.npz')
    for key in g.files:
        seed, n = key.split('_')
        got = lhs_oracle.lhs_params(int(n[1:]), seed=int(seed[4:]))
        assert np.array_equal(got, g[key]), key
        # latin property: one sample per stratum and per parameter
        lo = np.array([lhs_oracle.RANGES[p][0] for p in lhs_oracle.NAMES])
        hi = np.array([lhs_oracle.RANGES[p][1] for p in lhs_oracle.NAMES])
        strata = np.floor((got - lo) / (hi - lo) * len(got)).astype(int)
        assert all(sorted(np.clip(strata[:, j], 0, len(got) - 1)) == list(range(len(got))) for j in range(10))


def test_g4_objective_functions_pinned_by_example_database(example):
    """examples/out/ExampleDaily/ExampleDaily.SMART.lhs: 10 rows of float32 '%.8e' objective functions."""
    g = load_golden('g4_example_lhs.npz')
    dis, gw, _ = so.run_batch(example['area'], 3600.0, 87672, 8760, example['rain_hourly'], example['peva_hourly'],
                              g['params'], example['extra'], so.REPORT_SUMMARY, 24)
    assert np.array_equal(dis, g['discharge']) and np.array_equal(gw, g['gw'])
    got = objfn_oracle.objective_matrix(dis, example['flow_obs'], gw, float(g['gw_constraint']))
    assert got.shape == (10, 8)
    assert rel(got[:, :7], g['objfns'][:, :7]) < 5e-6          # float32 text precision of the database
    assert np.array_equal(got[:, 7], g['objfns'][:, 7])        # GW flags exactly
    # the shipped parameter set: NSE ~ 0.39 (examples/api_usage_example.ipynb cell 40)
    d0, g0, _ = so.run(example['area'], 3600.0, 87672, 8760, example['rain_hourly'], example['peva_hourly'],
                       example['params'], example['extra'], so.REPORT_SUMMARY, 24)
    o = objfn_oracle.objective_functions(d0, example['flow_obs'])
    assert abs(o[0] - 0.390445) < 1e-6 and abs(o[1] - 0.252085) < 1e-6 and abs(o[6] - 4.31063) < 1e-5
