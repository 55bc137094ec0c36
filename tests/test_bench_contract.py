"""bench.py: the synthetic workload of BASELINE.md section 4 (CPU) and the JSON line of the driver contract (GPU)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT


def test_synthetic_forcing_matches_the_survey_figures():
    """rain mean ~2.5 mm/d with ~20 % dry days (the shipped data: 2.56, 20 %), PE 0.22-2.72 mm/d,
    wet-step fraction w ~ 0.47."""
    sys.path.insert(0, ROOT)
    import bench
    daily, _ = bench.synthetic_forcing(0, hourly=False)
    assert daily.shape == (3653, 2)
    assert abs(daily[:, 0].mean() - 2.54) < 0.05 and abs((daily[:, 0] == 0).mean() - 0.20) < 0.01
    assert abs(daily[:, 1].min() - 0.22) < 0.01 and abs(daily[:, 1].max() - 2.72) < 0.01
    hourly, _ = bench.synthetic_forcing(0, hourly=True)
    assert hourly.shape == (87672, 2) and np.array_equal(hourly[:, 0], np.repeat(daily[:, 0] / 24, 24))
    w = bench.wet_fraction(hourly, 8760)
    assert abs(w - 0.473) < 0.002
    # against a direct count over a T sample
    T = np.linspace(0.9, 1.1, 2001)
    f = np.concatenate([hourly[:8760], hourly])[::97]
    direct = np.mean(f[:, 0][:, None] * T[None, :] - f[:, 1][:, None] >= 0)
    assert abs(bench.wet_fraction(hourly[::97], 0) - np.mean(hourly[::97, 0][:, None] * T - hourly[::97, 1][:, None] >= 0)) < 2e-3
    assert 0.4 < direct < 0.55
    other, _ = bench.synthetic_forcing(5, hourly=False)
    assert not np.array_equal(other, daily)


def test_pmc_figures_are_only_quoted_for_the_code_they_were_measured_on(tmp_path):
    """roofline.frac needs the vector-instruction count of exactly this workload on exactly this kernel build."""
    sys.path.insert(0, ROOT)
    import bench
    key = 'config3:runs_per_gpu=100000:discharge=1:math=fast'
    table = {'workloads': {key: {'source_hash': 'abc', 'valu_insts_per_launch': 5.0e9, 'hbm_bytes_per_launch': 3.0e9,
                                 'held_clock_hz': 2.1e9, 'issue_frac_at_held_clock': 0.8, 'source': 'somewhere.md'}}}
    path = tmp_path / 'traffic.json'
    path.write_text(json.dumps(table))
    got, note = bench.pmc_summary(key, str(path), source_hash='abc')
    assert got['valu_insts_per_launch'] == 5.0e9 and note == 'somewhere.md'
    got, note = bench.pmc_summary(key, str(path), source_hash='other')
    assert got == {} and 'other kernel sources' in note
    got, note = bench.pmc_summary('config3:runs_per_gpu=20000:discharge=1:math=fast', str(path), source_hash='abc')
    assert got == {} and 'no PMC summary for workload' in note
    got, note = bench.pmc_summary(key, str(tmp_path / 'missing.json'))
    assert got == {} and 'no PMC summary file' in note
    # the committed file, whatever its freshness, has the shape bench.py reads
    committed = json.load(open(os.path.join(ROOT, 'profiles', 'traffic_latest.json')))
    assert key in committed['workloads'] and len(bench.kernel_source_hash()) == 16
    for entry in committed['workloads'].values():
        # (a launch of several kernels side by side -- the daily leg -- carries the summed count and no fraction "at the
        # clock held": the counter passes serialise its kernels, tools/daily_roofline.py)
        held = entry['issue_frac_at_held_clock']
        assert (held is None or 0 < held <= 1.0) and entry['valu_insts_per_launch'] > 0


def test_bench_started_bare_with_gpus_2_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment (the way the driver starts N = 1) must not
    exit with "launch with torch.distributed.run": it becomes the launcher of its two ranks and relays their return
    code.  Without a GPU the ranks themselves stop at "bench.py needs a GPU" -- which is the evidence that they ran."""
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'],
                       cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    import torch
    if torch.cuda.is_available():
        pytest.skip('on a GPU box the GPU test below runs the same command to the end')
    assert r.returncode != 0
    assert b'bench.py needs a GPU' in r.stderr and b'launch with torch.distributed.run' not in r.stderr
    assert not [ln for ln in r.stdout.decode().splitlines() if ln.startswith('{')]


@pytest.mark.gpu
def test_bench_bare_command_with_two_gpus_spawns_two_ranks():
    """The driver's SCALE command may mirror its N = 1 command (no launcher): one JSON line, two ranks, two pids.  On
    a one-GPU box both ranks share GPU 0 and distributed.init() picks gloo by itself (more ranks than devices)."""
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'SMART_DIST_BACKEND')}
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2',
                                   '--warmup', '1', '--samples', '20000', '--no-flat', '--no-strong',
                                   '--no-cpu-baseline'], cwd=ROOT, env=env, stderr=subprocess.DEVNULL,
                                  timeout=900).decode()
    lines = [ln for ln in out.splitlines() if ln.startswith('{')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    r = d['ranks']
    assert d['n_gpus'] == 2 and r['world_size'] == 2 and len({x['pid'] for x in r['ranks']}) == 2
    import torch
    assert r['backend'] == ('nccl' if torch.cuda.device_count() >= 2 else 'gloo')
    assert r['rccl_version'] and all('current_device' in x for x in r['ranks'])
    assert d['series']['value'].startswith('weak') and 'strong' in d['series']['strong_1e6']


@pytest.mark.gpu
def test_bench_prints_one_json_line_with_the_contract_fields():
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '2', '--warmup', '1',
                                   '--samples', '20000'], cwd=ROOT, stderr=subprocess.DEVNULL).decode()
    lines = [ln for ln in out.splitlines() if ln.startswith('{')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
                'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert key in d, key
    assert d['n_gpus'] == 1 and d['steps'] == 2 and d['warmup'] == 1 and d['dtype'] == 'f64' and d['vs_baseline'] is None
    assert d['higher_is_better'] is True and d['scaling'] == 'weak' and d['data'] == 'synthetic'
    assert 'workload' in d['config'] and 'model' not in d['config']
    r = d['roofline']
    # the issue-slot fraction needs the PMC instruction count of this very workload and kernel build: absent for the
    # reduced sample count used here, and then it is null rather than a number measured on something else
    assert r['bound'] == 'valu-fp64-issue' and r['frac'] is None and 'no PMC summary for workload' in r['pmc_source']
    assert r['hbm']['unit'] == 'GB/s' and 0 < r['hbm']['frac'] < 1 and r['algorithmic_ratio']['ratio'] > 0
    assert 'smart_fast_intervals' in r['kernel'] and r['launch_ms'] > 0
    p = d['parity']
    assert p['ok'] is True and p['max_rel_discharge'] <= 1e-9 and p['gate'] == 1e-9 and p['contract'] == 1e-6 and p['rows'] >= 64
    print('bench parity: discharge %.2e relative, gw ratio %.2e absolute (gate %.0e, contract %.0e)' % (
        p['max_rel_discharge'], p['max_abs_gw_ratio'], p['gate'], p['contract']))
    # round 6: the launch that was TIMED is checked as well -- 48 rows of the stored matrix, their objective functions and
    # groundwater ratios against the oracle -- and says which kernel it was
    t = p['timed_launch']
    assert t['ok'] is True and t['kernel'] == r['kernel'] and t['ranks'] == 1 and t['rows'] == 48
    assert t['max_rel_discharge'] <= 1e-9 and t['max_abs_gw_ratio'] <= 1e-9 and t['max_rel_objfn'] <= 1e-8
    assert t['values'] == 48 * 3653 and d['observations'] == 'oracle'
    print('timed launch: discharge %.2e, objective functions %.2e, gw ratio %.2e' % (
        t['max_rel_discharge'], t['max_rel_objfn'], t['max_abs_gw_ratio']))
    # ... and a daily ensemble of 1e6 samples beside the headline: the literal rows in the throughput form
    dy = d['daily_1e6']
    assert 'smart_fast_illcond_lanes[' in dy['kernel'] and 'smart_fast_stiff' in dy['kernel']
    assert dy['parity']['ok'] and dy['parity']['rows'] == 48 and dy['rows_per_class']['literal'] > 100000
    assert dy['value'] > 1.5e11 and 'roofline' in dy          # (its issue fraction only for the profiled kernel build)
    assert 'smart_fast_intervals' in d['objectives_only']['kernel'] and d['objectives_only']['value'] > 0.9 * d['value']
    f = d['flat_forcing']
    assert 'smart_fast_steps' in f['kernel'] and 0 < f['value'] < d['value'] * 1.05
    assert abs(d['value'] - 20000 * 96432 / (d['ms_per_step'] * 1e-3)) < 1e-6 * d['value']
    r6 = d['runs_of_6']           # 6-hourly data in the hourly run: the interval engine over runs of six steps
    assert 'smart_fast_runs' in r6['kernel'] and 0 < r6['value'] < d['value'] * 1.05
    # round 4: raw reports and a report every step have kernels of their own (round 3: smart_fast_plain, unsliced)
    assert 'smart_fast_intervals_raw' in d['raw_gap24']['kernel'] and d['raw_gap24']['value'] > 0.5 * d['value']
    assert 'smart_fast_steps_every' in d['gap1']['kernel'] and d['gap1']['value'] > 0.25 * d['value']
    st = d['strong_1e6']          # config 4 beside config 3, one command for both series
    assert st['runs_total'] == st['runs_per_gpu'] == 1000000 and 'smart_fast_intervals_exits' in st['kernel']
    assert d['ranks']['world_size'] == 1 and len(d['ranks']['devices']) == 1
    assert 'useful_frac' in r and 'not a bound' in r['algorithmic_ratio']['note']
    assert 'executed_flops' in r and r['executed_flops'] is None      # counted flops: like frac, only for the profiled workload
    c = d['cpu_baseline']
    assert c['kind'] == 'port' and c['cores'] >= min(2, os.cpu_count()) and c['value'] > 1e6 and 'sample' in c   # (all host cores: the truth run must not pin OpenMP to one)
    assert d['value'] > 50 * c['value']          # sanity: the GPU path is orders of magnitude ahead of the host cores


@pytest.mark.gpu
@pytest.mark.parametrize('config, samples', [(3, 20000), (4, 40000), (5, 1000)])
def test_bench_with_two_ranks_launched_the_way_the_driver_does(config, samples):
    """`python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2` on a one-GPU box: both ranks share GPU 0
    and the collectives go through gloo (SMART_DIST_BACKEND; RCCL refuses two ranks on one device) -- the sharding,
    gathering, barrier + max-over-ranks timing and the single JSON line are the code of the multi-GPU run."""
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ, SMART_DIST_BACKEND='gloo')
    out = subprocess.check_output(
        [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr',
         '127.0.0.1', '--master-port', str(port), os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2',
         '--warmup', '1', '--config', str(config), '--samples', str(samples), '--no-flat'] +
        ([] if config == 3 else ['--no-cpu-baseline']),       # config 3: the whole line, cpu_baseline and parity included
        cwd=ROOT, env=env, stderr=subprocess.DEVNULL, timeout=900).decode()
    lines = [ln for ln in out.splitlines() if ln.startswith('{')]
    assert len(lines) == 1                                   # rank 0 prints, rank 1 does not
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['steps'] == 2 and d['value'] > 0
    c = d['config']
    if config == 3:         # weak: every rank its own block of the ensemble
        assert d['scaling'] == 'weak' and c['runs_per_gpu'] == samples and c['runs_total'] == 2 * samples
    elif config == 4:       # strong: one ensemble cut by rows
        assert d['scaling'] == 'strong' and c['runs_total'] == samples and c['runs_per_gpu'] == samples // 2
    else:                   # strong: 64 catchments cut by catchment
        assert d['scaling'] == 'strong' and c['runs_total'] == 64 * samples and c['runs_per_gpu'] == 32 * samples
    # the line says who took part: backend, one device and one launch time per rank
    r = d['ranks']
    assert r['backend'] == 'gloo' and r['world_size'] == 2 and len(r['devices']) == 2
    assert len(r['launch_ms_per_rank']) == 2 and all(ms > 0 for ms in r['launch_ms_per_rank'])
    assert sorted(x['rank'] for x in r['ranks']) == [0, 1] and len({x['pid'] for x in r['ranks']}) == 2
    # every rank has checked rows of its own timed launch against the oracle on its host
    t = d['parity']['timed_launch']
    assert t['ok'] and t['ranks'] == 2 and t['kernel'] == d['roofline']['kernel'] and t['max_rel_objfn'] <= 1e-8
    assert (t['max_rel_discharge'] is not None) == (config == 3)          # (configs 4 and 5 store no matrix)
    if config == 3:
        # N > 1 keeps rank 0's CPU baseline and the in-run parity check, and carries config 4's strong-scaled figure
        assert d['cpu_baseline']['kind'] == 'port' and d['cpu_baseline']['value'] > 1e6
        assert d['parity']['ok'] and d['parity']['max_rel_discharge'] <= 1e-9
        st = d['strong_1e6']
        assert st['scaling'] == 'strong' and st['runs_total'] == 1000000 and st['runs_per_gpu'] == 500000
        assert len(st['launch_ms_per_rank']) == 2 and st['value'] > 0
    per_step = c['runs_total'] * 96432            # sample-timesteps of one step, warm-up included
    assert abs(d['value'] - per_step / (d['ms_per_step'] * 1e-3)) < 1e-6 * d['value']


@pytest.mark.gpu
@pytest.mark.parametrize('config, samples', [(3, 8000), (4, 8000), (5, 250)])
def test_bench_with_eight_ranks_launched_the_way_the_driver_does(config, samples):
    """What the driver's 8-GPU node runs -- `python -m torch.distributed.run --nproc-per-node 8 bench.py --gpus 8` -- on
    a one-GPU box: eight ranks share GPU 0, distributed.init() finds that out by itself (the ranks compare the UUIDs of
    their devices) and stages the result blocks through gloo.  Everything else is the code of the real run: rank 0
    alone builds the observations and broadcasts them, shards of 1,000 rows (config 4) / 8 catchments (config 5), the
    all-gather, barrier + max-over-ranks timing, one JSON line."""
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k != 'SMART_DIST_BACKEND'}
    out = subprocess.check_output(
        [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '8', '--master-addr',
         '127.0.0.1', '--master-port', str(port), os.path.join(ROOT, 'bench.py'), '--gpus', '8', '--steps', '2',
         '--warmup', '1', '--config', str(config), '--samples', str(samples), '--no-flat', '--no-strong',
         '--no-cpu-baseline'], cwd=ROOT, env=env, stderr=subprocess.DEVNULL, timeout=1200).decode()
    lines = [ln for ln in out.splitlines() if ln.startswith('{')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    c, r = d['config'], d['ranks']
    assert d['n_gpus'] == 8 and r['world_size'] == 8 and len({x['pid'] for x in r['ranks']}) == 8
    assert sorted(x['rank'] for x in r['ranks']) == list(range(8)) and len(r['launch_ms_per_rank']) == 8
    import torch
    assert r['backend'] == ('nccl' if torch.cuda.device_count() >= 8 else 'gloo')
    if config == 3:
        assert d['scaling'] == 'weak' and c['runs_per_gpu'] == samples and c['runs_total'] == 8 * samples
    elif config == 4:
        assert d['scaling'] == 'strong' and c['runs_total'] == samples and c['runs_per_gpu'] == samples // 8
    else:
        assert d['scaling'] == 'strong' and c['runs_total'] == 64 * samples and c['runs_per_gpu'] == 8 * samples
    assert abs(d['value'] - c['runs_total'] * 96432 / (d['ms_per_step'] * 1e-3)) < 1e-6 * d['value']
    # round 6: per-rank correctness of an N-rank run -- each of the eight ranks has pushed rows of ITS OWN timed job
    # through the oracle, the line carries the largest difference and how many ranks took part
    t = d['parity']['timed_launch']
    assert t['ok'] and t['ranks'] == 8 and t['kernel'] == d['roofline']['kernel']
    assert t['rows'] == 8 * min(48, c['runs_per_gpu']) if config != 5 else t['rows'] >= 8 * 12
    assert t['max_abs_gw_ratio'] <= 1e-9 and t['max_rel_objfn'] <= 1e-8


@pytest.mark.gpu
def test_bench_without_a_checker_still_prints_its_line():
    """A box without gcc and without a built oracle (round 5: no line at all): the observations come from the committed
    fixture, the throughput line stands, `cpu_baseline` and `parity.timed_launch` are null with the reason."""
    env = dict(os.environ, SMART_BENCH_NO_ORACLE='1')
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '2', '--warmup', '1',
                                   '--samples', '20000', '--no-flat', '--no-strong'], cwd=ROOT, env=env,
                                  stderr=subprocess.DEVNULL).decode()
    d = json.loads([ln for ln in out.splitlines() if ln.startswith('{')][0])
    assert d['value'] > 1e11 and d['cpu_baseline'] is None and d['observations'].startswith('fixture')
    assert d['parity']['timed_launch'] is None and 'SMART_BENCH_NO_ORACLE' in d['parity']['why']


def test_the_committed_truth_fixture_is_what_the_oracle_computes():
    """tests/golden/bench_truth.npz (bench.py --obs-from-fixture) against a run of the oracle, bit for bit."""
    sys.path.insert(0, ROOT)
    import bench
    from oracle import smart_oracle as so
    for hourly in (True, False):
        forcing, _ = bench.synthetic_forcing(0, hourly=hourly)
        dt, gap = (3600.0, 24) if hourly else (86400.0, 1)
        W = bench.WARM_DAYS * (24 if hourly else 1)
        live, src = bench.truth_discharge(so, forcing, dt, forcing.shape[0], W, gap, hourly)
        kept, src2 = bench.truth_discharge(None, forcing, dt, forcing.shape[0], W, gap, hourly)
        assert src == 'oracle' and src2.startswith('fixture') and np.array_equal(live, kept)


def test_the_legs_of_the_line_are_priced_against_the_issue_roof_with_their_own_counts():
    """bench.leg_roofline: a leg's launch time against the issue cycles of 1,024 SIMDs with the vector-instruction count of
    that leg's own committed PMC summary (profiles/traffic_latest.json, key leg:<name>) -- and nothing at all when the
    kernel sources are not the ones that were profiled"""
    import bench
    for leg in ('flat_forcing', 'runs_of_6', 'raw_gap24', 'gap1'):
        pmc, _ = bench.pmc_summary('leg:' + leg)
        r = bench.leg_roofline(leg, 12.0)
        if not pmc:
            assert r == {}
            continue
        assert r['frac'] == pmc['valu_insts_per_launch'] * 4 / (1024 * 2.4e9 * 12.0e-3) and 0.3 < r['frac'] < 1.0
        assert 0.5 < r['frac_at_held_clock_profiled'] < 1.0
    assert bench.leg_roofline('no_such_leg', 12.0) == {} and bench.leg_roofline('flat_forcing', 0.0) == {}
