#!/usr/bin/env python3
"""Headline benchmark: Monte-Carlo sample-timesteps per second of the SMART time loop on MI355X.

Contract: `python bench.py --gpus N --steps K --warmup W` (N > 1: launched by torch.distributed.run, one rank per
GPU, RCCL).  One "step" = one pass of the hot path over one batch = the ensemble launch that advances every run of
this rank's shard through warm-up + simulation, followed, for N > 1, by the one RCCL all-gather of the objective /
groundwater arrays.  `--config` picks the BASELINE.json configuration (default 3, the one the metric is quoted on):

  2  1e4 LHS samples x daily 10 yr, per GPU (replicas for N > 1)                                   weak
  3  1e5 LHS samples x hourly 10 yr per GPU, sample-sharded                       (headline)       weak
  4  1e6 LHS samples x hourly 10 yr in total, sample-sharded over the N GPUs, gather of [N, 9]     strong
  5  64 synthetic catchments x 1e4 samples x hourly 10 yr in total, catchment-sharded              strong

Rank 0 prints ONE JSON line; DESIGN.md section 5 explains every field.  At N = 1 the line also carries, measured in
the same run: `parity` (GPU against the CPU oracle on the rows the cpu_baseline leg simulates anyway),
`flat_forcing` (the same workload with forcing that varies inside the day: the step loop instead of the interval
engine), `objectives_only` (the same runs without the stored discharge matrix, the Monte-Carlo default) and
`cpu_baseline`.
"""
import argparse
import hashlib
import json
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from smartpy_amd import distributed as sdist          # noqa: E402
from smartpy_amd import engine                        # noqa: E402
from smartpy_amd.parameters import Parameters         # noqa: E402
from smartpy_amd.sampling import latin_hypercube      # noqa: E402

AREA = 175.46e6
EXTRA = {'aar': 1200, 'r-o_ratio': 0.45, 'r-o_split': (0.10, 0.15, 0.15, 0.30, 0.30)}
TRUTH = [1.0, 1.0, 0.20845296027652363, 0.24606006380093334, 0.00012296588050682812, 105.25734595830215,
         46.81961454361724, 315.5490902162102, 1066.7332319333473, 10.640277777777778]   # Catchment.parameters
N_DAYS, WARM_DAYS = 3653, 365
FP64_VALU_PEAK_TFLOPS = 78.6    # 1/2 of the 157.3 TF fp32 vector peak (MI355X_MICROARCH.md chip table)
HBM_PEAK_GBS = 8000.0           # same table
N_SIMD = 256 * 4                # 256 CUs x 4 SIMDs
CLOCK_HZ = 2.4e9                # peak engine clock, same table
VALU_ISSUE_CYCLES = 4           # a wave64 fp64 (or any vector) instruction occupies its SIMD for 4 cycles
PARITY_CONTRACT = 1e-6          # BASELINE.json: discharge <= 1e-6 relative against the CPU reference path
PARITY_GATE = 1e-9              # what this run is held to: 1,000 x tighter (measured: 3e-12); non-zero exit above it
GW_OBS = 0.12667


def synthetic_forcing(catchment=0, hourly=True):
    """BASELINE.md section 4 / SURVEY.md 8(d): seeded daily rain and PE, hourly = daily / 24 repeated."""
    rng = np.random.default_rng(12345 + catchment)
    wet = rng.random(N_DAYS) < 0.80
    rain_d = wet * rng.gamma(0.70, 4.57, N_DAYS)
    doy = np.arange(N_DAYS) % 365.25
    pe_d = np.maximum(0.0, 1.47 * (1 + 0.85 * np.sin(2 * np.pi * (doy - 110) / 365.25)))
    if not hourly:
        return np.stack([rain_d, pe_d], axis=1), rng
    return np.stack([np.repeat(rain_d / 24, 24), np.repeat(pe_d / 24, 24)], axis=1), rng


def hourly_varying_forcing(base, seed=3):
    """The same daily totals spread UNEVENLY over the hours of each day (rain in ~6 random hours, PE on a daytime
    sine): genuinely sub-daily forcing, which the interval engine does not apply to."""
    rng = np.random.default_rng(seed)
    days = base.shape[0] // 24
    wts = rng.random((days, 24)) * (rng.random((days, 24)) < 0.25)
    wts[wts.sum(1) == 0, 0] = 1.0
    wts /= wts.sum(1, keepdims=True)
    out = base.copy()
    out[:, 0] = (base[::24, 0][:, None] * 24 * wts).ravel()
    day = np.maximum(0.0, np.sin(np.pi * (np.arange(24) - 5) / 14))
    day /= day.sum()
    out[:, 1] = (base[::24, 1][:, None] * 24 * day[None, :]).ravel()
    return out


def six_hourly_forcing(base, seed=4):
    """The same daily totals as four 6-hour values per day (rain in about half of the blocks, PE by the sun's share
    of each block), each spread equally over its six hourly steps -- what the reference's pipeline makes of 6-hourly
    input files in an hourly run (timeframe.py:167-186).  Constant over runs of six steps, not over the report
    interval: the run engine (smart_fast_runs)."""
    rng = np.random.default_rng(seed)
    days = base.shape[0] // 24
    wts = rng.random((days, 4)) * (rng.random((days, 4)) < 0.5)
    wts[wts.sum(1) == 0, 0] = 1.0
    wts /= wts.sum(1, keepdims=True)
    out = base.copy()
    out[:, 0] = np.repeat((base[::24, 0][:, None] * 24 * wts).ravel() / 6, 6)
    day = np.maximum(0.0, np.sin(np.pi * (np.arange(24) - 5) / 14)).reshape(4, 6).sum(1)
    day /= day.sum()
    out[:, 1] = np.repeat((base[::24, 1][:, None] * 24 * day[None, :]).ravel() / 6, 6)
    return out


def wet_fraction(forcing, n_warm, t_lo=0.9, t_hi=1.1):
    """Realised fraction of executed sample-steps on the wet branch (rain * T - peva >= 0, T ~ U[t_lo, t_hi])."""
    f = np.concatenate([forcing[:n_warm], forcing])
    rain, pe = f[:, 0], f[:, 1]
    with np.errstate(divide='ignore', invalid='ignore'):
        thr = np.where(rain > 0, pe / rain, np.where(pe <= 0, -np.inf, np.inf))
    return float(np.mean(np.clip((t_hi - thr) / (t_hi - t_lo), 0.0, 1.0)))


def _code_only(text):
    """C / C++ source without its comments, runs of white space collapsed (string literals left alone).  A ' opens a
    character literal only where one stands ('x', '\\n', '\\''): a digit separator (1'000'000) or an apostrophe in an
    #error text is an ordinary character -- round 4's scanner would have swallowed code up to the next quote."""
    import re
    char_literal = re.compile(r"'(\\.|[^\\'\n])'")
    out, i, n = [], 0, len(text)
    while i < n:
        c = text[i]
        if c == '"':
            j = i + 1
            while j < n and text[j] != c:
                j += 2 if text[j] == '\\' else 1
            out.append(text[i:j + 1])
            i = j + 1
        elif c == "'" and char_literal.match(text, i):
            j = char_literal.match(text, i).end()
            out.append(text[i:j])
            i = j
        elif text.startswith('//', i):
            i = text.find('\n', i)
            i = n if i < 0 else i
        elif text.startswith('/*', i):
            i = text.find('*/', i + 2)
            i = n if i < 0 else i + 2
        else:
            out.append(c)
            i += 1
    return ' '.join(''.join(out).split())


def kernel_source_hash():
    """sha256 over what decides the generated code: the kernel sources and the C ABI header, comments and white space
    aside -- a PMC summary is only quoted for the code it was measured on, and an edit of a comment is not an edit of
    the code -- and the compiler flags of every translation unit (smartpy_amd/build.py: -ffp-contract, -fno-honor-nans,
    the -D switches change the code as much as the text does)."""
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, 'smartpy_amd', 'csrc')
    files = [os.path.join(csrc, name) for name in sorted(os.listdir(csrc)) if name.endswith(('.hip', '.h', '.cpp'))]
    files.append(os.path.join(ROOT, 'include', 'smart_amd.h'))
    for path in files:
        with open(path, 'r', encoding='utf-8', errors='replace') as fh:
            h.update(os.path.basename(path).encode() + b'\0' + _code_only(fh.read()).encode())
    from smartpy_amd import build as hip_build
    for unit in sorted(hip_build.UNITS):
        h.update(('%s: %s\0' % (unit, ' '.join(hip_build.COMMON + hip_build.UNITS[unit]))).encode())
    return h.hexdigest()[:16]


def pmc_summary(pmc_key, path=None, source_hash=None):
    """The committed PMC figures of one workload (profiles/traffic_latest.json, written by tools/summarize_profile.py
    from rocprofv3 --pmc passes of this same command) -> (figures, where they come from).  They are properties of
    (kernel build, workload), so they are only handed out while the kernel sources still hash to what was profiled;
    otherwise ({}, the reason) and the bench line carries nulls rather than numbers measured on other code."""
    path = path or os.path.join(ROOT, 'profiles', 'traffic_latest.json')
    if not os.path.exists(path):
        return {}, 'no PMC summary file (profiles/traffic_latest.json)'
    with open(path) as fh:
        pmc = json.load(fh).get('workloads', {}).get(pmc_key)
    if pmc is None:
        return {}, 'no PMC summary for workload %s in profiles/traffic_latest.json' % pmc_key
    now = source_hash or kernel_source_hash()
    if pmc.get('source_hash') != now:
        return {}, 'profiles/traffic_latest.json was measured on other kernel sources (hash %s, now %s): not quoted' \
                   % (pmc.get('source_hash'), now)
    return pmc, pmc.get('source', 'profiles/traffic_latest.json')


def leg_roofline(leg, launch_ms):
    """Issue-roof figures of one leg of the line (the same runs under another mode) from that leg's own committed PMC
    summary (profiles/r04_<leg>.md through profiles/traffic_latest.json, key leg:<leg>): vector instructions x 4 issue
    cycles over the issue cycles 1,024 SIMDs have in the leg's launch at the peak clock; {} when the kernel sources are
    not the profiled ones."""
    pmc, _ = pmc_summary('leg:' + leg)
    if not pmc.get('valu_insts_per_launch') or not launch_ms:
        return {}
    return {'frac': pmc['valu_insts_per_launch'] * VALU_ISSUE_CYCLES / (N_SIMD * CLOCK_HZ * launch_ms * 1e-3),
            'frac_at_held_clock_profiled': pmc.get('issue_frac_at_held_clock'),
            'valu_insts_per_launch': pmc['valu_insts_per_launch'], 'bound': 'fp64 vector-ALU issue'}


def isa_model_summary(pmc_key, path=None, source_hash=None):
    """The vector-instruction count of one workload as tools/isa_model.py derives it from the TREE (hipcc's assembly of
    the hot loops weighed with the workload's path frequencies; cross-compiled, no GPU) -> (figures, source); like the
    PMC summary only for the kernel sources it was derived from."""
    path = path or os.path.join(ROOT, 'profiles', 'isa_model_latest.json')
    if not os.path.exists(path):
        return {}, 'no instruction model (profiles/isa_model_latest.json)'
    with open(path) as fh:
        model = json.load(fh).get('workloads', {}).get(pmc_key)
    if model is None:
        return {}, 'no instruction model for workload %s' % pmc_key
    now = source_hash or kernel_source_hash()
    if model.get('source_hash') != now:
        return {}, 'profiles/isa_model_latest.json was derived from other kernel sources (hash %s, now %s): not quoted' \
                   % (model.get('source_hash'), now)
    return model, model.get('source', 'profiles/isa_model_latest.json')


TRUTH_FIXTURE = os.path.join(ROOT, 'tests', 'golden', 'bench_truth.npz')
TIMED_ROWS = 48                  # rows of its own timed launch every rank pushes through the oracle afterwards
OBJFN_GATE = 1e-8                # objective functions (one-pass moments against numpy's two-pass): |diff| / max(|want|, 1)


def load_oracle():
    """The checker (oracle/smart_oracle.c through its ctypes binding), or (None, why not): a box without a C compiler and
    without a built oracle library still gets its throughput line -- with `cpu_baseline` and `parity` null and the
    reason beside them (round 5 died here)."""
    if os.environ.get('SMART_BENCH_NO_ORACLE'):      # (tests: what a box without gcc sees)
        return None, 'disabled by SMART_BENCH_NO_ORACLE'
    try:
        from oracle import smart_oracle as so
        so.max_threads()
        return so, None
    except Exception as e:      # noqa: BLE001 -- no gcc, no write access, a library that does not load: all the same here
        return None, '%s: %s' % (type(e).__name__, (str(e).strip().splitlines() or [''])[0][:200])


def truth_discharge(so, forcing, dt, T, W, gap, hourly, from_fixture=False):
    """Discharge of the "truth" parameter set, the base of the synthetic observations: from the oracle, or -- without one,
    or when --obs-from-fixture says so -- from the committed fixture the oracle produced in the build container
    (tests/golden/make_bench_truth.py; 58 KB).  Returns (series [R], where it came from)."""
    if so is not None and not from_fixture:
        return so.run_batch(AREA, dt, T, W, np.ascontiguousarray(forcing[:, 0]), np.ascontiguousarray(forcing[:, 1]),
                            np.array([TRUTH]), EXTRA, so.REPORT_SUMMARY, gap, want_discharge=True)[0][0], 'oracle'
    with np.load(TRUTH_FIXTURE) as z:
        return z['hourly' if hourly else 'daily'].copy(), 'fixture tests/golden/bench_truth.npz'


def synthetic_observations(truth, rng, R):
    obs = truth * np.exp(rng.normal(0.0, 0.2, R))
    obs[rng.random(R) < 0.12] = np.nan
    return obs


def rows_against_the_oracle(so, res, rows, params, forcing, area, obs, n_warm, gap, dt, catchment=None):
    """`rows` of a finished launch (`res`: its EnsembleResult, rows in the caller's order) against the oracle on the same
    inputs: the stored daily means where the launch stored them, the objective functions and the groundwater ratio
    always.  -> dict of the largest differences."""
    from oracle import objfn_oracle
    T = forcing.shape[0]
    want, want_gw, _ = so.run_batch(area, dt, T, n_warm, np.ascontiguousarray(forcing[:, 0]),
                                    np.ascontiguousarray(forcing[:, 1]), np.ascontiguousarray(params[rows]), EXTRA,
                                    so.REPORT_SUMMARY, gap, want_discharge=True, n_threads=so.max_threads())
    pick = torch.as_tensor(rows, device=res.gw.device)
    take = (lambda t: t[pick]) if catchment is None else (lambda t: t[catchment][pick])
    out = {'max_rel_discharge': None, 'values': 0}
    if res.discharge is not None:
        got = take(res.discharge).cpu().numpy()
        out['max_rel_discharge'] = float(np.max(np.abs(got - want) / np.maximum(np.abs(want), 1e-300)))
        out['values'] = int(got.size)
    out['max_abs_gw_ratio'] = float(np.max(np.abs(take(res.gw).cpu().numpy() - want_gw)))
    if res.objfn is not None and obs is not None:
        fn = objfn_oracle.objective_matrix(want, obs, want_gw, GW_OBS)
        got = take(res.objfn).cpu().numpy()
        out['max_rel_objfn'] = float(np.max(np.abs(got[:, :7] - fn[:, :7]) / np.maximum(np.abs(fn[:, :7]), 1.0)))
        out['gw_flag_equal'] = bool(np.array_equal(got[:, 7], fn[:, 7]))
    return out


def timed_launch_parity(so, why_not, job, kind, params_local, forcing_of, area_of, obs, n_warm, gap, dt):
    """EVERY rank checks the launch it timed: TIMED_ROWS rows (fixed seed per rank) of its OWN job -- the stored [R, N]
    rows where the matrix is stored, objective functions and groundwater ratio always -- against the oracle on its host;
    the largest difference over the ranks goes into the line (round 5 checked a separate, unsliced 4,096-row batch on
    rank 0 only: the launch the number came from was only looked at for NaNs).  kind: 'samples' (params_local = this
    rank's rows) or 'catchments' (params_local = the shared rows; this rank's catchments are job's)."""
    rank, world = sdist.rank_world()
    have = sdist.sum_over_ranks(1.0 if so is not None else 0.0)
    if have < world:
        return {'timed_launch': None, 'why': 'no oracle on %d of %d rank(s): %s' % (world - int(have), world, why_not)}
    worst = {'max_rel_discharge': -1.0, 'max_abs_gw_ratio': 0.0, 'max_rel_objfn': 0.0}
    n_rows, flags_ok, values = 0, True, 0
    if job.n_local > 0:
        res = job.prepared.result()
        rng = np.random.default_rng(6006 + rank)
        if kind == 'samples':
            rows = np.sort(rng.choice(job.n_local, size=min(TIMED_ROWS, job.n_local), replace=False))
            got = [rows_against_the_oracle(so, res, rows, params_local, forcing_of(0), area_of(0), obs, n_warm, gap, dt)]
            n_rows = len(rows)
        else:
            lo, _ = sdist.shard_bounds(job.n_total, world, rank)
            per_c = max(1, TIMED_ROWS // min(job.n_local, 4))
            got = []
            for c in sorted(rng.choice(job.n_local, size=min(job.n_local, 4), replace=False).tolist()):
                rows = np.sort(rng.choice(params_local.shape[0], size=min(per_c, params_local.shape[0]), replace=False))
                got.append(rows_against_the_oracle(so, res, rows, params_local, forcing_of(lo + c), area_of(lo + c), obs,
                                                   n_warm, gap, dt, catchment=c))
                n_rows += len(rows)
        for g in got:
            for k in worst:
                if g.get(k) is not None:
                    worst[k] = max(worst[k], g[k])
            flags_ok = flags_ok and g.get('gw_flag_equal', True)
            values += g['values']
    out = {k: sdist.max_over_ranks(v) for k, v in worst.items()}
    if out['max_rel_discharge'] < 0.0:
        out['max_rel_discharge'] = None         # (no matrix stored on any rank: objective functions and ratio only)
    out['gw_flag_equal'] = sdist.sum_over_ranks(0.0 if flags_ok else 1.0) == 0.0
    out['rows_per_rank'] = TIMED_ROWS
    out['rows'] = int(sdist.sum_over_ranks(n_rows))
    out['ranks'] = int(sdist.sum_over_ranks(1.0 if n_rows else 0.0))
    out['values'] = int(sdist.sum_over_ranks(values))
    out['kernel'] = job.prepared.describe()
    out['gate'] = {'discharge': PARITY_GATE, 'gw_ratio': PARITY_GATE, 'objfn': OBJFN_GATE, 'contract': PARITY_CONTRACT}
    out['ok'] = bool((out['max_rel_discharge'] is None or out['max_rel_discharge'] <= PARITY_GATE)
                     and out['max_abs_gw_ratio'] <= PARITY_GATE and out['max_rel_objfn'] <= OBJFN_GATE
                     and out['gw_flag_equal'] and out['ranks'] >= 1)
    out['against'] = 'oracle/smart_oracle.c on every rank\'s host (reference operation order, libm pow, numpy summation ' \
                     'order) + oracle/objfn_oracle.py, on rows of the timed job itself'
    return {'timed_launch': out}



def cpu_baseline_and_parity(so, forcing, n_warm, gap, dt, device, budget_s=12.0):
    """(1) The oracle's OpenMP batch runner (a C port of the reference loop) on the host cores, on a bounded sample of
    the same workload: reported next to the GPU number, never part of the timed GPU region.  (2) The same rows through
    the engine: the largest relative difference of the discharge and of the groundwater ratio -- parity measured in
    the run that reports the throughput."""
    cores = so.max_threads()
    rain, pe = np.ascontiguousarray(forcing[:, 0]), np.ascontiguousarray(forcing[:, 1])
    T = len(rain)
    ranges = Parameters().ranges

    def run(n):
        p = latin_hypercube(n, ranges, seed=99)
        t0 = time.perf_counter()
        dis, gw, _ = so.run_batch(AREA, dt, T, n_warm, rain, pe, p, EXTRA, so.REPORT_SUMMARY, gap,
                                  want_discharge=True, n_threads=cores)
        return time.perf_counter() - t0, p, dis, gw

    run(2 * cores)                      # first touch of the per-thread tables, thread start-up
    probe_n = 8 * cores
    probe = run(probe_n)[0]
    n = int(max(probe_n, min(4096, probe_n * budget_s / max(probe, 1e-3))))
    n -= n % cores
    secs, p, want, want_gw = run(n)
    cpu = {'value': n * (T + n_warm) / secs, 'unit': 'sample-timesteps/s', 'cores': cores, 'kind': 'port',
           'sample': '%d LHS samples x %d steps (%d + warm-up %d), oracle/smart_oracle.c with OpenMP, %.1f s'
                     % (n, T + n_warm, T, n_warm, secs),
           'reference_python_1core': 6.9e4}
    got = engine.run_ensemble(p, forcing, AREA, dt, n_warm, gap, extra=EXTRA, device=device)
    dis = got.discharge.cpu().numpy()
    gw = got.gw.cpu().numpy()
    rel = float(np.max(np.abs(dis - want) / np.maximum(np.abs(want), 1e-300)))
    gw_abs = float(np.max(np.abs(gw - want_gw)))
    parity = {'max_rel_discharge': rel, 'max_abs_gw_ratio': gw_abs, 'gate': PARITY_GATE, 'contract': PARITY_CONTRACT,
              'ok': bool(rel <= PARITY_GATE and gw_abs <= PARITY_GATE),
              'against': 'oracle/smart_oracle.c (reference operation order, libm pow, numpy summation order)',
              'rows': n, 'values': int(dis.size)}
    return cpu, parity


def api_legs(forcing, obs, W_days, device):
    """The package's own entry points on the driver's line (round 4's verdict: the API path had no number).
    simulate_ms: SMART.simulate() -- ONE parameter set per call, what a calibration loop written against the
    reference's per-sample protocol calls (smart.py:154-210, montecarlo.py:179-186) -- on the benchmark's hourly
    10-year series, repeated calls (the series stays on the device: 80 bytes go up per call); and the smartcpp.allsteps
    stand-in the unmodified reference binds, in the reference's own arithmetic (two calls per simulate(): warm-up, run).
    e2e_lhs: montecarlo.LHS('Catchment', ..., 100000) on the shipped example catchment (tests/golden/data/in: ten years
    of hourly steps, daily reports) -- constructor (file parsing, time axes, sampling) and run() (one launch, device ->
    host, the 23.5 MB sampling database written in the reference's format)."""
    import shutil
    import tempfile
    from datetime import datetime, timedelta
    from smartpy_amd import SMART, smartcpp
    from smartpy_amd.montecarlo import LHS
    legs = {}
    T = forcing.shape[0]
    start = datetime(2007, 1, 1, 9)
    # (the period's last day counts: 3,653 daily reports from 01/01/2007 to 31/12/2016, like the shipped example)
    sm = SMART.from_arrays(AREA, start, start + timedelta(hours=T - 24), timedelta(hours=1), timedelta(days=1), W_days,
                           forcing[:, 0], forcing[:, 1], nd_flow=obs)
    sm.extra = EXTRA
    names = sm.parameters.names
    ranges = Parameters().ranges
    rows = latin_hypercube(8, ranges, seed=5)
    sm.simulate(dict(zip(names, TRUTH)))                        # first call: upload, buffers, plan
    before, ts = engine.h2d_bytes, []
    for row in rows:
        t0 = time.perf_counter()
        dis, gw = sm.simulate(dict(zip(names, row)))
        ts.append((time.perf_counter() - t0) * 1e3)
    legs['simulate_ms'] = {
        'what': 'SMART.simulate(): one parameter set per call over %d hourly steps (+ %d of warm-up), daily means back on '
                'the host; forcing, buffers and plan kept on the device between calls' % (T, W_days * 24),
        'best': min(ts), 'median': float(np.median(ts)), 'calls': len(ts),
        'h2d_bytes_per_call': (engine.h2d_bytes - before) / len(ts), 'reports': int(len(dis)),
        'reference_python_1core_s': (T + W_days * 24) / 6.9e4}
    rain, peva = np.ascontiguousarray(forcing[:, 0]), np.ascontiguousarray(forcing[:, 1])
    hs = []
    for _ in range(3):
        t0 = time.perf_counter()
        smartcpp.allsteps(AREA, 3600.0, T, rain, peva, np.array(TRUTH), np.zeros(19), 1, 24)
        hs.append((time.perf_counter() - t0) * 1e3)
    legs['simulate_ms']['hook_literal_ms'] = min(hs)
    legs['simulate_ms']['hook_literal_what'] = ('smartcpp.allsteps over the same %d steps in the reference\'s own operation '
                                                'order (one sample per DPP row, smart_ensemble_literal_rows)' % T)
    data = os.path.join(ROOT, 'tests', 'golden', 'data', 'in')
    if os.path.isdir(data):
        tmp = tempfile.mkdtemp(prefix='smart_bench_')
        try:
            shutil.copytree(data, os.path.join(tmp, 'in'))
            np.random.seed(2718)
            t0 = time.perf_counter()
            lhs = LHS('Catchment', tmp, 'csv', 'csv', 100000, save_sim=False)
            lhs.model.extra = EXTRA
            t1 = time.perf_counter()
            lhs.run()
            torch.cuda.synchronize(device)
            t2 = time.perf_counter()
            lhs.run()
            t3 = time.perf_counter()
            # ... and SMART.simulate() on that very model (the shipped example's own series and parameter file)
            lhs.model.parameters.set_parameters_with_file(os.path.join(tmp, 'in', 'Catchment', 'Catchment.parameters'))
            lhs.model.simulate(lhs.model.parameters.values)
            es = []
            for row in rows[:5]:
                t4 = time.perf_counter()
                lhs.model.simulate(dict(zip(names, row)))
                es.append((time.perf_counter() - t4) * 1e3)
            legs['simulate_ms']['shipped_example_best'] = min(es)
            legs['e2e_lhs'] = {
                'what': "montecarlo.LHS('Catchment', root, 'csv', 'csv', 100000) on the shipped example (2007-2016, hourly "
                        "steps, daily reports, 365 days of warm-up): constructor, then run() = one launch + device -> host "
                        "+ the sampling database in the reference's format",
                'construct_s': t1 - t0, 'first_run_s': t2 - t1, 'run_s': t3 - t2,
                'database_mb': os.path.getsize(lhs.db_file) / 1e6, 'samples': 100000,
                'sample_timesteps_per_s': 100000 * (87648 + 8760) / (t3 - t2),
                'reference_core_hours': 100000 * (87648 + 8760) / 6.9e4 / 3600.0}
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    return legs


def daily_leg(so, n, device, steps, from_fixture=False):
    """A DAILY ensemble of n LHS samples beside the headline (round 5's verdict: daily data is the reference's normal use
    -- examples/in/ExampleDaily -- and no number existed above 1e4 samples): ten years of daily steps, a report every
    step, objective functions fused, no matrix stored.  With daily steps the default ranges put 11.6 % of the rows into
    the literal arithmetic (dt / RK > 2) and 20 % into the stiff variant; the launch picks the literal kernel's form from
    the load (smart_describe_launch says which).  48 rows of the timed launch, every class among them, against the
    oracle."""
    forcing, rng = synthetic_forcing(0, hourly=False)
    T, W, dt, gap = forcing.shape[0], WARM_DAYS, 86400.0, 1
    truth, _ = truth_discharge(so, forcing, dt, T, W, gap, False, from_fixture)
    obs = synthetic_observations(truth, rng, T)
    params = latin_hypercube(n, Parameters().ranges, seed=2718)
    prep = engine.prepare_ensemble(torch.from_numpy(params).to(device), forcing, AREA, dt, W, gap, obs=obs, gw_obs=GW_OBS,
                                   extra=EXTRA, want_discharge=False, device=device)
    _, ms, _ = timed_steps(prep.launch, steps, 1, device)
    res = prep.verify()
    cls = engine.variant_classes(torch.from_numpy(params), dt).numpy()
    leg = {'what': '%d-sample LHS ensemble x daily 10-yr synthetic forcing (T=%d + warm-up %d steps), a report every step, '
                   'objective functions fused, discharge not stored' % (n, T, W),
           'kernel': prep.describe(), 'launch_ms': ms, 'value': n * (T + W) / (ms * 1e-3), 'unit': 'sample-timesteps/s',
           'steps': steps, 'rows_per_class': {'regular': int((cls == 0).sum()), 'stiff': int((cls == 1).sum()),
                                              'guard': int((cls == 2).sum()), 'literal': int((cls == 3).sum())},
           # the launch is three kernels side by side: their vector instructions summed (tools/daily_roofline.py, from the
           # committed PMC passes of this workload) x 4 issue cycles over the SIMD cycles of this run's launch
           'roofline': leg_roofline('daily_%s' % ('1e6' if n == 1000000 else n), ms)}
    if so is not None:
        pick = np.random.default_rng(61)
        rows = np.sort(np.concatenate([pick.choice(np.nonzero(cls == c)[0], size=min(16, int((cls == c).sum())),
                                                   replace=False) for c in range(4) if (cls == c).any()]))
        leg['parity'] = dict(rows_against_the_oracle(so, res, rows, params, forcing, AREA, obs, W, gap, dt), rows=len(rows))
        leg['parity']['ok'] = bool(leg['parity']['max_abs_gw_ratio'] <= PARITY_GATE
                                   and leg['parity']['max_rel_objfn'] <= OBJFN_GATE and leg['parity']['gw_flag_equal'])
    return leg


def timed_steps(step, n_steps, n_warmup, device):
    """W untimed steps, then K steps between barrier + synchronize on both sides.  Returns (wall seconds, max over
    ranks; mean HIP-event milliseconds of a step on the stream the kernels are launched on)."""
    for _ in range(n_warmup):
        step()
    sdist.barrier()
    torch.cuda.synchronize(device)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_steps)]
    t0 = time.perf_counter()
    for k in range(n_steps):
        ev[k][0].record()
        res = step()
        ev[k][1].record()
    torch.cuda.synchronize(device)
    sdist.barrier()
    elapsed = sdist.max_over_ranks(time.perf_counter() - t0, device)
    return elapsed, float(np.mean([a.elapsed_time(b) for a, b in ev])), res


def rank_evidence(device, launch_ms):
    """Who took part: backend, and per rank the device it ran on and the mean HIP-event time of its own steps --
    gathered once, after the timed loop.  A line that says n_gpus = 8 then shows eight devices and eight timings."""
    import torch.distributed as dist
    props = torch.cuda.get_device_properties(device)
    mine = {'rank': sdist.rank_world()[0], 'device': '%s #%d' % (props.name, device.index),
            'current_device': int(torch.cuda.current_device()), 'local_rank': sdist.env_world()[2],
            'uuid': str(getattr(props, 'uuid', '')), 'pci_bus_id': int(getattr(props, 'pci_bus_id', -1)),
            'host': os.uname().nodename, 'pid': os.getpid(), 'launch_ms': launch_ms}
    try:        # the collective library torch was built against: RCCL's version on ROCm
        rccl = '.'.join(str(v) for v in torch.cuda.nccl.version())
    except Exception:       # noqa: BLE001
        rccl = None
    lib = {'rccl_version': rccl, 'hip': getattr(torch.version, 'hip', None), 'torch': torch.__version__}
    if not sdist.is_distributed():
        return dict(lib, backend=None, world_size=1, devices=[mine['device']], launch_ms_per_rank=[launch_ms],
                    ranks=[mine])
    everyone = [None] * dist.get_world_size()
    dist.all_gather_object(everyone, mine)
    return dict(lib, backend=sdist.data_backend(), rccl_failure=sdist.rccl_failure, group_backend=str(dist.get_backend()),
                world_size=dist.get_world_size(),
                devices=[r['device'] for r in everyone], launch_ms_per_rank=[r['launch_ms'] for r in everyone],
                ranks=everyone)


def spawn_ranks(n_gpus, argv):
    """`python bench.py --gpus N` started bare (no WORLD_SIZE in the environment): start the N ranks ourselves, the way
    the driver's launcher would -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
    127.0.0.1 --master-port <free> bench.py <same arguments>` -- as a CHILD process, before this process has made any
    GPU call (a process that has initialised the GPU must not exec another program on this pool, and need not: it only
    relays).  Rank 0's JSON line goes to stdout as it is, anything else the ranks print to stderr; returns the
    launcher's return code.  The reference's counterpart: `mpirun -np N python script.py` around spotpy's
    parallel='mpi' (montecarlo.py:153-154, docs/_doc_src/tutorial/montecarlo_experiment.rst:129-141)."""
    import socket
    import subprocess

    def start(extra_env):
        with socket.socket() as sock:
            sock.bind(('127.0.0.1', 0))
            port = sock.getsockname()[1]
        env = dict(os.environ)
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')   # dmabuf IPC: RCCL across processes needs it on this driver
        env.update(extra_env)
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n_gpus),
               '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + list(argv)
        child = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, cwd=os.getcwd())
        printed = False
        for raw in child.stdout:
            text = raw.decode('utf8', 'replace')
            printed = printed or text.startswith('{')
            out = sys.stdout if text.startswith('{') else sys.stderr
            out.write(text)
            out.flush()
        return child.wait(), printed

    rc, printed = start({})
    if rc != 0 and not printed and os.environ.get('SMART_DIST_BACKEND') is None:
        # The ranks ended without a line: whatever it was (an RCCL that aborted its process, a watchdog), this launcher
        # has made no GPU call and can start a FRESH job whose result blocks travel through the host (72 bytes per
        # sample once per step: the path does not need RCCL to be measured) -- and the line says why it did.
        why = 'the ranks started over RCCL exited with code %d before printing a line; restarted with ' \
              'SMART_DIST_BACKEND=gloo' % rc
        sys.stderr.write('bench.py: %s\n' % why)
        rc, printed = start({'SMART_DIST_BACKEND': 'gloo', 'SMART_DIST_RCCL_FAILURE': why})
    return rc


def arm_deadline():
    """Whatever hangs -- a collective, a driver call -- the run ends with a reason and a non-zero code after
    SMART_BENCH_DEADLINE seconds (1,500) instead of sitting silent until somebody's watchdog fires."""
    import threading
    limit = float(os.environ.get('SMART_BENCH_DEADLINE', '1500'))

    def fire():
        sys.stderr.write('bench.py: no result after %.0f s (SMART_BENCH_DEADLINE): rank %s gives up\n'
                         % (limit, os.environ.get('RANK', '0')))
        sys.stderr.flush()
        os._exit(3)
    t = threading.Timer(limit, fire)
    t.daemon = True
    t.start()
    return t


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--config', type=int, default=3, choices=[2, 3, 4, 5], help='BASELINE.json configuration')
    ap.add_argument('--samples', type=int, default=None, help='override the sample count of the configuration')
    ap.add_argument('--math', default='fast', choices=['fast', 'literal'])
    ap.add_argument('--no-discharge', action='store_true', help='do not write the [R, N] discharge matrix')
    ap.add_argument('--no-cpu-baseline', action='store_true', help='skip cpu_baseline and the in-run parity check')
    ap.add_argument('--no-flat', action='store_true', help='skip the flat_forcing / runs_of_6 / objectives_only legs')
    ap.add_argument('--no-strong', action='store_true', help='skip the strong_1e6 leg (config 4 beside config 3)')
    ap.add_argument('--no-daily', action='store_true', help='skip the daily_1e6 leg (a daily ensemble beside config 3)')
    ap.add_argument('--obs-from-fixture', action='store_true',
                    help='observations from tests/golden/bench_truth.npz instead of a run of the oracle (automatic when '
                         'the oracle cannot be built: a box without gcc)')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:    # started bare: this process becomes the launcher
        raise SystemExit(spawn_ranks(args.gpus, sys.argv[1:]))
    arm_deadline()
    rank, world, device = sdist.init()
    if world != args.gpus:
        raise SystemExit('bench.py: --gpus %d but WORLD_SIZE is %d' % (args.gpus, world))
    if device.type != 'cuda':
        raise SystemExit('bench.py needs a GPU: the engine has no CPU path')

    cfg = args.config
    hourly = cfg != 2
    dt, gap = (3600.0, 24) if hourly else (86400.0, 1)
    W = WARM_DAYS * (24 if hourly else 1)
    forcing, rng = synthetic_forcing(0, hourly=hourly)
    T = forcing.shape[0]
    R = T // gap
    ranges = Parameters().ranges
    d_forcing = torch.from_numpy(forcing).to(device)

    # observations: the CPU twin's discharge of the "truth" parameter set x lognormal noise, 12 % NaN.
    # Input data of the benchmark, made before anything is timed by the same checker the cpu_baseline leg times (round 3
    # took them from a run of the engine itself: circular) -- on RANK 0 ONLY, and broadcast (29 KB): the checker is a
    # gcc-built library next to its source, and N ranks of a freshly pushed tree building and loading it side by side
    # is a race the benchmark has no business running (round 4's advisor; oracle.smart_oracle.build() is atomic as well)
    obs, so, no_oracle, obs_source = None, None, None, None
    if rank == 0:
        so, no_oracle = load_oracle()
        truth, obs_source = truth_discharge(so, forcing, dt, T, W, gap, hourly, args.obs_from_fixture)
        obs = synthetic_observations(truth, rng, R)
    obs = sdist.broadcast_matrix(obs if rank == 0 else np.empty(0), src=0)
    if rank != 0:       # (behind the broadcast: rank 0 has built the library, the others only load it)
        so, no_oracle = load_oracle()

    store = not args.no_discharge and cfg in (2, 3)     # configs 4 and 5 gather objective functions only
    kw = dict(extra=EXTRA, math_mode=args.math, want_discharge=store, device=device)
    if cfg in (2, 3):
        n_local = args.samples or (100000 if cfg == 3 else 10000)
        n_runs_total = n_local * world
        params = latin_hypercube(n_local, ranges, seed=2718 + rank)       # this rank's own block of the ensemble
        d_params = torch.from_numpy(params).to(device)
        job = sdist.ShardedEnsemble(d_params, d_forcing, AREA, dt, W, gap, axis='samples', local_block=True,
                                    obs=obs, gw_obs=GW_OBS, **kw)
        scaling, shard = 'weak', 'sample-shard x%d' % world
        what = '%d-sample LHS ensemble per GPU x %s 10-yr synthetic forcing' % (n_local, 'hourly' if hourly else 'daily')
        blocks_local = math.ceil(n_local / 64)
    elif cfg == 4:
        n_total = args.samples or 1000000
        params = latin_hypercube(n_total, ranges, seed=2718)              # the same matrix on every rank, cut by rows
        lo, hi = sdist.shard_bounds(n_total, world, rank)
        d_params = torch.from_numpy(params[lo:hi]).to(device)
        job = sdist.ShardedEnsemble(torch.from_numpy(params).to(device), d_forcing, AREA, dt, W, gap, axis='samples',
                                    obs=obs, gw_obs=GW_OBS, **kw)
        n_runs_total, n_local = n_total, hi - lo
        scaling, shard = 'strong', 'sample-shard x%d' % world
        what = '%d-sample LHS ensemble in total (%d per GPU) x hourly 10-yr synthetic forcing' % (n_total, n_local)
        blocks_local = math.ceil(n_local / 64)
    else:
        C, n_per = 64, args.samples or 10000
        params = latin_hypercube(n_per, ranges, seed=2718)
        d_params = torch.from_numpy(params).to(device)
        areas = np.exp(np.random.default_rng(12345).uniform(np.log(20e6), np.log(2000e6), C))
        obs_c = np.tile(obs, (C, 1))
        job = sdist.ShardedEnsemble(torch.from_numpy(params).to(device), lambda c: synthetic_forcing(c, True)[0],
                                    areas, dt, W, gap, axis='catchments', n_catchments=C, obs=obs_c, gw_obs=GW_OBS,
                                    **kw)
        n_runs_total, n_local = C * n_per, job.n_local * n_per
        scaling, shard = 'strong', 'catchment-shard x%d' % world
        what = '%d synthetic catchments x %d samples each in total (%d catchments per GPU) x hourly 10-yr forcing' \
               % (C, n_per, job.n_local)
        blocks_local = math.ceil(n_per / 64) * job.n_local

    elapsed, launch_ms, res = timed_steps(job.step, args.steps, args.warmup, device)
    again = job.verify()                 # status word of the last launch; a repeated launch is gathered afresh
    res = res if again is None else again
    assert bool(torch.isfinite(res[..., :7]).all())
    ranks = rank_evidence(device, launch_ms)
    # every rank, on rows of the job it has just timed (cfg 5: four of its catchments)
    if cfg == 5:
        timed = timed_launch_parity(so, no_oracle, job, 'catchments', params, lambda c: synthetic_forcing(c, True)[0],
                                    lambda c: float(areas[c]), obs, W, gap, dt)
    else:
        timed = timed_launch_parity(so, no_oracle, job, 'samples', params[lo:hi] if cfg == 4 else params,
                                    lambda c: forcing, lambda c: AREA, obs, W, gap, dt)

    # config 4's strong-scaled figure in the same line (SURVEY.md 7.3-3 asks for both series from the driver's runs):
    # 1e6 samples IN TOTAL, cut by rows over the ranks, objective functions only, one all-gather
    strong = None
    if cfg == 3 and not args.no_strong and args.math == 'fast':
        n_total = 1000000
        p_all = latin_hypercube(n_total, ranges, seed=2718)
        lo, hi = sdist.shard_bounds(n_total, world, rank)
        sjob = sdist.ShardedEnsemble(torch.from_numpy(p_all).to(device), d_forcing, AREA, dt, W, gap, axis='samples',
                                     obs=obs, gw_obs=GW_OBS, **dict(kw, want_discharge=False))
        s_steps = max(2, args.steps // 4)
        s_elapsed, s_ms, s_res = timed_steps(sjob.step, s_steps, 1, device)
        s_again = sjob.verify()
        assert bool(torch.isfinite((s_res if s_again is None else s_again)[..., :7]).all())
        strong = {'what': 'configs[3]: 1e6-sample LHS ensemble in total, sample-sharded over the %d GPU(s), gather of '
                          '[N, 9]; discharge not stored' % world,
                  'value': n_total * (W + T) * s_steps / s_elapsed, 'unit': 'sample-timesteps/s', 'scaling': 'strong',
                  'ms_per_step': s_elapsed / s_steps * 1e3, 'steps': s_steps, 'runs_total': n_total,
                  'runs_per_gpu': hi - lo, 'kernel': sjob.prepared.describe(),
                  'launch_ms_per_rank': rank_evidence(device, s_ms)['launch_ms_per_rank']}
        del sjob, p_all

    # Every collective of the run lies behind this line: what follows on rank 0 (the legs, the CPU baseline: tens of
    # seconds of host work) is its own business, and its peers do not wait for it -- a wait would be bounded by the
    # group's timeout (SMART_DIST_TIMEOUT), and a slow host must not turn into "a peer never arrived".  The verdict of
    # the timed launch's check is the same number on every rank (a reduction): a failure there is every rank's exit code.
    sdist.barrier()
    failed = int(bool(timed.get('timed_launch')) and not timed['timed_launch']['ok'])
    if rank == 0:
        steps_per_run = W + T
        units_per_step = n_runs_total * steps_per_run            # executed sample-timesteps, all ranks, per step
        units_per_launch = n_local * steps_per_run               # ... of rank 0's launch
        value = units_per_step * args.steps / elapsed
        w = wet_fraction(forcing, W)
        flops_per_step = 105.0 + 121.0 * w                        # literal operation count
        bytes_per_step = (8.0 / gap * T / (W + T) if store else 0.0) + 16.0 / max(n_local, 1) \
            + (80.0 + 64.0 + 8.0) / (W + T)
        kern_s = launch_ms * 1e-3
        kernels = job.prepared.describe()
        pmc_key = 'config%d:runs_per_gpu=%d:discharge=%d:math=%s' % (cfg, n_local, int(store), args.math)

        # ---- roofline of the dominant kernel.  The binding roof is fp64 vector-ALU issue:
        # frac = vector instructions the kernel executes x 4 issue cycles / the issue cycles 1,024 SIMDs have in one
        # launch.  The instruction count is a property of (kernel build, workload): it comes from the committed PMC
        # summary of this same command and is only quoted while the kernel sources hash to what was profiled.
        pmc, pmc_note = pmc_summary(pmc_key)
        traffic, insts = pmc.get('hbm_bytes_per_launch'), pmc.get('valu_insts_per_launch')
        held_clock, held_frac = pmc.get('held_clock_hz'), pmc.get('issue_frac_at_held_clock')
        model, model_note = isa_model_summary(pmc_key)
        insts_model = model.get('valu_insts_per_launch')
        # the share of the vector instructions that are fp64 ARITHMETIC (fma / add / mul / min / max / ldexp): the
        # instruction model's, from the compiler's assembly; the PMC summary's figure (the same model over the MEASURED
        # instruction count) when there is no model for this build.  The counters' own share (FMA + ADD + MUL classes
        # only: min / max / ldexp are in none of them) is executed_flops' business, not this one's.
        fp64_share = model.get('fp64_share_of_valu', pmc.get('fp64_share_of_valu'))
        flops = pmc.get('fp64_flops_per_launch')
        issue = None if insts is None else insts * VALU_ISSUE_CYCLES / (N_SIMD * kern_s * CLOCK_HZ)
        roofline = {
            'bound': 'valu-fp64-issue',
            'achieved': None if insts is None else insts / kern_s / 1e12,
            'peak': N_SIMD * CLOCK_HZ / VALU_ISSUE_CYCLES / 1e12,
            'unit': 'T wave-instructions/s',
            'frac': issue,
            # ... of which fp64 ARITHMETIC (fma / add / mul / min / max / ldexp; no moves, compares, selects): the same
            # fraction times the arithmetic's share of the vector instructions, from the compiler's assembly of the
            # hot loop weighed with this workload's path frequencies (tools/isa_model.py, profiles/r03_isa_model_*.md)
            'useful_frac': None if issue is None or fp64_share is None else issue * fp64_share,
            'fp64_share_of_valu': fp64_share,
            # the flops the kernel EXECUTES, counted (SQ_INSTS_VALU_{FMA x 2, ADD, MUL}_F64 x 64 lanes; min / max / ldexp
            # and compares are in none of these counters), over this run's launch time, against the fp64 vector peak --
            # which is all-FMA: a kernel that issued an add or a mul on every cycle would read 0.5
            'executed_flops': None if flops is None else {
                'per_launch': flops, 'per_sample_step': flops / units_per_launch, 'tflops': flops / kern_s / 1e12,
                'peak_tflops': FP64_VALU_PEAK_TFLOPS, 'frac': flops / kern_s / 1e12 / FP64_VALU_PEAK_TFLOPS},
            'clock_basis_hz': CLOCK_HZ,
            # the same fraction over the shader cycles the chip actually ran (it lowers its clock under this load):
            # taken whole from the profiled runs (SQ_INSTS_VALU x 4 / (1,024 SIMDs x GRBM_GUI_ACTIVE / 8)), not mixed
            # with this run's timing
            'frac_at_held_clock': held_frac, 'held_clock_hz': held_clock,
            'valu_insts_per_launch': insts,
            'valu_insts_per_wave_step': None if insts is None else insts / (blocks_local * steps_per_run),
            # the same count derived from the tree alone (tools/isa_model.py): the two agree within 3 % or the fraction
            # above is not to be trusted (tests/test_isa_model.py holds them against each other)
            'valu_insts_model': insts_model,
            'valu_insts_model_per_wave_step': None if insts_model is None else insts_model / (blocks_local * steps_per_run),
            'valu_insts_model_vs_pmc': None if insts is None or insts_model is None else insts_model / insts,
            'frac_from_model': None if insts_model is None else
            insts_model * VALU_ISSUE_CYCLES / (N_SIMD * kern_s * CLOCK_HZ),
            'model_source': model_note,
            'kernel': kernels, 'launch_ms': launch_ms, 'traffic': traffic,
            # counters over algorithm: HBM bytes per launch as the PMC passes count them (FETCH_SIZE with the guide's gfx950
            # correction + WRITE_SIZE) against the algorithmic bytes of `hbm` below; the excess is the time slices'
            # hand-overs (1.3 x with the matrix stored; without one the hand-overs are all there is: 20 - 40 x of very little)
            'traffic_ratio': None if traffic is None else traffic / (units_per_launch * bytes_per_step),
            'pmc_key': pmc_key, 'pmc_source': pmc_note,
            # the reference's literal operation count against the fp64 vector peak: NOT a bound (the kernel executes
            # fewer operations than the reference writes down), kept as the algorithmic ratio
            'algorithmic_ratio': {'note': 'not a bound: the reference\'s operation count over the time of a kernel '
                                          'that executes fewer operations',
                                  'flops_per_sample_step': flops_per_step,
                                  'tflops': units_per_launch * flops_per_step / kern_s / 1e12,
                                  'peak_tflops': FP64_VALU_PEAK_TFLOPS,
                                  'ratio': units_per_launch * flops_per_step / kern_s / 1e12 / FP64_VALU_PEAK_TFLOPS},
            'hbm': {'bound': 'hbm', 'achieved': units_per_launch * bytes_per_step / kern_s / 1e9,
                    'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                    'frac': units_per_launch * bytes_per_step / kern_s / 1e9 / HBM_PEAK_GBS,
                    'bytes_per_sample_step': bytes_per_step},
        }
        line = {
            'metric': 'MC sample-timesteps/sec/GPU; 1e5 LHS x hourly 10-yr forcing',
            'value': value, 'unit': 'sample-timesteps/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True,
            'scaling': scaling, 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': 'configs[%d]: %s (T=%d + warm-up %d steps), summary report gap %d, objective '
                                   'functions fused, discharge %s; forcing constant within each report interval '
                                   '(daily values spread over the steps of the day, the reference\'s own input format)'
                                   % (cfg - 1, what, T, W, gap, 'stored [R,N]' if store else 'not stored'),
                       'runs_total': n_runs_total, 'runs_per_gpu': n_local, 'n_steps': T, 'n_warm': W,
                       'math_mode': args.math, 'wet_fraction': w, 'parallelism': shard},
            'per_gpu': value / world,
            # which scaling series this line belongs to; north_star's ">= 7x at 8 GPUs" reads on `value` for the weak
            # series (per-GPU work fixed) and on strong_1e6.value for the strong one (1e6 samples in total, configs[3])
            'series': {'value': '%s: %s' % (scaling, 'configs[%d]' % (cfg - 1)),
                       'strong_1e6': 'strong: configs[3], 1e6 samples in total over the %d GPU(s)' % world,
                       'target': '>= 7x the n_gpus=1 figure of the same series at n_gpus=8 (BASELINE.json north_star)'},
            'ranks': ranks,
            # the same rate counting the simulated steps only (the warm-up replays W of them)
            'value_without_warmup': n_runs_total * T * args.steps / elapsed,
            'roofline': roofline,
        }
        if world == 1 and cfg in (3, 4) and not args.no_flat and args.math == 'fast':
            # the same workload with forcing that varies inside the report interval: the step loop (smart_fast_steps)
            vary = hourly_varying_forcing(forcing)
            flat = engine.prepare_ensemble(d_params, vary, AREA, dt, W, gap, obs=obs, gw_obs=GW_OBS, **kw)
            f_elapsed, f_ms, _ = timed_steps(flat.launch, max(2, args.steps // 2), 1, device)
            flat.verify()
            line['flat_forcing'] = {
                'what': 'same runs, daily totals spread unevenly over the hours (rain in ~6 random hours, PE on a '
                        'daytime sine): forcing varies inside the report interval, the interval engine does not apply',
                'kernel': flat.describe(), 'launch_ms': f_ms, 'value': units_per_launch / (f_ms * 1e-3),
                'unit': 'sample-timesteps/s', 'wet_fraction': wet_fraction(vary, W),
                'roofline': leg_roofline('flat_forcing', f_ms)}
            del flat
        if strong is not None:
            line['strong_1e6'] = strong
        if world == 1 and cfg == 3 and not args.no_flat and args.math == 'fast':
            # 6-hourly data in the hourly run: forcing constant over runs of six steps (the run engine)
            six = six_hourly_forcing(forcing)
            runs = engine.prepare_ensemble(d_params, six, AREA, dt, W, gap, obs=obs, gw_obs=GW_OBS, **kw)
            r_elapsed, r_ms, _ = timed_steps(runs.launch, max(2, args.steps // 2), 1, device)
            runs.verify()
            line['runs_of_6'] = {
                'what': 'same runs, daily totals as four 6-hour values spread equally over their six steps (6-hourly '
                        'input files in an hourly run, timeframe.py:167-186): the interval engine over runs of 6 steps',
                'kernel': runs.describe(), 'launch_ms': r_ms, 'value': units_per_launch / (r_ms * 1e-3),
                'unit': 'sample-timesteps/s', 'wet_fraction': wet_fraction(six, W),
                'roofline': leg_roofline('runs_of_6', r_ms)}
            del runs
        if world == 1 and cfg == 3 and store and not args.no_flat and args.math == 'fast':
            # the same runs the way MonteCarlo.run() launches them by default (save_sim=False): objective functions and
            # groundwater ratios only, no discharge matrix -- which lets the engine order the rows
            lean = engine.prepare_ensemble(d_params, d_forcing, AREA, dt, W, gap, obs=obs, gw_obs=GW_OBS,
                                           **dict(kw, want_discharge=False))
            l_elapsed, l_ms, _ = timed_steps(lean.launch, max(2, args.steps // 2), 1, device)
            lean.verify()
            line['objectives_only'] = {
                'what': 'same runs without the stored discharge matrix (MonteCarlo.run with save_sim=False): rows '
                        'ordered by T and S*Z before the launch, results permuted back',
                'kernel': lean.describe(), 'launch_ms': l_ms, 'value': units_per_launch / (l_ms * 1e-3),
                'unit': 'sample-timesteps/s'}
            del lean
        if world == 1 and cfg == 3 and not args.no_flat and args.math == 'fast':
            # the two report modes of the reference that are not interval means (structure.py:192-195 and :190 with a
            # report every step), on the headline's runs and forcing, objective functions fused, no matrix stored.
            # Round 3 ran both through the general step loop, unsliced (24.8 / 43.7 ms)
            raw = engine.prepare_ensemble(d_params, d_forcing, AREA, dt, W, gap, obs=obs, gw_obs=GW_OBS, report='raw',
                                          **dict(kw, want_discharge=False))
            _, w_ms, _ = timed_steps(raw.launch, max(2, args.steps // 2), 1, device)
            raw.verify()
            line['raw_gap24'] = {
                'what': "same runs with report='raw' (the outflow of each day's last hour, groundwater ratio from those "
                        "steps only): the interval engine over 23 + 1 steps",
                'kernel': raw.describe(), 'launch_ms': w_ms, 'value': units_per_launch / (w_ms * 1e-3),
                'unit': 'sample-timesteps/s', 'roofline': leg_roofline('raw_gap24', w_ms)}
            del raw
            every = engine.prepare_ensemble(d_params, d_forcing, AREA, dt, W, 1, obs=np.repeat(obs, gap),
                                            gw_obs=GW_OBS, **dict(kw, want_discharge=False))
            _, e_ms, _ = timed_steps(every.launch, max(2, args.steps // 2), 1, device)
            every.verify()
            line['gap1'] = {
                'what': 'same runs with a report every step (hourly reports of the hourly run, %d observations): the '
                        'run as a stream of records, arm and report in one asm' % (R * gap),
                'kernel': every.describe(), 'launch_ms': e_ms, 'value': units_per_launch / (e_ms * 1e-3),
                'unit': 'sample-timesteps/s', 'roofline': leg_roofline('gap1', e_ms)}
            del every
        if world == 1 and cfg == 3 and not args.no_flat and args.math == 'fast':
            line.update(api_legs(forcing, obs, WARM_DAYS, device))
        if world == 1 and cfg == 3 and not args.no_daily and not args.no_flat and args.math == 'fast':
            line['daily_1e6'] = daily_leg(so, 1000000, device, max(2, args.steps // 4), args.obs_from_fixture)
        line['observations'] = obs_source
        if args.no_cpu_baseline:
            line['parity'] = dict(timed)
        elif so is None:
            # no checker on this box (no gcc and no built oracle): the throughput line stands, the baseline does not
            line['cpu_baseline'], line['parity'] = None, dict(timed, why='oracle unavailable: %s' % no_oracle)
        else:
            # rank 0's host cores and rank 0's GPU, whatever the world size
            line['cpu_baseline'], line['parity'] = cpu_baseline_and_parity(so, forcing, W, gap, dt, device)
            line['parity'].update(timed)
        print(json.dumps(line), flush=True)
        bad = [k for k in ('ok',) if line['parity'].get(k) is False]
        if line['parity'].get('timed_launch') and not line['parity']['timed_launch']['ok']:
            bad.append('timed_launch')
        if line.get('daily_1e6', {}).get('parity', {}).get('ok') is False:
            bad.append('daily_1e6')
        if bad:
            sys.stdout.flush()
            sys.stderr.write('bench.py: in-run parity check failed (%s): %r\n' % (', '.join(bad), line['parity']))
            failed = 1
    sdist.finish(failed)
    if failed:
        raise SystemExit(failed)


if __name__ == '__main__':
    main()
