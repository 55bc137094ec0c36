#!/usr/bin/env python3
"""Headline benchmark: Monte-Carlo sample-timesteps per second of the SMART time loop on MI355X.

Contract: `python bench.py --gpus N --steps K --warmup W` (N > 1: launched by torch.distributed.run, one rank per
GPU, RCCL).  One "step" = one pass of the hot path over one batch = one ensemble launch that advances every
sample of this rank's shard through warm-up + simulation (BASELINE config 3: 1e5 LHS samples x hourly 10-year
synthetic forcing per GPU, summary report, objective functions fused, discharge matrix written) followed, for
N > 1, by the one RCCL all-gather of the objective / groundwater arrays.  Weak scaling: 1e5 samples per GPU.

Rank 0 prints ONE JSON line; see DESIGN.md "Measurement" for every field.
"""
import argparse
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


def wet_fraction(forcing, n_warm, t_lo=0.9, t_hi=1.1):
    """Realised fraction of executed sample-steps on the wet branch (rain * T - peva >= 0, T ~ U[t_lo, t_hi])."""
    f = np.concatenate([forcing[:n_warm], forcing])
    rain, pe = f[:, 0], f[:, 1]
    with np.errstate(divide='ignore', invalid='ignore'):
        thr = np.where(rain > 0, pe / rain, np.where(pe <= 0, -np.inf, np.inf))
    return float(np.mean(np.clip((t_hi - thr) / (t_hi - t_lo), 0.0, 1.0)))


def cpu_baseline(forcing, n_warm, gap, budget_s=12.0):
    """The oracle's OpenMP batch runner (a C port of the reference loop) on the host cores, on a bounded sample of
    the same workload.  Reported next to the GPU number; never part of the timed GPU region."""
    from oracle import smart_oracle as so
    cores = so.max_threads()
    rain, pe = np.ascontiguousarray(forcing[:, 0]), np.ascontiguousarray(forcing[:, 1])
    T = len(rain)
    ranges = Parameters().ranges

    def run(n):
        p = latin_hypercube(n, ranges, seed=99)
        t0 = time.perf_counter()
        so.run_batch(AREA, 3600.0, T, n_warm, rain, pe, p, EXTRA, so.REPORT_SUMMARY, gap, want_discharge=True,
                     n_threads=cores)
        return time.perf_counter() - t0

    run(2 * cores)                      # first touch of the per-thread tables, thread start-up
    probe_n = 8 * cores
    probe = run(probe_n)
    n = int(max(probe_n, min(4096, probe_n * budget_s / max(probe, 1e-3))))
    n -= n % cores
    dt = run(n)
    return {'value': n * (T + n_warm) / dt, 'unit': 'sample-timesteps/s', 'cores': cores, 'kind': 'port',
            'sample': '%d LHS samples x %d steps (hourly 10 yr + 365 d warm-up), oracle/smart_oracle.c with OpenMP, '
                      '%.1f s' % (n, T + n_warm, dt),
            'reference_python_1core': 6.9e4}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--samples', type=int, default=100000, help='samples per GPU')
    ap.add_argument('--math', default='fast', choices=['fast', 'literal'])
    ap.add_argument('--no-discharge', action='store_true', help='do not write the [R, N] discharge matrix')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--sort-col', type=int, default=-1, help='experiment: sort the sample rows by this parameter column')
    args = ap.parse_args()

    rank, world, device = sdist.init()
    if world != args.gpus:
        raise SystemExit('bench.py: --gpus %d but WORLD_SIZE is %d (launch with torch.distributed.run)'
                         % (args.gpus, world))
    if device.type != 'cuda':
        raise SystemExit('bench.py needs a GPU: the engine has no CPU path')

    forcing, rng = synthetic_forcing(0, hourly=True)
    T, W, gap = forcing.shape[0], WARM_DAYS * 24, 24
    n_local = args.samples
    ranges = Parameters().ranges
    params = latin_hypercube(n_local, ranges, seed=2718 + rank)       # this rank's shard of the ensemble
    if args.sort_col >= 0:
        params = params[np.argsort(params[:, args.sort_col], kind='stable')]
    d_forcing = torch.from_numpy(forcing).to(device)
    d_params = torch.from_numpy(params).to(device)

    # observations: discharge of the "truth" parameter set (computed by the engine itself) x lognormal noise, 12 % NaN
    truth = engine.run_ensemble(np.array([TRUTH]), d_forcing, AREA, 3600.0, W, gap, extra=EXTRA, device=device)
    obs = truth.discharge[0].cpu().numpy() * np.exp(rng.normal(0.0, 0.2, T // gap))
    obs[rng.random(T // gap) < 0.12] = np.nan
    d_obs = torch.from_numpy(obs).to(device)
    R = T // gap
    d_dis = None if args.no_discharge else torch.empty((1, R, n_local), dtype=torch.float64, device=device)

    def one_step():
        out = engine.run_ensemble(d_params, d_forcing, AREA, 3600.0, W, gap, extra=EXTRA, obs=d_obs, gw_obs=0.12667,
                                  math_mode=args.math, want_discharge=False, discharge_out=d_dis, device=device)
        if world > 1:   # the path's only exchange: objective functions + gw of every shard, one all-gather
            res = torch.cat([out.objfn, out.gw.unsqueeze(1)], dim=1)
            return sdist.gather_rows(res, n_local * world)
        return out.objfn

    for _ in range(args.warmup):
        one_step()
    sdist.barrier()
    torch.cuda.synchronize(device)
    # HIP events on the stream the kernels are launched on (the engine launches on torch's current stream)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        ev[k][0].record()
        res = one_step()
        ev[k][1].record()
    torch.cuda.synchronize(device)
    sdist.barrier()
    elapsed = time.perf_counter() - t0
    elapsed = sdist.max_over_ranks(elapsed, device)
    launch_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    assert bool(torch.isfinite(res[:, :7]).all())

    if rank == 0:
        steps_per_launch = n_local * (W + T)                      # executed sample-timesteps per rank per step
        value = world * steps_per_launch * args.steps / elapsed
        w = wet_fraction(forcing, W)
        flops_per_step = 105.0 + 121.0 * w                         # literal operation count
        bytes_per_step = (0.0 if args.no_discharge else 8.0 / gap * T / (W + T)) \
            + 16.0 / n_local + (80.0 + 64.0 + 8.0) / (W + T)
        kern_s = launch_ms * 1e-3
        traffic = executed = None
        tpath = os.path.join(ROOT, 'profiles', 'traffic_latest.json')
        if os.path.exists(tpath):
            with open(tpath) as fh:
                pmc = json.load(fh)
            traffic = pmc.get('hbm_bytes_per_launch')
            if pmc.get('valu_insts_per_launch') and n_local == 100000 and not args.no_discharge and args.math == 'fast':
                # executed view: vector-ALU wave-instructions (PMC SQ_INSTS_VALU of the committed profile of this
                # same command) x 4 issue cycles, over the issue cycles the chip's 1,024 SIMDs have in one launch
                insts = float(pmc['valu_insts_per_launch'])
                executed = {'valu_insts_per_launch': insts,
                            'valu_insts_per_wave_step': insts / (math.ceil(n_local / 64) * (W + T)),
                            'issue_frac': insts * 4.0 / (N_SIMD * kern_s * CLOCK_HZ),
                            'source': 'profiles/traffic_latest.json'}
        line = {
            'metric': 'MC sample-timesteps/sec/GPU; 1e5 LHS x hourly 10-yr forcing',
            'value': value, 'unit': 'sample-timesteps/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': 'configs[2]: %d-sample LHS ensemble per GPU x hourly 10-yr synthetic forcing '
                                   '(T=%d + warm-up %d steps), summary report gap %d, objective functions fused, '
                                   'discharge %s' % (n_local, T, W, gap, 'not stored' if args.no_discharge
                                                     else 'stored [R,N]'),
                       'samples_per_gpu': n_local, 'n_steps': T, 'n_warm': W, 'math_mode': args.math,
                       'wet_fraction': w, 'parallelism': 'sample-shard x%d' % world},
            'per_gpu': value / world,
            # the same rate counting the simulated steps only (the warm-up replays W of them)
            'value_without_warmup': world * n_local * T * args.steps / elapsed,
            'roofline': {
                'bound': 'valu-fp64', 'achieved': steps_per_launch * flops_per_step / kern_s / 1e12,
                'peak': FP64_VALU_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                'frac': steps_per_launch * flops_per_step / kern_s / 1e12 / FP64_VALU_PEAK_TFLOPS,
                'flops_per_sample_step': flops_per_step, 'kernel': 'smart_ensemble_' + args.math,
                'launch_ms': launch_ms,
                'traffic': traffic,
                'executed': executed,
                'note': 'achieved = the reference\'s literal operation count F(w) = 105 + 121 w per sample-step '
                        '(SURVEY.md 8d) x sample-steps / launch time; the kernel executes fewer operations than '
                        'that count (DESIGN.md 4.1), so frac can exceed 1 -- `executed` is the instruction-issue view',
                'hbm': {'bound': 'hbm', 'achieved': steps_per_launch * bytes_per_step / kern_s / 1e9,
                        'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                        'frac': steps_per_launch * bytes_per_step / kern_s / 1e9 / HBM_PEAK_GBS,
                        'bytes_per_sample_step': bytes_per_step},
            },
        }
        if not args.no_cpu_baseline and world == 1:
            line['cpu_baseline'] = cpu_baseline(forcing, W, gap)
        elif not args.no_cpu_baseline:
            line['cpu_baseline'] = None
        print(json.dumps(line), flush=True)
    sdist.barrier()
    if sdist.is_distributed():
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
