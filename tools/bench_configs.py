#!/usr/bin/env python3
"""Kernel time of the other BASELINE.json configurations (parity-test cases, not the bench line)."""
import os, sys, json
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smartpy_amd import engine
from smartpy_amd.parameters import Parameters
from smartpy_amd.sampling import latin_hypercube

EXTRA = {'aar': 1200, 'r-o_ratio': 0.45, 'r-o_split': (0.10, 0.15, 0.15, 0.30, 0.30)}


def forcing(c, hourly):
    rng = np.random.default_rng(12345 + c)
    days = 3653
    rain = (rng.random(days) < 0.80) * rng.gamma(0.70, 4.57, days)
    pe = np.maximum(0.0, 1.47 * (1 + 0.85 * np.sin(2 * np.pi * ((np.arange(days) % 365.25) - 110) / 365.25)))
    if hourly:
        rain, pe = np.repeat(rain / 24, 24), np.repeat(pe / 24, 24)
    return np.stack([rain, pe], axis=1)


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


ranges = Parameters().ranges
out = []
# config 2: 1e4 samples, daily 10 yr (default ranges: every wavefront holds a dt/k > 2 row -> literal arithmetic)
p = torch.from_numpy(latin_hypercube(10000, ranges, seed=2718)).cuda()
f = torch.from_numpy(forcing(0, False)).cuda()
ms = timed(lambda: engine.run_ensemble(p, f, 175.46e6, 86400.0, 365, 1, extra=EXTRA))
out.append({'config': 'configs[1] 1e4 LHS x daily 10 yr (default ranges, ill-conditioned rows -> literal)', 'ms': ms,
            'steps_per_s': 10000 * 4018 / ms * 1e3})
stable = dict(ranges, SK=(12.0, 240.0), RK=(12.0, 96.0))
p2 = torch.from_numpy(latin_hypercube(10000, stable, seed=2718)).cuda()
ms = timed(lambda: engine.run_ensemble(p2, f, 175.46e6, 86400.0, 365, 1, extra=EXTRA))
out.append({'config': 'same with SK, RK >= 12 h (fast STIFF variant)', 'ms': ms, 'steps_per_s': 10000 * 4018 / ms * 1e3})
# config 5: 64 catchments x 1e4 samples, hourly 10 yr, objectives fused, no discharge
C = 64
fc = torch.from_numpy(np.stack([forcing(c, True) for c in range(C)])).cuda()
areas = np.exp(np.random.default_rng(99).uniform(np.log(20e6), np.log(2000e6), C))
obs = np.abs(np.random.default_rng(1).normal(2, 1, (C, 3653)))
ms = timed(lambda: engine.run_ensemble(p, fc, areas, 3600.0, 8760, 24, extra=EXTRA, obs=obs, gw_obs=0.12667,
                                       want_discharge=False), reps=2)
out.append({'config': 'configs[4] 64 catchments x 1e4 samples x hourly 10 yr, one launch, one GPU', 'ms': ms,
            'steps_per_s': C * 10000 * 96432 / ms * 1e3})
# config 4 per-GPU shard: 125,000 samples
p3 = torch.from_numpy(latin_hypercube(125000, ranges, seed=1)).cuda()
f1 = fc[0]
ms = timed(lambda: engine.run_ensemble(p3, f1, 175.46e6, 3600.0, 8760, 24, extra=EXTRA, obs=obs[0], gw_obs=0.12667,
                                       want_discharge=False))
out.append({'config': 'configs[3] per-GPU shard: 125,000 samples x hourly 10 yr', 'ms': ms,
            'steps_per_s': 125000 * 96432 / ms * 1e3})
for o in out:
    print(json.dumps(o))
