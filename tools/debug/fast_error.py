"""Largest relative difference of the fast mode from the reference-exact oracle over random LHS rows, headline forcing."""
import sys
sys.path.insert(0, '.')
import numpy as np
import torch
import bench
from smartpy_amd import engine, sampling
from smartpy_amd.parameters import Parameters
from oracle import smart_oracle as so
forcing = bench.synthetic_forcing(0, True)[0]
T, W = forcing.shape[0], 8760
extra = {'aar': 1200, 'r-o_ratio': 0.45, 'r-o_split': (0.10, 0.15, 0.15, 0.30, 0.30)}
n = 256
params = sampling.latin_hypercube(n, Parameters().ranges, seed=77)
out = engine.run_ensemble(params, forcing, 175.46e6, 3600.0, W, 24, extra=extra)
dis, gw, _ = so.run_batch(175.46e6, 3600.0, T, W, forcing[:, 0].copy(), forcing[:, 1].copy(), params, extra, so.REPORT_SUMMARY, 24)
got = out.discharge.cpu().numpy()
rel = np.abs(got - dis) / np.maximum(np.abs(dis), 1e-300)
print('discharge: max rel %.3e, 99.9th pct %.3e, median %.3e over %d values' % (rel.max(), np.quantile(rel, 0.999), np.median(rel), rel.size))
print('gw ratio : max abs %.3e' % np.max(np.abs(out.gw.cpu().numpy() - gw)))
