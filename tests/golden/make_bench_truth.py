"""bench.py --obs-from-fixture: the discharge of the benchmark's "truth" parameter set under the synthetic forcing of
BASELINE.md section 4 (hourly and daily runs, 3,653 daily values each), made by the oracle in the build container, so
that a box without a C compiler still gets the benchmark's observations (and its throughput line).
`python tests/golden/make_bench_truth.py` rewrites tests/golden/bench_truth.npz."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench                                        # noqa: E402
from oracle import smart_oracle as so               # noqa: E402

out = {}
for name, hourly in (('hourly', True), ('daily', False)):
    forcing, _ = bench.synthetic_forcing(0, hourly=hourly)
    dt, gap = (3600.0, 24) if hourly else (86400.0, 1)
    W = bench.WARM_DAYS * (24 if hourly else 1)
    out[name] = bench.truth_discharge(so, forcing, dt, forcing.shape[0], W, gap, hourly)[0]
    assert out[name].shape == (bench.N_DAYS,)
np.savez(os.path.join(ROOT, 'tests', 'golden', 'bench_truth.npz'), **out)
print({k: (v.shape, float(v.mean())) for k, v in out.items()})
