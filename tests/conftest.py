import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope='session')
def example():
    """Forcing / observations / parameters of the reference's shipped example (tests/golden/forcing_example.npz)."""
    g = load_golden('forcing_example.npz')
    extra = {'aar': float(g['extra'][0]), 'r-o_ratio': float(g['extra'][1]), 'r-o_split': tuple(g['extra'][2:])}
    return {
        'rain_daily': g['rain_daily'], 'peva_daily': g['peva_daily'], 'flow_obs': g['flow_obs'],
        'rain_hourly': np.repeat(g['rain_daily'] / 24, 24), 'peva_hourly': np.repeat(g['peva_daily'] / 24, 24),
        'area': float(g['area']), 'params': g['params'], 'extra': extra,
    }


@pytest.fixture(scope='session', autouse=True)
def _native_libraries():
    """Make sure the in-tree HIP library and the oracle are built before any test needs them (no-ops when they are
    up to date; hipcc cross-compiles without a GPU).  The product itself never builds anything implicitly."""
    from smartpy_amd import build as hip_build
    hip_build.build()
    from oracle import smart_oracle
    smart_oracle.build()
