"""A `smartcpp`-compatible module: what the reference's optional accelerator hook expects to import.

The reference does `import smartcpp` (structure.py:22-27) and, if that works, calls `smartcpp.allsteps(...)`
with the arguments of run_all_steps (structure.py:56-62,118-121,143-146) or, for old versions, `smartcpp.onestep`
per step (structure.py:171-187).  Registering this module under that name makes the unmodified reference run its
time loop on the MI355X:

    import sys, smartpy_amd.smartcpp
    sys.modules['smartcpp'] = smartpy_amd.smartcpp      # before `import smartpy`

(INTEGRATION.md has the variant that binds the C ABI directly with ctypes, without this package.)
"""
from .engine import allsteps, onestep  # noqa: F401

__version__ = '0.2.0'      # >= 0.2.0: `allsteps` is available (structure.py:57)


def install():
    """Register this module as `smartcpp` so that a subsequently imported reference picks it up."""
    import sys
    sys.modules['smartcpp'] = sys.modules[__name__]
