"""smartpy_amd -- MI355X-native engine for the SMART rainfall-runoff model's Monte-Carlo time loop.

A from-scratch implementation of the hot path of ThibHlln/smartpy (structure.run -> run_all_steps ->
run_one_step, driven per sample by SMART.simulate / montecarlo.LHS): the whole parameter ensemble advances
in lockstep inside one hand-written HIP kernel (gfx950), behind the C ABI of include/smart_amd.h, with the
reference's Python surface on top (SMART, Parameters, objfunctions, montecarlo.*, and a `smartcpp`-compatible
module).  There is no CPU fallback: the engine fails loudly without the HIP extension or without a GPU.
"""
from .version import __version__
from .parameters import Parameters

__all__ = ['__version__', 'Parameters', 'SMART', 'objfunctions', 'engine']


def __getattr__(name):
    # torch and the HIP library are only needed by the compute modules: import them on first use
    if name in ('SMART',):
        from .smart import SMART
        return SMART
    if name in ('objfunctions', 'engine', 'structure', 'smartcpp', 'montecarlo', 'distributed', 'sampling'):
        import importlib
        return importlib.import_module('.' + name, __name__)
    raise AttributeError(name)
