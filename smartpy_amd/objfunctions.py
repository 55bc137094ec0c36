"""Objective functions of the calibration pipeline (mirror of smartpy/objfunctions.py + montecarlo.py:193-209).

In the reference only `groundwater_constraint` lives in-tree; NSE / KGE (+ its three components) / PBias / RMSE
are spotpy functions called per sample.  Here they are computed for the whole ensemble on the GPU, either fused
into the time-loop kernel (one-pass moments, engine.run_ensemble(obs=...)) or from a stored discharge matrix
(`objective_functions`, two-pass like spotpy).  Column order: montecarlo.py:71-74.
"""
from .engine import OBJ_FN_NAMES, objective_functions  # noqa: F401


def groundwater_constraint(evaluation, simulation):
    """1.0 when the simulated groundwater contribution to runoff lies within 0.1 of the constraint, both bounds
    included, else 0.0 (objfunctions.py:20-24); both arguments are one-element sequences, as the Monte-Carlo protocol
    passes them (montecarlo.py:186,191,207).  The same rule is fused into the ensemble kernel (finish_objectives in
    csrc/smart_device.h)."""
    target, value = evaluation[0], simulation[0]
    return float(target - 0.1 <= value <= target + 0.1)
