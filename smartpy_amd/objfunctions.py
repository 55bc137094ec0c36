"""Objective functions of the calibration pipeline (mirror of smartpy/objfunctions.py + montecarlo.py:193-209).

In the reference only `groundwater_constraint` lives in-tree; NSE / KGE (+ its three components) / PBias / RMSE
are spotpy functions called per sample.  Here they are computed for the whole ensemble on the GPU, either fused
into the time-loop kernel (one-pass moments, engine.run_ensemble(obs=...)) or from a stored discharge matrix
(`objective_functions`, two-pass like spotpy).  Column order: montecarlo.py:71-74.
"""
from .engine import OBJ_FN_NAMES, objective_functions  # noqa: F401


def groundwater_constraint(evaluation, simulation):
    """objfunctions.py:20-24: 1.0 if the simulated groundwater ratio is within +/- 0.1 of the constraint."""
    if (evaluation[0] - 0.1 <= simulation[0]) and (simulation[0] <= evaluation[0] + 0.1):
        return 1.0
    else:
        return 0.0
