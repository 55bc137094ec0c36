"""Objective functions as the reference calls them -- numpy restatement.  TEST INFRASTRUCTURE ONLY.

The arithmetic lives in a third-party dependency that is absent from /root/reference and from this
image: spotpy (setup.py:56 `spotpy>=1.5.14`, unpinned, un-vendored).  Its published formulas
(spotpy/objectivefunctions.py: nashsutcliffe, kge(return_all=True), pbias, rmse) are restated here in
the two-pass numpy form spotpy uses, at the reference's call sites montecarlo.py:193-209.

Parity status: PINNED to float32 file precision by the reference's committed example output
examples/out/ExampleDaily/ExampleDaily.SMART.lhs (tests/golden/g4_example_lhs.npz): re-running the
oracle with that file's 10 parameter rows reproduces its NSE, KGE, KGEc, KGEa, KGEb, PBias, RMSE and
GW columns (tests/test_oracle_golden.py::test_g4_*).
"""
import numpy as np

NAMES = ['NSE', 'KGE', 'KGEc', 'KGEa', 'KGEb', 'PBias', 'RMSE', 'GW']  # montecarlo.py:71-74


def groundwater_constraint(gw_obs, gw_sim):
    """objfunctions.py:20-24."""
    return 1.0 if (gw_obs - 0.1 <= gw_sim) and (gw_sim <= gw_obs + 0.1) else 0.0


def objective_functions(sim, obs, gw_sim=None, gw_obs=None):
    """montecarlo.py:193-209 for one sample: sim[R], obs[R] (NaN = missing) -> list of 7 or 8 floats."""
    obs = np.asarray(obs, dtype=np.float64)
    mask = ~np.isnan(obs)                                   # montecarlo.py:195-196
    e = obs[mask]
    s = np.asarray(sim, dtype=np.float64)[mask]
    nse = 1 - np.sum((e - s) ** 2) / np.sum((e - np.mean(e)) ** 2)          # :199
    cc = np.corrcoef(e, s)[0, 1]                                            # :200-201
    alpha = np.std(s) / np.std(e)
    beta = np.sum(s) / np.sum(e)
    kge = 1 - np.sqrt((cc - 1) ** 2 + (alpha - 1) ** 2 + (beta - 1) ** 2)
    pbias = 100 * (float(np.sum(s - e)) / float(np.sum(e)))                 # :202
    rmse = np.sqrt(np.mean((e - s) ** 2))                                   # :203
    out = [nse, kge, cc, alpha, beta, pbias, rmse]
    if gw_obs:                                                              # :205-207
        out.append(groundwater_constraint(gw_obs, gw_sim))
    return out


def objective_matrix(sims, obs, gw_sims=None, gw_obs=None):
    """[N,R] discharges -> [N,7|8] float64."""
    return np.array([objective_functions(sims[n], obs, None if gw_sims is None else gw_sims[n], gw_obs)
                     for n in range(len(sims))], dtype=np.float64)
