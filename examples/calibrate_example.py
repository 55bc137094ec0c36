#!/usr/bin/env python3
"""The workflow of the reference's examples/api_usage_example.ipynb on the GPU engine: one simulation with the
shipped parameter set, then a Latin-hypercube calibration and the two second stages (GLUE, Best).

    python examples/calibrate_example.py [sample_size] [root]

`root` must contain in/Catchment/Catchment.{rain,peva,flow,sttngs,parameters} (default: a scratch copy of
tests/golden/data, the example catchment of the reference).  Start it with
`python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 examples/calibrate_example.py`
to shard the sample over the GPUs of a node.
"""
import os
import shutil
import sys
import tempfile
import time
from datetime import datetime, timedelta

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

import smartpy_amd                                   # noqa: E402
from smartpy_amd import distributed                  # noqa: E402
from smartpy_amd.montecarlo import LHS, GLUE, Best   # noqa: E402

EXTRA = {'aar': 1200, 'r-o_ratio': 0.45, 'r-o_split': (0.10, 0.15, 0.15, 0.30, 0.30)}


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    rank, world, device = distributed.init()
    if len(sys.argv) > 2:
        root = sys.argv[2]
    else:
        root = os.path.join(tempfile.gettempdir(), 'smartpy_amd_example')
        if rank == 0 and not os.path.isdir(os.path.join(root, 'in')):
            shutil.copytree(os.path.join(os.path.dirname(HERE), 'tests', 'golden', 'data', 'in'),
                            os.path.join(root, 'in'))
        distributed.barrier()

    # ---- one simulation, like smartpy.SMART(...).simulate(...) ------------------------------------------------
    sm = smartpy_amd.SMART('Catchment', 175.46e6, datetime(2007, 1, 1, 9), datetime(2016, 12, 31, 9),
                           timedelta(hours=1), timedelta(days=1), 365, 'csv', 'csv', root, gauged_area_m2=175.97e6)
    sm.extra = EXTRA
    sm.parameters.set_parameters_with_file(sm.in_f + 'Catchment.parameters')
    discharge, gw = sm.simulate(sm.parameters.values)
    obs = sm.get_evaluation_array()
    ok = ~np.isnan(obs)
    nse = 1 - np.sum((obs[ok] - discharge[ok]) ** 2) / np.sum((obs[ok] - obs[ok].mean()) ** 2)
    if rank == 0:
        print('single run: %d daily discharges, NSE = %.6f, groundwater contribution = %.4f' % (len(discharge), nse, gw))

    # ---- calibration: LHS over the default ranges, every row in one launch per GPU ---------------------------------
    np.random.seed(2718)                    # reproducible; with several ranks run() takes rank 0's sample either way
    t0 = time.perf_counter()
    lhs = LHS('Catchment', root, 'csv', 'csv', n)
    lhs.model.extra = EXTRA
    t1 = time.perf_counter()
    lhs.run()
    t2 = time.perf_counter()
    if rank == 0:
        best = int(np.nanargmax(lhs.obj_fns[:, 0]))
        print('LHS: %d samples x %d hourly steps on %d GPU(s): set-up %.2f s, run + database %.2f s -> %s'
              % (n, len(sm.nd_rain) + 8760, world, t1 - t0, t2 - t1, lhs.db_file))
        print('     best NSE %.4f (KGE %.4f) at %s' % (lhs.obj_fns[best, 0], lhs.obj_fns[best, 1],
                                                     {k: round(float(v), 4) for k, v in zip(lhs.param_names, lhs.lhs_params[best])}))

    # ---- second stages: from the finished run itself (selection on the GPU, only the chosen rows travel) or, like the
    # ---- reference's GLUE / Best, from the database file (leave `sampling=` out) ---------------------------------------
    glue = GLUE('Catchment', root, 'csv', 'csv', conditioning={'NSE': ('min', (0.4,)), 'KGEc': ('min', (0.9,))},
                sampling=lhs)
    glue.model.extra = EXTRA
    glue.run()
    top = Best('Catchment', root, 'csv', 'csv', target='KGE', nb_best=10, constraining={'GW': ('equal', (1.0,))})
    top.model.extra = EXTRA
    top.run()
    if rank == 0:
        print('GLUE: %d behavioural sets -> %s' % (len(glue.behavioural_params), glue.db_file))
        print('Best: 10 best KGE among the sets meeting the groundwater constraint -> %s' % top.db_file)
    distributed.finish()        # (several ranks: leave the process group -- without waiting for a communicator that never answered)


if __name__ == '__main__':
    main()
