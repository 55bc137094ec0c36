"""The N > 1 path on CPU: two processes, gloo backend.  Covers the sharding + single all-gather used by
bench.py and MonteCarlo.run() (on the GPU box the same code runs over RCCL with one rank per GPU)."""
import os
import shutil
import socket
import sys
from datetime import timedelta

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, GOLDEN


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _init(rank, world, port):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    dist.init_process_group('gloo', rank=rank, world_size=world)


def _worker_gather(rank, world, port, out_dir):
    _init(rank, world, port)
    from smartpy_amd import distributed as sdist
    assert sdist.is_distributed() and sdist.rank_world() == (rank, world)
    ok = True
    for n in (7, 8, 1, 2, 101):
        full = torch.arange(n * 9, dtype=torch.float64).reshape(n, 9) * 1.5
        lo, hi = sdist.shard_bounds(n, world, rank)
        got = sdist.gather_rows(full[lo:hi].clone(), n)
        ok = ok and torch.equal(got, full)
        got1 = sdist.gather_rows(full[lo:hi, 0].clone(), n)                  # 1-D rows (gw)
        ok = ok and torch.equal(got1, full[:, 0])
    ok = ok and sdist.max_over_ranks(10.0 + rank, torch.device('cpu')) == 10.0 + world - 1
    ok = ok and sdist.sum_over_ranks(1.0 + rank, torch.device('cpu')) == sum(1.0 + r for r in range(world))
    sdist.barrier()
    open(os.path.join(out_dir, 'gather_%d.txt' % rank), 'w').write('ok' if ok else 'FAILED')
    dist.destroy_process_group()


def test_gather_rows_two_ranks(tmp_path):
    port = _free_port()
    mp.spawn(_worker_gather, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert [open(tmp_path / ('gather_%d.txt' % r)).read() for r in range(2)] == ['ok', 'ok']


# ---- MonteCarlo.run() sharded over two ranks, with the GPU launch replaced by the CPU oracle ------------------
class _OracleResult(object):
    def __init__(self, discharge, gw, objfn):
        self.discharge, self.gw, self.objfn = discharge, gw, objfn


def _oracle_simulate_ensemble(self, parameters, report='summary', objective_functions=False, gw_constraint=None,
                              save_discharge=True, math_mode='fast', device=None):
    """Stand-in for SMART.simulate_ensemble on a machine without a GPU (tests only)."""
    from oracle import smart_oracle as so, objfn_oracle
    from smartpy_amd import structure
    dt = self.delta_simu.total_seconds()
    T = len(self.timeseries) - 1
    W = structure.warm_up_length(self.warm_up, dt, T) if self.warm_up else 0
    gap = T // (len(self.timeseries_report) - 1)
    dis, gw, _ = so.run_batch(self.area, dt, T, W, self.nd_rain, self.nd_peva, np.asarray(parameters, float),
                              self.extra, so.REPORT_SUMMARY, gap)
    obj = objfn_oracle.objective_matrix(dis, self.nd_flow, gw, gw_constraint)
    if obj.shape[1] == 7:
        obj = np.concatenate([obj, np.full((len(obj), 1), np.nan)], axis=1)
    return _OracleResult(torch.from_numpy(dis), torch.from_numpy(gw), torch.from_numpy(obj))


def _make_root(tmp, days=120):
    root = os.path.join(tmp, 'data')
    shutil.copytree(os.path.join(GOLDEN, 'data', 'in'), os.path.join(root, 'in'))
    with open(os.path.join(root, 'in', 'Catchment', 'Catchment.sttngs'), 'w') as f:
        f.write('ARGUMENT,VALUE\ncatchment_area_km2,175.46\ngauged_area_km2,175.97\n'
                'start_datetime,01/01/2007 09:00:00\nend_datetime,%s 09:00:00\nsimu_timedelta_min,60\n'
                'report_timedelta_min,1440\nwarm_up_days,30\ngw_constraint,0.12667\n'
                % ('30/04/2007' if days == 120 else '31/12/2007'))
    return root


def _worker_lhs(rank, world, port, root, save_sim):
    if world > 1:
        _init(rank, world, port)
    elif ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    from smartpy_amd.smart import SMART
    from smartpy_amd.montecarlo import LHS
    SMART.simulate_ensemble = _oracle_simulate_ensemble
    np.random.seed(2718)                    # every rank draws the same sample, like every MPI rank of the reference
    lhs = LHS('Catchment', root, 'csv', 'csv', 13, save_sim=save_sim)
    lhs.model.extra = {'aar': 1200, 'r-o_ratio': 0.45, 'r-o_split': (0.10, 0.15, 0.15, 0.30, 0.30)}
    lhs.run()
    np.save(os.path.join(root, 'objfns_w%d_r%d.npy' % (world, rank)), lhs.obj_fns)
    if world > 1:
        dist.destroy_process_group()


@pytest.mark.parametrize('save_sim', [False, True])
def test_lhs_run_sharded_equals_single_process(tmp_path, save_sim):
    root1 = _make_root(str(tmp_path / 'one'))
    root2 = _make_root(str(tmp_path / 'two'))
    _worker_lhs(0, 1, 0, root1, save_sim)
    mp.spawn(_worker_lhs, args=(2, _free_port(), root2, save_sim), nprocs=2, join=True)
    db1 = open(os.path.join(root1, 'out', 'Catchment', 'Catchment.SMART.lhs')).read()
    db2 = open(os.path.join(root2, 'out', 'Catchment', 'Catchment.SMART.lhs')).read()
    assert db1 == db2                                               # rank 0 wrote the same database, row for row
    lines = db1.strip().split('\n')
    head = lines[0].split(',')
    assert head[:18] == ['NSE', 'KGE', 'KGEc', 'KGEa', 'KGEb', 'PBias', 'RMSE', 'GW',
                         'T', 'C', 'H', 'D', 'S', 'Z', 'SK', 'FK', 'GK', 'RK']
    assert len(lines) == 14 and len(head) == 18 + (120 if save_sim else 0)
    if save_sim:
        assert head[18] == '2007-01-01 09:00:00' and head[-1] == '2007-04-30 09:00:00'
    a = np.load(os.path.join(root1, 'objfns_w1_r0.npy'))
    for r in range(2):                                              # every rank holds the full gathered matrix
        assert np.array_equal(a, np.load(os.path.join(root2, 'objfns_w2_r%d.npy' % r)))
    # the observed series was written next to it (montecarlo.py:88)
    assert os.path.exists(os.path.join(root2, 'out', 'Catchment', 'Catchment.obs.flow'))


def test_second_stage_with_no_behavioural_set_writes_header_only(tmp_path):
    """GLUE on a sampling database with a condition nothing meets: zero simulations, header-only output, like the
    reference's sampler looping zero times (glue.py:222-289 can return an empty array)."""
    root = _make_root(str(tmp_path / 'one'))
    _worker_lhs(0, 1, 0, root, False)
    from smartpy_amd.montecarlo import GLUE, Best
    glue = GLUE('Catchment', root, 'csv', 'csv', conditioning={'NSE': ('min', (2.0,))})
    assert glue.behavioural_params.shape == (0, 10)
    glue.run()
    lines = open(os.path.join(root, 'out', 'Catchment', 'Catchment.SMART.glue')).read().split('\n')
    assert lines[0].startswith('NSE,KGE') and lines[1:] == ['']
    assert glue.obj_fns.shape == (0, 8)
    with pytest.raises(Exception, match='higher than the sample size'):
        Best('Catchment', root, 'csv', 'csv', target='NSE', nb_best=14)
    with pytest.raises(Exception, match='not recognised'):
        Best('Catchment', root, 'csv', 'csv', target='nse', nb_best=2)
    best = Best('Catchment', root, 'csv', 'csv', target='KGE', nb_best=3)
    assert best.best_params.shape == (3, 10) and best.db_file.endswith('Catchment.SMART.3best')
