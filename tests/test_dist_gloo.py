"""The N > 1 path on CPU: two and EIGHT processes, gloo backend.  Covers the sharding + single all-gather used by
bench.py and MonteCarlo.run() (on the GPU box the same code runs over RCCL with one rank per GPU), the rank-0-only
collection of the saved series, empty shards, and a rank that fails."""
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


def _failing_simulate_ensemble(self, parameters, **kw):
    """... and one whose launch on rank 3 reports a status its repeated launch could not clear."""
    if dist.is_initialized() and dist.get_rank() == 3:
        from smartpy_amd.engine import SmartEngineError
        raise SmartEngineError(-6, 'the repeated launch reports status 0x1 as well')
    return _oracle_simulate_ensemble(self, parameters, **kw)


def _oom_simulate_ensemble(self, parameters, **kw):
    """... and one that runs out of device memory on rank 3 (what torch raises is a RuntimeError, not the engine's own)."""
    if dist.is_initialized() and dist.get_rank() == 3:
        raise RuntimeError('HIP out of memory. Tried to allocate 27.22 GiB')
    return _oracle_simulate_ensemble(self, parameters, **kw)


def _worker_lhs(rank, world, port, root, save_sim, seeded=True, n=13, failing=False):
    if world > 1:
        _init(rank, world, port)
    elif ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    from smartpy_amd.smart import SMART
    from smartpy_amd.montecarlo import LHS
    from smartpy_amd import distributed as sdist
    from smartpy_amd.engine import SmartEngineError
    SMART.simulate_ensemble = {False: _oracle_simulate_ensemble, True: _failing_simulate_ensemble,
                               'oom': _oom_simulate_ensemble}[failing]
    # seeded: every rank draws the same sample.  Unseeded (what the reference's own scripts do): every process draws
    # another one from NumPy's global stream, and rank 0's has to win
    np.random.seed(2718 if seeded else 1000 + rank)
    lhs = LHS('Catchment', root, 'csv', 'csv', n, save_sim=save_sim)
    lhs.model.extra = {'aar': 1200, 'r-o_ratio': 0.45, 'r-o_split': (0.10, 0.15, 0.15, 0.30, 0.30)}
    if failing:
        # every rank raises, in the same call: the failed one its own error, the others one that says a peer failed --
        # and all of them are still in step afterwards (the barrier below would hang otherwise)
        try:
            lhs.run()
            outcome = 'returned'
        except SmartEngineError as e:
            outcome = 'peer' if 'another rank' in str(e) else 'own'
        except RuntimeError as e:
            outcome = 'own: ' + str(e)[:18]
        dist.barrier()
        open(os.path.join(root, 'outcome_r%d.txt' % rank), 'w').write(outcome)
        dist.destroy_process_group()
        return
    lhs.run()
    np.save(os.path.join(root, 'objfns_w%d_r%d.npy' % (world, rank)), lhs.obj_fns)
    np.save(os.path.join(root, 'sample_w%d_r%d.npy' % (world, rank)), lhs._sample)
    lo, hi = sdist.shard_bounds(n, world, rank)
    open(os.path.join(root, 'bytes_w%d_r%d.txt' % (world, rank)), 'w').write(
        '%d %d' % (sdist.last_collect_bytes, hi - lo))
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


def test_unseeded_ranks_run_rank_zeros_sample(tmp_path):
    """Two processes that drew DIFFERENT samples (unseeded NumPy streams): run() broadcasts rank 0's matrix before
    sharding, so the database pairs every parameter row with the results of that very row -- the same file a single
    process with rank 0's stream writes."""
    root1 = _make_root(str(tmp_path / 'one'))
    root2 = _make_root(str(tmp_path / 'two'))
    _worker_lhs(0, 1, 0, root1, False, False)                       # one process, seed 1000 = rank 0's stream
    mp.spawn(_worker_lhs, args=(2, _free_port(), root2, False, False), nprocs=2, join=True)
    assert open(os.path.join(root1, 'out', 'Catchment', 'Catchment.SMART.lhs')).read() == \
        open(os.path.join(root2, 'out', 'Catchment', 'Catchment.SMART.lhs')).read()
    s0, s1 = (np.load(os.path.join(root2, 'sample_w2_r%d.npy' % r)) for r in range(2))
    assert np.array_equal(s0, s1) and np.array_equal(s0, np.load(os.path.join(root1, 'sample_w1_r0.npy')))


# ---- ShardedEnsemble: sample blocks and catchment blocks, with the launch replaced by the CPU oracle ---------------
class _OraclePrepared(object):
    """Stand-in for engine.prepare_ensemble on a machine without a GPU: same arguments, launch() through the oracle."""

    def __init__(self, params, forcing, area_m2, delta_sec, n_warm, report_gap, obs=None, gw_obs=None, extra=None,
                 device=None, **kw):
        self.a = (np.asarray(params, float), np.asarray(forcing, float), area_m2, delta_sec, n_warm, report_gap,
                  obs, gw_obs, extra)

    repeated = False

    def enqueue(self):
        self._last = self.launch()

    def result(self):
        return self._last

    def launch(self):
        from oracle import smart_oracle as so, objfn_oracle
        params, forcing, area, dt, W, gap, obs, gw_obs, extra = self.a
        squeeze = forcing.ndim == 2
        forcing = forcing[None] if squeeze else forcing
        C = forcing.shape[0]
        objfn, gws = [], []
        for c in range(C):
            p = params if params.ndim == 2 else params[c]
            a_c = area if np.ndim(area) == 0 else area[c]
            dis, gw, _ = so.run_batch(float(a_c), dt, forcing.shape[1], W, forcing[c, :, 0].copy(),
                                      forcing[c, :, 1].copy(), p, extra, so.REPORT_SUMMARY, gap)
            o = np.full((len(p), 8), np.nan)
            if obs is not None:
                g = gw_obs if gw_obs is None or np.ndim(gw_obs) == 0 else gw_obs[c]
                m = objfn_oracle.objective_matrix(dis, obs if np.ndim(obs) == 1 else obs[c], gw, g)
                o[:, :m.shape[1]] = m
            objfn.append(o)
            gws.append(gw)
        objfn, gws = torch.from_numpy(np.stack(objfn)), torch.from_numpy(np.stack(gws))
        return _OracleResult(None, gws[0] if squeeze else gws, objfn[0] if squeeze else objfn)

    def verify(self):
        return self._last

    def describe(self):
        return 'oracle stand-in[%d rows]' % len(self.a[0])


class _FlakyPrepared(_OraclePrepared):
    """... whose first launch on rank 1 comes back poisoned (a time slice that timed out writes NaN), and whose
    verify() repeats it -- what engine.PreparedEnsemble.verify() does on a non-zero status word."""

    def enqueue(self):
        _OraclePrepared.enqueue(self)
        self._poisoned = dist.is_initialized() and dist.get_rank() == 1 and not getattr(self, '_done', False)
        if self._poisoned:
            self._last.gw[...] = float('nan')
            self._last.objfn[...] = float('nan')

    def verify(self):
        self.repeated = self._poisoned
        if self._poisoned:
            self._done = True
            self._poisoned = False          # the status word read next is that of the repeated, clean launch
            _OraclePrepared.enqueue(self)
        return self._last


class _BrokenPrepared(_OraclePrepared):
    """... whose verify() gives up on rank 1 (the repeated launch reports a status as well)."""

    def verify(self):
        if dist.is_initialized() and dist.get_rank() == 1:
            from smartpy_amd.engine import SmartEngineError
            raise SmartEngineError(-6, 'the repeated launch reports status 0x1 as well')
        return self._last


def _sharded_setup():
    rng = np.random.default_rng(5)
    C, N, days = 5, 6, 40
    forcing = np.stack([np.stack([np.repeat(rng.gamma(0.7, 4.5, days) / 24, 24),
                                  np.repeat(rng.uniform(0.2, 2.5, days) / 24, 24)], axis=1) for _ in range(C)])
    from oracle import lhs_oracle
    params = lhs_oracle.lhs_params(N, seed=3)
    areas = rng.uniform(50e6, 500e6, C)
    obs = rng.uniform(0.5, 4.0, (C, days))
    obs[rng.random((C, days)) < 0.1] = np.nan
    gw_obs = rng.uniform(0.05, 0.3, C)
    extra = {'aar': 1200, 'r-o_ratio': 0.45, 'r-o_split': (0.10, 0.15, 0.15, 0.30, 0.30)}
    return C, N, forcing, params, areas, obs, gw_obs, extra


def _worker_sharded(rank, world, port, out_dir):
    if world > 1:
        _init(rank, world, port)
    elif ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    from smartpy_amd import distributed as sdist, engine
    engine.prepare_ensemble = _OraclePrepared
    C, N, forcing, params, areas, obs, gw_obs, extra = _sharded_setup()
    asked = []

    def forcing_of(c):          # a rank must only ever be asked for its own catchments
        asked.append(c)
        return forcing[c]
    cat = sdist.ShardedEnsemble(params, forcing_of, areas, 3600.0, 240, 24, axis='catchments', n_catchments=C,
                                obs=obs, gw_obs=gw_obs, extra=extra, device='cpu')
    lo, hi = sdist.shard_bounds(C, world, rank)
    assert sorted(asked) == list(range(lo, hi)) and cat.n_local == hi - lo
    got_c = cat.step()
    sam = sdist.ShardedEnsemble(params, forcing[1], areas[1], 3600.0, 240, 24, axis='samples', obs=obs[1],
                                gw_obs=float(gw_obs[1]), extra=extra, device='cpu')
    got_s = sam.step()
    own = params[rank::world][:3]       # weak mode: every rank brings its own block of 3 rows
    weak = sdist.ShardedEnsemble(own, forcing[2], areas[2], 3600.0, 240, 24, axis='samples', local_block=True,
                                 obs=obs[2], extra=extra, device='cpu')
    got_w = weak.step()
    # a launch that one rank had to repeat: step() gathered that rank's NaN rows, verify() gathers again -- on every
    # rank, the ones whose own launch was clean included
    engine.prepare_ensemble = _FlakyPrepared
    flaky = sdist.ShardedEnsemble(params, forcing[1], areas[1], 3600.0, 240, 24, axis='samples', obs=obs[1],
                                  gw_obs=float(gw_obs[1]), extra=extra, device='cpu')
    first = flaky.step()
    again = flaky.verify()
    if world > 1:
        assert torch.isnan(first[:, 8]).any() and again is not None and not torch.isnan(again[:, 8]).any()
        assert torch.equal(again, got_s)
        assert flaky.verify() is None                   # nothing to repeat the second time
    else:
        assert again is None and torch.equal(first, got_s)
    # a rank whose repeated launch fails too: EVERY rank raises, in the same verify() call -- none is left waiting in
    # a reduction or a gather its failed peer never joins (advisor, round 3)
    if world > 1:
        engine.prepare_ensemble = _BrokenPrepared
        broken = sdist.ShardedEnsemble(params, forcing[1], areas[1], 3600.0, 240, 24, axis='samples', obs=obs[1],
                                       gw_obs=float(gw_obs[1]), extra=extra, device='cpu')
        broken.step()
        try:
            broken.verify()
            raise AssertionError('rank %d: verify() returned although rank 1 failed' % rank)
        except engine.SmartEngineError as e:
            assert ('another rank' in str(e)) == (rank == 0)
        dist.barrier()                                  # both ranks are still in step with each other
    np.savez(os.path.join(out_dir, 'sharded_w%d_r%d.npz' % (world, rank)), c=got_c.numpy(), s=got_s.numpy(),
             w=got_w.numpy())
    if world > 1:
        dist.destroy_process_group()


def test_sharded_ensemble_over_catchments_and_samples(tmp_path):
    """BASELINE config 5's split: 5 catchments over 2 ranks (3 + 2), each rank holding only its own forcing; and the
    sample-axis split of configs 3 / 4.  Every rank ends up with the single-process result, bit for bit."""
    _worker_sharded(0, 1, 0, str(tmp_path))
    mp.spawn(_worker_sharded, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    one = np.load(tmp_path / 'sharded_w1_r0.npz')
    C, N = 5, 6
    assert one['c'].shape == (C, N, 9) and one['s'].shape == (N, 9)
    assert np.all(np.isfinite(one['c'])) and np.array_equal(one['c'][1], one['s'])
    for r in range(2):
        two = np.load(tmp_path / ('sharded_w2_r%d.npz' % r))
        assert np.array_equal(two['c'], one['c']) and np.array_equal(two['s'], one['s'])
        # weak mode: the gathered matrix is the ranks' own blocks in rank order
        assert two['w'].shape == (6, 9)
    _, _, forcing, params, areas, obs, gw_obs, extra = _sharded_setup()
    two = np.load(tmp_path / 'sharded_w2_r0.npz')
    whole = _OraclePrepared(np.concatenate([params[0::2][:3], params[1::2][:3]]), forcing[2], areas[2], 3600.0, 240, 24,
                            obs=obs[2], extra=extra).launch()
    assert np.array_equal(two['w'][:, :8], whole.objfn.numpy(), equal_nan=True)
    assert np.array_equal(two['w'][:, 8], whole.gw.numpy())


# ---- eight ranks: what the driver's 8-GPU run starts (one process per GPU), on the CPU ---------------------------------
def _worker_gather8(rank, world, port, out_dir):
    _init(rank, world, port)
    from smartpy_amd import distributed as sdist
    ok = True
    for n in (5, 8, 9, 64, 101):                    # 5 rows over 8 ranks: three empty shards
        full = torch.arange(n * 9, dtype=torch.float64).reshape(n, 9) * 0.5
        lo, hi = sdist.shard_bounds(n, world, rank)
        ok = ok and torch.equal(sdist.gather_rows(full[lo:hi].clone(), n), full)
        series = (torch.arange(n * 7, dtype=torch.float32).reshape(n, 7) + 0.25)
        got = sdist.collect_rows(series[lo:hi].clone(), n, dst=0)
        if rank == 0:
            ok = ok and got.dtype == np.float32 and np.array_equal(got, series.numpy())
            ok = ok and sdist.last_collect_bytes == (n - (hi - lo)) * 7 * 4      # everybody else's rows came in
        else:
            ok = ok and got is None and sdist.last_collect_bytes == (hi - lo) * 7 * 4   # its own rows went out, no more
    ok = ok and sdist.max_over_ranks(rank) == world - 1 and sdist.data_backend() == 'gloo'
    sdist.barrier()
    open(os.path.join(out_dir, 'gather8_%d.txt' % rank), 'w').write('ok' if ok else 'FAILED')
    dist.destroy_process_group()


def test_gather_and_collect_rows_eight_ranks(tmp_path):
    mp.spawn(_worker_gather8, args=(8, _free_port(), str(tmp_path)), nprocs=8, join=True)
    assert [open(tmp_path / ('gather8_%d.txt' % r)).read() for r in range(8)] == ['ok'] * 8


def _worker_probe(rank, world, port, out_dir, failing_rank, hangs=False):
    _init(rank, world, port)
    import warnings
    from smartpy_amd import distributed as sdist

    import time

    def probe(device, group):   # stands for the group's first device collective
        if rank == failing_rank and hangs:
            time.sleep(600)     # a communicator set-up that never returns (the commoner RCCL failure)
        if rank == failing_rank:
            raise RuntimeError("NCCL error: unhandled cuda error\nhipIpcGetMemHandle: invalid argument")
        return float(world)
    t0 = time.monotonic()
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        works = sdist.rccl_answers(torch.device('cpu'), probe=probe)
    took = time.monotonic() - t0
    ok = works == (failing_rank is None)
    if failing_rank is None:
        ok = ok and sdist.rccl_failure is None and not w and not sdist._ABANDONED
    else:
        mine = ('the first device collective over 4 ranks did not answer within 3 s' if hangs
                else 'RuntimeError: NCCL error: unhandled cuda error')
        ok = ok and sdist.rccl_failure == (mine if rank == failing_rank else 'RCCL failed on another rank')
        ok = ok and (len(w) == 1) == (rank == 0) and sdist._ABANDONED
        ok = ok and took < 20.0          # bounded: the probe's 3 s + the agreement, not the ten minutes of a watchdog
    open(os.path.join(out_dir, 'probe_%d.txt' % rank), 'w').write('ok' if ok else 'FAILED %r %r %.1f' % (works, sdist.rccl_failure, took))
    # (a rank whose stand-in still sleeps ends the way finish() ends a process that left a communicator behind)
    sdist.finish(0)


@pytest.mark.parametrize('failing_rank, hangs', [(None, False), (0, False), (2, False), (1, True)])
def test_ranks_agree_on_staging_when_rccl_does_not_start_on_one_of_them(tmp_path, monkeypatch, failing_rank, hangs):
    """distributed.init()'s first device collective, replaced by a stand-in that raises on one rank the way a refused
    IPC handle does -- or that never RETURNS (round 5 caught the raise only; a hang waited for RCCL's watchdog, ten
    minutes of silence and no line): every rank gives the same answer (stage through the host) within the probe's
    bound, rank 0 alone warns, each keeps a reason, and the process ends without waiting for the thread that hangs."""
    monkeypatch.setenv('SMART_DIST_PROBE_TIMEOUT', '3')
    mp.spawn(_worker_probe, args=(4, _free_port(), str(tmp_path), failing_rank, hangs), nprocs=4, join=True)
    assert [open(tmp_path / ('probe_%d.txt' % r)).read() for r in range(4)] == ['ok'] * 4


def _worker_absent(rank, world, port, out_dir):
    """distributed.init() itself (no GPU here: the gloo default group with its timeout), then one rank never joins."""
    import time
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    from smartpy_amd import distributed as sdist
    sdist.init()
    assert sdist.data_backend() == 'gloo' and sdist.max_over_ranks(rank) == world - 1
    t0 = time.monotonic()
    outcome = 'returned'
    if rank == 2:
        time.sleep(12.0)        # (asleep through the collective: a rank that died, or one stuck in a driver call)
        outcome = 'slept'
    else:
        try:
            sdist.agree_or_raise(None)
        except Exception as e:      # noqa: BLE001 -- gloo's timeout surfaces as a RuntimeError / DistBackendError
            outcome = 'raised after %.0f s' % (time.monotonic() - t0) if time.monotonic() - t0 < 11.0 else 'raised late'
    open(os.path.join(out_dir, 'absent_%d.txt' % rank), 'w').write(outcome)
    os._exit(0)     # (the group is broken by now: no orderly shutdown to be had)


def test_a_rank_that_never_arrives_makes_its_peers_raise_within_the_bound(tmp_path, monkeypatch):
    """init() gives the default group a timeout (SMART_DIST_TIMEOUT, 120 s by default; round 5 passed none, so a peer
    that never arrived meant ten minutes of silence): the ranks that wait raise within it."""
    monkeypatch.setenv('SMART_DIST_TIMEOUT', '5')
    mp.spawn(_worker_absent, args=(4, _free_port(), str(tmp_path)), nprocs=4, join=True)
    got = [open(tmp_path / ('absent_%d.txt' % r)).read() for r in range(4)]
    assert got[2] == 'slept' and all(g.startswith('raised after') for i, g in enumerate(got) if i != 2), got


def test_importing_the_package_leaves_the_hsa_ipc_mode_alone(tmp_path):
    """Round 5 set HSA_ENABLE_IPC_MODE_LEGACY=0 on import of smartpy_amd.distributed (advisor: every user process got
    its environment changed).  Now: untouched on import; init() sets it only when SMART_DIST_IPC_DMABUF=1 asks."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ('HSA_ENABLE_IPC_MODE_LEGACY', 'SMART_DIST_IPC_DMABUF')}
    code = ("import os, sys; sys.path.insert(0, %r); import smartpy_amd.montecarlo, smartpy_amd.distributed as d; "
            "a = os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY'); d.init(); b = os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY'); "
            "os.environ['SMART_DIST_IPC_DMABUF'] = '1'; d.init(); print(a, b, os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY'))" % ROOT)
    out = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    assert out.stdout.split() == ['None', 'None', '0']


@pytest.mark.parametrize('n', [13, 5])
def test_lhs_run_over_eight_ranks_with_saved_series(tmp_path, n):
    """MonteCarlo.run(save_sim=True) over eight ranks (n = 5: three of them without a row): the database rank 0 writes
    is the single process's, byte for byte; every rank holds the [N, 8] objective functions; and the simulated series
    travelled to rank 0 ONLY -- a rank other than 0 moved its own rows as float32, nothing else (round 4 all-gathered
    the fp64 matrix to every rank: 29 GB x 8 at config 4)."""
    root1 = _make_root(str(tmp_path / 'one'))
    root8 = _make_root(str(tmp_path / 'eight'))
    _worker_lhs(0, 1, 0, root1, True, True, n)
    mp.spawn(_worker_lhs, args=(8, _free_port(), root8, True, True, n), nprocs=8, join=True)
    db1 = open(os.path.join(root1, 'out', 'Catchment', 'Catchment.SMART.lhs')).read()
    db8 = open(os.path.join(root8, 'out', 'Catchment', 'Catchment.SMART.lhs')).read()
    assert db1 == db8 and len(db1.strip().split('\n')) == n + 1
    a = np.load(os.path.join(root1, 'objfns_w1_r0.npy'))
    R = 120
    for r in range(8):
        assert np.array_equal(a, np.load(os.path.join(root8, 'objfns_w8_r%d.npy' % r)))
        moved, n_local = (int(v) for v in open(os.path.join(root8, 'bytes_w8_r%d.txt' % r)).read().split())
        if r == 0:
            assert moved == (n - n_local) * R * 4
        else:
            assert moved == n_local * R * 4 <= n_local * (9 + R) * 8


def test_a_failing_rank_makes_all_eight_raise(tmp_path):
    """One rank's launch fails for good (rank 3): MonteCarlo.run() raises on EVERY rank in the same call -- the failed
    one its own error, the seven others one that names a peer -- instead of leaving them in the gather until the
    process group times out."""
    root = _make_root(str(tmp_path / 'eight'))
    mp.spawn(_worker_lhs, args=(8, _free_port(), root, True, True, 13, True), nprocs=8, join=True)
    got = [open(os.path.join(root, 'outcome_r%d.txt' % r)).read() for r in range(8)]
    assert got == ['peer'] * 3 + ['own'] + ['peer'] * 4


def test_a_rank_out_of_memory_makes_all_eight_raise(tmp_path):
    """... and so does an exception that is not the engine's own (round 5 caught SmartEngineError only: a HIP
    out-of-memory on one rank left the seven others in the gather): the rank re-raises its own RuntimeError."""
    root = _make_root(str(tmp_path / 'eight'))
    mp.spawn(_worker_lhs, args=(8, _free_port(), root, True, True, 13, 'oom'), nprocs=8, join=True)
    got = [open(os.path.join(root, 'outcome_r%d.txt' % r)).read() for r in range(8)]
    assert got == ['peer'] * 3 + ['own: HIP out of memory.'] + ['peer'] * 4


# ---- bench.py: every rank checks rows of the job it timed (parity.timed_launch) -----------------------------------------
def _worker_timed_parity(rank, world, port, out_dir):
    _init(rank, world, port)
    import bench
    from oracle import smart_oracle as so
    from smartpy_amd import distributed as sdist, engine
    engine.prepare_ensemble = _OraclePrepared
    C, _, forcing, _, areas, obs, _, _ = _sharded_setup()
    from oracle import lhs_oracle
    params = lhs_oracle.lhs_params(150, seed=8)
    W, gap, dt = 240, 24, 3600.0
    ok = True
    # sample axis: 75 rows per rank, 48 of them checked on each
    job = sdist.ShardedEnsemble(params, forcing[1], float(areas[1]), dt, W, gap, axis='samples', obs=obs[1],
                                gw_obs=bench.GW_OBS, extra=bench.EXTRA, device='cpu')
    job.step()
    lo, hi = sdist.shard_bounds(150, world, rank)
    args = (job, 'samples', params[lo:hi], lambda c: forcing[1], lambda c: float(areas[1]), obs[1], W, gap, dt)
    t = bench.timed_launch_parity(so, None, *args)['timed_launch']
    ok = ok and t['ok'] and t['ranks'] == world and t['rows'] == 48 * world and t['max_rel_discharge'] is None
    ok = ok and t['max_rel_objfn'] <= 1e-12 and t['max_abs_gw_ratio'] == 0.0 and t['kernel'].startswith('oracle stand-in')
    # one rank's launch is off by 1e-6 in one groundwater ratio: every rank's line says so, and says not ok
    if rank == 1:
        job.prepared.result().gw[:] += 1e-6
    t = bench.timed_launch_parity(so, None, *args)['timed_launch']
    ok = ok and not t['ok'] and abs(t['max_abs_gw_ratio'] - 1e-6) < 1e-9 and t['ranks'] == world
    # a rank without a checker: nobody claims a check that was not made
    t = bench.timed_launch_parity(so if rank == 0 else None, 'no gcc on this rank', *args)
    ok = ok and t['timed_launch'] is None and 'no oracle on 1 of %d' % world in t['why']
    # catchment axis: five catchments over the ranks, up to four of each rank's checked
    cat = sdist.ShardedEnsemble(params[:20], lambda c: forcing[c], areas, dt, W, gap, axis='catchments', n_catchments=C,
                                obs=np.tile(obs[1], (C, 1)), gw_obs=bench.GW_OBS, extra=bench.EXTRA, device='cpu')
    cat.step()
    t = bench.timed_launch_parity(so, None, cat, 'catchments', params[:20], lambda c: forcing[c],
                                  lambda c: float(areas[c]), obs[1], W, gap, dt)['timed_launch']
    ok = ok and t['ok'] and t['ranks'] == world and t['rows'] >= 12 * world and t['max_rel_objfn'] <= 1e-12
    open(os.path.join(out_dir, 'timed_%d.txt' % rank), 'w').write('ok' if ok else 'FAILED %r' % (t,))
    dist.destroy_process_group()


def test_every_rank_checks_rows_of_its_own_timed_launch(tmp_path):
    """bench.timed_launch_parity over two ranks, with the launch replaced by the oracle: how many ranks and rows took
    part, the largest difference over the ranks (a deviation planted on one rank shows on the line of all), a rank
    without a checker, the catchment axis."""
    mp.spawn(_worker_timed_parity, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert [open(tmp_path / ('timed_%d.txt' % r)).read() for r in range(2)] == ['ok', 'ok']
