"""Base class of the Monte-Carlo workflows (counterpart of smartpy/montecarlo/montecarlo.py).

The reference hands this object to spotpy's `mc` sampler, which calls parameters() / simulation() /
objectivefunction() / save() once per sample (montecarlo.py:153-154), optionally farmed out over MPI.  Here
`run()` evaluates the whole sample in one launch per GPU -- time loop and objective functions fused
(SMART.simulate_ensemble) -- shards the rows over the ranks of torch.distributed when the script was started with
one process per GPU, gathers the [N, 9] result block with one RCCL all-gather and lets rank 0 write the sampling
database (database.py) in the reference's format.  The per-sample protocol methods are kept, with the reference's
signatures, for code written against them.

After run() the results also stay where they were computed: `device_obj_fns` / `device_gw` / `device_sample` are
the gathered matrices as device tensors, which GLUE / Best take directly (`sampling=<this object>`) instead of
re-reading the database file.
"""
from os import sep

import numpy as np

from ..smart import SMART
from ..inout import get_dict_simulation_settings
from ..objfunctions import groundwater_constraint
from .. import distributed as sdist
from ..engine import SmartEngineError
from .database import database_for

OBJ_FN_NAMES = ['NSE', 'KGE', 'KGEc', 'KGEa', 'KGEb', 'PBias', 'RMSE']      # montecarlo.py:71-74; 'GW' is appended
                                                                             # when the settings hold a gw_constraint


class ParameterList(object):
    """Stand-in for spotpy.parameter.List (lhs.py:120-131): a named column of the sample matrix, consumed in
    order."""

    def __init__(self, name, values):
        self.name = name
        self.values = np.asarray(values)
        self._next = 0

    def __call__(self):
        v = self.values[self._next % len(self.values)]
        self._next += 1
        return v


class MonteCarlo(object):
    def __init__(self, catchment, root_f, in_format, out_format,
                 parallel, save_sim, func, settings_filename):
        # the settings file names the period, the time steps, the areas and (optionally) the groundwater constraint
        folder = sep.join([root_f, 'in', catchment, sep])
        settings = get_dict_simulation_settings(folder + (settings_filename or catchment + '.sttngs'))
        area, gauged_area, start, end, delta_simu, delta_report, warm_up_days, gw_constraint = settings
        self.model = SMART(catchment, area, start, end, delta_simu, delta_report, warm_up_days,
                           in_format, out_format, root_f, gauged_area)

        # 'seq' | 'mpi' in the reference; here the rows are sharded over the ranks of torch.distributed whenever it is
        # initialised, and `p` only keeps its meaning for the NetCDF files (opened in parallel mode, :93)
        self.parallel, self.p = parallel, parallel == 'mpi'
        self.save_sim = save_sim
        self.constraints = {'gw': gw_constraint}
        self.param_names = self.model.parameters.names
        self.obj_fn_names = OBJ_FN_NAMES + (['GW'] if gw_constraint else [])
        self.out_format = out_format
        self.db_file = '{}{}.SMART.{}{}'.format(self.model.out_f, catchment, func,
                                                '.nc' if out_format == 'netcdf' else '')
        self.database = None
        self.math_mode = 'fast'

        self._sample = None         # [N, 10] float64 on the host: the rows to simulate, set by the subclasses
        self._device_sample = None  # the same matrix when it was drawn / selected on the device
        self._p_map = None
        self.params = None
        #: results of the last run(): objective functions [N, 7|8] and groundwater ratios [N] (float64, host) ...
        self.obj_fns = None
        self.gw_contributions = None
        #: ... and the same on the device, next to the parameter rows they belong to (second stages take these)
        self.device_obj_fns = None
        self.device_gw = None

        self.model.write_output_files(which='observed', parallel=self.p)

    # ---- the sample ------------------------------------------------------------------------------------
    def _set_sample(self, matrix):
        """matrix: [N, 10] numpy array, or a device tensor (sampled / selected on the GPU: it stays there for the
        launch; the host copy is what the database and the per-sample protocol read)."""
        self._device_sample = None
        if type(matrix).__module__.startswith('torch'):
            self._device_sample = matrix.to(dtype=_f64()).contiguous()
            matrix = self._device_sample.cpu().numpy()
        self._sample = np.ascontiguousarray(matrix, dtype=np.float64)
        self._p_map = None
        self.params = [ParameterList(name, self._sample[:, j]) for j, name in enumerate(self.param_names)]

    @property
    def device_sample(self):
        return self._device_sample

    @property
    def p_map(self):
        """parameter tuple -> row index (lhs.py:117); built on first use (a dict of 1e6 tuples is not free)."""
        if self._p_map is None and self._sample is not None:
            self._p_map = {tuple(self._sample[r, :].tolist()): r for r in range(self._sample.shape[0])}
        return self._p_map

    # ---- run -------------------------------------------------------------------------------------------
    def _open_database(self):
        db = database_for(self.out_format, self.db_file, self.obj_fn_names, self.param_names)
        self.database = db.create(self._sample.shape[0],
                                  self.model.timeseries_report[1:] if self.save_sim else None, parallel=self.p)
        return db

    def _finish_database(self, db, compression):
        db.close()
        self.database = None
        if self.out_format in ('netcdf', 'csv'):
            db.compress(compression)

    def run(self, compression=None):
        """Simulate every row of the sample and write the sampling database (montecarlo.py:132-177)."""
        import torch
        rank, world = sdist.rank_world()
        if world > 1:
            # the sampler draws from NumPy's unseeded global stream (lhs.py:149,154): every process holds another
            # matrix unless the caller seeded them alike.  Rank 0's is THE sample, as in the reference's MPI mode,
            # where only the master draws.
            theirs = sdist.broadcast_matrix(self._sample, src=0)
            if theirs.shape != self._sample.shape or not np.array_equal(theirs, self._sample):
                self._set_sample(theirs)
        n = self._sample.shape[0]
        n_obj = len(self.obj_fn_names)
        if n == 0:      # e.g. GLUE with no behavioural set: the reference's sampler loops zero times, header only
            self.obj_fns, self.gw_contributions = np.empty((0, n_obj)), np.empty(0)
            self.device_obj_fns = self.device_gw = None
            if rank == 0:
                self._finish_database(self._open_database(), compression)
            sdist.barrier()
            return
        lo, hi = sdist.shard_bounds(n, world, rank)
        rows = self._device_sample if self._device_sample is not None else self._sample
        # one rank's launch failing -- a status word its repeated launch could not clear, a HIP out-of-memory at
        # save_sim=True, any other exception (round 5 caught SmartEngineError only: advisor) -- must not leave its peers
        # waiting in the gather below: the outcome is agreed over the ranks first, and every rank raises if any did (a
        # failing MPI worker ends the reference's run as well, montecarlo.py:153-154 -- here promptly, and on every rank)
        failure, out, block, mine = None, None, None, None
        try:
            out = self.model.simulate_ensemble(rows[lo:hi] if hi > lo else rows[:1],
                                               objective_functions=True, gw_constraint=self.constraints['gw'],
                                               save_discharge=self.save_sim, math_mode=self.math_mode)
            block = torch.cat([out.objfn[:hi - lo, :n_obj], out.gw[:hi - lo].unsqueeze(1)], dim=1)
            if self.save_sim:       # (the float32 the database keeps, montecarlo.py:225: a device copy that can fail too)
                mine = out.discharge[:hi - lo].to(torch.float32)
        except Exception as e:      # noqa: BLE001 -- re-raised below, on this rank as it is
            failure = e
        if world > 1:
            sdist.agree_or_raise(failure)
        elif failure is not None:
            raise failure
        # the [N, 9] block of objective functions and groundwater ratios goes to every rank (72 bytes per sample: second
        # stages select from it on any rank) ...
        gathered = sdist.gather_rows(block, n)
        self.device_obj_fns, self.device_gw = gathered[:, :n_obj], gathered[:, n_obj]
        host = gathered.cpu().numpy()
        self.obj_fns, self.gw_contributions = host[:, :n_obj], host[:, n_obj]
        # ... the simulated series, when they are to be saved, to the rank that writes the database ONLY, and as the
        # float32 the database keeps (montecarlo.py:225; the reference's workers send theirs to the master alone:
        # :211-231).  Never all-gathered: [N, R] is 29 GB of fp64 at N = 1e6.
        series = None
        if self.save_sim:
            series = sdist.collect_rows(mine, n, dst=0)
        failure = None
        if rank == 0:
            try:
                db = self._open_database()
                db.write_table(self.obj_fns, self._sample, series)
                self._finish_database(db, compression)
            except Exception as e:      # noqa: BLE001 -- a full disk on rank 0 ends the call on every rank
                failure = e
        if world > 1:
            sdist.agree_or_raise(failure)       # (every rank has got here: the barrier of the call, with an outcome)
        elif failure is not None:
            raise failure

    # ---- the per-sample protocol of the reference (spotpy setup class) ------------------------------------
    def parameters(self):
        return np.array([p() for p in self.params])

    def simulation(self, vector):
        """montecarlo.py:179-186."""
        discharge, groundwater_component = self.model.simulate(dict(zip(self.param_names, vector)))
        return discharge, [groundwater_component]

    def evaluation(self):
        return self.model.nd_flow, [self.constraints['gw']]

    def objectivefunction(self, simulation, evaluation):
        """montecarlo.py:193-209 for one sample, on the GPU (two-pass kernel over an [R, 1] matrix)."""
        from ..engine import objective_functions
        sim = np.ascontiguousarray(np.asarray(simulation[0], dtype=np.float64)[:, None])
        o = objective_functions(sim, evaluation[0]).cpu().numpy()[0, :7].tolist()
        if self.constraints['gw']:
            o.append(groundwater_constraint(evaluation=evaluation[1], simulation=simulation[1]))
        return o

    def _init_db(self):
        """Open the database for the per-sample protocol (save() below); run() does this itself."""
        self._open_database()

    def save(self, obj_fns, parameters, simulations, *args, **kwargs):
        """One row into the open database (montecarlo.py:211-231); NetCDF rows land at the sample's own index."""
        params = np.asarray(parameters).tolist()
        index = self.p_map[tuple(params)] if self.out_format == 'netcdf' else None
        self.database.write_sample(index, obj_fns, params, np.asarray(simulations[0]) if self.save_sim else None)

    def _get_sampled_sets_from_file(self, file_location, param_names, obj_fn_names, decompression_csv):
        """-> (params float32 [N, 10], obj_fns float32 [N, k]) of a previous run's database (montecarlo.py:233-262)."""
        return database_for(self.out_format, file_location, obj_fn_names, param_names).read(decompression_csv)


def _f64():
    import torch
    return torch.float64
