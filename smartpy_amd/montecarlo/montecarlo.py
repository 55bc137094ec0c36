"""Base class of the Monte-Carlo workflows (counterpart of smartpy/montecarlo/montecarlo.py).

The reference hands this object to spotpy's `mc` sampler, which calls parameters() / simulation() /
objectivefunction() / save() once per sample (montecarlo.py:153-154), optionally farmed out over MPI.  Here
`run()` evaluates the whole sample in one launch per GPU -- time loop and objective functions fused
(SMART.simulate_ensemble) -- shards the rows over the ranks of torch.distributed when the script was started with
one process per GPU, gathers the [N, 8] objective matrix with one RCCL all-gather and lets rank 0 write the
sampling database in the reference's format (float32 values printed '%.6e'; NetCDF schema of :91-118).
The per-sample protocol methods are kept, with the reference's signatures, for code written against them.
"""
import gzip
import shutil
from io import open
from os import sep, remove, rename

import numpy as np

try:
    from netCDF4 import Dataset
except ImportError:
    Dataset = None

from ..smart import SMART
from ..inout import get_dict_simulation_settings
from ..objfunctions import groundwater_constraint
from ..version import __version__
from .. import distributed as sdist

_NO_NETCDF = "The use of 'netcdf' as the output file format requires the package 'netCDF4', " \
             "please install it and retry, or choose another file format."


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
        in_f = sep.join([root_f, 'in', catchment, sep])
        settings = ''.join([in_f, settings_filename]) if settings_filename else \
            ''.join([in_f, catchment, '.sttngs'])
        c_area, g_area, start, end, delta_simu, delta_report, warm_up, gw_constraint = \
            get_dict_simulation_settings(settings)

        self.model = SMART(catchment, c_area, start, end, delta_simu, delta_report, warm_up,
                           in_format, out_format, root_f, g_area)

        self.parallel = parallel            # 'seq' | 'mpi' in the reference; here any value shards over the
        self.p = parallel == 'mpi'          # ranks of torch.distributed when it is initialised
        self.save_sim = save_sim
        self.constraints = {'gw': gw_constraint}
        self.param_names = self.model.parameters.names
        self.obj_fn_names = \
            ['NSE', 'KGE', 'KGEc', 'KGEa', 'KGEb', 'PBias', 'RMSE', 'GW'] \
            if self.constraints['gw'] else \
            ['NSE', 'KGE', 'KGEc', 'KGEa', 'KGEb', 'PBias', 'RMSE']

        self._sample = None                 # [N, 10] float64: the rows to simulate, set by the subclasses
        self._p_map = None
        self.params = None

        self.out_format = out_format
        self.db_file = \
            self.model.out_f + '{}.SMART.{}.nc'.format(catchment, func) if self.out_format == 'netcdf' else \
            self.model.out_f + '{}.SMART.{}'.format(catchment, func)
        self.database = None
        self.math_mode = 'fast'
        #: results of the last run(): objective functions [N, 7|8] and groundwater ratios [N] (float64, host)
        self.obj_fns = None
        self.gw_contributions = None

        self.model.write_output_files(which='observed', parallel=self.p)

    # ---- the sample ------------------------------------------------------------------------------------
    def _set_sample(self, matrix):
        self._sample = np.ascontiguousarray(matrix, dtype=np.float64)
        self._p_map = None
        self.params = [ParameterList(name, self._sample[:, j]) for j, name in enumerate(self.param_names)]

    @property
    def p_map(self):
        """parameter tuple -> row index (lhs.py:117); built on first use (a dict of 1e6 tuples is not free)."""
        if self._p_map is None and self._sample is not None:
            self._p_map = {tuple(self._sample[r, :].tolist()): r for r in range(self._sample.shape[0])}
        return self._p_map

    # ---- database --------------------------------------------------------------------------------------
    def _simu_stamps(self):
        return self.model.timeseries_report[1:]

    def _init_db(self):
        """montecarlo.py:90-127."""
        n = self._sample.shape[0]
        if self.out_format == 'netcdf':
            if not Dataset:
                raise Exception(_NO_NETCDF)
            self.database = Dataset(self.db_file, 'w', format='NETCDF4', parallel=self.p)
            self.database.description = "Monte Carlo Simulation outputs with SMARTpy v{}.".format(__version__)
            self.database.createDimension('NbSamples', n)
            self.database.createDimension('NbParameters', len(self.param_names))
            self.database.createDimension('NbObjFunctions', len(self.obj_fn_names))
            params = self.database.createVariable('Parameters', np.float32, ('NbSamples', 'NbParameters'))
            params.units = ', '.join(self.param_names)
            objfns = self.database.createVariable('ObjFunctions', np.float32, ('NbSamples', 'NbObjFunctions'))
            objfns.units = ', '.join(self.obj_fn_names)
            if self.save_sim:
                stamps = self._simu_stamps()
                self.database.createDimension('DateTime', len(stamps))
                times = self.database.createVariable('DateTime', np.float64, ('DateTime',))
                times.units = "seconds since 1970-01-01 00:00:00.0"
                simu = self.database.createVariable('Simulations', np.float32, ('NbSamples', 'DateTime'))
                simu.units = "Discharge in m3/s"
                self.database.variables['DateTime'][0:len(stamps)] = \
                    (np.asarray(stamps, dtype='datetime64[us]') - np.datetime64('1970-01-01T00:00:00')) / \
                    np.timedelta64(1, 's')
        else:
            self.database = open(self.db_file, 'w', newline='', encoding='utf8')
            simu_steps = [dt.strftime('%Y-%m-%d %H:%M:%S') for dt in self._simu_stamps()] if self.save_sim else []
            self.database.write(','.join(self.obj_fn_names + self.param_names + simu_steps) + '\n')

    def _write_rows(self, obj_fns, params, sims):
        """Bulk form of save() (montecarlo.py:211-231): everything cast to float32, CSV values as '%.6e'."""
        if self.out_format == 'netcdf':
            self.database.variables['Parameters'][:, 0:len(self.param_names)] = params
            self.database.variables['ObjFunctions'][:, 0:len(self.obj_fn_names)] = obj_fns
            if self.save_sim:
                self.database.variables['Simulations'][:, 0:sims.shape[1]] = sims
        else:
            cols = [obj_fns, params] + ([sims] if self.save_sim else [])
            table = np.ascontiguousarray(np.concatenate([np.asarray(c, dtype=np.float32) for c in cols], axis=1))
            # the rows are formatted by the library (smart_db_append_rows: same characters as '%.6e' % float32,
            # ~15x faster than numpy.savetxt, which took 30x the GPU run at 1e5 samples)
            import ctypes
            from .. import _lib
            self.database.flush()
            _lib.check(_lib.lib().smart_db_append_rows(
                self.db_file.encode('utf8'), table.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                table.shape[0], table.shape[1], 0))
            self.database.seek(0, 2)

    # ---- run -------------------------------------------------------------------------------------------
    def run(self, compression=None):
        """Simulate every row of the sample and write the sampling database (montecarlo.py:132-177)."""
        n = self._sample.shape[0]
        rank, world = sdist.rank_world()
        n_obj = len(self.obj_fn_names)
        if n == 0:      # e.g. GLUE with no behavioural set: the reference's sampler loops zero times, header only
            self.obj_fns, self.gw_contributions = np.empty((0, n_obj)), np.empty(0)
            if rank == 0:
                self._init_db()
                self.database.close()
                self._compress(compression)
            sdist.barrier()
            return
        lo, hi = sdist.shard_bounds(n, world, rank)
        out = self.model.simulate_ensemble(self._sample[lo:hi] if hi > lo else self._sample[:1],
                                           objective_functions=True, gw_constraint=self.constraints['gw'],
                                           save_discharge=self.save_sim, math_mode=self.math_mode)
        local = out.objfn[:hi - lo, :n_obj]
        import torch
        packed = torch.cat([local, out.gw[:hi - lo].unsqueeze(1)], dim=1)
        if self.save_sim:       # float32 is all the database keeps (montecarlo.py:225)
            packed = torch.cat([packed, out.discharge[:hi - lo].to(torch.float64)], dim=1)
        packed = sdist.gather_rows(packed, n)
        host = packed.cpu().numpy()
        self.obj_fns = host[:, :n_obj]
        self.gw_contributions = host[:, n_obj]
        sims = host[:, n_obj + 1:] if self.save_sim else None
        if rank == 0:
            self._init_db()
            self._write_rows(self.obj_fns, self._sample, sims)
            self.database.close()
            self._compress(compression)
        sdist.barrier()

    def _compress(self, compression):
        """montecarlo.py:157-177."""
        if self.out_format == 'netcdf':
            if compression is True:
                compression = 6
            if not isinstance(compression, bool) and isinstance(compression, (int, float)):
                with Dataset(self.db_file, 'r') as src, Dataset(self.db_file.replace('.nc', '_.nc'), 'w') as dst:
                    dst.description = src.description
                    for name, dimension in src.dimensions.items():
                        dst.createDimension(name, len(dimension))
                    for name, variable in src.variables.items():
                        v = dst.createVariable(name, variable.datatype, variable.dimensions,
                                               zlib=True, complevel=compression)
                        v.units = src.variables[name].units
                        dst.variables[name][:] = src.variables[name][:]
                remove(self.db_file)
                rename(self.db_file.replace('.nc', '_.nc'), self.db_file)
        elif self.out_format == 'csv':
            if compression is True:
                with open(self.db_file, 'rb') as f_in:
                    with gzip.open(self.db_file + '.gz', 'wb') as f_out:
                        shutil.copyfileobj(f_in, f_out)
                remove(self.db_file)

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

    def save(self, obj_fns, parameters, simulations, *args, **kwargs):
        """One row into an open database (montecarlo.py:211-231)."""
        params = np.asarray(parameters).tolist()
        if self.out_format == 'netcdf':
            index = self.p_map[tuple(params)]
            self.database.variables['Parameters'][index, 0:len(self.param_names)] = params
            self.database.variables['ObjFunctions'][index, 0:len(self.obj_fn_names)] = obj_fns
            if self.save_sim:
                self.database.variables['Simulations'][index, 0:len(simulations[0])] = simulations[0]
        else:
            row = list(obj_fns) + params + (np.asarray(simulations[0]).tolist() if self.save_sim else [])
            self.database.write(','.join('%.6e' % np.float32(x) for x in row) + '\n')

    def _get_sampled_sets_from_file(self, file_location, param_names, obj_fn_names, decompression_csv):
        """-> (params float32 [N, 10], obj_fns float32 [N, k]) (montecarlo.py:233-262)."""
        if self.out_format == 'netcdf':
            if not Dataset:
                raise Exception(_NO_NETCDF)
            with Dataset(file_location, 'r') as f:
                return (np.array(f.variables['Parameters'][:, :], dtype=np.float32),
                        np.array(f.variables['ObjFunctions'][:, :], dtype=np.float32))
        opener = (lambda: gzip.open(file_location + '.gz', 'rb')) if decompression_csv else \
            (lambda: open(file_location, 'rb'))
        # columns are looked up by header name like the reference's DictReader; the table itself is parsed in bulk by
        # the library (smart_db_parse_rows: text -> float64 -> float32, as np.array(str) does; a 1e6-row database
        # takes a fraction of a second instead of minutes)
        import ctypes
        from .. import _lib
        with opener() as f:
            header = f.readline().decode('utf8').rstrip('\r\n').split(',')
            try:
                cols = [header.index(name) for name in list(param_names) + list(obj_fn_names)]
            except ValueError as e:
                raise KeyError(str(e))
            body = f.read()
        max_rows = body.count(b'\n') + 1
        table = np.empty((max_rows, len(cols)), dtype=np.float32)
        idx = np.asarray(cols, dtype=np.int32)
        n = _lib.lib().smart_db_parse_rows(body, len(body), len(header), idx.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
                                           len(cols), table.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), max_rows, 0)
        if n < 0:
            _lib.check(int(n))
        table = table[:n]
        return np.ascontiguousarray(table[:, :len(param_names)]), np.ascontiguousarray(table[:, len(param_names):])
