"""The sampling database of a Monte-Carlo run: the file LHS writes and GLUE / Best / Total read back.

Two flavours with one interface, selected by the `out_format` of the run:

  SamplingCsv     `<out>/<catchment>.SMART.<func>`      header = objective-function names + parameter names
                  (+ one '%Y-%m-%d %H:%M:%S' stamp per report step when simulations are saved), then one line per
                  sample, every value cast to float32 and printed '%.6e' (montecarlo.py:122-127, 225-231); optional
                  gzip of the whole file afterwards (:170-177).
  SamplingNetcdf  `<out>/<catchment>.SMART.<func>.nc`   dimensions NbSamples / NbParameters / NbObjFunctions
                  (/ DateTime), float32 variables Parameters, ObjFunctions (, Simulations), float64 DateTime in
                  seconds since the epoch (:91-118); optional rewrite with zlib (:158-169).  Needs netCDF4.

What is observable by the second-stage tools and by users -- file names, header, column order, float32 rounding,
number formatting, variable names -- is the reference's; how the rows get there is not: the reference appends one
sample at a time from spotpy's loop, here the whole table is formatted by the library in one call
(smart_db_append_rows) and read back the same way (smart_db_parse_rows).
"""
import ctypes
import gzip
import os
import shutil

import numpy as np

try:
    from netCDF4 import Dataset
except ImportError:
    Dataset = None

from .. import _lib
from ..version import __version__

NO_NETCDF = "The use of 'netcdf' as the output file format requires the package 'netCDF4', " \
            "please install it and retry, or choose another file format."


def database_for(out_format, path, obj_fn_names, param_names):
    """The flavour that goes with an output format ('netcdf' -> NetCDF, anything else -> text)."""
    kind = SamplingNetcdf if out_format == 'netcdf' else SamplingCsv
    return kind(path, list(obj_fn_names), list(param_names))


class SamplingCsv(object):
    def __init__(self, path, obj_fn_names, param_names):
        self.path, self.obj_fn_names, self.param_names = path, obj_fn_names, param_names
        self.handle = None
        self._with_series = False

    # ---- writing
    def create(self, n_samples, report_stamps=None, parallel=False):
        """Start the file: the header line.  report_stamps: datetimes of the saved series, or None."""
        self._with_series = report_stamps is not None
        columns = self.obj_fn_names + self.param_names
        if self._with_series:
            columns = columns + [stamp.strftime('%Y-%m-%d %H:%M:%S') for stamp in report_stamps]
        self.handle = open(self.path, 'w', newline='', encoding='utf8')
        self.handle.write(','.join(columns) + '\n')
        return self

    def write_table(self, obj_fns, params, series=None):
        """All samples at once.  The three blocks are cast to float32 and laid side by side; the library formats
        the lines (characters identical to '%.6e' % numpy.float32(x); numpy.savetxt took 25x the GPU run)."""
        blocks = [obj_fns, params] + ([series] if self._with_series else [])
        table = np.ascontiguousarray(np.concatenate([np.asarray(b, dtype=np.float32) for b in blocks], axis=1))
        self.handle.flush()
        _lib.check(_lib.lib().smart_db_append_rows(
            self.path.encode('utf8'), table.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
            table.shape[0], table.shape[1], 0))
        self.handle.seek(0, os.SEEK_END)

    def write_sample(self, index, obj_fns, params, series=None):
        """One sample (the per-sample protocol of the reference, MonteCarlo.save)."""
        values = list(obj_fns) + list(params) + (list(series) if self._with_series else [])
        self.handle.write(','.join('%.6e' % np.float32(v) for v in values) + '\n')

    def close(self):
        if self.handle is not None:
            self.handle.close()
            self.handle = None

    def compress(self, compression):
        """compression=True gzips the file and removes the plain one; anything else leaves it alone."""
        if compression is True:
            with open(self.path, 'rb') as plain, gzip.open(self.path + '.gz', 'wb') as packed:
                shutil.copyfileobj(plain, packed)
            os.remove(self.path)

    # ---- reading
    def read(self, gzipped=False):
        """-> (params float32 [N, n_par], obj_fns float32 [N, n_obj]).  Columns are found by header name (so files of
        older releases with more objective functions still read); the table is parsed in bulk by the library, text ->
        correctly rounded double -> float32, which is what numpy.array(list_of_strings, dtype=float32) does."""
        opener = gzip.open(self.path + '.gz', 'rb') if gzipped else open(self.path, 'rb')
        with opener as f:
            header = f.readline().decode('utf8').rstrip('\r\n').split(',')
            body = f.read()
        wanted = self.param_names + self.obj_fn_names
        missing = [name for name in wanted if name not in header]
        if missing:
            raise KeyError(missing[0])
        columns = np.asarray([header.index(name) for name in wanted], dtype=np.int32)
        room = body.count(b'\n') + 1
        table = np.empty((room, len(wanted)), dtype=np.float32)
        n = _lib.lib().smart_db_parse_rows(body, len(body), len(header),
                                           columns.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), len(wanted),
                                           table.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), room, 0)
        if n < 0:
            _lib.check(int(n))
        k = len(self.param_names)
        return np.ascontiguousarray(table[:n, :k]), np.ascontiguousarray(table[:n, k:])


# NetCDF layout: (variable, dtype, dimensions, units); DateTime / Simulations only when the series are saved
_EPOCH = np.datetime64('1970-01-01T00:00:00')


class SamplingNetcdf(object):
    def __init__(self, path, obj_fn_names, param_names):
        self.path, self.obj_fn_names, self.param_names = path, obj_fn_names, param_names
        self.handle = None
        self._with_series = False

    def _layout(self):
        yield 'Parameters', np.float32, ('NbSamples', 'NbParameters'), ', '.join(self.param_names)
        yield 'ObjFunctions', np.float32, ('NbSamples', 'NbObjFunctions'), ', '.join(self.obj_fn_names)
        if self._with_series:
            yield 'DateTime', np.float64, ('DateTime',), "seconds since 1970-01-01 00:00:00.0"
            yield 'Simulations', np.float32, ('NbSamples', 'DateTime'), "Discharge in m3/s"

    def create(self, n_samples, report_stamps=None, parallel=False):
        if Dataset is None:
            raise Exception(NO_NETCDF)
        self._with_series = report_stamps is not None
        sizes = {'NbSamples': n_samples, 'NbParameters': len(self.param_names),
                 'NbObjFunctions': len(self.obj_fn_names)}
        if self._with_series:
            sizes['DateTime'] = len(report_stamps)
        nc = self.handle = Dataset(self.path, 'w', format='NETCDF4', parallel=parallel)
        nc.description = "Monte Carlo Simulation outputs with SMARTpy v{}.".format(__version__)
        for name, size in sizes.items():
            nc.createDimension(name, size)
        for name, dtype, dims, units in self._layout():
            nc.createVariable(name, dtype, dims).units = units
        if self._with_series:
            seconds = (np.asarray(report_stamps, dtype='datetime64[us]') - _EPOCH) / np.timedelta64(1, 's')
            nc.variables['DateTime'][0:len(report_stamps)] = seconds
        return self

    def write_table(self, obj_fns, params, series=None):
        self._put(slice(None), obj_fns, params, series)

    def write_sample(self, index, obj_fns, params, series=None):
        self._put(index, obj_fns, params, series)

    def _put(self, where, obj_fns, params, series):
        v = self.handle.variables
        v['Parameters'][where, 0:len(self.param_names)] = params
        v['ObjFunctions'][where, 0:len(self.obj_fn_names)] = obj_fns
        if self._with_series:
            v['Simulations'][where, 0:np.shape(series)[-1]] = series

    def close(self):
        if self.handle is not None:
            self.handle.close()
            self.handle = None

    def compress(self, compression):
        """compression = True (level 6) or a zlib level: the file is rewritten variable by variable with zlib."""
        level = 6 if compression is True else compression
        if isinstance(level, bool) or not isinstance(level, (int, float)):
            return
        packed_path = self.path.replace('.nc', '_.nc')
        with Dataset(self.path, 'r') as plain, Dataset(packed_path, 'w') as packed:
            packed.description = plain.description
            for name, dim in plain.dimensions.items():
                packed.createDimension(name, len(dim))
            for name, var in plain.variables.items():
                out = packed.createVariable(name, var.datatype, var.dimensions, zlib=True, complevel=level)
                out.units = var.units
                out[:] = var[:]
        os.remove(self.path)
        os.rename(packed_path, self.path)

    def read(self, gzipped=False):
        if Dataset is None:
            raise Exception(NO_NETCDF)
        with Dataset(self.path, 'r') as nc:
            return (np.array(nc.variables['Parameters'][:, :], dtype=np.float32),
                    np.array(nc.variables['ObjFunctions'][:, :], dtype=np.float32))
