"""A test double for netCDF4.Dataset: just the calls smartpy_amd makes, backed by a pickle file.

netCDF4 is not installed in this image, so the NetCDF branches of smartpy_amd (sampling database, flow files, forcing
readers) could not run at all.  This double lets the tests drive that code -- schema, dtypes, slicing, attribute names,
the zlib rewrite -- against an in-memory model of a NetCDF-4 file.  It checks what a real file would: a variable's
dimensions must exist, shapes must fit, values are cast to the variable's dtype on assignment (float32 variables round).
It is test infrastructure only; the product never imports it.
"""
import os
import pickle

import numpy as np


class _Dimension(object):
    def __init__(self, size):
        self.size = size

    def __len__(self):
        return self.size


class _Variable(object):
    def __init__(self, name, datatype, dimensions, sizes, zlib=False, complevel=None):
        self.__dict__['_attrs'] = {}
        self.__dict__['name'] = name
        self.__dict__['datatype'] = np.dtype(datatype)
        self.__dict__['dimensions'] = tuple(dimensions)
        self.__dict__['filters'] = {'zlib': bool(zlib), 'complevel': complevel}
        fill = np.nan if np.dtype(datatype).kind == 'f' else 0
        self.__dict__['data'] = np.full([sizes[d] for d in dimensions], fill, dtype=datatype)

    def __setattr__(self, key, value):          # e.g. var.units = '...'
        self._attrs[key] = value

    def __getattr__(self, key):
        try:
            return self.__dict__['_attrs'][key]
        except KeyError:
            raise AttributeError(key)

    def __getitem__(self, index):
        return self.data[index]

    def __setitem__(self, index, value):
        self.data[index] = np.asarray(value, dtype=self.datatype)     # shape mismatches raise, like the library

    def __len__(self):
        return len(self.data)


class FakeDataset(object):
    """Dataset(path, 'w', format='NETCDF4', parallel=False) / Dataset(path, 'r'), usable as a context manager."""
    opened = []          # (path, mode, parallel): what the code under test asked for

    def __init__(self, path, mode='r', format='NETCDF4', parallel=False):
        self.__dict__['_path'], self.__dict__['_mode'] = path, mode
        self.__dict__['_attrs'] = {}
        self.__dict__['dimensions'], self.__dict__['variables'] = {}, {}
        FakeDataset.opened.append((path, mode, parallel))
        if mode == 'r':
            if not os.path.exists(path):
                raise IOError('No such file: ' + path)
            with open(path, 'rb') as f:
                attrs, dims, variables = pickle.load(f)
            self._attrs.update(attrs)
            self.dimensions.update(dims)
            self.variables.update(variables)
        elif mode != 'w':
            raise ValueError(mode)
        else:
            open(path, 'wb').close()

    def __setattr__(self, key, value):          # e.g. nc.description = '...'
        self._attrs[key] = value

    def __getattr__(self, key):
        try:
            return self.__dict__['_attrs'][key]
        except KeyError:
            raise AttributeError(key)

    def set_auto_mask(self, flag):
        pass

    def createDimension(self, name, size):
        self.dimensions[name] = _Dimension(size)

    def createVariable(self, name, datatype, dimensions=(), zlib=False, complevel=None):
        sizes = {d: len(self.dimensions[d]) for d in dimensions}       # KeyError: dimension not defined
        var = _Variable(name, datatype, dimensions, sizes, zlib, complevel)
        self.variables[name] = var
        return var

    def close(self):
        if self._mode == 'w':
            with open(self._path, 'wb') as f:
                pickle.dump((self._attrs, self.dimensions, self.variables), f)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False
